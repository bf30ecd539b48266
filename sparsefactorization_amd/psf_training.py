"""Adding / Temporal-Order training driver — counterpart of ``SyntheticExperiments/psf_training.py``.

    python -m sparsefactorization_amd.psf_training --problem order --n-vec 16384 --epochs 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
           -m sparsefactorization_amd.psf_training --problem order --n-vec 16384

Same model construction as the reference driver (psf_training.py:29-45: ``n_W = int(log2(n_vec))``, config keys
of synthetic_training_config.py), same optimiser / loss choice (50-58), ``seed_everything(42)`` (16),
``drop_last=True`` loaders (80-114). What changes: data come from the on-device generators
(``synth_data``: there are no ``.pt`` files here), the loaders are ``train.DeviceBatches`` — DataLoader semantics for
tensors that already sit on the GPU, without the per-sample collate (the
reference's ``torch_geometric.data.DataLoader`` is a subclass that adds nothing for tensor pairs), and under
``torch.distributed.run`` every rank trains on its shard of each global batch with ONE flat gradient
all-reduce per step (``dp.FlatGradAllReduce``, RCCL over xGMI).
"""
from __future__ import annotations

import argparse
import json
import time

import numpy as np
import torch
from torch import nn

from . import dp, synth_data
from .synthetic_psf import PSFNet
from .train import DeviceBatches, GraphedStep, TrainModel, count_params, make_adam, seed_everything, train_epoch

# PSF entries of SyntheticExperiments/synthetic_training_config.py (same keys and values)
config = {
    "adding": {
        "model": {"add_init_linear_layer": True, "vocab_size": 1, "dim": 32, "Ws": [32, 'GELU'], "V": [32, 'GELU'],
                  "pooling_type": "FLATTEN", "head": ['linear'], "n_class": 1, "n_channels_V": 8, "use_cuda": True,
                  "use_residuals": True, "use_pos_embedding": False, "problem": "adding"},
        "training": {"device_id": 0, "batch_size": 40, "learning_rate": 0.001, "eval_frequency": 1,
                     "num_train_steps": 20},
    },
    "order": {
        "model": {"add_init_linear_layer": False, "vocab_size": 6, "dim": 32, "Ws": [32, 'GELU'], "V": [32, 'GELU'],
                  "pooling_type": "FLATTEN", "head": ['linear'], "n_class": 4, "n_channels_V": 8, "use_cuda": True,
                  "use_residuals": True, "use_pos_embedding": True, "problem": "order"},
        "training": {"device_id": 0, "batch_size": 40, "learning_rate": 0.001, "eval_frequency": 1,
                     "num_train_steps": 20},
    },
}


def build_model(problem: str, n_vec: int, use_cuda: bool = True) -> PSFNet:
    cfg = config[problem]["model"]
    return PSFNet(vocab_size=cfg["vocab_size"], add_init_linear_layer=cfg["add_init_linear_layer"],
                  embedding_size=cfg["dim"], n_vec=n_vec, n_W=int(np.log2(n_vec)), Ws=cfg["Ws"], V=cfg["V"],
                  n_channels_V=cfg["n_channels_V"], n_class=cfg["n_class"], pooling_type=cfg["pooling_type"],
                  head=cfg["head"], use_cuda=use_cuda, use_residuals=cfg["use_residuals"],
                  use_pos_embedding=cfg["use_pos_embedding"], problem=problem)


def make_split(problem: str, n_seq: int, n_vec: int, device, seed: int):
    g = torch.Generator(device=device).manual_seed(seed)
    fn = synth_data.adding if problem == "adding" else synth_data.temporal_order
    return fn(n_seq, n_vec, device=device, generator=g)


def train_benchmark(problem: str, n_vec: int, batch: int, steps: int, warmup: int, device, rank: int = 0,
                    world: int = 1, graph: bool = False, n_batches: int = 4, group=None) -> dict:
    """Time ``steps`` optimisation steps of the reference training loop body (psf_utils.py:62-71: zero_grad, forward,
    loss, backward, [gradient all-reduce,] Adam step) on this rank's shard: ``batch`` sequences per rank per step,
    synthetic data of the task's distribution generated on the device (rank-offset seed), replicas started from rank
    0's weights. Returns per-rank numbers; the caller aggregates over ranks (bench.py --train, ``--json`` below).
    The region is bracketed by a barrier + device synchronisation on both sides when a process group exists. ``group``:
    the process group of the parameter broadcast and the gradient all-reduce (None = the default group); the barriers use
    the default group (bench.py keeps that one on gloo and hands an RCCL group in here)."""
    import torch.distributed as dist
    seed_everything(42)
    net = build_model(problem, n_vec).to(device)
    dp.broadcast_parameters(net, group=group)
    lr = config[problem]["training"]["learning_rate"]
    use_graph_dp = graph and world > 1
    optimizer = make_adam(net.parameters(), lr, capturable=graph and world == 1)
    loss = nn.MSELoss() if problem == "adding" else nn.CrossEntropyLoss()
    reducer = dp.FlatGradAllReduce(net.parameters(), group=group, timing=True) if world > 1 else None
    X, Y = make_split(problem, batch * n_batches, n_vec, device, 1000 + rank)
    batches = [(X[i * batch:(i + 1) * batch], Y[i * batch:(i + 1) * batch]) for i in range(n_batches)]
    graphed, graph_error = None, None
    if graph:
        try:
            graphed = GraphedStep(net, optimizer, loss, *batches[0], reducer=reducer)
        except Exception as exc:
            if world == 1:
                raise
            # Several ranks: a capture that fails next to a live process group (never seen: no multi-GPU box yet) must not
            # cost the measurement. Every rank has made the same number of reducer calls (the warm-up steps), so an eager
            # continuation stays in step with ranks whose capture worked.
            graph_error = repr(exc)
            torch.cuda.synchronize(device)

    def one(i):
        x, y = batches[i % n_batches]
        if graphed is not None:
            return graphed(x, y)
        optimizer.zero_grad(set_to_none=True)
        out = loss(net(x).squeeze(), y)
        out.backward()
        if reducer is not None:
            reducer()
        optimizer.step()
        return out.detach()

    def sync():
        torch.cuda.synchronize(device)
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize(device)

    for i in range(warmup):
        one(i)
    if reducer is not None:
        sync()
        reducer.reset_timing()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    sync()
    t0 = time.perf_counter()
    e0.record()
    last = None
    for i in range(steps):
        last = one(i)
    e1.record()
    sync()
    dt = time.perf_counter() - t0
    return {"seconds": dt, "event_ms": e0.elapsed_time(e1), "steps": steps, "loss": float(last),
            "allreduce_us": reducer.mean_us() if reducer is not None else None,
            "grad_bytes": 4 * sum(p.numel() for p in net.parameters() if p.requires_grad),
            "hip_graph": (("fwd+bwd" if use_graph_dp else "step") if graphed is not None else f"eager after: {graph_error}")
            if graph else None}


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--problem", choices=["adding", "order"], default="adding")
    ap.add_argument("--n-vec", type=int, default=128)
    ap.add_argument("--epochs", type=int, default=None)
    ap.add_argument("--train-seqs", type=int, default=4000)
    ap.add_argument("--eval-seqs", type=int, default=400)
    ap.add_argument("--batch-size", type=int, default=None, help="per-rank batch (reference: 40)")
    ap.add_argument("--max-steps", type=int, default=None, help="stop an epoch early (benchmarking)")
    ap.add_argument("--json", action="store_true", help="print one JSON line with training throughput")
    ap.add_argument("--graph", action="store_true",
                    help="capture the training step in a HIP graph and replay it (train.GraphedStep; with several ranks the "
                         "gradient all-reduce and the optimizer step stay outside the graph)")
    ap.add_argument("--force-allreduce", action="store_true",
                    help="run the flat gradient all-reduce even in a 1-rank group (RCCL path check on a 1-GPU box)")
    args = ap.parse_args(argv)

    rank, world, device = dp.init_from_env()
    if device.type != "cuda":
        raise SystemExit("training needs an MI355X: the chord-spmm path has no CPU implementation")
    cfg_training = config[args.problem]["training"]
    batch = args.batch_size or cfg_training["batch_size"]
    epochs = args.epochs if args.epochs is not None else cfg_training["num_train_steps"]

    seed_everything(42)
    net = build_model(args.problem, args.n_vec).to(device)
    dp.broadcast_parameters(net)
    if rank == 0:
        print('Number of trainable parameters', count_params(net))
    # --graph: one process captures the whole step (optimizer included: capturable Adam); under data parallelism the
    # graph holds forward + backward and the all-reduce + optimizer step stay eager (train.GraphedStep)
    optimizer = make_adam(net.parameters(), cfg_training["learning_rate"], capturable=args.graph and world == 1)
    loss = nn.MSELoss() if args.problem == "adding" else nn.CrossEntropyLoss()
    force = args.force_allreduce and torch.distributed.is_initialized()
    reducer = dp.FlatGradAllReduce(net.parameters(), force=force) if (world > 1 or force) else None

    # every rank draws its own shard of the global data (rank-offset seed), validation/test are replicated
    lo, hi = dp.shard_bounds(args.train_seqs, rank, world)
    Xtr, Ytr = make_split(args.problem, hi - lo, args.n_vec, device, 1000 + rank)
    Xva, Yva = make_split(args.problem, args.eval_seqs, args.n_vec, device, 2000)
    Xte, Yte = make_split(args.problem, args.eval_seqs, args.n_vec, device, 3000)
    mk = lambda X, Y, shuffle: DeviceBatches(X, Y, batch, shuffle=shuffle, drop_last=True)  # noqa: E731
    trainloader, valloader, testloader = mk(Xtr, Ytr, True), mk(Xva, Yva, False), mk(Xte, Yte, False)

    graphed = GraphedStep(net, optimizer, loss, *next(iter(trainloader)), reducer=reducer) if args.graph else None

    if args.json:
        train_epoch(net, trainloader, optimizer, loss, reducer, max_steps=3, graphed=graphed)  # warm-up
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        stats = train_epoch(net, trainloader, optimizer, loss, reducer, max_steps=args.max_steps, graphed=graphed)
        torch.cuda.synchronize(device)
        dt = time.perf_counter() - t0
        if rank == 0:
            print(json.dumps({"metric": "PSF train tokens/sec", "problem": args.problem, "n_vec": args.n_vec,
                              "n_gpus": world, "batch_per_gpu": batch, "steps": stats["steps"],
                              "ms_per_step": dt * 1e3 / max(stats["steps"], 1),
                              "value": world * batch * args.n_vec * stats["steps"] / dt, "unit": "tokens/s",
                              "loss": stats["loss"], "hip_graph": bool(args.graph)}))
        return {"net": net, "reducer": reducer, "stats": stats}

    history = TrainModel(net=net, trainloader=trainloader, valloader=valloader, testloader=testloader, n_epochs=epochs,
               test_freq=cfg_training["eval_frequency"], optimizer=optimizer, loss=loss, problem=args.problem,
               saving_criteria=99.5, reducer=reducer, is_main=rank == 0, graphed=graphed)
    return {"net": net, "reducer": reducer, "history": history}


if __name__ == "__main__":
    main()
