"""LRA training driver — counterpart of ``LRA/{listops,pathfinder,cifar10,imdb}_training.py`` (four scripts
that differ only in config key, file names, the CLS block and the checkpoint threshold).

    python -m sparsefactorization_amd.lra_training --task listops --epochs 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
           -m sparsefactorization_amd.lra_training --task listops --json

Model configs are the values of ``LRA/psf_training_config.py`` (2-118); the loop is ``TrainPSF``
(``LRA/psf_utils.py:48-128``); a CLS token (id ``vocab_size - 1``) is prepended when ``pooling_type == 'CLS'``
(``listops_training.py:65-72``), which is why those tasks have ``n_vec`` = sequence length + 1.
The LRA datasets are not available here (7.7 GB download + TensorFlow preprocessing): the driver
trains on synthetic token tensors of the task's shape, dtype and value range, or on ``--data-dir`` holding the
reference's ``train_clean.pt`` / ``target_train_clean.pt`` / ``val_…`` / ``test_…`` files when they exist.
"""
from __future__ import annotations

import argparse
import json
import os
import time

import torch
from torch import nn

from . import dp
from .lra_psf import PSFNet
from .train import DeviceBatches, GraphedStep, TrainPSF, count_params, make_adam, seed_everything, train_epoch

config = {
    "listops": {
        "model": {"vocab_size": 15 + 1 + 1, "embedding_size": 512, "n_vec": 1999 + 1, "n_W": 11, "Ws": [128, 'GELU'],
                  "V": [128, 'GELU'], "n_channels_V": 128, "n_class": 10, "pooling_type": "CLS", "head": ['linear'],
                  "use_cuda": True, "use_residuals": False, "dropout1_p": 0, "dropout2_p": 0, "dropout3_p": 0,
                  "init_embedding_weights": False, "use_pos_embedding": True, "problem": "listops"},
        "training": {"device_id": 0, "batch_size": 32, "learning_rate": 0.001, "eval_frequency": 1, "num_train_steps": 7},
        "saving_criteria": 38,
    },
    "cifar10": {
        "model": {"vocab_size": 256, "embedding_size": 16, "n_vec": 1024, "n_W": 10, "Ws": [16, 'GELU'],
                  "V": [16, 'GELU'], "n_channels_V": 16, "n_class": 10, "pooling_type": "FLATTEN",
                  "head": ['non-linear', 16], "use_cuda": True, "use_residuals": False, "dropout1_p": 0,
                  "dropout2_p": 0.2, "dropout3_p": 0.8, "init_embedding_weights": False, "use_pos_embedding": True,
                  "problem": "cifar10"},
        "training": {"device_id": 0, "batch_size": 32, "learning_rate": 0.001, "eval_frequency": 1, "num_train_steps": 35},
        "saving_criteria": 44,
    },
    "pathfinder": {
        "model": {"vocab_size": 225, "embedding_size": 32, "n_vec": 1024, "n_W": 11, "Ws": [128, 'GELU'],
                  "V": [128, 'GELU'], "n_channels_V": 32, "n_class": 2, "pooling_type": "FLATTEN", "head": ['linear'],
                  "use_cuda": True, "use_residuals": False, "dropout1_p": 0, "dropout2_p": 0, "dropout3_p": 0,
                  "init_embedding_weights": False, "use_pos_embedding": True, "problem": "pathfinder"},
        "training": {"device_id": 0, "batch_size": 64, "learning_rate": 0.001, "eval_frequency": 1, "num_train_steps": 45},
        "saving_criteria": 80,
    },
    "imdb": {
        "model": {"vocab_size": 95 + 1 + 1, "embedding_size": 32, "n_vec": 4096 + 1, "n_W": 12, "Ws": [128, 'GELU'],
                  "V": [128, 'GELU'], "n_channels_V": 32, "n_class": 2, "pooling_type": "CLS", "head": ['linear'],
                  "use_cuda": True, "use_residuals": True, "dropout1_p": 0.4, "dropout2_p": 0, "dropout3_p": 0,
                  "init_embedding_weights": True, "use_pos_embedding": False, "problem": "imdb"},
        "training": {"device_id": 0, "batch_size": 32, "learning_rate": 0.0001, "eval_frequency": 1, "num_train_steps": 145},
        "saving_criteria": 76,
    },
}


def build_model(task: str, use_cuda: bool = True, **overrides) -> PSFNet:
    cfg = dict(config[task]["model"], use_cuda=use_cuda)
    cfg.update(overrides)
    return PSFNet(**cfg)


def add_cls_token(data: torch.Tensor, vocab_size: int) -> torch.Tensor:
    """Prepend the CLS column (token id vocab_size - 1) — listops_training.py:65-72."""
    cls = torch.full((data.size(0), 1), vocab_size - 1, dtype=data.dtype, device=data.device)
    return torch.cat([cls, data], dim=-1)


def synthetic_split(task: str, n_seq: int, device, seed: int):
    """Token tensor [n_seq, n_vec (without CLS)] int64 in the task's value range + labels; PAD/CLS ids unused."""
    cfg = config[task]["model"]
    cls = cfg["pooling_type"] == "CLS"
    seq_len = cfg["n_vec"] - (1 if cls else 0)
    n_tokens = cfg["vocab_size"] - (2 if task in ("listops", "imdb") else 0)
    g = torch.Generator(device=device).manual_seed(seed)
    data = torch.randint(0, n_tokens, (n_seq, seq_len), device=device, generator=g)
    labels = torch.randint(0, cfg["n_class"], (n_seq,), device=device, generator=g)
    return data, labels


def load_split(task: str, split: str, data_dir: str, device):
    data = torch.load(os.path.join(data_dir, f"{split}_clean.pt")).to(torch.int64).to(device)
    labels = torch.load(os.path.join(data_dir, f"target_{split}_clean.pt")).to(torch.int64).to(device)
    return data, labels


def train_benchmark(task: str, steps: int, warmup: int, device, rank: int = 0, world: int = 1, graph: bool = False,
                    batch: int = None, n_batches: int = 4, **model_overrides) -> dict:
    """Time ``steps`` optimisation steps of the LRA loop body (LRA/psf_utils.py:60-74: zero_grad, forward, loss, backward,
    [gradient all-reduce,] Adam) on synthetic token batches of the task's shape (CLS prepended where the task pools on
    it). Per-rank numbers, as ``psf_training.train_benchmark``; ``model_overrides`` change configuration entries
    (BASELINE.json's ListOps wording: n_vec=2048, embedding_size=64, n_channels_V=64)."""
    import torch.distributed as dist
    cfg_model, cfg_training = dict(config[task]["model"], **model_overrides), config[task]["training"]
    batch = batch or cfg_training["batch_size"]
    seed_everything(42)
    net = PSFNet(**cfg_model).to(device)
    dp.broadcast_parameters(net)
    optimizer = make_adam(net.parameters(), cfg_training["learning_rate"], capturable=graph and world == 1)
    loss = nn.CrossEntropyLoss()
    reducer = dp.FlatGradAllReduce(net.parameters(), timing=True) if world > 1 else None
    cls = cfg_model["pooling_type"] == "CLS"
    g = torch.Generator(device=device).manual_seed(1000 + rank)
    n_tokens = cfg_model["vocab_size"] - (2 if task in ("listops", "imdb") else 0)
    X = torch.randint(0, n_tokens, (batch * n_batches, cfg_model["n_vec"] - (1 if cls else 0)), device=device, generator=g)
    Y = torch.randint(0, cfg_model["n_class"], (batch * n_batches,), device=device, generator=g)
    if cls:
        X = add_cls_token(X, cfg_model["vocab_size"])
    batches = [(X[i * batch:(i + 1) * batch], Y[i * batch:(i + 1) * batch]) for i in range(n_batches)]
    graphed = GraphedStep(net, optimizer, loss, *batches[0], reducer=reducer) if graph else None

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
    return {"seconds": dt, "event_ms": e0.elapsed_time(e1), "steps": steps, "loss": float(last), "batch": batch,
            "n_vec": cfg_model["n_vec"], "embedding_size": cfg_model["embedding_size"], "n_channels_V": cfg_model["n_channels_V"],
            "hip_graph": bool(graph)}


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--task", choices=sorted(config), default="listops")
    ap.add_argument("--epochs", type=int, default=None)
    ap.add_argument("--train-seqs", type=int, default=640)
    ap.add_argument("--eval-seqs", type=int, default=128)
    ap.add_argument("--batch-size", type=int, default=None)
    ap.add_argument("--data-dir", default=None, help="directory with the reference's *_clean.pt tensors")
    ap.add_argument("--max-steps", type=int, default=None)
    ap.add_argument("--json", action="store_true", help="print one JSON line with training throughput")
    ap.add_argument("--graph", action="store_true",
                    help="capture the training step in a HIP graph and replay it (train.GraphedStep; with several ranks the "
                         "gradient all-reduce and the optimizer step stay outside the graph)")
    args = ap.parse_args(argv)

    rank, world, device = dp.init_from_env()
    if device.type != "cuda":
        raise SystemExit("training needs an MI355X: the chord-spmm path has no CPU implementation")
    cfg_model, cfg_training = config[args.task]["model"], config[args.task]["training"]
    batch = args.batch_size or cfg_training["batch_size"]
    epochs = args.epochs if args.epochs is not None else cfg_training["num_train_steps"]

    seed_everything(42)
    net = build_model(args.task).to(device)
    dp.broadcast_parameters(net)
    if rank == 0:
        print('Number of trainable parameters', count_params(net))
    loss = nn.CrossEntropyLoss()
    optimizer = make_adam(net.parameters(), cfg_training['learning_rate'], capturable=args.graph and world == 1)
    reducer = dp.FlatGradAllReduce(net.parameters()) if world > 1 else None

    splits = {}
    for split, n, seed in (("train", args.train_seqs, 100 + rank), ("val", args.eval_seqs, 200), ("test", args.eval_seqs, 300)):
        if args.data_dir:
            X, Y = load_split(args.task, split, args.data_dir, device)
            if split == "train":
                lo, hi = dp.shard_bounds(X.size(0), rank, world)
                X, Y = X[lo:hi], Y[lo:hi]
        else:
            lo, hi = dp.shard_bounds(n, rank, world) if split == "train" else (0, n)
            X, Y = synthetic_split(args.task, hi - lo, device, seed)
        if cfg_model['pooling_type'] == 'CLS':
            X = add_cls_token(X, cfg_model['vocab_size'])
        splits[split] = (X, Y)
    mk = lambda s, shuffle: DeviceBatches(*splits[s], batch, shuffle=shuffle, drop_last=True)  # noqa: E731
    trainloader, valloader, testloader = mk("train", True), mk("val", False), mk("test", False)

    graphed = GraphedStep(net, optimizer, loss, *next(iter(trainloader)), reducer=reducer) if args.graph else None

    if args.json:
        train_epoch(net, trainloader, optimizer, loss, reducer, max_steps=3, graphed=graphed)
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        stats = train_epoch(net, trainloader, optimizer, loss, reducer, max_steps=args.max_steps, graphed=graphed)
        torch.cuda.synchronize(device)
        dt = time.perf_counter() - t0
        if rank == 0:
            print(json.dumps({"metric": "PSF train tokens/sec", "task": args.task, "n_vec": cfg_model["n_vec"],
                              "n_gpus": world, "batch_per_gpu": batch, "steps": stats["steps"],
                              "ms_per_step": dt * 1e3 / max(stats["steps"], 1),
                              "value": world * batch * cfg_model["n_vec"] * stats["steps"] / dt, "unit": "tokens/s",
                              "loss": stats["loss"], "data": "files" if args.data_dir else "synthetic",
                              "hip_graph": bool(args.graph)}))
        return

    TrainPSF(net=net, trainloader=trainloader, valloader=valloader, testloader=testloader, n_epochs=epochs,
             test_freq=cfg_training['eval_frequency'], optimizer=optimizer, loss=loss, problem=cfg_model['problem'],
             saving_criteria=config[args.task]["saving_criteria"], reducer=reducer, is_main=rank == 0, graphed=graphed)


if __name__ == "__main__":
    main()
