"""Genome classification driver — counterpart of ``Genome_Clf/genome_clf_training.py`` for its ``"psf"`` model.

    python -m sparsefactorization_amd.genome_training --epochs 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
           -m sparsefactorization_amd.genome_training --json

The model configuration is the ``"PSF"`` entry of ``Genome_Clf/genome_training_config.py`` (2-22: dog / donkey cDNA,
N = 16384 nucleotides, 14 factors, 32 channels, FLATTEN pooling), the training entry is 113-119 (batch 16, Adam 1e-4); the
loop is ``TrainPSF`` of ``Genome_Clf/psf_utils.py:48-151`` (``train.TrainGenomePSF``: gradient-norm clip at 1.0, ROC-AUC in
both evaluation loops); a CLS column is prepended when ``pooling_type == 'CLS'`` (genome_clf_training.py:127-134; the shipped
configuration pools by FLATTEN). The dataset (``DDcDNA16384_{train,val,test}[_targets].pt``, built by
``genome_preprocessing.py`` from a download) is not available here: the driver trains on synthetic token tensors of the
same shape, dtype and value range, or on ``--data-dir`` holding those files when they exist.
"""
from __future__ import annotations

import argparse
import json
import os
import time

import torch
from torch import nn

from . import dp
from .genome_psf import PSFNet
from .train import DeviceBatches, GraphedStep, TrainGenomePSF, count_params, make_adam, seed_everything, train_epoch

config = {
    "DDcDNA": {
        "PSF": {"name": "psf", "vocab_size": 6, "embedding_size": 32, "n_vec": 16384, "n_W": 14, "Ws": [32, 'GELU'],
                "V": [32, 'GELU'], "n_channels_V": 32, "n_class": 2, "pooling_type": "FLATTEN", "head": ['linear'],
                "use_cuda": True, "use_residuals": False, "dropout1_p": 0.2, "dropout2_p": 0, "dropout3_p": 0.8,
                "init_embedding_weights": False, "use_pos_embedding": False},
        "training": {"device_id": 0, "batch_size": 16, "learning_rate": 0.0001, "eval_frequency": 1, "num_train_steps": 100},
        "saving_criteria": 100,  # genome_clf_training.py:198
        "grad_clip_norm": 1.0,   # psf_utils.py:73
    },
}


def build_model(use_cuda: bool = True, **overrides) -> PSFNet:
    cfg = {k: v for k, v in config["DDcDNA"]["PSF"].items() if k != "name"}
    cfg.update(use_cuda=use_cuda, **overrides)
    return PSFNet(**cfg)


def add_cls_token(data: torch.Tensor, vocab_size: int) -> torch.Tensor:
    """Prepend the CLS column (token id vocab_size - 1) — genome_clf_training.py:127-134."""
    cls = torch.full((data.size(0), 1), vocab_size - 1, dtype=data.dtype, device=data.device)
    return torch.cat([cls, data], dim=-1)


def synthetic_split(n_seq: int, device, seed: int, n_vec: int = None):
    """Token tensor [n_seq, n_vec] int64 over the nucleotide alphabet (ids 0..4; the last id is the CLS token's) + binary
    labels, both uniform."""
    cfg = config["DDcDNA"]["PSF"]
    n_vec = n_vec or cfg["n_vec"] - (1 if cfg["pooling_type"] == "CLS" else 0)
    g = torch.Generator(device=device).manual_seed(seed)
    data = torch.randint(0, cfg["vocab_size"] - 1, (n_seq, n_vec), device=device, generator=g)
    labels = torch.randint(0, cfg["n_class"], (n_seq,), device=device, generator=g)
    return data, labels


def load_split(split: str, data_dir: str, device):
    """genome_clf_training.py:117-125."""
    data = torch.load(os.path.join(data_dir, f"DDcDNA16384_{split}.pt")).to(torch.int64).to(device)
    labels = torch.load(os.path.join(data_dir, f"DDcDNA16384_{split}_targets.pt")).to(torch.int64).to(device)
    return data, labels


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=None)
    ap.add_argument("--train-seqs", type=int, default=256)
    ap.add_argument("--eval-seqs", type=int, default=64)
    ap.add_argument("--batch-size", type=int, default=None)
    ap.add_argument("--n-vec", type=int, default=None, help="sequence length (default: the configuration's 16384)")
    ap.add_argument("--data-dir", default=None, help="directory with the reference's DDcDNA16384_*.pt tensors")
    ap.add_argument("--max-steps", type=int, default=None)
    ap.add_argument("--json", action="store_true", help="print one JSON line with training throughput")
    ap.add_argument("--graph", action="store_true",
                    help="capture the training step (gradient clip included) in a HIP graph and replay it; with several ranks "
                         "the gradient all-reduce, the clip and the optimizer step stay outside the graph")
    args = ap.parse_args(argv)

    rank, world, device = dp.init_from_env()
    if device.type != "cuda":
        raise SystemExit("training needs an MI355X: the chord-spmm path has no CPU implementation")
    cfg_model, cfg_training = dict(config["DDcDNA"]["PSF"]), config["DDcDNA"]["training"]
    if args.n_vec:
        cfg_model["n_vec"] = args.n_vec
        cfg_model["n_W"] = max(1, (args.n_vec - 1).bit_length())  # log2 factors, as 14 for 16384
    batch = args.batch_size or cfg_training["batch_size"]
    epochs = args.epochs if args.epochs is not None else cfg_training["num_train_steps"]
    clip = config["DDcDNA"]["grad_clip_norm"]

    seed_everything(42)
    net = build_model(n_vec=cfg_model["n_vec"], n_W=cfg_model["n_W"]).to(device)
    dp.broadcast_parameters(net)
    if rank == 0:
        print('Number of trainable parameters', count_params(net))
    loss = nn.CrossEntropyLoss()
    optimizer = make_adam(net.parameters(), cfg_training['learning_rate'], capturable=args.graph and world == 1)
    reducer = dp.FlatGradAllReduce(net.parameters()) if world > 1 else None

    cls = cfg_model['pooling_type'] == 'CLS'
    splits = {}
    for split, n, seed in (("train", args.train_seqs, 100 + rank), ("val", args.eval_seqs, 200), ("test", args.eval_seqs, 300)):
        if args.data_dir:
            X, Y = load_split(split, args.data_dir, device)
            if split == "train":
                lo, hi = dp.shard_bounds(X.size(0), rank, world)
                X, Y = X[lo:hi], Y[lo:hi]
        else:
            lo, hi = dp.shard_bounds(n, rank, world) if split == "train" else (0, n)
            X, Y = synthetic_split(hi - lo, device, seed, cfg_model["n_vec"] - (1 if cls else 0))
        if cls:
            X = add_cls_token(X, cfg_model['vocab_size'])
        splits[split] = (X, Y)
    # genome_clf_training.py:144-182: train and validation drop the last partial batch, the test loader keeps it
    trainloader = DeviceBatches(*splits["train"], batch, shuffle=True, drop_last=True)
    valloader = DeviceBatches(*splits["val"], batch, shuffle=False, drop_last=True)
    testloader = DeviceBatches(*splits["test"], batch, shuffle=False, drop_last=False)

    graphed = (GraphedStep(net, optimizer, loss, *next(iter(trainloader)), reducer=reducer, grad_clip_norm=clip)
               if args.graph else None)

    if args.json:
        train_epoch(net, trainloader, optimizer, loss, reducer, max_steps=3, graphed=graphed, grad_clip_norm=clip)
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        stats = train_epoch(net, trainloader, optimizer, loss, reducer, max_steps=args.max_steps, graphed=graphed,
                            grad_clip_norm=clip)
        torch.cuda.synchronize(device)
        dt = time.perf_counter() - t0
        if rank == 0:
            print(json.dumps({"metric": "PSF train tokens/sec", "task": "genome (DDcDNA)", "n_vec": cfg_model["n_vec"],
                              "n_gpus": world, "batch_per_gpu": batch, "steps": stats["steps"],
                              "ms_per_step": dt * 1e3 / max(stats["steps"], 1),
                              "value": world * batch * cfg_model["n_vec"] * stats["steps"] / dt, "unit": "tokens/s",
                              "loss": stats["loss"], "data": "files" if args.data_dir else "synthetic",
                              "hip_graph": bool(args.graph)}))
        return

    TrainGenomePSF(net=net, trainloader=trainloader, valloader=valloader, testloader=testloader, n_epochs=epochs,
                   test_freq=cfg_training['eval_frequency'], optimizer=optimizer, loss=loss,
                   saving_criteria=config["DDcDNA"]["saving_criteria"], reducer=reducer, is_main=rank == 0, graphed=graphed,
                   grad_clip_norm=clip)


if __name__ == "__main__":
    main()
