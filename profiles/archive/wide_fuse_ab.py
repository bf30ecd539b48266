#!/usr/bin/env python3
"""A/B in one process: second layers in the forward GEMM's epilogue (wide_fuse = 1) vs the separate kernel (0); ms per
training step and per no-grad forward, ListOps reference configuration and BASELINE's wording (N = 2048, E = C = 64).
    python profiles/wide_fuse_ab.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsefactorization_amd as sfa  # noqa: E402
from sparsefactorization_amd import lra_training  # noqa: E402

dev = torch.device("cuda:0")
for tag, over in (("listops reference config", {}), ("listops N=2048 E=C=64", dict(n_vec=2048, embedding_size=64, n_channels_V=64))):
    for rnd in range(2):
        row = []
        for fuse in (1, 0):
            sfa.set_tuning("wide_fuse", fuse)
            r = lra_training.train_benchmark("listops", steps=30, warmup=5, device=dev, **over)
            net = lra_training.build_model("listops", **over).to(dev).eval()
            X = torch.randint(0, 15, (32, net.n_vec), device=dev)
            with torch.no_grad():
                for _ in range(3):
                    net(X)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(20):
                    net(X)
                torch.cuda.synchronize()
                inf = (time.perf_counter() - t0) / 20 * 1e3
            row.append(f"wide_fuse={fuse}: train {r['event_ms'] / r['steps']:.3f} ms/step, inference forward {inf:.3f} ms")
        print(f"{tag} (round {rnd}): " + " | ".join(row))
sfa.set_tuning("wide_fuse", 1)
