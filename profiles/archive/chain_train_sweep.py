#!/usr/bin/env python3
"""Forward chain in TRAINING mode (every step's output is kept for the backward): the single-launch LDS-resident chain
(chain_fused = 1) against M per-step launches (0), us per chain.

    python profiles/chain_train_sweep.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsefactorization_amd as sfa  # noqa: E402

SHAPES = {  # B, N, M, C, residual
    "cfg1 Adding N=128": (40, 128, 7, 8, True),
    "CIFAR N=1024 C=16": (32, 1024, 10, 16, False),
    "Pathfinder N=1024 C=32": (64, 1024, 11, 32, False),
    "ListOps N=2000 C=128": (32, 2000, 11, 128, False),
    "N=2048 C=64": (32, 2048, 11, 64, False),
    "N=2048 C=8": (64, 2048, 11, 8, True),
    "N=512 C=32": (64, 512, 9, 32, True),
}


def main():
    dev = torch.device("cuda:0")
    for name, (B, N, M, C, res) in SHAPES.items():
        g = torch.Generator(device=dev).manual_seed(0)
        Ws = [(0.1 * torch.randn(B, N, M + 1, device=dev, generator=g)).requires_grad_(True) for _ in range(M)]
        V0 = torch.randn(B, N, C, device=dev, generator=g).requires_grad_(True)
        row = []
        for mode, grad in (("train", True), ("infer", False)):
            for fused in (1, 0):
                sfa.set_tuning("chain_fused", fused)
                with torch.set_grad_enabled(grad):
                    for _ in range(5):
                        sfa.chord_chain(Ws, V0, res)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    torch.cuda.synchronize()
                    e0.record()
                    for _ in range(30):
                        sfa.chord_chain(Ws, V0, res)
                    e1.record()
                    torch.cuda.synchronize()
                row.append(f"{mode} {'one launch' if fused else 'per step'} {e0.elapsed_time(e1) / 30 * 1e3:.0f} us")
        print(f"{name:26s} " + " | ".join(row))
    sfa.set_tuning("chain_fused", 1)


if __name__ == "__main__":
    main()
