#!/usr/bin/env python3
"""A/B in one process: x3_fwd_k storing Y through the LDS-transposed contiguous bursts (mlp_fwd_store = 0) vs straight from
the accumulator registers (1). ms per call; results must be bit-equal.   python profiles/mlp_fwd_store_ab.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsefactorization_amd as sfa  # noqa: E402
from sparsefactorization_amd import fused_mlp  # noqa: E402
from sparsefactorization_amd.psfnet import MLPBlock  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
for name, T, layers in (("Order B=40", 40 * 16384, [(32, 8)] + [(32, 15)] * 14), ("Adding B=64", 64 * 16384, [(32, 8)] + [(32, 15)] * 14),
                        ("Pathfinder B=64", 64 * 1024, [(128, 32)] + [(128, 12)] * 11)):
    blocks = [MLPBlock([h, 'GELU'], 32, o).to(dev) for h, o in layers]
    x = torch.randn(T, 32, device=dev)
    outs = {}
    for rnd in range(3):
        row = []
        for store in (0, 1):
            sfa.set_tuning("mlp_fwd_store", store)
            with torch.no_grad():
                for _ in range(3):
                    ys = fused_mlp.fused_mlp_forward(x, blocks)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(20):
                    ys = fused_mlp.fused_mlp_forward(x, blocks)
                e1.record()
                torch.cuda.synchronize()
            outs[store] = ys
            row.append(f"store={store}: {e0.elapsed_time(e1) / 20:.4f} ms")
        print(f"{name} (round {rnd}): " + " | ".join(row), "| bit-equal", all(torch.equal(a, b) for a, b in zip(outs[0], outs[1])))
sfa.set_tuning("mlp_fwd_store", 0)
