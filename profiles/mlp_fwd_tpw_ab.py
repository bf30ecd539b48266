#!/usr/bin/env python3
"""A/B in one process: x3_fwd_k with two tiles per wave (two workgroups per CU) vs one tile per wave (146 registers: three
workgroups per CU), knob mlp_fwd_tpw. After a warm-up the two settings alternate (A B A B ...), 20 calls per reading, six
readings each; ms per call, median and all readings; results must be bit-equal.   python profiles/mlp_fwd_tpw_ab.py"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsefactorization_amd as sfa  # noqa: E402
from sparsefactorization_amd import fused_mlp  # noqa: E402
from sparsefactorization_amd.psfnet import MLPBlock  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)


def reading(x, blocks, calls=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.no_grad():
        torch.cuda.synchronize()
        e0.record()
        for _ in range(calls):
            ys = fused_mlp.fused_mlp_forward(x, blocks)
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / calls, ys


for name, T, layers in (("Order B=40", 40 * 16384, [(32, 8)] + [(32, 15)] * 14), ("Adding B=64", 64 * 16384, [(32, 8)] + [(32, 15)] * 14),
                        ("Pathfinder B=64", 64 * 1024, [(128, 32)] + [(128, 12)] * 11), ("N=2048 B=32", 32 * 2048, [(32, 8)] + [(32, 12)] * 11)):
    blocks = [MLPBlock([h, 'GELU'], 32, o).to(dev) for h, o in layers]
    x = torch.randn(T, 32, device=dev)
    reading(x, blocks, 200)  # clocks up
    times = {1: [], 2: []}
    outs = {}
    for rnd in range(6):
        for tpw in ((2, 1) if rnd % 2 == 0 else (1, 2)):
            sfa.set_tuning("mlp_fwd_tpw", tpw)
            t, outs[tpw] = reading(x, blocks)
            times[tpw].append(t)
    print(f"{name}: tpw=2 median {statistics.median(times[2]):.4f} ms {['%.4f' % t for t in times[2]]} | tpw=1 median "
          f"{statistics.median(times[1]):.4f} ms {['%.4f' % t for t in times[1]]} | bit-equal",
          all(torch.equal(a, b) for a, b in zip(outs[2], outs[1])))
sfa.set_tuning("mlp_fwd_tpw", 0)
