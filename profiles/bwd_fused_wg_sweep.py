#!/usr/bin/env python3
"""The fused backward step on COLD operands (10 operand sets walked round-robin, as a training step sees them): workgroups
per CU (knob bwd_fused_wg_limit; 0 = what fits) x threads per workgroup (bwd_fused_nt), us per step; settings alternate
after a warm-up, median of five readings of 100 steps.   python profiles/bwd_fused_wg_sweep.py [B N L C]"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsefactorization_amd as sfa  # noqa: E402
from sparsefactorization_amd import chord  # noqa: E402

B, N, L, C = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (40, 16384, 15, 8)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
sets = 10
Ws = [0.1 * torch.randn(B, N, L, device=dev, generator=g) for _ in range(sets)]
Vs = [torch.randn(B, N, C, device=dev, generator=g) for _ in range(sets)]
dZs = [torch.randn(B, N, C, device=dev, generator=g) for _ in range(sets)]
dWs = [torch.empty_like(Ws[0]) for _ in range(sets)]
dVs = [torch.empty_like(Vs[0]) for _ in range(sets)]


def reading(steps=100):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for i in range(steps):
        s = i % sets
        chord._launch_bwd(dZs[s], Ws[s], Vs[s], dWs[s], dVs[s], B, N, L, C, N * C, None)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3


reading(300)
settings = [(nt, w) for nt in (1, 0) for w in (0, 2, 3, 4, 5)]
times = {s: [] for s in settings}
for rnd in range(5):
    for s in (settings if rnd % 2 == 0 else settings[::-1]):
        sfa.set_tuning("bwd_fused_nt", s[0])
        sfa.set_tuning("bwd_fused_wg_limit", s[1])
        times[s].append(reading())
sfa.set_tuning("bwd_fused_nt", 1)
sfa.set_tuning("bwd_fused_wg_limit", 0)
print(f"B={B} N={N} L={L} C={C}, cold operands")
for s in settings:
    print(f"threads={256 if s[0] else 512} wg_limit={s[1]}: median {statistics.median(times[s]):.2f} us  {['%.2f' % t for t in times[s]]}")
