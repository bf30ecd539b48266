#!/usr/bin/env python3
"""cfg2 forward step: us per launch for rows-per-thread x workgroups-per-CU (knobs fwd_rows, fwd_wg_limit)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsefactorization_amd as sfa  # noqa: E402
from sparsefactorization_amd import chord  # noqa: E402

B, N, L, C = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (64, 16384, 15, 8)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
W = 0.1 * torch.randn(B, N, L, device=dev, generator=g)
V = torch.randn(B, N, C, device=dev, generator=g)
R = torch.randn(B, N, C, device=dev, generator=g)
out = torch.empty_like(V)


def time_us(iters=100):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(20):
        chord._launch_fwd(W, V, R, out, B, N, L, C, N * C, None)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        chord._launch_fwd(W, V, R, out, B, N, L, C, N * C, None)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for rows in (1, 2):
    for lim in (1, 4, 3, 2):
        sfa.set_tuning("fwd_rows", rows)
        sfa.set_tuning("fwd_wg_limit", lim)
        print(f"B={B} N={N} L={L} C={C} rows={rows} wg_per_cu<={'any' if lim == 1 else lim}: {time_us():.2f} us", flush=True)
