#!/usr/bin/env python3
"""The forward step kernel launched again and again on ONE operand set (what profiles/bwd_pmc_run.py does under the profiler:
everything L2 / Infinity-Cache resident) with two and four rows per thread; us per launch, arms interleaved."""
import os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsefactorization_amd as sfa  # noqa: E402
from sparsefactorization_amd import chord  # noqa: E402

dev = torch.device("cuda:0")
for B, N, L, C in ((64, 1024, 12, 32), (32, 2048, 12, 32), (32, 4096, 13, 32), (16, 16384, 15, 32), (32, 1024, 11, 16), (64, 4096, 13, 16), (8, 4096, 13, 64)):
    g = torch.Generator(device=dev).manual_seed(0)
    W = 0.1 * torch.randn(B, N, L, device=dev, generator=g)
    V = torch.randn(B, N, C, device=dev, generator=g)
    out = torch.empty_like(V)
    times = {2: [], 4: []}
    for rnd in range(7):
        for r in ((2, 4) if rnd % 2 == 0 else (4, 2)):
            sfa.set_tuning("fwd_rows", r)
            for _ in range(10):
                chord._launch_fwd(W, V, None, out, B, N, L, C, N * C, None)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(200):
                chord._launch_fwd(W, V, None, out, B, N, L, C, N * C, None)
            e1.record()
            torch.cuda.synchronize()
            times[r].append(e0.elapsed_time(e1) / 200 * 1e3)
    sfa.set_tuning("fwd_rows", 0)
    print(f"B={B} N={N} L={L} C={C}: two rows {statistics.median(times[2]):.2f} us   four rows {statistics.median(times[4]):.2f} us", flush=True)
