#!/usr/bin/env python3
"""What a sequence length of 2^k + 1 (LRA's CLS-token column, LRA/listops_training.py:65-72: IMDb N = 4097) costs against 2^k:
forward chain and backward step, us per step, median of seven readings.   python profiles/ragged_penalty.py"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsefactorization_amd as sfa  # noqa: E402
from sparsefactorization_amd import chord  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)


def timed(fn, n):
    for i in range(max(3, n // 4)):
        fn(i)
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for i in range(n):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / n)
    return statistics.median(ts)


for tag, B, N0, L, C in (("IMDb-like", 32, 4096, 13, 32), ("Pathfinder-like", 64, 1024, 11, 32), ("ListOps-like", 32, 2048, 12, 64),
                         ("narrow rows", 32, 16384, 15, 8)):
    for N in (N0, N0 + 1):
        M = L - 1
        Ws = [0.1 * torch.randn(B, N, L, device=dev, generator=g) for _ in range(M)]
        V0 = torch.randn(B, N, C, device=dev, generator=g)
        with torch.no_grad():
            t_f = timed(lambda i: sfa.chord_chain(Ws, V0, False), 20) / M
        sets = min(8, M)
        Vs = [torch.randn(B, N, C, device=dev, generator=g) for _ in range(sets)]
        dZ = torch.randn(B, N, C, device=dev, generator=g)
        dWs = [torch.empty_like(Ws[0]) for _ in range(sets)]
        dVs = [torch.empty_like(V0) for _ in range(sets)]
        t_b = timed(lambda i: chord._launch_bwd(dZ, Ws[i % sets], Vs[i % sets], dWs[i % sets], dVs[i % sets], B, N, L, C, N * C, None), 50)
        print(f"{tag:16s} B={B} N={N:6d} L={L} C={C}: forward {t_f:7.2f} us/step ({sfa.describe_fwd(B, N, L, C)[:70]})   backward {t_b:7.2f} us/step", flush=True)
        del Ws, Vs, dWs, dVs
        torch.cuda.empty_cache()
