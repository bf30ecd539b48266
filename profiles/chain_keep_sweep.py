#!/usr/bin/env python3
"""Chains that keep EVERY step's result (training) and chains that keep the last one (inference): M per-step launches (knob
chain_fused = 0) against the one-launch chain (chain_fused = 2, where it fits), W rotating; us per step, arms interleaved.
Round 6: re-measured after the one-launch chain got its XCD-aware workgroup order.   python profiles/chain_keep_sweep.py"""
import os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsefactorization_amd as sfa  # noqa: E402
from sparsefactorization_amd import chord  # noqa: E402

dev = torch.device("cuda:0")
SHAPES = [(64, 1024, 12, 32), (32, 1024, 11, 16), (64, 1025, 12, 32), (32, 2048, 12, 32), (32, 2048, 12, 64), (32, 2000, 12, 128), (32, 2049, 13, 128),
          (64, 2048, 12, 8), (64, 2048, 12, 16), (32, 1024, 11, 64), (32, 1024, 11, 128), (64, 512, 10, 32), (32, 2000, 12, 64)]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for B, N, L, C in SHAPES:
    M = L - 1
    g = torch.Generator(device=dev).manual_seed(1)
    sets = max(2, min(12, int(640e6 // (M * 4 * B * N * L))))
    Wsets = [[0.1 * torch.randn(B, N, L, device=dev, generator=g) for _ in range(M)] for _ in range(sets)]
    V0 = torch.randn(B, N, C, device=dev, generator=g)
    res = C <= 8
    out = {}
    for keep in (True, False):
        times, it = {0: [], 2: []}, [0]
        with torch.no_grad():
            for rnd in range(5):
                for cf in ((0, 2) if rnd % 2 == 0 else (2, 0)):
                    sfa.set_tuning("chain_fused", cf)
                    for _ in range(2):
                        chord._chain_forward_raw(V0, res, None, Wsets[it[0] % sets], keep)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    torch.cuda.synchronize()
                    e0.record()
                    for _ in range(16):
                        it[0] += 1
                        chord._chain_forward_raw(V0, res, None, Wsets[it[0] % sets], keep)
                    e1.record()
                    torch.cuda.synchronize()
                    times[cf].append(e0.elapsed_time(e1) / 16 / M * 1e3)
        sfa.set_tuning("chain_fused", 1)
        out[keep] = (statistics.median(times[0]), statistics.median(times[2]))
    print(f"B={B} N={N} L={L} C={C} ({N * C} elements per sequence): every step kept: per-step {out[True][0]:.2f} / one launch {out[True][1]:.2f} us"
          f"   last kept: per-step {out[False][0]:.2f} / one launch {out[False][1]:.2f}", flush=True)
