#!/usr/bin/env python3
"""Fused backward step, Order shape (and genome's): workgroups per CU (knob bwd_fused_wg_limit) x fronts (bwd_fronts), operands
rotating through ten sets, dZ = the dV of the launch before; us per step, median of five readings, arms interleaved.
    python profiles/bwd_wg_fronts_sweep.py"""
import os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsefactorization_amd as sfa  # noqa: E402
from sparsefactorization_amd import chord  # noqa: E402

dev = torch.device("cuda:0")
SHAPES = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(40, 16384, 15, 8), (64, 16384, 15, 8), (16, 16384, 15, 32)]
for B, N, L, C in SHAPES:
    g = torch.Generator(device=dev).manual_seed(0)
    sets = 10
    Ws = [0.1 * torch.randn(B, N, L, device=dev, generator=g) for _ in range(sets)]
    Vs = [torch.randn(B, N, C, device=dev, generator=g) for _ in range(sets)]
    zz = [torch.randn(B, N, C, device=dev, generator=g), torch.empty(B, N, C, device=dev)]
    dWs = [torch.empty_like(Ws[0]) for _ in range(sets)]
    it = [0]

    def reading(wg, fr, steps=100):
        sfa.set_tuning("bwd_fused_wg_limit", wg)
        sfa.set_tuning("bwd_fronts", fr)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(steps):
            i = it[0] % sets
            it[0] += 1
            chord._launch_bwd(zz[it[0] & 1], Ws[i], Vs[i], dWs[i], zz[1 - (it[0] & 1)], B, N, L, C, N * C, None)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / steps * 1e3

    arms = [(wg, fr) for wg in (0, 2, 3, 4, 5) for fr in (1, 2)]  # wg 0 = the automatic rule
    reading(0, 0, 200)
    times = {a: [] for a in arms}
    for rnd in range(5):
        for a in (arms if rnd % 2 == 0 else arms[::-1]):
            times[a].append(reading(*a))
    sfa.set_tuning("bwd_fused_wg_limit", 0)
    sfa.set_tuning("bwd_fronts", 0)
    print(f"B={B} N={N} L={L} C={C}: " + "  ".join(f"wg{wg}/fronts{fr}: {statistics.median(t):.2f}" for (wg, fr), t in times.items()), flush=True)
