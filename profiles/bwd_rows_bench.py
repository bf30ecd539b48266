#!/usr/bin/env python3
"""Fused backward step with one or two rows per thread (lab: profiles/r06l_bwd_fused_two_rows.patch adds the template parameter R and the knob bwd_rows; not shipped) x workgroups per
CU (bwd_fused_wg_limit), operands rotating (W, V, dW through ten sets; dZ = the dV of the launch before); us per step, median
of five readings, arms interleaved; gradients compared bit for bit.   python profiles/bwd_rows_bench.py [BxNxLxC ...]"""
import os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsefactorization_amd as sfa  # noqa: E402
from sparsefactorization_amd import chord  # noqa: E402

dev = torch.device("cuda:0")
SHAPES = [(16, 16384, 15, 32), (32, 4096, 13, 32), (64, 1024, 12, 32), (64, 16384, 15, 16), (32, 1024, 11, 16), (64, 16384, 15, 32)]  # (wide rows: pass BxNxLxC)
if len(sys.argv) > 1:
    SHAPES = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for B, N, L, C in SHAPES:
    g = torch.Generator(device=dev).manual_seed(0)
    sets = 10
    Ws = [0.1 * torch.randn(B, N, L, device=dev, generator=g) for _ in range(sets)]
    Vs = [torch.randn(B, N, C, device=dev, generator=g) for _ in range(sets)]
    zz = [torch.randn(B, N, C, device=dev, generator=g), torch.empty(B, N, C, device=dev)]
    dWs = [torch.empty_like(Ws[0]) for _ in range(sets)]
    it = [0]

    def reading(rows, wg, steps=100):
        sfa.set_tuning("bwd_rows", rows)
        sfa.set_tuning("bwd_fused_wg_limit", wg)
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

    ref = None
    for rows in (1, 2):
        sfa.set_tuning("bwd_rows", rows)
        dW, dV = torch.empty_like(Ws[0]), torch.empty_like(Vs[0])
        chord._launch_bwd(zz[0], Ws[0], Vs[0], dW, dV, B, N, L, C, N * C, None)
        torch.cuda.synchronize()
        if ref is None:
            ref = (dW, dV)
        else:
            assert torch.equal(dV, ref[1]), "dV differs"
            assert torch.equal(dW, ref[0]), "dW differs"
    arms = [(r, w) for r in (1, 2) for w in (0, 2, 3, 4)]
    reading(1, 0, 200)
    times = {a: [] for a in arms}
    for rnd in range(5):
        for a in (arms if rnd % 2 == 0 else arms[::-1]):
            times[a].append(reading(*a))
    sfa.set_tuning("bwd_rows", 0)
    sfa.set_tuning("bwd_fused_wg_limit", 0)
    alg = 4 * B * N * (2 * L + 3 * C)
    print(f"B={B} N={N} L={L} C={C} ({alg / 1e6:.1f} MB algorithmic): " + "  ".join(f"r{r}/wg{w}: {statistics.median(t):.2f}" for (r, w), t in times.items()), flush=True)
