#!/usr/bin/env python3
"""Lab: the fused backward step for rows of 64 / 128 channels (TG = 16 / 32 lanes per row, tiles of 16 / 8 rows: since round 6 the automatic choice
for 64 channels and for 128 up to N = 4096; knob bwd_fused = 1) against the two-kernel path (chord_dv_win_k + chord_dw_chunk_k; knob bwd_fused = 0). Operands rotating, dZ chained; us per step,
median of five, arms interleaved; dV compared bit for bit, dW against the oracle.   python profiles/bwd_fused_wide_ab.py"""
import os, statistics, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsefactorization_amd as sfa  # noqa: E402
from sparsefactorization_amd import chord  # noqa: E402
from oracle import chord_oracle as oc  # noqa: E402  (lab script: the oracle is the checker)

dev = torch.device("cuda:0")
SHAPES = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(32, 2000, 12, 128), (32, 2048, 12, 64), (8, 16384, 15, 64), (8, 16384, 15, 128), (32, 4096, 13, 64)]
for B, N, L, C in SHAPES:
    g = torch.Generator(device=dev).manual_seed(0)
    sets = 10
    Ws = [0.1 * torch.randn(B, N, L, device=dev, generator=g) for _ in range(sets)]
    Vs = [torch.randn(B, N, C, device=dev, generator=g) for _ in range(sets)]
    zz = [torch.randn(B, N, C, device=dev, generator=g), torch.empty(B, N, C, device=dev)]
    dWs = [torch.empty_like(Ws[0]) for _ in range(sets)]
    it = [0]
    outs = {}
    for wide in (0, 1):
        sfa.set_tuning("bwd_fused", wide)
        dW, dV = torch.full_like(Ws[0], float("nan")), torch.full_like(Vs[0], float("nan"))
        chord._launch_bwd(zz[0], Ws[0], Vs[0], dW, dV, B, N, L, C, N * C, None)
        torch.cuda.synchronize()
        outs[wide] = (dW, dV)
    dF, dVo = oc.spmul_bwd(zz[0][:2].cpu().numpy(), Ws[0][:2].cpu().numpy(), Vs[0][:2].cpu().numpy())
    ok_dv = torch.equal(outs[0][1], outs[1][1]) and np.array_equal(outs[1][1][:2].cpu().numpy(), dVo)
    err_dw = float(np.abs(outs[1][0][:2].cpu().numpy() - dF).max() / np.abs(dF).max())

    def reading(wide, steps=100):
        sfa.set_tuning("bwd_fused", wide)
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

    reading(0, 100)
    times = {0: [], 1: []}
    for rnd in range(5):
        for w in ((0, 1) if rnd % 2 == 0 else (1, 0)):
            times[w].append(reading(w))
    sfa.set_tuning("bwd_fused", 1)
    print(f"B={B} N={N} L={L} C={C}: two kernels {statistics.median(times[0]):.2f} us   fused {statistics.median(times[1]):.2f} us   "
          f"dV bit-equal (and = oracle): {ok_dv}   dW rel err vs oracle {err_dw:.1e}", flush=True)
