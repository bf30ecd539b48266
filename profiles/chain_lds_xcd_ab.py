#!/usr/bin/env python3
"""The one-launch chain (chord_chain_lds_k) with its workgroups dealt to XCDs so that the workgroups of a sequence share an
XCD (knob xcd_remap = 1, round 6) against blockIdx order (xcd_remap = 0); no-grad chains (only the last result kept), W
rotating; us per step-equivalent, arms interleaved, results compared bit for bit.   python profiles/chain_lds_xcd_ab.py"""
import os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsefactorization_amd as sfa  # noqa: E402
from sparsefactorization_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
for B, N, L, C, amap in ((32, 2000, 12, 128, False), (32, 2048, 12, 64, False), (64, 1024, 12, 32, False), (8, 1024, 12, 1024, True), (32, 1025, 12, 32, False), (32, 2049, 13, 128, False), (40, 128, 8, 8, False)):
    M = L - 1
    g = torch.Generator(device=dev).manual_seed(1)
    sets = max(2, min(12, int(640e6 // (M * 4 * B * N * L))))
    Wsets = [[0.1 * torch.randn(B, N, L, device=dev, generator=g) for _ in range(M)] for _ in range(sets)]
    V0 = torch.eye(N, device=dev) if amap else torch.randn(B, N, C, device=dev, generator=g)
    desc = _lib.describe_chain_fwd(B, N, L, N if amap else C, M)
    ref, times, it = None, {1: [], 0: []}, [0]
    with torch.no_grad():
        for r in (1, 0):
            sfa.set_tuning("xcd_remap", r)
            out = sfa.chord_chain(Wsets[0], V0, False)
            if ref is None:
                ref = out.clone()
            assert torch.equal(out, ref)
        for rnd in range(7):
            for r in ((1, 0) if rnd % 2 == 0 else (0, 1)):
                sfa.set_tuning("xcd_remap", r)
                for _ in range(2):
                    sfa.chord_chain(Wsets[it[0] % sets], V0, False)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(20):
                    it[0] += 1
                    sfa.chord_chain(Wsets[it[0] % sets], V0, False)
                e1.record()
                torch.cuda.synchronize()
                times[r].append(e0.elapsed_time(e1) / 20 / M * 1e3)
        sfa.set_tuning("xcd_remap", 1)
    print(f"B={B} N={N} L={L} C={N if amap else C}: XCD-aware {statistics.median(times[1]):.2f} us per step   blockIdx order {statistics.median(times[0]):.2f}   [{desc[:90]}]", flush=True)
