#!/usr/bin/env python3
"""Rows of 32 channels on the 1024-thread / 256-row tile of the wide-row configuration against the automatic 64-row tile (needs a
library in which fwd_wide = 1 also applies to C / 4 >= 8: psf_chord.hip, pick_window, `wide == 1 && CG >= 8` — the shipped rule
starts at 64 channels): forward chain us per step, cache-resident and with operands rotating; dV likewise.  python profiles/fwd_c32_wide_tile_ab.py"""
import os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsefactorization_amd as sfa  # noqa: E402

dev = torch.device("cuda:0")
for tag, B, N, L, C in (("genome-like", 16, 16384, 15, 32), ("IMDb-like", 32, 4096, 13, 32), ("IMDb 4097", 32, 4097, 13, 32), ("C=32 N=8192", 16, 8192, 14, 32),
                        ("C=48 N=16384", 8, 16384, 15, 48)):
    M = L - 1
    g = torch.Generator(device=dev).manual_seed(0)
    chain_bytes = M * 4 * B * N * L + 4 * B * N * C
    sets = max(2, min(24, int(640e6 / chain_bytes) + 1))
    Wsets = [[0.1 * torch.randn(B, N, L, device=dev, generator=g) for _ in range(M)] for _ in range(sets)]
    V0s = [torch.randn(B, N, C, device=dev, generator=g) for _ in range(sets)]

    def reading(rot, n=12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        with torch.no_grad():
            for i in range(n):
                s = i % sets if rot else 0
                out = sfa.chord_chain(Wsets[s], V0s[s], False)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n / M, out
    res, outs = {}, {}
    for rd in range(6):
        for w in ((0, 1) if rd % 2 == 0 else (1, 0)):
            sfa.set_tuning("fwd_wide", w)
            try:
                for rot in (False, True):
                    t, out = reading(rot)
                    if rd:
                        res.setdefault((w, rot), []).append(t)
                outs[w] = out.clone()
            finally:
                sfa.set_tuning("fwd_wide", 0)
    sfa.set_tuning("fwd_wide", 1)
    desc = sfa.describe_fwd(B, N, L, C)[:60]
    sfa.set_tuning("fwd_wide", 0)
    med = {k: statistics.median(v) for k, v in res.items()}
    print(f"{tag:14s} B={B} N={N} L={L} C={C}: auto {med[(0, False)]:.2f} / {med[(0, True)]:.2f} us  1024-thread tile {med[(1, False)]:.2f} / {med[(1, True)]:.2f} us "
          f"(cache-resident / rotating)  equal bits {bool(torch.equal(outs[0], outs[1]))}  [{desc}]", flush=True)
    del Wsets, V0s
    torch.cuda.empty_cache()
