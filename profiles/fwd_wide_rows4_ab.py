#!/usr/bin/env python3
"""Rows of 64..256 channels at N <= 4096: the automatic choice (32-channel chunks on 1024-thread workgroups) against whole rows
with two and with four rows per thread (knobs fwd_wide = 4, fwd_rows); chains that keep every step, W rotating; us per step."""
import os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsefactorization_amd as sfa  # noqa: E402
from sparsefactorization_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
for B, N, L, C in ((32, 2048, 12, 64), (32, 2048, 12, 128), (32, 4096, 13, 64), (16, 4096, 13, 128), (32, 1024, 11, 64)):
    M = L - 1
    g = torch.Generator(device=dev).manual_seed(1)
    sets = max(2, min(12, int(640e6 // (M * 4 * B * N * L))))
    Wsets = [[0.1 * torch.randn(B, N, L, device=dev, generator=g) for _ in range(M)] for _ in range(sets)]
    V0 = torch.randn(B, N, C, device=dev, generator=g)
    arms = {"auto": (0, 0), "whole rows, 2 per thread": (4, 2), "whole rows, 4 per thread": (4, 4)}
    ref, desc, times = None, {}, {a: [] for a in arms}
    it = [0]
    with torch.no_grad():
        sfa.set_tuning("chain_fused", 0)
        for a, (w, r) in arms.items():
            sfa.set_tuning("fwd_wide", w)
            sfa.set_tuning("fwd_rows", r)
            desc[a] = _lib.describe_fwd(B, N, L, C)
            out = sfa.chord_chain(Wsets[0], V0, False)
            if ref is None:
                ref = out.clone()
            assert torch.equal(out, ref), a
        for rnd in range(5):
            for a in (list(arms) if rnd % 2 == 0 else list(arms)[::-1]):
                sfa.set_tuning("fwd_wide", arms[a][0])
                sfa.set_tuning("fwd_rows", arms[a][1])
                for _ in range(2):
                    sfa.chord_chain(Wsets[it[0] % sets], V0, False)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(16):
                    it[0] += 1
                    sfa.chord_chain(Wsets[it[0] % sets], V0, False)
                e1.record()
                torch.cuda.synchronize()
                times[a].append(e0.elapsed_time(e1) / 16 / M * 1e3)
        sfa.set_tuning("fwd_wide", 0)
        sfa.set_tuning("fwd_rows", 0)
        sfa.set_tuning("chain_fused", 1)
    print(f"B={B} N={N} L={L} C={C}: " + "   ".join(f"{a}: {statistics.median(t):.2f} us" for a, t in times.items()), flush=True)
    for a in arms:
        print("      ", a, "->", desc[a], flush=True)
