#!/usr/bin/env python3
"""Rows of 32 channels, forward chain in inference mode: 256-thread tiles (64 rows) against 512-thread tiles (128 rows, knob
fwd_wide = 3), 1 and 2 rows per thread; us per step, median of seven, settings interleaved, results compared bit for bit."""
import itertools
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsefactorization_amd as sfa  # noqa: E402

dev = torch.device("cuda:0")
SHAPES = (("genome-like", 16, 16384, 14, 32, 0), ("IMDb-like", 32, 4096, 12, 32, 0), ("IMDb N=4097", 32, 4097, 12, 32, 0),
          ("genome B=64", 64, 16384, 14, 32, 1))
WIDES = (0, 3)
if len(sys.argv) > 1 and sys.argv[1] == "wide-rows":  # rows of >= 64 channels: whole row per workgroup against 32-channel chunks
    SHAPES = (("ListOps ref", 32, 2000, 11, 128, 0), ("ListOps 2048x64", 32, 2048, 11, 64, 0), ("attention map", 8, 1024, 11, 1024, 0))
    WIDES = (0, 1, 2)
for tag, B, N, M, C, res in SHAPES:
    g = torch.Generator(device=dev).manual_seed(0)
    Ws = [0.1 * torch.randn(B, N, M + 1, device=dev, generator=g) for _ in range(M)]
    V0 = torch.randn(B, N, C, device=dev, generator=g)

    def reading(chains=20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        with torch.no_grad():
            for _ in range(chains):
                out = sfa.chord_chain(Ws, V0, bool(res))
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / chains / M * 1e3, out

    reading(30)
    settings = list(itertools.product(WIDES, (1, 2), (0, 3)))
    times, desc, ref = {s: [] for s in settings}, {}, None
    for rnd in range(7):
        for s in (settings if rnd % 2 == 0 else settings[::-1]):
            sfa.set_tuning("fwd_wide", s[0]), sfa.set_tuning("fwd_rows", s[1]), sfa.set_tuning("fwd_wg_limit", s[2])
            desc[s] = sfa.describe_fwd(B, N, M + 1, C)
            t, out = reading()
            times[s].append(t)
            ref = out.clone() if ref is None else ref
            assert torch.equal(out, ref)
    for k in ("fwd_wide", "fwd_rows", "fwd_wg_limit"):
        sfa.set_tuning(k, 0)
    for s in settings:
        print(f"{tag:14s} wide={s[0]} rows={s[1]} wg_limit={s[2]}: {statistics.median(times[s]):6.2f} us per step   {desc[s][:75]}", flush=True)
    del Ws, V0
    torch.cuda.empty_cache()
