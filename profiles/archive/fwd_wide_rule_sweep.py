#!/usr/bin/env python3
"""Rows of >= 64 channels, forward chain in inference mode: one workgroup per whole row (fwd_wide = 0 before round 4) against
32-channel chunks on 1024-thread workgroups (fwd_wide = 1), 2 rows per thread; us per step, median of seven, interleaved."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsefactorization_amd as sfa  # noqa: E402

dev = torch.device("cuda:0")
WHOLE = int(sys.argv[1]) if len(sys.argv) > 1 else 0  # knob value that means "whole row per workgroup"
for B, N, M, C in ((32, 2048, 11, 64), (32, 2048, 11, 96), (32, 2048, 11, 128), (32, 2000, 11, 128), (32, 4096, 12, 64), (16, 4096, 12, 128),
                   (8, 16384, 14, 64), (32, 2048, 11, 192), (32, 2048, 11, 256), (8, 2048, 11, 512), (32, 1024, 10, 64), (32, 600, 9, 64)):
    g = torch.Generator(device=dev).manual_seed(0)
    Ws = [0.1 * torch.randn(B, N, M + 1, device=dev, generator=g) for _ in range(M)]
    V0 = torch.randn(B, N, C, device=dev, generator=g)

    def reading(chains=20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        with torch.no_grad():
            for _ in range(chains):
                out = sfa.chord_chain(Ws, V0, False)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / chains / M * 1e3, out

    sfa.set_tuning("chain_fused", 0)  # per-step kernels (the single-launch chain has its own rule)
    reading(20)
    times, ref = {WHOLE: [], 1: [], 0: []}, None
    for rnd in range(7):
        for w in ((WHOLE, 1, 0) if rnd % 2 == 0 else (0, 1, WHOLE)):
            sfa.set_tuning("fwd_wide", w)
            t, out = reading()
            times[w].append(t)
            ref = out.clone() if ref is None else ref
            assert torch.equal(out, ref)
    sfa.set_tuning("fwd_wide", 0)
    sfa.set_tuning("chain_fused", 1)
    a, b = statistics.median(times[WHOLE]), statistics.median(times[1])
    auto = f"   automatic (knob 0) {statistics.median(times[0]):7.2f} us" if WHOLE != 0 else ""
    print(f"B={B:3d} N={N:6d} L={M + 1:2d} C={C:4d}: whole row {a:7.2f} us   32-channel chunks {b:7.2f} us   ({(a / b - 1) * 100:+5.1f} %){auto}", flush=True)
    del Ws, V0
    torch.cuda.empty_cache()
