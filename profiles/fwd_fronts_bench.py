#!/usr/bin/env python3
"""Forward chain with the tiles of a batch element walked as 1, 2, 4, 8 interleaved fronts (Geom::ileave; needs the lab knob fwd_fronts, which round 6 removed after this measurement: no gain):
us per launch at cfg2 (N = 16384, L = 15, C = 8, B = 64, residual, M = 14), Order's batch (B = 40, every step kept) and the
genome shape; arms interleaved, outputs compared bit for bit.   python profiles/fwd_fronts_bench.py"""
import os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsefactorization_amd as sfa  # noqa: E402

dev = torch.device("cuda:0")
for B, N, L, C, res in ((64, 16384, 15, 8, True), (40, 16384, 15, 8, True), (16, 16384, 15, 32, False), (32, 4096, 13, 32, False)):
    M = L - 1
    g = torch.Generator(device=dev).manual_seed(1)
    Ws = [0.1 * torch.randn(B, N, L, device=dev, generator=g) for _ in range(M)]
    V0 = torch.randn(B, N, C, device=dev, generator=g)
    ref = None
    times = {f: [] for f in (1, 2, 4, 8)}
    with torch.no_grad():
        for f in times:
            sfa.set_tuning("fwd_fronts", f)
            out = sfa.chord_chain(Ws, V0, res)
            if ref is None:
                ref = out.clone()
            assert torch.equal(out, ref), f
        for rnd in range(7):
            for f in (list(times) if rnd % 2 == 0 else list(times)[::-1]):
                sfa.set_tuning("fwd_fronts", f)
                for _ in range(3):
                    sfa.chord_chain(Ws, V0, res)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(20):
                    sfa.chord_chain(Ws, V0, res)
                e1.record()
                torch.cuda.synchronize()
                times[f].append(e0.elapsed_time(e1) / 20 / M * 1e3)
    sfa.set_tuning("fwd_fronts", 0)
    alg = 4 * B * N * (L + 2 * C + (C if res else 0))
    print(f"B={B} N={N} L={L} C={C}: " + "   ".join(f"fronts {f}: {statistics.median(t):.2f} us ({alg / statistics.median(t) / 8e6:.3f})" for f, t in times.items()), flush=True)
