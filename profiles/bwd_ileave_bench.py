#!/usr/bin/env python3
"""Fused backward step with the tiles of a batch element walked as 2^s interleaved fronts (knob bwd_fronts = 2^s;
csrc/bwd_fused.h, Geom::ileave): rows N / 2^s apart are then in flight together, so the longest links' sources are hot in
the XCD's L2. us per step with operands rotating (W, V, dW through `sets`; dZ = the dV of the launch before), arms interleaved.
    python profiles/bwd_ileave_bench.py [B N L C]"""
import os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsefactorization_amd as sfa  # noqa: E402
from sparsefactorization_amd import chord  # noqa: E402

shapes = [tuple(int(a) for a in sys.argv[1:5])] if len(sys.argv) >= 5 else [(16, 16384, 15, 32), (40, 16384, 15, 8), (32, 4096, 13, 32)]
dev = torch.device("cuda:0")
for B, N, L, C in shapes:
    g = torch.Generator(device=dev).manual_seed(0)
    sets = 10  # as bench.py: backward_vs_stream (W, V, dW: 1.4 GB at the Order shape)
    Ws = [0.1 * torch.randn(B, N, L, device=dev, generator=g) for _ in range(sets)]
    Vs = [torch.randn(B, N, C, device=dev, generator=g) for _ in range(sets)]
    zz = [torch.randn(B, N, C, device=dev, generator=g), torch.empty(B, N, C, device=dev)]
    dWs = [torch.empty_like(Ws[0]) for _ in range(sets)]
    it = [0]

    def reading(s, steps=100):
        sfa.set_tuning("bwd_fronts", 1 << s)
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

    # correctness: the interleave only re-orders workgroups
    ref = None
    for s in (0, 1, 2, 3):
        sfa.set_tuning("bwd_fronts", 1 << s)
        dW, dV = torch.empty_like(Ws[0]), torch.empty_like(Vs[0])
        chord._launch_bwd(zz[0], Ws[0], Vs[0], dW, dV, B, N, L, C, N * C, None)
        torch.cuda.synchronize()
        if ref is None:
            ref = (dW, dV)
        else:
            assert torch.equal(dW, ref[0]) and torch.equal(dV, ref[1]), s
    reading(0, 200)
    times = {s: [] for s in (0, 1, 2, 3)}
    for rnd in range(5):
        for s in ((0, 1, 2, 3) if rnd % 2 == 0 else (3, 2, 1, 0)):
            times[s].append(reading(s))
    sfa.set_tuning("bwd_fronts", 0)
    alg = 4 * B * N * (2 * L + 3 * C)
    print(f"B={B} N={N} L={L} C={C} ({sets} operand sets, {alg / 1e6:.1f} MB algorithmic): " +
          "   ".join(f"fronts {1 << s}: {statistics.median(times[s]):.2f} us ({alg / statistics.median(times[s]) / 8e6:.3f} of 8 TB/s)" for s in times), flush=True)
