#!/usr/bin/env python3
"""Short sequences, forward chain in inference mode: M per-step launches (chain_fused = 0) against ONE LDS-resident launch
wherever it fits (2) and the library's automatic choice (1); us per step, median of seven, interleaved, bit-compared."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsefactorization_amd as sfa  # noqa: E402

dev = torch.device("cuda:0")
TRAIN = len(sys.argv) > 1 and sys.argv[1] == "train"  # every step's output kept (what a training forward does)
for B, N, M, C in ((32, 2048, 11, 64), (32, 2000, 11, 128), (32, 2048, 11, 128), (64, 1024, 11, 32), (32, 1024, 10, 16), (40, 128, 7, 8),
                   (64, 2048, 11, 8), (64, 512, 9, 8), (8, 1024, 11, 1024), (32, 2048, 11, 32), (32, 1025, 11, 32)):
    g = torch.Generator(device=dev).manual_seed(0)
    Ws = [(0.1 * torch.randn(B, N, M + 1, device=dev, generator=g)).requires_grad_(TRAIN) for _ in range(M)]
    V0 = torch.randn(B, N, C, device=dev, generator=g)

    def reading(chains=20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        with torch.set_grad_enabled(TRAIN):
            for _ in range(chains):
                out = sfa.chord_chain(Ws, V0, False)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / chains / M * 1e3, out.detach()

    reading(20)
    times, ref = {0: [], 2: [], 1: []}, None
    for rnd in range(7):
        for k in ((0, 2, 1) if rnd % 2 == 0 else (1, 2, 0)):
            sfa.set_tuning("chain_fused", k)
            t, out = reading()
            times[k].append(t)
            ref = out.clone() if ref is None else ref
            assert torch.equal(out, ref)
    sfa.set_tuning("chain_fused", 1)
    cc_t = {}
    for cc in (0, 1):  # channel groups per workgroup of the single launch: automatic (two where the rows fit) or one
        sfa.set_tuning("chain_cc", cc)
        sfa.set_tuning("chain_fused", 2)
        cc_t[cc] = statistics.median([reading()[0] for _ in range(5)])
    sfa.set_tuning("chain_cc", 0)
    sfa.set_tuning("chain_fused", 1)
    m = {k: statistics.median(v) for k, v in times.items()}
    print(f"B={B:3d} N={N:5d} L={M + 1:2d} C={C:4d}: per-step {m[0]:7.2f} us   one launch where it fits {m[2]:7.2f} us   automatic {m[1]:7.2f} us   one launch with cc auto / 1: {cc_t[0]:6.2f} / {cc_t[1]:6.2f}   "
          f"{sfa._lib.describe_chain_fwd(B, N, M + 1, C, M)[:60] if hasattr(sfa, 'describe_chain_fwd') else ''}", flush=True)
    del Ws, V0
    torch.cuda.empty_cache()
