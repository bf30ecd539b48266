#!/usr/bin/env python3
"""Forward chain of the Temporal-Order training shape (N = 16384, L = 15, C = 8, B = 40; every step's output kept, operands
rotating) over workgroups per CU (knob fwd_wg_limit: 0 = the rule, 1 = what fits, 2..4), us per step-launch, interleaved."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsefactorization_amd as sfa  # noqa: E402
from sparsefactorization_amd import chord  # noqa: E402

dev = torch.device("cuda:0")
for B in (40, 48, 64):
    N, L, C, M = 16384, 15, 8, 14
    g = torch.Generator(device=dev).manual_seed(1)
    Ws = [0.1 * torch.randn(B, N, L, device=dev, generator=g) for _ in range(M)]
    V0 = torch.randn(B, N, C, device=dev, generator=g)
    outs = [torch.empty_like(V0) for _ in range(M)]

    def chain():
        x = V0
        for m in range(M):
            chord._launch_fwd(Ws[m], x, V0, outs[m], B, N, L, C, N * C, None)
            x = outs[m]

    res = {}
    for rd in range(5):
        for lim in (0, 1, 2, 3, 4):
            sfa.set_tuning("fwd_wg_limit", lim)
            for _ in range(3):
                chain()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(20):
                chain()
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(lim, []).append(e0.elapsed_time(e1) * 1e3 / 20 / M)
    sfa.set_tuning("fwd_wg_limit", 0)
    print(f"B={B}: " + "  ".join(f"limit {k}: {min(v):.2f}" for k, v in res.items()) + "   us per step (best of 5)", flush=True)
