#!/usr/bin/env python3
"""cfg2 chain (14 per-step launches): ms per chain with and without the zigzag tile order (knob chain_zigzag)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsefactorization_amd as sfa  # noqa: E402

B, N, M, L, C = 64, 16384, 14, 15, 8
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
Ws = [0.1 * torch.randn(B, N, L, device=dev, generator=g) for _ in range(M)]
V0 = torch.randn(B, N, C, device=dev, generator=g)


def time_ms(iters=50):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.no_grad():
        for _ in range(10):
            sfa.chord_chain(Ws, V0, True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(iters):
            out = sfa.chord_chain(Ws, V0, True)
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters, out


ref = None
for rep in range(5):
    for z in (0, 1):
        sfa.set_tuning("chain_zigzag", z)
        ms, out = time_ms()
        ref = out.clone() if ref is None else ref
        assert torch.equal(out, ref)
        print(f"chain_zigzag={z}: {ms:.4f} ms per chain = {ms / M * 1e3:.2f} us per step", flush=True)
