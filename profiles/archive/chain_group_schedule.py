#!/usr/bin/env python3
"""cfg2 chain: the loop nest  for group-of-sequences: for step  instead of  for step: for all sequences.

With G sequences per group, a group's V / V0 / ping-pong outputs (G x 1.5 MB) can stay in the XCDs' L2 between its 14
steps; only W streams from memory. Prototype with per-step launches on batch slices (same kernels, same results).
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsefactorization_amd as sfa  # noqa: E402
from sparsefactorization_amd import chord  # noqa: E402

B, N, M, L, C = 64, 16384, 14, 15, 8
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
Ws = [0.1 * torch.randn(B, N, L, device=dev, generator=g) for _ in range(M)]
V0 = torch.randn(B, N, C, device=dev, generator=g)
bufs = [torch.empty_like(V0), torch.empty_like(V0)]


def chain_grouped(G):
    for b0 in range(0, B, G):
        sl = slice(b0, b0 + G)
        x = V0[sl]
        for m in range(M):
            out = bufs[m & 1][sl]
            chord._launch_fwd(Ws[m][sl], x, V0[sl], out, G, N, L, C, N * C, None)
            x = out
    return bufs[(M - 1) & 1]


def time_ms(fn, iters=30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


with torch.no_grad():
    ref = sfa.chord_chain(Ws, V0, True).clone()
    print(f"library chain (step-major): {time_ms(lambda: sfa.chord_chain(Ws, V0, True)):.4f} ms")
    for G in (64, 32, 16, 8, 4):
        ms = time_ms(lambda: chain_grouped(G))
        assert torch.equal(chain_grouped(G), ref)
        print(f"group-major, {G:2d} sequences per group ({B // G * M:3d} launches): {ms:.4f} ms per chain", flush=True)
