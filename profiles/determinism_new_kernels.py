#!/usr/bin/env python3
"""Bit stability of the kernels added late in round 6: 300 runs each of chord_chain_rows_k (eight channels per workgroup, ListOps
shape; one channel group x five rows, the text task's shape) and of chord_chain_bwd_lds_k (cfg1 and N = 1024), every result
compared with the first run's bit for bit.    python profiles/determinism_new_kernels.py [runs]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsefactorization_amd as sfa  # noqa: E402
from sparsefactorization_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 300
g = torch.Generator(device=dev).manual_seed(5)
bad = 0
for B, N, L, C, M in ((32, 2000, 12, 128, 11), (32, 4097, 14, 32, 12)):
    Ws = [0.2 * torch.randn(B, N, L, device=dev, generator=g) for _ in range(M)]
    V0 = torch.randn(B, N, C, device=dev, generator=g)
    desc = _lib.describe_chain_fwd(B, N, L, C, M)
    assert "chord_chain_rows_k" in desc, desc
    with torch.no_grad():
        ref = sfa.chord_chain(Ws, V0, True).clone()
        diff = sum(int(not torch.equal(sfa.chord_chain(Ws, V0, True), ref)) for _ in range(runs))
    bad += diff
    print(f"{desc[:48]}: {runs} runs, {diff} differ", flush=True)
for B, N, L, C, M in ((40, 128, 8, 8, 7), (40, 1024, 11, 8, 10)):
    assert _lib.load().psf_chord_chain_bwd_supported(N, L, C, M) == 1
    W0 = [0.2 * torch.randn(B, N, L, device=dev, generator=g) for _ in range(M)]
    V0 = torch.randn(B, N, C, device=dev, generator=g)
    gout = torch.randn(B, N, C, device=dev, generator=g)

    def grads():
        Wg = [w.clone().requires_grad_(True) for w in W0]
        Vg = V0.clone().requires_grad_(True)
        sfa.chord_chain(Wg, Vg, True).backward(gout)
        return [Vg.grad] + [w.grad for w in Wg]
    ref = grads()
    diff = 0
    for _ in range(runs):
        diff += int(any(not torch.equal(a, b) for a, b in zip(grads(), ref)))
    bad += diff
    print(f"chord_chain_bwd_lds_k N={N}: {runs} runs, {diff} differ", flush=True)
sys.exit(1 if bad else 0)
