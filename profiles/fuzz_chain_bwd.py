#!/usr/bin/env python3
"""Differential fuzz of the one-launch backward chain (chord_chain_bwd_lds_k behind psf_chord_chain_bwd_f32) beyond the suite's
seeds: random N <= 1024, L in 2..20, C = 4 or 8, M in 1..12, with and without the residual, explicit offsets now and then —
dV0 and every dW_m against the CPU oracle's per-step backward, bit for bit.    python profiles/fuzz_chain_bwd.py [cases]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sparsefactorization_amd as sfa  # noqa: E402
from sparsefactorization_amd import _lib  # noqa: E402
from oracle import chord_oracle as oc  # noqa: E402  (a lab script: the oracle is the checker here, as in tests/)

dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(707)
t0, bad = time.time(), 0
for i in range(cases):
    N = int(rng.integers(1, 1025)) if rng.integers(0, 4) else int(rng.choice([1, 2, 63, 64, 65, 128, 512, 1023, 1024]))
    L = int(rng.integers(2, 21))
    C = int(rng.choice([4, 8]))
    B = int(rng.integers(1, 5))
    M = int(rng.integers(1, 13))
    residual = bool(rng.integers(0, 2))
    offsets = [int(v) for v in rng.integers(-N, 2 * N + 1, size=L)] if rng.integers(0, 4) == 0 else None
    assert _lib.load().psf_chord_chain_bwd_supported(N, L, C, M) == 1
    W = (0.4 * rng.standard_normal((M, B, N, L))).astype(np.float32)
    V0 = rng.standard_normal((B, N, C)).astype(np.float32)
    gout = rng.standard_normal((B, N, C)).astype(np.float32)
    X = [V0]
    for m in range(M):
        nxt = oc.spmul_fwd(W[m], X[-1], offsets)
        X.append((nxt + V0).astype(np.float32) if residual else nxt)
    g, want_dW, terms = gout, [None] * M, []
    for m in range(M - 1, -1, -1):
        terms.append(g)
        want_dW[m], g = oc.spmul_bwd(g, W[m], X[m], offsets)
    want = g
    if residual:
        want = terms[0]
        for t in terms[1:] + [g]:
            want = (want + t).astype(np.float32)
    Wg = [torch.from_numpy(W[m]).to(dev).requires_grad_(True) for m in range(M)]
    Vg = torch.from_numpy(V0).to(dev).requires_grad_(True)
    out = sfa.chord_chain(Wg, Vg, residual, offsets=offsets)
    out.backward(torch.from_numpy(gout).to(dev))
    ok = np.array_equal(out.detach().cpu().numpy(), X[-1]) and np.array_equal(Vg.grad.cpu().numpy(), want) and all(np.array_equal(w.grad.cpu().numpy(), d) for w, d in zip(Wg, want_dW))
    if not ok:
        bad += 1
        print(f"FAIL B={B} N={N} L={L} C={C} M={M} residual={residual} offsets={offsets is not None}", flush=True)
print(f"{cases} random backward chains in one launch (forward result, dV0, every dW_m against the oracle, bit for bit): {bad} failures, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
