#!/usr/bin/env python3
"""Differential fuzz of chord_chain_rows_k (both instances, forced by chain_cc = 2) beyond the suite's seeds: random lengths in
1057..2048 (eight channels per workgroup) and 2113..4160 (one channel group, five rows per thread), any link count 2..20, any
number of channel groups (odd ones leave the last workgroup one group), with and without the residual, every step kept and
ping-pong storage, explicit offsets now and then — against the CPU oracle, bit for bit.
    python profiles/fuzz_chain_rows.py [cases]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sparsefactorization_amd as sfa  # noqa: E402
from sparsefactorization_amd import _lib  # noqa: E402
from oracle import chord_oracle as oc  # noqa: E402  (a lab script: the oracle is the checker here, as in tests/)

dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(606)
t0, bad = time.time(), 0
sfa.set_tuning("chain_fused", 2)
sfa.set_tuning("chain_cc", 2)
try:
    for i in range(cases):
        long_ = bool(rng.integers(0, 2))
        N = int(rng.integers(2113, 4161)) if long_ else int(rng.integers(1057, 2049))
        if rng.integers(0, 5) == 0:
            N = int(rng.choice([4097, 4096, 4160, 2113] if long_ else [2000, 2048, 1057, 2001]))
        L = int(rng.integers(2, 21))
        C = 4 * int(rng.integers(1, 7) if long_ else rng.integers(2, 10))  # (one channel group below 2113 rows is the older kernel)
        B = int(rng.integers(1, 4))
        M = int(rng.integers(2, 6))
        residual = bool(rng.integers(0, 2))
        offsets = None
        if rng.integers(0, 4) == 0:
            offsets = [int(v) for v in rng.integers(-N, 2 * N, size=L)]
        W = (0.4 * rng.standard_normal((M, B, N, L))).astype(np.float32)
        V0 = rng.standard_normal((B, N, C)).astype(np.float32)
        X, want = V0.copy(), []
        for m in range(M):
            X = oc.spmul_fwd(W[m], X, offsets) + (V0 if residual else 0)
            X = X.astype(np.float32)
            want.append(X)
        desc = _lib.describe_chain_fwd(B, N, L, C, M)
        assert "chord_chain_rows_k" in desc, desc
        Ws = [torch.from_numpy(W[m]).to(dev) for m in range(M)]
        with torch.no_grad():
            got = sfa.chord_chain(Ws, torch.from_numpy(V0).to(dev), residual, offsets=offsets).cpu().numpy()
        Wg = [w.clone().requires_grad_(True) for w in Ws]
        out = sfa.chord_chain(Wg, torch.from_numpy(V0).to(dev), residual, offsets=offsets)
        steps = [t.detach().cpu().numpy() for t in list(out.grad_fn.saved_tensors)[1 + M:2 * M]]
        ok = np.array_equal(got, want[-1]) and np.array_equal(out.detach().cpu().numpy(), want[-1]) and all(np.array_equal(s, w) for s, w in zip(steps, want))
        if not ok:
            bad += 1
            print(f"FAIL B={B} N={N} L={L} C={C} M={M} residual={residual} offsets={offsets is not None}  [{desc[:60]}]", flush=True)
        if i % 20 == 19:
            print(f"{i + 1} cases, {bad} failures, {time.time() - t0:.0f} s", flush=True)
finally:
    sfa.set_tuning("chain_cc", 0)
    sfa.set_tuning("chain_fused", 1)
print(f"{cases} random chains on chord_chain_rows_k (both instances; ping-pong and every-step storage): {bad} failures, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
