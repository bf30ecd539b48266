#!/usr/bin/env python3
"""Differential fuzz of the two launch choices round 6 added, beyond the suite's seeds: the forward with four rows per thread
(knob fwd_rows = 4 forced: full tiles, aligned or not, and ragged lengths on the EDGE instance) and the fused backward step with
2 / 4 / 8 interleaved fronts, random shapes, against the CPU oracle (forward and dV bit for bit, dW <= 1e-5).
    python profiles/fuzz_new_paths.py [cases]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sparsefactorization_amd as sfa  # noqa: E402
from sparsefactorization_amd.chord import _launch_bwd  # noqa: E402
from oracle import chord_oracle as oc  # noqa: E402  (a lab script: the oracle is the checker here, as in tests/)

dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(2026)
t0 = time.time()
bad = 0
for i in range(cases):
    # forward, four rows per thread: rows of 16 / 32 / 64 channels; N a multiple of the four-row tile, of the two-row tile only, or ragged
    C = int(rng.choice([16, 32, 64]))
    tr4 = {16: 256, 32: 128, 64: 64}[C]
    kind = int(rng.integers(0, 3))
    N = int(rng.integers(2, 40)) * tr4 if kind == 0 else (int(rng.integers(4, 80)) * (tr4 // 2) if kind == 1 else int(rng.integers(2 * tr4, 6000)))
    L = int(rng.integers(4, 21))
    B = int(rng.integers(1, 4))
    W = (0.5 * rng.standard_normal((B, N, L))).astype(np.float32)
    V = rng.standard_normal((B, N, C)).astype(np.float32)
    R = rng.standard_normal((B, N, C)).astype(np.float32) if rng.integers(0, 2) else None
    want = oc.spmul_fwd(W, V) + (R if R is not None else 0)
    sfa.set_tuning("fwd_rows", 4)
    got = sfa.chord_spmm(torch.from_numpy(W).to(dev), torch.from_numpy(V).to(dev), None if R is None else torch.from_numpy(R).to(dev)).cpu().numpy()
    sfa.set_tuning("fwd_rows", 0)
    if not np.array_equal(got, want.astype(np.float32)):
        bad += 1
        print(f"FORWARD FAIL B={B} N={N} L={L} C={C} residual={R is not None}", flush=True)
    # fused backward, interleaved fronts: rows of 4 / 8 / 16 / 32 channels, N a multiple of fronts x tile
    C = int(rng.choice([4, 8, 16, 32]))
    tr = 256 // (C // 4)
    fronts = int(rng.choice([2, 4, 8]))
    L = int(rng.integers(4, 21))
    N = int(rng.integers(1, 12)) * fronts * tr
    while (1 << (L - 2)) % tr and L > 4:  # far offsets must be multiples of the tile (else the two-kernel / edge path: not the subject)
        L -= 1
    B = int(rng.integers(1, 4))
    if (B * N * L) % 4:
        B = 4
    W = (0.5 * rng.standard_normal((B, N, L))).astype(np.float32)
    V = rng.standard_normal((B, N, C)).astype(np.float32)
    dZ = rng.standard_normal((B, N, C)).astype(np.float32)
    dF, dV = oc.spmul_bwd(dZ, W, V)
    gW = torch.full((B, N, L), float("nan"), device=dev)
    gV = torch.full((B, N, C), float("nan"), device=dev)
    sfa.set_tuning("bwd_fronts", fronts)
    _launch_bwd(torch.from_numpy(dZ).to(dev), torch.from_numpy(W).to(dev), torch.from_numpy(V).to(dev), gW, gV, B, N, L, C, N * C, None)
    sfa.set_tuning("bwd_fronts", 0)
    ok = np.array_equal(gV.cpu().numpy(), dV) and float(np.abs(gW.cpu().numpy() - dF).max() / max(np.abs(dF).max(), 1e-30)) <= 1e-5
    if not ok:
        bad += 1
        print(f"BACKWARD FAIL B={B} N={N} L={L} C={C} fronts={fronts}", flush=True)
print(f"{cases} forward cases with four rows per thread forced + {cases} fused backward cases with 2 / 4 / 8 fronts: {bad} failures, {time.time() - t0:.0f} s")
