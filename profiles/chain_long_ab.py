#!/usr/bin/env python3
"""The one-launch chain for 2113 <= N <= 4160 (chord_chain_rows_k with one channel group and five rows per thread; knob
chain_cc = 2 forces it) against M per-step launches (chain_fused = 0); every step kept (training) and last kept (inference),
W rotating; us per step, arms interleaved, results compared bit for bit.   python profiles/chain_long_ab.py [BxNxLxC ...]"""
import os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsefactorization_amd as sfa  # noqa: E402
from sparsefactorization_amd import chord, _lib  # noqa: E402

dev = torch.device("cuda:0")
SHAPES = [(32, 4097, 14, 32), (32, 4096, 13, 32), (64, 4097, 14, 32), (32, 4097, 14, 64), (32, 4097, 14, 8), (16, 4097, 14, 32), (32, 3000, 13, 32), (64, 4096, 13, 8)]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
ARMS = {"steps": (0, 0), "one launch": (2, 2)}
for B, N, L, C in SHAPES:
    M = L - 2 if N == 4097 else L - 1
    g = torch.Generator(device=dev).manual_seed(1)
    sets = max(2, min(12, int(640e6 // (M * 4 * B * N * L))))
    Wsets = [[0.1 * torch.randn(B, N, L, device=dev, generator=g) for _ in range(M)] for _ in range(sets)]
    V0 = torch.randn(B, N, C, device=dev, generator=g)
    res = True
    line = []
    sfa.set_tuning("chain_fused", 2)
    sfa.set_tuning("chain_cc", 0)
    auto = "rows_k" in _lib.describe_chain_fwd(B, N, L, C, M)
    for keep in (True, False):
        times, it, ref = {k: [] for k in ARMS}, [0], None
        with torch.no_grad():
            for rnd in range(5):
                order = list(ARMS) if rnd % 2 == 0 else list(ARMS)[::-1]
                for arm in order:
                    cf, cc = ARMS[arm]
                    sfa.set_tuning("chain_fused", cf)
                    sfa.set_tuning("chain_cc", cc)
                    last = chord._chain_forward_raw(V0, res, None, Wsets[0], keep)[2][-1]
                    if ref is None:
                        ref = last.clone()
                    assert torch.equal(last, ref), (arm, keep)
                    chord._chain_forward_raw(V0, res, None, Wsets[it[0] % sets], keep)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    torch.cuda.synchronize()
                    e0.record()
                    for _ in range(16):
                        it[0] += 1
                        chord._chain_forward_raw(V0, res, None, Wsets[it[0] % sets], keep)
                    e1.record()
                    torch.cuda.synchronize()
                    times[arm].append(e0.elapsed_time(e1) / 16 / M * 1e3)
        line.append(("every step kept" if keep else "last kept") + ": " + " / ".join(f"{statistics.median(times[a]):.2f}" for a in ARMS))
    sfa.set_tuning("chain_fused", 1)
    sfa.set_tuning("chain_cc", 0)
    print(f"B={B} N={N} L={L} C={C} M={M} (per-step launches / one launch, us per step; automatic: {'one launch' if auto else 'steps'}): " + "   ".join(line), flush=True)
