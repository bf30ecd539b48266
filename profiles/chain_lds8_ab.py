#!/usr/bin/env python3
"""The one-launch chain for 1057 <= N <= 2048 with TWO channel groups per workgroup (chord_chain_lds8_k, knob chain_cc = 2)
against one group per workgroup (chain_cc = 1, what ran there before) and against M per-step launches (chain_fused = 0);
every step kept (training) and last kept (inference), W rotating; us per step, arms interleaved, results compared bit for
bit.   python profiles/chain_lds8_ab.py [BxNxLxC ...]"""
import os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsefactorization_amd as sfa  # noqa: E402
from sparsefactorization_amd import chord, _lib  # noqa: E402

dev = torch.device("cuda:0")
SHAPES = [(32, 2000, 12, 128), (32, 2048, 12, 64), (32, 2000, 12, 64), (32, 2048, 12, 32), (64, 2048, 12, 16), (64, 2048, 12, 8),
          (128, 2048, 12, 8), (32, 2001, 12, 128), (32, 1500, 12, 32), (16, 2000, 12, 128), (8, 2000, 12, 128)]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
ARMS = {"steps": (0, 0), "one group": (2, 1), "two groups": (2, 2)}
for B, N, L, C in SHAPES:
    M = L - 1
    g = torch.Generator(device=dev).manual_seed(1)
    sets = max(2, min(12, int(640e6 // (M * 4 * B * N * L))))
    Wsets = [[0.1 * torch.randn(B, N, L, device=dev, generator=g) for _ in range(M)] for _ in range(sets)]
    V0 = torch.randn(B, N, C, device=dev, generator=g)
    res = C <= 8
    line = []
    sfa.set_tuning("chain_fused", 2)
    sfa.set_tuning("chain_cc", 0)
    auto = "lds8" in _lib.describe_chain_fwd(B, N, L, C, M)
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
    print(f"B={B} N={N} L={L} C={C} (steps / one group / two groups, us per step; automatic: {'two' if auto else 'one'}): " + "   ".join(line), flush=True)
