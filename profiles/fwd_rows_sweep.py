#!/usr/bin/env python3
"""Forward window kernel: rows per thread (knob fwd_rows: 2, 4) x workgroups per CU (fwd_wg_limit: 0 = auto, 2, 3, 4), chains
that keep every step (training) with W rotating beyond the Infinity Cache; us per step, median of five, arms interleaved,
outputs compared bit for bit. A lab build (PSF_HIPCC_EXTRA="-DPSF_ROWS4_TGS_MIN=1 -DPSF_ROWS4_TGS_MAX=5") compiles R = 4 for
more widths than the product.   python profiles/fwd_rows_sweep.py"""
import os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsefactorization_amd as sfa  # noqa: E402
from sparsefactorization_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
SHAPES = [(64, 16384, 15, 8), (40, 16384, 15, 8), (64, 16384, 15, 16), (16, 16384, 15, 32), (32, 4096, 13, 32), (64, 1024, 12, 32),
          (8, 16384, 15, 64), (32, 2048, 12, 64), (8, 16384, 15, 128)]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for B, N, L, C in SHAPES:
    M = L - 1
    g = torch.Generator(device=dev).manual_seed(1)
    sets = 1 if os.environ.get("PSF_SWEEP_RESIDENT") else max(2, min(12, int(640e6 // (M * 4 * B * N * L))))  # PSF_SWEEP_RESIDENT=1: one operand set (cache-resident: what a small model's step sees)
    Wsets = [[0.1 * torch.randn(B, N, L, device=dev, generator=g) for _ in range(M)] for _ in range(sets)]
    V0 = torch.randn(B, N, C, device=dev, generator=g)
    res = C <= 8
    arms = [(r, w) for r in (2, 4) for w in (0, 2, 3, 4)]
    ref, desc, times = None, {}, {a: [] for a in arms}
    it = [0]
    with torch.no_grad():
        sfa.set_tuning("chain_fused", 0)
        for a in arms:
            sfa.set_tuning("fwd_rows", a[0])
            sfa.set_tuning("fwd_wg_limit", a[1])
            desc[a] = _lib.describe_fwd(B, N, L, C)
            out = sfa.chord_chain(Wsets[0], V0, res)
            if ref is None:
                ref = out.clone()
            assert torch.equal(out, ref), a
        for rnd in range(5):
            for a in (arms if rnd % 2 == 0 else arms[::-1]):
                sfa.set_tuning("fwd_rows", a[0])
                sfa.set_tuning("fwd_wg_limit", a[1])
                for _ in range(2):
                    sfa.chord_chain(Wsets[it[0] % sets], V0, res)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(16):
                    it[0] += 1
                    sfa.chord_chain(Wsets[it[0] % sets], V0, res)
                e1.record()
                torch.cuda.synchronize()
                times[a].append(e0.elapsed_time(e1) / 16 / M * 1e3)
        sfa.set_tuning("fwd_rows", 0)
        sfa.set_tuning("fwd_wg_limit", 0)
        sfa.set_tuning("chain_fused", 1)
    r4 = "R=4" in desc[(4, 0)]
    print(f"B={B} N={N} L={L} C={C} ({sets} W sets; rows 4 {'compiled' if r4 else 'NOT compiled: same kernel'}): " +
          "  ".join(f"r{r}/wg{w}: {statistics.median(t):.2f}" for (r, w), t in times.items()), flush=True)
