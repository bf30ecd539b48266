#!/usr/bin/env python3
"""Lab (library built with PSF_HIPCC_EXTRA=-DPSF_FWD_R4_LAB): the forward window kernel with FOUR rows per thread for rows of
16 / 32 channels (tile 256 / 128 rows at 256 threads: one far link fewer) against the shipped two rows per thread. The lab
selects R = 4 with knob fwd_wg_limit = 4. us per step of a rotating chain, arms interleaved, outputs compared bit for bit.
    python profiles/fwd_r4_bench.py"""
import os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsefactorization_amd as sfa  # noqa: E402
from sparsefactorization_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
for B, N, L, C in ((16, 16384, 15, 32), (32, 4096, 13, 32), (64, 1024, 12, 32), (64, 16384, 15, 16), (64, 16384, 15, 32)):
    M = L - 1
    g = torch.Generator(device=dev).manual_seed(1)
    sets = max(2, min(12, int(640e6 // (M * 4 * B * N * L))))
    Wsets = [[0.1 * torch.randn(B, N, L, device=dev, generator=g) for _ in range(M)] for _ in range(sets)]
    V0 = torch.randn(B, N, C, device=dev, generator=g)
    arms = {"auto": 0, "three per CU, R=2": 3, "R=4 (lab)": 4}
    ref, names, times = None, {}, {a: [] for a in arms}
    it = [0]
    with torch.no_grad():
        sfa.set_tuning("chain_fused", 0)
        for a, v in arms.items():
            sfa.set_tuning("fwd_wg_limit", v)
            names[a] = _lib.describe_fwd(B, N, L, C)
            out = sfa.chord_chain(Wsets[0], V0, False)
            if ref is None:
                ref = out.clone()
            assert torch.equal(out, ref), a
        for rnd in range(7):
            for a in (list(arms) if rnd % 2 == 0 else list(arms)[::-1]):
                sfa.set_tuning("fwd_wg_limit", arms[a])
                for _ in range(2):
                    sfa.chord_chain(Wsets[it[0] % sets], V0, False)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(20):
                    it[0] += 1
                    sfa.chord_chain(Wsets[it[0] % sets], V0, False)
                e1.record()
                torch.cuda.synchronize()
                times[a].append(e0.elapsed_time(e1) / 20 / M * 1e3)
        sfa.set_tuning("fwd_wg_limit", 0)
        sfa.set_tuning("chain_fused", 1)
    alg = 4 * B * N * (L + 2 * C)
    print(f"B={B} N={N} L={L} C={C} ({sets} W sets): " + "   ".join(f"{a}: {statistics.median(t):.2f} us ({alg / statistics.median(t) / 8e6:.3f})" for a, t in times.items()), flush=True)
    print("      R=4 arm runs:", names["R=4 (lab)"], flush=True)
