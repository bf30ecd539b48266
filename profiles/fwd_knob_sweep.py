#!/usr/bin/env python3
"""Forward chain (training mode: every output kept) over the launcher's knobs: fwd_wide x fwd_rows x fwd_wg_limit, us per
step-launch; settings alternate after a warm-up, median of five readings.   python profiles/fwd_knob_sweep.py B N M C res"""
import itertools
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsefactorization_amd as sfa  # noqa: E402

B, N, M, C, res = (int(a) for a in sys.argv[1:6]) if len(sys.argv) >= 6 else (32, 2000, 11, 128, 0)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
Ws = [(0.1 * torch.randn(B, N, M + 1, device=dev, generator=g)).requires_grad_(True) for _ in range(M)]
V0 = torch.randn(B, N, C, device=dev, generator=g).requires_grad_(True)
sfa.set_tuning("chain_fused", 0)  # per-step launches (what training uses at these sizes)


def reading(chains=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(chains):
        out = sfa.chord_chain(Ws, V0, bool(res))
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / chains / M * 1e3, out


reading(100)
settings = list(itertools.product((0, 1, 2), (0, 1, 2), (0, 2, 3, 4)))
times = {s: [] for s in settings}
desc = {}
ref = None
for rnd in range(5):
    for s in (settings if rnd % 2 == 0 else settings[::-1]):
        sfa.set_tuning("fwd_wide", s[0])
        sfa.set_tuning("fwd_rows", s[1])
        sfa.set_tuning("fwd_wg_limit", s[2])
        desc[s] = sfa.describe_fwd(B, N, M + 1, C)
        t, out = reading()
        times[s].append(t)
        if ref is None:
            ref = out.detach().clone()
        assert torch.equal(out, ref)
for k in ("fwd_wide", "fwd_rows", "fwd_wg_limit"):
    sfa.set_tuning(k, 0)
sfa.set_tuning("chain_fused", 1)
print(f"B={B} N={N} M={M} C={C} residual={res}")
for s in sorted(settings, key=lambda s: statistics.median(times[s])):
    print(f"wide={s[0]} rows={s[1]} wg_limit={s[2]}: median {statistics.median(times[s]):6.2f} us  {desc[s]}")
