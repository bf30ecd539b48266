#!/usr/bin/env python3
"""The headline: forward chain in inference mode (two output buffers) over fwd_rows x fwd_wg_limit, us per step-launch;
settings alternate, median of seven readings of 20 chains.   python profiles/fwd_infer_knob_sweep.py [B N M C res]"""
import itertools
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsefactorization_amd as sfa  # noqa: E402

B, N, M, C, res = (int(a) for a in sys.argv[1:6]) if len(sys.argv) >= 6 else (64, 16384, 14, 8, 1)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
Ws = [0.1 * torch.randn(B, N, M + 1, device=dev, generator=g) for _ in range(M)]
V0 = torch.randn(B, N, C, device=dev, generator=g)


def reading(chains=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    with torch.no_grad():
        for _ in range(chains):
            out = sfa.chord_chain(Ws, V0, bool(res))
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / chains / M * 1e3, out


reading(50)
settings = list(itertools.product((1, 2), (0, 1, 2, 3, 4)))
times = {s: [] for s in settings}
desc, ref = {}, None
for rnd in range(7):
    for s in (settings if rnd % 2 == 0 else settings[::-1]):
        sfa.set_tuning("fwd_rows", s[0])
        sfa.set_tuning("fwd_wg_limit", s[1])
        desc[s] = sfa.describe_fwd(B, N, M + 1, C)
        t, out = reading()
        times[s].append(t)
        if ref is None:
            ref = out.clone()
        assert torch.equal(out, ref)
for k in ("fwd_rows", "fwd_wg_limit"):
    sfa.set_tuning(k, 0)
print(f"B={B} N={N} M={M} C={C} residual={res}")
for s in settings:
    lim = {0: "auto", 1: "no limit"}.get(s[1], f"<= {s[1]} per CU")
    print(f"rows per thread {s[0]}, workgroups {lim:12s}: median {statistics.median(times[s]):6.2f} us per step   {desc[s]}")
