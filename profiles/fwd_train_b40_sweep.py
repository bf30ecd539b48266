#!/usr/bin/env python3
"""Forward chain as the Temporal-Order training step runs it (B = 40, N = 16384, M = 14, L = 15, C = 8, residual, every
output kept): rows per thread (fwd_rows) x workgroups per CU (fwd_wg_limit), us per step-launch; settings alternate after a
warm-up, median of five readings of 10 chains.   python profiles/fwd_train_b40_sweep.py [B]"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsefactorization_amd as sfa  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 40
N, M, C = 16384, 14, 8
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
Ws = [(0.1 * torch.randn(B, N, M + 1, device=dev, generator=g)).requires_grad_(True) for _ in range(M)]
V0 = torch.randn(B, N, C, device=dev, generator=g).requires_grad_(True)


def reading(chains=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(chains):
        out = sfa.chord_chain(Ws, V0, True)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / chains / M * 1e3, out


reading(100)
settings = [(r, w) for r in (0, 1, 2) for w in (0, 2, 3, 4)]
times = {s: [] for s in settings}
ref = None
for rnd in range(5):
    for s in (settings if rnd % 2 == 0 else settings[::-1]):
        sfa.set_tuning("fwd_rows", s[0])
        sfa.set_tuning("fwd_wg_limit", s[1])
        t, out = reading()
        times[s].append(t)
        if ref is None:
            ref = out.detach().clone()
        assert torch.equal(out, ref)
sfa.set_tuning("fwd_rows", 0)
sfa.set_tuning("fwd_wg_limit", 0)
print(f"B={B}: {sfa.describe_fwd(B, N, M + 1, C)}")
for s in settings:
    print(f"fwd_rows={s[0]} fwd_wg_limit={s[1]}: median {statistics.median(times[s]):.2f} us per launch  {['%.2f' % t for t in times[s]]}")
