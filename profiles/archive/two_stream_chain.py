#!/usr/bin/env python3
"""Experiment: the cfg2 forward chain as ONE chain of 64 sequences vs 2 / 4 independent sub-chains (batch halves /
quarters) on separate HIP streams, so that one sub-chain's kernel tails overlap with another's kernel heads.
    python profiles/two_stream_chain.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsefactorization_amd as sfa  # noqa: E402

dev = torch.device("cuda:0")
B, N, M, C = 64, 16384, 14, 8
L = M + 1
g = torch.Generator(device=dev).manual_seed(0)
Ws = [0.1 * torch.randn(B, N, L, device=dev, generator=g) for _ in range(M)]
V0 = torch.randn(B, N, C, device=dev, generator=g)


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


import statistics
import time

with torch.no_grad():
    ref = sfa.chord_chain(Ws, V0, True)
    variants = {"one chain of 64": lambda: sfa.chord_chain(Ws, V0, True)}
    for parts in (2, 4):
        for limit in (0, 3):  # forward kernel's workgroups-per-CU knob (auto caps at 3 only for >= 4096 tiles)
            streams = [torch.cuda.Stream(device=dev) for _ in range(parts)]
            per = B // parts
            subW = [[w[i * per:(i + 1) * per] for w in Ws] for i in range(parts)]
            subV = [V0[i * per:(i + 1) * per] for i in range(parts)]

            def run(streams=streams, subW=subW, subV=subV, limit=limit, parts=parts):
                sfa.set_tuning("fwd_wg_limit", limit)
                cur = torch.cuda.current_stream(dev)
                outs = []
                for i, s in enumerate(streams):
                    s.wait_stream(cur)
                    with torch.cuda.stream(s):
                        outs.append(sfa.chord_chain(subW[i], subV[i], True))
                for s in streams:
                    cur.wait_stream(s)
                sfa.set_tuning("fwd_wg_limit", 0)
                return outs

            assert torch.equal(torch.cat(run()), ref)
            variants[f"{parts} sub-chains on {parts} streams, wg_limit={limit}"] = run
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.5:  # clocks up
        sfa.chord_chain(Ws, V0, True)
    torch.cuda.synchronize()
    samples = {k: [] for k in variants}
    for _ in range(7):
        for k, fn in variants.items():
            samples[k].append(timeit(fn, 40))
    for k, v in samples.items():
        print(f"{k:50s} {statistics.median(v):7.1f} us/chain (min {min(v):.1f})")
