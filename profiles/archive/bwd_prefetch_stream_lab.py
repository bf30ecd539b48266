#!/usr/bin/env python3
"""Can a second stream warm the NEXT backward step's cold operands (W, V: 60 MB at Order B = 40) while this step's fused
kernel runs? A reduction over the next set's W and V (torch.sum: a pure read stream) is launched on a side stream right
before each step; us per step on the main stream (events around 100 steps, all work joined at the end), against the same
loop without the toucher. Operand sets rotate (10), outputs rotate, dZ fixed: the training step's situation.
    python profiles/bwd_prefetch_stream_lab.py"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsefactorization_amd import chord  # noqa: E402

B, N, L, C = 40, 16384, 15, 8
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
sets = 10
Ws = [0.1 * torch.randn(B, N, L, device=dev, generator=g) for _ in range(sets)]
Vs = [torch.randn(B, N, C, device=dev, generator=g) for _ in range(sets)]
dZ = torch.randn(B, N, C, device=dev, generator=g)
dWs = [torch.empty_like(Ws[0]) for _ in range(sets)]
dVs = [torch.empty_like(Vs[0]) for _ in range(sets)]
side = torch.cuda.Stream()
main = torch.cuda.current_stream()
sink = torch.zeros(2, device=dev)


def reading(touch, steps=100):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for i in range(steps):
        s = i % sets
        if touch:
            nxt = (i + 1) % sets
            ev = torch.cuda.Event()
            ev.record(main)          # the toucher of step i + 1 starts when step i is about to
            side.wait_event(ev)
            with torch.cuda.stream(side):
                if touch & 1:
                    sink[0] = Ws[nxt].sum()
                if touch & 2:
                    sink[1] = Vs[nxt].sum()
        chord._launch_bwd(dZ, Ws[s], Vs[s], dWs[s], dVs[s], B, N, L, C, N * C, None)
    main.wait_stream(side)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3


reading(0, 300)
modes = {"no toucher": 0, "touch next W": 1, "touch next V": 2, "touch next W and V": 3}
times = {k: [] for k in modes}
names = list(modes)
for rnd in range(5):
    for k in (names if rnd % 2 == 0 else names[::-1]):
        times[k].append(reading(modes[k]))
for k in names:
    print(f"{k:22s} median {statistics.median(times[k]):.2f} us per step  {['%.2f' % t for t in times[k]]}")
