#!/usr/bin/env python3
"""Where the cold-operand penalty of the fused backward step comes from: rotate (10 sets) only the read operands W and V,
only dZ, only the outputs, everything, or nothing; us per step, median of five readings of 100 steps, alternating.
    python profiles/bwd_fused_coldness.py [B N L C]"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsefactorization_amd as sfa  # noqa: E402
from sparsefactorization_amd import chord  # noqa: E402

B, N, L, C = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (40, 16384, 15, 8)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
sets = 10
Ws = [0.1 * torch.randn(B, N, L, device=dev, generator=g) for _ in range(sets)]
Vs = [torch.randn(B, N, C, device=dev, generator=g) for _ in range(sets)]
dZs = [torch.randn(B, N, C, device=dev, generator=g) for _ in range(sets)]
dWs = [torch.empty_like(Ws[0]) for _ in range(sets)]
dVs = [torch.empty_like(Vs[0]) for _ in range(sets)]
MODES = {"nothing rotates (warm)": (0, 0, 0, 0), "W rotates": (1, 0, 0, 0), "V rotates": (0, 1, 0, 0), "dZ rotates": (0, 0, 1, 0),
         "dW, dV (outputs) rotate": (0, 0, 0, 1), "W, V rotate; dZ, outputs fixed": (1, 1, 0, 0),
         "W, V, outputs rotate; dZ fixed (a training step)": (1, 1, 0, 1), "everything rotates": (1, 1, 1, 1)}


def reading(mode, steps=100):
    rw, rv, rz, ro = mode
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for i in range(steps):
        s = i % sets
        chord._launch_bwd(dZs[s * rz], Ws[s * rw], Vs[s * rv], dWs[s * ro], dVs[s * ro], B, N, L, C, N * C, None)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3


reading(MODES["everything rotates"], 300)
times = {k: [] for k in MODES}
names = list(MODES)
for rnd in range(5):
    for k in (names if rnd % 2 == 0 else names[::-1]):
        times[k].append(reading(MODES[k]))
print(f"B={B} N={N} L={L} C={C}: W {B*N*L*4/1e6:.0f} MB, V = dZ = dV {B*N*C*4/1e6:.0f} MB, dW {B*N*L*4/1e6:.0f} MB per step")
for k in names:
    print(f"{k:52s} median {statistics.median(times[k]):.2f} us  {['%.2f' % t for t in times[k]]}")
