#!/usr/bin/env python3
"""What the fused backward step's time is made of: the cfg2 / Temporal-Order instance of chord_bwd_fused_k with parts left
out (csrc/bwd_fused.h, ABL; needs a library built with PSF_HIPCC_EXTRA=-DPSF_BWD_ABLATE_LAB), warm operands (one set) and
cold ones (10 sets, W / V / outputs rotating, dZ fixed: what a training step sees). us per step, median of five readings of
100 steps, arms interleaved.
    python profiles/bwd_ablate_bench.py [B N L C]"""
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
dZ = torch.randn(B, N, C, device=dev, generator=g)
dWs = [torch.empty_like(Ws[0]) for _ in range(sets)]
dVs = [torch.empty_like(Vs[0]) for _ in range(sets)]
ARMS = [(0, "everything"), (2, "no far W elements"), (1, "no far dZ / V rows"), (3, "no far loads at all"),
        (4, "no dZ / V windows"), (8, "no W tiles"), (12, "no staged operands"), (15, "no loads"),
        (16, "no dV arithmetic"), (32, "no dW gathers"), (48, "no arithmetic on staged operands"),
        (64, "no stores"), (79, "no loads, no stores"), (112, "loads only"), (115, "staged loads only"),
        (124, "far loads only"), (63, "stores only"),
        (128, "variant: dW stored non-temporally"), (512, "variant: dW dots as FMAs"), (640, "variant: both"),
        (256, "variant: far dZ / V rows by LDS-DMA"), (258, "the same, no far W elements"), (320, "the same, no stores")]
if len(sys.argv) > 5:
    keep = {int(a) for a in sys.argv[5].split(",")}
    ARMS = [a for a in ARMS if a[0] in keep]


def reading(abl, cold, steps=100):
    sfa.set_tuning("bwd_ablate", abl)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for i in range(steps):
        s = i % sets if cold else 0
        chord._launch_bwd(dZ, Ws[s], Vs[s], dWs[s], dVs[s], B, N, L, C, N * C, None)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3


from sparsefactorization_amd import _lib  # noqa: E402
lib = _lib.load()
nv = B * N * C // 4
# the same operand sets through the plain streaming kernel with the backward step's byte mix (W rows of 15 floats here, 16 in
# the stream: its W-like streams are 2 vectors per V vector, so it moves 112 B per 4 channels against the step's 108)
sW = [torch.empty(2 * nv * 4, device=dev).fill_(0.5) for _ in range(sets)]
sdW = [torch.empty(2 * nv * 4, device=dev) for _ in range(sets)]


def stream_reading(cold, steps=100):
    st = torch.cuda.current_stream().cuda_stream
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for i in range(steps):
        s = i % sets if cold else 0
        _lib.check(lib.psf_stream_mix_bwd_f32(sW[s].data_ptr(), Vs[s].data_ptr(), dZ.data_ptr(), sdW[s].data_ptr(),
                                              dVs[s].data_ptr(), nv, st), "stream")
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3


print(sfa.build_info().split("csrc=")[-1])
reading(0, True, 300)
times = {(a, c): [] for a, _ in ARMS for c in (False, True)}
for rnd in range(5):
    order = ARMS if rnd % 2 == 0 else ARMS[::-1]
    for a, _ in order:
        for c in (False, True):
            times[(a, c)].append(reading(a, c))
sfa.set_tuning("bwd_ablate", 0)
for cold in (False, True):
    ts = [stream_reading(cold) for _ in range(5)]
    t = statistics.median(ts)
    print(f"stream kernel, backward mix ({112 * nv / 1e6:.1f} MB): {'cold' if cold else 'warm'} {t:.2f} us = {112 * nv / t / 1e6:.2f} TB/s")
print(f"B={B} N={N} L={L} C={C}; algorithmic bytes per step {4 * B * N * (2 * L + 3 * C) / 1e6:.1f} MB")
for a, name in ARMS:
    print(f"ABL={a:3d} {name:36s} warm {statistics.median(times[(a, False)]):6.2f} us   cold {statistics.median(times[(a, True)]):6.2f} us")
