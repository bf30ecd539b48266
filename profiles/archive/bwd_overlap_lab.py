#!/usr/bin/env python3
"""Can the MLP backward run beside the chord backward chain? The chain's kernels are latency / memory bound (vector unit ~25 %
busy, matrix pipe idle), the MLP backward is vector / matrix bound with the memory system idle; MLP m's backward needs dW_m only,
which the chain produces first for the LAST step. Temporal-Order shape (B = 40, N = 16384, 15 MLPs of 32 hidden units):
  A   = 14 fused backward steps (operands rotating as in a training step),
  B   = the MLP backward of all 15 MLPs in one launch,   B3 = the same in three launches of five MLPs (dX added up),
alone, one after the other, and on two streams. us, median of seven.   python profiles/bwd_overlap_lab.py"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsefactorization_amd as sfa  # noqa: E402
from sparsefactorization_amd import chord, fused_mlp  # noqa: E402
from sparsefactorization_amd.psfnet import MLPBlock  # noqa: E402

dev = torch.device("cuda:0")
B, N, L, C, M = 40, 16384, 15, 8, 14
g = torch.Generator(device=dev).manual_seed(0)
Ws = [0.1 * torch.randn(B, N, L, device=dev, generator=g) for _ in range(M)]
Vs = [torch.randn(B, N, C, device=dev, generator=g) for _ in range(M)]
dWs = [torch.empty_like(Ws[0]) for _ in range(M)]
dZ = [torch.randn(B, N, C, device=dev, generator=g), torch.empty(B, N, C, device=dev)]
torch.manual_seed(0)
blocks = [MLPBlock([32, 'GELU'], 32, C).to(dev)] + [MLPBlock([32, 'GELU'], 32, L).to(dev) for _ in range(M)]
params = [p.detach().contiguous() for p in fused_mlp._params_of(blocks)]
x2 = torch.randn(B * N, 32, device=dev, generator=g)
gys = [torch.randn(B * N, C, device=dev, generator=g)] + [w.reshape(B * N, L) for w in dWs]
for w in dWs:
    w.normal_(generator=g)


def chain_bwd():
    for m in reversed(range(M)):
        chord._launch_bwd(dZ[m & 1], Ws[m], Vs[m], dWs[m], dZ[(m & 1) ^ 1], B, N, L, C, N * C, None)


def mlp_bwd(groups):
    dX = None
    for grp in groups:
        sub = [params[4 * k + i] for k in grp for i in range(4)]
        d, _ = fused_mlp._backward_raw(x2, sub, [gys[k] for k in grp], True)
        dX = d if dX is None else dX.add_(d)
    return dX


ALL = [list(range(15))]
THREE = [list(range(10, 15)), list(range(5, 10)), list(range(0, 5))]
side = torch.cuda.Stream(device=dev)


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def both(groups):
    main = torch.cuda.current_stream(dev)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        mlp_bwd(groups)
    chain_bwd()
    main.wait_stream(side)


ARMS = {"A: chain backward alone": chain_bwd, "B: MLP backward, one launch": lambda: mlp_bwd(ALL),
        "B3: MLP backward, three launches": lambda: mlp_bwd(THREE),
        "A then B (one stream)": lambda: (chain_bwd(), mlp_bwd(ALL)),
        "A beside B (two streams)": lambda: both(ALL), "A beside B3 (two streams)": lambda: both(THREE)}
for lim in (0, 3, 2):
    sfa.set_tuning("bwd_fused_wg_limit", lim)
    times = {k: [] for k in ARMS}
    for rnd in range(7):
        for k in (list(ARMS) if rnd % 2 == 0 else list(ARMS)[::-1]):
            times[k].append(timed(ARMS[k]))
    print(f"fused backward step: workgroups per CU {'as many as fit' if lim == 0 else '<= %d' % lim}")
    for k in ARMS:
        print(f"  {k:36s} {statistics.median(times[k]):8.1f} us")
sfa.set_tuning("bwd_fused_wg_limit", 0)
