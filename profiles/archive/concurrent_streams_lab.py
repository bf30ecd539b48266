#!/usr/bin/env python3
"""Could the producer MLPs (matrix / vector bound) and the chord chain (memory bound) share the chip? The existing kernels on
two HIP streams, independent operands, Temporal-Order training shape (B = 40, N = 16384, L = 15, C = 8, 15 MLPs 32-32-15):
time of each alone, of both back to back on one stream, and of both at once on two streams.  python profiles/concurrent_streams_lab.py"""
import os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsefactorization_amd as sfa  # noqa: E402
from sparsefactorization_amd import chord, fused_mlp  # noqa: E402
from sparsefactorization_amd.psfnet import MLPBlock  # noqa: E402

dev = torch.device("cuda:0")
B, N, L, C, M = 40, 16384, 15, 8, 14
torch.manual_seed(0)
blocks = [MLPBlock([32, 'GELU'], 32, C).to(dev)] + [MLPBlock([32, 'GELU'], 32, L).to(dev) for _ in range(M)]
x = torch.randn(B, N, 32, device=dev)
Ws = [0.1 * torch.randn(B, N, L, device=dev) for _ in range(M)]
Vs = [torch.randn(B, N, C, device=dev) for _ in range(M + 1)]
dWs = [torch.empty_like(Ws[0]) for _ in range(M)]
zz = [torch.randn(B, N, C, device=dev), torch.empty(B, N, C, device=dev)]
sA, sB = torch.cuda.Stream(dev), torch.cuda.Stream(dev)

x2 = x.reshape(B * N, 32)
params = fused_mlp._params_of(blocks)
gys = [torch.randn_like(y) for y in fused_mlp._forward_raw(x2, params)]


def producer_fwd():
    with torch.no_grad():
        fused_mlp.fused_mlp_forward(x, blocks)


def chain_fwd():
    with torch.no_grad():
        sfa.chord_chain(Ws, Vs[0], True)


def producer_bwd():
    with torch.no_grad():
        fused_mlp._backward_raw(x2, params, gys, True)


def chain_bwd():
    for m in range(M):
        chord._launch_bwd(zz[m & 1], Ws[m], Vs[m], dWs[m], zz[1 - (m & 1)], B, N, L, C, N * C, None)


def timed(fa, fb, mode, n=20):
    ts = []
    for _ in range(5):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            if mode == "a":
                fa()
            elif mode == "b":
                fb()
            elif mode == "seq":
                fa(); fb()
            else:
                ready = torch.cuda.Event(); ready.record()
                with torch.cuda.stream(sA):
                    sA.wait_event(ready); fa(); da = torch.cuda.Event(); da.record()
                with torch.cuda.stream(sB):
                    sB.wait_event(ready); fb(); db = torch.cuda.Event(); db.record()
                torch.cuda.current_stream().wait_event(da); torch.cuda.current_stream().wait_event(db)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n)
    return statistics.median(ts)


for name, fa, fb in (("forward: 15 producer MLPs | 14 chord steps", producer_fwd, chain_fwd),
                     ("backward: MLP backward | 14 fused backward steps", producer_bwd, chain_bwd)):
    for f in (fa, fb):
        for _ in range(3):
            f()
    r = {m: timed(fa, fb, m) for m in ("a", "b", "seq", "par")}
    print(f"{name}: producer alone {r['a']:.3f} ms, chain alone {r['b']:.3f} ms, one stream {r['seq']:.3f} ms, two streams {r['par']:.3f} ms "
          f"(of the sum {r['par'] / (r['a'] + r['b']):.2f}, of the larger {r['par'] / max(r['a'], r['b']):.2f})", flush=True)
