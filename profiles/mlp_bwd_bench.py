#!/usr/bin/env python3
"""Fused producer-MLP backward (csrc/mlp_bwd.hip): the all-f32-MFMA kernel (mlp_bwd_variant=1) vs the split-bf16 kernels
(2: steps 1, 2, 6 on the bf16 matrix pipe; 3 = auto: all steps, operands on dual-use LDS planes), interleaved in one process, at the Temporal-Order training shape
(E = h = 32, g with 8 outputs + 14 link MLPs with 15, T = B*N tokens) and at the Pathfinder shape (E = 32, h = 128).
Gradients of both variants are compared with float64 autograd through nn modules on a slice of the tokens.

    python profiles/mlp_bwd_bench.py [--tokens 655360] [--rounds 5] [--iters 10]
"""
import argparse
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sparsefactorization_amd as sfa  # noqa: E402
from sparsefactorization_amd import fused_mlp  # noqa: E402


LABELS = {0: "auto", 1: "f32-MFMA", 2: "split-bf16 steps 1,2,6", 3: "split-bf16 on dual-use planes"}


def make(E, h, outs, dev, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed)
    params = []
    for O in outs:
        params += [torch.randn(h, E, device=dev, generator=g) / E ** 0.5, 0.1 * torch.randn(h, device=dev, generator=g),
                   torch.randn(O, h, device=dev, generator=g) / h ** 0.5, 0.1 * torch.randn(O, device=dev, generator=g)]
    return params


def reference(x, params, gys):
    """float64 autograd through the same math: Y_k = GELU(x A^T + a) B^T + b."""
    xd = x.double().requires_grad_(True)
    pd = [p.double().requires_grad_(True) for p in params]
    total = 0
    for k in range(len(params) // 4):
        A, a, B, b = pd[4 * k:4 * k + 4]
        y = torch.nn.functional.gelu(xd @ A.t() + a) @ B.t() + b
        total = total + (y * gys[k].double()).sum()
    total.backward()
    return xd.grad, [p.grad for p in pd]


def rel(a, b):
    return float((a.double() - b).abs().max() / b.abs().max().clamp_min(1e-30))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", type=int, default=40 * 16384)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--variants", type=int, nargs="+", default=[1, 2, 0],
                    help="mlp_bwd_variant values to interleave; speed-ups are relative to the first")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    for name, E, h, outs, T in (("order_n16384_b40", 32, 32, [8] + [15] * 14, args.tokens),
                                ("pathfinder_b64", 32, 128, [12] * 11, 64 * 1024),
                                ("wide_out_o32", 32, 32, [32] * 4, args.tokens)):
        params = make(E, h, outs, dev)
        g = torch.Generator(device=dev).manual_seed(1)
        x = torch.randn(T, E, device=dev, generator=g)
        gys = [torch.randn(T, O, device=dev, generator=g) for O in outs]
        # accuracy on the first 4096 tokens
        n = 4096
        dx_ref, gp_ref = reference(x[:n], params, [gy[:n] for gy in gys])
        errs = {}
        VAR = tuple(args.variants)
        for v in VAR:
            sfa.set_tuning("mlp_bwd_variant", v)
            dX, grads = fused_mlp._backward_raw(x[:n].contiguous(), params, [gy[:n].contiguous() for gy in gys], True)
            errs[v] = max([rel(dX, dx_ref)] + [rel(gq, gr) for gq, gr in zip(grads, gp_ref)])
            dX2, grads2 = fused_mlp._backward_raw(x[:n].contiguous(), params, [gy[:n].contiguous() for gy in gys], True)
            assert torch.equal(dX, dX2) and all(torch.equal(p, q) for p, q in zip(grads, grads2)), "not bit-reproducible"
        times = {v: [] for v in VAR}
        for r in range(args.rounds + 1):
            for v in VAR:
                sfa.set_tuning("mlp_bwd_variant", v)
                fused_mlp._backward_raw(x, params, gys, True)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(args.iters):
                    fused_mlp._backward_raw(x, params, gys, True)
                e1.record()
                torch.cuda.synchronize()
                if r:
                    times[v].append(e0.elapsed_time(e1) / args.iters)
        sfa.set_tuning("mlp_bwd_variant", 0)
        med = {v: statistics.median(times[v]) for v in VAR}
        base = med[VAR[0]]
        print(f"{name}: T={T} E={E} h={h} K={len(outs)}   " + "   ".join(
            f"{LABELS.get(v, v)} {med[v]:.3f} ms (max rel err {errs[v]:.2e}, {base / med[v]:.2f}x)" for v in VAR), flush=True)


if __name__ == "__main__":
    main()
