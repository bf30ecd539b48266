#!/usr/bin/env python3
"""The backward step as two kernels (dW, then dV) vs the fused kernel (csrc/bwd_fused.h), with and without the link-major
side copy of W's far columns; warm operands (one set, repeated) and cold ones (10 operand sets walked round-robin, as a
training step sees them). us per step-pair, bit / tolerance check of the fused kernel against the two-kernel path.

    python profiles/bwd_fused_bench.py [B N L C]          default cfg2: 64 16384 15 8
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsefactorization_amd as sfa  # noqa: E402
from sparsefactorization_amd import _lib, chord  # noqa: E402


def main():
    B, N, L, C = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (64, 16384, 15, 8)
    dev = torch.device("cuda:0")
    lib = _lib.load()
    g = torch.Generator(device=dev).manual_seed(0)
    sets = 10
    Ws = [0.1 * torch.randn(B, N, L, device=dev, generator=g) for _ in range(sets)]
    Vs = [torch.randn(B, N, C, device=dev, generator=g) for _ in range(sets)]
    dZs = [torch.randn(B, N, C, device=dev, generator=g) for _ in range(sets)]
    k0 = lib.psf_chord_bwd_far_first_link(B, N, L, C)
    fars = [w[:, :, k0:].permute(0, 2, 1).contiguous() if 0 <= k0 < L else None for w in Ws]
    dV, dW = torch.empty_like(Vs[0]), torch.empty_like(Ws[0])
    dV2, dW2 = torch.empty_like(Vs[0]), torch.empty_like(Ws[0])

    def time_us(fn, iters=100):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for i in range(10):
            fn(i)
        torch.cuda.synchronize()
        e0.record()
        for i in range(iters):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3

    def step(i, far, cold, outs=(None, None)):
        s = i % sets if cold else 0
        chord._launch_bwd(dZs[s], Ws[s], Vs[s], outs[0] if outs[0] is not None else dW, outs[1] if outs[1] is not None else dV,
                          B, N, L, C, N * C, None, fars[s] if far else None, k0 if far else 0)

    # correctness: fused vs two kernels on the same operands
    sfa.set_tuning("bwd_fused", 0)
    step(0, False, False, (dW2, dV2))
    sfa.set_tuning("bwd_fused", 2)
    for far in (False, True):
        dW.zero_(), dV.zero_()
        step(0, far, False)
        torch.cuda.synchronize()
        err = float((dW - dW2).abs().max() / dW2.abs().max())
        print(f"fused (far copy {far}): dV bit-equal {bool(torch.equal(dV, dV2))}, dW rel err {err:.2e}")
    alg = 4 * B * N * (2 * L + 3 * C)
    print(f"B={B} N={N} L={L} C={C}  far links from k0={k0}; algorithmic bytes of a fused step {alg / 1e6:.1f} MB")
    for cold in (False, True):
        row = []
        for fused, nt in ((0, 0), (2, 0), (2, 1)):
            sfa.set_tuning("bwd_fused", fused)
            sfa.set_tuning("bwd_fused_nt", nt)
            for far in (False, True):
                if nt and far:
                    continue  # the copy starts at the 512-thread tile's first far link
                t = time_us(lambda i: step(i, far, cold))
                name = ("fused256" if nt else "fused") if fused else "dW+dV"
                row.append(f"{name}{'+far' if far else ''} {t:.1f} us ({alg / t / 1e6:.2f} TB/s)")
        print(("cold: " if cold else "warm: ") + " | ".join(row))
    sfa.set_tuning("bwd_fused", 1)
    sfa.set_tuning("bwd_fused_nt", 0)


if __name__ == "__main__":
    main()
