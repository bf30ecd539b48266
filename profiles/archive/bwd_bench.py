#!/usr/bin/env python3
"""One step of the chord-spmm at a BASELINE shape: us per launch of the forward (+residual), dV alone, dW alone and
dV + dW, with the algorithmic TB/s of each backward kernel (4*B*N*(L+2C) bytes).

    python profiles/bwd_bench.py [B N L C]          default cfg2: 64 16384 15 8
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsefactorization_amd import chord  # noqa: E402


def main():
    B, N, L, C = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (64, 16384, 15, 8)
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    W = 0.1 * torch.randn(B, N, L, device=dev, generator=g)
    V = torch.randn(B, N, C, device=dev, generator=g)
    dZ = torch.randn(B, N, C, device=dev, generator=g)

    def time_us(fn, iters=50):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3

    dV, out, dW = torch.empty_like(V), torch.empty_like(V), torch.empty_like(W)
    alg = 4 * B * N * (L + 2 * C)
    t_fw = time_us(lambda: chord._launch_fwd(W, V, dZ, out, B, N, L, C, N * C, None))
    t_dv = time_us(lambda: chord._launch_bwd(dZ, W, V, None, dV, B, N, L, C, N * C, None))
    t_dw = time_us(lambda: chord._launch_bwd(dZ, W, V, dW, None, B, N, L, C, N * C, None))
    t_both = time_us(lambda: chord._launch_bwd(dZ, W, V, dW, dV, B, N, L, C, N * C, None))
    print(f"B={B} N={N} L={L} C={C}: fwd(+res) {t_fw:.1f} us  dV {t_dv:.1f} us ({alg / t_dv / 1e6:.2f} TB/s)  "
          f"dW {t_dw:.1f} us ({alg / t_dw / 1e6:.2f} TB/s)  both {t_both:.1f} us")
if __name__ == "__main__":
    main()
