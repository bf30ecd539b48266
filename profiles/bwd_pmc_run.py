#!/usr/bin/env python3
"""Workload for the per-shape counter passes: `iters` launches of dV alone, then of dW alone, then of the step that wants both
(the fused kernel, csrc/bwd_fused.h, where it applies), then of the forward step, then (round 6) `iters` no-grad chains of
M = L - 1 steps where the library runs those as ONE launch (chord_chain_lds_k: what inference runs at the LRA lengths), at one shape.

    python3 profiles/bwd_pmc_run.py B N L C [iters]

Run under `rocprofv3 --kernel-trace --stats` or one `--pmc` group at a time (profiles/collect_bwd.sh).
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsefactorization_amd import chord  # noqa: E402


def main():
    B, N, L, C = (int(a) for a in sys.argv[1:5])
    iters = int(sys.argv[5]) if len(sys.argv) > 5 else 100
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    W = 0.1 * torch.randn(B, N, L, device=dev, generator=g)
    V = torch.randn(B, N, C, device=dev, generator=g)
    dZ = torch.randn(B, N, C, device=dev, generator=g)
    dV, dW = torch.empty_like(V), torch.empty_like(W)
    for _ in range(iters):
        chord._launch_bwd(dZ, W, V, None, dV, B, N, L, C, N * C, None)
    torch.cuda.synchronize()
    for _ in range(iters):
        chord._launch_bwd(dZ, W, V, dW, None, B, N, L, C, N * C, None)
    torch.cuda.synchronize()
    for _ in range(iters):
        chord._launch_bwd(dZ, W, V, dW, dV, B, N, L, C, N * C, None)
    torch.cuda.synchronize()
    out = torch.empty_like(V)
    for _ in range(iters):  # (round 5) the forward step of the shape too, no residual: counters for every forward instance
        chord._launch_fwd(W, V, None, out, B, N, L, C, N * C, None)
    torch.cuda.synchronize()
    from sparsefactorization_amd import _lib
    M = L - 1
    if "chord_chain_" in _lib.describe_chain_fwd(B, N, L, C, M):  # the one-launch chain: bench.py's `fwd_chain_kernel`
        Ws = [0.1 * torch.randn(B, N, L, device=dev, generator=g) for _ in range(M)]
        with torch.no_grad():
            for _ in range(iters):
                chord.chord_chain(Ws, V, False)
        torch.cuda.synchronize()
    print(f"done B={B} N={N} L={L} C={C} iters={iters} alg_bytes={4 * B * N * (L + 2 * C)}")


if __name__ == "__main__":
    main()
