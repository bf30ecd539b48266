#!/usr/bin/env python3
"""Run under rocprofv3 --kernel-trace --stats: the cfg2 chain 20 times with G sequences per group (argv[1])."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsefactorization_amd import chord  # noqa: E402

G = int(sys.argv[1])
B, N, M, L, C = 64, 16384, 14, 15, 8
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
Ws = [0.1 * torch.randn(B, N, L, device=dev, generator=g) for _ in range(M)]
V0 = torch.randn(B, N, C, device=dev, generator=g)
bufs = [torch.empty_like(V0), torch.empty_like(V0)]
for _ in range(20):
    for b0 in range(0, B, G):
        sl = slice(b0, b0 + G)
        x = V0[sl]
        for m in range(M):
            out = bufs[m & 1][sl]
            chord._launch_fwd(Ws[m][sl], x, V0[sl], out, G, N, L, C, N * C, None)
            x = out
torch.cuda.synchronize()
