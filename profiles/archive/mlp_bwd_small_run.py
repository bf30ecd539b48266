#!/usr/bin/env python3
"""Workload for rocprofv3 --kernel-trace --stats: the producer MLPs' forward + backward at an LRA width and token count
(default CIFAR-10: T = 32 x 1024 tokens, E = 16, eleven MLPs 16-16-{16, 11}).  python3 profiles/mlp_bwd_small_run.py [T E h C L M]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsefactorization_amd import fused_mlp  # noqa: E402
from sparsefactorization_amd.psfnet import MLPBlock  # noqa: E402

T, E, h, C, L, M = (int(v) for v in sys.argv[1:7]) if len(sys.argv) >= 7 else (32768, 16, 16, 16, 11, 10)
dev = torch.device("cuda:0")
torch.manual_seed(0)
blocks = [MLPBlock([h, 'GELU'], E, C).to(dev)] + [MLPBlock([h, 'GELU'], E, L).to(dev) for _ in range(M)]
x = torch.randn(T, E, device=dev)
params = fused_mlp._params_of(blocks)
gys = [torch.randn_like(y) for y in fused_mlp._forward_raw(x, params)]
for _ in range(60):
    with torch.no_grad():
        fused_mlp._forward_raw(x, params)
        fused_mlp._backward_raw(x, params, gys, True)
torch.cuda.synchronize()
