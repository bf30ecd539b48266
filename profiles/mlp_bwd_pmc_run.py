#!/usr/bin/env python3
"""Workload for counter passes on the fused MLP backward: `iters` calls at the Temporal-Order training shape.
    python3 profiles/mlp_bwd_pmc_run.py <mlp_bwd_variant 0|1> [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsefactorization_amd as sfa  # noqa: E402
from sparsefactorization_amd import fused_mlp  # noqa: E402
from profiles.mlp_bwd_bench import make  # noqa: E402

variant = int(sys.argv[1])
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
E, h, outs, T = 32, 32, [8] + [15] * 14, 40 * 16384
params = make(E, h, outs, dev)
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(T, E, device=dev, generator=g)
gys = [torch.randn(T, O, device=dev, generator=g) for O in outs]
sfa.set_tuning("mlp_bwd_variant", variant)
for _ in range(iters):
    fused_mlp._backward_raw(x, params, gys, True)
torch.cuda.synchronize()
print("done", variant)
