#!/usr/bin/env python3
"""The fused producer-MLP forward alone at the headline token count (15 MLPs, E = h = 32, T = 64 * 16384):
ms per call for the variant selected by PSF_MLP_VARIANT (0 auto, 1/2 f32 MFMA, 3 split-bf16). Used under rocprofv3."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsefactorization_amd as sfa  # noqa: E402
from sparsefactorization_amd import fused_mlp  # noqa: E402
from sparsefactorization_amd.psfnet import MLPBlock  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
blocks = [MLPBlock([32, 'GELU'], 32, 8).to(dev)] + [MLPBlock([32, 'GELU'], 32, 15).to(dev) for _ in range(14)]
x = torch.randn(64 * 16384, 32, device=dev)
sfa.set_tuning("mlp_variant", int(os.environ.get("PSF_MLP_VARIANT", "0")))
iters = int(os.environ.get("PSF_ITERS", "10"))
with torch.no_grad():
    fused_mlp.fused_mlp_forward(x, blocks)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fused_mlp.fused_mlp_forward(x, blocks)
    e1.record()
    torch.cuda.synchronize()
print("variant", os.environ.get("PSF_MLP_VARIANT", "0"), "ms/call", e0.elapsed_time(e1) / iters)
