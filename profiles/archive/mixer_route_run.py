#!/usr/bin/env python3
"""Workload for rocprofv3: 20 no-grad forwards of the Adding PSFNet (seed 42, N = 16384, B = 64) through ONE route of the
mixer:  python3 profiles/mixer_route_run.py never|always|recipe   (never = W through memory, always = W computed inside the
step, recipe = the same with the affine input layer evaluated inside the kernels too: `data` never written)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsefactorization_amd import fused_mixer, psf_training  # noqa: E402
from sparsefactorization_amd.train import seed_everything  # noqa: E402

fused_mixer.route = sys.argv[1] if len(sys.argv) > 1 else "never"
problem = sys.argv[2] if len(sys.argv) > 2 else "adding"
dev = torch.device("cuda:0")
seed_everything(42)
net = psf_training.build_model(problem, 16384).to(dev).eval()
x, _ = psf_training.make_split(problem, 64, 16384, dev, 42)
with torch.no_grad():
    for _ in range(20):
        y = net(x)
torch.cuda.synchronize()
print("route", fused_mixer.route, "logits", float(y.float().abs().sum()))
