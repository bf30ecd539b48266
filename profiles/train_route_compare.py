#!/usr/bin/env python3
"""Is the training trajectory at the headline length a property of the kernels or of the optimisation? The Adding network at
N = 16384 (reference configuration: batch 40, Adam lr 1e-3), the same seed and batches, trained twice: with the fused producer
kernels (split-bf16 MLPs, A&S GELU) and with the MLPs in plain PyTorch f32 (library GEMMs + torch GELU); the chord chain is this
library's in both. Loss per step side by side.   python profiles/train_route_compare.py [steps]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsefactorization_amd import fused_mlp, psf_training  # noqa: E402
from sparsefactorization_amd.train import make_adam, seed_everything  # noqa: E402

dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
problem, N, batch = "adding", 16384, 40
X, Y = psf_training.make_split(problem, batch * 50, N, dev, 1000)
traj = {}
for route in ("fused", "pytorch"):
    fused_mlp.enabled = fused_mlp.train_enabled = fused_mlp.wide_enabled = (route == "fused")
    seed_everything(42)
    net = psf_training.build_model(problem, N).to(dev)
    opt = make_adam(net.parameters(), 1e-3)
    loss = torch.nn.MSELoss()
    out = []
    for i in range(steps):
        j = i % 50
        x, y = X[j * batch:(j + 1) * batch], Y[j * batch:(j + 1) * batch]
        opt.zero_grad(set_to_none=True)
        l = loss(net(x).squeeze(), y)
        l.backward()
        opt.step()
        out.append(float(l.detach()))
    traj[route] = out
fused_mlp.enabled = fused_mlp.train_enabled = fused_mlp.wide_enabled = True
a, b = traj["fused"], traj["pytorch"]
first = next((i for i in range(steps) if abs(a[i] - b[i]) > 1e-3 * max(abs(b[i]), 1e-3)), None)
print(f"first step at which the two losses differ by more than 1e-3 relative: {first}")
for i in list(range(0, 20)) + list(range(20, steps, max(1, steps // 40))):
    print(f"step {i:4d}: fused {a[i]:.6f}   pytorch MLPs {b[i]:.6f}   rel diff {abs(a[i] - b[i]) / max(abs(b[i]), 1e-12):.2e}")
print(f"max loss over the run: fused {max(a):.3f} (step {a.index(max(a))}), pytorch MLPs {max(b):.3f} (step {b.index(max(b))})")
