#!/usr/bin/env python3
"""Is the training trajectory at the headline length a property of the kernels or of the optimisation? The Adding network at
N = 16384 (reference configuration: batch 40, Adam lr 1e-3), the same seed and batches, trained three times: with this library
throughout; with the MLPs in plain PyTorch f32 (library GEMMs + torch GELU) and this library's chord chain; and with the chain in
plain PyTorch too (out = sum_k W[..., k] * roll(V, -off_k) (+ V0) under torch autograd — the arithmetic of
torch_sparse.spmm(chord index) written with dense ops). Loss per step side by side.
    python profiles/train_route_compare.py [steps]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsefactorization_amd import fused_mixer, fused_mlp, psf_training, psfnet  # noqa: E402
from sparsefactorization_amd._lib import chord_offsets  # noqa: E402
from sparsefactorization_amd.train import make_adam, seed_everything  # noqa: E402

dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
problem, N, batch = "adding", 16384, 40
X, Y = psf_training.make_split(problem, batch * 50, N, dev, 1000)
library_mix = psfnet._ChordMixer.mix


def torch_mix(self, data, V, use_residuals, links=None):
    Ws = self.link_weights(data) if links is None else links
    offs = [int(o) for o in chord_offsets(V.shape[1], Ws[0].shape[-1])]
    V0 = V
    for W in Ws:
        acc = W[..., 0:1] * torch.roll(V, -offs[0], 1)
        for k in range(1, len(offs)):
            acc = acc + W[..., k:k + 1] * torch.roll(V, -offs[k], 1)
        V = acc + V0 if use_residuals else acc
    return V


traj, ms = {}, {}
for route in ("fused", "pytorch", "pure"):
    fused_mlp.enabled = fused_mlp.train_enabled = fused_mlp.wide_enabled = fused_mixer.enabled = (route == "fused")
    psfnet._ChordMixer.mix = torch_mix if route == "pure" else library_mix
    seed_everything(42)
    net = psf_training.build_model(problem, N).to(dev)
    opt = make_adam(net.parameters(), 1e-3)
    loss = torch.nn.MSELoss()
    out = []
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    for i in range(steps):
        j = i % 50
        x, y = X[j * batch:(j + 1) * batch], Y[j * batch:(j + 1) * batch]
        opt.zero_grad(set_to_none=True)
        l = loss(net(x).squeeze(), y)
        l.backward()
        opt.step()
        out.append(l.detach())
    torch.cuda.synchronize()
    ms[route] = (time.perf_counter() - t_start) / steps * 1e3
    traj[route] = [float(v) for v in out]
fused_mlp.enabled = fused_mlp.train_enabled = fused_mlp.wide_enabled = fused_mixer.enabled = True
psfnet._ChordMixer.mix = library_mix
a, b, c = traj["fused"], traj["pytorch"], traj["pure"]
first = next((i for i in range(steps) if abs(a[i] - b[i]) > 1e-3 * max(abs(b[i]), 1e-3)), None)
print(f"first step at which the two losses differ by more than 1e-3 relative: {first}")
for i in list(range(0, 20)) + list(range(20, steps, max(1, steps // 40))):
    print(f"step {i:4d}: library {a[i]:.6f}   pytorch MLPs {b[i]:.6f} (rel {abs(a[i] - b[i]) / max(abs(b[i]), 1e-12):.1e})   "
          f"pytorch MLPs and chain {c[i]:.6f} (rel {abs(a[i] - c[i]) / max(abs(c[i]), 1e-12):.1e})")
print(f"ms per training step (wall, {steps} steps): library {ms['fused']:.2f}, pytorch MLPs + library chain {ms['pytorch']:.2f}, "
      f"pure pytorch (dense-op chain, library GEMMs) {ms['pure']:.2f}")
print(f"max loss over the run: library {max(a):.3f} (step {a.index(max(a))}), pytorch MLPs {max(b):.3f} (step {b.index(max(b))}), "
      f"pure pytorch {max(c):.3f} (step {c.index(max(c))})")
