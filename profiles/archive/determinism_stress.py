"""Run-to-run bit stability of the chord kernels under load: every kernel is run `reps` times on the same operands and each
result compared bit for bit with the first. Any difference is a hardware / code-generation hazard, not arithmetic."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import sparsefactorization_amd as sfa  # noqa: E402
from sparsefactorization_amd import fused_mixer, fused_mlp  # noqa: E402
from sparsefactorization_amd.chord import _launch_bwd  # noqa: E402
from sparsefactorization_amd.psfnet import MLPBlock  # noqa: E402

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
mixer_only = len(sys.argv) > 2 and sys.argv[2] == "mixer"
SHAPES = [("cfg2", 64, 16384, 15, 8, True), ("order", 40, 16384, 15, 8, True), ("genome", 16, 16384, 15, 32, False),
          ("imdb", 32, 4097, 13, 32, True), ("c16", 32, 16384, 15, 16, True), ("listops", 32, 2000, 12, 128, False)]


def where(a, b):
    d = (a != b)
    idx = d.reshape(-1, d.shape[-1]).any(-1).nonzero().flatten()
    return f"{idx.numel()} rows, first {idx[:4].tolist()}, columns {d.reshape(-1, d.shape[-1])[idx[0]].nonzero().flatten().tolist()[:8]}"


for name, B, N, L, C, res in ([] if mixer_only else SHAPES):
    torch.manual_seed(1)
    W = 0.1 * torch.randn(B, N, L, device=dev)
    V = torch.randn(B, N, C, device=dev)
    dZ = torch.randn(B, N, C, device=dev)
    with torch.no_grad():
        first = sfa.chord_spmm(W, V, V if res else None)
        bad = 0
        for r in range(reps):
            out = sfa.chord_spmm(W, V, V if res else None)
            if not torch.equal(out, first):
                bad += 1
                if bad <= 3:
                    print(f"  {name} forward step rep {r}: {where(out, first)}", flush=True)
        print(f"{name}: forward step {bad} of {reps} runs differ", flush=True)
        dW0, dV0 = torch.empty_like(W), torch.empty_like(V)
        _launch_bwd(dZ, W, V, dW0, dV0, B, N, L, C, N * C, None)
        bad = 0
        for r in range(reps):
            dW, dV = torch.empty_like(W), torch.empty_like(V)
            _launch_bwd(dZ, W, V, dW, dV, B, N, L, C, N * C, None)
            if not (torch.equal(dW, dW0) and torch.equal(dV, dV0)):
                bad += 1
                if bad <= 3:
                    print(f"  {name} backward step rep {r}: dW {'same' if torch.equal(dW, dW0) else where(dW, dW0)}; "
                          f"dV {'same' if torch.equal(dV, dV0) else where(dV, dV0)}", flush=True)
        print(f"{name}: backward step {bad} of {reps} runs differ", flush=True)

for name, B, N, E, h, C, L, res in [("genome", 16, 16384, 32, 32, 32, 15, False), ("cfg2", 64, 16384, 32, 32, 8, 15, True),
                                    ("c16", 32, 16384, 32, 32, 16, 15, True), ("pathfinder", 64, 1024, 32, 128, 32, 12, False),
                                    ("imdb", 32, 4097, 32, 128, 32, 13, True)]:
    torch.manual_seed(0)
    g = MLPBlock([h, 'GELU'], E, C).to(dev)
    f = MLPBlock([h, 'GELU'], E, L).to(dev)
    x = torch.randn(B, N, E, device=dev)
    with torch.no_grad():
        first = fused_mixer.mixer_forward(x, g, [f], res).clone()
        p0 = [t.clone() for t in fused_mlp.fused_mlp_forward(x, [g, f])]
        bad = badp = 0
        for r in range(reps):
            out = fused_mixer.mixer_forward(x, g, [f], res)
            if not torch.equal(out, first):
                bad += 1
                if bad <= 3:
                    print(f"  {name} mixer step rep {r}: {where(out, first)}", flush=True)
            p = fused_mlp.fused_mlp_forward(x, [g, f])
            if not all(torch.equal(a, b) for a, b in zip(p, p0)):
                badp += 1
        print(f"{name}: mixer step {bad} of {reps} runs differ; producer MLPs {badp} of {reps}", flush=True)
