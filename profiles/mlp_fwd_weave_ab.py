#!/usr/bin/env python3
"""Producer MLPs (psf_mlp_fwd_f32: 15 MLPs, E = h = 32) — x3_fwd_k (knob mlp_variant = 0) against the woven kernel x3w_fwd_k
(mlp_variant = 4: units software-pipelined inside each wave, csrc/mlp_fwd_x3w.hip), arms interleaved in one process:
    python profiles/mlp_fwd_weave_ab.py
ms per call at T = 1 M and 655 k tokens (and two LRA-like sizes); every output compared bit for bit."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsefactorization_amd import _lib, fused_mlp  # noqa: E402
from sparsefactorization_amd.psfnet import MLPBlock  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
cases = [("adding 15 MLPs h=32", [MLPBlock([32, 'GELU'], 32, 8)] + [MLPBlock([32, 'GELU'], 32, 15) for _ in range(14)], [(64, 16384), (40, 16384), (3, 1000)]),
         ("pathfinder 12 MLPs h=128", [MLPBlock([128, 'GELU'], 32, 32)] + [MLPBlock([128, 'GELU'], 32, 12) for _ in range(11)], [(64, 1024), (8, 1025)]),
         ("cifar 11 MLPs h=16, E=8", [MLPBlock([16, 'GELU'], 8, 8)] + [MLPBlock([16, 'GELU'], 8, 11) for _ in range(10)], [(64, 1024)])]
for name, blocks, sizes in cases:
    blocks = [b.to(dev) for b in blocks]
    for B, N in sizes:
        x = torch.randn(B, N, blocks[0].network[0].in_features, device=dev)
        res, outs = {}, {}
        with torch.no_grad():
            for rd in range(5):
                for variant in (0, 4):
                    _lib.set_tuning("mlp_variant", variant)
                    for _ in range(3):
                        y = fused_mlp.fused_mlp_forward(x, blocks)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    torch.cuda.synchronize()
                    e0.record()
                    for _ in range(20):
                        y = fused_mlp.fused_mlp_forward(x, blocks)
                    e1.record()
                    torch.cuda.synchronize()
                    res.setdefault(variant, []).append(e0.elapsed_time(e1) / 20)
                    outs[variant] = [t.clone() for t in y]
        _lib.set_tuning("mlp_variant", 0)
        same = all(torch.equal(a, b) for a, b in zip(outs[0], outs[4]))
        diff = max(float((a - b).abs().max()) for a, b in zip(outs[0], outs[4]))
        finite = all(bool(torch.isfinite(t).all()) for t in outs[4])
        print(f"{name} T={B * N}: x3_fwd_k {min(res[0]):.4f} ms   x3w_fwd_k (woven) {min(res[4]):.4f} ms   bit-equal={same}  "
              f"max |diff| = {diff:.3g}  finite={finite}", flush=True)
