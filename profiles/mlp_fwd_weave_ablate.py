#!/usr/bin/env python3
"""x3w_fwd_k (csrc/mlp_fwd_x3w.hip) with parts compiled out — library built with PSF_HIPCC_EXTRA=-DPSF_X3W_ABLATE_LAB:
    python profiles/mlp_fwd_weave_ablate.py
ms per call at T = 1 M tokens, 15 MLPs of E = h = 32; knob mlp_variant = 4 + ABL (1 = no Y stores, 2 = no GELU / split
arithmetic, 4 = no MFMAs, 8 = no per-slot barrier and image fetch); 0 = x3_fwd_k."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsefactorization_amd import _lib, fused_mlp  # noqa: E402
from sparsefactorization_amd.psfnet import MLPBlock  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
blocks = [MLPBlock([32, 'GELU'], 32, 8)] + [MLPBlock([32, 'GELU'], 32, 15) for _ in range(14)]
blocks = [b.to(dev) for b in blocks]
x = torch.randn(64, 16384, 32, device=dev)
names = {0: "x3_fwd_k"}
for abl in range(16):
    parts = [n for bit, n in ((1, "no stores"), (2, "no GELU/split"), (4, "no MFMAs"), (8, "no barrier/fetch")) if abl & bit]
    names[4 + abl] = "x3w_fwd_k " + (", ".join(parts) if parts else "(everything)")
variants = [int(v) for v in sys.argv[1:]] or sorted(names)
res = {}
with torch.no_grad():
    for rd in range(3):
        for v in variants:
            _lib.set_tuning("mlp_variant", v)
            for _ in range(3):
                fused_mlp.fused_mlp_forward(x, blocks)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(20):
                fused_mlp.fused_mlp_forward(x, blocks)
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(v, []).append(e0.elapsed_time(e1) / 20)
_lib.set_tuning("mlp_variant", 0)
for v in variants:
    print(f"mlp_variant={v:2d}  {names[v]:60s} {min(res[v]):.4f} ms", flush=True)
