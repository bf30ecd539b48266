#!/usr/bin/env python3
"""Producer MLPs (psf_mlp_fwd_f32: 15 MLPs, E = h = 32) — two builds of the library in one process, arms interleaved:
    python profiles/mlp_fwd_ab.py other/libpsf_chord.so
ms per call at T = 1 M and 655 k tokens; every output of the two builds compared bit for bit."""
import ctypes, os, sys
import torch
from torch import nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsefactorization_amd import _lib, fused_mlp  # noqa: E402
from sparsefactorization_amd.psfnet import MLPBlock  # noqa: E402

dev = torch.device("cuda:0")
new = _lib.load()
old = ctypes.CDLL(os.path.abspath(sys.argv[1]))
for fname, (argtypes, restype) in _lib.SIGNATURES.items():
    fn = getattr(old, fname, None)
    if fn is not None:
        fn.argtypes, fn.restype = argtypes, restype
torch.manual_seed(0)
blocks = [MLPBlock([32, 'GELU'], 32, 8)] + [MLPBlock([32, 'GELU'], 32, 15) for _ in range(14)]
blocks = [b.to(dev) for b in blocks]
for T, B in ((1048576, 64), (655360, 40)):
    x = torch.randn(B, 16384, 32, device=dev)
    res, outs = {}, {}
    with torch.no_grad():
        for rd in range(5):
            for name, lib in (("this", new), ("other", old)):
                _lib._lib = lib
                for _ in range(3):
                    y = fused_mlp.fused_mlp_forward(x, blocks)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(20):
                    y = fused_mlp.fused_mlp_forward(x, blocks)
                e1.record()
                torch.cuda.synchronize()
                res.setdefault(name, []).append(e0.elapsed_time(e1) / 20)
                outs[name] = [t.clone() for t in y]
    _lib._lib = new
    same = all(torch.equal(a, b) for a, b in zip(outs["this"], outs["other"]))
    diff = max(float((a - b).abs().max()) for a, b in zip(outs["this"], outs["other"]))
    print(f"T={T}: " + "  ".join(f"{k}: {min(v):.4f} ms" for k, v in res.items()) + f"   bit-equal={same}  max |diff| = {diff:.3g}", flush=True)
