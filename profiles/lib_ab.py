#!/usr/bin/env python3
"""A/B of two builds of the library in one process, arms interleaved: the forward chain at the headline shape (and Order's),
the fused backward step warm and with rotating operands, the mixer with W computed in the step. us per call, median of seven
readings; results of the two builds are compared bit for bit.
    python profiles/lib_ab.py sparsefactorization_amd/libpsf_chord_prev.so"""
import ctypes
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsefactorization_amd as sfa  # noqa: E402
from sparsefactorization_amd import _lib, chord  # noqa: E402

dev = torch.device("cuda:0")
new = _lib.load()
old = ctypes.CDLL(os.path.abspath(sys.argv[1]))
for fname, (argtypes, restype) in _lib.SIGNATURES.items():
    fn = getattr(old, fname, None)
    if fn is not None:
        fn.argtypes, fn.restype = argtypes, restype


def with_lib(lib, fn):
    _lib._lib = lib
    try:
        return fn()
    finally:
        _lib._lib = new


def timed(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def ab(name, fn, n, check=None):
    for lib in (old, new):
        with_lib(lib, lambda: timed(fn, max(3, n // 4)))
    t = {"old": [], "new": []}
    for rnd in range(7):
        for key, lib in ((("old", old), ("new", new)) if rnd % 2 == 0 else (("new", new), ("old", old))):
            t[key].append(with_lib(lib, lambda: timed(fn, n)))
    o, w = statistics.median(t["old"]), statistics.median(t["new"])
    extra = ""
    if check is not None:
        a = with_lib(old, check)
        b = with_lib(new, check)
        extra = "  equal bits: " + str(all(bool(torch.equal(x, y)) for x, y in zip(a, b)))
    print(f"{name:44s} old {o:8.2f} us   new {w:8.2f} us   ({(o / w - 1) * 100:+5.1f} %){extra}", flush=True)


from sparsefactorization_amd import fused_mlp  # noqa: E402
from sparsefactorization_amd.psfnet import MLPBlock  # noqa: E402

torch.manual_seed(0)
for tag, T, O in (("cfg2: 15 MLPs, 1 M tokens", 64 * 16384, 15), ("Order B=40", 40 * 16384, 15), ("L=12 links", 64 * 2048, 12)):
    blocks = [MLPBlock([32, 'GELU'], 32, 8).to(dev)] + [MLPBlock([32, 'GELU'], 32, O).to(dev) for _ in range(O - 1)]
    x = torch.randn(T, 32, device=dev)
    with torch.no_grad():
        ab(f"producer MLPs forward, {tag}", lambda i: fused_mlp.fused_mlp_forward(x, blocks), 10,
           check=lambda: list(fused_mlp.fused_mlp_forward(x, blocks)))
    del x, blocks
    torch.cuda.empty_cache()

g = torch.Generator(device=dev).manual_seed(0)
for tag, B, N, L, C in (("cfg2 B=64", 64, 16384, 15, 8), ("Order B=40", 40, 16384, 15, 8), ("genome B=16 C=32", 16, 16384, 15, 32),
                        ("pathfinder B=64 C=32", 64, 1024, 11, 32), ("C=16 N=4096 B=64", 64, 4096, 13, 16)):
    M = L - 1
    Ws = [0.1 * torch.randn(B, N, L, device=dev, generator=g) for _ in range(M)]
    V0 = torch.randn(B, N, C, device=dev, generator=g)
    with torch.no_grad():
        ab(f"forward chain, {tag} (per chain)", lambda i: sfa.chord_chain(Ws, V0, True), 20,
           check=lambda: [sfa.chord_chain(Ws, V0, True)])
    sets = min(10, M)
    Vs = [torch.randn(B, N, C, device=dev, generator=g) for _ in range(sets)]
    dZ = torch.randn(B, N, C, device=dev, generator=g)
    dWs = [torch.empty_like(Ws[0]) for _ in range(sets)]
    dVs = [torch.empty_like(V0) for _ in range(sets)]

    def bwd(i, cold):
        s = i % sets if cold else 0
        chord._launch_bwd(dZ, Ws[s], Vs[s], dWs[s], dVs[s], B, N, L, C, N * C, None)

    def bwd_check():
        bwd(0, False)
        torch.cuda.synchronize()
        return [dWs[0].clone(), dVs[0].clone()]

    ab(f"backward step, {tag}, warm", lambda i: bwd(i, False), 100, check=bwd_check)
    ab(f"backward step, {tag}, operands rotate", lambda i: bwd(i, True), 100)
    del Ws, Vs, dWs, dVs
    torch.cuda.empty_cache()
