#!/usr/bin/env python3
"""Backward step (dV window + dW chunk kernels, C >= 64; fused step for C <= 32) of two builds of the library in one process,
arms interleaved, operands rotating through sets that span 2.5 x the Infinity Cache:
    python profiles/bwd_wide_nt_ab.py other/libpsf_chord.so
us per step, median of seven readings; dV and dW of the two builds compared bit for bit."""
import ctypes, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsefactorization_amd import _lib, chord  # noqa: E402

dev = torch.device("cuda:0")
new = _lib.load()
old = ctypes.CDLL(os.path.abspath(sys.argv[1]))
for fname, (argtypes, restype) in _lib.SIGNATURES.items():
    fn = getattr(old, fname, None)
    if fn is not None:
        fn.argtypes, fn.restype = argtypes, restype
FOOT = 640e6
for tag, B, N, L, C in (("cfg3 reference", 32, 2000, 12, 128), ("cfg3 wording", 32, 2048, 12, 64), ("ListOps-like 2049", 32, 2049, 12, 64),
                        ("N=4096 C=128", 16, 4096, 13, 128), ("genome-like C=64", 8, 16384, 15, 64), ("cfg4 (fused)", 64, 1024, 12, 32),
                        ("IMDb 4097 (fused edge)", 32, 4097, 13, 32)):
    g = torch.Generator(device=dev).manual_seed(0)
    per_set = 4 * B * N * (2 * L + 4 * C)
    sets = max(2, min(48, int(FOOT / per_set) + 1))
    Ws = [0.1 * torch.randn(B, N, L, device=dev, generator=g) for _ in range(sets)]
    Vs = [torch.randn(B, N, C, device=dev, generator=g) for _ in range(sets)]
    dZs = [torch.randn(B, N, C, device=dev, generator=g) for _ in range(sets)]
    dWs = [torch.empty_like(Ws[0]) for _ in range(sets)]
    dVs = [torch.empty_like(Vs[0]) for _ in range(sets)]

    def reading(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for i in range(n):
            s = i % sets
            chord._launch_bwd(dZs[s], Ws[s], Vs[s], dWs[s], dVs[s], B, N, L, C, N * C, None)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n
    t, outs = {"old": [], "new": []}, {}
    for rd in range(8):
        for key, lib in ((("old", old), ("new", new)) if rd % 2 == 0 else (("new", new), ("old", old))):
            _lib._lib = lib
            r = reading(3 * sets)
            if rd:
                t[key].append(r)
            outs[key] = (dWs[0].clone(), dVs[0].clone())
    _lib._lib = new
    same = all(torch.equal(a, b) for a, b in zip(outs["old"], outs["new"]))
    o, w = statistics.median(t["old"]), statistics.median(t["new"])
    print(f"{tag:24s} B={B} N={N} L={L} C={C} ({sets} sets): old {o:7.2f} us  new {w:7.2f} us  ({(o / w - 1) * 100:+5.1f} %)  equal bits: {same}", flush=True)
    del Ws, Vs, dZs, dWs, dVs
    torch.cuda.empty_cache()
