#!/usr/bin/env python3
"""Fused backward step at rows of 32 channels, two builds of the library in one process (arms interleaved, five rounds):
    python profiles/bwd_c32_ab.py other/libpsf_chord.so
us per step, cache-resident (one operand set) / rotating operands; dV and dW of the two builds compared bit for bit."""
import ctypes, os, sys
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


def timed(fn, n):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for B, N, L, C in [(16, 16384, 15, 32), (32, 16384, 15, 32), (32, 4096, 13, 32), (16, 16384, 15, 16), (40, 16384, 15, 8)]:
    g = torch.Generator(device=dev).manual_seed(3)
    per = 4 * B * N * (2 * L + 3 * C)
    sets = max(2, min(24, -(-640_000_000 // per)))
    Ws = [0.1 * torch.randn(B, N, L, device=dev, generator=g) for _ in range(sets)]
    Vs = [torch.randn(B, N, C, device=dev, generator=g) for _ in range(sets)]
    dZ = torch.randn(B, N, C, device=dev, generator=g)
    dWs = [torch.empty_like(Ws[0]) for _ in range(sets)]
    dVs = [torch.empty_like(Vs[0]) for _ in range(sets)]
    res, outs = {}, {}
    for rd in range(5):
        for name, lib in (("this", new), ("other", old)):
            _lib._lib = lib
            it = [0]

            def rot():
                s = it[0] % sets
                it[0] += 1
                chord._launch_bwd(dZ, Ws[s], Vs[s], dWs[s], dVs[s], B, N, L, C, N * C, None)

            warm = timed(lambda: chord._launch_bwd(dZ, Ws[0], Vs[0], dWs[0], dVs[0], B, N, L, C, N * C, None), 40)
            r = timed(rot, max(40, 2 * sets))
            res.setdefault(name, []).append((warm, r))
            outs[name] = (dWs[0].clone(), dVs[0].clone())
    _lib._lib = new
    same = all(torch.equal(a, b) for a, b in zip(outs["this"], outs["other"]))
    print(f"B={B} N={N} L={L} C={C} ({sets} sets): " + "  ".join(
        f"{k}: {min(x[0] for x in v):.2f} / {min(x[1] for x in v):.2f}" for k, v in res.items()) + f"   bit-equal={same}", flush=True)
    del Ws, Vs, dWs, dVs
    torch.cuda.empty_cache()
