#!/usr/bin/env python3
"""Forward step at rows of 64..256 channels: fwd_wide = 0 (automatic), 1 (1024-thread chunks), 4 (whole rows),
us per step, one process, settings interleaved, five rounds; cache-resident (one operand set) and rotating operands."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsefactorization_amd as sfa  # noqa: E402
from sparsefactorization_amd import chord  # noqa: E402

dev = torch.device("cuda:0")
SHAPES = [(32, 2048, 12, 64), (32, 2000, 12, 128), (32, 2048, 12, 128), (32, 4096, 13, 64), (16, 2048, 12, 256), (8, 16384, 15, 64)]


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


for B, N, L, C in SHAPES:
    g = torch.Generator(device=dev).manual_seed(1)
    per = 4 * B * N * (L + 2 * C)
    sets = max(2, min(48, -(-640_000_000 // per)))
    Ws = [0.1 * torch.randn(B, N, L, device=dev, generator=g) for _ in range(sets)]
    Vs = [torch.randn(B, N, C, device=dev, generator=g) for _ in range(sets)]
    outs = [torch.empty_like(Vs[0]) for _ in range(sets)]
    res = {}
    ref = None
    for rd in range(5):
        for wide in (0, 1, 4):
            sfa.set_tuning("fwd_wide", wide)
            it = [0]

            def rot():
                s = it[0] % sets
                it[0] += 1
                chord._launch_fwd(Ws[s], Vs[s], None, outs[s], B, N, L, C, N * C, None)

            warm = timed(lambda: chord._launch_fwd(Ws[0], Vs[0], None, outs[0], B, N, L, C, N * C, None), 50)
            r = timed(rot, max(50, 2 * sets))
            res.setdefault(wide, []).append((warm, r))
            if ref is None:
                ref = outs[0].clone()
            else:
                assert torch.equal(ref, outs[0]), "results differ between configurations"
    sfa.set_tuning("fwd_wide", 0)
    line = f"B={B} N={N} L={L} C={C} ({sets} sets):"
    for wide, v in res.items():
        line += f"  wide={wide}: {min(x[0] for x in v):.2f} / {min(x[1] for x in v):.2f}"
    print(line + "   (us per step: cache-resident / rotating, best of 5)", flush=True)
    del Ws, Vs, outs
    torch.cuda.empty_cache()
