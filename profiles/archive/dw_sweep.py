#!/usr/bin/env python3
"""dW kernels A/B on the BASELINE shapes: whole-row window kernel (dw_variant=1) vs the chunk-looping kernel
(dw_variant=2; dw_tgs 4 / 5 = 8 / 16 lanes per row chunk; fwd_split 2 / 0 = ragged tiles in a second launch / one
predicated launch), interleaved in one process.
Reports median us per launch, algorithmic TB/s (4*B*N*(L+2C) bytes) and max|err|/max|ref| against a float64 reference.

    python profiles/dw_sweep.py [--rounds 5] [--iters 30] [--shapes cfg2,cfg3_ref,...]
"""
import argparse
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sparsefactorization_amd as sfa  # noqa: E402
from sparsefactorization_amd import chord  # noqa: E402

SHAPES = {  # name: (B, N, L, C)
    "cfg2": (64, 16384, 15, 8),
    "order_b40": (40, 16384, 15, 8),
    "cfg3_ref": (32, 2000, 12, 128),
    "cfg3_base": (32, 2048, 12, 64),
    "cfg4_train": (64, 1024, 12, 32),
    "genome": (16, 16384, 15, 32),
    "imdb": (32, 4097, 13, 32),
    "c16": (32, 4096, 13, 16),
}
VARIANTS = [("whole_row", {"dw_variant": 1}), ("auto", {}),
            ("chunk_tg8", {"dw_variant": 2, "dw_tgs": 4}), ("chunk_tg16", {"dw_variant": 2, "dw_tgs": 5}),
            ("chunk_tg8_2launch", {"dw_variant": 2, "dw_tgs": 4, "fwd_split": 2}),
            ("chunk_tg8_1launch", {"dw_variant": 2, "dw_tgs": 4, "fwd_split": 0})]
DEFAULTS = {"dw_variant": 0, "dw_tgs": 0, "fwd_split": 1}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--shapes", default=",".join(SHAPES))
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "dw_sweep.json"))
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    res = {}
    for name in args.shapes.split(","):
        B, N, L, C = SHAPES[name]
        g = torch.Generator(device=dev).manual_seed(1)
        W = 0.1 * torch.randn(B, N, L, device=dev, generator=g)
        V = torch.randn(B, N, C, device=dev, generator=g)
        dZ = torch.randn(B, N, C, device=dev, generator=g)
        dW = torch.empty_like(W)
        offs = [0] + [(1 << (k - 1)) % N for k in range(1, L)]
        ref = torch.stack([(dZ.double() * torch.roll(V.double(), -o, 1)).sum(-1) for o in offs], -1)
        alg = 4 * B * N * (L + 2 * C)
        times = {v: [] for v, _ in VARIANTS}
        errs = {}
        for r in range(args.rounds + 1):
            for vname, knobs in VARIANTS:
                for k, val in {**DEFAULTS, **knobs}.items():
                    sfa.set_tuning(k, val)
                try:
                    dW.zero_()
                    chord._launch_bwd(dZ, W, V, dW, None, B, N, L, C, N * C, None)
                except RuntimeError as exc:
                    times[vname] = None
                    errs[vname] = str(exc)[:80]
                    continue
                if r == 0:
                    errs[vname] = float((dW.double() - ref).abs().max() / ref.abs().max())
                    continue
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(args.iters):
                    chord._launch_bwd(dZ, W, V, dW, None, B, N, L, C, N * C, None)
                e1.record()
                torch.cuda.synchronize()
                times[vname].append(e0.elapsed_time(e1) / args.iters * 1e3)
        for k, val in DEFAULTS.items():
            sfa.set_tuning(k, val)
        print(f"{name}: B={B} N={N} L={L} C={C}  alg {alg / 1e6:.1f} MB")
        res[name] = {}
        for vname, _ in VARIANTS:
            t = times[vname]
            if not t:
                print(f"    {vname:14s} n/a ({errs.get(vname)})")
                continue
            med = statistics.median(t)
            res[name][vname] = {"us": med, "tbs": alg / med / 1e6, "err": errs[vname]}
            print(f"    {vname:14s} {med:7.2f} us  {alg / med / 1e6:5.2f} TB/s  ({alg / med / 1e6 / 8:.3f} of 8)  err {errs[vname]:.2e}")
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as fh:
        json.dump(res, fh, indent=1)


if __name__ == "__main__":
    main()
