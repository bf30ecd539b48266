#!/usr/bin/env python3
"""Step kernels on COLD operands: the way they run inside a training step, where W_m and the saved V_m were written long
ago and come from HBM, not from the 256 MB Infinity Cache that a single-buffer timing loop keeps them in.

    python profiles/cold_sweep.py [--shapes order_b40,cfg2] [--rounds 5] [--iters 24]

Each shape gets S complete operand sets (S x set bytes >= 1.5 GB), walked round-robin, so every launch reads operands
that ~1.4 GB of other traffic have passed over since their last use. Variants (tuning knobs) are interleaved in one
process; "warm" repeats the default on ONE set for comparison. Reports median us per launch and TB/s on algorithmic bytes.
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
    "order_b40": (40, 16384, 15, 8),
    "cfg2": (64, 16384, 15, 8),
    "genome": (16, 16384, 15, 32),
}
KERNELS = {
    "fwd": {"bytes": lambda B, N, L, C: 4 * B * N * (L + 3 * C),  # with residual
            "variants": [("auto", {}), ("rows1", {"fwd_variant": 2, "fwd_rows": 1}), ("rows2", {"fwd_variant": 2, "fwd_rows": 2}),
                         ("wg_limit1", {"fwd_wg_limit": 1}), ("wg_limit2", {"fwd_wg_limit": 2}), ("wg_limit3", {"fwd_wg_limit": 3}),
                         ("wg_limit4", {"fwd_wg_limit": 4})]},
    "dV": {"bytes": lambda B, N, L, C: 4 * B * N * (L + 2 * C),
           "variants": [("auto", {}), ("nt256_r2", {"dv_threads": 1, "bwd_rows": 2}), ("nt256_r1", {"dv_threads": 1, "bwd_rows": 1}),
                        ("nt512_r1", {"dv_threads": 2})]},
    "dW": {"bytes": lambda B, N, L, C: 4 * B * N * (L + 2 * C),
           "variants": [("auto", {}), ("rows1", {"bwd_rows": 1}), ("rows2", {"bwd_rows": 2}), ("whole_row", {"dw_variant": 1}),
                        ("chunk", {"dw_variant": 2})]},
}
DEFAULTS = {"fwd_variant": 0, "fwd_rows": 0, "fwd_wg_limit": 0, "dv_threads": 0, "bwd_rows": 0, "dw_variant": 0}


def set_knobs(kn):
    for k, v in {**DEFAULTS, **kn}.items():
        sfa.set_tuning(k, v)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="order_b40,cfg2")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=24)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "cold_sweep.json"))
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    res = {}
    for name in args.shapes.split(","):
        B, N, L, C = SHAPES[name]
        set_bytes = 4 * B * N * (2 * L + 4 * C)
        S = max(2, -(-1_500_000_000 // set_bytes))
        g = torch.Generator(device=dev).manual_seed(1)
        sets = []
        for _ in range(S):
            sets.append({"W": 0.1 * torch.randn(B, N, L, device=dev, generator=g), "V": torch.randn(B, N, C, device=dev, generator=g),
                         "dZ": torch.randn(B, N, C, device=dev, generator=g), "R": torch.randn(B, N, C, device=dev, generator=g),
                         "out": torch.empty(B, N, C, device=dev), "dW": torch.empty(B, N, L, device=dev)})
        print(f"{name}: B={B} N={N} L={L} C={C}; {S} operand sets of {set_bytes / 1e6:.0f} MB", flush=True)
        res[name] = {}

        def launch(kernel, s):
            if kernel == "fwd":
                chord._launch_fwd(s["W"], s["V"], s["R"], s["out"], B, N, L, C, N * C, None)
            elif kernel == "dV":
                chord._launch_bwd(s["dZ"], s["W"], s["V"], None, s["out"], B, N, L, C, N * C, None)
            else:
                chord._launch_bwd(s["dZ"], s["W"], s["V"], s["dW"], None, B, N, L, C, N * C, None)

        for kernel, spec in KERNELS.items():
            alg = spec["bytes"](B, N, L, C)
            variants = spec["variants"] + [("warm (one set, auto)", {})]
            times = {v: [] for v, _ in variants}
            for r in range(args.rounds + 1):
                for vname, knobs in variants:
                    set_knobs(knobs)
                    warm = vname.startswith("warm")
                    try:
                        launch(kernel, sets[0])
                    except RuntimeError as exc:
                        times[vname] = None
                        continue
                    if times[vname] is None:
                        continue
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    torch.cuda.synchronize()
                    e0.record()
                    for i in range(args.iters):
                        launch(kernel, sets[0] if warm else sets[i % S])
                    e1.record()
                    torch.cuda.synchronize()
                    if r:
                        times[vname].append(e0.elapsed_time(e1) / args.iters * 1e3)
            set_knobs({})
            res[name][kernel] = {}
            for vname, _ in variants:
                t = times[vname]
                if not t:
                    print(f"  {kernel:3s} {vname:22s} n/a")
                    continue
                med = statistics.median(t)
                res[name][kernel][vname] = {"us": med, "tbs": alg / med / 1e6}
                print(f"  {kernel:3s} {vname:22s} {med:7.2f} us  {alg / med / 1e6:5.2f} TB/s ({alg / med / 1e6 / 8:.3f} of 8)", flush=True)
        del sets
        torch.cuda.empty_cache()
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as fh:
        json.dump(res, fh, indent=1)


if __name__ == "__main__":
    main()
