#!/usr/bin/env python3
"""Interleaved A/B timing of kernel variants in ONE process (cdna_hip_programming.md §5.4 rule 24).

    python profiles/sweep.py [--out gpurun_out/sweep.json] [--rounds 5] [--shapes cfg2,...]

For each shape and each variant (tuning-knob setting) it times the forward chain with HIP events, `rounds`
times, variants interleaved, and reports median / min microseconds per kernel launch and achieved GB/s on
algorithmic bytes (4*B*N*(L + 2C + [res]C) per launch). Also times the backward step and a
device-copy ceiling (out = in, same bytes) on the same box as the second denominator.
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sparsefactorization_amd as sfa  # noqa: E402

SHAPES = {
    # name: (B, N, M, C, residual)
    "cfg2": (64, 16384, 14, 8, True),          # BASELINE.json configs[1] (headline)
    "cfg1": (40, 128, 7, 8, True),             # configs[0]
    "cfg3_ref": (32, 2000, 11, 128, False),    # ListOps, reference config
    "cfg3_base": (32, 2048, 11, 64, False),    # ListOps, BASELINE wording
    "cfg4_train": (64, 1024, 11, 32, False),   # Pathfinder32
    "cfg4_map": (8, 1024, 11, 1024, False),    # attention map, C = N
    "genome": (16, 16384, 14, 32, True),
    "imdb": (32, 4097, 12, 32, True),          # ragged last tile (4097 = 64*64 + 1)
    "n2000_c16": (32, 2000, 11, 16, True),     # ragged, narrow rows
}

PS = {"chain_fused": 0}  # per-step launches (the shipped default fuses short sequences into one launch)
VARIANTS = [
    ("auto", {}),
    ("fused_cc1", {"chain_cc": 1}),
    ("per_step", {**PS}),
    ("generic", {**PS, "fwd_variant": 1}),
    ("win_r1", {**PS, "fwd_variant": 2, "fwd_rows": 1}),
    ("win_r2", {**PS, "fwd_variant": 2, "fwd_rows": 2}),
    ("noremap", {**PS, "xcd_remap": 0}),
    ("alledge", {**PS, "fwd_split": 0}),
    ("wide1_nt1024", {**PS, "fwd_wide": 1}),
    ("wide2_chunk256", {**PS, "fwd_wide": 2}),
]
DEFAULTS = {"fwd_variant": 0, "fwd_rows": 0, "xcd_remap": 1, "fwd_split": 1, "fwd_wide": 0, "chain_fused": 1, "chain_cc": 0}


def set_knobs(kn):
    for k, v in {**DEFAULTS, **kn}.items():
        sfa.set_tuning(k, v)


def time_ms(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "sweep.json"))
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--shapes", default=",".join(SHAPES))
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    results = {"device": torch.cuda.get_device_name(0), "shapes": {}}

    # device copy ceiling: read X bytes + write X bytes, float4 per lane (torch's copy kernel)
    a = torch.empty(256 * 1024 * 1024 // 4, device=dev)
    b = torch.empty_like(a)
    b.copy_(a)
    ms = min(time_ms(lambda: b.copy_(a), 10) for _ in range(3))
    results["copy_GBs"] = 2 * a.numel() * 4 / ms / 1e6
    del a, b
    print(f"device copy ceiling: {results['copy_GBs']:.0f} GB/s", flush=True)

    for name in args.shapes.split(","):
        B, N, M, C, res = SHAPES[name]
        L = M + 1
        g = torch.Generator(device=dev).manual_seed(1)
        Ws = [0.1 * torch.randn(B, N, L, device=dev, generator=g) for _ in range(M)]
        V0 = torch.randn(B, N, C, device=dev, generator=g)
        bytes_launch = 4 * B * N * (L + 2 * C + (C if res else 0))
        entry = {"B": B, "N": N, "M": M, "L": L, "C": C, "residual": res, "bytes_per_launch": bytes_launch,
                 "variants": {}}
        usable = []
        for vname, kn in VARIANTS:
            set_knobs(kn)
            try:
                desc = sfa.describe_fwd(B, N, L, C)
                with torch.no_grad():
                    ref = sfa.chord_chain(Ws, V0, res)
                usable.append((vname, kn, desc))
            except sfa.PSFLibraryError:
                continue
        seen = {}
        uniq = []
        for vname, kn, desc in usable:  # drop knob settings that resolve to an identical kernel
            key = (desc, kn.get("xcd_remap", 1), kn.get("chain_fused", 1), kn.get("chain_cc", 0))  # desc names tile shape, NT, full/edge
            if key in seen:
                continue
            seen[key] = vname
            uniq.append((vname, kn, desc))
        samples = {v[0]: [] for v in uniq}
        with torch.no_grad():
            for _ in range(args.rounds):
                for vname, kn, desc in uniq:
                    set_knobs(kn)
                    sfa.chord_chain(Ws, V0, res)
                    samples[vname].append(time_ms(lambda: sfa.chord_chain(Ws, V0, res), args.iters) * 1e3 / M)
        for vname, kn, desc in uniq:
            med, mn = statistics.median(samples[vname]), min(samples[vname])
            entry["variants"][vname] = {"kernel": desc, "us_per_launch_median": med, "us_per_launch_min": mn,
                                        "GBs_median": bytes_launch / med / 1e3, "GBs_best": bytes_launch / mn / 1e3}
            print(f"{name:11s} {vname:13s} {med:8.2f} us/launch  {bytes_launch / med / 1e3:7.0f} GB/s   {desc}", flush=True)
        set_knobs({})

        # backward step (dW + dV) at this shape
        W = Ws[0].clone().requires_grad_(True)
        V = V0.clone().requires_grad_(True)
        out = sfa.chord_spmm(W, V)
        dZ = torch.randn_like(out)
        from sparsefactorization_amd.chord import _launch_bwd
        dW, dV = torch.empty_like(W), torch.empty_like(V)
        Wd, Vd = W.detach(), V.detach()
        bytes_dv = 4 * B * N * (L + 2 * C)   # read W, dZ; write dV
        bytes_dw = 4 * B * N * (L + 2 * C)   # read dZ, V; write dW
        entry["bwd"] = {}
        for label, kn in (("win", {"bwd_variant": 0}), ("win_r1", {"bwd_variant": 0, "bwd_rows": 1}),
                          ("generic", {"bwd_variant": 1})):
            for k, v in {"bwd_variant": 0, "bwd_rows": 0, **kn}.items():
                sfa.set_tuning(k, v)
            f_dv = lambda: _launch_bwd(dZ, Wd, Vd, None, dV, B, N, L, C, N * C, None)  # noqa: E731
            f_dw = lambda: _launch_bwd(dZ, Wd, Vd, dW, None, B, N, L, C, N * C, None)  # noqa: E731
            f_dv(), f_dw()
            t_dv = statistics.median([time_ms(f_dv, args.iters) * 1e3 for _ in range(args.rounds)])
            t_dw = statistics.median([time_ms(f_dw, args.iters) * 1e3 for _ in range(args.rounds)])
            entry["bwd"][label] = {"dV_us": t_dv, "dW_us": t_dw, "dV_GBs": bytes_dv / t_dv / 1e3,
                                   "dW_GBs": bytes_dw / t_dw / 1e3}
            print(f"{name:11s} bwd {label:8s} dV {t_dv:8.2f} us {bytes_dv / t_dv / 1e3:6.0f} GB/s   "
                  f"dW {t_dw:8.2f} us {bytes_dw / t_dw / 1e3:6.0f} GB/s", flush=True)
        sfa.set_tuning("bwd_variant", 0)
        sfa.set_tuning("bwd_rows", 0)
        results["shapes"][name] = entry
        del Ws, V0

    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as fh:
        json.dump(results, fh, indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
