#!/usr/bin/env python3
"""Backward of one chain step = dV (needs dZ, W) and dW (needs dZ, V): independent of each other. Do they finish sooner
side by side on two streams than one after the other? Temporal-Order training shape, cold operand sets walked round-robin
(profiles/cold_sweep.py), each schedule captured once into a HIP graph (no host launch time in the comparison).

    python profiles/dvdw_overlap_lab.py [--steps 14] [--rounds 7]
"""
import argparse
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sparsefactorization_amd import chord  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=14)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--batch", type=int, default=40)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    B, N, L, C = args.batch, 16384, 15, 8
    S = args.steps
    g = torch.Generator(device=dev).manual_seed(1)
    sets = [{"W": 0.1 * torch.randn(B, N, L, device=dev, generator=g), "V": torch.randn(B, N, C, device=dev, generator=g),
             "dZ": torch.randn(B, N, C, device=dev, generator=g), "dV": torch.empty(B, N, C, device=dev),
             "dW": torch.empty(B, N, L, device=dev)} for _ in range(S)]
    side = torch.cuda.Stream(device=dev)

    def dv(s):
        chord._launch_bwd(s["dZ"], s["W"], s["V"], None, s["dV"], B, N, L, C, N * C, None)

    def dw(s):
        chord._launch_bwd(s["dZ"], s["W"], s["V"], s["dW"], None, B, N, L, C, N * C, None)

    def sequential():
        for s in sets:
            dv(s)
            dw(s)

    def two_streams():
        # like the real backward: dV_m first (the next step waits for its result), dW_m beside dV_{m-1}
        main = torch.cuda.current_stream()
        for s in sets:
            ev = torch.cuda.Event()
            ev.record(main)
            dv(s)
            side.wait_event(ev)
            with torch.cuda.stream(side):
                dw(s)
        main.wait_stream(side)

    graphs = {}
    for name, fn in (("one stream: dV, dW, dV, dW, ...", sequential), ("two streams: dW_m beside dV_(m-1)", two_streams)):
        fn()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            fn()
        graphs[name] = gr
    times = {k: [] for k in graphs}
    for r in range(args.rounds + 1):
        for name, gr in graphs.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            gr.replay()
            e1.record()
            torch.cuda.synchronize()
            if r:
                times[name].append(e0.elapsed_time(e1) * 1e3 / S)
    for name, t in times.items():
        print(f"B={B}: {name:40s} {statistics.median(t):7.2f} us per step (dV + dW)")


if __name__ == "__main__":
    main()
