#!/usr/bin/env python3
"""Is the eager Temporal-Order training step host-bound? ms/step eager vs replayed from a HIP graph on one box, and the
Python functions the eager step spends its host time in (cProfile over 30 steps).    python profiles/host_step_profile.py"""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsefactorization_amd import psf_training  # noqa: E402

dev = torch.device("cuda:0")
for graph in (False, True):
    r = psf_training.train_benchmark("order", 16384, 40, steps=40, warmup=5, device=dev, graph=graph)
    print(f"graph={graph}: {r['seconds'] * 1e3 / r['steps']:.3f} ms/step wall, {r['event_ms'] / r['steps']:.3f} ms/step device")
pr = cProfile.Profile()
pr.enable()
psf_training.train_benchmark("order", 16384, 40, steps=30, warmup=2, device=dev, graph=False)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
