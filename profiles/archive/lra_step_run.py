#!/usr/bin/env python3
"""Workload for `rocprofv3 --kernel-trace --stats`: N training steps of the LRA driver's loop body on one GPU.
    python3 profiles/lra_step_run.py [listops|listops64|pathfinder|imdb|cifar10] [steps] [graph]
listops64 = BASELINE.json's wording of configs[2]: N = 2048, E = C = 64."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsefactorization_amd import lra_training  # noqa: E402

task = sys.argv[1] if len(sys.argv) > 1 else "listops"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
over = dict(n_vec=2048, embedding_size=64, n_channels_V=64) if task == "listops64" else {}
graph = len(sys.argv) > 3 and sys.argv[3] == "graph"  # the step replayed from a HIP graph (train.GraphedStep)
r = lra_training.train_benchmark("listops" if task == "listops64" else task, steps=steps, warmup=3, device=torch.device("cuda:0"),
                                 graph=graph, **over)
print(f"{task}{' (HIP graph)' if graph else ''}: {r['seconds'] * 1e3 / r['steps']:.3f} ms/step wall, {r['event_ms'] / r['steps']:.3f} ms/step device, "
      f"N={r['n_vec']} E={r['embedding_size']} C={r['n_channels_V']} B={r['batch']} loss {r['loss']:.4f}")
