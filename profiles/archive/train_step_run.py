#!/usr/bin/env python3
"""Workload for `rocprofv3 --kernel-trace --stats`: the real training driver (psf_training.main --json) for a few steps.
    python3 profiles/train_step_run.py [order|adding] [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsefactorization_amd import psf_training  # noqa: E402

problem = sys.argv[1] if len(sys.argv) > 1 else "order"
steps = sys.argv[2] if len(sys.argv) > 2 else "40"
psf_training.main(["--problem", problem, "--n-vec", "16384", "--json", "--max-steps", steps, "--train-seqs", "1600",
                   "--eval-seqs", "40"])
