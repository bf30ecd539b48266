#!/usr/bin/env python3
"""Workload for `rocprofv3 --kernel-trace --stats`: training steps of the genome driver (N = 16384, C = 32, B = 16, gradient
clip at 1.0) on one GPU.    python3 profiles/genome_step_run.py [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsefactorization_amd import genome_training  # noqa: E402

steps = sys.argv[1] if len(sys.argv) > 1 else "40"
genome_training.main(["--json", "--max-steps", steps, "--train-seqs", "704"])
