#!/usr/bin/env python3
"""Table of the synthetic experiments' logs (profiles/length_sweep_graph.sh): per length and task, test accuracy at the
best-validation epoch (the reference's model selection), best and last test accuracy, seconds per epoch.
    python profiles/summarize_length_sweep.py gpurun_out/r06s_sweep"""
import re
import sys
import os

prefix = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r06s_sweep"


def read(path):
    val, test, secs = [], [], []
    for line in open(path, errors="replace"):
        if m := re.match(r"Epoch (\d+) - Training loss:\s+\S+ — Time:\s+([0-9.]+)sec", line):
            secs.append(float(m.group(2)))
        elif m := re.match(r"Val\s+accuracy:\s+([0-9.]+)", line):
            val.append(float(m.group(1)))
        elif m := re.match(r"Test accuracy:\s+([0-9.]+)", line):
            test.append(float(m.group(1)))
    return val, test, secs


def cell(path):
    if not os.path.exists(path):
        return "—", "—"
    val, test, secs = read(path)
    n = min(len(val), len(test))
    if n == 0:
        return "no epochs", "—"
    best = max(range(n), key=lambda e: (val[e], -e))
    steady = sorted(secs[1:] or secs)[len(secs[1:] or secs) // 2]
    return f"{test[best]:.2f} ({best}) / {max(test[:n]):.2f} / {test[n - 1]:.2f}", f"{steady:.2f}"


print("| N | Adding: test accuracy at the best-validation epoch (epoch) / best / last | s per epoch | Temporal Order: the same | s per epoch |")
print("|---|---|---|---|---|")
for n in (128, 256, 512, 1024, 2048, 4096, 8192, 16384):
    a, ta = cell(f"{prefix}_adding_n{n}.log")
    o, to = cell(f"{prefix}_order_n{n}.log")
    print(f"| {n} | {a} | {ta} | {o} | {to} |")
