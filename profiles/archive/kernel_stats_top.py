#!/usr/bin/env python3
"""Top kernels of a `rocprofv3 --kernel-trace --stats` run, one line each (total ms, share, calls, average us, name).
    python profiles/kernel_stats_top.py <dir with *_kernel_stats.csv> [rows]"""
import csv
import glob
import os
import sys

d = sys.argv[1]
rows_max = int(sys.argv[2]) if len(sys.argv) > 2 else 16
files = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
rows = [r for f in files for r in csv.DictReader(open(f))]
total = sum(float(r["TotalDurationNs"]) for r in rows)
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:rows_max]:
    t = float(r["TotalDurationNs"])
    print(f"{t / 1e6:9.3f} ms {100 * t / total:6.2f}%  calls {int(r['Calls']):5d}  avg {float(r['AverageNs']) / 1e3:9.2f} us  {r['Name'][:110]}")
