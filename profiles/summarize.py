#!/usr/bin/env python3
"""Turns the raw rocprofv3 CSVs under gpurun_out/prof_<tag>/ into the committed summaries

    profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (verbatim)
    profiles/<tag>_pmc.json           per-launch counter averages for the dominant kernel + derived traffic
    profiles/<tag>_summary.md         the numbers bench.py's roofline block should agree with

HBM traffic follows MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports exactly half of a wide coalesced streaming read, so traffic = (2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes.
The counters sit on the L2's memory-side port: Infinity-Cache hits are included (they are not HBM-only bytes).
"""
from __future__ import annotations

import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL_KEY = "chord_"  # our kernels


def find(d, pattern):
    return sorted(glob.glob(os.path.join(d, "**", pattern), recursive=True))


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    # hash of the kernel sources the counters were collected on, written by collect.sh on the GPU box at collection
    # time (sparsefactorization_amd/build.py: csrc_hash); bench.py withholds roofline.traffic when csrc has changed
    hfile = os.path.join(src, "csrc_hash.txt")
    chash = open(hfile).read().strip() if os.path.exists(hfile) else None
    out = {"tag": tag, "csrc_hash": chash, "kernels": {}}
    md = [f"# rocprofv3 summary — {tag}", "", "Command: `python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline` "
          "(N=16384, M=14, L=15, C=8, B=64; 25 chains x 14 launches)", ""]

    stats = find(os.path.join(src, "stats"), "*kernel_stats.csv")
    if stats:
        shutil.copy(stats[0], os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"))
        md += ["## Kernel durations (`--kernel-trace --stats`)", "", "| kernel | calls | avg us | total ms | % |", "|---|---|---|---|---|"]
        with open(stats[0]) as fh:
            for row in csv.DictReader(fh):
                name = row.get("Name", "")
                calls = int(float(row.get("Calls", 0)))
                avg = float(row.get("AverageNs", 0)) / 1e3
                tot = float(row.get("TotalDurationNs", 0)) / 1e6
                pct = row.get("Percentage", "")
                md.append(f"| `{name[:110]}` | {calls} | {avg:.2f} | {tot:.3f} | {pct} |")
                if KERNEL_KEY in name:
                    out["kernels"].setdefault(name, {})["avg_us"] = avg
                    out["kernels"][name]["calls"] = calls
        md.append("")

    counters = defaultdict(lambda: defaultdict(list))  # kernel -> counter -> values
    for f in find(src, "*counter_collection.csv"):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name", "")
                if KERNEL_KEY not in name:
                    continue
                try:
                    counters[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
                except (KeyError, ValueError):
                    pass
    dominant = None
    for name, cs in counters.items():
        k = out["kernels"].setdefault(name, {})
        k["counters_per_launch"] = {c: sum(v) / len(v) for c, v in cs.items()}
        k["counter_launches"] = {c: len(v) for c, v in cs.items()}
        if dominant is None or len(next(iter(cs.values()))) > len(next(iter(counters[dominant].values()))):
            dominant = name
    if dominant:
        c = out["kernels"][dominant]["counters_per_launch"]
        out["dominant_kernel"] = dominant
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            rd = 2.0 * c["FETCH_SIZE"] * 1024.0
            wr = c["WRITE_SIZE"] * 1024.0
            out["hbm_read_bytes_per_launch"] = rd
            out["hbm_write_bytes_per_launch"] = wr
            out["hbm_bytes_per_launch"] = rd + wr
        if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c and c["TCC_HIT_sum"] + c["TCC_MISS_sum"] > 0:
            out["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
        md += [f"## PMC, per launch of `{dominant[:100]}`", "", "| counter | per launch |", "|---|---|"]
        for k2, v in sorted(c.items()):
            md.append(f"| {k2} | {v:,.1f} |")
        md.append("")
        alg = 4 * 64 * 16384 * (15 + 3 * 8)
        if "hbm_bytes_per_launch" in out:
            md += [f"Memory-side traffic per launch (2*FETCH_SIZE + WRITE_SIZE, KiB -> bytes): "
                   f"**{out['hbm_bytes_per_launch'] / 1e6:.1f} MB** (read {out['hbm_read_bytes_per_launch'] / 1e6:.1f}, "
                   f"write {out['hbm_write_bytes_per_launch'] / 1e6:.1f}) vs algorithmic {alg / 1e6:.1f} MB.", ""]
        if "l2_hit_rate" in out:
            md.append(f"L2 hit rate: {out['l2_hit_rate'] * 100:.1f} %")
        avg = out["kernels"][dominant].get("avg_us")
        if avg:
            md += ["", f"Algorithmic GB/s from the rocprof average duration: {alg / avg / 1e3:.0f} GB/s "
                       f"({alg / avg / 1e3 / 8000 * 100:.1f} % of 8 TB/s)."]
    with open(os.path.join(ROOT, "profiles", f"{tag}_pmc.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    with open(os.path.join(ROOT, "profiles", f"{tag}_summary.md"), "w") as fh:
        fh.write("\n".join(md) + "\n")
    print("\n".join(md))


if __name__ == "__main__":
    main()
