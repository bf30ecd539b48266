#!/usr/bin/env python3
"""Summarise the counter passes of profiles/collect_mlp_bwd.sh: per mlp_bwd_variant, the main kernel's average duration
(--kernel-trace --stats pass) and its counters per launch (separate --pmc passes, summed over the dispatch's rows).
    python profiles/summarize_mlp_bwd.py <tag>   reads gpurun_out/prof_<tag>/, writes profiles/<tag>_mlp_bwd_pmc.json"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main_kernel(name):
    return "mlp_bwd" in name and "pack" not in name and "reduce" not in name


def main():
    tag = sys.argv[1]
    base = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    out = {}
    for vdir in sorted(glob.glob(os.path.join(base, "v*"))):
        v = os.path.basename(vdir)
        entry = {"kernel": None, "avg_us": None, "calls": None, "counters_per_launch": {}}
        for f in glob.glob(os.path.join(vdir, "stats", "**", "*kernel_stats.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if main_kernel(row["Name"]):
                    m = re.search(r"mlp_bwd\w*(<[^>]*>)?", row["Name"])
                    entry["kernel"] = m.group(0) if m else row["Name"][:80]
                    entry["avg_us"] = float(row["AverageNs"]) / 1e3
                    entry["calls"] = int(row["Calls"])
        for f in glob.glob(os.path.join(vdir, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
            sums, dispatches = defaultdict(float), set()
            for row in csv.DictReader(open(f)):
                if main_kernel(row["Kernel_Name"]):
                    sums[row["Counter_Name"]] += float(row["Counter_Value"])
                    dispatches.add(row["Dispatch_Id"])
            for k, val in sums.items():
                entry["counters_per_launch"][k] = val / max(1, len(dispatches))
        c = entry["counters_per_launch"]
        if c.get("SQ_BUSY_CYCLES") and c.get("GRBM_GUI_ACTIVE"):
            # SQ_* cycle counters tick once per 4 clocks and are summed over the XCDs' shader engines; per-SIMD busy
            # fractions: instruction-busy cycles x 4 / (1024 SIMDs x kernel clocks)
            clk = c["GRBM_GUI_ACTIVE"] / 8.0  # GRBM_GUI_ACTIVE is summed over the 8 XCDs
            entry["derived"] = {
                "kernel_clocks": clk,
                "valu_busy_frac": c.get("SQ_ACTIVE_INST_VALU", 0) * 4 / (1024 * clk),
                "mfma_busy_frac": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * clk) if "SQ_VALU_MFMA_BUSY_CYCLES" in c else None,
                "lds_array_busy_frac": c.get("SQ_LDS_IDX_ACTIVE", 0) / (256 * clk) if "SQ_LDS_IDX_ACTIVE" in c else None,
                "lds_conflict_frac_of_active": (c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"]) if c.get("SQ_LDS_IDX_ACTIVE") else None,
            }
        out[f"mlp_bwd_variant_{v[1:]}"] = entry
    dst = os.path.join(ROOT, "profiles", f"{tag}_mlp_bwd_pmc.json")
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
