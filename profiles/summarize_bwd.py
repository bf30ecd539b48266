#!/usr/bin/env python3
"""gpurun_out/prof_<tag>/<shape>/<pass>/ (profiles/collect_bwd.sh) -> profiles/<tag>_pmc.json + profiles/<tag>_summary.md.

Per shape and per chord kernel: rocprofv3 average duration, counters per launch, memory-side traffic
(2*FETCH_SIZE + WRITE_SIZE KiB, the gfx950 correction of MI355X_MICROARCH.md) against the algorithmic bytes
4*B*N*(L+2C) of one backward kernel, L2 hit rate, VMEM instructions per wave.
"""
from __future__ import annotations

import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = {"cfg2": (64, 16384, 15, 8), "order_train": (40, 16384, 15, 8), "genome_like": (16, 16384, 15, 32),
          "cfg3_ref": (32, 2000, 12, 128), "cfg3_baseline": (32, 2048, 12, 64), "cfg4": (64, 1024, 12, 32)}


def short(name: str) -> str:
    i = name.find("chord_")
    j = name.find(">(", i)
    return name[i:j + 1] if i >= 0 and j >= 0 else name[:80]


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02_bwd"
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    sys.path.insert(0, ROOT)
    from sparsefactorization_amd.build import csrc_hash
    out = {"tag": tag, "csrc_hash": csrc_hash(), "shapes": {}}  # (bench.py attaches these numbers only to the sources they came from)
    md = [f"# rocprofv3 summary, per shape: forward step, dV, dW and fused backward kernels — {tag}", "",
          "Workload: `python3 profiles/bwd_pmc_run.py B N L C 100` (100 launches of dV alone, 100 of dW alone, 100 of the fused "
          "step where it applies — its algorithmic bytes are 4BN(2L+3C)); one "
          "`--kernel-trace --stats` pass and one pass per `--pmc` group. Traffic = (2*FETCH_SIZE + WRITE_SIZE) KiB.", ""]
    for shape, (B, N, L, C) in SHAPES.items():
        sdir = os.path.join(src, shape)
        if not os.path.isdir(sdir):
            continue
        alg = 4 * B * N * (L + 2 * C)
        kernels = defaultdict(dict)
        for f in glob.glob(os.path.join(sdir, "stats", "**", "*kernel_stats.csv"), recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    if "chord_" in row.get("Name", ""):
                        k = kernels[short(row["Name"])]
                        k["avg_us"] = float(row["AverageNs"]) / 1e3
                        k["calls"] = int(float(row["Calls"]))
        vals = defaultdict(lambda: defaultdict(list))
        for f in glob.glob(os.path.join(sdir, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    if "chord_" in row.get("Kernel_Name", ""):
                        try:
                            vals[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
                        except (KeyError, ValueError):
                            pass
        for name, cs in vals.items():
            kernels[name]["counters_per_launch"] = {c: sum(v) / len(v) for c, v in cs.items()}
        md += [f"## {shape}: B={B} N={N} L={L} C={C}; algorithmic bytes per kernel 4BN(L+2C) = {alg / 1e6:.2f} MB", "",
               "| kernel | calls | avg us | alg TB/s | of 8 TB/s | traffic MB (rd+wr) | traffic/alg | L2 hit | VMEM rd/wave | wait-inst/wave-cycles | L2 requests / row |",
               "|---|---|---|---|---|---|---|---|---|---|---|"]
        for name, k in sorted(kernels.items()):
            c = k.get("counters_per_launch", {})
            row = [f"`{name}`", str(k.get("calls", "")), f"{k.get('avg_us', float('nan')):.2f}"]
            fused = "fused" in name
            fwd = "chord_fwd_" in name
            chain = "chord_chain_" in name  # ONE launch for all L - 1 steps of a chain (round 6)
            full = k.get("calls", 0) >= 50 and ("false>" in name or fused or fwd or chain)  # the main (full-tile) launch of the shape
            alg_k = 4 * B * N * (2 * L + 3 * C) if fused else ((L - 1) * alg if chain else alg)  # the fused step: 4BN(2L+3C)
            k["alg_bytes_per_launch"] = alg_k
            if "avg_us" in k and full:
                tbs = alg_k / k["avg_us"] / 1e6
                k["alg_tbs"], k["frac"] = tbs, tbs / 8.0
                row += [f"{tbs:.2f}", f"{tbs / 8.0:.3f}"]
            else:
                row += ["", ""]
            if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                rd, wr = 2.0 * c["FETCH_SIZE"] * 1024.0, c["WRITE_SIZE"] * 1024.0
                k["traffic_bytes"], k["traffic_read"], k["traffic_write"] = rd + wr, rd, wr
                k["traffic_over_alg"] = (rd + wr) / alg_k
                row += [f"{(rd + wr) / 1e6:.1f} ({rd / 1e6:.1f}+{wr / 1e6:.1f})", f"{(rd + wr) / alg_k:.3f}" if full else ""]
            else:
                row += ["", ""]
            if c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0) > 0:
                k["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
                row.append(f"{k['l2_hit_rate'] * 100:.1f} %")
            else:
                row.append("")
            row.append(f"{c['SQ_INSTS_VMEM_RD'] / c['SQ_WAVES']:.1f}" if "SQ_INSTS_VMEM_RD" in c and c.get("SQ_WAVES") else
                       (f"{c['SQ_INSTS_VMEM_RD']:.0f}/launch" if "SQ_INSTS_VMEM_RD" in c else ""))
            row.append(f"{c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES']:.2f}" if c.get("SQ_WAVE_CYCLES") and "SQ_WAIT_INST_ANY" in c else "")
            if "TCC_REQ_sum" in c:
                k["tcc_req_per_row"] = c["TCC_REQ_sum"] / (B * N)
            row.append(f"{c['TCC_REQ_sum'] / (B * N):.2f}" if "TCC_REQ_sum" in c else "")
            md.append("| " + " | ".join(row) + " |")
        md.append("")
        out["shapes"][shape] = {"B": B, "N": N, "L": L, "C": C, "alg_bytes": alg, "kernels": kernels}
    with open(os.path.join(ROOT, "profiles", f"{tag}_pmc.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    with open(os.path.join(ROOT, "profiles", f"{tag}_summary.md"), "w") as fh:
        fh.write("\n".join(md) + "\n")
    print("\n".join(md))


if __name__ == "__main__":
    main()
