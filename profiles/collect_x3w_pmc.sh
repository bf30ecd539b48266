#!/usr/bin/env bash
# SQ counter passes on the producer-MLP forward kernels x3_fwd_k and the woven x3w_fwd_k (lab build with
# PSF_HIPCC_EXTRA=-DPSF_X3W_ABLATE_LAB), T = 1 M tokens, 15 MLPs of E = h = 32:   bash profiles/collect_x3w_pmc.sh <tag>
set -u
TAG=${1:-r06b}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_${TAG}_x3w_pmc
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {
  local name=$1; shift
  timeout -k 10 200 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -- python3 $ROOT/profiles/mlp_fwd_weave_ablate.py 0 4 > "$OUT/$name.log" 2>&1
  echo "$name rc=$?"
}
run stats --kernel-trace --stats &&
run pmc_time --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE &&
run pmc_inst --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
cd "$ROOT"
python3 - "$OUT" <<'PY' | tee gpurun_out/${TAG}_x3w_pmc.txt
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "x3w_fwd_k" in k or "x3_fwd_k" in k:
            acc[k.split("(")[0][-40:]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "x3" in row["Name"]:
            print(f"stats  {row['Name'][:60]:60s} calls {row['Calls']:>5s}  avg {float(row['AverageNs']) / 1e3:9.1f} us")
for k, d in acc.items():
    print("==", k)
    for c, v in sorted(d.items()):
        print(f"  {c:32s} per launch {sum(v) / len(v):16.1f}   launches {len(v)}")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "SQ_BUSY_CYCLES" in d:
        m = lambda c: sum(d[c]) / len(d[c])
        print(f"  matrix busy / SQ busy = {m('SQ_VALU_MFMA_BUSY_CYCLES') / m('SQ_BUSY_CYCLES'):.3f}   both at once / SQ busy = {m('SQ_VALU_MFMA_COEXEC_CYCLES') / m('SQ_BUSY_CYCLES'):.3f}"
              f"   both at once / matrix busy = {m('SQ_VALU_MFMA_COEXEC_CYCLES') / m('SQ_VALU_MFMA_BUSY_CYCLES'):.3f}")
PY
