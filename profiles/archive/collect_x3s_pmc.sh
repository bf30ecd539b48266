#!/usr/bin/env bash
# SQ counter passes on the producer-MLP forward kernels (x3_fwd_k and the unit-stationary x3s_fwd_k) in the lab binary
# (profiles/x3flab.hip built with -DPSF_X3F_NOTRACE), T = 1 M tokens, 15 MLPs of E = h = 32:
#     bash profiles/collect_x3s_pmc.sh <tag>
set -u
TAG=${1:-r05}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_${TAG}_x3s_pmc
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {
  local name=$1; shift
  timeout -k 10 200 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -- $ROOT/profiles/bin/x3flab 1048576 quick > "$OUT/$name.log" 2>&1
  echo "$name rc=$?"
}
run pmc_time --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
run pmc_inst --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run pmc_mem --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES
run pmc_misc --pmc SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_TRANS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_FLAT SQ_WAIT_INST_ANY
cd "$ROOT"
python3 - "$OUT" <<'PY' | tee gpurun_out/${TAG}_x3s_pmc.txt
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "x3s_fwd_k" in k or "x3_fwd_k" in k:
            acc[k.split("(")[0][-60:]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print("==", k)
    for c, v in sorted(d.items()):
        print(f"  {c:32s} per launch {sum(v) / len(v):16.1f}   launches {len(v)}")
PY
