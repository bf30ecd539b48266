#!/usr/bin/env bash
# Collects the rocprofv3 evidence for bench.py's roofline numbers. Run on the GPU box from the repo root:
#     bash profiles/collect.sh <tag>            e.g.  bash profiles/collect.sh r01
# Pass 1: --kernel-trace --stats (per-kernel durations). Passes 2..: one --pmc group each (TCC has 4 slots:
# FETCH_SIZE takes 3, WRITE_SIZE 2 — MI355X_MICROARCH.md "rocprofv3 PMC slots"), never combined with traces.
# Raw output goes to gpurun_out/prof_<tag>/ (scratch); profiles/summarize.py turns it into profiles/<tag>_*.
set -u
TAG=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf "$OUT"  # (gpurun MERGES gpurun_out/ back: delete the local copy of an earlier collection under the same tag too, or the
              #  summariser averages both)
mkdir -p "$OUT"
python3 -c "from sparsefactorization_amd.build import csrc_hash; print(csrc_hash())" > "$OUT/csrc_hash.txt"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train --no-extra-legs"

run() {  # name, rocprofv3 args...
  local name=$1; shift
  echo "== $name" | tee -a "$OUT/collect.log"
  timeout -k 10 300 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -- $BENCH >> "$OUT/collect.log" 2>&1
  echo "rc=$?" | tee -a "$OUT/collect.log"
}

run stats --kernel-trace --stats
run pmc_fetch --pmc FETCH_SIZE
run pmc_write --pmc WRITE_SIZE
run pmc_l2 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run pmc_ea --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
run pmc_sq --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
run pmc_lds --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
cd "$ROOT"
python3 profiles/summarize.py "$TAG" | tee -a "$OUT/collect.log"
