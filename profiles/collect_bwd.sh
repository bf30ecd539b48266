#!/usr/bin/env bash
# rocprofv3 evidence for the backward kernels (chord_dv_win_k / chord_dw_win_k / chord_bwd_fused_k) and, since round 5, the forward
# step kernel of every shape of bench.py's `shapes` leg.
#     bash profiles/collect_bwd.sh <tag>        e.g. r02_bwd  -> gpurun_out/prof_<tag>/<shape>/<pass>/
# One --kernel-trace --stats pass and one pass per --pmc group (never combined), program directly after `--`.
set -u
TAG=${1:-r02_bwd}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf "$OUT"  # (gpurun MERGES gpurun_out/ back: delete the local copy of an earlier collection under the same tag too, or the
              #  summariser averages both)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp

run() {  # shape-name, pass-name, "B N L C", rocprofv3 args...
  local shape=$1 name=$2 dims=$3; shift 3
  echo "== $shape $name" | tee -a "$OUT/collect.log"
  timeout -k 10 240 rocprofv3 "$@" --output-format csv -d "$OUT/$shape/$name" -- python3 $ROOT/profiles/bwd_pmc_run.py $dims 100 >> "$OUT/collect.log" 2>&1
  local rc=$?
  echo "rc=$rc" | tee -a "$OUT/collect.log"
  return $rc
}

for spec in "cfg2:64 16384 15 8" "order_train:40 16384 15 8" "genome_like:16 16384 15 32" "cfg3_ref:32 2000 12 128" "cfg3_baseline:32 2048 12 64" "cfg4:64 1024 12 32"; do
  shape=${spec%%:*}; dims=${spec#*:}
  run $shape stats "$dims" --kernel-trace --stats &&
  run $shape pmc_fetch "$dims" --pmc FETCH_SIZE &&
  run $shape pmc_write "$dims" --pmc WRITE_SIZE &&
  run $shape pmc_l2 "$dims" --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum &&
  run $shape pmc_sq "$dims" --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE &&
  run $shape pmc_inst "$dims" --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR || exit 1
done
cd "$ROOT"
python3 profiles/summarize_bwd.py "$TAG" | tee -a "$OUT/collect.log"
