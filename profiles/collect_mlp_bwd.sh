#!/usr/bin/env bash
# Counter passes on the fused MLP backward kernels (mlp_bwd_k / mlp_bwd_x3_k) at the Temporal-Order training shape.
#     [VARIANTS="1 2 0"] bash profiles/collect_mlp_bwd.sh <tag>   -> gpurun_out/prof_<tag>/v<variant>/<pass>/
# then, here:  python profiles/summarize_mlp_bwd.py <tag>   -> profiles/<tag>_mlp_bwd_pmc.json
set -u
TAG=${1:-r02_mlp_bwd}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {  # variant, pass-name, rocprofv3 args...
  local v=$1 name=$2; shift 2
  echo "== v$v $name" | tee -a "$OUT/collect.log"
  timeout -k 10 240 rocprofv3 "$@" --output-format csv -d "$OUT/v$v/$name" -- python3 $ROOT/profiles/mlp_bwd_pmc_run.py $v 5 >> "$OUT/collect.log" 2>&1
  local rc=$?
  echo "rc=$rc" | tee -a "$OUT/collect.log"
  return $rc
}
for v in ${VARIANTS:-1 2 0}; do
  run $v stats --kernel-trace --stats &&
  run $v pmc_time --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE &&
  run $v pmc_inst --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE || exit 1
done
