#!/usr/bin/env bash
# Counter passes on the fused MLP forward kernel (x3_fwd_k) at the Adding inference shape (T = 64*16384, 15 MLPs).
set -u
TAG=${1:-r02_mlp_fwd}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {
  local name=$1; shift
  echo "== $name" | tee -a "$OUT/collect.log"
  timeout -k 10 240 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -- python3 $ROOT/profiles/mlp_fwd_bench.py >> "$OUT/collect.log" 2>&1
  echo "rc=$?" | tee -a "$OUT/collect.log"
}
run stats --kernel-trace --stats
run pmc_time --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE
run pmc_inst --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run pmc_mem --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY
