#!/usr/bin/env bash
# Round-3 evidence, run on the GPU box from the repo root:   bash profiles/collect_r03.sh <tag>     (e.g. r03z)
#   1. profiles/collect.sh <tag>          forward headline: kernel stats + PMC passes + summary (bench.py's traffic source)
#   2. kernel stats of the training steps: Temporal Order N=16384 B=40, ListOps (reference config), ListOps N=2048 E=C=64,
#      genome N=16384 C=32 B=16 (the other families: profiles/family_step_profiles.sh)
#   3. profiles/collect_bwd.sh <tag>_bwd  backward kernels incl. the fused step: stats + PMC (L2 requests per row)
# Every rocprofv3 call has the program directly after `--`; counters and traces are never combined.
set -u
TAG=${1:-r03z}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_${TAG}_steps
mkdir -p "$OUT"
bash profiles/collect.sh "$TAG" > "$OUT/collect_fwd.log" 2>&1
echo "collect.sh rc=$?"
cd /tmp && export TMPDIR=/tmp
stats() {  # name, program args...
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$name" -- python3 "$@" > "$OUT/$name.log" 2>&1
  echo "$name rc=$?"
  { grep -E "ms/step|tokens" "$OUT/$name.log" | tail -2; python3 $ROOT/profiles/kernel_stats_top.py "$OUT/$name" 18; } > "$ROOT/gpurun_out/${TAG}_${name}_kernels.log" 2>&1
}
stats train_order $ROOT/profiles/train_step_run.py order 40
stats listops $ROOT/profiles/lra_step_run.py listops 40
stats listops64 $ROOT/profiles/lra_step_run.py listops64 40
stats genome $ROOT/profiles/genome_step_run.py 40
cd "$ROOT"
bash profiles/collect_bwd.sh "${TAG}_bwd" > "$OUT/collect_bwd.log" 2>&1
echo "collect_bwd.sh rc=$?"
# only gpurun_out/ travels back from the GPU box: the summaries the scripts wrote under profiles/ go there too
mkdir -p gpurun_out/profiles_$TAG
cp profiles/${TAG}_* profiles/${TAG}_bwd_* gpurun_out/profiles_$TAG/ 2>/dev/null
cp gpurun_out/${TAG}_*_kernels.log gpurun_out/profiles_$TAG/ 2>/dev/null
ls gpurun_out/profiles_$TAG
