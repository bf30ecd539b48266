#!/usr/bin/env bash
# rocprofv3 kernel stats of every model family's no-grad forward (profiles/family_infer_run.py): top kernels
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_infer; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for fam in order genome listops pathfinder imdb cifar10 pathfinder_map imdb_map; do
  rm -rf $OUT/$fam
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$fam -- python3 $ROOT/profiles/family_infer_run.py $fam > $OUT/$fam.log 2>&1
  echo "== $fam rc=$?"; grep "ms per forward" $OUT/$fam.log | tail -1
  python3 $ROOT/profiles/kernel_stats_top.py $OUT/$fam 7
done
