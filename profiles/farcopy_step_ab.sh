set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_farcopy
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for fc in 0 1 0 1; do
  export PSF_FAR_COPY=$fc
  rm -rf $OUT/run$fc
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/run$fc -- python3 $ROOT/profiles/train_step_run.py order 40 > $OUT/run$fc.log 2>&1
  echo "far_copy=$fc rc=$?"
  grep -E "ms_per_step" $OUT/run$fc.log | tail -1 | cut -c1-200
  python3 $ROOT/profiles/kernel_stats_top.py $OUT/run$fc 6
done
