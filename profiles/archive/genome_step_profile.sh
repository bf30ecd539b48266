set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_genome; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/run -- python3 $ROOT/profiles/genome_step_run.py 40 > $OUT/run.log 2>&1
echo rc=$?
grep ms_per_step $OUT/run.log | tail -1 | cut -c1-260
python3 $ROOT/profiles/kernel_stats_top.py $OUT/run 22
