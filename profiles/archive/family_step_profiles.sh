set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_families; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; rm -rf $OUT/$name; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -- python3 "$@" > $OUT/$name.log 2>&1; echo "== $name rc=$?"; grep -E "ms_per_step|ms/step" $OUT/$name.log | tail -1 | cut -c1-200; python3 $ROOT/profiles/kernel_stats_top.py $OUT/$name 9; }
run genome $ROOT/profiles/genome_step_run.py 40
run pathfinder $ROOT/profiles/lra_step_run.py pathfinder 40
run imdb $ROOT/profiles/lra_step_run.py imdb 40
run cifar10 $ROOT/profiles/lra_step_run.py cifar10 40
run adding $ROOT/profiles/train_step_run.py adding 40
