set -u
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/prof_r06w_families; rm -rf $OUT; mkdir -p $OUT
export PYTHONPATH=$ROOT
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -- python3 "$@" > $OUT/$name.log 2>&1; echo "== $name rc=$?"; grep -E "ms/step" $OUT/$name.log | tail -1 | cut -c1-200; python3 $ROOT/profiles/archive/kernel_stats_top.py $OUT/$name 12; }
run listops $ROOT/profiles/archive/lra_step_run.py listops 40
run listops64 $ROOT/profiles/archive/lra_step_run.py listops64 40
run pathfinder $ROOT/profiles/archive/lra_step_run.py pathfinder 40
run imdb $ROOT/profiles/archive/lra_step_run.py imdb 40
