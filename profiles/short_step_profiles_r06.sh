set -u
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/prof_r06x_short; rm -rf $OUT; mkdir -p $OUT
export PYTHONPATH=$ROOT
cd /tmp && export TMPDIR=/tmp
for n in 128 1024 2048; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/n$n -- python3 -m sparsefactorization_amd.psf_training --problem order --n-vec $n --train-seqs 40000 --eval-seqs 400 --epochs 1 --graph > $OUT/n$n.log 2>&1
  echo "== N=$n rc=$? $(grep 'Training loss' $OUT/n$n.log | tail -1 | cut -c1-120)"
  python3 $ROOT/profiles/archive/kernel_stats_top.py $OUT/n$n 14
done
