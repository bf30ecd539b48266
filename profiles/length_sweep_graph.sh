#!/usr/bin/env bash
# The reference's synthetic experiments (SyntheticExperiments/synth_data_generation.py:75-80: 200 000 / 5 000 / 5 000 sequences;
# synthetic_training_config.py: batch 40, Adam lr 1e-3, 20 epochs) at the lengths given on the command line, both tasks, data
# generated on the device, the optimisation step replayed from a HIP graph:   bash profiles/length_sweep_graph.sh 128 256 ...   -> gpurun_out/r06s_sweep_<task>_n<N>.log
set -u
for n in "$@"; do
  for task in adding order; do
    out=gpurun_out/r06s_sweep_${task}_n${n}.log
    python -m sparsefactorization_amd.psf_training --problem $task --n-vec $n --train-seqs 200000 --eval-seqs 5000 --epochs 20 --graph > $out 2>&1
    echo "$task N=$n: $(grep 'Test accuracy' $out | sort -t: -k2 -n | tail -1) (best epoch); last: $(grep 'Test accuracy' $out | tail -1); $(grep -c 'Training loss' $out) epochs, $(grep 'Training loss' $out | tail -1 | sed 's/.*Time: *//')"
  done
done
