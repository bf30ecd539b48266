for n in 128 1024; do
  for knob in 1 0; do
    python - <<PY
import time, torch, sparsefactorization_amd as sfa
from sparsefactorization_amd import psf_training
sfa.set_tuning("chain_bwd_fused", $knob)
import sys
sys.argv = ["x", "--problem", "order", "--n-vec", "$n", "--train-seqs", "40000", "--eval-seqs", "400", "--epochs", "2"]
import io, contextlib
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    psf_training.main()
lines = [l for l in buf.getvalue().splitlines() if "Training loss" in l]
print("N=$n chain_bwd_fused=$knob eager:", lines[-1])
PY
  done
done
