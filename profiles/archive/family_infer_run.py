#!/usr/bin/env python3
"""Workload for `rocprofv3 --kernel-trace --stats`: the no-grad forward of one model family (eval mode), 40 calls.
    python3 profiles/family_infer_run.py [order|adding|genome|listops|pathfinder|imdb|cifar10|pathfinder_map|imdb_map]
*_map = ChangedPSF (LRA/attention_maps/*_inference.py): the forward plus the dense N x N attention map, batch 8 / 2."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsefactorization_amd import genome_training, lra_training, psf_training, psfnet  # noqa: E402
from sparsefactorization_amd.train import seed_everything  # noqa: E402

fam = sys.argv[1] if len(sys.argv) > 1 else "order"
for kv in filter(None, os.environ.get("PSF_TUNE", "").split(",")):  # e.g. PSF_TUNE=fwd_wide=1,fwd_wg_limit=3
    import sparsefactorization_amd as sfa
    key, val = kv.split("=")
    sfa.set_tuning(key, int(val))
dev = torch.device("cuda:0")
seed_everything(42)
g = torch.Generator(device=dev).manual_seed(1)
if fam in ("order", "adding"):
    net = psf_training.build_model(fam, 16384).to(dev)
    x, _ = psf_training.make_split(fam, 40, 16384, dev, 7)
elif fam == "genome":
    net = genome_training.build_model().to(dev)
    x, _ = genome_training.synthetic_split(16, dev, 7)
elif fam.endswith("_map"):
    task = fam[:-4]
    cfg = dict(lra_training.config[task]["model"])
    net = psfnet.ChangedPSF(**cfg).to(dev)
    B = 8 if task == "pathfinder" else 2
    x, _ = lra_training.synthetic_split(task, B, dev, 7)
    if cfg["pooling_type"] == "CLS":
        x = lra_training.add_cls_token(x, cfg["vocab_size"])
else:
    cfg = lra_training.config[fam]
    net = lra_training.build_model(fam).to(dev)
    x, _ = lra_training.synthetic_split(fam, cfg["training"]["batch_size"], dev, 7)
    if cfg["model"]["pooling_type"] == "CLS":
        x = lra_training.add_cls_token(x, cfg["model"]["vocab_size"])
net.eval()
with torch.no_grad():
    for _ in range(3):
        net(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(40):
        out = net(x)
    torch.cuda.synchronize()
print(f"{fam}: {(time.perf_counter() - t0) / 40 * 1e3:.3f} ms per forward, input {tuple(x.shape)}")
