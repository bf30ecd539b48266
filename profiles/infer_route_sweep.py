#!/usr/bin/env python3
"""Does the automatic route of the fused mixer hold in the REAL no-grad forward of every network family? Each family's forward
(eval mode, the reference's batch size) with fused_mixer.route = auto / always / never, interleaved in one process: GPU ms per
forward (50 forwards per reading, best and median of five), logits compared.   python profiles/infer_route_sweep.py [family ...]"""
import os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsefactorization_amd import fused_mixer, genome_training, lra_training, psf_training  # noqa: E402
from sparsefactorization_amd.train import seed_everything  # noqa: E402

dev = torch.device("cuda:0")
for spec in (sys.argv[1:] or ["adding", "order", "genome", "imdb", "pathfinder", "cifar10", "listops"]):
    fam, _, bs = spec.partition(":")  # family[:batch]
    seed_everything(42)
    if fam in ("order", "adding"):
        net = psf_training.build_model(fam, 16384).to(dev)
        x, _ = psf_training.make_split(fam, int(bs or 64), 16384, dev, 7)
    elif fam == "genome":
        net = genome_training.build_model().to(dev)
        x, _ = genome_training.synthetic_split(int(bs or 16), dev, 7)
    else:
        cfg = lra_training.config[fam]
        net = lra_training.build_model(fam).to(dev)
        x, _ = lra_training.synthetic_split(fam, int(bs or cfg["training"]["batch_size"]), dev, 7)
        if cfg["model"]["pooling_type"] == "CLS":
            x = lra_training.add_cls_token(x, cfg["model"]["vocab_size"])
    net.eval()

    def reading(n=50):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        with torch.no_grad():
            for _ in range(n):
                y = net(x)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n, y
    t, outs = {}, {}
    try:
        for rd in range(6):
            for route in (("auto", "always", "never") if rd % 2 == 0 else ("never", "always", "auto")):
                fused_mixer.route = route
                r, y = reading(10 if rd == 0 else 50)
                if rd:
                    t.setdefault(route, []).append(r)
                outs[route] = y.float().clone()
    finally:
        fused_mixer.route = "auto"
    ref = outs["never"]
    scale = float(ref.abs().max())
    print(f"{spec:14s} tokens {x.shape[0] * x.shape[1]:8d} " + "  ".join(f"{k}: {min(v):.3f} ({statistics.median(v):.3f}) ms" for k, v in t.items())
          + "   max |logit difference| vs never / max |logit|: " + ", ".join(f"{k} {float((outs[k] - ref).abs().max()) / scale:.1e}" for k in ("auto", "always")), flush=True)
    del net
    torch.cuda.empty_cache()
