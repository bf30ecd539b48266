#!/usr/bin/env python3
"""Do the automatic launch rules hold inside REAL training steps? Each knob's values, interleaved in one process, per task:
GPU ms per training step (100 steps per reading, best and median of five).   python profiles/step_knob_sweep.py [task ...]
(The rules were set on single kernels with rotating operands; round 5 found one — non-temporal dV — that the step reverses.)"""
import os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsefactorization_amd as sfa  # noqa: E402
from sparsefactorization_amd import genome_training, lra_training, psf_training  # noqa: E402
from sparsefactorization_amd.train import make_adam  # noqa: E402

dev = torch.device("cuda:0")
KNOBS = {  # knob: values (first = default)
    "bwd_fused_wg_limit": (0, 2, 3, 4, 5),
    "fwd_wg_limit": (0, 1, 2, 3, 4),
    "bwd_fused": (1, 0),
    "chain_zigzag": (1, 0),
    "xcd_remap": (1, 0),
    "fwd_wide": (0, 1, 2, 4),
    "dw_variant": (0, 1),
    "dw_tgs": (0, 4, 5),
    "dv_threads": (0, 1),
    "chain_fused": (1, 0, 2),
    "fwd_split": (1, 0, 2),
}
if os.environ.get("PSF_SWEEP_KNOBS"):  # e.g. PSF_SWEEP_KNOBS="bwd_ablate=0,1024,2048" (lab knobs; first value = the default)
    KNOBS = {kv.split("=")[0]: tuple(int(v) for v in kv.split("=")[1].split(",")) for kv in os.environ["PSF_SWEEP_KNOBS"].split(";")}
for task in (sys.argv[1:] or ["order", "genome", "imdb", "pathfinder", "listops", "cifar10"]):
    torch.manual_seed(42)
    loss = torch.nn.CrossEntropyLoss()
    if task == "genome":
        batch = 16
        net = genome_training.build_model().to(dev)
        opt = make_adam(net.parameters(), 1e-4)
        X, Y = genome_training.synthetic_split(batch, dev, 1)
    elif task == "order":
        batch = 40
        net = psf_training.build_model("order", 16384).to(dev)
        opt = make_adam(net.parameters(), psf_training.config["order"]["training"]["learning_rate"])
        X, Y = psf_training.make_split("order", batch, 16384, dev, 1000)
    else:
        cfg = lra_training.config[task]
        batch = cfg["training"]["batch_size"]
        net = lra_training.build_model(task).to(dev)
        opt = make_adam(net.parameters(), cfg["training"]["learning_rate"])
        X, Y = lra_training.synthetic_split(task, batch, dev, 1)
        if cfg["model"]["pooling_type"] == "CLS":
            X = lra_training.add_cls_token(X, cfg["model"]["vocab_size"])

    def step():
        opt.zero_grad(set_to_none=True)
        out = loss(net(X).squeeze(), Y)
        out.backward()
        opt.step()

    def reading(n=100):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(n):
            step()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    reading(30)
    for knob, values in KNOBS.items():
        t = {v: [] for v in values}
        try:
            for rd in range(5):
                for v in (values if rd % 2 == 0 else values[::-1]):
                    sfa.set_tuning(knob, v)
                    try:
                        t[v].append(reading())
                    except Exception as exc:  # a forced variant that does not apply to this network's shapes
                        t[v].append(float("nan"))
        finally:
            sfa.set_tuning(knob, values[0])
        base = min(t[values[0]])
        print(f"{task:10s} {knob:20s} " + "  ".join(f"{v}: {min(t[v]):.3f} ({statistics.median(t[v]):.3f}){'' if v == values[0] else f' {(base / min(t[v]) - 1) * 100:+.1f}%'}" for v in values), flush=True)
    del net, opt
    torch.cuda.empty_cache()
