#!/usr/bin/env python3
"""Training step of the LRA networks and the genome network (synthetic tokens, the reference's batch sizes) with two builds of
the library in one process, arms interleaved:   python profiles/lra_step_ab.py other/libpsf_chord.so [task ...]
ms per step (wall, and GPU time between two events; 100 steps per reading), median of seven readings per arm."""
import ctypes, os, statistics, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsefactorization_amd import _lib, genome_training, lra_training, psf_training  # noqa: E402
from sparsefactorization_amd.train import make_adam  # noqa: E402

dev = torch.device("cuda:0")
new = _lib.load()
old = ctypes.CDLL(os.path.abspath(sys.argv[1]))
for fname, (argtypes, restype) in _lib.SIGNATURES.items():
    fn = getattr(old, fname, None)
    if fn is not None:
        fn.argtypes, fn.restype = argtypes, restype
for task in (sys.argv[2:] or ["listops", "imdb", "pathfinder", "cifar10", "genome", "order"]):
    torch.manual_seed(42)
    loss = torch.nn.CrossEntropyLoss()
    if task == "genome":
        batch = genome_training.config["DDcDNA"]["training"]["batch_size"]
        net = genome_training.build_model().to(dev)
        opt = make_adam(net.parameters(), 1e-4)
        X, Y = genome_training.synthetic_split(batch, dev, 1)
    elif task == "order":  # BASELINE configs[4]: Temporal Order, N = 16384, 40 sequences per step
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
        t0 = time.perf_counter()
        e0.record()
        for _ in range(n):
            step()
        e1.record()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3, e0.elapsed_time(e1) / n
    t = {"old": [], "new": []}
    for rd in range(8):
        for key, lib in ((("old", old), ("new", new)) if rd % 2 == 0 else (("new", new), ("old", old))):
            _lib._lib = lib
            r = reading(20 if rd == 0 else 100)
            if rd:
                t[key].append(r)
    _lib._lib = new
    o, w = statistics.median(x[0] for x in t["old"]), statistics.median(x[0] for x in t["new"])
    og, wg = statistics.median(x[1] for x in t["old"]), statistics.median(x[1] for x in t["new"])
    print(f"{task:11s} GPU ms/step, best of seven: old {min(x[1] for x in t['old']):.3f} new {min(x[1] for x in t['new']):.3f}", flush=True)
    print(f"{task:11s} batch {batch}: wall old {o:.3f} new {w:.3f} ms/step ({(o / w - 1) * 100:+5.1f} %)   GPU old {og:.3f} new {wg:.3f} ({(og / wg - 1) * 100:+5.1f} %)"
          f"   new readings {' '.join(f'{x[1]:.3f}' for x in t['new'])} | old {' '.join(f'{x[1]:.3f}' for x in t['old'])}", flush=True)
    del net, opt
    torch.cuda.empty_cache()
