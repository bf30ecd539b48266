#!/usr/bin/env python3
"""Host time of the eager no-grad forward of an LRA network: ms per forward (host issue vs device), then cProfile over 300
forwards (Python functions by own time).   python profiles/lra_infer_host_profile.py [task] [batch]"""
import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsefactorization_amd import lra_training  # noqa: E402

task = sys.argv[1] if len(sys.argv) > 1 else "cifar10"
dev = torch.device("cuda:0")
cfg = lra_training.config[task]
batch = int(sys.argv[2]) if len(sys.argv) > 2 else cfg["training"]["batch_size"]
torch.manual_seed(42)
net = lra_training.build_model(task).to(dev).eval()
X, _ = lra_training.synthetic_split(task, batch, dev, 1)
if cfg["model"]["pooling_type"] == "CLS":
    X = lra_training.add_cls_token(X, cfg["model"]["vocab_size"])
with torch.no_grad():
    for _ in range(30):
        net(X)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(300):
        net(X)
    t_host = time.perf_counter() - t0
    e1.record(); torch.cuda.synchronize()
    print(f"{task} batch {batch}: host issues a forward in {t_host / 300 * 1e3:.3f} ms, the device takes {e0.elapsed_time(e1) / 300:.3f} ms")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(300):
        net(X)
    pr.disable()
    torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(26)
