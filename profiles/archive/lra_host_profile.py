#!/usr/bin/env python3
"""Where does the host time of an eager LRA training step go? wall vs device ms per step, then cProfile over 100 steps without
device synchronisation inside the loop (the Python functions by own time).   python profiles/lra_host_profile.py [task]"""
import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsefactorization_amd import lra_training  # noqa: E402
from sparsefactorization_amd.train import make_adam  # noqa: E402

task = sys.argv[1] if len(sys.argv) > 1 else "cifar10"
dev = torch.device("cuda:0")
cfg = lra_training.config[task]
batch = cfg["training"]["batch_size"]
torch.manual_seed(42)
net = lra_training.build_model(task).to(dev)
opt = make_adam(net.parameters(), cfg["training"]["learning_rate"])
loss = torch.nn.CrossEntropyLoss()
X, Y = lra_training.synthetic_split(task, batch, dev, 1)
if cfg["model"]["pooling_type"] == "CLS":
    X = lra_training.add_cls_token(X, cfg["model"]["vocab_size"])


def step():
    opt.zero_grad(set_to_none=True)
    out = loss(net(X).squeeze(), Y)
    out.backward()
    opt.step()


for _ in range(20):
    step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter(); e0.record()
for _ in range(200):
    step()
t_host = time.perf_counter() - t0
e1.record(); torch.cuda.synchronize()
print(f"{task}: host issues a step in {t_host / 200 * 1e3:.3f} ms, the device takes {e0.elapsed_time(e1) / 200:.3f} ms per step")
pr = cProfile.Profile()
pr.enable()
for _ in range(100):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
