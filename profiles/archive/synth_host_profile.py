#!/usr/bin/env python3
"""Host time of the eager training step of the synthetic tasks at a short length (the reference's length sweep is bound by it up to
N = 4096): ms per step host vs device, torch profiler's CPU table.   python profiles/synth_host_profile.py [problem] [N]"""
import os, sys, time
import torch
from torch.profiler import ProfilerActivity, profile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsefactorization_amd import psf_training  # noqa: E402
from sparsefactorization_amd.train import make_adam  # noqa: E402

problem = sys.argv[1] if len(sys.argv) > 1 else "order"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = torch.device("cuda:0")
torch.manual_seed(42)
net = psf_training.build_model(problem, N).to(dev)
opt = make_adam(net.parameters(), 1e-3)
loss = torch.nn.MSELoss() if problem == "adding" else torch.nn.CrossEntropyLoss()
X, Y = psf_training.make_split(problem, 40, N, dev, 1)


def step():
    opt.zero_grad(set_to_none=True)
    out = loss(net(X).squeeze(), Y)
    out.backward()
    opt.step()


for _ in range(30):
    step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter(); e0.record()
for _ in range(300):
    step()
t_host = time.perf_counter() - t0
e1.record(); torch.cuda.synchronize()
print(f"{problem} N={N}: host issues a step in {t_host / 300 * 1e3:.3f} ms, the device takes {e0.elapsed_time(e1) / 300:.3f} ms per step")
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(5):
        step()
    torch.cuda.synchronize()
ka = prof.key_averages()
print(f"GPU busy {sum(k.self_device_time_total for k in ka) / 5 / 1e3:.3f} ms per step")
print(ka.table(sort_by="self_cpu_time_total", row_limit=26, max_name_column_width=60))
