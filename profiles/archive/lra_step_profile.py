#!/usr/bin/env python3
"""Training step of an LRA task's PSFNet (synthetic tokens): wall ms/step vs GPU-busy ms/step and the top kernels.

    python profiles/lra_step_profile.py listops [batch]
"""
import os
import sys
import time

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsefactorization_amd import lra_training  # noqa: E402


def main():
    task = sys.argv[1] if len(sys.argv) > 1 else "listops"
    cfg = lra_training.config[task]
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else cfg["training"]["batch_size"]
    dev = torch.device("cuda:0")
    torch.manual_seed(42)
    net = lra_training.build_model(task).to(dev)
    if os.environ.get("PSF_PROFILE_FOREACH_ADAM"):
        opt = torch.optim.Adam(net.parameters(), lr=cfg["training"]["learning_rate"])
    else:
        from sparsefactorization_amd.train import make_adam
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

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        step()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 200 * 1e3
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for _ in range(3):
            step()
        torch.cuda.synchronize()
    ka = prof.key_averages()
    busy = sum(k.self_device_time_total for k in ka) / 3 / 1e3
    print(f"{task} batch {batch}: wall {wall:.3f} ms/step, GPU busy {busy:.3f} ms/step, tokens/step {X.numel()}")
    print(ka.table(sort_by="self_cuda_time_total", row_limit=22, max_name_column_width=64))
    if os.environ.get("PSF_PROFILE_CPU"):
        print(ka.table(sort_by="self_cpu_time_total", row_limit=30, max_name_column_width=64))


if __name__ == "__main__":
    main()
