#!/usr/bin/env python3
"""Training step of BASELINE.json configs[2] as worded there (ListOps N = 2048, dim = 64: E = C = 64, batch 32): wall and
GPU-busy ms/step and the top kernels.    python profiles/lra_step_profile_cfg3.py"""
import os
import sys
import time

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsefactorization_amd import lra_training  # noqa: E402
from sparsefactorization_amd.train import make_adam  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(42)
    over = dict(n_vec=2048, embedding_size=64, n_channels_V=64)
    net = lra_training.build_model("listops", **over).to(dev)
    opt = make_adam(net.parameters(), 1e-3)
    loss = torch.nn.CrossEntropyLoss()
    g = torch.Generator(device=dev).manual_seed(1)
    X = torch.randint(0, 15, (32, 2047), device=dev, generator=g)
    Y = torch.randint(0, 10, (32,), device=dev, generator=g)
    X = lra_training.add_cls_token(X, 17)

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
    print(f"listops N=2048 E=C=64 batch 32: wall {wall:.3f} ms/step, GPU busy {busy:.3f} ms/step, tokens/step {X.numel()}")
    print(ka.table(sort_by="self_cuda_time_total", row_limit=24, max_name_column_width=64))


if __name__ == "__main__":
    main()
