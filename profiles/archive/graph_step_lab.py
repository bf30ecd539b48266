#!/usr/bin/env python3
"""Does a whole training step (forward, loss, backward, Adam) of an LRA PSFNet capture into a HIP graph, and what
does replaying it buy on the launch-bound small models?

    python profiles/graph_step_lab.py pathfinder

RESULTS (round 1):
* first attempt (profiles/r01_graph_step_lab.log): capture succeeded and the first replays reproduced the eager
  losses exactly, but a later replay faulted (memory aperture violation) inside PyTorch's own
  aten::embedding_dense_backward (rocprim partition kernel whose sizes were read back to the host at capture time).
* after the table gradient moved to psf_embed_tokens_bwd_f32 (no sort, no host read-back;
  profiles/r01_graph_step_lab2.log): 48 replays with changing batches, no fault, loss after 8 steps 0.060913 vs
  0.060912 eager; Pathfinder 2.11 ms/step eager (foreach Adam) -> 1.37 ms/step replayed.
The guard below stays: the script replays a captured graph of a WHOLE step on whatever the installed PyTorch does for
every op in it, and one non-capturable op is a GPU fault, not an exception. train.GraphedStep is the supported form.
"""
import copy
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsefactorization_amd import lra_training  # noqa: E402


def main():
    if os.environ.get("PSF_GRAPH_LAB_UNSAFE") != "1":
        raise SystemExit("graph_step_lab: a replay faulted the GPU in round 1 (see the docstring); "
                         "set PSF_GRAPH_LAB_UNSAFE=1 only after nn.Embedding's backward has been replaced")
    task = sys.argv[1] if len(sys.argv) > 1 else "pathfinder"
    cfg = lra_training.config[task]
    batch = cfg["training"]["batch_size"]
    dev = torch.device("cuda:0")
    torch.manual_seed(42)
    net = lra_training.build_model(task).to(dev)
    net_g = copy.deepcopy(net)
    loss = torch.nn.CrossEntropyLoss()
    X, Y = lra_training.synthetic_split(task, batch * 4, dev, 1)
    if cfg["model"]["pooling_type"] == "CLS":
        X = lra_training.add_cls_token(X, cfg["model"]["vocab_size"])
    batches = [(X[i * batch:(i + 1) * batch], Y[i * batch:(i + 1) * batch]) for i in range(4)]

    # eager
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)

    def eager_step(x, y):
        opt.zero_grad(set_to_none=True)
        out = loss(net(x).squeeze(), y)
        out.backward()
        opt.step()
        return out

    for i in range(8):
        l_e = eager_step(*batches[i % 4])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(40):
        eager_step(*batches[i % 4])
    torch.cuda.synchronize()
    t_eager = (time.perf_counter() - t0) / 40 * 1e3

    # graphed: static inputs, capturable Adam, 3 eager warm-up steps on a side stream, then capture one step
    opt_g = torch.optim.Adam(net_g.parameters(), lr=1e-3, capturable=True)
    sx, sy = batches[0][0].clone(), batches[0][1].clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for i in range(3):
            sx.copy_(batches[i % 4][0])
            sy.copy_(batches[i % 4][1])
            opt_g.zero_grad(set_to_none=True)
            out = loss(net_g(sx).squeeze(), sy)
            out.backward()
            opt_g.step()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    opt_g.zero_grad(set_to_none=True)
    with torch.cuda.graph(g):
        s_out = loss(net_g(sx).squeeze(), sy)
        s_out.backward()
        opt_g.step()

    def graph_step(x, y):
        sx.copy_(x)
        sy.copy_(y)
        g.replay()
        return s_out

    for i in range(3, 8):
        l_g = graph_step(*batches[i % 4])
    torch.cuda.synchronize()
    print(f"loss after 8 steps: eager {float(l_e):.6f}  graphed {float(l_g):.6f}")
    t0 = time.perf_counter()
    for i in range(40):
        graph_step(*batches[i % 4])
    torch.cuda.synchronize()
    t_graph = (time.perf_counter() - t0) / 40 * 1e3
    print(f"{task}: eager {t_eager:.3f} ms/step, graph replay {t_graph:.3f} ms/step")


if __name__ == "__main__":
    main()
