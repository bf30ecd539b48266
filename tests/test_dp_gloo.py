"""CPU, 2 processes over gloo: the batch-shard + flat gradient all-reduce logic of sparsefactorization_amd/dp.py.

Checks the property the 8-GPU run relies on: averaged shard gradients == gradients of the mean loss over the
concatenated (global) batch, including parameters that receive no gradient on a step.
"""
import os
import socket

import torch
import torch.multiprocessing as mp

from sparsefactorization_amd import dp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _Tiny(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.used = torch.nn.Linear(6, 5)
        self.head = torch.nn.Linear(5, 1)
        self.unused = torch.nn.Embedding(4, 3)  # like PSFNet.embedding for problem='adding': never touched

    def forward(self, x):
        return self.head(torch.nn.functional.gelu(self.used(x)))


def _global_batch():
    g = torch.Generator().manual_seed(3)
    return torch.randn(8, 6, generator=g), torch.randn(8, 1, generator=g)


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, w, dev = dp.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and dev.type == "cpu"
    torch.manual_seed(100 + rank)  # replicas start different on purpose ...
    net = _Tiny()
    dp.broadcast_parameters(net, src=0)  # ... and must end up identical to rank 0
    x, y = _global_batch()
    xs, ys = dp.shard_batch([x, y], rank, world)
    assert xs.shape[0] == 4
    reducer = dp.FlatGradAllReduce(net.parameters())
    opt = torch.optim.Adam(net.parameters(), lr=1e-2)
    for _ in range(3):
        opt.zero_grad(set_to_none=True)
        torch.nn.functional.mse_loss(net(xs), ys).backward()
        reducer()
        opt.step()
    assert net.unused.weight.grad is None
    torch.save({k: v.clone() for k, v in net.state_dict().items()}, os.path.join(out_dir, f"rank{rank}.pt"))
    torch.distributed.destroy_process_group()


def test_two_rank_data_parallel_equals_single_process(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    sd0 = torch.load(tmp_path / "rank0.pt")
    sd1 = torch.load(tmp_path / "rank1.pt")
    for k in sd0:
        assert torch.equal(sd0[k], sd1[k]), f"replicas diverged at {k}"

    # single process on the global batch, same initial weights as rank 0
    torch.manual_seed(100)
    net = _Tiny()
    x, y = _global_batch()
    opt = torch.optim.Adam(net.parameters(), lr=1e-2)
    for _ in range(3):
        opt.zero_grad(set_to_none=True)
        torch.nn.functional.mse_loss(net(x), y).backward()
        opt.step()
    for k, v in net.state_dict().items():
        assert torch.allclose(sd0[k], v, rtol=1e-5, atol=1e-6), k


def test_shard_bounds():
    assert dp.shard_bounds(64, 3, 8) == (24, 32)
    assert dp.shard_bounds(10, 1, 4) == (2, 4)                 # drop_last: 2 each, 2 dropped
    assert dp.shard_bounds(10, 3, 4, drop_last=False) == (9, 10)
    assert dp.shard_bounds(2, 3, 4, drop_last=False) == (2, 2)


def test_world_size_one_is_a_no_op():
    net = _Tiny()
    net(torch.randn(2, 6)).sum().backward()
    before = net.used.weight.grad.clone()
    dp.FlatGradAllReduce(net.parameters())()
    assert torch.equal(net.used.weight.grad, before)
