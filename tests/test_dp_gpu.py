"""GPU, 2 processes sharing the one GPU of the box over gloo (RCCL refuses two ranks on one device): the data-parallel
training path on the REAL model — broadcast of the replicas, batch shards, dp.FlatGradAllReduce over the HIP kernels'
gradients, parameters without a gradient — against single-process training on the concatenated batch.
(The 2-rank CPU test, tests/test_dp_gloo.py, can only use a toy model: the chord path has no CPU implementation.)"""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

CFG = dict(vocab_size=6, add_init_linear_layer=False, embedding_size=32, n_vec=512, n_W=9, Ws=[32, 'GELU'], V=[32, 'GELU'],
           n_channels_V=8, n_class=4, pooling_type="FLATTEN", head=['linear'], use_cuda=True, use_residuals=True,
           use_pos_embedding=True, problem="order")
STEPS, GLOBAL_B = 3, 16


def _batches(device):
    g = torch.Generator().manual_seed(5)
    return [(torch.randint(0, 6, (GLOBAL_B, 512, 1), generator=g).to(device), torch.randint(0, 4, (GLOBAL_B,), generator=g).to(device))
            for _ in range(STEPS)]


def _train(net, batches, reducer, rank, world):
    from sparsefactorization_amd import dp
    from sparsefactorization_amd.train import make_adam
    opt = make_adam(net.parameters(), 1e-3)
    loss = torch.nn.CrossEntropyLoss()
    for X, Y in batches:
        Xs, Ys = dp.shard_batch([X, Y], rank, world)
        opt.zero_grad(set_to_none=True)
        loss(net(Xs).squeeze(), Ys).backward()
        if reducer is not None:
            reducer()
        opt.step()


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from sparsefactorization_amd import dp
    from sparsefactorization_amd.synthetic_psf import PSFNet
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)  # replicas start different on purpose; the broadcast makes them rank 0's
    net = PSFNet(**CFG).to(dev)
    dp.broadcast_parameters(net, src=0)
    reducer = dp.FlatGradAllReduce(net.parameters())
    _train(net, _batches(dev), reducer, rank, world)
    torch.save({k: v.cpu() for k, v in net.state_dict().items()}, os.path.join(out_dir, f"rank{rank}.pt"))
    torch.save(reducer.calls, os.path.join(out_dir, f"calls{rank}.pt"))
    dist.destroy_process_group()


def test_two_rank_training_of_the_real_model_equals_single_process(gpu, tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    sd0, sd1 = torch.load(tmp_path / "rank0.pt"), torch.load(tmp_path / "rank1.pt")
    assert torch.load(tmp_path / "calls0.pt") == STEPS
    for k in sd0:
        assert torch.equal(sd0[k], sd1[k]), f"replicas diverged at {k}"
    from sparsefactorization_amd.synthetic_psf import PSFNet
    torch.manual_seed(100)
    net = PSFNet(**CFG).to(gpu)
    _train(net, _batches(gpu), None, 0, 1)
    for k, v in net.state_dict().items():
        assert torch.allclose(sd0[k], v.cpu(), rtol=2e-4, atol=2e-6), k


def test_bench_two_rank_rehearsal_replays_the_step_from_a_graph(gpu):
    """bench.py --gpus 2 on this one-GPU box in rehearsal mode (two ranks share the GPU over gloo: plumbing only, the
    numbers mean nothing): the self-launch, the barriers and gathers, and — the default with several ranks — the training
    leg replaying forward + backward from a HIP graph with the flat gradient all-reduce and Adam eager behind it."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PSF_BENCH_REHEARSAL="1")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--train-steps", "2", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{"metric"')][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and "rehearsal" in line
    # every rank names its device as the library's HIP runtime sees it; the rehearsal's two ranks share the one GPU
    assert len(line["devices"]) == 2 and all("pci=" in d and "xcds=8" in d and "cus=256" in d for d in line["devices"]), line["devices"]
    assert line["distinct_pci_devices"] == 1
    tr = line["train"]
    assert "error" not in tr, tr
    assert tr["hip_graph"] == "fwd+bwd" and tr["global_batch"] == 80 and tr["allreduce_us"] is not None
    assert tr["allreduce_bytes"] > 4_000_000 and tr["loss"] == tr["loss"]  # ~1.07 M fp32 gradients, finite loss


def test_bench_keeps_the_headline_when_a_rank_stalls_in_the_training_leg(gpu):
    """Several ranks: the training leg's collectives have never run on real multi-GPU hardware. If a rank never arrives, a
    deadline makes rank 0 print the finished headline with the leg marked as abandoned — and every rank exits NON-zero (4):
    a rank stuck in a kernel or a collective must not be recorded as a success by torch.distributed.run or the driver."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PSF_BENCH_REHEARSAL="1", PSF_BENCH_TEST_STALL_RANK="1", PSF_BENCH_TRAIN_DEADLINE_S="20")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--train-steps", "2", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode != 0, "a stalled rank must not end as rc 0"
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, out.stdout[-2000:] + out.stderr[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["roofline"]["frac"] > 0
    assert "did not finish" in line["train"]["error"] and line["train"]["rc"] == 4


def test_rccl_group_beside_a_gloo_default_group(gpu):
    """bench.py keeps its default process group on gloo (barriers and timing gathers: the headline must not depend on RCCL)
    and gives the training leg's reducer an RCCL group of its own. The same combination with one rank: a forced flat
    all-reduce through the RCCL group leaves the gradients as they were."""
    import subprocess
    import sys
    import textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import os, sys, torch, torch.distributed as dist
        sys.path.insert(0, %r)
        from sparsefactorization_amd import dp
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", RANK="0", WORLD_SIZE="1")
        dist.init_process_group("gloo", rank=0, world_size=1)
        g = dist.new_group(backend="nccl")
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        lin = torch.nn.Linear(64, 32).to(dev)
        lin(torch.randn(8, 64, device=dev)).square().sum().backward()
        before = [p.grad.clone() for p in lin.parameters()]
        red = dp.FlatGradAllReduce(lin.parameters(), group=g, timing=True, force=True)
        red()
        torch.cuda.synchronize()
        assert dist.get_backend(g) == "nccl" and dist.get_backend() == "gloo"
        assert all(torch.equal(a, p.grad) for a, p in zip(before, lin.parameters()))
        assert red.mean_us() > 0
        dist.barrier()
        dist.destroy_process_group()
        print("ok")
    """ % root)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-3000:]
