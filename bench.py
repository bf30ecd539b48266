#!/usr/bin/env python3
"""Headline benchmark: PSF-Attn forward chain throughput (BASELINE.json metric) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch: the M = 14 dependent chord-spmm kernels
V <- W_m V + V_0 (SyntheticExperiments/psf.py:172-188) on the Adding configuration N = 16384, L = 15, C = 8,
B = 64 per GPU (BASELINE.json configs[1]) with all operands already resident in HBM (pre-materialised W_1..W_M
and V_0). With N GPUs every rank runs its own batch shard; the path has no data-path
collective ("weak" scaling: per-GPU work fixed). Rank 0 prints ONE JSON line.

roofline.achieved = algorithmic bytes per kernel launch / average launch duration, where
  bytes per launch = 4*B*N*(L + 2C + C) (read W row, read V row, read residual row, write out row; index = 0 B)
  average launch duration = HIP-event time over the timed region / (K*M) launches (events recorded on the
  stream the kernels run on, so inter-kernel gaps are included — conservative).
cpu_baseline = the reference's CPU op sequence (index_select -> mul -> scatter_add, + residual: what
torch_sparse.spmm executes on CPU tensors) restated in torch CPU ops (oracle/), timed on this host's cores on a
bounded sample of the same workload. It is reported, not targeted.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_SEQ, M_FACTORS, C_CH, B_PER_GPU = 16384, 14, 8, 64
L_LINKS = M_FACTORS + 1
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6290 GB/s is the measured copy rate


def make_inputs(device, B, seed_base=1234):
    """W_i = 0.1*randn(B,N,L) (seed 1234+i), V_0 = randn(B,N,C) (seed 1234) — SURVEY.md §8d."""
    g = torch.Generator(device=device)
    Ws = []
    for i in range(M_FACTORS):
        g.manual_seed(seed_base + 1 + i)
        Ws.append(0.1 * torch.randn(B, N_SEQ, L_LINKS, device=device, generator=g))
    g.manual_seed(seed_base)
    V0 = torch.randn(B, N_SEQ, C_CH, device=device, generator=g)
    return Ws, V0


def cpu_baseline(sample_B=16, chains=2):
    """Reference CPU path (torch-sparse op sequence) on a bounded sample: sample_B sequences, `chains` timed passes."""
    from oracle.chord_oracle import chord_indices, torch_chain_port
    import numpy as np

    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    rows, cols = chord_indices(N_SEQ, L_LINKS)
    index = torch.from_numpy(np.stack([rows, cols]))
    g = torch.Generator().manual_seed(1234)
    Ws = [0.1 * torch.randn(sample_B, N_SEQ, L_LINKS, generator=g) for _ in range(M_FACTORS)]
    V0 = torch.randn(sample_B, N_SEQ, C_CH, generator=g)
    with torch.no_grad():
        torch_chain_port(index, Ws[:2], V0, True)  # warm-up (allocator, thread pool)
        t0 = time.perf_counter()
        for _ in range(chains):
            torch_chain_port(index, Ws, V0, True)
        dt = (time.perf_counter() - t0) / chains
    return {
        "value": sample_B * N_SEQ / dt,
        "unit": "tokens/s",
        "cores": torch.get_num_threads(),
        "kind": "port",
        "sample": f"B={sample_B} of the same workload (N={N_SEQ}, M={M_FACTORS}, C={C_CH}, residual), "
                  f"{chains} timed chains, {dt * 1e3:.0f} ms/chain; torch CPU index_select->mul->index_add",
    }


def pmc_traffic():
    """HBM bytes per launch from the committed rocprofv3 PMC summary, if there is one (profiles/*_pmc.json)."""
    try:
        cands = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc.json"))
        if not cands:
            return None
        with open(os.path.join(ROOT, "profiles", cands[-1])) as fh:
            return json.load(fh).get("hbm_bytes_per_launch")
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        sys.exit(f"--gpus {args.gpus} needs torch.distributed.run with --nproc-per-node {args.gpus} (WORLD_SIZE={world})")

    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: torch.cuda.is_available() is False (there is no CPU path)")
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    dist = None
    if world > 1 or "RANK" in os.environ:  # launched by torch.distributed.run (also with a single rank)
        import torch.distributed as dist  # backend "nccl" is RCCL on ROCm
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    import sparsefactorization_amd as sfa

    Ws, V0 = make_inputs(device, B_PER_GPU, seed_base=1234 + 1000 * rank)
    kernel = sfa.describe_fwd(B_PER_GPU, N_SEQ, L_LINKS, C_CH)

    def step():
        return sfa.chord_chain(Ws, V0, True)

    def sync_all():
        torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(device)

    with torch.no_grad():
        # bring the clocks up before the W warm-up steps (a cold process measured ~10 % low): ~0.3 s of chains
        t_pre = time.perf_counter()
        while time.perf_counter() - t_pre < 0.3:
            for _ in range(8):
                step()
            torch.cuda.synchronize(device)
        for _ in range(args.warmup):
            step()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        sync_all()
        t0 = time.perf_counter()
        ev0.record()
        for _ in range(args.steps):
            out = step()
        ev1.record()
        torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(device)
        elapsed = time.perf_counter() - t0
    ev_ms = ev0.elapsed_time(ev1)
    assert torch.isfinite(out).all()

    if dist is not None:
        t = torch.tensor([elapsed, ev_ms], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, ev_ms = float(t[0]), float(t[1])

    if rank == 0:
        tokens = world * B_PER_GPU * N_SEQ * args.steps
        launches = args.steps * M_FACTORS
        bytes_per_launch = 4 * B_PER_GPU * N_SEQ * (L_LINKS + 2 * C_CH + C_CH)
        launch_s = ev_ms * 1e-3 / launches
        achieved = bytes_per_launch / launch_s / 1e9
        line = {
            "metric": "PSF-attn fwd tokens/sec @ N=16384, M=14, B=64",
            "value": tokens / elapsed,
            "unit": "tokens/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"adding_fwd_chain N={N_SEQ} M={M_FACTORS} L={L_LINKS} C={C_CH} B={B_PER_GPU}/gpu residual=on",
                "global_batch": world * B_PER_GPU,
                "seq_len": N_SEQ,
                "parallelism": f"batch-sharded x{world}, no data-path collective",
                "kernel": kernel,
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": pmc_traffic(),
                "bytes_per_launch": bytes_per_launch,
                "launch_us": launch_s * 1e6,
                "launches": launches,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)

    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
