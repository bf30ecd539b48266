#!/usr/bin/env python3
"""Headline benchmark: PSF-Attn forward chain throughput (BASELINE.json metric) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--no-train] [--train-graph] [--no-cpu-baseline]

``--gpus N`` with N > 1 started as a plain ``python bench.py`` launches N ranks by itself: the parent process
(which never touches the GPU) starts ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
127.0.0.1 --master-port P bench.py --gpus N ...`` as a CHILD, relays its output and exits with its return code.
Started under ``torch.distributed.run`` already (RANK in the environment) it is one rank of that job.

One "step" = one pass of the hot path over one batch: the M = 14 dependent chord-spmm kernels
V <- W_m V + V_0 (SyntheticExperiments/psf.py:172-188) on the Adding configuration N = 16384, L = 15, C = 8,
B = 64 per GPU (BASELINE.json configs[1]) with all operands already resident in HBM (pre-materialised W_1..W_M
and V_0). With N GPUs every rank runs its own batch shard; the path has no data-path
collective ("weak" scaling: per-GPU work fixed). Rank 0 prints ONE JSON line.

roofline.achieved = algorithmic bytes per kernel launch / average launch duration, where
  bytes per launch = 4*B*N*(L + 2C + C) (read W row, read V row, read residual row, write out row; index = 0 B)
  average launch duration = HIP-event time over the timed region / (K*M) launches (events recorded on the
  stream the kernels run on, so inter-kernel gaps are included — conservative). With N ranks: the slowest rank's.
roofline.traffic = memory-side bytes per launch from the committed rocprofv3 PMC summary named in
  roofline.traffic_source — null when the kernel sources changed since that collection.
cpu_baseline = the reference's CPU op sequence (index_select -> mul -> scatter_add, + residual: what
  torch_sparse.spmm executes on CPU tensors) restated in torch CPU ops (oracle/), timed on this host's cores on a
  bounded sample of the same workload, best of a thread-count sweep; cpu_baseline_cfg1 = the same at BASELINE.json
  configs[0] (Adding N = 128, M = 7, B = 40 — the reference's own CPU-runnable case). Reported, not targeted.
roofline.peak_measured / frac_of_measured = the same achieved rate against what the library's plain streaming kernel with
  the step's byte mix (psf_stream_mix_f32) sustains on this box with the operands where the chain has them, measured in this
  run outside the timed region; peak_measured_hbm_only = the same kernel with every operand beyond the Infinity Cache.
e2e_forward = SURVEY.md §8(d)'s end-to-end variant (one GPU): the real PSFNet (seed 42) on generated Adding / Temporal-Order
  batches at N = 16384, B = 64, both routes of the mixer (W through memory; W computed inside each chain step).
shapes = the other timed shapes of SURVEY.md §8(d) (cfg1, cfg3 reference and BASELINE wording, cfg4 chain and attention map,
  genome-like): us per step and algorithmic-bytes fraction (one GPU), twice: `*_cache_resident` (one operand set re-used: it
  sits in the 256 MB Infinity Cache — NOT an HBM rate) and `*_rotating` (operand sets spanning 2.5 x the Infinity Cache: the
  fraction of the HBM roofline). One entry PER KERNEL, each with the time of that kernel and the counters of that kernel:
  `fwd_step_kernel` (the per-step chord_fwd_win_k: what training runs, every step kept), `fwd_chain_kernel` (what a no-grad
  chord_chain call runs: chord_chain_lds_k / chord_chain_rows_k, one launch for all steps, where it applies — else "same_as"), `bwd_step_kernel`
  (the fused backward step), `bwd_chain_kernel` (the whole backward chain in one launch where the library has it: cfg1). `counters`: memory-side bytes and L2 requests per launch of THAT kernel from the committed
  per-shape counter summary, only when it was collected on these kernel sources. `bound`: "hbm", or "l2" when the counters say
  traffic / algorithmic <= 1.1 and the kernel is under 0.6 of HBM — then `l2_to_cu_GBps` (L2 requests x 128 B / rocprof time)
  and `frac_of_l2_roof` (of the guide's 16.8-18.8 TB/s for rows served out of the XCDs' L2) say where it sits.
devices = every rank's HIP device as the library sees it (PCI bus id, XCDs, CUs); distinct_pci_devices must equal n_gpus in a
  real multi-GPU run.
train = the data-parallel leg (BASELINE.json configs[4]): Temporal Order N = 16384, B = 40 per
  GPU, full training step of the reference loop (psf_utils.py:62-71) with ONE flat RCCL gradient all-reduce;
  tokens/s over all ranks and the mean device time of the all-reduce. Not part of the headline's timed region.
  With several ranks the step is replayed from a HIP graph by default (--no-train-graph: eager).
train_listops = BASELINE.json configs[2] (one GPU only): the ListOps training step at the reference's configuration
  and at BASELINE's wording (N = 2048, dim = 64).
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_SEQ, M_FACTORS, C_CH, B_PER_GPU = 16384, 14, 8, 64
L_LINKS = M_FACTORS + 1
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); the rate this box sustains is measured below
# rows served out of an XCD's L2 to its CUs, chip-wide (MI355X_MICROARCH.md, "L2 (per XCD)": 2 048 rows shared by every
# workgroup: 16.8-18.8 TB/s): the roof of a chord kernel whose memory-side traffic is its algorithmic bytes and which still
# sits under 0.6 of HBM — every link of a row is an L2 -> CU request
L2_ROOF_GBS = (16800.0, 18800.0)
L2_REQUEST_BYTES = 128
# several ranks: the data-parallel training leg is abandoned after this long (see main); the variable exists for the test of it
TRAIN_DEADLINE_S = float(os.environ.get("PSF_BENCH_TRAIN_DEADLINE_S", "300"))
TRAIN_ABANDONED_RC = 4  # exit code of every rank when that deadline passes (the headline line is still printed by rank 0)
PREHEAT_S = 0.3        # seconds of untimed chains before the warm-up steps (brings the clocks up; printed in the line)
TRAIN_PROBLEM, TRAIN_N, TRAIN_B = "order", 16384, 40  # SyntheticExperiments/synthetic_training_config.py:72-86


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--train", dest="train", action="store_true", default=True,
                    help="also time the data-parallel training leg (default)")
    ap.add_argument("--no-train", dest="train", action="store_false")
    ap.add_argument("--train-steps", type=int, default=20)
    ap.add_argument("--train-graph", dest="train_graph", action="store_true", default=None,
                    help="training leg: replay the step from a HIP graph (with several ranks: forward+backward; all-reduce + "
                         "Adam stay eager). Default: on when --gpus > 1 — ~150 launches per step from each of N host "
                         "processes is where the scaling of a 2.4 ms step is decided — off on one GPU")
    ap.add_argument("--no-train-graph", dest="train_graph", action="store_false")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the e2e_forward and shapes legs (one GPU only)")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------
# parent side: start N ranks as a child job (no GPU call may happen in this process before or after)
# ---------------------------------------------------------------------------------------------------------
def rehearsal() -> bool:
    """PSF_BENCH_REHEARSAL=1: run the N-rank code path (self-launch, barriers, gathers, the gradient reducer) on a box
    with fewer GPUs than ranks — ranks share the GPUs round-robin and talk over gloo, because RCCL refuses two ranks on
    one device. For checking the multi-rank plumbing only: the line is marked and its numbers mean nothing."""
    return os.environ.get("PSF_BENCH_REHEARSAL", "0") == "1"


def self_launch(args) -> int:
    import socket

    import torch  # importing torch does not initialise the GPU; device_count() does not either on this image
    have = torch.cuda.device_count()
    if have < args.gpus and not (rehearsal() and have >= 1):
        print(f"bench.py: --gpus {args.gpus} but this host exposes {have} GPU(s); "
              "one rank per GPU is required (no oversubscription)", file=sys.stderr, flush=True)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    # The GPU pool's own requirement for multi-process GPU work (its host driver only supports dmabuf IPC; the pool
    # exports this variable on every box). Kept as a default so that a child started from a stripped environment still
    # has it. No run of ours has shown it matter: no box with more than one GPU was available.
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=None, text=True, env=env)
    line = None
    for out in proc.stdout:  # relay; remember the JSON line of rank 0
        sys.stdout.write(out)
        sys.stdout.flush()
        if out.lstrip().startswith('{"metric"'):
            line = out
    rc = proc.wait()
    if rc == 0 and line is None:
        print("bench.py: the ranks exited cleanly but printed no result line", file=sys.stderr, flush=True)
        return 3
    return rc


# ---------------------------------------------------------------------------------------------------------
# rank side
# ---------------------------------------------------------------------------------------------------------
def make_inputs(device, B, seed_base=1234):
    """W_i = 0.1*randn(B,N,L) (seed 1234+i), V_0 = randn(B,N,C) (seed 1234) — SURVEY.md §8d."""
    import torch
    g = torch.Generator(device=device)
    Ws = []
    for i in range(M_FACTORS):
        g.manual_seed(seed_base + 1 + i)
        Ws.append(0.1 * torch.randn(B, N_SEQ, L_LINKS, device=device, generator=g))
    g.manual_seed(seed_base)
    V0 = torch.randn(B, N_SEQ, C_CH, device=device, generator=g)
    return Ws, V0


def _cpu_chain_time(N, M, C, B, threads, chains, residual=True):
    import numpy as np
    import torch
    from oracle.chord_oracle import chord_indices, torch_chain_port
    L = M + 1
    torch.set_num_threads(threads)
    rows, cols = chord_indices(N, L)
    index = torch.from_numpy(np.stack([rows, cols]))
    g = torch.Generator().manual_seed(1234)
    Ws = [0.1 * torch.randn(B, N, L, generator=g) for _ in range(M)]
    V0 = torch.randn(B, N, C, generator=g)
    with torch.no_grad():
        torch_chain_port(index, Ws[:2], V0, residual)  # warm-up (allocator, thread pool)
        t0 = time.perf_counter()
        for _ in range(chains):
            torch_chain_port(index, Ws, V0, residual)
        return (time.perf_counter() - t0) / chains


def cpu_baseline():
    """Reference CPU path (torch-sparse op sequence) on a bounded sample of the headline workload. The thread count is
    swept ({8, 32, 64, all host cores}, B = 8, one chain each): ``index_add_`` stops scaling early and the all-cores
    setting is NOT the fastest on a 256-core host; the best setting is then timed on B = 16, two chains."""
    host = os.cpu_count() or 1
    sweep = {}
    for t in sorted({min(8, host), min(32, host), min(64, host), host}):
        sweep[t] = 8 * N_SEQ / _cpu_chain_time(N_SEQ, M_FACTORS, C_CH, 8, t, 1)
    best = max(sweep, key=sweep.get)
    sample_B, chains = 16, 2
    dt = _cpu_chain_time(N_SEQ, M_FACTORS, C_CH, sample_B, best, chains)
    return {
        "value": sample_B * N_SEQ / dt,
        "unit": "tokens/s",
        "cores": best,
        "host_cores": host,
        "kind": "port",
        "sample": f"B={sample_B} of the same workload (N={N_SEQ}, M={M_FACTORS}, C={C_CH}, residual), "
                  f"{chains} timed chains, {dt * 1e3:.0f} ms/chain; torch CPU index_select->mul->index_add",
        "thread_sweep_tokens_per_s": {str(k): round(v) for k, v in sweep.items()},
    }


def cpu_baseline_cfg1():
    """BASELINE.json configs[0]: Adding N = 128, M = 7 (L = 8), C = 8, B = 40, residual — defined as the CPU path."""
    host = os.cpu_count() or 1
    N, M, C, B, chains = 128, 7, 8, 40, 200
    sweep = {t: B * N / _cpu_chain_time(N, M, C, B, t, chains) for t in sorted({1, min(8, host), min(32, host)})}
    best = max(sweep, key=sweep.get)
    return {"value": sweep[best], "unit": "tokens/s", "cores": best, "host_cores": host, "kind": "port",
            "sample": f"full cfg1 workload (N={N}, M={M}, C={C}, B={B}, residual), {chains} timed chains per setting",
            "thread_sweep_tokens_per_s": {str(k): round(v) for k, v in sweep.items()}}


def gpu_cfg1(device):
    """The same cfg1 chain through the HIP path (one LDS-resident launch per chain), for the line next to the CPU's."""
    import torch
    import sparsefactorization_amd as sfa
    N, M, C, B = 128, 7, 8, 40
    g = torch.Generator(device=device).manual_seed(1234)
    Ws = [0.1 * torch.randn(B, N, M + 1, device=device, generator=g) for _ in range(M)]
    V0 = torch.randn(B, N, C, device=device, generator=g)
    with torch.no_grad():
        for _ in range(20):
            sfa.chord_chain(Ws, V0, True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(device)
        e0.record()
        for _ in range(200):
            sfa.chord_chain(Ws, V0, True)
        e1.record()
        torch.cuda.synchronize(device)
    return B * N * 200 / (e0.elapsed_time(e1) * 1e-3)


def measured_bandwidth(device):
    """What this GPU's memory system gives the forward step's byte mix, measured with the library's plain streaming kernel
    (psf_stream_mix_f32, csrc/stream_mix.hip: reads W : V : residual = 2 : 1 : 1 and writes 1, the step's 15 : 8 : 8 : 8),
    outside the headline's timed region, HIP events over the launches:

    * ``stream_hbm``: every operand beyond the 256 MiB Infinity Cache (2.5 GB footprint: W 1 GiB, the other three 512 MiB
      each) — what HBM alone sustains for the mix;
    * ``stream_chain_residency``: the operands placed as the chain places them — 14 steps, each with its own 63 MB W
      slab (881 MB, streamed from HBM once per chain) and the three V-sized streams (33.5 MB each: input = the previous
      step's output, residual, output) re-used from step to step, i.e. resident in the Infinity Cache. This is the
      denominator of ``frac_of_measured``: the same bytes from the same places, with no gather and no LDS.
    GB/s of bytes read + written."""
    import torch
    from sparsefactorization_amd import _lib
    lib = _lib.load()
    stream = torch.cuda.current_stream(device).cuda_stream

    def launch(w, v, r, o, n):
        rc = lib.psf_stream_mix_f32(w.data_ptr(), v.data_ptr(), r.data_ptr(), o.data_ptr(), n, stream)
        _lib.check(rc, "psf_stream_mix_f32")

    def timed(fn, nbytes, reps):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(device)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize(device)
        return nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9

    out = {}
    with torch.cuda.device(device):
        n = 32 * 1024 * 1024  # vectors of 16 bytes: V-sized streams of 512 MiB, W of 1 GiB
        w = torch.empty(8 * n, dtype=torch.float32, device=device).fill_(0.5)
        v, r, o = (torch.empty(4 * n, dtype=torch.float32, device=device).fill_(1.0) for _ in range(3))
        out["stream_hbm"] = timed(lambda: launch(w, v, r, o, n), 80 * n, 10)
        del w, v, r, o
        torch.cuda.empty_cache()
        nv = B_PER_GPU * N_SEQ * C_CH // 4  # vectors per V-sized stream at the headline shape (33.5 MB)
        ws = [torch.empty(8 * nv, dtype=torch.float32, device=device).fill_(0.5) for _ in range(M_FACTORS)]
        v0, pa, pb = (torch.empty(4 * nv, dtype=torch.float32, device=device).fill_(1.0) for _ in range(3))

        def chain():
            for m in range(M_FACTORS):
                launch(ws[m], v0 if m == 0 else (pa, pb)[(m - 1) & 1], v0, (pa, pb)[m & 1], nv)

        out["stream_chain_residency"] = timed(chain, 80 * nv * M_FACTORS, 20)
        del ws, v0, pa, pb
        torch.cuda.empty_cache()
    return out


def pmc_traffic():
    """(bytes per launch, source) from the newest committed forward-kernel PMC summary (profiles/*_pmc.json with
    ``hbm_bytes_per_launch``). The summary records the hash of the kernel sources it was collected on
    (``csrc_hash``, sparsefactorization_amd/build.py); if that differs from the sources of THIS tree the number
    describes another kernel and is withheld (None), with the reason in the source string."""
    try:
        from sparsefactorization_amd.build import csrc_hash
        now = csrc_hash()
        best = None
        pdir = os.path.join(ROOT, "profiles")
        for f in sorted(os.listdir(pdir)):
            if not f.endswith("_pmc.json"):
                continue
            with open(os.path.join(pdir, f)) as fh:
                d = json.load(fh)
            if "hbm_bytes_per_launch" in d:
                best = (f, d)
        if best is None:
            return None, "no profiles/*_pmc.json with hbm_bytes_per_launch"
        f, d = best
        then = d.get("csrc_hash")
        if then != now:
            return None, f"profiles/{f} @ csrc {str(then)[:12]} is stale: csrc is now {now[:12]}"
        return d["hbm_bytes_per_launch"], f"profiles/{f} @ csrc {now[:12]}"
    except Exception as exc:  # never let bookkeeping kill the measurement
        return None, f"unavailable: {exc!r}"


def run_rank(args) -> int:
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr, flush=True)
        return 2
    n_dev = torch.cuda.device_count()
    if n_dev <= local_rank and not (rehearsal() and n_dev >= 1):
        print(f"bench.py: rank {rank} needs GPU {local_rank} but this host exposes {n_dev} GPU(s)",
              file=sys.stderr, flush=True)
        return 2
    if not torch.cuda.is_available():
        print("bench.py needs an MI355X: torch.cuda.is_available() is False (there is no CPU path)", file=sys.stderr)
        return 2
    device = torch.device("cuda", local_rank % n_dev if rehearsal() else local_rank)
    torch.cuda.set_device(device)

    dist = None
    if "RANK" in os.environ:  # launched by torch.distributed.run (also with a single rank)
        import torch.distributed as dist  # backend "nccl" is RCCL on ROCm
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # The default group is gloo: the forward path has NO data-path collective, its ranks only meet at barriers and at
        # two small gathers of timings (CPU tensors), so the headline of an N-GPU run does not depend on RCCL coming up.
        # RCCL carries what needs it — the gradient all-reduce of the training leg — in its own group (train_leg).
        dist.init_process_group("gloo", rank=rank, world_size=world)

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
        while time.perf_counter() - t_pre < PREHEAT_S:
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
        sync_all()
        elapsed = time.perf_counter() - t0
    ev_ms = ev0.elapsed_time(ev1)
    assert torch.isfinite(out).all()

    # which device each rank ran on, as the library's HIP runtime names it: a multi-GPU record must show `world` distinct PCI
    # bus ids (and 8 XCDs / 256 CUs each) — checkable from the line alone
    with torch.cuda.device(device):
        try:
            from sparsefactorization_amd import _lib as _psf_lib
            my_dev = f"rank={rank} local_rank={local_rank} {_psf_lib.device_info()}"
        except Exception as exc:
            my_dev = f"rank={rank} local_rank={local_rank} unavailable: {exc!r}"
    devices = [my_dev]
    if dist is not None:
        devices = [None] * world
        dist.all_gather_object(devices, my_dev)
    per_rank_ms = [ev_ms]
    if dist is not None:
        t = torch.tensor([elapsed, ev_ms], device="cpu", dtype=torch.float64)  # the default group is gloo
        gathered = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(gathered, t)
        elapsed = max(float(g[0]) for g in gathered)  # MAX over ranks
        per_rank_ms = [float(g[1]) for g in gathered]
        ev_ms = max(per_rank_ms)
    del Ws, V0, out
    torch.cuda.empty_cache()

    bw = None
    if rank == 0:
        try:
            bw = measured_bandwidth(device)
        except Exception as exc:  # bookkeeping must not lose the headline
            bw = {"error": repr(exc)}
    if args.train_graph is None:
        args.train_graph = world > 1
    line = None
    if rank == 0:
        tokens = world * B_PER_GPU * N_SEQ * args.steps
        launches = args.steps * M_FACTORS
        bytes_per_launch = 4 * B_PER_GPU * N_SEQ * (L_LINKS + 2 * C_CH + C_CH)
        launch_s = ev_ms * 1e-3 / launches
        achieved = bytes_per_launch / launch_s / 1e9
        traffic, traffic_source = pmc_traffic()
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
                "traffic": traffic,
                "traffic_source": traffic_source,
                "bytes_per_launch": bytes_per_launch,
                "launch_us": launch_s * 1e6,
                "launches": launches,
                "frac_per_gpu": [bytes_per_launch / (ms * 1e-3 / launches) / 1e9 / HBM_PEAK_GBS for ms in per_rank_ms],
                # the second denominator: what a plain streaming kernel with the same byte mix
                # sustains on THIS box with the operands where the chain has them (measured_bandwidth)
                "peak_measured": bw.get("stream_chain_residency") if bw else None,
                "peak_measured_hbm_only": bw.get("stream_hbm") if bw else None,
                "peak_measured_how": "psf_stream_mix_f32 (in-tree, W : V : res : out = 2 : 1 : 1 : 1, no gather): "
                                     "peak_measured = 14 steps with a 63 MB W slab each from HBM and the three 33.5 MB V-sized "
                                     "streams re-used step to step (Infinity-Cache resident, as in the chain); "
                                     "peak_measured_hbm_only = every operand beyond the Infinity Cache (2.5 GB footprint); "
                                     "bytes read + written per second, HIP events, rank 0, outside the timed region",
                "frac_of_measured": (achieved / bw["stream_chain_residency"]) if bw and bw.get("stream_chain_residency") else None,
                "frac_of_measured_hbm_only": (achieved / bw["stream_hbm"]) if bw and bw.get("stream_hbm") else None,
                # of the step's algorithmic bytes, the part that is re-used within 235 MB of traffic and therefore served by
                # the 256 MiB Infinity Cache rather than HBM: input row + residual row of the 4 (L + 3C) bytes per row
                "infinity_cache_resident_read_fraction": 2 * C_CH / (L_LINKS + 2 * C_CH),
            },
            "preheat_s": PREHEAT_S,
            "rank_ms_per_step": {"min": min(per_rank_ms) / args.steps, "max": max(per_rank_ms) / args.steps},
            "devices": devices,
            "distinct_pci_devices": len({d.split("pci=")[1].split()[0] for d in devices if "pci=" in d}),
        }
        if rehearsal():
            line["rehearsal"] = "ranks share GPUs over gloo: plumbing check only, the numbers are meaningless"

    # The secondary legs run AFTER the headline is complete. With several ranks the training leg contains collectives
    # that no box with more than one GPU has ever run: a rank that fails inside one leaves the others
    # waiting. A deadline guards the headline: when it passes, rank 0 prints the finished line with the leg marked as
    # abandoned — and every rank then leaves with TRAIN_ABANDONED_RC, NOT 0: a rank that is stuck in a kernel or a collective
    # must not be recorded as a success (torch.distributed.run, self_launch and the driver all see the non-zero code; the
    # line carries it too, as train.rc). If a real multi-GPU run ever trips this, read its records; do not re-run to see it again.
    train = None
    train_listops = None
    if args.train:
        guard = None
        if world > 1:
            import threading

            def abandon():
                if rank == 0 and line is not None:
                    line["train"] = {"error": f"the training leg did not finish within {TRAIN_DEADLINE_S} s on {world} ranks; "
                                              "abandoned so that the headline line is kept",
                                     "rc": TRAIN_ABANDONED_RC}
                    print(json.dumps(line), flush=True)
                if rank != 0:
                    time.sleep(5)  # rank 0 prints first: a peer that leaves earlier closes its sockets under rank 0's collective,
                                   # which then fails with "connection closed" before rank 0's own deadline has fired
                os._exit(TRAIN_ABANDONED_RC)

            guard = threading.Timer(TRAIN_DEADLINE_S, abandon)
            guard.daemon = True
            guard.start()
        train = train_leg(args, device, rank, world, dist)
        if guard is not None:
            guard.cancel()
        if world == 1:
            train_listops = listops_leg(args, device)
    e2e, shapes = None, None
    if world == 1 and rank == 0 and not args.no_extra_legs:
        e2e = e2e_forward_leg(device)
        shapes = shapes_leg(device)

    if rank == 0:
        if train is not None:
            line["train"] = train
        if train_listops is not None:
            line["train_listops"] = train_listops
        if e2e is not None:
            line["e2e_forward"] = e2e
        if shapes is not None:
            line["shapes"] = shapes
        if world == 1 and not args.no_cpu_baseline:
            # the GPU side of cfg1 first: its 200 chains are launch-bound, and the CPU legs leave up to 256 OpenMP threads
            # spinning beside the launching thread (one run read 18.7 M tokens/s behind them where it reads 220-250 M alone)
            try:
                gpu1, gpu1_err = gpu_cfg1(device), None
            except Exception as exc:
                gpu1, gpu1_err = None, repr(exc)
            line["cpu_baseline"] = cpu_baseline()
            cfg1 = cpu_baseline_cfg1()
            cfg1["gpu_value"] = gpu1
            if gpu1_err is not None:
                cfg1["gpu_error"] = gpu1_err
            line["cpu_baseline_cfg1"] = cfg1
        print(json.dumps(line), flush=True)

    if dist is not None:
        dist.destroy_process_group()
    return 0


def train_leg(args, device, rank, world, dist):
    """Temporal Order N = 16384, B = 40 per GPU: whole-job training tokens/s and the gradient all-reduce's device time."""
    import torch
    r, err = None, None
    if os.environ.get("PSF_BENCH_TEST_STALL_RANK") == str(rank):  # test hook: this rank never reaches the leg's collectives
        time.sleep(3600)
    try:
        from sparsefactorization_amd.psf_training import train_benchmark
        # gradients travel over RCCL ("nccl" on ROCm), in a group of their own; the rehearsal (ranks sharing a GPU) stays on gloo
        group = dist.new_group(backend="nccl") if (dist is not None and world > 1 and not rehearsal()) else None
        r = train_benchmark(TRAIN_PROBLEM, TRAIN_N, TRAIN_B, steps=args.train_steps, warmup=5, device=device,
                            rank=rank, world=world, graph=args.train_graph, group=group)
        vals = [r["seconds"], r["event_ms"], r["allreduce_us"] if r["allreduce_us"] is not None else -1.0, 1.0]
    except Exception as exc:  # this rank still takes part in the gather below: the others must not wait for it
        err = repr(exc)
        vals = [0.0, 0.0, -1.0, 0.0]
    try:
        if dist is not None:
            t = torch.tensor(vals, device="cpu", dtype=torch.float64)
            gathered = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(gathered, t)
            failed = [i for i, g in enumerate(gathered) if float(g[3]) == 0.0]
            if failed:
                return {"error": err if err is not None else f"the training leg failed on rank(s) {failed}"}
            secs = max(float(g[0]) for g in gathered)
            ms = [float(g[1]) / r["steps"] for g in gathered]
            ar = max(float(g[2]) for g in gathered)
        else:
            if err is not None:
                return {"error": err}
            secs, ms, ar = vals[0], [vals[1] / r["steps"]], vals[2]
        return {
            "metric": "PSF train tokens/sec, Temporal Order N=16384, B=40/GPU, Adam",
            "value": world * TRAIN_B * TRAIN_N * r["steps"] / secs,
            "unit": "tokens/s",
            "steps": r["steps"],
            "ms_per_step": secs * 1e3 / r["steps"],
            "rank_ms_per_step": {"min": min(ms), "max": max(ms)},
            "global_batch": world * TRAIN_B,
            "allreduce_us": None if ar < 0 else ar,
            "allreduce_bytes": r["grad_bytes"],
            "allreduce": "one flat fp32 bucket, RCCL all_reduce(AVG); device time of bucket fill + collective + copy back"
                         if world > 1 else "not run (one rank)",
            "hip_graph": r["hip_graph"],
            "loss": r["loss"],
            "scaling": "weak",
        }
    except Exception as exc:  # the headline line must survive a failure of the secondary leg
        return {"error": repr(exc)}


def listops_leg(args, device):
    """BASELINE.json configs[2], "LRA ListOps — full PSF model training loop on 1 x MI355X", twice: the reference's own
    configuration (LRA/psf_training_config.py:2-30: N = 1 + 1999, E = 512, 128 channels, batch 32) and BASELINE's wording
    (N = 2048, dim = 64). One GPU; outside the headline's timed region. The whole step (zero_grad, forward, loss, backward, Adam)
    is replayed from a HIP graph (train.GraphedStep: the same kernels in the same order; a step of the small configuration is
    ~200 launches for ~1.3 ms of GPU work, and issued one by one it takes 1.4-1.9 ms depending on the host); the eagerly
    stepped time is reported beside it."""
    out = {}
    try:
        from sparsefactorization_amd.lra_training import train_benchmark
        for tag, over in (("reference_config", {}), ("baseline_wording", dict(n_vec=2048, embedding_size=64, n_channels_V=64))):
            eager = train_benchmark("listops", steps=args.train_steps, warmup=5, device=device, graph=False, **over)
            r = train_benchmark("listops", steps=args.train_steps, warmup=5, device=device, graph=True, **over)
            out[tag] = {"metric": f"PSF train tokens/sec, ListOps N={r['n_vec']}, E={r['embedding_size']}, "
                                  f"C={r['n_channels_V']}, B={r['batch']}, Adam",
                        "value": r["batch"] * r["n_vec"] * r["steps"] / r["seconds"], "unit": "tokens/s",
                        "ms_per_step": r["seconds"] * 1e3 / r["steps"], "device_ms_per_step": r["event_ms"] / r["steps"],
                        "steps": r["steps"], "loss": r["loss"], "hip_graph": r["hip_graph"],
                        "eager_ms_per_step": eager["seconds"] * 1e3 / eager["steps"], "eager_loss": eager["loss"]}
    except Exception as exc:
        out["error"] = repr(exc)
    return out


def e2e_forward_leg(device, reps=20):
    """SURVEY.md §8(d) end-to-end variant: the real PSFNet (seed 42, SyntheticExperiments/psf_training.py:16,29-45) on generated
    Adding and Temporal-Order batches (synth_data_generation.py:8-70), N = 16384, B = 64, no_grad, whole ``net(x)``:
    tokens/s for the two routes of the mixer — W_m written by the producer kernel and read by the chain
    (``w_through_memory``) and W_m computed inside each chain step, never in memory (``w_in_step``, psf_mixer_fwd_f32) — their
    HBM bytes per token by the algorithmic model, the device time of the parts of the first route, and the largest relative
    difference between the two routes' logits. One GPU, outside the headline's timed region."""
    import torch
    out = {}
    try:
        from sparsefactorization_amd import fused_mixer
        from sparsefactorization_amd.psf_training import build_model, make_split
        from sparsefactorization_amd.psfnet import _flat_head
        from sparsefactorization_amd.token_linear import embed_tokens
        from sparsefactorization_amd.train import seed_everything

        def timed(fn, n=reps):
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(device)
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize(device)
            return e0.elapsed_time(e1) / n  # ms

        B, N = B_PER_GPU, N_SEQ
        for problem in ("adding", "order"):
            seed_everything(42)
            net = build_model(problem, N).to(device).eval()
            x, _ = make_split(problem, B, N, device, 42)
            E, C, M, L = net.embedding_size, net.n_channels_V, net.n_W, net.n_links
            n_cls = net.n_class
            in_bytes = 8 if problem == "adding" else 8  # [x, marker] f32 / one int64 token
            with torch.no_grad():
                def embed():
                    if problem == "order":
                        return embed_tokens(x.squeeze(-1), net.embedding, net.pos_embedding.weight)
                    return net.init_linear(x)

                data = embed()
                V0, links = net.produce(data)
                VM = net.mix(data, V0, net.use_residuals, links)
                parts = {"embed": timed(embed), "producer_mlps": timed(lambda: net.produce(data)),
                         "chain": timed(lambda: net.mix(data, V0, net.use_residuals, links)),
                         "head": timed(lambda: _flat_head(net.final, VM.reshape(B, -1)))}
                res = {}
                logits = {}
                for route, tag in (("never", "w_through_memory"), ("always", "w_in_step")):
                    fused_mixer.route = route
                    try:
                        logits[tag] = net(x).float().clone()
                        ms = timed(lambda: net(x))
                    finally:
                        fused_mixer.route = "auto"
                    # algorithmic HBM bytes per token: input, data written + read, [W written + read,] V traffic of the M steps, head
                    common = in_bytes + 4 * E + 4 * C
                    if tag == "w_through_memory":
                        model = common + 4 * E + 4 * (C + M * L) + M * 4 * (L + 3 * C if net.use_residuals else L + 2 * C)
                    else:
                        model = common + 4 * E + 4 * C + M * 4 * (E + (3 if net.use_residuals else 2) * C)
                    res[tag] = {"ms_per_forward": ms, "tokens_per_s": B * N / (ms * 1e-3), "hbm_bytes_per_token_model": model,
                                "achieved_GBps_of_model_bytes": model * B * N / (ms * 1e-3) / 1e9,
                                "frac_of_hbm_peak": model * B * N / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
                ref = logits["w_through_memory"]
                rel = float((logits["w_in_step"] - ref).abs().max() / ref.abs().max().clamp_min(1e-30))
            out[problem] = {"metric": f"PSFNet forward tokens/s, {problem} N={N} B={B} (E={E}, C={C}, M={M}, n_class={n_cls})",
                            "unit": "tokens/s", **res, "w_through_memory_parts_ms": parts,
                            "speedup_w_in_step": res["w_through_memory"]["ms_per_forward"] / res["w_in_step"]["ms_per_forward"],
                            "logits_rel_diff_between_routes": rel}
            del net, x, data, V0, links, VM
            torch.cuda.empty_cache()
    except Exception as exc:
        out["error"] = repr(exc)
    return out


def pmc_shapes():
    """Per-shape counter summary of the newest committed profiles/*_bwd_pmc.json (profiles/collect_bwd.sh: forward step, dV, dW
    and fused backward kernels of the `shapes` leg's shapes) IF it was collected on this tree's kernel sources (csrc_hash);
    otherwise (None, reason) — as pmc_traffic() does for the headline kernel."""
    try:
        from sparsefactorization_amd.build import csrc_hash
        now = csrc_hash()
        pdir = os.path.join(ROOT, "profiles")
        best = None
        for f in sorted(os.listdir(pdir)):
            if f.endswith("_bwd_pmc.json"):
                with open(os.path.join(pdir, f)) as fh:
                    d = json.load(fh)
                if "shapes" in d and "csrc_hash" in d:
                    best = (f, d)
        if best is None:
            return None, "no profiles/*_bwd_pmc.json with a csrc_hash"
        f, d = best
        if d["csrc_hash"] != now:
            return None, f"profiles/{f} @ csrc {d['csrc_hash'][:12]} is stale: csrc is now {now[:12]}"
        return d["shapes"], f"profiles/{f} @ csrc {now[:12]}"
    except Exception as exc:
        return None, f"unavailable: {exc!r}"


def _counter_traffic(pmc, shape_key, kernel_prefix, alg_bytes_per_launch=None):
    """Counters of the shape's main launch of ONE kernel family from the per-shape summary, or None: memory-side bytes and L2
    requests per launch, the rocprofv3 duration they were collected beside, and the L2 -> CU rate they amount to."""
    if not pmc or shape_key not in pmc:
        return None
    best = None
    for name, k in pmc[shape_key].get("kernels", {}).items():
        if name.startswith(kernel_prefix) and "traffic_bytes" in k and k.get("calls", 0) >= 50:
            if best is None or k["calls"] > best[1]["calls"]:
                best = (name, k)
    if best is None:
        return None
    name, k = best
    rows = pmc[shape_key]["B"] * pmc[shape_key]["N"]
    out = {"kernel": name, "rocprof_avg_us": k.get("avg_us"), "traffic_bytes": k["traffic_bytes"],
           "traffic_over_algorithmic": (k["traffic_bytes"] / alg_bytes_per_launch if alg_bytes_per_launch else k.get("traffic_over_alg")),
           "l2_requests_per_row": k.get("tcc_req_per_row"), "l2_hit_rate": k.get("l2_hit_rate")}
    if k.get("tcc_req_per_row") and k.get("avg_us"):
        out["l2_to_cu_GBps"] = k["tcc_req_per_row"] * rows * L2_REQUEST_BYTES / k["avg_us"] / 1e3
    return out


def _with_roof(entry, counters):
    """Attach a kernel's own counters and state the roof that applies (module docstring, `bound`)."""
    entry["bound"] = "hbm"
    if counters is None:
        return entry
    entry["counters"] = counters
    toa, frac = counters.get("traffic_over_algorithmic"), entry.get("frac_of_hbm_peak_rotating", 1.0)
    if toa is not None and toa <= 1.1 and frac < 0.6 and "l2_to_cu_GBps" in counters:
        entry["bound"] = "l2"
        entry["l2_to_cu_GBps"] = counters["l2_to_cu_GBps"]
        entry["l2_roof_GBps"] = list(L2_ROOF_GBS)
        # (every request priced at a full 128-byte line, so the rate is an upper estimate — 4-byte far-W loads and partial
        # lines move less — and a fraction above 1 reads "at the roof"; the roof is the guide's range, low end .. high end)
        entry["l2_request_bytes_assumed"] = L2_REQUEST_BYTES
        entry["l2_to_cu_reading"] = "upper estimate: every request priced at a full line; > 1 of the roof reads 'at the roof'"
        entry["frac_of_l2_roof"] = [counters["l2_to_cu_GBps"] / L2_ROOF_GBS[1], counters["l2_to_cu_GBps"] / L2_ROOF_GBS[0]]
    elif toa is not None and toa > 1.1:
        entry["bound"] = "hbm (re-reads: traffic over algorithmic %.2f)" % toa
    return entry


ROTATE_FOOTPRINT = 640 * 1000 * 1000  # bytes an operand rotation has to span: 2.5 x the 256 MB Infinity Cache
ROTATE_MAX_SETS = 48


def shapes_leg(device):
    """SURVEY.md §8(d) "other timed shapes" (reference shapes: LRA/psf_training_config.py:2-30,60-88, Genome_Clf/
    genome_training_config.py:6-16): us per step and the fraction of 8 TB/s that the step's ALGORITHMIC bytes amount to, for
    the forward chain (bytes 4 B N (L + 2C [+ C residual]) per step) and for one backward step with both gradients (bytes
    4 B N (2L + 3C)). Two timings each, HIP events over back-to-back calls:
      * `*_cache_resident`: ONE operand set re-used by every call. Every LRA shape's working set (<= 70 MB) then sits in the
        256 MB Infinity Cache: the figure is a cache-resident rate, NOT an HBM rate (the round-4 review's point), and is named so;
      * `*_rotating`: the calls rotate through enough operand sets to span 2.5 x the Infinity Cache (or 48 sets for the tiny
        cfg1), as a training step or a stream of batches sees them: the fraction of the HBM roofline. Backward: W, V and dW
        rotate; dZ is the dV the launch before wrote (two buffers taking turns), as in the chain's backward.
    `*_counters`: memory-side bytes and L2 requests per row of the same kernels from the newest committed per-shape counter
    summary, when it was collected on these sources."""
    import torch
    out = {}
    try:
        import sparsefactorization_amd as sfa
        from sparsefactorization_amd import _lib
        from sparsefactorization_amd.chord import _launch_bwd
        pmc, pmc_src = pmc_shapes()
        out["counters_source"] = pmc_src
        # name, B, N, L, C, residual, attention-map mode (first operand eye(N), unbatched), key in the counter summary
        shapes = [("cfg1_adding_n128", 40, 128, 8, 8, True, False, None),
                  ("cfg3_listops_reference", 32, 2000, 12, 128, False, False, "cfg3_ref"),
                  ("cfg3_listops_baseline_wording", 32, 2048, 12, 64, False, False, "cfg3_baseline"),
                  ("cfg4_pathfinder_chain", 64, 1024, 12, 32, False, False, "cfg4"),
                  ("cfg4_pathfinder_attention_map", 8, 1024, 12, 1024, False, True, None),
                  ("genome_like", 16, 16384, 15, 32, False, False, "genome_like")]

        def timed(fn, n):
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(device)
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize(device)
            return e0.elapsed_time(e1) * 1e3 / n  # us

        for name, B, N, L, C, res, amap, pkey in shapes:
            M = L - 1
            g = torch.Generator(device=device).manual_seed(7)
            chain_bytes = M * 4 * B * N * L + 4 * B * N * (C if not amap else N)  # the W_m of one chain + its V0
            sets = max(2, min(ROTATE_MAX_SETS, -(-ROTATE_FOOTPRINT // chain_bytes)))
            Wsets = [[0.1 * torch.randn(B, N, L, device=device, generator=g) for _ in range(M)] for _ in range(sets)]
            V0 = torch.eye(N, device=device) if amap else torch.randn(B, N, C, device=device, generator=g)
            V0s = [V0] if amap else [V0] + [torch.randn(B, N, C, device=device, generator=g) for _ in range(sets - 1)]
            it = [0]

            def chain_rot():
                s_ = it[0] % sets
                it[0] += 1
                sfa.chord_chain(Wsets[s_], V0s[s_ % len(V0s)], res)

            with torch.no_grad():
                fwd_bytes = 4 * B * N * (L + 2 * C + (C if res else 0))

                def fwd_entry(kernel, what, launches_per_chain):
                    t_res = timed(lambda: sfa.chord_chain(Wsets[0], V0, res), 20)
                    t_rot = timed(chain_rot, max(20, 2 * sets))
                    return {"kernel": kernel, "what": what, "launches_per_chain": launches_per_chain,
                            "us_per_step_cache_resident": t_res / M,
                            "frac_of_hbm_peak_cache_resident": fwd_bytes / (t_res / M * 1e-6) / 1e9 / HBM_PEAK_GBS,
                            "us_per_step_rotating": t_rot / M,
                            "frac_of_hbm_peak_rotating": fwd_bytes / (t_rot / M * 1e-6) / 1e9 / HBM_PEAK_GBS}

                entry = {"B": B, "N": N, "L": L, "C": C, "M": M, "residual": res, "fwd_bytes_per_step": fwd_bytes,
                         "rotation": {"operand_sets": sets, "footprint_MB": sets * chain_bytes / 1e6}}
                # (a) the per-step kernel — what training runs (every step's result kept): M launches per chain, forced here
                _lib.set_tuning("chain_fused", 0)
                try:
                    step = fwd_entry(_lib.describe_fwd(B, N, L, N if amap else C), "M per-step launches (knob chain_fused = 0): "
                                     "what a training forward runs", M)
                finally:
                    _lib.set_tuning("chain_fused", 1)
                # counters of the per-step kernel: one launch = one step without residual (profiles/bwd_pmc_run.py)
                entry["fwd_step_kernel"] = _with_roof(step, _counter_traffic(pmc, pkey, "chord_fwd_", 4 * B * N * (L + 2 * C)))
                # (b) what a no-grad chord_chain call runs by itself: the single LDS-resident launch where the library takes it
                chain_desc = _lib.describe_chain_fwd(B, N, L, N if amap else C, M)
                if "chord_chain_" in chain_desc:  # chord_chain_lds_k or chord_chain_rows_k
                    chain = fwd_entry(chain_desc, "ONE launch for all M steps, only the last result kept: what a no-grad "
                                      "chord_chain (inference) runs", 1)
                    cc = _counter_traffic(pmc, pkey, "chord_chain_", M * 4 * B * N * (L + 2 * C))
                    if cc is not None and cc.get("l2_requests_per_row"):
                        cc["l2_requests_per_row_per_step"] = cc["l2_requests_per_row"] / M
                    entry["fwd_chain_kernel"] = _with_roof(chain, cc)
                else:
                    entry["fwd_chain_kernel"] = {"same_as": "fwd_step_kernel", "kernel": chain_desc}
                if not amap:
                    step_bytes_in = 4 * B * N * (L + 2 * C)
                    bsets = max(2, min(ROTATE_MAX_SETS, -(-ROTATE_FOOTPRINT // (2 * step_bytes_in))))
                    bsets = min(bsets, sets * M)
                    flatW = [w for ws in Wsets for w in ws][:bsets]
                    Vs = [torch.randn(B, N, C, device=device, generator=g) for _ in range(bsets)]
                    zz = [torch.randn(B, N, C, device=device, generator=g), torch.empty(B, N, C, device=device)]
                    dWs = [torch.empty_like(flatW[0]) for _ in range(bsets)]
                    jt = [0]

                    def bwd_rot():  # W, V, dW new every launch; dZ = the dV of the launch before (as in the chain's backward)
                        s_ = jt[0] % bsets
                        jt[0] += 1
                        _launch_bwd(zz[jt[0] & 1], flatW[s_], Vs[s_], dWs[s_], zz[1 - (jt[0] & 1)], B, N, L, C, N * C, None)

                    t_bwd = timed(lambda: _launch_bwd(zz[0], flatW[0], Vs[0], dWs[0], zz[1], B, N, L, C, N * C, None), 40)
                    t_brot = timed(bwd_rot, max(40, 2 * bsets))
                    bwd_bytes = 4 * B * N * (2 * L + 3 * C)
                    entry["bwd_bytes_per_step"] = bwd_bytes
                    fused_bwd = C <= 64 or (C == 128 and N <= 4096)  # psf_chord.hip: fused_step_width
                    bwd = {"kernel": "chord_bwd_fused_k" if fused_bwd else "chord_dv_win_k + chord_dw_chunk_k (two launches per step)",
                           "what": "one backward step, both gradients; rotating: W, V, dW new every launch, dZ = the dV of the launch before",
                           "us_per_step_cache_resident": t_bwd,
                           "frac_of_hbm_peak_cache_resident": bwd_bytes / (t_bwd * 1e-6) / 1e9 / HBM_PEAK_GBS,
                           "us_per_step_rotating": t_brot,
                           "frac_of_hbm_peak_rotating": bwd_bytes / (t_brot * 1e-6) / 1e9 / HBM_PEAK_GBS,
                           "rotation_operand_sets": bsets}
                    bc = _counter_traffic(pmc, pkey, "chord_bwd_fused", bwd_bytes)
                    if bc is None:  # wide rows: two kernels per step — the dV kernel's counters, against its own bytes
                        bc = _counter_traffic(pmc, pkey, "chord_dv_", 4 * B * N * (L + 2 * C))
                    if bc is not None and fused_bwd and "fused" in bc["kernel"]:
                        bwd["kernel"] = bc["kernel"]
                    if fused_bwd and N < 2 * (256 >> ((C // 4).bit_length() - 1)):  # psf_chord.hip: pick_fused_step wants two tiles of rows
                        bwd["kernel"] = "chord_dv_generic_k + chord_dw_generic_k (two launches per step: under two tiles of rows)"
                    entry["bwd_step_kernel"] = _with_roof(bwd, bc)
                    # (c) the whole backward chain in ONE launch where the library has it (N <= 1024, C <= 8: csrc/bwd_chain_lds.h) —
                    # what chord_chain's autograd node runs there; every step's X kept by the forward, dW_m and dV0 written
                    lib = _lib.load()
                    if lib.psf_chord_chain_bwd_supported(N, L, C, M):
                        import ctypes
                        Xs = [torch.randn(B, N, C, device=device, generator=g) for _ in range(M)]
                        dWc = [torch.empty(B, N, L, device=device) for _ in range(M)]
                        gout, dv0 = torch.randn(B, N, C, device=device, generator=g), torch.empty(B, N, C, device=device)
                        tabs = [(ctypes.c_void_p * M)(*[t.data_ptr() for t in ts]) for ts in (Wsets[0][:M], Xs, dWc)]
                        stream = torch.cuda.current_stream(device).cuda_stream

                        def bwd_chain():
                            _lib.check(lib.psf_chord_chain_bwd_f32(gout.data_ptr(), tabs[0], Xs[0].data_ptr(), tabs[1], tabs[2], dv0.data_ptr(),
                                                                   None, M, 1, B, N, L, C, None, stream), "psf_chord_chain_bwd_f32")
                        t_chain = timed(bwd_chain, 40)
                        entry["bwd_chain_kernel"] = {"kernel": "chord_chain_bwd_lds_k<L=%d,G=%d,residual>" % (L, C // 4),
                                                     "what": "ONE launch for all M backward steps: what chord_chain's backward runs for short "
                                                             "sequences of narrow rows (the per-step path: M x bwd_step_kernel)",
                                                     "us_per_chain": t_chain, "us_per_step_equivalent": t_chain / M, "steps": M}
                        del Xs, dWc, gout, dv0
                    del Vs, dWs, zz, flatW
            out[name] = entry
            del Wsets, V0, V0s
            torch.cuda.empty_cache()
        bvs = backward_vs_stream(device, timed)
        bc = _counter_traffic(pmc, "order_train", "chord_bwd_fused", bvs["step_bytes"])
        bvs["frac_of_hbm_peak_rotating"] = bvs["step_frac_of_hbm_peak"]
        _with_roof(bvs, bc)  # (cache-resident operands under the profiler: its traffic is the memory side of L2, not HBM alone)
        out["backward_step_vs_stream"] = bvs
    except Exception as exc:
        out["error"] = repr(exc)
    return out


def backward_vs_stream(device, timed):
    """The fused backward step of the Temporal-Order training shape (N = 16384, L = 15, C = 8, B = 40) as the chain's backward
    runs it: W_m, V_m and dW_m are new every launch (ten sets, 1.4 GB), dZ is the dV the previous launch wrote (two buffers
    taking turns) — beside the library's gather-free streaming kernel with the same byte mix on the same operands
    (psf_stream_mix_bwd_f32: reads W : V : dZ = 2 : 1 : 1, writes dW : dV = 2 : 1). DESIGN.md 4.3. (Until round 5 dZ was one
    fixed buffer and dV rotated with the rest: a launch rule that helps there — non-temporal dV — cost time in the real step.)"""
    import torch
    from sparsefactorization_amd import _lib
    from sparsefactorization_amd.chord import _launch_bwd
    B, N, L, C, sets = 40, 16384, 15, 8, 10
    g = torch.Generator(device=device).manual_seed(11)
    Ws = [0.1 * torch.randn(B, N, L, device=device, generator=g) for _ in range(sets)]
    Vs = [torch.randn(B, N, C, device=device, generator=g) for _ in range(sets)]
    zz = [torch.randn(B, N, C, device=device, generator=g), torch.empty(B, N, C, device=device)]
    dWs = [torch.empty_like(Ws[0]) for _ in range(sets)]
    nv = B * N * C // 4
    sW = [torch.empty(8 * nv, device=device).fill_(0.5) for _ in range(sets)]  # the stream kernel's W-like operands: 2 vectors per V vector
    sdW = [torch.empty(8 * nv, device=device) for _ in range(sets)]
    lib = _lib.load()
    stream = torch.cuda.current_stream(device).cuda_stream
    it = [0]

    def step():
        s = it[0] % sets
        it[0] += 1
        _launch_bwd(zz[it[0] & 1], Ws[s], Vs[s], dWs[s], zz[1 - (it[0] & 1)], B, N, L, C, N * C, None)

    def stream_step():
        s = it[0] % sets
        it[0] += 1
        _lib.check(lib.psf_stream_mix_bwd_f32(sW[s].data_ptr(), Vs[s].data_ptr(), zz[it[0] & 1].data_ptr(), sdW[s].data_ptr(),
                                              zz[1 - (it[0] & 1)].data_ptr(), nv, stream), "psf_stream_mix_bwd_f32")

    t_step, t_stream = timed(step, 100), timed(stream_step, 100)
    step_bytes, stream_bytes = 4 * B * N * (2 * L + 3 * C), 112 * nv
    return {"shape": {"B": B, "N": N, "L": L, "C": C}, "operands": f"W, V, dW rotating through {sets} sets; dZ = the dV of the previous launch (two buffers)",
            "step_us": t_step, "step_bytes": step_bytes, "step_GBps": step_bytes / t_step / 1e3,
            "step_frac_of_hbm_peak": step_bytes / t_step / 1e3 / HBM_PEAK_GBS,
            "stream_us": t_stream, "stream_bytes": stream_bytes, "stream_GBps": stream_bytes / t_stream / 1e3,
            "step_frac_of_stream": (step_bytes / t_step) / (stream_bytes / t_stream)}


def main() -> int:
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        return self_launch(args)
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
