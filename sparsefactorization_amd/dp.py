"""Batch-sharded data parallelism for the PSF path: one process per GPU, replicas of the (small) model, one
flat gradient all-reduce per step over RCCL/xGMI.

The reference has no multi-GPU code at all; this is the one parallelism the path admits:
every batch element of ``out[b] = W[b] (.) V[b]`` is independent (spmul_cuda.cu:20 outer loop), so the forward
and backward chains need no communication, and only the parameter gradients are exchanged. The sequence axis
is never sharded: link offsets reach N/2, every step would need a near-global exchange.

Sizing (Order @ N=16384: ~1.07 M parameters = 4.3 MB fp32): a single bucket, a single ``all_reduce(SUM)``,
latency-bound on xGMI (7 point-to-point links x ~153 GB/s per GPU) — bucketing/overlap would only add launches.
Backend ``"nccl"`` is RCCL on ROCm builds of PyTorch; ``"gloo"`` works for CPU rehearsal (tests).
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional, Sequence

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> tuple:
    """Join the process group described by RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torch.distributed.run).

    Returns (rank, world_size, device). Without RANK in the environment (a plain ``python`` start) nothing is
    initialised; under ``torch.distributed.run`` the group is created even for a single rank, so a 1-GPU box
    exercises the same RCCL code path as an 8-GPU node.
    """
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_gpu = torch.cuda.is_available()
    device = torch.device("cuda", local) if use_gpu else torch.device("cpu")
    if use_gpu:
        torch.cuda.set_device(device)
    if (world > 1 or "RANK" in os.environ) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or ("nccl" if use_gpu else "gloo")
        kwargs = {"device_id": device} if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kwargs)
    return rank, world, device


def shard_bounds(n_items: int, rank: int, world: int, drop_last: bool = True) -> tuple:
    """[lo, hi) of this rank's contiguous share of ``n_items``. ``drop_last`` gives every rank the same count
    (the reference loaders all use drop_last=True, SyntheticExperiments/psf_training.py:84)."""
    if drop_last:
        per = n_items // world
        return rank * per, (rank + 1) * per
    per = (n_items + world - 1) // world
    return min(rank * per, n_items), min((rank + 1) * per, n_items)


def shard_batch(tensors: Sequence[torch.Tensor], rank: int, world: int) -> List[torch.Tensor]:
    """Slice every tensor's leading (batch) axis to this rank's shard of the GLOBAL batch."""
    lo, hi = shard_bounds(tensors[0].shape[0], rank, world)
    return [t[lo:hi] for t in tensors]


def broadcast_parameters(module: torch.nn.Module, src: int = 0, group=None) -> None:
    """Make every replica start from rank ``src``'s weights (and buffers)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t, src=src, group=group)


class FlatGradAllReduce:
    """Average every parameter gradient over the ranks with ONE collective on ONE flat fp32 bucket.

    Per step: one ``torch.cat`` of all gradients into the bucket, one ``all_reduce`` (AVG on RCCL; SUM then a scale
    on backends without AVG), one multi-tensor copy back into the ``.grad`` tensors — three device operations whatever
    the parameter count (a PSFNet has ~60 small parameters; the per-parameter copies of a naive bucket are ~120
    launches, more host time than the 4.3 MB collective itself).

    Parameters whose ``.grad`` is None on this step (e.g. ``pos_embedding`` with use_pos_embedding=False, or
    ``embedding`` for problem='adding' — constructed but unused, SyntheticExperiments/psf.py:98-107) contribute
    zeros to the bucket (the bucket layout is the same on every rank and every step) and are left at None
    afterwards, so the optimizer treats them exactly as on one GPU.

    ``force=True`` runs the three operations in a 1-rank group too (a 1-rank all-reduce is the identity).
    ``timing=True`` records a HIP event pair around every call (bucket fill + collective + copy back);
    ``mean_us()`` reads them back — the "µs per all-reduce" that bench.py reports next to the step time.
    """

    def __init__(self, params: Iterable[torch.nn.Parameter], group=None, timing: bool = False, force: bool = False):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.force = force  # run the collective even in a 1-rank group (exercises the RCCL path on a 1-GPU box)
        self.numel = sum(p.numel() for p in self.params)
        self.timing = timing
        self._flat: Optional[torch.Tensor] = None
        self._views: List[torch.Tensor] = []
        self._zeros: Optional[torch.Tensor] = None
        self._events: List[tuple] = []
        self.calls = 0

    def _buffers(self, like: torch.Tensor) -> None:
        if self._flat is None or self._flat.device != like.device or self._flat.dtype != like.dtype:
            self._flat = torch.zeros(self.numel, dtype=like.dtype, device=like.device)
            self._views, off = [], 0
            for p in self.params:
                self._views.append(self._flat[off:off + p.numel()].view(p.shape))
                off += p.numel()
            self._zeros = torch.zeros(max([p.numel() for p in self.params] + [1]), dtype=like.dtype, device=like.device)

    @torch.no_grad()
    def __call__(self) -> None:
        if not self.params or not dist.is_initialized():
            return
        world = dist.get_world_size(self.group)
        if world == 1 and not self.force:
            return
        self._buffers(self.params[0])
        timed = self.timing and self._flat.is_cuda
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        live = [i for i, p in enumerate(self.params) if p.grad is not None]
        pieces = [p.grad.reshape(-1) if p.grad is not None else self._zeros[:p.numel()] for p in self.params]
        torch.cat(pieces, out=self._flat)
        backend = dist.get_backend(self.group)
        if backend == "nccl":  # RCCL: the average is part of the collective
            dist.all_reduce(self._flat, op=dist.ReduceOp.AVG, group=self.group)
        else:
            dist.all_reduce(self._flat, op=dist.ReduceOp.SUM, group=self.group)
            self._flat.mul_(1.0 / world)
        if live:
            torch._foreach_copy_([self.params[i].grad for i in live], [self._views[i] for i in live])
        if timed:
            e1.record()
            self._events.append((e0, e1))
        self.calls += 1

    def mean_us(self, reset: bool = True) -> Optional[float]:
        """Mean device time of the recorded calls in microseconds (synchronises); None when nothing was timed."""
        if not self._events:
            return None
        torch.cuda.synchronize()
        total = sum(a.elapsed_time(b) for a, b in self._events)
        n = len(self._events)
        if reset:
            self._events = []
        return total * 1e3 / n

    def reset_timing(self) -> None:
        self._events = []
