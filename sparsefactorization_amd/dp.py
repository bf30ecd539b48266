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

    Returns (rank, world_size, device). With WORLD_SIZE unset or 1 nothing is initialised.
    """
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_gpu = torch.cuda.is_available()
    device = torch.device("cuda", local) if use_gpu else torch.device("cpu")
    if use_gpu:
        torch.cuda.set_device(device)
    if world > 1 and not dist.is_initialized():
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
    """Sum-all-reduce every parameter gradient in ONE flat fp32 bucket, then divide by the world size.

    Parameters whose ``.grad`` is None on this step (e.g. ``pos_embedding`` with use_pos_embedding=False, or
    ``embedding`` for problem='adding' — constructed but unused, SyntheticExperiments/psf.py:98-107) contribute
    zeros to the bucket and are left at None afterwards, so the optimizer treats them exactly as on one GPU.
    """

    def __init__(self, params: Iterable[torch.nn.Parameter], group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.numel = sum(p.numel() for p in self.params)
        self._flat: Optional[torch.Tensor] = None

    def _buffer(self, like: torch.Tensor) -> torch.Tensor:
        if self._flat is None or self._flat.device != like.device or self._flat.dtype != like.dtype:
            self._flat = torch.zeros(self.numel, dtype=like.dtype, device=like.device)
        return self._flat

    @torch.no_grad()
    def __call__(self) -> None:
        if not self.params or not dist.is_initialized():
            return
        world = dist.get_world_size(self.group)
        if world == 1:
            return
        flat = self._buffer(self.params[0])
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                flat[off:off + n].zero_()
            else:
                flat[off:off + n].copy_(p.grad.reshape(-1))
            off += n
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        flat.mul_(1.0 / world)
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is not None:
                p.grad.copy_(flat[off:off + n].view_as(p.grad))
            off += n
