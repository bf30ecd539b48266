"""The mixer with W produced inside the chain step — ``psf_mixer_fwd_f32`` (csrc/fwd_mlp_step.h).

``V = g(data); for m: W = fs[m](data); V = spmm(idx, W, V) (+ V0)`` (SyntheticExperiments/psf.py:165-188) as M + 2
launches that never write a W_m: every step kernel computes its tile's rows of W_m on chip from the tile's rows of ``data``.
Inference only (nothing is kept for a backward); ``eligible`` says whether a call can take it, otherwise the caller
produces the W_m (fused_mlp.py) and runs the chain (chord.chord_chain).
"""
from __future__ import annotations

import ctypes
from typing import Sequence

import torch
from torch import nn

from . import _lib
from .fused_mlp import _needs_grad, _params_of, _ptrs, _two_layer

enabled = True  # module-level switch (tests / A-B timing)
#: "auto": take the fused path where it is measured faster than psf_mlp_fwd_f32 + psf_chord_chain_fwd_f32 end to end
#: (profiles/r04c_mixer_ablate.log: long sequences of narrow rows with one hidden unit — the step kernel stages its MLP's
#: weight image per tile, 14 KB per 32 hidden rows, which short tiles (C = 32: 64 rows) and 128-wide hidden layers do not
#: amortise); "always": wherever the shape is covered; "never".
route = "auto"


def _sizes(x: torch.Tensor, g: nn.Module, fs: Sequence[nn.Module]):
    """(N, E, M, h table, C, L) when every block is Linear, GELU(erf), Linear on x's width and the link MLPs agree on L."""
    if x.dim() != 3 or not len(fs):
        return None
    pairs = [_two_layer(b) for b in [g, *fs]]
    if any(p is None for p in pairs):
        return None
    E = x.shape[-1]
    if any(l1.in_features != E or l1.weight.dtype != torch.float32 for l1, _ in pairs):
        return None
    L = pairs[1][1].out_features
    if any(l2.out_features != L for _, l2 in pairs[1:]):
        return None
    h = (ctypes.c_int32 * len(pairs))(*[l1.out_features for l1, _ in pairs])
    return x.shape[1], E, len(fs), h, pairs[0][1].out_features, L


def eligible(x: torch.Tensor, g: nn.Module, fs: Sequence[nn.Module]) -> bool:
    if not enabled or not x.is_cuda or x.dtype != torch.float32 or _needs_grad(x, [g, *fs]):
        return False
    sz = _sizes(x, g, fs)
    if sz is None:
        return False
    N, E, M, h, C, L = sz
    if route == "never" or (route == "auto" and not (N >= 8192 and C <= 16 and max(h) <= 32)):
        return False
    return _lib.load().psf_mixer_fwd_workspace(N, E, M, h, C, L) >= 0


def covered(x: torch.Tensor, g: nn.Module, fs: Sequence[nn.Module]) -> bool:
    """The shape is inside the fused path's limits (whatever ``route`` says about using it)."""
    sz = _sizes(x, g, fs)
    return sz is not None and x.is_cuda and x.dtype == torch.float32 and _lib.load().psf_mixer_fwd_workspace(*sz) >= 0


def mixer_forward(x: torch.Tensor, g: nn.Module, fs: Sequence[nn.Module], use_residual: bool) -> torch.Tensor:
    """V_M [B, N, C] from ``data`` [B, N, E]. Caller checks ``eligible`` first."""
    N, E, M, h, C, L = _sizes(x, g, fs)
    B = x.shape[0]
    dev = x.device
    lib = _lib.load()
    x3 = x.detach().contiguous()
    if x3.data_ptr() % 16:
        x3 = x3.clone()
    params = [p.detach().contiguous() for p in _params_of([g, *fs])]
    ws_bytes = lib.psf_mixer_fwd_workspace(N, E, M, h, C, L)
    if ws_bytes < 0:
        raise ValueError("psf_mixer_fwd does not cover this shape")
    ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=dev)
    V0 = torch.empty((B, N, C), dtype=torch.float32, device=dev)
    bufs = [torch.empty_like(V0) for _ in range(min(M, 2))]
    o_tab = (ctypes.c_void_p * M)(*[bufs[m % len(bufs)].data_ptr() for m in range(M)])
    with torch.cuda.device(dev):
        rc = lib.psf_mixer_fwd_f32(x3.data_ptr(), B, N, E, M, _ptrs(params[0::4]), _ptrs(params[1::4]), _ptrs(params[2::4]),
                                   _ptrs(params[3::4]), h, C, L, 1 if use_residual else 0, V0.data_ptr(), o_tab, ws.data_ptr(),
                                   ws_bytes, torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(rc, "psf_mixer_fwd_f32")
    return bufs[(M - 1) % len(bufs)]
