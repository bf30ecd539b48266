"""Fused forward of PSFNet's producer MLPs (``g`` and ``fs[0..M)``) — inference path.

``MLPBlock`` is ``Linear(E, h) -> GELU -> Linear(h, out)`` (SyntheticExperiments/psf.py:35-60); PSFNet applies
M+1 of them to the same ``data`` (psf.py:165,175). ``psf_mlp_fwd_f32`` (csrc/mlp_fwd.hip) evaluates all of them
in one launch from one read of ``data`` on the f32 matrix core, the hidden layer staying in registers.

Used only when nothing needs a gradient (eval / ``torch.no_grad()``): training keeps the PyTorch layers so that
autograd has its saved activations (their weight gradients run on ``psf_linear_wgrad_f32``, token_linear.py).
MLPs of another form, fp64, E > 64, h > 128, out > 32 or CPU tensors use the stock modules.
"""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence

import torch
from torch import nn

from . import _lib

MAX_E, MAX_H, MAX_O, MAX_K = 64, 128, 32, 32
enabled = True  # module-level switch (tests / A-B timing)


def _two_layer(block: nn.Module) -> Optional[tuple]:
    """(lin1, lin2) if ``block.network`` is exactly Linear, GELU(erf), Linear with biases."""
    net = getattr(block, "network", None)
    if not isinstance(net, nn.Sequential) or len(net) != 3:
        return None
    l1, act, l2 = net[0], net[1], net[2]
    if not (isinstance(l1, nn.Linear) and isinstance(l2, nn.Linear) and isinstance(act, nn.GELU)):
        return None
    if getattr(act, "approximate", "none") != "none" or l1.bias is None or l2.bias is None:
        return None
    return l1, l2


def eligible(x: torch.Tensor, blocks: Sequence[nn.Module]) -> bool:
    if not enabled or torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for b in blocks for p in b.parameters())):
        return False
    if not x.is_cuda or x.dtype != torch.float32 or x.dim() < 2:
        return False
    E = x.shape[-1]
    if E < 4 or E > MAX_E or E % 4:
        return False
    for b in blocks:
        pair = _two_layer(b)
        if pair is None:
            return False
        l1, l2 = pair
        if l1.in_features != E or l1.out_features > MAX_H or l2.out_features > MAX_O or l1.weight.dtype != torch.float32:
            return False
    return True


def fused_mlp_forward(x: torch.Tensor, blocks: Sequence[nn.Module]) -> List[torch.Tensor]:
    """[block(x) for block in blocks], computed by the fused kernel. Caller checks ``eligible`` first."""
    lead = x.shape[:-1]
    E = x.shape[-1]
    x2 = x.reshape(-1, E).contiguous()
    T = x2.shape[0]
    outs: List[torch.Tensor] = []
    lib = _lib.load()
    dev = x.device
    for start in range(0, len(blocks), MAX_K):
        group = blocks[start:start + MAX_K]
        K = len(group)
        pairs = [_two_layer(b) for b in group]
        ys = [torch.empty((T, l2.out_features), dtype=torch.float32, device=dev) for _, l2 in pairs]
        keep = [t.detach().contiguous() for l1, l2 in pairs for t in (l1.weight, l1.bias, l2.weight, l2.bias)]
        vp = ctypes.c_void_p
        A = (vp * K)(*[keep[4 * k].data_ptr() for k in range(K)])
        a = (vp * K)(*[keep[4 * k + 1].data_ptr() for k in range(K)])
        B = (vp * K)(*[keep[4 * k + 2].data_ptr() for k in range(K)])
        b = (vp * K)(*[keep[4 * k + 3].data_ptr() for k in range(K)])
        h = (ctypes.c_int32 * K)(*[l1.out_features for l1, _ in pairs])
        O = (ctypes.c_int32 * K)(*[l2.out_features for _, l2 in pairs])
        Y = (vp * K)(*[y.data_ptr() for y in ys])
        ws_bytes = lib.psf_mlp_fwd_workspace(E, K, h, O)
        if ws_bytes < 0:
            raise ValueError("psf_mlp_fwd does not support these layer sizes")
        ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=dev)  # packed weight images
        with torch.cuda.device(dev):
            rc = lib.psf_mlp_fwd_f32(x2.data_ptr(), T, E, K, A, a, B, b, h, O, Y, ws.data_ptr(), ws_bytes,
                                     torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(rc, "psf_mlp_fwd_f32")
        outs.extend(y.reshape(*lead, y.shape[1]) for y in ys)
    return outs
