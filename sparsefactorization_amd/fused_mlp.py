"""Fused PSFNet producer MLPs (``g`` and ``fs[0..M)``) — forward and backward in one launch each.

``MLPBlock`` is ``Linear(E, h) -> GELU -> Linear(h, out)`` (SyntheticExperiments/psf.py:35-60); PSFNet applies
M+1 of them to the same ``data`` (psf.py:165,175).

* ``psf_mlp_fwd_f32`` (csrc/mlp_fwd.hip) evaluates all of them from one read of ``data`` on the f32 matrix
  core, the hidden layer staying in registers.
* ``psf_mlp_bwd_f32`` (csrc/mlp_bwd.hip) is their backward: the hidden layer is recomputed, dX is accumulated
  over all MLPs in registers, weight gradients are reduced in a fixed order. The forward therefore saves only
  ``data`` and the parameters (autograd through the PyTorch layers keeps 2 x [T, (M+1) h] activations).

``eligible`` (no gradient needed) / ``trainable`` (gradient needed) say whether a call can take these kernels;
MLPs of another form, fp64, E > 64 (E > 32 when training), h > 128, out > 32 or CPU tensors use the stock modules.

``wide_ok`` / ``wide_apply`` (csrc/mlp_wide.hip) cover the LRA widths — E up to 1024, outputs up to 128; the reference
ListOps network has E = 512, out = 12 and 128 (LRA/psf_training_config.py:2-30): the M+1 first layers are ONE stacked GEMM
on the bf16 matrix pipe at f32 accuracy, forward, input gradient and weight gradient; the forward keeps the hidden
pre-activations for the backward instead of recomputing them.

``stackable`` / ``stacked_apply`` cover whatever is left (fp64, odd widths) with library GEMMs laid out for them: the M+1 first layers share their input, so they run as ONE Linear(E, sum h) — one GEMM
forward, one GEMM for the input gradient (K = sum h, instead of M+1 GEMMs plus M accumulations of a [T, E]
tensor) and one for the weight gradient; GELU and its backward are one kernel each.
"""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence

import torch
import torch.nn.functional as F
from torch import nn

from . import _lib

MAX_E, MAX_E_TRAIN, MAX_H, MAX_O, MAX_K = 64, 32, 128, 32, 32
enabled = True        # module-level switches (tests / A-B timing)
train_enabled = True

_vp = ctypes.c_void_p


def _two_layer(block: nn.Module) -> Optional[tuple]:
    """(lin1, lin2) if ``block.network`` is exactly Linear, GELU(erf), Linear with biases.
    The answer is remembered on the block and re-validated by identity (six dictionary look-ups): the eligibility checks ask
    it up to four times per block and step, and indexing an nn.Sequential costs microseconds — 0.17 ms of the 1.5 ms the host
    needs to issue a CIFAR-10 training step (profiles/lra_host_profile.py)."""
    seen = block.__dict__.get("_psf_two_layer")
    if seen is not None:
        net, l1, act, l2 = seen
        mods = net._modules
        if (block._modules.get("network") is net and len(mods) == 3 and mods.get("0") is l1 and mods.get("1") is act
                and mods.get("2") is l2 and act.approximate == "none" and l1._parameters.get("bias") is not None
                and l2._parameters.get("bias") is not None):
            return l1, l2
        del block.__dict__["_psf_two_layer"]
    net = getattr(block, "network", None)
    if not isinstance(net, nn.Sequential) or len(net) != 3:
        return None
    l1, act, l2 = net[0], net[1], net[2]
    if not (isinstance(l1, nn.Linear) and isinstance(l2, nn.Linear) and isinstance(act, nn.GELU)):
        return None
    if getattr(act, "approximate", "none") != "none" or l1.bias is None or l2.bias is None:
        return None
    if type(net) is nn.Sequential and all(k in net._modules for k in ("0", "1", "2")):  # (plain containers with the default keys)
        block.__dict__["_psf_two_layer"] = (net, l1, act, l2)
    return l1, l2


def _needs_grad(x: torch.Tensor, blocks: Sequence[nn.Module]) -> bool:
    return torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for b in blocks for p in b.parameters()))


def _shapes_ok(x: torch.Tensor, blocks: Sequence[nn.Module], max_e: int) -> bool:
    if not x.is_cuda or x.dtype != torch.float32 or x.dim() < 2 or not len(blocks):
        return False
    E = x.shape[-1]
    if E < 4 or E > max_e or E % 4:
        return False
    for b in blocks:
        pair = _two_layer(b)
        if pair is None:
            return False
        l1, l2 = pair
        if l1.in_features != E or l1.out_features > MAX_H or l2.out_features > MAX_O or l1.weight.dtype != torch.float32:
            return False
    return True


def eligible(x: torch.Tensor, blocks: Sequence[nn.Module]) -> bool:
    """Inference: the fused forward can replace ``[b(x) for b in blocks]`` and nothing needs a gradient."""
    return enabled and not _needs_grad(x, blocks) and _shapes_ok(x, blocks, MAX_E)


def trainable(x: torch.Tensor, blocks: Sequence[nn.Module]) -> bool:
    """Training: fused forward + fused backward (``fused_mlp_apply``) can replace the PyTorch layers."""
    return enabled and train_enabled and _needs_grad(x, blocks) and len(blocks) <= MAX_K and _shapes_ok(x, blocks, MAX_E_TRAIN)


def _ptrs(tensors: Sequence[torch.Tensor]):
    return (_vp * len(tensors))(*[t.data_ptr() for t in tensors])


def _forward_raw(x2: torch.Tensor, params: Sequence[torch.Tensor]) -> List[torch.Tensor]:
    """x2 [T, E] contiguous; params = (A0, a0, B0, b0, A1, ...) contiguous fp32. One launch per <= MAX_K MLPs."""
    T, E = x2.shape
    dev = x2.device
    lib = _lib.load()
    outs: List[torch.Tensor] = []
    n_mlp = len(params) // 4
    for start in range(0, n_mlp, MAX_K):
        grp = params[4 * start:4 * min(start + MAX_K, n_mlp)]
        K = len(grp) // 4
        As, as_, Bs, bs = grp[0::4], grp[1::4], grp[2::4], grp[3::4]
        ys = [torch.empty((T, B.shape[0]), dtype=torch.float32, device=dev) for B in Bs]
        h = (ctypes.c_int32 * K)(*[A.shape[0] for A in As])
        O = (ctypes.c_int32 * K)(*[B.shape[0] for B in Bs])
        ws_bytes = lib.psf_mlp_fwd_workspace(E, K, h, O)
        if ws_bytes < 0:
            raise ValueError("psf_mlp_fwd does not support these layer sizes")
        ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=dev)  # packed weight images
        with torch.cuda.device(dev):
            rc = lib.psf_mlp_fwd_f32(x2.data_ptr(), T, E, K, _ptrs(As), _ptrs(as_), _ptrs(Bs), _ptrs(bs), h, O, _ptrs(ys),
                                     ws.data_ptr(), ws_bytes, _lib.stream_ptr(dev))
        _lib.check(rc, "psf_mlp_fwd_f32")
        outs.extend(ys)
    return outs


def _backward_raw(x2: torch.Tensor, params: Sequence[torch.Tensor], gys: Sequence[torch.Tensor], need_dx: bool):
    T, E = x2.shape
    dev = x2.device
    lib = _lib.load()
    K = len(params) // 4
    As, as_, Bs = params[0::4], params[1::4], params[2::4]
    h = (ctypes.c_int32 * K)(*[A.shape[0] for A in As])
    O = (ctypes.c_int32 * K)(*[B.shape[0] for B in Bs])
    grads = [torch.empty_like(p) for p in params]
    dX = torch.empty_like(x2) if need_dx else None
    ws_bytes = lib.psf_mlp_bwd_workspace(T, E, K, h, O)
    if ws_bytes < 0:
        raise ValueError("psf_mlp_bwd does not support these layer sizes")
    ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=dev)  # packed weights + per-wave partial sums
    with torch.cuda.device(dev):
        rc = lib.psf_mlp_bwd_f32(x2.data_ptr(), T, E, K, _ptrs(As), _ptrs(as_), _ptrs(Bs), h, O, _ptrs(gys),
                                 dX.data_ptr() if need_dx else None, _ptrs(grads[0::4]), _ptrs(grads[1::4]),
                                 _ptrs(grads[2::4]), _ptrs(grads[3::4]), ws.data_ptr(), ws_bytes,
                                 _lib.stream_ptr(dev))
    _lib.check(rc, "psf_mlp_bwd_f32")
    return dX, grads


class _FusedMLPFn(torch.autograd.Function):
    """(Y_0, ..., Y_{K-1}) = MLPs(x2); saves x2 and the parameters only."""

    @staticmethod
    def forward(ctx, x2, *params):
        params = tuple(p if p.is_contiguous() else p.contiguous() for p in params)  # (no graph is recorded in here: no detach)
        ctx.save_for_backward(x2, *params)
        return tuple(_forward_raw(x2, params))

    @staticmethod
    def backward(ctx, *gys):
        x2, *params = ctx.saved_tensors
        Bs = params[2::4]
        gys = [torch.zeros((x2.shape[0], B.shape[0]), dtype=torch.float32, device=x2.device) if g is None else g.contiguous()
               for g, B in zip(gys, Bs)]
        need_dx = ctx.needs_input_grad[0]
        # The backward kernel sizes its dY prefetch and its weight buffering for the WIDEST output of a call: one MLP
        # with more than 16 outputs (g of the LRA networks: 32 channels) would put all ~50 units of the 12 link MLPs
        # on the slower configuration. Wide and narrow MLPs therefore go in two launches and their dX are added.
        narrow = [k for k, B in enumerate(Bs) if B.shape[0] <= 16]
        wide = [k for k, B in enumerate(Bs) if B.shape[0] > 16]
        if not narrow or not wide:
            dX, grads = _backward_raw(x2, params, gys, need_dx)
            return (dX, *grads)
        grads: List[Optional[torch.Tensor]] = [None] * len(params)
        dX = None
        for group in (narrow, wide):
            sub_params = [params[4 * k + i] for k in group for i in range(4)]
            d, g_sub = _backward_raw(x2, sub_params, [gys[k] for k in group], need_dx)
            for n, k in enumerate(group):
                grads[4 * k:4 * k + 4] = g_sub[4 * n:4 * n + 4]
            if need_dx:
                dX = d if dX is None else dX.add_(d)
        return (dX, *grads)


def _params_of(blocks: Sequence[nn.Module]) -> List[torch.Tensor]:
    out: List[torch.Tensor] = []
    for b in blocks:
        l1, l2 = _two_layer(b)
        out += [l1.weight, l1.bias, l2.weight, l2.bias]
    return out


def fused_mlp_forward(x: torch.Tensor, blocks: Sequence[nn.Module]) -> List[torch.Tensor]:
    """[block(x) for block in blocks] without autograd, by the fused kernel. Caller checks ``eligible`` first."""
    lead, E = x.shape[:-1], x.shape[-1]
    x2 = x.detach().reshape(-1, E).contiguous()
    params = [p.detach().contiguous() for p in _params_of(blocks)]
    return [y.reshape(*lead, y.shape[1]) for y in _forward_raw(x2, params)]


def fused_mlp_apply(x: torch.Tensor, blocks: Sequence[nn.Module]) -> List[torch.Tensor]:
    """[block(x) for block in blocks] under autograd (fused forward and backward). Caller checks ``trainable``."""
    lead, E = x.shape[:-1], x.shape[-1]
    x2 = x.reshape(-1, E).contiguous()
    ys = _FusedMLPFn.apply(x2, *_params_of(blocks))
    return [y.reshape(*lead, y.shape[1]) for y in ys]


WIDE_MAX_E, WIDE_MAX_O, WIDE_MAX_K = 1024, 128, 24
wide_enabled = True


def wide_ok(x: torch.Tensor, blocks: Sequence[nn.Module]) -> bool:
    """The wide kernels (psf_mlp_wide_*) can replace ``[b(x) for b in blocks]``, with or without autograd. Callers try
    ``eligible`` / ``trainable`` (the narrow kernels, which never let a hidden activation reach memory) first."""
    if not (enabled and wide_enabled) or not x.is_cuda or x.dtype != torch.float32 or x.dim() < 2 or not 1 <= len(blocks) <= WIDE_MAX_K:
        return False
    E = x.shape[-1]
    if E < 16 or E > WIDE_MAX_E or E % 16:
        return False
    J = 0
    for b in blocks:
        pair = _two_layer(b)
        if pair is None:
            return False
        l1, l2 = pair
        if l1.in_features != E or l1.out_features > MAX_H or l2.out_features > WIDE_MAX_O or l1.weight.dtype != torch.float32:
            return False
        J += (l1.out_features + 31) // 32 * 32
    T = x.numel() // E
    return T * max(E, J) < 2 ** 30  # 32-bit lane offsets into a bf16 plane


def _wide_sizes(x2: torch.Tensor, params: Sequence[torch.Tensor]):
    K = len(params) // 4
    h = (ctypes.c_int32 * K)(*[A.shape[0] for A in params[0::4]])
    O = (ctypes.c_int32 * K)(*[B.shape[0] for B in params[2::4]])
    return x2.shape[0], x2.shape[1], K, h, O


def _scratch(nbytes: int, dev) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=dev)  # the caching allocator aligns to 512 bytes


def _wide_forward_raw(x2: torch.Tensor, params: Sequence[torch.Tensor], keep: bool = True):
    """(Y_0..Y_{K-1}, saved): ``saved`` holds X as bf16 term planes and the hidden pre-activations (mlp_wide.hip).
    ``keep`` False (inference): no record is kept — the library works in scratch and skips what only a backward reads."""
    T, E, K, h, O = _wide_sizes(x2, params)
    dev = x2.device
    lib = _lib.load()
    As, as_, Bs, bs = params[0::4], params[1::4], params[2::4], params[3::4]
    n_saved, n_ws = lib.psf_mlp_wide_saved_bytes(T, E, K, h, O), lib.psf_mlp_wide_fwd_workspace(T, E, K, h, O)
    if n_saved < 0 or n_ws < 0:
        raise ValueError("psf_mlp_wide_fwd does not support these layer sizes")
    saved = _scratch(n_saved, dev) if keep else None
    ws = _scratch(n_ws + (0 if keep else n_saved + 256), dev)
    ys = [torch.empty((T, B.shape[0]), dtype=torch.float32, device=dev) for B in Bs]
    with torch.cuda.device(dev):
        rc = lib.psf_mlp_wide_fwd_f32(x2.data_ptr(), T, E, K, _ptrs(As), _ptrs(as_), _ptrs(Bs), _ptrs(bs), h, O, _ptrs(ys),
                                      saved.data_ptr() if keep else None, saved.numel() if keep else 0, ws.data_ptr(), ws.numel(),
                                      _lib.stream_ptr(dev))
    _lib.check(rc, "psf_mlp_wide_fwd_f32")
    return ys, saved


def _wide_backward_raw(saved: torch.Tensor, T: int, E: int, params: Sequence[torch.Tensor], gys: Sequence[torch.Tensor], need_dx: bool):
    dev = saved.device
    lib = _lib.load()
    K = len(params) // 4
    As, Bs = params[0::4], params[2::4]
    h = (ctypes.c_int32 * K)(*[A.shape[0] for A in As])
    O = (ctypes.c_int32 * K)(*[B.shape[0] for B in Bs])
    grads = [torch.empty_like(p) for p in params]
    dX = torch.empty((T, E), dtype=torch.float32, device=dev) if need_dx else None
    n_ws = lib.psf_mlp_wide_bwd_workspace(T, E, K, h, O)
    if n_ws < 0:
        raise ValueError("psf_mlp_wide_bwd does not support these layer sizes")
    ws = _scratch(n_ws, dev)
    with torch.cuda.device(dev):
        rc = lib.psf_mlp_wide_bwd_f32(saved.data_ptr(), saved.numel(), T, E, K, _ptrs(As), _ptrs(Bs), h, O, _ptrs(gys),
                                      dX.data_ptr() if need_dx else None, _ptrs(grads[0::4]), _ptrs(grads[1::4]),
                                      _ptrs(grads[2::4]), _ptrs(grads[3::4]), ws.data_ptr(), ws.numel(),
                                      _lib.stream_ptr(dev))
    _lib.check(rc, "psf_mlp_wide_bwd_f32")
    return dX, grads


class _WideMLPFn(torch.autograd.Function):
    """(Y_0, ..., Y_{K-1}) = MLPs(x2) on the wide kernels; saves the kernels' own record of X and the hidden layer."""

    @staticmethod
    def forward(ctx, x2, *params):
        params = tuple(p if p.is_contiguous() else p.contiguous() for p in params)  # (inside Function.forward: nothing to detach from)
        ys, saved = _wide_forward_raw(x2, params)
        ctx.save_for_backward(saved, *params)
        ctx.x_shape = tuple(x2.shape)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *gys):
        saved, *params = ctx.saved_tensors
        T, E = ctx.x_shape
        gys = [torch.zeros((T, B.shape[0]), dtype=torch.float32, device=saved.device) if g is None else g.contiguous()
               for g, B in zip(gys, params[2::4])]
        dX, grads = _wide_backward_raw(saved, T, E, params, gys, ctx.needs_input_grad[0])
        return (dX, *grads)


def wide_apply(x: torch.Tensor, blocks: Sequence[nn.Module]) -> List[torch.Tensor]:
    """[block(x) for block in blocks] on the wide kernels (autograd-aware). Caller checks ``wide_ok`` first."""
    lead, E = x.shape[:-1], x.shape[-1]
    if _needs_grad(x, blocks):
        ys = _WideMLPFn.apply(x.reshape(-1, E).contiguous(), *_params_of(blocks))
    else:
        ys, _ = _wide_forward_raw(x.detach().reshape(-1, E).contiguous(), [p.detach().contiguous() for p in _params_of(blocks)], keep=False)
    return [y.reshape(*lead, y.shape[1]) for y in ys]


def stackable(x: torch.Tensor, blocks: Sequence[nn.Module]) -> bool:
    """GPU call of >= 2 two-layer MLPs that share their input and that the fused kernels do not take."""
    if not enabled or not x.is_cuda or len(blocks) < 2 or x.dim() < 2:
        return False
    pairs = [_two_layer(b) for b in blocks]
    return all(p is not None and p[0].in_features == x.shape[-1] and p[0].weight.dtype == x.dtype for p in pairs)


def stacked_apply(x: torch.Tensor, blocks: Sequence[nn.Module]) -> List[torch.Tensor]:
    """[block(x) for block in blocks] with the first layers stacked into one Linear (autograd-transparent: the
    parameters stay the modules' own tensors; ``torch.cat`` routes their gradients back)."""
    pairs = [_two_layer(b) for b in blocks]
    lead, E = x.shape[:-1], x.shape[-1]
    x2 = x.reshape(-1, E)
    hidden = F.gelu(F.linear(x2, torch.cat([l1.weight for l1, _ in pairs], 0), torch.cat([l1.bias for l1, _ in pairs], 0)))
    parts = hidden.split([l1.out_features for l1, _ in pairs], dim=-1)  # column slices: GEMM operands with lda = sum h
    from .token_linear import _TokenLinearFn, wgrad_supported  # (token_linear imports nothing from here)
    outs = []
    for p, (_, l2) in zip(parts, pairs):
        # second-layer weight gradients are [out x T] * [T x h] reductions into a tiny tile: on the tall-skinny MFMA
        # kernel (reads the slice with its row stride) where it applies — 135 us each through hipBLASLt at ListOps sizes
        if torch.is_grad_enabled() and l2.weight.requires_grad and wgrad_supported(p, l2.out_features):
            y = _TokenLinearFn.apply(p, l2.weight, l2.bias)
        else:
            y = F.linear(p, l2.weight, l2.bias)
        outs.append(y.reshape(*lead, l2.out_features))
    return outs
