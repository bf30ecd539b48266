"""The mixer with W produced inside the chain step — ``psf_mixer_fwd_f32`` (csrc/fwd_mlp_step.h).

``V = g(data); for m: W = fs[m](data); V = spmm(idx, W, V) (+ V0)`` (SyntheticExperiments/psf.py:165-188) as M + 2
launches that never write a W_m: every step kernel computes its tile's rows of W_m on chip from the tile's rows of ``data``.
Inference only (nothing is kept for a backward); ``eligible`` says whether a call can take it, otherwise the caller
produces the W_m (fused_mlp.py) and runs the chain (chord.chord_chain).
"""
from __future__ import annotations

import ctypes
import threading
from typing import Sequence

import torch
from torch import nn

from . import _lib
from .fused_mlp import _needs_grad, _ptrs, _two_layer

enabled = True  # module-level switch (tests / A-B timing)
#: "auto": take the fused path where it is measured faster than psf_mlp_fwd_f32 + psf_chord_chain_fwd_f32 end to end
#: (profiles/r04l_mixer_bench.log: long sequences of narrow rows with one hidden unit — the step kernel stages its MLP's
#: weight image per tile, 14 KB per 32 hidden rows, which short tiles (C = 32: 64 rows) and 128-wide hidden layers do not
#: amortise — and the short sequences that run as one LDS-resident launch); "always": wherever the shape is covered; "never".
route = "auto"
#: Hand the single-launch kernel (short sequences, csrc/mixer_lds.h) the RECIPE of ``data`` — the affine input layer or the
#: embedding lookup, include/psf_chord.h: psf_mixer_input — instead of its rows. Off: ``data`` is written once
#: (psf_affine_rows_f32 / psf_embed_tokens_f32) and passed as rows. The per-step kernels take rows only: evaluating a recipe
#: inside them measured slower than the one pass that writes the rows (712 against 609 + 28 us at Temporal Order's shape,
#: 0.97 against 0.82 ms per Adding forward; profiles/r04h_mixer_bench.log, r04w_bench_stdout.log) and round 5 removed those
#: instances (two thirds of that translation unit).
recipe_in_kernel = False


class Recipe:
    """How ``data`` [B, N, E] comes about (include/psf_chord.h: psf_mixer_input): given (``data``), an affine map of a few
    inputs per position (``affine``: init_linear of the Adding network, psf.py:153-154) or an embedding lookup (``tokens``:
    psf.py:151-152), each optionally plus a positional row. The single-launch kernel evaluates the last two itself (``data``
    is then never written); for the per-step kernels ``rows()`` writes it once."""

    def __init__(self, kind, src, weight=None, bias=None, pos=None, K=0, E=None):
        self.kind, self.src, self.weight, self.bias, self.pos, self.K = kind, src, weight, bias, pos, K
        self.E = E if E is not None else (src.shape[-1] if kind == _lib.MIXER_IN_DATA else weight.shape[0 if kind == _lib.MIXER_IN_AFFINE else 1])

    @staticmethod
    def data(x):
        return Recipe(_lib.MIXER_IN_DATA, x)

    @staticmethod
    def affine(inp, linear: nn.Linear, pos=None):
        return Recipe(_lib.MIXER_IN_AFFINE, inp, linear.weight, linear.bias, pos, K=inp.shape[-1])

    @staticmethod
    def tokens(idx, table: torch.Tensor, pos=None):
        return Recipe(_lib.MIXER_IN_TOKENS, idx, table, None, pos, K=table.shape[0])

    @property
    def B(self):
        return self.src.shape[0]

    @property
    def N(self):
        return self.src.shape[1]

    def tensors(self):
        return [t for t in (self.src, self.weight, self.bias, self.pos) if t is not None]

    def rows(self) -> torch.Tensor:
        """``data`` [B, N, E] written once: psf_affine_rows_f32 / psf_embed_tokens_f32 (+ the positional rows)."""
        from .token_linear import _EmbedTokensFn, affine_rows
        if self.kind == _lib.MIXER_IN_DATA:
            return self.src
        with torch.no_grad():
            if self.kind == _lib.MIXER_IN_AFFINE:
                x = affine_rows(self.src, self.weight, self.bias)
                return x if self.pos is None else x + self.pos.unsqueeze(0)
            return _EmbedTokensFn.apply(self.src, self.weight, self.pos, None)

    def ok(self):
        """Shapes and dtypes the library takes; anything else makes the caller materialise ``data`` instead."""
        s = self.src
        if not s.is_cuda:
            return False
        if self.kind == _lib.MIXER_IN_DATA:
            return s.dim() == 3 and s.dtype == torch.float32
        if self.pos is not None and (self.pos.dtype != torch.float32 or tuple(self.pos.shape) != (self.N, self.E)):
            return False
        if self.kind == _lib.MIXER_IN_AFFINE:
            return (s.dim() == 3 and s.dtype == torch.float32 and 1 <= s.shape[-1] <= 3 and self.weight.dtype == torch.float32
                    and tuple(self.weight.shape) == (self.E, s.shape[-1]))
        return s.dim() == 2 and s.dtype == torch.int64 and self.weight.dtype == torch.float32 and self.weight.dim() == 2


def _block_pairs(E: int, g: nn.Module, fs: Sequence[nn.Module]):
    """((M, h table, C, L), [(lin1, lin2), ...]) when every block is Linear, GELU(erf), Linear of input width E and the link
    MLPs agree on L; None otherwise."""
    if not len(fs):
        return None
    pairs = [_two_layer(b) for b in [g, *fs]]
    if any(p is None for p in pairs):
        return None
    if any(l1.in_features != E or l1.weight.dtype != torch.float32 for l1, _ in pairs):
        return None
    L = pairs[1][1].out_features
    if any(l2.out_features != L for _, l2 in pairs[1:]):
        return None
    h = (ctypes.c_int32 * len(pairs))(*[l1.out_features for l1, _ in pairs])
    return (len(fs), h, pairs[0][1].out_features, L), pairs


def _block_sizes(E: int, g: nn.Module, fs: Sequence[nn.Module]):
    """(M, h table, C, L), or None (see _block_pairs)."""
    found = _block_pairs(E, g, fs)
    return None if found is None else found[0]


def _key(E: int, g: nn.Module, fs: Sequence[nn.Module]):
    return (E, id(g), tuple(map(id, fs)))


# What a successful eligibility check found, for the mixer_forward_in that follows it at once ON THE SAME THREAD (a third
# fewer Python calls per no-grad forward of an LRA network): (key, sizes, pairs). Thread-local — two threads serving two models
# never see each other's layer pairs — set only when the check says yes, consumed (and cleared) by the next forward, so it
# keeps no module alive beyond that.
_handoff = threading.local()


def _route_ok(N: int, E: int, M: int, C: int, L: int, h, tokens: int = 1 << 62) -> bool:
    if route != "auto":
        return route == "always"
    if N >= 8192 and C <= 32 and max(h) <= 32:  # (32-channel rows since the step kernel's diet: genome shape 488 -> 451 us)
        return True
    # short sequences: ONE launch with V resident in LDS (csrc/mixer_lds.h) against producer + chain
    # (profiles/r04n_mixer_bench_short.log: cfg1 115 -> 89 us per forward, N = 512: 134 -> 106)
    plan = _lib.load().psf_mixer_fwd_plan(N, E, M, h, C, L)
    if plan == 2:
        return True
    # An EAGER forward of a small network is bound by the host, and the fused route is one library call for all its M + 2
    # launches: in the no-grad forward of the LRA networks (profiles/infer_route_sweep.py, ms per forward, fused / through
    # memory) CIFAR-10's widths (hidden 16) win at every batch size — 0.227 / 0.270 at 32 k tokens, 0.319 / 0.339 at 524 k —
    # and hidden 128 wins while the GPU time of the heavier step kernels stays under the host's: Pathfinder 0.173 / 0.222 at
    # 16 k tokens, IMDb 0.247 / 0.291 at 32 k, but 0.377 / 0.280 at 65 k. Under stream capture there is no host in the replay
    # and the GPU-time rule above stands.
    if plan == 1 and not torch.cuda.is_current_stream_capturing():
        return (max(h) <= 32 and C <= 32) or tokens <= 40000
    return False


def eligible_recipe(r: Recipe, g: nn.Module, fs: Sequence[nn.Module]) -> bool:
    """The fused mixer can run from this recipe, nothing needs a gradient, and ``route`` wants it."""
    if not enabled or not r.ok() or torch.is_grad_enabled() and any(t.requires_grad for t in r.tensors() if t.is_floating_point()):
        return False
    if _needs_grad(torch.empty(0), [g, *fs]):
        return False
    _handoff.pending = None
    found = _block_pairs(r.E, g, fs)
    if found is None:
        return False
    (M, h, C, L), pairs = found
    ok = _route_ok(r.N, r.E, M, C, L, h, r.B * r.N) and _lib.load().psf_mixer_fwd_workspace(r.N, r.E, M, h, C, L) >= 0
    if ok:
        _handoff.pending = (_key(r.E, g, fs), found[0], pairs)
    return ok


def mixer_forward_in(r: Recipe, g: nn.Module, fs: Sequence[nn.Module], use_residual: bool) -> torch.Tensor:
    """V_M [B, N, C] from the recipe of ``data``. Caller checks ``eligible_recipe`` (or ``covered``) first."""
    pend = getattr(_handoff, "pending", None)
    _handoff.pending = None
    if pend is not None and pend[0] == _key(r.E, g, fs):  # straight after this thread's eligibility check of the same blocks
        (M, h, C, L), pairs = pend[1], pend[2]
    else:
        found = _block_pairs(r.E, g, fs)
        if found is None:
            raise ValueError("psf_mixer_fwd does not cover these blocks: every block must be Linear(E, h) -> GELU(erf) -> "
                             "Linear(h, out) in f32 on input width E, the link MLPs agreeing on L (check eligible() / covered())")
        (M, h, C, L), pairs = found
    B, N, E = r.B, r.N, r.E
    dev = r.src.device
    lib = _lib.load()
    if r.kind != _lib.MIXER_IN_DATA and not (recipe_in_kernel and lib.psf_mixer_fwd_plan(N, E, M, h, C, L) == 2):
        r = Recipe.data(r.rows())  # the per-step kernels take rows (psf_mixer_fwd_plan: 2 = the single-launch kernel runs)

    def prep(t, align):
        if t is None:
            return None
        t = t.detach().contiguous()
        return t if t.data_ptr() % align == 0 else t.clone()

    keep = [prep(r.src, 16 if r.kind == _lib.MIXER_IN_DATA else 8), prep(r.weight, 16), prep(r.bias, 4), prep(r.pos, 16)]
    spec = _lib.MixerInput(r.kind, int(r.K), *[t.data_ptr() if t is not None else None for t in keep])
    params = [p if p.is_contiguous() else p.contiguous() for l1, l2 in pairs for p in (l1.weight, l1.bias, l2.weight, l2.bias)]
    ws_bytes = lib.psf_mixer_fwd_workspace(N, E, M, h, C, L)
    if ws_bytes < 0:
        raise ValueError("psf_mixer_fwd does not cover this shape")
    ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=dev)
    V0 = torch.empty((B, N, C), dtype=torch.float32, device=dev)
    bufs = [torch.empty_like(V0) for _ in range(min(M, 2))]
    o_tab = (ctypes.c_void_p * M)(*[bufs[m % len(bufs)].data_ptr() for m in range(M)])
    with torch.cuda.device(dev):
        rc = lib.psf_mixer_fwd_in_f32(ctypes.byref(spec), B, N, E, M, _ptrs(params[0::4]), _ptrs(params[1::4]), _ptrs(params[2::4]),
                                      _ptrs(params[3::4]), h, C, L, 1 if use_residual else 0, V0.data_ptr(), o_tab, ws.data_ptr(),
                                      ws_bytes, _lib.stream_ptr(dev))
    _lib.check(rc, "psf_mixer_fwd_in_f32")
    return bufs[(M - 1) % len(bufs)]


def _sizes(x: torch.Tensor, g: nn.Module, fs: Sequence[nn.Module]):
    """(N, E, M, h table, C, L) for ``data`` given as a tensor, or None."""
    if x.dim() != 3:
        return None
    sz = _block_sizes(x.shape[-1], g, fs)
    return None if sz is None else (x.shape[1], x.shape[-1], *sz)


def eligible(x: torch.Tensor, g: nn.Module, fs: Sequence[nn.Module]) -> bool:
    return eligible_recipe(Recipe.data(x), g, fs) and not x.requires_grad


def covered(x: torch.Tensor, g: nn.Module, fs: Sequence[nn.Module]) -> bool:
    """The shape is inside the fused path's limits (whatever ``route`` says about using it)."""
    sz = _sizes(x, g, fs)
    return sz is not None and x.is_cuda and x.dtype == torch.float32 and _lib.load().psf_mixer_fwd_workspace(*sz) >= 0


def mixer_forward(x: torch.Tensor, g: nn.Module, fs: Sequence[nn.Module], use_residual: bool) -> torch.Tensor:
    """V_M [B, N, C] from ``data`` [B, N, E]. Caller checks ``eligible`` (or ``covered``) first."""
    return mixer_forward_in(Recipe.data(x), g, fs, use_residual)
