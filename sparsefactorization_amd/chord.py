"""PyTorch-facing operators of the chord-sparse matmul path, backed by libpsf_chord.so (HIP, gfx950).

Drop-ins for the reference's call sites:

* ``spmm(index, value, m, n, matrix)`` — same signature and meaning as ``torch_sparse.spmm`` as it is called
  at SyntheticExperiments/psf.py:178-184, LRA/psf.py:230-236, Genome_Clf/psf.py:220-226,
  attention_block.py:164-170, LRA/attention_maps/pathfinder_inference.py:66-81 and imdb_inference.py:45-59.
* ``chord_spmm(W, V, residual=None)`` — the same step without the index tensor (the pattern is implicit),
  with the residual add of SyntheticExperiments/psf.py:187-188 fused in.
* ``chord_chain(W_list, V0, use_residual)`` — the whole hot loop of PSFNet.forward (psf.py:172-188).

All of them are differentiable (gradients per spmul/spmul_cuda.cu:61-112) and run only on a HIP device:
PyTorch provides device memory, the stream and autograd bookkeeping; every FLOP of the path is in the
hand-written kernels. There is deliberately no CPU or eager-PyTorch fallback.
"""
from __future__ import annotations

import ctypes
import weakref
from typing import List, Optional, Sequence, Tuple

import torch
from torch.autograd.function import once_differentiable

from . import _lib

_SUFFIX = {torch.float32: "_f32", torch.float64: "_f64"}


def _suffix(t: torch.Tensor) -> str:
    try:
        return _SUFFIX[t.dtype]
    except KeyError:
        raise TypeError(f"chord spmm computes in float32 (or float64); got {t.dtype}") from None


def _require_hip(*tensors: torch.Tensor) -> torch.device:
    dev = tensors[0].device
    for t in tensors:
        if not t.is_cuda:
            raise RuntimeError(
                "chord spmm runs only on an MI355X (HIP) device: got a tensor on "
                f"'{t.device}'. Move the module and its inputs to the GPU (net.cuda()); there is no CPU path.")
        if t.device != dev:
            raise RuntimeError(f"all operands must live on one device (got {dev} and {t.device})")
    return dev


def _stream_ptr(dev: torch.device) -> int:
    return _lib.stream_ptr(dev)


def _norm_offsets(offsets) -> Optional[Tuple[int, ...]]:
    if offsets is None:
        return None
    if isinstance(offsets, torch.Tensor):
        offsets = offsets.detach().cpu().tolist()
    return tuple(int(o) for o in offsets)


def _shapes(W: torch.Tensor, V: torch.Tensor):
    if W.dim() != 3:
        raise ValueError(f"W must be [B, N, L], got {tuple(W.shape)}")
    B, N, L = W.shape
    if V.dim() == 2:
        V = V.unsqueeze(0)
    if V.dim() != 3 or V.shape[1] != N:
        raise ValueError(f"V must be [B, N, C] or [N, C] with N={N}, got {tuple(V.shape)}")
    if V.shape[0] == B and not (B == 1):
        stride = N * V.shape[2]
    elif V.shape[0] == 1:
        stride = 0 if B != 1 else N * V.shape[2]
    else:
        raise ValueError(f"batch of V ({V.shape[0]}) must be {B} or 1")
    if L > _lib.MAX_LINKS:
        raise ValueError(f"L={L} exceeds {_lib.MAX_LINKS}")
    return B, N, L, V.shape[2], stride


def _launch_fwd(W, V, res, out, B, N, L, C, stride, offsets):
    dev = _require_hip(W, V, out) if res is None else _require_hip(W, V, out, res)
    fn = getattr(_lib.load(), "psf_chord_spmm_fwd" + _suffix(W))
    off = _lib.offsets_array(offsets)
    with torch.cuda.device(dev):
        rc = fn(W.data_ptr(), V.data_ptr(), res.data_ptr() if res is not None else None, out.data_ptr(),
                B, N, L, C, stride, off, _stream_ptr(dev))
    _lib.check(rc, "psf_chord_spmm_fwd")


def _launch_bwd(dZ, W, V, dW, dV, B, N, L, C, stride, offsets):
    dev = _require_hip(dZ, W, V)
    off = _lib.offsets_array(offsets)
    with torch.cuda.device(dev):
        fn = getattr(_lib.load(), "psf_chord_spmm_bwd" + _suffix(dZ))
        rc = fn(dZ.data_ptr(), W.data_ptr(), V.data_ptr(), dW.data_ptr() if dW is not None else None,
                dV.data_ptr() if dV is not None else None, B, N, L, C, stride, off, _stream_ptr(dev))
    _lib.check(rc, "psf_chord_spmm_bwd")


def spmm_forward_raw(W, V, residual, offsets):
    """Launch one forward step (no autograd). Returns (out, W_contig, V_contig, geom)."""
    if W.dtype != V.dtype:
        raise TypeError(f"W ({W.dtype}) and V ({V.dtype}) must share a dtype")
    W = W.contiguous()
    Vc = V.contiguous()
    B, N, L, C, stride = _shapes(W, Vc)
    res = None
    if residual is not None:
        if residual.shape != (B, N, C) or residual.dtype != W.dtype:
            raise ValueError(f"residual must be [{B}, {N}, {C}] {W.dtype}, got {tuple(residual.shape)} {residual.dtype}")
        res = residual.contiguous()
    out = torch.empty((B, N, C), dtype=W.dtype, device=W.device)
    _launch_fwd(W, Vc, res, out, B, N, L, C, stride, offsets)
    return out, W, Vc, (B, N, L, C, stride)


def spmm_backward_raw(dZ, W, V, geom, offsets, need_dW, need_dV, v_shape):
    """Launch the backward step (no autograd). Returns (dW, dV) with dV reduced/reshaped to ``v_shape``."""
    B, N, L, C, stride = geom
    dZ = dZ.contiguous()
    dW = torch.empty_like(W) if need_dW else None
    dV = torch.empty((B, N, C), dtype=W.dtype, device=W.device) if need_dV else None
    if need_dW or need_dV:
        _launch_bwd(dZ, W, V, dW, dV, B, N, L, C, stride, offsets)
    if dV is not None:
        if stride == 0 and B != 1:
            dV = dV.sum(dim=0)
        dV = dV.reshape(v_shape)
    return dW, dV


class _ChordSpmm(torch.autograd.Function):
    """One step; autograd boundary in the shape of spmul/spmul.py:12-31 (SparseMultiply)."""

    @staticmethod
    def forward(ctx, W, V, residual, offsets):
        out, Wc, Vc, geom = spmm_forward_raw(W, V, residual, offsets)
        ctx.save_for_backward(Wc, Vc)
        ctx.offsets, ctx.geom, ctx.v_shape = offsets, geom, V.shape
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dZ):
        W, V = ctx.saved_tensors
        need_dW, need_dV, need_res = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        dW, dV = spmm_backward_raw(dZ, W, V, ctx.geom, ctx.offsets, need_dW, need_dV, ctx.v_shape)
        return dW, dV, (dZ if need_res else None), None


def chord_spmm(W: torch.Tensor, V: torch.Tensor, residual: Optional[torch.Tensor] = None,
               offsets: Optional[Sequence[int]] = None) -> torch.Tensor:
    """``out[b,p,:] = sum_k W[b,p,k] * V[b,(p+off_k) mod N,:] (+ residual[b,p,:])``.

    W [B,N,L]; V [B,N,C] or [N,C] (broadcast over the batch, as the unbatched ``eye(N)`` of
    pathfinder_inference.py:57); ``offsets`` defaults to the chord pattern 0,1,2,4,... of
    get_chord_indices_assym (SyntheticExperiments/psf.py:7-32).
    """
    return _ChordSpmm.apply(W, V, residual, _norm_offsets(offsets))


# ----------------------------------------------------------------------------------------------------
# torch_sparse.spmm drop-in
# ----------------------------------------------------------------------------------------------------
_index_cache: dict = {}


def offsets_from_index(index: torch.Tensor, m: int) -> Tuple[int, ...]:
    """Recover the link offsets from a COO index list and verify it is the affine (circulant) pattern the
    kernels assume: rows[i*L+k] == i and cols[i*L+k] == (i + off_k) mod m for every i (psf.py:7-32)."""
    # cache per tensor OBJECT (weak reference) and in-place version: a freed tensor's address can be reused by a
    # different index list, so data_ptr() is not an identity
    key = id(index)
    hit = _index_cache.get(key)
    if hit is not None and hit[0]() is index and hit[1] == (index._version, m):
        return hit[2]
    if index.dim() != 2 or index.shape[0] != 2:
        raise ValueError(f"index must be [2, nnz], got {tuple(index.shape)}")
    nnz = index.shape[1]
    if m < 1 or nnz % m != 0 or nnz == 0:
        raise ValueError(f"index with nnz={nnz} is not a fixed number of links per row for m={m}")
    L = nnz // m
    rows = index[0].reshape(m, L)
    cols = index[1].reshape(m, L)
    ar = torch.arange(m, device=index.device, dtype=index.dtype).unsqueeze(1)
    if not bool((rows == ar).all()):
        raise ValueError("index rows are not [i]*L for i in range(m): not a chord-style pattern")
    off = cols[0].tolist()  # row 0: cols == offsets mod m
    want = (ar + cols[0].unsqueeze(0)) % m
    if not bool((cols == want).all()):
        raise ValueError("index cols are not (i + off_k) mod m: not a chord-style (circulant) pattern; "
                         "this library implements only the structured pattern of get_chord_indices_assym")
    out = tuple(int(o) for o in off)
    if len(_index_cache) > 64:
        _index_cache.clear()
    _index_cache[key] = (weakref.ref(index), (index._version, m), out)
    return out


def spmm(index: torch.Tensor, value: torch.Tensor, m: int, n: int, matrix: torch.Tensor) -> torch.Tensor:
    """Drop-in for ``torch_sparse.spmm`` on chord-structured index lists.

    ``index`` [2, m*L] is the tensor built from get_chord_indices_assym; ``value`` [B, m*L] (or [m*L]);
    ``matrix`` [B, n, C] or [n, C]. Requires m == n. Returns [B, m, C] ([m, C] when nothing is batched).
    The index tensor is only inspected once (to recover and verify the offsets); it is never read on the
    hot path.
    """
    if m != n:
        raise ValueError(f"chord spmm is square: m={m} != n={n}")
    offsets = offsets_from_index(index, m)
    L = len(offsets)
    unbatched = value.dim() == 1
    if unbatched:
        value = value.unsqueeze(0)
    if value.dim() != 2 or value.shape[1] != m * L:
        raise ValueError(f"value must be [B, {m * L}], got {tuple(value.shape)}")
    W = value.reshape(value.shape[0], m, L)
    out = _ChordSpmm.apply(W, matrix, None, offsets)
    if unbatched and matrix.dim() == 2:
        out = out.squeeze(0)
    return out


# ----------------------------------------------------------------------------------------------------
# whole chain
# ----------------------------------------------------------------------------------------------------
def _sum_tensors(terms: Sequence[torch.Tensor]) -> torch.Tensor:
    """((t0 + t1) + t2) + ... in one pass (csrc/sum_tensors.hip); all terms fp32, same shape, on one HIP device."""
    dev = _require_hip(*terms)
    # the kernel reads 16-byte vectors: a contiguous VIEW at a storage offset (an upstream gradient) may be misaligned
    terms = [t.contiguous() for t in terms]
    terms = [t if t.data_ptr() % 16 == 0 else t.clone() for t in terms]
    out = torch.empty_like(terms[0])
    tab = (ctypes.c_void_p * len(terms))(*[t.data_ptr() for t in terms])
    with torch.cuda.device(dev):
        rc = _lib.load().psf_sum_tensors_f32(tab, len(terms), out.numel(), out.data_ptr(), _stream_ptr(dev))
    _lib.check(rc, "psf_sum_tensors_f32")
    return out


def _chain_forward_raw(V0, use_residual, offsets, Ws, keep_all):
    """The M launches of the chain (one library call). keep_all: every step's result in its own buffer (they are the saved
    inputs of the backward steps); otherwise two buffers take turns. Returns (V0 contiguous, [W_m contiguous], outs, geometry)."""
    M = len(Ws)
    Ws = [w if w.is_contiguous() else w.contiguous() for w in Ws]
    V0c = V0 if V0.is_contiguous() else V0.contiguous()
    B, N, L, C, stride0 = _shapes(Ws[0], V0c)
    for w in Ws:
        if w.shape != Ws[0].shape or w.dtype != V0c.dtype:
            raise ValueError("every W_m must be [B, N, L] with V0's dtype")
    if use_residual and stride0 == 0 and B != 1:
        raise ValueError("a broadcast V0 cannot be the residual")
    dev = _require_hip(V0c, *Ws)
    nbuf = M if keep_all else min(M, 2)
    bufs = [torch.empty((B, N, C), dtype=V0c.dtype, device=dev) for _ in range(nbuf)]
    outs = [bufs[m % nbuf] for m in range(M)]
    lib = _lib.load()
    w_tab = (ctypes.c_void_p * M)(*[w.data_ptr() for w in Ws])
    o_tab = (ctypes.c_void_p * M)(*[o.data_ptr() for o in outs])
    with torch.cuda.device(dev):
        fn = getattr(lib, "psf_chord_chain_fwd" + _suffix(V0c))
        rc = fn(w_tab, V0c.data_ptr(), o_tab, M, 1 if use_residual else 0, B, N, L, C, stride0,
                _lib.offsets_array(offsets), _stream_ptr(dev))
    _lib.check(rc, "psf_chord_chain_fwd")
    return V0c, Ws, outs, (B, N, L, C, stride0)


class _ChordChain(torch.autograd.Function):
    """X_0 = V0; X_{m+1} = W_m (.) X_m (+ V0) — SyntheticExperiments/psf.py:167-188 as one autograd node."""

    @staticmethod
    def forward(ctx, V0, use_residual, offsets, *Ws):
        if not Ws:
            return V0
        keep_all = any(ctx.needs_input_grad)
        V0c, Ws, outs, geom = _chain_forward_raw(V0, use_residual, offsets, Ws, keep_all)
        if keep_all:
            ctx.save_for_backward(V0c, *Ws, *outs[:-1])
        ctx.M, ctx.geom, ctx.offsets, ctx.use_residual, ctx.v_shape = len(Ws), geom, offsets, use_residual, V0.shape
        return outs[-1]

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        M = ctx.M
        B, N, L, C, stride0 = ctx.geom
        saved = ctx.saved_tensors
        V0, Ws, steps = saved[0], saved[1:1 + M], saved[1 + M:2 * M]
        need_v0 = ctx.needs_input_grad[0]
        need_w = ctx.needs_input_grad[3:]
        g = g.contiguous()
        dWs: List[Optional[torch.Tensor]] = [None] * M
        res_acc = None
        # the residual sends every step's output gradient to V0 too: dV0 = dX_0 + sum_m dX_m. The dX_m exist anyway
        # (each is the next step's dZ): they are kept and summed ONCE at the end (psf_sum_tensors_f32) instead of one
        # accumulate kernel per step (14 x 9.5 us per Temporal-Order training step)
        sum_once = ctx.use_residual and need_v0 and g.dtype == torch.float32 and (B * N * C) % 4 == 0 and M + 1 <= 32
        res_terms: List[torch.Tensor] = []
        # everything that is the same for the M launches is looked up once: the small LRA models are bound by the host, and
        # the device guard, the stream object and the symbol look-up of _launch_bwd are most of what a launch costs it
        dev = _require_hip(g, V0, *Ws)
        lib = _lib.load()
        off = _lib.offsets_array(ctx.offsets)
        # the whole backward chain in ONE library call (psf_chord_chain_bwd_f32): one launch with the running gradient resident in
        # LDS for short sequences of narrow rows (the synthetic tasks up to N = 1024; csrc/bwd_chain_lds.h), the M per-step launches
        # and the one-pass residual sum issued by the library otherwise — same kernels, same order and same bits as the loop
        # below, without M trips through ctypes and 2 M allocations (the small LRA models are bound by the host)
        if (g.dtype == torch.float32 and stride0 == N * C and all(need_w) and M >= 1
                and not any(t.data_ptr() % 16 for t in (g, V0, *steps))):  # (views at odd offsets: the loop's kernels take them)
            one_launch = bool(lib.psf_chord_chain_bwd_supported(N, L, C, M))
            dWb = torch.empty((M, B, N, L), dtype=g.dtype, device=dev)
            dWs = list(dWb.unbind(0))
            dV0 = torch.empty((B, N, C), dtype=g.dtype, device=dev)
            w_tab = (ctypes.c_void_p * M)(*[w.data_ptr() for w in Ws])
            x_tab = (ctypes.c_void_p * M)(V0.data_ptr(), *[s.data_ptr() for s in steps])
            dw_tab = (ctypes.c_void_p * M)(*[d.data_ptr() for d in dWs])
            dx_tab = None
            if not one_launch:
                dXb = torch.empty((M, B, N, C), dtype=g.dtype, device=dev)
                dx_tab = (ctypes.c_void_p * M)(*[dXb[m].data_ptr() for m in range(M)])
            with torch.cuda.device(dev):
                rc = lib.psf_chord_chain_bwd_f32(g.data_ptr(), w_tab, V0.data_ptr(), x_tab, dw_tab, dV0.data_ptr(), dx_tab, M,
                                                 1 if ctx.use_residual else 0, B, N, L, C, off, _stream_ptr(dev))
            if rc != _lib.PSF_E_UNSUPPORTED:
                _lib.check(rc, "psf_chord_chain_bwd_f32")
                return (dV0.reshape(ctx.v_shape) if need_v0 else None, None, None, *dWs)
            dWs = [None] * M
        fn = getattr(lib, "psf_chord_spmm_bwd" + _suffix(g))
        rc = 0
        with torch.cuda.device(dev):
            stream = _stream_ptr(dev)
            for m in range(M - 1, -1, -1):
                x_in = V0 if m == 0 else steps[m - 1]
                stride = stride0 if m == 0 else N * C
                if sum_once:
                    res_terms.append(g)
                elif ctx.use_residual and need_v0:
                    res_acc = g.clone() if res_acc is None else res_acc.add_(g)
                want_dx = m > 0 or need_v0
                dW = torch.empty_like(Ws[m]) if need_w[m] else None
                dX = torch.empty((B, N, C), dtype=g.dtype, device=dev) if want_dx else None
                if dW is not None or dX is not None:
                    rc = fn(g.data_ptr(), Ws[m].data_ptr(), x_in.data_ptr(), dW.data_ptr() if dW is not None else None,
                            dX.data_ptr() if dX is not None else None, B, N, L, C, stride, off, stream)
                    if rc:
                        break
                dWs[m] = dW
                g = dX
        _lib.check(rc, "psf_chord_spmm_bwd")
        dV0 = None
        if need_v0:
            dV0 = g
            if stride0 == 0 and B != 1:
                dV0 = dV0.sum(dim=0)
            if res_terms:
                dV0 = _sum_tensors(res_terms + [dV0])  # ((g_M + g_{M-1}) + ... + g_1) + dX_0: the order of the loop above
            elif res_acc is not None:
                dV0 = dV0 + res_acc
            dV0 = dV0.reshape(ctx.v_shape)
        return (dV0, None, None, *dWs)


def chord_chain(W_list: Sequence[torch.Tensor], V0: torch.Tensor, use_residual: bool = False,
                offsets: Optional[Sequence[int]] = None) -> torch.Tensor:
    """Run ``for m: V = W_m (.) V (+ V0)`` for all factors in one call (M dependent HIP launches on the
    current stream, no Python or allocator work between them). Returns the final V [B, N, C]."""
    if not len(W_list):
        return V0
    if not torch.is_grad_enabled() or not (V0.requires_grad or any(w.requires_grad for w in W_list)):
        # nothing to differentiate: no autograd node (its bookkeeping is most of the host time of a short chain)
        return _chain_forward_raw(V0, bool(use_residual), _norm_offsets(offsets), W_list, False)[2][-1]
    return _ChordChain.apply(V0, bool(use_residual), _norm_offsets(offsets), *W_list)


def get_chord_indices_assym(n_vec: int, n_link: int):
    """Same return value as the reference helper (SyntheticExperiments/psf.py:7-32): two Python lists
    ``rows``/``cols`` of length n_vec*n_link, computed by the library's host routine psf_chord_indices."""
    size = n_vec * n_link
    rows = (ctypes.c_int64 * size)()
    cols = (ctypes.c_int64 * size)()
    _lib.check(_lib.load().psf_chord_indices(n_vec, n_link, rows, cols), "psf_chord_indices")
    return list(rows), list(cols)
