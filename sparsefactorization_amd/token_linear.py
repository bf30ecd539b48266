"""``TokenLinear`` — ``nn.Linear`` for the token-wise MLPs that produce the chain's operands
(``MLPBlock``, SyntheticExperiments/psf.py:35-60: ``W_m = fs[m](data)``, ``V = g(data)``).

Same parameters, same state_dict keys, same forward (``addmm`` on rocBLAS/hipBLASLt). Only the weight and bias
gradients differ: for T = B*N tokens and layer widths of 2..128 they are a reduction over ~10^6 rows into a tile
of a few hundred numbers, for which library GEMMs take ~1 ms per layer (84 % of a training step at
Order/Adding N = 16384 — profiles/r01_train_step_profile.log). ``psf_linear_wgrad_f32`` (csrc/linear_wgrad.hip)
streams X and dY once through the f32 matrix core instead. Layers outside its range (wide ListOps layers,
fp64, CPU tensors) use the stock autograd formulas.
"""
from __future__ import annotations

import torch
from torch import nn

from . import _lib

MAX_WIDTH = 128   # kernel limit on in_features / out_features
MIN_TOKENS = 4096  # below this the stock path is fine


def wgrad_supported(x2d: torch.Tensor, out_features: int) -> bool:
    return (x2d.is_cuda and x2d.dtype == torch.float32 and x2d.shape[0] >= MIN_TOKENS
            and x2d.shape[1] <= MAX_WIDTH and out_features <= MAX_WIDTH)


def linear_wgrad(x2d: torch.Tensor, dy2d: torch.Tensor, need_bias: bool = True):
    """(dWeight [n, m], dBias [n] or None) for X [T, m], dY [T, n] on the HIP kernel."""
    T, m = x2d.shape
    n = dy2d.shape[1]
    lib = _lib.load()
    ws_bytes = lib.psf_linear_wgrad_workspace(T, m, n)
    if ws_bytes < 0:
        raise ValueError(f"psf_linear_wgrad does not support T={T}, m={m}, n={n}")
    # column slices of a wider row-major array (unit inner stride) are taken as they are, with their row stride
    if x2d.stride(1) != 1 or x2d.stride(0) < m:
        x2d = x2d.contiguous()
    if dy2d.stride(1) != 1 or dy2d.stride(0) < n:
        dy2d = dy2d.contiguous()
    ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=x2d.device)
    dW = torch.empty((n, m), dtype=torch.float32, device=x2d.device)
    db = torch.empty(n, dtype=torch.float32, device=x2d.device) if need_bias else None
    with torch.cuda.device(x2d.device):
        rc = lib.psf_linear_wgrad_strided_f32(x2d.data_ptr(), x2d.stride(0), dy2d.data_ptr(), dy2d.stride(0), T, m, n,
                                              dW.data_ptr(), db.data_ptr() if db is not None else None, ws.data_ptr(),
                                              ws_bytes, _lib.stream_ptr(x2d.device))
    _lib.check(rc, "psf_linear_wgrad_strided_f32")
    return dW, db


def affine_rows_supported(x: torch.Tensor, weight: torch.Tensor, bias) -> bool:
    """``x @ weight.T + bias`` with at most three inputs per position on the GPU in fp32: psf_affine_rows_f32 (csrc/embed.hip)
    writes the rows in one pass (init_linear of the synthetic PSFNet, SyntheticExperiments/psf.py:153-154: a library GEMM
    takes 80 us for that K = 2 product at 1 M positions, the pass 25)."""
    return (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.dim() >= 2 and 1 <= x.shape[-1] <= 3
            and weight.dim() == 2 and weight.shape[1] == x.shape[-1] and weight.shape[0] % 4 == 0 and 4 <= weight.shape[0] <= 1024
            and (bias is None or bias.dtype == torch.float32))


def affine_rows(x: torch.Tensor, weight: torch.Tensor, bias) -> torch.Tensor:
    """The rows themselves (no autograd). Products summed in input order with fused adds, then one rounded add of the bias —
    the arithmetic of the mixer kernels' affine recipe, bit for bit."""
    K, E = x.shape[-1], weight.shape[0]
    x2 = x.detach().reshape(-1, K).contiguous()
    w = weight.detach().contiguous()
    b = bias.detach().contiguous() if bias is not None else None
    out = torch.empty((x2.shape[0], E), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = _lib.load().psf_affine_rows_f32(x2.data_ptr(), w.data_ptr(), b.data_ptr() if b is not None else None, out.data_ptr(),
                                             x2.shape[0], K, E, _lib.stream_ptr(x.device))
    _lib.check(rc, "psf_affine_rows_f32")
    return out.reshape(*x.shape[:-1], E)


class _TokenLinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        if affine_rows_supported(x, weight, bias):
            return affine_rows(x, weight, bias)
        return torch.nn.functional.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        dx = dy.matmul(weight) if need_x else None
        dW = db = None
        if need_w or need_b:
            dW, db = linear_wgrad(x.reshape(-1, x.shape[-1]), dy.reshape(-1, dy.shape[-1]), need_b)
            if not need_w:
                dW = None
        return dx, dW, db


EMB_MAX_VOCAB, EMB_MAX_WIDTH = 512, 4096  # limits of psf_embed_tokens_bwd_f32


def embedding_wgrad(idx: torch.Tensor, dout: torch.Tensor, vocab: int, padding_idx) -> torch.Tensor:
    """d(table) [vocab, E] of ``table[idx]`` given d(out) [..., E]: ``psf_embed_tokens_bwd_f32`` (csrc/embed.hip —
    per-slice LDS tables with single-owner accumulation, fixed-order reduction: deterministic, no host read-back, so
    a training step stays capturable in a HIP graph). Vocabularies beyond its LDS table use PyTorch's kernel — in
    eager mode only: under stream capture that case raises (the aten operator is not capturable)."""
    E = dout.shape[-1]
    d2 = dout.reshape(-1, E).contiguous()
    T = d2.shape[0]
    if vocab > EMB_MAX_VOCAB or E > EMB_MAX_WIDTH or dout.dtype != torch.float32 or T == 0:
        if dout.is_cuda and torch.cuda.is_current_stream_capturing():
            # aten::embedding_dense_backward sizes a rocprim partition from a host read-back: not capturable, and a
            # replay of such a capture faulted the GPU (profiles/r01_graph_step_lab.log). Refuse instead.
            raise RuntimeError(
                f"embedding gradient for a {vocab} x {E} {dout.dtype} table cannot be captured in a HIP graph: the "
                f"capturable kernel psf_embed_tokens_bwd_f32 is limited to vocab <= {EMB_MAX_VOCAB}, width <= "
                f"{EMB_MAX_WIDTH}, float32, and the fallback aten::embedding_dense_backward reads sizes back to the host")
        pad = -1 if padding_idx is None else padding_idx
        return torch.ops.aten.embedding_dense_backward(dout.contiguous(), idx, vocab, pad, False)
    lib = _lib.load()
    idx_c = idx.reshape(-1).contiguous()
    ws_bytes = lib.psf_embed_tokens_bwd_workspace(T, vocab, E)
    ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=d2.device)
    dW = torch.empty((vocab, E), dtype=torch.float32, device=d2.device)
    with torch.cuda.device(d2.device):
        rc = lib.psf_embed_tokens_bwd_f32(idx_c.data_ptr(), d2.data_ptr(), T, vocab, E, dW.data_ptr(), ws.data_ptr(),
                                          ws_bytes, _lib.stream_ptr(d2.device))
    _lib.check(rc, "psf_embed_tokens_bwd_f32")
    if padding_idx is not None and padding_idx >= 0:
        dW[padding_idx].zero_()
    return dW


class _TokenEmbeddingFn(torch.autograd.Function):
    """weight[idx] whose weight gradient runs on ``embedding_wgrad`` instead of PyTorch's sort-and-scatter (4.5 ms
    per step for the 6-token vocabulary of Temporal Order at T = 655 360)."""

    @staticmethod
    def forward(ctx, idx, weight, padding_idx):
        ctx.save_for_backward(idx)
        ctx.vocab, ctx.padding_idx = weight.shape[0], padding_idx
        return torch.nn.functional.embedding(idx, weight, padding_idx)

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        return None, embedding_wgrad(idx, dout, ctx.vocab, ctx.padding_idx), None


class _EmbedTokensFn(torch.autograd.Function):
    """out = weight[idx] (+ pos broadcast over the batch) in one pass (csrc/embed.hip). Gradients: the table's by
    ``embedding_wgrad``; pos's is the sum of the output gradient over the batch."""

    @staticmethod
    def forward(ctx, idx, weight, pos, padding_idx):
        lib = _lib.load()
        V, E = weight.shape
        idx_c = idx.contiguous()
        T = idx_c.numel()
        N = pos.shape[0] if pos is not None else 1
        out = torch.empty(*idx.shape, E, dtype=torch.float32, device=idx.device)
        w, p = weight.detach().contiguous(), (pos.detach().contiguous() if pos is not None else None)
        with torch.cuda.device(idx.device):
            rc = lib.psf_embed_tokens_f32(idx_c.data_ptr(), w.data_ptr(), p.data_ptr() if p is not None else None,
                                          out.data_ptr(), T, N, V, E, _lib.stream_ptr(idx.device))
        _lib.check(rc, "psf_embed_tokens_f32")
        ctx.save_for_backward(idx_c)
        ctx.vocab, ctx.padding_idx, ctx.has_pos, ctx.n_pos = V, padding_idx, pos is not None, N
        return out

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        E = dout.shape[-1]
        dW = dpos = None
        if ctx.needs_input_grad[1]:
            dW = embedding_wgrad(idx, dout, ctx.vocab, ctx.padding_idx)
        if ctx.has_pos and ctx.needs_input_grad[2]:
            dpos = dout.reshape(-1, ctx.n_pos, E).sum(0)
        return None, dW, dpos, None


def embed_tokens(idx: torch.Tensor, embedding: nn.Embedding, pos: torch.Tensor = None) -> torch.Tensor:
    """``embedding(idx) + pos.unsqueeze(0)`` (``pos`` [N, E] with N = idx.shape[-1], or None) — the opening lines of
    every PSFNet.forward. One fused pass on the GPU for plain fp32 embeddings; anything else (CPU, max_norm, sparse
    gradients, E not a multiple of 4) takes the stock modules."""
    w = embedding.weight
    if (idx.is_cuda and idx.dtype == torch.int64 and w.dtype == torch.float32 and w.shape[1] % 4 == 0
            and embedding.max_norm is None and not embedding.scale_grad_by_freq and not embedding.sparse
            and (pos is None or (pos.dtype == torch.float32 and pos.dim() == 2 and pos.shape[0] == idx.shape[-1]
                                 and pos.shape[1] == w.shape[1]))):
        return _EmbedTokensFn.apply(idx, w, pos, embedding.padding_idx)
    out = embedding(idx)
    return out if pos is None else out + pos.unsqueeze(0)


class TokenEmbedding(nn.Embedding):
    """Drop-in ``nn.Embedding`` (same parameters / state_dict / forward values) for small vocabularies looked up
    at ~1e6 positions; larger vocabularies, CPU tensors and exotic options use the stock path."""

    def forward(self, idx: torch.Tensor) -> torch.Tensor:
        if (torch.is_grad_enabled() and self.weight.requires_grad and idx.is_cuda and self.weight.dtype == torch.float32
                and self.num_embeddings <= EMB_MAX_VOCAB and self.embedding_dim <= EMB_MAX_WIDTH
                and self.max_norm is None and not self.scale_grad_by_freq and not self.sparse):
            return _TokenEmbeddingFn.apply(idx, self.weight, self.padding_idx)
        return super().forward(idx)


class TokenLinear(nn.Linear):
    """Drop-in ``nn.Linear`` (``isinstance(layer, nn.Linear)`` holds; identical parameters and state_dict)."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if torch.is_grad_enabled() and (self.weight.requires_grad or (self.bias is not None and self.bias.requires_grad)) \
                and x.dim() >= 2 and wgrad_supported(x.reshape(-1, x.shape[-1]), self.out_features):
            return _TokenLinearFn.apply(x, self.weight, self.bias)
        if not torch.is_grad_enabled() and affine_rows_supported(x, self.weight, self.bias):
            return affine_rows(x, self.weight, self.bias)
        return super().forward(x)
