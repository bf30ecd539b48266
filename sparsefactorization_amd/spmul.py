"""Drop-in for the reference's optional native module ``spmul`` (spmul/spmul.py:1-31).

``SparseMultiply.apply(F, V, offsets, n_block, n_thread_vec, n_thread_dim, n_thread_link)`` keeps its
signature and its 7-slot gradient tuple ``(dJdF, dJdV, None x5)`` (spmul/spmul.py:15,31) but runs the gfx950
kernels of libpsf_chord.so. The four launch-shape arguments of the CUDA version (a grid of ``n_block``
workgroups of ``n_thread_vec`` x ``n_thread_dim`` threads, spmul_cuda.cu:44-47) are accepted and ignored:
tile sizes here are chosen per shape for 64-wide wavefronts and the 8 XCDs.
"""
from __future__ import annotations

import torch
from torch.autograd.function import once_differentiable

from .chord import _norm_offsets, spmm_backward_raw, spmm_forward_raw


def get_offsets(n_link_all: int) -> torch.Tensor:
    """[0, 1, 2, 4, ..., 2**(n_link_all-2)] as an int64 tensor (spmul/spmul.py:8-9), on the GPU when there is one."""
    off = torch.tensor([0] + [2 ** k for k in range(n_link_all - 1)], dtype=torch.int64)
    return off.cuda() if torch.cuda.is_available() else off


class SparseMultiply(torch.autograd.Function):
    """Z[i,p,:] = sum_k F[i,p,k] * V[i,(p+offsets[k]) % n_vec,:]  (spmul_cuda.cu:24) and its two gradients."""

    @staticmethod
    def forward(ctx, F, V, offsets=None, n_block=16, n_thread_vec=64, n_thread_dim=16, n_thread_link=16):
        off = _norm_offsets(offsets)  # None: chord pattern == get_offsets(F.shape[-1])
        Z, Fc, Vc, geom = spmm_forward_raw(F, V, None, off)
        ctx.save_for_backward(Fc, Vc)
        ctx.off, ctx.geom, ctx.v_shape = off, geom, V.shape
        return Z

    @staticmethod
    @once_differentiable
    def backward(ctx, dJdZ):
        F, V = ctx.saved_tensors
        dJdF, dJdV = spmm_backward_raw(dJdZ, F, V, ctx.geom, ctx.off, ctx.needs_input_grad[0],
                                       ctx.needs_input_grad[1], ctx.v_shape)
        return dJdF, dJdV, None, None, None, None, None
