"""ctypes/numpy front end of the CPU oracle (oracle/chord_oracle.c) plus a torch-CPU port of the
reference op sequence.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg —
never by sparsefactorization_amd/ (the product path has no CPU fallback).

Each function cites the reference lines it restates; see the header of chord_oracle.c for what pins it.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile chord_oracle.c with the committed Makefile (gcc, -ffp-contract=off)."""
    src_m = max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("chord_oracle.c", "chord_oracle_impl.h", "Makefile"))
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < src_m:
        subprocess.run(["make", "-C", _HERE, "-B", "liboracle.so"], check=True, capture_output=True)
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


_I64P = ctypes.POINTER(ctypes.c_int64)


def _p(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


def _suffix(dtype) -> str:
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return "_f32"
    if dtype == np.float64:
        return "_f64"
    raise TypeError(f"oracle supports float32/float64, got {dtype}")


def _c(a, dtype) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=dtype)


def chord_indices(n_vec: int, n_link: int):
    """get_chord_indices_assym — SyntheticExperiments/psf.py:7-32. Returns (rows, cols), int64 [n_vec*n_link]."""
    rows = np.empty(n_vec * n_link, dtype=np.int64)
    cols = np.empty(n_vec * n_link, dtype=np.int64)
    rc = lib().oracle_chord_indices(ctypes.c_int64(n_vec), ctypes.c_int64(n_link), _p(rows), _p(cols))
    if rc:
        raise ValueError("oracle_chord_indices: invalid arguments")
    return rows, cols


def spmul_offsets(n_link_all: int) -> np.ndarray:
    """get_offsets — spmul/spmul.py:8-9."""
    off = np.empty(n_link_all, dtype=np.int64)
    if lib().oracle_spmul_offsets(ctypes.c_int64(n_link_all), _p(off)):
        raise ValueError("oracle_spmul_offsets: invalid arguments")
    return off


def spmm(index, value, m: int, n: int, matrix) -> np.ndarray:
    """torch_sparse.spmm(index, value, m, n, matrix) restated (see chord_oracle_impl.h).

    index [2, nnz] int64; value [B, nnz]; matrix [B, n, C] or [n, C] (broadcast). Returns [B, m, C].
    """
    index = np.asarray(index)
    value = np.asarray(value)
    dtype = value.dtype
    suf = _suffix(dtype)
    rows = _c(index[0], np.int64)
    cols = _c(index[1], np.int64)
    value = _c(value, dtype)
    matrix = _c(matrix, dtype)
    if value.ndim != 2:
        raise ValueError("value must be [B, nnz]")
    B, nnz = value.shape
    if rows.shape != (nnz,) or cols.shape != (nnz,):
        raise ValueError("index must be [2, nnz]")
    if matrix.ndim == 2:
        bstride, C = 0, matrix.shape[1]
        if matrix.shape[0] != n:
            raise ValueError("matrix must have n rows")
    else:
        if matrix.shape[0] != B or matrix.shape[1] != n:
            raise ValueError("matrix must be [B, n, C]")
        bstride, C = n * matrix.shape[2], matrix.shape[2]
    out = np.empty((B, m, C), dtype=dtype)
    fn = getattr(lib(), "oracle_spmm" + suf)
    rc = fn(_p(rows), _p(cols), ctypes.c_int64(nnz), _p(value), ctypes.c_int64(B), ctypes.c_int64(m),
            ctypes.c_int64(n), _p(matrix), ctypes.c_int64(bstride), ctypes.c_int64(C), _p(out))
    if rc:
        raise ValueError("oracle_spmm: index out of range")
    return out


def _offsets_arg(offsets: Optional[Sequence[int]], L: int) -> np.ndarray:
    if offsets is None:
        return spmul_offsets(L)
    off = _c(offsets, np.int64)
    if off.shape != (L,):
        raise ValueError("offsets must hold L entries")
    return off


def spmul_fwd(F, V, offsets=None) -> np.ndarray:
    """forward_kernel — spmul/spmul_cuda.cu:20-27. F [B,N,L], V [B,N,C] or [N,C]."""
    F = np.asarray(F)
    dtype = F.dtype
    suf = _suffix(dtype)
    F = _c(F, dtype)
    V = _c(V, dtype)
    B, N, L = F.shape
    C = V.shape[-1]
    bstride = 0 if V.ndim == 2 else N * C
    off = _offsets_arg(offsets, L)
    Z = np.empty((B, N, C), dtype=dtype)
    getattr(lib(), "oracle_spmul_fwd" + suf)(_p(F), _p(V), ctypes.c_int64(bstride), _p(off), ctypes.c_int64(B),
                                            ctypes.c_int64(N), ctypes.c_int64(L), ctypes.c_int64(C), _p(Z))
    return Z


def spmul_bwd(dZ, F, V, offsets=None):
    """backward_dJdV_kernel / backward_dJdF_kernel — spmul/spmul_cuda.cu:75-84, 102-111. Returns (dF, dV)."""
    F = np.asarray(F)
    dtype = F.dtype
    suf = _suffix(dtype)
    F = _c(F, dtype)
    V = _c(V, dtype)
    dZ = _c(dZ, dtype)
    B, N, L = F.shape
    C = dZ.shape[-1]
    bstride = 0 if V.ndim == 2 else N * C
    off = _offsets_arg(offsets, L)
    dV = np.empty((B, N, C), dtype=dtype)
    dF = np.empty((B, N, L), dtype=dtype)
    getattr(lib(), "oracle_spmul_bwd_dv" + suf)(_p(dZ), _p(F), _p(off), ctypes.c_int64(B), ctypes.c_int64(N),
                                               ctypes.c_int64(L), ctypes.c_int64(C), _p(dV))
    getattr(lib(), "oracle_spmul_bwd_df" + suf)(_p(dZ), _p(V), ctypes.c_int64(bstride), _p(off),
                                               ctypes.c_int64(B), ctypes.c_int64(N), ctypes.c_int64(L),
                                               ctypes.c_int64(C), _p(dF))
    return dF, dV


def chain(index, W_all, V0, use_residual: bool) -> np.ndarray:
    """Hot loop of PSFNet.forward — SyntheticExperiments/psf.py:167-188.

    W_all [M, B, N, L] (or [M, B, N*L]); V0 [B, N, C]. Returns every step's V: [M, B, N, C].
    """
    W_all = np.asarray(W_all)
    dtype = W_all.dtype
    suf = _suffix(dtype)
    index = np.asarray(index)
    rows = _c(index[0], np.int64)
    cols = _c(index[1], np.int64)
    nnz = rows.shape[0]
    V0 = _c(V0, dtype)
    B, N, C = V0.shape
    M = W_all.shape[0]
    W_all = _c(W_all.reshape(M, B, nnz), dtype)
    steps = np.empty((M, B, N, C), dtype=dtype)
    rc = getattr(lib(), "oracle_chain" + suf)(_p(rows), _p(cols), ctypes.c_int64(nnz), _p(W_all),
                                             ctypes.c_int64(M), _p(V0), ctypes.c_int64(B), ctypes.c_int64(N),
                                             ctypes.c_int64(C), ctypes.c_int(1 if use_residual else 0), _p(steps))
    if rc:
        raise ValueError("oracle_chain: index out of range")
    return steps


# ----------------------------------------------------------------------------------------------------
# torch-CPU port of the reference op sequence: what torch_sparse.spmm executes on CPU tensors.
# Used (a) as the cpu_baseline of bench.py (multi-threaded, `kind: "port"`), (b) as the autograd
# reference for gradients in tests. Differentiable.
# ----------------------------------------------------------------------------------------------------
def torch_spmm_port(index, value, m, n, matrix):
    """index_select(-2, col) -> * value.unsqueeze(-1) -> scatter_add(row, dim=-2, dim_size=m).

    torch-sparse==0.6.11 `spmm` (requirements.txt:146) as called at SyntheticExperiments/psf.py:178-184;
    torch.Tensor.index_add on dim -2 stands in for torch_scatter.scatter_add (same sums, same order on CPU).
    """
    import torch

    row, col = index[0], index[1]
    out = matrix.index_select(-2, col)
    out = out * value.unsqueeze(-1)
    zeros = torch.zeros(out.shape[:-2] + (m, out.shape[-1]), dtype=out.dtype, device=out.device)
    return zeros.index_add(-2, row, out)


def torch_chain_port(index, W_list, V0, use_residual: bool):
    """for m: V = spmm(idx, W_m.reshape(B, N*L), N, N, V); V = V + res_conn — SyntheticExperiments/psf.py:172-188."""
    V = V0
    N = V0.shape[-2]
    for W in W_list:
        V = torch_spmm_port(index, W.reshape(W.size(0), W.size(1) * W.size(2)), N, N, V)
        if use_residual:
            V = V + V0
    return V
