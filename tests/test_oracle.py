"""CPU: pins the oracle (oracle/chord_oracle.c) before anything is allowed to trust it.

 1. index pattern == the reference's get_chord_indices_assym output (golden, generated from /root/reference)
 2. generic COO spmm == the reference's own statement of the operator (spmul_cuda.cu formulas) == dense matmul
 3. chain == what the reference's PSFNet.forward produced here (captured operands of its hot loop)
 4. gradients == autograd through the reference forward
"""
import hashlib

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_inf
from oracle import chord_oracle as oc


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_indices_match_reference_small():
    g = load_golden("chord_indices.npz")
    for n, l in g["full_cases"]:
        rows, cols = oc.chord_indices(int(n), int(l))
        assert np.array_equal(rows, g[f"rows_{n}_{l}"]), (n, l)
        assert np.array_equal(cols, g[f"cols_{n}_{l}"]), (n, l)


def test_indices_match_reference_large_by_hash():
    g = load_golden("chord_indices.npz")
    for line in g["hashed"]:
        n, l, hr, hc = str(line).split(",")
        rows, cols = oc.chord_indices(int(n), int(l))
        assert _sha(rows) == hr and _sha(cols) == hc, (n, l)
        assert np.array_equal(cols[: 4 * int(l)], g[f"colshead_{n}_{l}"])
        assert np.array_equal(cols[-4 * int(l):], g[f"colstail_{n}_{l}"])


def test_duplicate_links_are_kept():
    # N=1024, L=12: 2^10 == 0 (mod N) -> the last link is a second self link
    rows, cols = oc.chord_indices(1024, 12)
    c = cols.reshape(1024, 12)
    assert np.array_equal(c[:, 0], c[:, 11])
    assert len(rows) == 1024 * 12


def _dense_W(W, offsets, N):
    """Dense [B,N,N] matrix with W[b,p,k] added at column (p+off_k) mod N (chord_mask_mat.m:19-37 pattern)."""
    B, _, L = W.shape
    D = np.zeros((B, N, N), dtype=np.float64)
    p = np.arange(N)
    for k in range(L):
        np.add.at(D, (slice(None), p, (p + int(offsets[k])) % N), W[:, :, k].astype(np.float64))
    return D


@pytest.mark.parametrize("N,L,C,B", [(8, 4, 3, 2), (128, 8, 8, 3), (100, 9, 5, 2), (257, 10, 4, 1), (64, 7, 16, 2)])
def test_spmm_equals_spmul_formula_and_dense(N, L, C, B):
    rng = np.random.default_rng(N * 131 + L)
    W = rng.standard_normal((B, N, L), dtype=np.float32)
    V = rng.standard_normal((B, N, C), dtype=np.float32)
    rows, cols = oc.chord_indices(N, L)
    out = oc.spmm(np.stack([rows, cols]), W.reshape(B, N * L), N, N, V)
    # the reference's own operator statement (spmul_cuda.cu:24), same summation order -> bit-exact
    off = oc.spmul_offsets(L)
    z = oc.spmul_fwd(W, V, off)
    assert np.array_equal(out, z)
    # dense matmul in float64
    dense = np.einsum("bpq,bqc->bpc", _dense_W(W, off, N), V.astype(np.float64))
    assert rel_inf(out, dense) < 2e-6


def test_spmm_broadcast_matrix():
    N, L, B = 32, 6, 3
    rng = np.random.default_rng(5)
    W = rng.standard_normal((B, N, L), dtype=np.float32)
    eye = np.eye(N, dtype=np.float32)
    rows, cols = oc.chord_indices(N, L)
    out = oc.spmm(np.stack([rows, cols]), W.reshape(B, N * L), N, N, eye)
    assert rel_inf(out, _dense_W(W, oc.spmul_offsets(L), N)) < 1e-7


def test_spmm_matches_torch_port_bitwise():
    N, L, C, B = 128, 8, 8, 4
    rng = np.random.default_rng(9)
    W = rng.standard_normal((B, N, L), dtype=np.float32)
    V = rng.standard_normal((B, N, C), dtype=np.float32)
    rows, cols = oc.chord_indices(N, L)
    idx = np.stack([rows, cols])
    out = oc.spmm(idx, W.reshape(B, N * L), N, N, V)
    torch.set_num_threads(1)
    t = oc.torch_spmm_port(torch.from_numpy(idx), torch.from_numpy(W.reshape(B, N * L)), N, N, torch.from_numpy(V))
    assert np.array_equal(out, t.numpy())


@pytest.mark.parametrize("fixture,residual", [("psfnet_adding_n128.npz", True), ("psfnet_order_n128.npz", True),
                                              ("lra_pathfinder_ckpt.npz", False)])
def test_chain_matches_reference_forward(fixture, residual):
    g = load_golden(fixture)
    W, V0, Vfin = g["W"], g["V0"], g["Vfin"]
    M, B, N, L = W.shape
    rows, cols = oc.chord_indices(N, L)
    steps = oc.chain(np.stack([rows, cols]), W, V0, residual)
    # the generator ran the reference single-threaded with the same op order: expect (near) bit equality
    assert rel_inf(steps[-1], Vfin) <= 1e-6
    if "chord_indicies" in g.files:
        assert np.array_equal(g["chord_indicies"], np.stack([rows, cols]))


def test_attention_map_matches_reference_restatement():
    g = load_golden("lra_pathfinder_ckpt.npz")
    W = g["W"]
    M, B, N, L = W.shape
    rows, cols = oc.chord_indices(N, L)
    idx = np.stack([rows, cols])
    Wf = np.eye(N, dtype=np.float32)
    for m in range(M):
        Wf = oc.spmm(idx, W[m].reshape(B, N * L), N, N, Wf)
    assert rel_inf(Wf[0, ::16, :], g["Wfinal_rows"]) <= 1e-6
    assert rel_inf(Wf.sum(-1), g["Wfinal_rowsum"]) <= 1e-5


@pytest.mark.parametrize("fixture", ["psfnet_adding_n128.npz", "psfnet_order_n128.npz"])
def test_backward_matches_reference_autograd(fixture):
    g = load_golden(fixture)
    W, V0, gV = g["W"], g["V0"], g["gVfin"]
    M, B, N, L = W.shape
    rows, cols = oc.chord_indices(N, L)
    steps = oc.chain(np.stack([rows, cols]), W, V0, True)
    off = oc.spmul_offsets(L)
    grad = gV.copy()
    res_acc = np.zeros_like(V0)
    dW = np.zeros_like(W)
    for m in range(M - 1, -1, -1):
        x_in = V0 if m == 0 else steps[m - 1]
        res_acc += grad
        dW[m], grad = oc.spmul_bwd(grad, W[m], x_in, off)
    dV0 = grad + res_acc
    assert rel_inf(dW, g["dW"]) <= 1e-5
    assert rel_inf(dV0, g["dV0"]) <= 1e-5


def test_f64_variants():
    N, L, C, B = 64, 7, 4, 2
    rng = np.random.default_rng(3)
    W = rng.standard_normal((B, N, L))
    V = rng.standard_normal((B, N, C))
    rows, cols = oc.chord_indices(N, L)
    out = oc.spmm(np.stack([rows, cols]), W.reshape(B, N * L), N, N, V)
    assert out.dtype == np.float64
    dense = np.einsum("bpq,bqc->bpc", _dense_W(W, oc.spmul_offsets(L), N), V)
    assert rel_inf(out, dense) < 1e-14
