"""CPU: property tests of the oracle (hypothesis) — size-independent identities the operator must satisfy.
They guard the checker itself: linearity, the adjoint identity that ties the two gradient formulas of
spmul_cuda.cu:75-111 to its forward formula (:20-27), shift equivariance of the circulant pattern, and autograd
of the torch port (float64 gradcheck).
"""
import numpy as np
import torch
from hypothesis import given, settings, strategies as st

from oracle import chord_oracle as oc

shapes = st.tuples(st.integers(1, 3), st.integers(1, 70), st.integers(1, 9), st.integers(1, 6))


def _rand(shape, seed):
    return np.random.default_rng(seed).standard_normal(shape)


@settings(max_examples=40, deadline=None)
@given(shapes, st.integers(0, 2 ** 16))
def test_forward_is_linear_in_both_operands(shape, seed):
    B, N, L, C = shape
    W1, W2, V1, V2 = _rand((B, N, L), seed), _rand((B, N, L), seed + 1), _rand((B, N, C), seed + 2), _rand((B, N, C), seed + 3)
    f = oc.spmul_fwd
    assert np.allclose(f(W1 + 2 * W2, V1), f(W1, V1) + 2 * f(W2, V1), rtol=1e-10, atol=1e-10)
    assert np.allclose(f(W1, V1 - 3 * V2), f(W1, V1) - 3 * f(W1, V2), rtol=1e-10, atol=1e-10)


@settings(max_examples=40, deadline=None)
@given(shapes, st.integers(0, 2 ** 16))
def test_adjoint_identities(shape, seed):
    """<dZ, W.V> = <dV, V> = <dW, W> for the bilinear op: ties bwd_dv / bwd_df to fwd."""
    B, N, L, C = shape
    W, V, dZ = _rand((B, N, L), seed), _rand((B, N, C), seed + 1), _rand((B, N, C), seed + 2)
    Z = oc.spmul_fwd(W, V)
    dW, dV = oc.spmul_bwd(dZ, W, V)
    lhs = float((dZ * Z).sum())
    assert np.isclose(lhs, float((dV * V).sum()), rtol=1e-9, atol=1e-9)
    assert np.isclose(lhs, float((dW * W).sum()), rtol=1e-9, atol=1e-9)


@settings(max_examples=30, deadline=None)
@given(shapes, st.integers(0, 2 ** 16), st.integers(0, 100))
def test_shift_equivariance(shape, seed, shift):
    """The pattern is circulant: rolling W and V along the sequence rolls the output."""
    B, N, L, C = shape
    W, V = _rand((B, N, L), seed), _rand((B, N, C), seed + 1)
    out = oc.spmul_fwd(W, V)
    rolled = oc.spmul_fwd(np.roll(W, shift, 1), np.roll(V, shift, 1))
    assert np.array_equal(np.roll(out, shift, 1), rolled)


@settings(max_examples=25, deadline=None)
@given(st.integers(1, 200), st.integers(1, 12))
def test_index_list_structure(n_vec, n_link):
    """rows[i*L+k] = i; cols[i*L] = i; cols[i*L+k] = (i + 2^(k-1)) mod n — psf.py:7-32 in closed form."""
    rows, cols = oc.chord_indices(n_vec, n_link)
    r, c = rows.reshape(n_vec, n_link), cols.reshape(n_vec, n_link)
    i = np.arange(n_vec)
    assert np.array_equal(r, np.repeat(i[:, None], n_link, 1))
    assert np.array_equal(c[:, 0], i)
    for k in range(1, n_link):
        assert np.array_equal(c[:, k], (i + pow(2, k - 1, n_vec)) % n_vec)


def test_torch_port_gradcheck_float64():
    N, L, C, B = 12, 4, 3, 2
    rows, cols = oc.chord_indices(N, L)
    idx = torch.from_numpy(np.stack([rows, cols]))
    g = torch.Generator().manual_seed(0)
    W = torch.randn(B, N * L, dtype=torch.float64, generator=g, requires_grad=True)
    V = torch.randn(B, N, C, dtype=torch.float64, generator=g, requires_grad=True)
    assert torch.autograd.gradcheck(lambda w, v: oc.torch_spmm_port(idx, w, N, N, v), (W, V), eps=1e-6, atol=1e-8)
    # and the port's autograd agrees with the oracle's explicit gradient formulas
    dZ = torch.randn(B, N, C, dtype=torch.float64, generator=g)
    oc.torch_spmm_port(idx, W, N, N, V).backward(dZ)
    dF, dV = oc.spmul_bwd(dZ.numpy(), W.detach().numpy().reshape(B, N, L), V.detach().numpy())
    assert np.allclose(W.grad.numpy().reshape(B, N, L), dF, rtol=1e-12, atol=1e-12)
    assert np.allclose(V.grad.numpy(), dV, rtol=1e-12, atol=1e-12)
