"""`sparsefactorization_amd.lazy.spmm`: the reference's unmodified hot loop (one spmm + one residual add per factor,
SyntheticExperiments/psf.py:172-188) recorded and run as ONE chord_chain call.

CPU: the recording logic, with the chain evaluators replaced by the oracle's torch port of the reference op sequence.
GPU: the real kernels — values, gradients, number of library calls."""
import numpy as np
import pytest
import torch

from oracle import chord_oracle as oc


def _reference_loop(spmm, idx, Ws, V, n_vec, use_residuals):
    """The loop body of PSFNet.forward, verbatim in structure (psf.py:167-188)."""
    if use_residuals:
        res_conn = V
    for W in Ws:
        V = spmm(idx, W.reshape(W.size(0), W.size(1) * W.size(2)), n_vec, n_vec, V)
        if use_residuals:
            V = V + res_conn
    return V


@pytest.fixture
def cpu_chain(monkeypatch):
    """lazy.chord_chain / chord_spmm -> the oracle's torch port (CPU), counting calls."""
    from sparsefactorization_amd import lazy
    calls = {"chain": 0, "step": 0}

    def index_for(N, L):
        rows, cols = oc.chord_indices(N, L)
        return torch.from_numpy(np.stack([rows, cols]))

    def chain(Ws, V0, use_residual, offsets=None):
        calls["chain"] += 1
        B, N, L = Ws[0].shape
        base = V0 if V0.dim() == 3 else V0.unsqueeze(0).expand(B, -1, -1)
        return oc.torch_chain_port(index_for(N, L), list(Ws), base, use_residual)

    def step(W, V, residual=None, offsets=None):
        calls["step"] += 1
        B, N, L = W.shape
        out = oc.torch_spmm_port(index_for(N, L), W.reshape(B, N * L), N, N, V)
        return out if residual is None else out + residual

    monkeypatch.setattr(lazy, "chord_chain", chain)
    monkeypatch.setattr(lazy, "chord_spmm", step)
    return calls


def _inputs(B=3, N=64, M=5, C=8, seed=0):
    g = torch.Generator().manual_seed(seed)
    L = M + 1
    Ws = [0.3 * torch.randn(B, N, L, generator=g) for _ in range(M)]
    V0 = torch.randn(B, N, C, generator=g)
    rows, cols = oc.chord_indices(N, L)
    return torch.from_numpy(np.stack([rows, cols])), Ws, V0


@pytest.mark.parametrize("res", [True, False])
def test_reference_loop_becomes_one_chain_call(cpu_chain, res):
    from sparsefactorization_amd import lazy
    idx, Ws, V0 = _inputs()
    V = _reference_loop(lazy.spmm, idx, Ws, V0, 64, res)
    assert isinstance(V, lazy.LazyChordChain) and cpu_chain == {"chain": 0, "step": 0}
    # shape / dtype / device queries do not run anything
    assert V.shape == (3, 64, 8) and V.size(0) == 3 and V.dim() == 3 and V.dtype == torch.float32 and not V.is_cuda
    assert cpu_chain == {"chain": 0, "step": 0}
    flat = V.view(V.size(0), -1)  # first use of values: the chain runs once
    assert cpu_chain == {"chain": 1, "step": 0} and type(flat) is torch.Tensor
    want = oc.torch_chain_port(idx, Ws, V0, res)
    assert torch.equal(flat, want.view(3, -1))
    assert torch.equal(V[:, 0, :], want[:, 0, :]) and torch.equal(torch.nn.functional.dropout(V, 0.0), want)
    assert cpu_chain == {"chain": 1, "step": 0}  # cached


def test_residual_on_some_steps_only_and_foreign_adds(cpu_chain):
    from sparsefactorization_amd import lazy
    idx, Ws, V0 = _inputs(M=3)
    reshape = lambda W: W.reshape(W.size(0), -1)  # noqa: E731
    V = lazy.spmm(idx, reshape(Ws[0]), 64, 64, V0)
    V = V + V0                                        # fused residual
    V = lazy.spmm(idx, reshape(Ws[1]), 64, 64, V)    # no residual on this step
    V = lazy.spmm(idx, reshape(Ws[2]), 64, 64, V)
    V = V0 + V                                        # commuted add: fused too
    got = V * 1.0
    assert cpu_chain == {"chain": 0, "step": 3}
    x = oc.torch_spmm_port(idx, reshape(Ws[0]), 64, 64, V0) + V0
    x = oc.torch_spmm_port(idx, reshape(Ws[1]), 64, 64, x)
    x = oc.torch_spmm_port(idx, reshape(Ws[2]), 64, 64, x) + V0
    assert torch.equal(got, x)
    # adding something that is not the start tensor is an ordinary add on the evaluated chain
    other = torch.ones_like(V0)
    V2 = lazy.spmm(idx, reshape(Ws[0]), 64, 64, V0) + other
    assert type(V2) is torch.Tensor
    assert torch.equal(V2, oc.torch_spmm_port(idx, reshape(Ws[0]), 64, 64, V0) + other)
    # a clone of the start tensor is a different tensor: no fusion either (values equal, storage not)
    V3 = lazy.spmm(idx, reshape(Ws[0]), 64, 64, V0) + V0.clone()
    assert type(V3) is torch.Tensor


def test_broadcast_start_tensor_and_intermediate_reuse(cpu_chain):
    from sparsefactorization_amd import lazy
    idx, Ws, _ = _inputs(M=2, N=32, C=32)
    eye = torch.eye(32)
    reshape = lambda W: W.reshape(W.size(0), -1)  # noqa: E731
    A1 = lazy.spmm(idx, reshape(Ws[0]), 32, 32, eye)       # unbatched eye(N): pathfinder_inference.py:57,75-81
    A2 = lazy.spmm(idx, reshape(Ws[1]), 32, 32, A1)
    assert A2.shape == (3, 32, 32)
    m1 = oc.torch_spmm_port(idx, reshape(Ws[0]), 32, 32, eye)
    assert torch.equal(A2.contiguous(), oc.torch_spmm_port(idx, reshape(Ws[1]), 32, 32, m1))
    assert torch.equal(A1.contiguous(), m1)  # an intermediate that is read later evaluates its own prefix


@pytest.mark.gpu
@pytest.mark.parametrize("B,N,M,C,res", [(40, 128, 7, 8, True), (4, 1024, 11, 32, False), (2, 4097, 12, 8, True)])
def test_lazy_loop_on_the_gpu_matches_chord_chain(gpu, B, N, M, C, res):
    import sparsefactorization_amd as sfa
    from sparsefactorization_amd import chord, lazy
    g = torch.Generator(device=gpu).manual_seed(0)
    L = M + 1
    idx = torch.tensor(sfa.get_chord_indices_assym(N, L)).to(gpu)
    Wa = [(0.3 * torch.randn(B, N, L, device=gpu, generator=g)).requires_grad_(True) for _ in range(M)]
    Va = torch.randn(B, N, C, device=gpu, generator=g).requires_grad_(True)
    Wb = [w.detach().clone().requires_grad_(True) for w in Wa]
    Vb = Va.detach().clone().requires_grad_(True)
    calls = []
    real_apply = chord._ChordChain.apply
    chord._ChordChain.apply = lambda *a: (calls.append(1), real_apply(*a))[1]
    try:
        out = _reference_loop(lazy.spmm, idx, Wa, Va, N, res)
        assert isinstance(out, lazy.LazyChordChain) and not calls
        head = torch.nn.functional.dropout(out, 0.0).reshape(B, -1)  # what the models do next (dropout3, view)
        assert len(calls) == 1
    finally:
        chord._ChordChain.apply = real_apply
    want = sfa.chord_chain(Wb, Vb, res)
    assert torch.equal(head, want.reshape(B, -1))
    gz = torch.randn(B, N * C, device=gpu, generator=g)
    head.backward(gz)
    want.reshape(B, -1).backward(gz)
    assert all(torch.equal(a.grad, b.grad) for a, b in zip(Wa, Wb)) and torch.equal(Va.grad, Vb.grad)
    with torch.no_grad():  # evaluation mode: recorded and run without autograd
        ev = _reference_loop(lazy.spmm, idx, Wa, Va, N, res)
        assert torch.equal(ev[:, 0, :], want[:, 0, :]) and not ev.materialize().requires_grad


def test_in_place_adds_reach_every_alias_and_properties_come_from_the_value(cpu_chain):
    """``V.add_(res)`` / ``V += res`` are not recorded (a recorded add returns a NEW chain and would leave aliases of V
    without the residual): the chain runs and the add happens in place on its value. Properties other than the
    shape / dtype / device family are read from the evaluated tensor (requires_grad, T, data ...)."""
    from sparsefactorization_amd import lazy
    idx, Ws, V0 = _inputs(M=2)
    reshape = lambda W: W.reshape(W.size(0), -1)  # noqa: E731
    want = oc.torch_spmm_port(idx, reshape(Ws[0]), 64, 64, V0) + V0
    V = lazy.spmm(idx, reshape(Ws[0]), 64, 64, V0)
    alias = V
    V.add_(V0)                                    # statement form: V itself must now hold the sum
    assert torch.equal(alias * 1.0, want) and torch.equal(V * 1.0, want)
    V = lazy.spmm(idx, reshape(Ws[0]), 64, 64, V0)
    alias = V
    V += V0
    assert torch.equal(V * 1.0, want) and torch.equal(alias * 1.0, want)
    Wg = Ws[0].clone().requires_grad_(True)
    with torch.enable_grad():
        Vg = lazy.spmm(idx, reshape(Wg), 64, 64, V0)
        assert Vg.shape == (3, 64, 8) and cpu_chain["step"] + cpu_chain["chain"] == 2  # shape: still nothing new ran
        assert Vg.requires_grad and Vg.grad_fn is not None and not Vg.is_leaf     # read from the evaluated chain
    assert torch.equal(Vg.mT, Vg.materialize().mT) and torch.equal(Vg.data, Vg.materialize().data)
