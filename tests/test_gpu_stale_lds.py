"""GPU: results must not depend on what earlier kernels left in LDS — or on what freshly allocated memory holds.

Every kernel here stages only what it needs; lanes or rows outside a ragged shape read LDS slots nobody wrote. Such a lane
must leave a sum by a select — a zero factor times a stale NaN is NaN. (Round 5 found the dW window kernel doing the
latter: a dW element stayed NaN once in many runs, depending on which kernels had used the CU before;
test_gpu_parity.py::test_dw_window_kernel_ignores_what_earlier_kernels_left_in_lds pins that kernel.) Here every family of
kernels is run right after launches that leave NaN in the LDS of every CU: the LDS-resident chain on all-NaN operands
(2 x 66 KB per CU) and the wide-row forward step on all-NaN rows (its own allocation sizes, so other offsets). The second half
of the file does the same for global memory: NaN is left in the blocks torch's caching allocator hands out next, so every
workspace (packed weight images, bf16 term planes, per-workgroup partial sums, saved activations) starts as NaN; a whole
training step must come out bit-identical with and without."""
import numpy as np
import pytest
import torch

from conftest import rel_inf
from oracle import chord_oracle as oc

pytestmark = pytest.mark.gpu

TOL = 1e-5
_poison_cache = {}


def _poison(gpu):
    import sparsefactorization_amd as sfa
    if not _poison_cache:
        B, N, C, L = 2048, 512, 16, 10
        _poison_cache["chain"] = ([torch.zeros(B, N, L, device=gpu) for _ in range(2)], torch.full((B, N, C), float("nan"), device=gpu))
        assert "chord_chain_lds_k" in sfa._lib.describe_chain_fwd(B, N, L, C, 2)
        _poison_cache["wide"] = (torch.zeros(64, 2048, 12, device=gpu), torch.full((64, 2048, 128), float("nan"), device=gpu))
    with torch.no_grad():
        Ws, V0 = _poison_cache["chain"]
        out = sfa.chord_chain(Ws, V0, True)
        assert bool(torch.isnan(out[0, 0, 0]))
        W, V = _poison_cache["wide"]
        sfa.chord_spmm(W, V)


def _t(a, gpu):
    return torch.from_numpy(a).to(gpu)


def _mk(shape, seed, scale=1.0):
    return (scale * np.random.default_rng(seed).standard_normal(shape)).astype(np.float32)


# channel groups that are not a power of two (idle lanes), ragged and tiny N, narrow rows on the fused backward step (aligned
# and edge instances), rows of >= 32 channels (chunk-looping dW), L at both ends
SHAPES = [(2, 54, 8, 88), (3, 515, 9, 24), (2, 1031, 11, 28), (1, 433, 16, 136), (2, 300, 12, 20), (2, 2048, 12, 8), (2, 1025, 11, 8),
          (2, 4097, 13, 32), (2, 1000, 12, 128), (3, 37, 5, 12), (1, 700, 20, 4), (2, 640, 4, 36), (2, 777, 9, 16), (1, 2000, 12, 100)]


@pytest.mark.parametrize("B,N,L,C", SHAPES)
def test_chord_steps_after_nan_in_every_lds(gpu, B, N, L, C):
    import sparsefactorization_amd as sfa
    from sparsefactorization_amd.chord import _launch_bwd
    W, V, R, dZ = _mk((B, N, L), 1, 0.5), _mk((B, N, C), 2), _mk((B, N, C), 3), _mk((B, N, C), 4)
    Wt, Vt, Rt, dZt = _t(W, gpu), _t(V, gpu), _t(R, gpu), _t(dZ, gpu)
    dF, dV = oc.spmul_bwd(dZ, W, V)
    _poison(gpu)
    got = sfa.chord_spmm(Wt, Vt, Rt).cpu().numpy()
    assert np.array_equal(got, oc.spmul_fwd(W, V, None) + R)
    for fused in (1, 0):
        sfa.set_tuning("bwd_fused", fused)
        try:
            _poison(gpu)
            gW = torch.full_like(Wt, float("nan"))
            gV = torch.full_like(Vt, float("nan"))
            _launch_bwd(dZt, Wt, Vt, gW, gV, B, N, L, C, N * C, None)
        finally:
            sfa.set_tuning("bwd_fused", 1)
        assert np.array_equal(gV.cpu().numpy(), dV), fused
        assert rel_inf(gW.cpu().numpy(), dF) <= TOL, fused


@pytest.mark.parametrize("B,N,L,C,M", [(3, 128, 8, 8, 7), (2, 100, 7, 12, 3), (5, 512, 10, 4, 9), (2, 37, 5, 20, 4)])
def test_lds_resident_chain_after_nan_in_every_lds(gpu, B, N, L, C, M):
    import sparsefactorization_amd as sfa
    Ws = np.stack([_mk((B, N, L), 10 + m, 0.3) for m in range(M)])
    V0 = _mk((B, N, C), 9)
    rows, cols = oc.chord_indices(N, L)
    want = oc.chain(np.stack([rows, cols]), Ws, V0, True)[-1]
    Wts, V0t = [_t(Ws[m], gpu) for m in range(M)], _t(V0, gpu)
    _poison(gpu)
    with torch.no_grad():
        got = sfa.chord_chain(Wts, V0t, True).cpu().numpy()
    assert np.array_equal(got, want)


@pytest.mark.parametrize("case", ["odd", "c4", "imdb", "lds_c4_h128", "cfg1", "l4_e4"])
def test_fused_mixer_after_nan_in_every_lds(gpu, case):
    import test_gpu_mixer as tm
    args = next(c for c in tm.CASES if c[0] == case)
    _poison(gpu)
    tm.test_mixer_matches_the_oracle_chain_fed_with_float64_mlp_w(gpu, *args)


@pytest.mark.parametrize("i", [2, 3, 4, 5])
def test_producer_mlps_after_nan_in_every_lds(gpu, i):
    import test_gpu_producer as tpd
    _poison(gpu)
    tpd.test_fused_mlp_forward_matches_pytorch(gpu, *tpd.MLP_CASES[i])
    _poison(gpu)
    tpd.test_fused_mlp_backward_matches_float64_autograd(gpu, *tpd.MLP_TRAIN_CASES[i])


def _poison_free_memory(gpu):
    """NaN in the blocks torch's caching allocator will hand out next: every workspace the wrappers get from torch.empty then
    starts as NaN (the allocator splits the large blocks for smaller requests)."""
    torch.cuda.empty_cache()
    blocks = [torch.full((n,), float("nan"), device=gpu) for n in (1 << 28, 1 << 26, 1 << 26, 1 << 24, 1 << 22, 1 << 20, 1 << 18, 1 << 16, 1 << 14)]
    small = [torch.full((n,), float("nan"), device=gpu) for n in (64, 256, 1024, 4096) for _ in range(64)]
    del blocks, small
    # best effort beyond that: the free remainders of segments that live tensors keep open are handed out first (best fit) — walk
    # a few request sizes and leave NaN in whatever comes back
    for n in (1 << 22, 1 << 20, 1 << 18, 1 << 16, 1 << 12, 256):
        probes = [torch.empty(n, device=gpu) for _ in range(4)]
        for p in probes:
            p.fill_(float("nan"))
        del probes


@pytest.mark.parametrize("i", [1, 2, 3, 4, 5])
def test_producer_mlps_with_nan_in_every_fresh_allocation(gpu, i):
    """Narrow producers (workspaces: packed images, per-workgroup partial sums) on ragged token counts."""
    import test_gpu_producer as tpd
    _poison_free_memory(gpu)
    tpd.test_fused_mlp_forward_matches_pytorch(gpu, *tpd.MLP_CASES[i])
    _poison_free_memory(gpu)
    tpd.test_fused_mlp_backward_matches_float64_autograd(gpu, *tpd.MLP_TRAIN_CASES[i])


@pytest.mark.parametrize("T,E,layers", [(1000, 64, [(96, 32), (33, 1), (128, 20)]), (4097, 512, [(128, 128)] + [(128, 12)] * 3),
                                        (37, 16, [(5, 3), (128, 32)]), (2049 * 2 + 1, 128, [(100, 17), (64, 64)])])
def test_wide_mlps_with_nan_in_every_fresh_allocation(gpu, T, E, layers):
    """The wide producers keep bf16 term planes, Hpre and partial sums in scratch memory: rows past T and columns past the
    widths must not reach a sum through a zero factor."""
    import copy
    from sparsefactorization_amd import fused_mlp
    from sparsefactorization_amd.psfnet import MLPBlock
    torch.manual_seed(7)
    blocks = [MLPBlock([h, 'GELU'], E, o).to(gpu) for h, o in layers]
    ref_blocks = [copy.deepcopy(b).double() for b in blocks]
    x = torch.randn(T, E, device=gpu).requires_grad_(True)
    xr = x.detach().double().requires_grad_(True)
    gys = [torch.randn(T, o, device=gpu) for _, o in layers]
    assert fused_mlp.wide_ok(x, blocks)
    _poison_free_memory(gpu)
    ys = fused_mlp.wide_apply(x, blocks)
    refs = [b(xr) for b in ref_blocks]
    for y, r in zip(ys, refs):
        assert rel_inf(y.detach().cpu().numpy(), r.detach().cpu().numpy()) <= TOL
    _poison_free_memory(gpu)
    torch.autograd.backward(ys, gys)
    torch.autograd.backward(refs, [g.double() for g in gys])
    assert rel_inf(x.grad.cpu().numpy(), xr.grad.cpu().numpy()) <= 2e-5
    for b, rb in zip(blocks, ref_blocks):
        for (name, p), (_, rp) in zip(b.named_parameters(), rb.named_parameters()):
            assert rel_inf(p.grad.cpu().numpy(), rp.grad.cpu().numpy()) <= 2e-5, name


@pytest.mark.parametrize("case", ["odd", "imdb", "cfg1", "l4_e4"])
def test_fused_mixer_with_nan_in_every_fresh_allocation(gpu, case):
    import test_gpu_mixer as tm
    args = next(c for c in tm.CASES if c[0] == case)
    _poison_free_memory(gpu)
    tm.test_mixer_matches_the_oracle_chain_fed_with_float64_mlp_w(gpu, *args)


def test_model_training_step_with_nan_in_every_fresh_allocation(gpu):
    """One optimisation step of a small PSFNet (tokens, positional rows, residual, FLATTEN head) twice from the same state: once
    as it comes, once with NaN in everything the allocator hands out — same loss, same updated weights, bit for bit."""
    import copy
    from sparsefactorization_amd import psf_training
    from sparsefactorization_amd.train import make_adam
    torch.manual_seed(3)
    net0 = psf_training.build_model("order", 257).to(gpu)
    X, Y = psf_training.make_split("order", 6, 257, gpu, 5)
    results = []
    for poison in (False, True):
        net = copy.deepcopy(net0)
        opt = make_adam(net.parameters(), 1e-3)
        loss = torch.nn.CrossEntropyLoss()
        if poison:
            _poison_free_memory(gpu)
        out = loss(net(X).squeeze(), Y)
        out.backward()
        opt.step()
        results.append((float(out.detach()), [p.detach().clone() for p in net.parameters()]))
    assert results[0][0] == results[1][0]
    for a, b in zip(results[0][1], results[1][1]):
        assert torch.equal(a, b)
