"""GPU: the wide producer-MLP kernels (csrc/mlp_wide.hip, csrc/x3_gemm.h) against float64 PyTorch modules.

The wide path covers the LRA widths of MLPBlock (LRA/psf.py:35-60; reference ListOps: E = 512, hidden 128, outputs 12 and
128, LRA/psf_training_config.py:2-30): forward <= 1e-5, gradients <= 2e-5 of max|reference| per tensor, bit-reproducible.
"""
import copy
import ctypes

import numpy as np
import pytest
import torch

from conftest import rel_inf

pytestmark = pytest.mark.gpu


def _blocks(gpu, E, layers):
    from sparsefactorization_amd.psfnet import MLPBlock
    return [MLPBlock([h, 'GELU'], E, o).to(gpu) for h, o in layers]


def _check(gpu, T, E, layers, seed, lead=None, scale=1.0, tol_fwd=1e-5, tol_bwd=2e-5):
    from sparsefactorization_amd import fused_mlp
    torch.manual_seed(seed)
    blocks = _blocks(gpu, E, layers)
    refs = [copy.deepcopy(b).double() for b in blocks]
    shape = (T, E) if lead is None else (*lead, E)
    x = (torch.randn(*shape, device=gpu) * scale).requires_grad_(True)
    xr = x.detach().double().requires_grad_(True)
    assert fused_mlp.wide_ok(x, blocks), (T, E, layers)
    ys = fused_mlp.wide_apply(x, blocks)
    yr = [b(xr) for b in refs]
    for y, r in zip(ys, yr):
        assert y.shape == r.shape
        assert rel_inf(y.detach().cpu().numpy(), r.detach().cpu().numpy()) <= tol_fwd, (T, E, layers)
    gys = [torch.randn_like(y) for y in ys]
    torch.autograd.backward(ys, gys)
    torch.autograd.backward(yr, [g.double() for g in gys])
    assert rel_inf(x.grad.cpu().numpy(), xr.grad.cpu().numpy()) <= tol_bwd, (T, E, layers)
    for k, (b, rb) in enumerate(zip(blocks, refs)):
        for (name, p), (_, rp) in zip(b.named_parameters(), rb.named_parameters()):
            assert p.grad is not None and p.grad.shape == rp.grad.shape, (k, name)
            assert rel_inf(p.grad.cpu().numpy(), rp.grad.cpu().numpy()) <= tol_bwd, (k, name, T, E, layers)
    return blocks, x, gys


WIDE_CASES = [  # (T, E, [(h, out), ...])
    (64000, 512, [(128, 128)] + [(128, 12)] * 11),   # reference ListOps: g + 11 link MLPs (psf_training_config.py:2-30)
    (32 * 2048, 64, [(64, 64)] + [(64, 12)] * 11),   # BASELINE.json cfg3 wording: N = 2048, dim = 64
    (4097 * 2, 64, [(128, 32)] + [(128, 13)] * 12),  # IMDb-like links at E = 64
    (1000, 48, [(96, 33), (33, 1), (128, 20)]),      # ragged token count, odd widths, outputs not multiples of 4
    (255, 16, [(5, 3)]),                             # smaller than one GEMM tile
    (257, 512, [(128, 128), (100, 127)]),            # one token past a tile, two wide outputs
]


@pytest.mark.parametrize("T,E,layers", WIDE_CASES)
def test_wide_mlp_forward_backward_match_float64(gpu, T, E, layers):
    _check(gpu, T, E, layers, seed=T + E)


def _random_wide_case(seed):
    rng = np.random.default_rng(1000 + seed)
    E = int(rng.choice([16, 32, 48, 64, 80, 128, 256, 272, 512]))
    K = int(rng.integers(1, 7)) if seed < 32 else int(rng.integers(13, 25))  # the last seeds: many MLPs (up to the limit of 24)
    layers = [(int(rng.choice([1, 7, 31, 32, 33, 64, 65, 96, 127, 128])),
               int(rng.choice([1, 2, 8, 12, 15, 16, 17, 31, 32, 33, 64, 100, 128]))) for _ in range(K)]
    T = int(rng.choice([1, 31, 33, 255, 256, 257, 1000, 4096, 4097, 9999]))
    return T, E, layers


@pytest.mark.parametrize("seed", range(40))
def test_wide_mlp_random_shapes(gpu, seed):
    """Seeded random widths around every boundary of the kernels: 16-column blocks of E, 32-row hidden units, 16-output
    k-steps and 32-output tiles of the second layer, 32-token tiles and the 256-token GEMM tile."""
    T, E, layers = _random_wide_case(seed)
    _check(gpu, T, E, layers, seed=seed, scale=1.5)


def test_wide_mlp_is_bit_reproducible_and_handles_leading_dims(gpu):
    from sparsefactorization_amd import fused_mlp
    blocks, x, gys = _check(gpu, 3 * 700, 128, [(128, 64), (64, 12), (64, 12)], seed=5, lead=(3, 700))
    first = [p.grad.clone() for b in blocks for p in b.parameters()] + [x.grad.clone()]
    for b in blocks:
        b.zero_grad()
    x.grad = None
    ys = fused_mlp.wide_apply(x, blocks)
    assert ys[0].shape == (3, 700, 64)
    torch.autograd.backward(ys, gys)
    again = [p.grad for b in blocks for p in b.parameters()] + [x.grad]
    assert all(torch.equal(u, v) for u, v in zip(first, again))


def test_wide_mlp_inference_partial_outputs_and_frozen_input(gpu):
    """No-grad calls skip the autograd node; outputs that receive no gradient count as zero; an input that needs no
    gradient gets none (dX = NULL skips that GEMM)."""
    from sparsefactorization_amd import fused_mlp
    torch.manual_seed(3)
    blocks = _blocks(gpu, 64, [(64, 16), (64, 12), (64, 12)])
    x = torch.randn(3000, 64, device=gpu)  # no grad on the input
    with torch.no_grad():
        y0 = fused_mlp.wide_apply(x, blocks)
        for b, y in zip(blocks, y0):
            assert rel_inf(y.cpu().numpy(), copy.deepcopy(b).double()(x.double()).cpu().numpy()) <= 1e-5
    ys = fused_mlp.wide_apply(x, blocks)
    assert all(torch.equal(a, b) for a, b in zip(ys, y0))
    ys[1].square().sum().backward()
    ref = copy.deepcopy(blocks[1]).double()
    ref.zero_grad()
    ref(x.double()).square().sum().backward()
    for p, r in zip(blocks[1].parameters(), ref.parameters()):
        assert rel_inf(p.grad.cpu().numpy(), r.grad.cpu().numpy()) <= 2e-5
    for b in (blocks[0], blocks[2]):
        assert all(float(p.grad.abs().max()) == 0.0 for p in b.parameters())


def test_wide_mlp_rejects_bad_arguments(gpu):
    from sparsefactorization_amd import _lib, fused_mlp
    lib = _lib.load()
    h, O = (ctypes.c_int32 * 1)(128), (ctypes.c_int32 * 1)(12)
    assert lib.psf_mlp_wide_fwd_workspace(1000, 520, 1, h, O) == -1      # E not a multiple of 16
    assert lib.psf_mlp_wide_saved_bytes(1000, 512, 1, h, (ctypes.c_int32 * 1)(129)) == -1
    assert lib.psf_mlp_wide_bwd_workspace(1000, 512, 25, h, O) == -1
    assert lib.psf_mlp_wide_saved_bytes(1000, 512, 1, h, O) > 0
    blocks = _blocks(gpu, 24, [(32, 8)])
    assert not fused_mlp.wide_ok(torch.randn(10, 24, device=gpu), blocks)  # E = 24
    assert not fused_mlp.wide_ok(torch.randn(10, 32, device=gpu).double(), _blocks(gpu, 32, [(32, 8)]))


def test_listops_network_takes_the_wide_path(gpu):
    """The reference ListOps configuration (LRA/psf_training_config.py:2-30) trains through psf_mlp_wide_*: its step
    matches a float64 copy of the same network running the stock PyTorch layers."""
    from sparsefactorization_amd import fused_mlp, lra_training
    torch.manual_seed(0)
    net = lra_training.build_model("listops").to(gpu)
    ref = copy.deepcopy(net)
    X, Y = lra_training.synthetic_split("listops", 2, gpu, 1)
    X = lra_training.add_cls_token(X, lra_training.config["listops"]["model"]["vocab_size"])
    calls = []
    orig = fused_mlp.wide_apply
    fused_mlp.wide_apply = lambda x, b: (calls.append(len(b)), orig(x, b))[1]
    try:
        out = net(X)
    finally:
        fused_mlp.wide_apply = orig
    assert calls == [12]
    loss = torch.nn.functional.cross_entropy(out, Y)
    loss.backward()
    fused_mlp.wide_enabled = False
    try:
        out_ref = ref(X)
        torch.nn.functional.cross_entropy(out_ref, Y).backward()
    finally:
        fused_mlp.wide_enabled = True
    assert rel_inf(out.detach().cpu().numpy(), out_ref.detach().cpu().numpy()) <= 1e-4
    for (n, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        if q.grad is None:
            assert p.grad is None, n
        else:
            assert rel_inf(p.grad.cpu().numpy(), q.grad.cpu().numpy()) <= 2e-4, n


def test_second_layer_in_the_gemm_epilogue_equals_the_separate_kernel(gpu):
    """Networks whose MLPs all have 97..128 hidden rows run the second layers of their narrow-output MLPs inside the first
    layers' GEMM epilogue (knob wide_fuse): same arithmetic in the same order as wide_out_k — equal bits —, with a
    wide-output MLP in the same call taking the separate kernel, in training and in inference (which keeps no record)."""
    import sparsefactorization_amd as sfa
    from sparsefactorization_amd import fused_mlp
    torch.manual_seed(21)
    blocks = _blocks(gpu, 128, [(128, 96), (128, 12), (100, 15), (128, 32), (97, 1)])
    x = torch.randn(3, 777, 128, device=gpu)
    outs = {}
    for fuse in (1, 0):
        sfa.set_tuning("wide_fuse", fuse)
        try:
            with torch.no_grad():
                inf = fused_mlp.wide_apply(x, blocks)
            xg = x.clone().requires_grad_(True)
            tr = fused_mlp.wide_apply(xg, blocks)
            torch.autograd.backward(tr, [torch.ones_like(t) for t in tr])
            outs[fuse] = (inf, [t.detach() for t in tr], xg.grad, [p.grad.clone() for b in blocks for p in b.parameters()])
            for b in blocks:
                b.zero_grad()
        finally:
            sfa.set_tuning("wide_fuse", 1)
    for a, b in zip(outs[1][0] + outs[1][1], outs[0][0] + outs[0][1]):
        assert torch.equal(a, b)
    assert all(torch.equal(a, b) for a, b in zip(outs[1][0], outs[1][1]))  # inference == training forward
    assert torch.equal(outs[1][2], outs[0][2]) and all(torch.equal(a, b) for a, b in zip(outs[1][3], outs[0][3]))
    for blk, y in zip(blocks, outs[1][0]):
        ref = copy.deepcopy(blk).double()(x.double())
        assert rel_inf(y.cpu().numpy(), ref.detach().cpu().numpy()) <= 1e-5
