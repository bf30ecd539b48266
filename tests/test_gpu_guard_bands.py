"""GPU: no kernel writes outside its outputs.

Every output tensor is a view into a larger buffer with 256 bytes of NaN on either side (the view itself 16-byte aligned, as
the window kernels want, and — for the chord step — also one float off, which sends the launch to the general instances).
After the call the bands must still be NaN and the outputs fully written. The operands sit inside NaN bands too: a kernel that
reads past an operand's ends and USES what it finds shows it in its result (reads that are discarded stay invisible). Shapes: ragged N, channel groups that are not a power
of two, the fused backward step (aligned and edge instances), wide rows, the LDS-resident chain, the mixers, the producer
MLPs with partial token tiles."""
import numpy as np
import pytest
import torch

from conftest import rel_inf
from oracle import chord_oracle as oc

pytestmark = pytest.mark.gpu

TOL = 1e-5
BAND = 64  # floats


def _banded(shape, gpu, shift=0):
    n = int(np.prod(shape))
    buf = torch.full((n + 2 * BAND + 4,), float("nan"), device=gpu)
    view = buf[BAND + shift:BAND + shift + n].view(*shape)
    return buf, view, (BAND + shift, BAND + shift + n)


def _bands_intact(buf, span):
    lo, hi = span
    return bool(torch.isnan(buf[:lo]).all()) and bool(torch.isnan(buf[hi:]).all()) and not bool(torch.isnan(buf[lo:hi]).any())


def _banded_input(a, gpu, shift=0):
    """The array inside NaN bands: a kernel that reads past an operand's ends and uses what it finds shows NaN in its result."""
    _, view, _ = _banded(a.shape, gpu, shift)
    view.copy_(torch.from_numpy(a))
    return view


def _mk(shape, seed, scale=1.0):
    return (scale * np.random.default_rng(seed).standard_normal(shape)).astype(np.float32)


def _t(a, gpu):
    return torch.from_numpy(a).to(gpu)


SHAPES = [(2, 54, 8, 88), (3, 515, 9, 24), (2, 2048, 12, 8), (2, 1025, 11, 8), (2, 4097, 13, 32), (2, 1000, 12, 128), (3, 37, 5, 12),
          (1, 700, 20, 4), (2, 640, 4, 36), (1, 2000, 12, 100), (3, 16384, 15, 8), (1, 300, 12, 260), (2, 5, 3, 3), (1, 1, 1, 1)]


@pytest.mark.parametrize("shift", [0, 1])
@pytest.mark.parametrize("B,N,L,C", SHAPES)
def test_chord_step_and_gradients_stay_inside_their_outputs(gpu, B, N, L, C, shift):
    from sparsefactorization_amd import _lib
    from sparsefactorization_amd.chord import _launch_bwd
    W, V, R, dZ = _mk((B, N, L), 1, 0.5), _mk((B, N, C), 2), _mk((B, N, C), 3), _mk((B, N, C), 4)
    Wt, Vt, Rt, dZt = (_banded_input(a, gpu, shift) for a in (W, V, R, dZ))
    obuf, out, ospan = _banded((B, N, C), gpu, shift)
    lib = _lib.load()
    s = torch.cuda.current_stream(gpu).cuda_stream
    _lib.check(lib.psf_chord_spmm_fwd_f32(Wt.data_ptr(), Vt.data_ptr(), Rt.data_ptr(), out.data_ptr(), B, N, L, C, N * C, None, s), "fwd")
    assert _bands_intact(obuf, ospan)
    assert np.array_equal(out.cpu().numpy(), oc.spmul_fwd(W, V, None) + R)
    wbuf, gW, wspan = _banded((B, N, L), gpu, shift)
    vbuf, gV, vspan = _banded((B, N, C), gpu, shift)
    _launch_bwd(dZt, Wt, Vt, gW, gV, B, N, L, C, N * C, None)
    assert _bands_intact(wbuf, wspan) and _bands_intact(vbuf, vspan)
    dF, dV = oc.spmul_bwd(dZ, W, V)
    assert np.array_equal(gV.cpu().numpy(), dV) and rel_inf(gW.cpu().numpy(), dF) <= TOL


@pytest.mark.parametrize("B,N,L,C,M", [(3, 128, 8, 8, 7), (2, 100, 7, 12, 3), (5, 512, 10, 4, 9), (2, 37, 5, 20, 4), (2, 2048, 12, 8, 3),
                                       (2, 1025, 11, 32, 3)])
def test_chain_steps_stay_inside_their_outputs(gpu, B, N, L, C, M):
    """psf_chord_chain_fwd_f32 with every step's result kept (the training form): M banded outputs."""
    import ctypes
    from sparsefactorization_amd import _lib
    Ws = np.stack([_mk((B, N, L), 10 + m, 0.3) for m in range(M)])
    V0 = _mk((B, N, C), 9)
    rows, cols = oc.chord_indices(N, L)
    want = oc.chain(np.stack([rows, cols]), Ws, V0, True)
    Wts, V0t = [_banded_input(Ws[m], gpu) for m in range(M)], _banded_input(V0, gpu)
    outs = [_banded((B, N, C), gpu) for _ in range(M)]
    lib = _lib.load()
    vp = ctypes.c_void_p
    rc = lib.psf_chord_chain_fwd_f32((vp * M)(*[w.data_ptr() for w in Wts]), V0t.data_ptr(), (vp * M)(*[o[1].data_ptr() for o in outs]),
                                     M, 1, B, N, L, C, N * C, None, torch.cuda.current_stream(gpu).cuda_stream)
    _lib.check(rc, "chain")
    for m, (buf, view, span) in enumerate(outs):
        assert _bands_intact(buf, span), m
        assert np.array_equal(view.cpu().numpy(), want[m]), m


@pytest.mark.parametrize("T,E,layers", [(31, 4, [(5, 3)]), (2049, 8, [(7, 2), (40, 31)]), (1000, 28, [(96, 32), (33, 1), (128, 20)]),
                                        (4097 * 2 + 3, 32, [(32, 8)] + [(32, 15)] * 3), (1024 * 3 + 7, 16, [(16, 16)] + [(16, 11)] * 2)])
def test_producer_mlps_stay_inside_their_outputs(gpu, T, E, layers):
    """psf_mlp_fwd_f32 / psf_mlp_bwd_f32 through the raw entry points: every Y, dX and every weight gradient banded."""
    import ctypes
    from sparsefactorization_amd import _lib, fused_mlp
    from sparsefactorization_amd.psfnet import MLPBlock
    torch.manual_seed(3)
    blocks = [MLPBlock([h, 'GELU'], E, o).to(gpu) for h, o in layers]
    x = _banded_input(np.random.default_rng(5).standard_normal((T, E)).astype(np.float32), gpu)
    params = fused_mlp._params_of(blocks)
    K = len(layers)
    As, as_, Bs, bs = params[0::4], params[1::4], params[2::4], params[3::4]
    h = (ctypes.c_int32 * K)(*[hh for hh, _ in layers])
    O = (ctypes.c_int32 * K)(*[oo for _, oo in layers])
    lib = _lib.load()
    s = torch.cuda.current_stream(gpu).cuda_stream
    ptrs = fused_mlp._ptrs
    ys = [_banded((T, o), gpu) for _, o in layers]
    ws_bytes = lib.psf_mlp_fwd_workspace(E, K, h, O)
    assert ws_bytes >= 0
    ws = torch.empty(ws_bytes // 4 + 4, device=gpu)
    _lib.check(lib.psf_mlp_fwd_f32(x.data_ptr(), T, E, K, ptrs(As), ptrs(as_), ptrs(Bs), ptrs(bs), h, O, ptrs([y[1] for y in ys]),
                                   ws.data_ptr(), ws_bytes, s), "mlp fwd")
    with torch.no_grad():
        for (buf, view, span), blk in zip(ys, blocks):
            assert _bands_intact(buf, span)
            assert rel_inf(view.cpu().numpy(), blk.double()(x.double()).float().cpu().numpy()) <= TOL
            blk.float()
    gys = [_banded_input(np.random.default_rng(6 + i).standard_normal((T, o)).astype(np.float32), gpu) for i, (_, o) in enumerate(layers)]
    grads = [_banded(tuple(p.shape), gpu) for p in params]
    dxb = _banded((T, E), gpu)
    wsb = lib.psf_mlp_bwd_workspace(T, E, K, h, O)
    assert wsb >= 0
    ws2 = torch.empty(wsb // 4 + 4, device=gpu)
    g = [gr[1] for gr in grads]
    _lib.check(lib.psf_mlp_bwd_f32(x.data_ptr(), T, E, K, ptrs(As), ptrs(as_), ptrs(Bs), h, O, ptrs(gys), dxb[1].data_ptr(),
                                   ptrs(g[0::4]), ptrs(g[1::4]), ptrs(g[2::4]), ptrs(g[3::4]), ws2.data_ptr(), wsb, s), "mlp bwd")
    assert _bands_intact(dxb[0], dxb[2])
    for buf, _, span in grads:
        assert _bands_intact(buf, span)
    xr = x.clone().requires_grad_(True)
    ref = torch.autograd.grad([blk.double()(xr.double()) for blk in blocks], [xr] + [p for blk in blocks for p in blk.parameters()],
                              [gy.double() for gy in gys])
    assert rel_inf(dxb[1].cpu().numpy(), ref[0].cpu().numpy()) <= 2e-5
    for (_, view, _), r in zip(grads, ref[1:]):
        assert rel_inf(view.cpu().numpy(), r.float().cpu().numpy()) <= 2e-5


@pytest.mark.parametrize("case", ["odd", "c4", "imdb", "two_tiles", "lds_c4_h128", "cfg1", "lds_n64", "l4_e4"])
def test_fused_mixer_stays_inside_its_outputs(gpu, case):
    """psf_mixer_fwd_in_f32 (per-step kernels and the single-launch LDS-resident kernel) with V0 and both alternating step
    buffers banded; the result against the oracle chain fed with float64-MLP W (test_gpu_mixer's reference)."""
    import ctypes
    import test_gpu_mixer as tm
    from sparsefactorization_amd import _lib, fused_mixer, fused_mlp
    _, B, N, E, h, C, L, M, residual = next(c for c in tm.CASES if c[0] == case)
    g, fs = tm._blocks(E, h, C, L, M, seed=11)
    x = torch.randn(B, N, E, generator=torch.Generator().manual_seed(5))
    want, _ = tm._reference(x, g, fs, residual)
    g.to(gpu)
    for f in fs:
        f.to(gpu)
    xd = x.to(gpu)
    Mh, htab, C2, L2 = fused_mixer._block_sizes(E, g, fs)
    assert (Mh, C2, L2) == (M, C, L)
    lib = _lib.load()
    spec = _lib.MixerInput(_lib.MIXER_IN_DATA, 0, xd.data_ptr(), None, None, None)
    params = [p.detach().contiguous() for p in fused_mlp._params_of([g, *fs])]
    ws_bytes = lib.psf_mixer_fwd_workspace(N, E, M, htab, C, L)
    assert ws_bytes >= 0
    ws = torch.empty(ws_bytes // 4 + 4, device=gpu)
    v0 = _banded((B, N, C), gpu)
    bufs = [_banded((B, N, C), gpu) for _ in range(min(M, 2))]
    o_tab = (ctypes.c_void_p * M)(*[bufs[m % len(bufs)][1].data_ptr() for m in range(M)])
    ptrs = fused_mixer._ptrs
    rc = lib.psf_mixer_fwd_in_f32(ctypes.byref(spec), B, N, E, M, ptrs(params[0::4]), ptrs(params[1::4]), ptrs(params[2::4]),
                                  ptrs(params[3::4]), htab, C, L, 1 if residual else 0, v0[1].data_ptr(), o_tab, ws.data_ptr(),
                                  ws_bytes, torch.cuda.current_stream(gpu).cuda_stream)
    _lib.check(rc, "psf_mixer_fwd_in_f32")
    last = bufs[(M - 1) % len(bufs)]
    assert _bands_intact(last[0], last[2])
    for buf, _, span in bufs:
        lo, hi = span
        assert bool(torch.isnan(buf[:lo]).all()) and bool(torch.isnan(buf[hi:]).all())
    lo, hi = v0[2]
    assert bool(torch.isnan(v0[0][:lo]).all()) and bool(torch.isnan(v0[0][hi:]).all())
    assert rel_inf(last[1].cpu().numpy(), want) <= TOL
