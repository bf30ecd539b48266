"""GPU: the HIP path (through the C ABI, via the ctypes host side) against the CPU oracle.

Tolerance: BASELINE.json asks for <= 1e-5 relative fp32. The forward kernels use the oracle's summation order
with uncontracted mul/add, so most forward checks demand BIT equality; gradients (different association in dW's
cross-lane reduction) are held to max|a-b| <= 1e-5 * max|ref|.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_inf
from oracle import chord_oracle as oc

pytestmark = pytest.mark.gpu

TOL = 1e-5


def _mk(shape, seed, scale=1.0):
    return (np.random.default_rng(seed).standard_normal(shape, dtype=np.float32) * scale).astype(np.float32)


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _oracle_fwd(W, V, res=None, offsets=None):
    out = oc.spmul_fwd(W, V, offsets)
    if res is not None:
        out = out + res
    return out


FWD_SHAPES = [
    # (B, N, L, C) — cfg1; non-pow2 N; N smaller than a tile; duplicate self link (N=1024, L=12); wide rows;
    # C not a multiple of 4 (scalar path); L beyond the window kernels; ragged last tile
    (40, 128, 8, 8), (3, 2000, 12, 16), (2, 100, 9, 8), (2, 1024, 12, 32), (2, 1024, 12, 8), (2, 513, 10, 128),
    (2, 300, 9, 6), (1, 64, 7, 1), (2, 2048, 12, 64), (1, 4097, 13, 8), (2, 777, 22, 4), (1, 16384, 15, 8),
    (2, 256, 9, 260), (1, 1000, 4, 12), (2, 96, 3, 8), (1, 1, 1, 4), (3, 2, 2, 8), (1, 5000, 24, 8),
]


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("B,N,L,C", FWD_SHAPES)
def test_forward_step_bit_exact(gpu, B, N, L, C, variant):
    import sparsefactorization_amd as sfa
    W, V, R = _mk((B, N, L), 1), _mk((B, N, C), 2), _mk((B, N, C), 3)
    sfa.set_tuning("fwd_variant", variant)
    try:
        for res in (None, R):
            got = sfa.chord_spmm(_t(W, gpu), _t(V, gpu), None if res is None else _t(res, gpu)).cpu().numpy()
            want = _oracle_fwd(W, V, res)
            assert np.array_equal(got, want), f"variant={variant} res={res is not None} rel={rel_inf(got, want):.3e}"
    finally:
        sfa.set_tuning("fwd_variant", 0)


@pytest.mark.parametrize("split", [0, 1, 2])  # one predicated launch / auto / full tiles + ragged tiles in two launches
@pytest.mark.parametrize("B,N,L,C,shift", [(3, 1101, 11, 8, 0), (3, 1101, 11, 8, 1), (2, 2000, 12, 16, 3),
                                           (1, 515, 9, 32, 2), (5, 777, 13, 8, 1)])
def test_forward_window_w_alignment_edges(gpu, B, N, L, C, shift, split):
    """W tiles are staged in 16-byte chunks: cover a W buffer whose start is not 16-byte aligned (a view
    `shift` elements into an allocation), whose size is not a multiple of 16 bytes, and both staging paths."""
    import sparsefactorization_amd as sfa
    W, V, R = _mk((B, N, L), 31), _mk((B, N, C), 32), _mk((B, N, C), 33)
    backing = torch.full((B * N * L + 8,), float("nan"), device=gpu)
    Wt = backing[shift:shift + B * N * L].view(B, N, L)
    Wt.copy_(_t(W, gpu))
    assert Wt.data_ptr() % 16 == (4 * shift) % 16
    sfa.set_tuning("fwd_split", split)
    try:
        assert "win" in sfa.describe_fwd(B, N, L, C)
        got = sfa.chord_spmm(Wt, _t(V, gpu), _t(R, gpu)).cpu().numpy()
    finally:
        sfa.set_tuning("fwd_split", 1)
    assert np.array_equal(got, _oracle_fwd(W, V, R))


@pytest.mark.parametrize("split", [0, 1, 2])
def test_forward_split_and_edge_kernels_agree_on_all_shapes(gpu, split):
    import sparsefactorization_amd as sfa
    sfa.set_tuning("fwd_split", split)
    try:
        for (B, N, L, C) in [(40, 128, 8, 8), (2, 1024, 12, 32), (2, 513, 10, 128), (1, 16384, 15, 8), (2, 2048, 12, 64)]:
            W, V = _mk((B, N, L), 34), _mk((B, N, C), 35)
            got = sfa.chord_spmm(_t(W, gpu), _t(V, gpu)).cpu().numpy()
            assert np.array_equal(got, _oracle_fwd(W, V)), (B, N, L, C)
    finally:
        sfa.set_tuning("fwd_split", 1)


@pytest.mark.parametrize("wide", [0, 1, 2, 4])
@pytest.mark.parametrize("B,N,L,C", [(2, 2000, 12, 128), (1, 1024, 12, 1024), (3, 600, 10, 64), (2, 4097, 13, 260),
                                     (2, 512, 9, 64), (2, 511, 9, 64), (1, 1536, 20, 96)])
def test_forward_wide_rows(gpu, B, N, L, C, wide):
    """Rows of >= 64 channels: channel-chunked 1024-thread tiles (fwd_wide = 1; the automatic choice, 0, for 64..256 channels
    and N <= 4096), chunks on 256 threads (2), one workgroup per whole row (4)."""
    import sparsefactorization_amd as sfa
    W, V, R = _mk((B, N, L), 51), _mk((B, N, C), 52), _mk((B, N, C), 53)
    sfa.set_tuning("fwd_wide", wide)
    try:
        desc = sfa.describe_fwd(B, N, L, C)
        got = sfa.chord_spmm(_t(W, gpu), _t(V, gpu), _t(R, gpu)).cpu().numpy()
        got_nores = sfa.chord_spmm(_t(W, gpu), _t(V, gpu)).cpu().numpy()
    finally:
        sfa.set_tuning("fwd_wide", 0)
    if wide == 1 and N >= 512:
        assert "NT=1024" in desc, desc
    if wide == 4:
        assert "NT=256" in desc and "TG=8," not in desc, desc
    assert np.array_equal(got, _oracle_fwd(W, V, R)), desc
    assert np.array_equal(got_nores, _oracle_fwd(W, V)), desc


def test_forward_xcd_remap_off(gpu):
    import sparsefactorization_amd as sfa
    W, V = _mk((5, 1000, 11), 7), _mk((5, 1000, 8), 8)
    sfa.set_tuning("xcd_remap", 0)
    try:
        got = sfa.chord_spmm(_t(W, gpu), _t(V, gpu)).cpu().numpy()
    finally:
        sfa.set_tuning("xcd_remap", 1)
    assert np.array_equal(got, _oracle_fwd(W, V))


@pytest.mark.parametrize("limit", [0, 1, 2, 3, 4])
def test_forward_workgroups_per_cu_limit(gpu, limit):
    """fwd_wg_limit only changes how much LDS a launch asks for (an occupancy cap): results stay bit-exact."""
    import sparsefactorization_amd as sfa
    W, V, R = _mk((3, 2048, 12), 21), _mk((3, 2048, 8), 22), _mk((3, 2048, 8), 23)
    sfa.set_tuning("fwd_wg_limit", limit)
    try:
        got = sfa.chord_spmm(_t(W, gpu), _t(V, gpu), _t(R, gpu)).cpu().numpy()
    finally:
        sfa.set_tuning("fwd_wg_limit", 0)
    assert np.array_equal(got, _oracle_fwd(W, V, R))


@pytest.mark.parametrize("zigzag", [0, 1])
def test_chain_zigzag_tile_order(gpu, zigzag):
    """chain_zigzag only changes which workgroup computes which tile on odd steps (also with a ragged last tile and
    a workgroup count that is not a multiple of 8): the chain stays bit-exact against the oracle."""
    import sparsefactorization_amd as sfa
    B, N, M, L, C = 3, 2304 + 37, 4, 12, 8
    Ws = [_mk((B, N, L), 31 + m) * 0.3 for m in range(M)]
    V0 = _mk((B, N, C), 40)
    sfa.set_tuning("chain_zigzag", zigzag)
    sfa.set_tuning("chain_fused", 0)  # per-step launches also for this short sequence
    try:
        got = sfa.chord_chain([_t(w, gpu) for w in Ws], _t(V0, gpu), True).cpu().numpy()
    finally:
        sfa.set_tuning("chain_zigzag", 1)
        sfa.set_tuning("chain_fused", 1)
    rows, cols = oc.chord_indices(N, L)
    assert np.array_equal(got, oc.chain(np.stack([rows, cols]), np.stack(Ws), V0, True)[-1])


def test_forward_broadcast_first_operand(gpu):
    """Unbatched eye(N) as in ChangedPSF (pathfinder_inference.py:57,75-81)."""
    import sparsefactorization_amd as sfa
    B, N, L = 3, 256, 9
    W = _mk((B, N, L), 9)
    eye = np.eye(N, dtype=np.float32)
    got = sfa.chord_spmm(_t(W, gpu), _t(eye, gpu)).cpu().numpy()
    assert np.array_equal(got, oc.spmul_fwd(W, eye))


def test_forward_explicit_offsets(gpu):
    """Arbitrary / negative / >= N offsets, the `offsets` argument of spmul/spmul.py:15."""
    import sparsefactorization_amd as sfa
    B, N, L, C = 2, 500, 6, 8
    W, V = _mk((B, N, L), 10), _mk((B, N, C), 11)
    off = [3, 0, 499, 1000, -7, 250]
    got = sfa.chord_spmm(_t(W, gpu), _t(V, gpu), offsets=off).cpu().numpy()
    assert np.array_equal(got, oc.spmul_fwd(W, V, off))


@pytest.mark.parametrize("B,N,L,C,far", [(2, 4096, 12, 8, [777, 3001]), (2, 2048, 11, 32, [200, 1000, 1500]),
                                           (3, 16384, 15, 8, [512, 1024, 2048, 4096, 8192])])
def test_window_kernels_with_chord_near_links_and_any_far_offsets(gpu, B, N, L, C, far):
    """The window kernels take the near links as the chord pattern's constants and the far ones as run-time values. Far
    offsets that are multiples of the tile length — every chord offset is — let a launch of full tiles compute its row-block
    addresses on the scalar unit (Geom::aligned, fwd_window.h / bwd_fused.h); any others keep the per-lane form. Both against
    the oracle: forward and dV bit for bit, dW to the parity bar."""
    import sparsefactorization_amd as sfa
    off = [0] + [1 << k for k in range(L - 1 - len(far))] + list(far)
    assert len(off) == L
    W, V, dZ = _mk((B, N, L), 91), _mk((B, N, C), 92), _mk((B, N, C), 93)
    Wt, Vt = _t(W, gpu).requires_grad_(True), _t(V, gpu).requires_grad_(True)
    out = sfa.chord_spmm(Wt, Vt, offsets=off)
    assert np.array_equal(out.detach().cpu().numpy(), oc.spmul_fwd(W, V, off))
    out.backward(_t(dZ, gpu))
    dF, dV = oc.spmul_bwd(dZ, W, V, off)
    assert np.array_equal(Vt.grad.cpu().numpy(), dV)
    assert rel_inf(Wt.grad.cpu().numpy(), dF) <= TOL


def test_forward_f64(gpu):
    import sparsefactorization_amd as sfa
    rng = np.random.default_rng(12)
    W, V = rng.standard_normal((2, 300, 10)), rng.standard_normal((2, 300, 6))
    got = sfa.chord_spmm(_t(W, gpu), _t(V, gpu)).cpu().numpy()
    assert np.array_equal(got, oc.spmul_fwd(W, V))


def test_spmm_dropin_signature(gpu):
    """spmm(index, value, m, n, matrix) exactly as SyntheticExperiments/psf.py:178-184 calls it."""
    import sparsefactorization_amd as sfa
    B, N, L, C = 4, 128, 8, 8
    W, V = _mk((B, N, L), 13), _mk((B, N, C), 14)
    idx = torch.tensor(sfa.get_chord_indices_assym(N, L)).to(gpu)
    Wt = _t(W, gpu)
    got = sfa.spmm(idx, Wt.reshape(Wt.size(0), Wt.size(1) * Wt.size(2)), N, N, _t(V, gpu)).cpu().numpy()
    rows, cols = oc.chord_indices(N, L)
    want = oc.spmm(np.stack([rows, cols]), W.reshape(B, N * L), N, N, V)
    assert np.array_equal(got, want)


BWD_SHAPES = [(4, 128, 8, 8), (2, 2000, 12, 16), (2, 100, 9, 8), (2, 1024, 12, 32), (1, 513, 10, 128),
              (2, 300, 9, 6), (1, 64, 7, 1), (1, 4097, 13, 8), (2, 256, 9, 260)]


@pytest.mark.parametrize("B,N,L,C", BWD_SHAPES)
def test_backward_step(gpu, B, N, L, C):
    import sparsefactorization_amd as sfa
    W, V, dZ = _mk((B, N, L), 15), _mk((B, N, C), 16), _mk((B, N, C), 17)
    Wt = _t(W, gpu).requires_grad_(True)
    Vt = _t(V, gpu).requires_grad_(True)
    Rt = _t(np.zeros_like(V), gpu).requires_grad_(True)
    out = sfa.chord_spmm(Wt, Vt, Rt)
    out.backward(_t(dZ, gpu))
    dF, dV = oc.spmul_bwd(dZ, W, V)
    assert np.array_equal(Vt.grad.cpu().numpy(), dV)  # same order, uncontracted -> exact
    assert rel_inf(Wt.grad.cpu().numpy(), dF) <= TOL
    assert np.array_equal(Rt.grad.cpu().numpy(), dZ)


BWD_WIN_SHAPES = [(3, 16384, 15, 8), (2, 2048, 12, 64), (2, 1101, 11, 8), (2, 2000, 12, 128), (1, 4097, 13, 8),
                  (2, 1024, 12, 32), (2, 777, 9, 16), (2, 640, 10, 260), (5, 1000, 7, 4), (2, 515, 20, 24)]


@pytest.mark.parametrize("dv_threads", [0, 1])  # dV: automatic (512 threads x 1 row for C <= 8) / 256 threads x 2 rows
@pytest.mark.parametrize("B,N,L,C", BWD_WIN_SHAPES)
def test_backward_window_kernels(gpu, B, N, L, C, dv_threads):
    """LDS-window dV / dW kernels (full, ragged and all-edge launches; the fused step switched off) vs the oracle and vs the
    generic kernels."""
    import sparsefactorization_amd as sfa
    from sparsefactorization_amd.chord import _launch_bwd
    W, V, dZ = _mk((B, N, L), 41), _mk((B, N, C), 42), _mk((B, N, C), 43)
    Wt, Vt, dZt = _t(W, gpu), _t(V, gpu), _t(dZ, gpu)
    dF, dV = oc.spmul_bwd(dZ, W, V)
    got = {}
    for variant in (0, 1):
        sfa.set_tuning("bwd_variant", variant)
        sfa.set_tuning("dv_threads", dv_threads)
        sfa.set_tuning("bwd_fused", 0)
        try:
            gW = torch.full_like(Wt, float("nan"))
            gV = torch.full_like(Vt, float("nan"))
            _launch_bwd(dZt, Wt, Vt, gW, gV, B, N, L, C, N * C, None)
            got[variant] = (gW.cpu().numpy(), gV.cpu().numpy())
        finally:
            sfa.set_tuning("bwd_variant", 0)
            sfa.set_tuning("dv_threads", 0)
            sfa.set_tuning("bwd_fused", 1)
    for variant, (gW, gV) in got.items():
        assert np.array_equal(gV, dV), f"dV variant={variant}"
        assert rel_inf(gW, dF) <= TOL, f"dW variant={variant}"


@pytest.mark.parametrize("split", [0, 2])
@pytest.mark.parametrize("B,N,L,C", [(2, 1101, 11, 8), (1, 4097, 13, 8), (2, 2000, 12, 128), (2, 777, 9, 16)])
def test_backward_window_kernels_launch_modes(gpu, B, N, L, C, split):
    """Ragged shapes: one predicated launch (fwd_split=0) and full tiles + ragged tiles in two launches (2); the default
    (1) picks between them by size and is what every other test runs."""
    import sparsefactorization_amd as sfa
    from sparsefactorization_amd.chord import _launch_bwd
    W, V, dZ = _mk((B, N, L), 61), _mk((B, N, C), 62), _mk((B, N, C), 63)
    dF, dV = oc.spmul_bwd(dZ, W, V)
    for dw_variant in (0, 1):
        sfa.set_tuning("fwd_split", split)
        sfa.set_tuning("dw_variant", dw_variant)
        try:
            gW = torch.full((B, N, L), float("nan"), device=gpu)
            gV = torch.full((B, N, C), float("nan"), device=gpu)
            _launch_bwd(_t(dZ, gpu), _t(W, gpu), _t(V, gpu), gW, gV, B, N, L, C, N * C, None)
        finally:
            sfa.set_tuning("fwd_split", 1)
            sfa.set_tuning("dw_variant", 0)
        assert np.array_equal(gV.cpu().numpy(), dV)
        assert rel_inf(gW.cpu().numpy(), dF) <= TOL


@pytest.mark.parametrize("B,N,L,C", [(2, 54, 8, 88), (3, 515, 9, 24), (2, 1031, 11, 28), (1, 433, 16, 136), (2, 300, 12, 20)])
def test_dw_window_kernel_ignores_what_earlier_kernels_left_in_lds(gpu, B, N, L, C):
    """Rows whose channel groups (C / 4) are not a power of two leave lanes of the dW window kernel idle; their window slots
    are never staged, so those lanes must drop out of the row sum by a select — a zero factor times a stale NaN is NaN (seen
    as a dW element that stayed NaN once in many runs, depending on the kernels that had used the CU before). Here the
    same kernel configuration is first run on all-NaN operands of the padded width, on enough workgroups to leave NaN in
    every LDS slot of every CU, then on the real shape."""
    import sparsefactorization_amd as sfa
    from sparsefactorization_amd.chord import _launch_bwd
    Cp = 4 << int(np.ceil(np.log2(C // 4)))  # the same lanes per row, every one of them staging
    Bp = 4096 * 8 // max(N // 8, 1) + 8
    nanV = torch.full((Bp, N, Cp), float("nan"), device=gpu)
    nanW = torch.zeros(Bp, N, L, device=gpu)
    sink = torch.empty(Bp, N, L, device=gpu)
    W, V, dZ = _mk((B, N, L), 81), _mk((B, N, C), 82), _mk((B, N, C), 83)
    dF, dV = oc.spmul_bwd(dZ, W, V)
    Wt, Vt, dZt = _t(W, gpu), _t(V, gpu), _t(dZ, gpu)
    sfa.set_tuning("dw_variant", 1)  # the window kernel for the padded width too (its default is the chunk-looping kernel)
    try:
        for _ in range(3):
            _launch_bwd(nanV, nanW, nanV, sink, None, Bp, N, L, Cp, N * Cp, None)
            gW = torch.full_like(Wt, float("nan"))
            gV = torch.full_like(Vt, float("nan"))
            _launch_bwd(dZt, Wt, Vt, gW, gV, B, N, L, C, N * C, None)
            assert np.array_equal(gV.cpu().numpy(), dV)
            assert rel_inf(gW.cpu().numpy(), dF) <= TOL
    finally:
        sfa.set_tuning("dw_variant", 0)


@pytest.mark.parametrize("tgs", [0, 4, 5])  # auto / 8 lanes per row chunk / 16
@pytest.mark.parametrize("split", [0, 1, 2])
@pytest.mark.parametrize("B,N,L,C", [(2, 2000, 12, 128), (2, 2048, 12, 64), (3, 1000, 11, 32), (1, 4097, 13, 32),
                                     (2, 515, 10, 96), (2, 700, 15, 160), (1, 16384, 15, 32), (2, 300, 20, 64)])
def test_backward_dw_chunk_kernel(gpu, B, N, L, C, split, tgs):
    """Chunk-looping dW (csrc/bwd_dw_chunk.h: rows of >= 32 channels, 8 or 16 lanes per 32- / 64-channel chunk, DPP
    row sums) vs the oracle, <= 1e-5; forced on (dw_variant=2), every launch mode, chunk counts 1..5, ragged and
    full tilings, non-power-of-two N; a view that starts 8 bytes into an allocation (element-wise tile ends)."""
    import sparsefactorization_amd as sfa
    from sparsefactorization_amd.chord import _launch_bwd
    if tgs == 5 and (C // 4) % 16 != 0:
        pytest.skip("16 lanes per chunk need C % 64 == 0")
    W, V, dZ = _mk((B, N, L), 71), _mk((B, N, C), 72), _mk((B, N, C), 73)
    dF, _ = oc.spmul_bwd(dZ, W, V)
    for shift in (0, 2):
        gbuf = torch.full((B * N * L + 8,), float("nan"), device=gpu)
        gW = gbuf[shift:shift + B * N * L].view(B, N, L)
        sfa.set_tuning("dw_variant", 2)
        sfa.set_tuning("dw_tgs", tgs)
        sfa.set_tuning("fwd_split", split)
        try:
            _launch_bwd(_t(dZ, gpu), _t(W, gpu), _t(V, gpu), gW, None, B, N, L, C, N * C, None)
        finally:
            sfa.set_tuning("dw_variant", 0)
            sfa.set_tuning("dw_tgs", 0)
            sfa.set_tuning("fwd_split", 1)
        assert rel_inf(gW.cpu().numpy(), dF) <= TOL, shift
        assert torch.isnan(gbuf[:shift]).all() and torch.isnan(gbuf[shift + B * N * L:]).all()


def test_backward_dw_chunk_kernel_is_the_default_for_wide_rows(gpu):
    """dw_variant=2 raises where the chunk kernel does not apply, so: it applies at C = 32 / 128 and not at C = 8."""
    import sparsefactorization_amd as sfa
    from sparsefactorization_amd.chord import _launch_bwd
    sfa.set_tuning("dw_variant", 2)
    try:
        for C, ok in ((32, True), (128, True), (8, False), (36, False)):
            B, N, L = 1, 512, 10
            args = [torch.zeros(B, N, C, device=gpu), torch.zeros(B, N, L, device=gpu), torch.zeros(B, N, C, device=gpu),
                    torch.empty(B, N, L, device=gpu), None, B, N, L, C, N * C, None]
            if ok:
                _launch_bwd(*args)
            else:
                with pytest.raises(RuntimeError, match="dw_variant=2"):
                    _launch_bwd(*args)
    finally:
        sfa.set_tuning("dw_variant", 0)


@pytest.mark.parametrize("wide", [1, 2])
@pytest.mark.parametrize("B,N,L,C", [(2, 2000, 12, 128), (2, 640, 10, 260), (2, 2048, 12, 64)])
def test_backward_dv_wide_row_configs(gpu, B, N, L, C, wide):
    import sparsefactorization_amd as sfa
    from sparsefactorization_amd.chord import _launch_bwd
    W, V, dZ = _mk((B, N, L), 47), _mk((B, N, C), 48), _mk((B, N, C), 49)
    sfa.set_tuning("fwd_wide", wide)
    try:
        gV = torch.full((B, N, C), float("nan"), device=gpu)
        _launch_bwd(_t(dZ, gpu), _t(W, gpu), _t(V, gpu), None, gV, B, N, L, C, N * C, None)
    finally:
        sfa.set_tuning("fwd_wide", 0)
    assert np.array_equal(gV.cpu().numpy(), oc.spmul_bwd(dZ, W, V)[1])


def test_backward_window_misaligned_buffers(gpu):
    """W and dW as views that start 4 bytes into an allocation (all-edge launches, partial 16-byte chunks)."""
    import sparsefactorization_amd as sfa
    from sparsefactorization_amd.chord import _launch_bwd
    B, N, L, C = 3, 1101, 11, 8
    W, V, dZ = _mk((B, N, L), 44), _mk((B, N, C), 45), _mk((B, N, C), 46)
    backing = torch.zeros(B * N * L + 8, device=gpu)
    Wt = backing[1:1 + B * N * L].view(B, N, L)
    Wt.copy_(_t(W, gpu))
    gbuf = torch.full((B * N * L + 8,), float("nan"), device=gpu)
    gW = gbuf[3:3 + B * N * L].view(B, N, L)
    gV = torch.empty(B, N, C, device=gpu)
    _launch_bwd(_t(dZ, gpu), Wt, _t(V, gpu), gW, gV, B, N, L, C, N * C, None)
    dF, dV = oc.spmul_bwd(dZ, W, V)
    assert np.array_equal(gV.cpu().numpy(), dV)
    assert rel_inf(gW.cpu().numpy(), dF) <= TOL
    assert torch.isnan(gbuf[:3]).all() and torch.isnan(gbuf[3 + B * N * L:]).all()  # nothing written outside


def test_backward_broadcast_V(gpu):
    import sparsefactorization_amd as sfa
    B, N, L, C = 3, 128, 8, 16
    W, V, dZ = _mk((B, N, L), 18), _mk((N, C), 19), _mk((B, N, C), 20)
    Wt = _t(W, gpu).requires_grad_(True)
    Vt = _t(V, gpu).requires_grad_(True)
    sfa.chord_spmm(Wt, Vt).backward(_t(dZ, gpu))
    dF, dV = oc.spmul_bwd(dZ, W, V)
    assert rel_inf(Wt.grad.cpu().numpy(), dF) <= TOL
    assert rel_inf(Vt.grad.cpu().numpy(), dV.sum(0)) <= TOL


def test_gradcheck_f64(gpu):
    import sparsefactorization_amd as sfa
    g = torch.Generator().manual_seed(21)
    W = torch.randn(2, 24, 5, dtype=torch.float64, generator=g).to(gpu).requires_grad_(True)
    V = torch.randn(2, 24, 3, dtype=torch.float64, generator=g).to(gpu).requires_grad_(True)
    R = torch.randn(2, 24, 3, dtype=torch.float64, generator=g).to(gpu).requires_grad_(True)
    assert torch.autograd.gradcheck(lambda w, v, r: sfa.chord_spmm(w, v, r), (W, V, R), eps=1e-6, atol=1e-7)
    assert torch.autograd.gradcheck(lambda w1, w2, v: sfa.chord_chain([w1, w2], v, True),
                                    (W, W.detach().clone().requires_grad_(True), V), eps=1e-6, atol=1e-7)


def test_sparse_multiply_dropin(gpu):
    """SparseMultiply.apply(F, V, offsets, n_block, ...) — spmul/spmul.py:12-31."""
    from sparsefactorization_amd.spmul import SparseMultiply, get_offsets
    B, N, L, C = 2, 200, 8, 12
    F, V, dZ = _mk((B, N, L), 22), _mk((B, N, C), 23), _mk((B, N, C), 24)
    Ft, Vt = _t(F, gpu).requires_grad_(True), _t(V, gpu).requires_grad_(True)
    off = get_offsets(L)
    assert off.tolist() == [0, 1, 2, 4, 8, 16, 32, 64]
    Z = SparseMultiply.apply(Ft, Vt, off, 16, 64, 16, 16)
    assert np.array_equal(Z.detach().cpu().numpy(), oc.spmul_fwd(F, V))
    Z.backward(_t(dZ, gpu))
    dF, dV = oc.spmul_bwd(dZ, F, V)
    assert rel_inf(Ft.grad.cpu().numpy(), dF) <= TOL
    assert np.array_equal(Vt.grad.cpu().numpy(), dV)


# ---------------------------------------------------------------------------------------------------
# golden fixtures: operands captured from the reference's PSFNet.forward
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("fixture,residual", [("psfnet_adding_n128.npz", True), ("psfnet_order_n128.npz", True),
                                              ("lra_pathfinder_ckpt.npz", False)])
def test_chain_against_reference_fixture(gpu, fixture, residual):
    import sparsefactorization_amd as sfa
    g = load_golden(fixture)
    W, V0 = g["W"], g["V0"]
    Ws = [_t(W[m], gpu) for m in range(W.shape[0])]
    got = sfa.chord_chain(Ws, _t(V0, gpu), residual).cpu().numpy()
    assert rel_inf(got, g["Vfin"]) <= TOL
    # and bit-exact against the oracle's chain
    M, B, N, L = W.shape
    rows, cols = oc.chord_indices(N, L)
    assert np.array_equal(got, oc.chain(np.stack([rows, cols]), W, V0, residual)[-1])
    # step-by-step API gives the same bits as the fused call
    V = _t(V0, gpu)
    res = V if residual else None
    for w in Ws:
        V = sfa.chord_spmm(w, V, res)
    assert np.array_equal(V.cpu().numpy(), got)


@pytest.mark.parametrize("fixture", ["psfnet_adding_n128.npz", "psfnet_order_n128.npz"])
def test_chain_gradients_against_reference_autograd(gpu, fixture):
    import sparsefactorization_amd as sfa
    g = load_golden(fixture)
    W, V0 = g["W"], g["V0"]
    Ws = [_t(W[m], gpu).requires_grad_(True) for m in range(W.shape[0])]
    V0t = _t(V0, gpu).requires_grad_(True)
    out = sfa.chord_chain(Ws, V0t, True)
    out.backward(_t(g["gVfin"], gpu))
    dW = np.stack([w.grad.cpu().numpy() for w in Ws])
    assert rel_inf(dW, g["dW"]) <= TOL
    assert rel_inf(V0t.grad.cpu().numpy(), g["dV0"]) <= TOL
    # unfused path agrees
    Ws2 = [_t(W[m], gpu).requires_grad_(True) for m in range(W.shape[0])]
    V02 = _t(V0, gpu).requires_grad_(True)
    V = V02
    for w in Ws2:
        V = sfa.chord_spmm(w, V, V02)
    V.backward(_t(g["gVfin"], gpu))
    assert rel_inf(np.stack([w.grad.cpu().numpy() for w in Ws2]), g["dW"]) <= TOL
    assert rel_inf(V02.grad.cpu().numpy(), g["dV0"]) <= TOL


CHAIN_SHAPES = [  # (B, N, M, L, C, residual)
    (40, 128, 7, 8, 8, True), (3, 2000, 11, 12, 16, False), (2, 1024, 11, 12, 32, True), (2, 2048, 5, 12, 8, True),
    (5, 777, 4, 9, 12, True), (1, 64, 3, 7, 4, False), (2, 1500, 6, 20, 8, True), (2, 300, 9, 3, 24, False),
    (3, 1025, 2, 11, 8, True), (2, 2049, 3, 12, 8, True), (1, 100, 12, 9, 260, True),
    (2, 1025, 11, 12, 32, False), (2, 2049, 4, 13, 64, True), (2, 1056, 3, 12, 16, True),  # three rows per thread (CLS-token lengths)
]


@pytest.mark.parametrize("B,N,M,L,C,residual", CHAIN_SHAPES)
def test_fused_lds_chain_matches_oracle_and_per_step(gpu, B, N, M, L, C, residual):
    """The one-launch LDS-resident chain (N*CC <= 2048) against the oracle, bit for bit, in both storage modes:
    every step kept (training) and two ping-pong buffers (inference); also equal to the per-step kernels."""
    import sparsefactorization_amd as sfa
    W = _mk((M, B, N, L), 61, 0.4)
    V0 = _mk((B, N, C), 62)
    rows, cols = oc.chord_indices(N, L)
    want = oc.chain(np.stack([rows, cols]), W, V0, residual)  # [M, B, N, C]
    Ws = [_t(W[m], gpu) for m in range(M)]
    got = {}
    for fused in (2, 0):  # 2: the single launch wherever it fits (the default, 1, leaves rows of > 64 channels to the steps)
        sfa.set_tuning("chain_fused", fused)
        try:
            with torch.no_grad():
                got[fused] = sfa.chord_chain(Ws, _t(V0, gpu), residual).cpu().numpy()       # ping-pong storage
            Wg = [w.clone().requires_grad_(True) for w in Ws]
            out = sfa.chord_chain(Wg, _t(V0, gpu), residual)                                # every step stored
            steps = [t for t in out.grad_fn.saved_tensors][1 + M:2 * M]
            for m, t in enumerate(steps):
                assert np.array_equal(t.cpu().numpy(), want[m]), f"fused={fused} stored step {m}"
            assert np.array_equal(out.detach().cpu().numpy(), want[-1])
        finally:
            sfa.set_tuning("chain_fused", 1)
    assert np.array_equal(got[2], want[-1])
    assert np.array_equal(got[0], want[-1])


BIG_CHAIN_SHAPES = [  # (B, N, M, L, C, residual): 1057 <= N <= 2048, two channel groups per workgroup, a thread owns both
    (2, 2000, 11, 12, 128, True), (3, 2048, 5, 12, 8, True), (2, 1057, 4, 11, 16, False), (2, 2001, 3, 12, 12, True),
    (1, 1500, 6, 20, 24, True), (2, 2047, 3, 2, 8, False), (2, 1999, 4, 15, 20, True),
]


@pytest.mark.parametrize("B,N,M,L,C,residual", BIG_CHAIN_SHAPES)
def test_fused_lds_chain_large_instance(gpu, B, N, M, L, C, residual):
    """chord_chain_rows_k with two channel groups per thread (knob chain_cc = 2 forces it; automatic from 256 workgroups per launch on): bit-equal to the oracle
    and to the one-group-per-workgroup launch, with every step kept and with ping-pong storage; an odd number of channel
    groups (C = 12, 20) leaves the last workgroup one group."""
    import sparsefactorization_amd as sfa
    from sparsefactorization_amd import _lib
    W = _mk((M, B, N, L), 71, 0.4)
    V0 = _mk((B, N, C), 72)
    rows, cols = oc.chord_indices(N, L)
    want = oc.chain(np.stack([rows, cols]), W, V0, residual)
    Ws = [_t(W[m], gpu) for m in range(M)]
    got = {}
    sfa.set_tuning("chain_fused", 2)
    try:
        for cc in (2, 1):
            sfa.set_tuning("chain_cc", cc)
            assert ("chord_chain_rows_k" in _lib.describe_chain_fwd(B, N, L, C, M)) == (cc == 2)
            with torch.no_grad():
                got[cc] = sfa.chord_chain(Ws, _t(V0, gpu), residual).cpu().numpy()
            Wg = [w.clone().requires_grad_(True) for w in Ws]
            out = sfa.chord_chain(Wg, _t(V0, gpu), residual)
            steps = [t for t in out.grad_fn.saved_tensors][1 + M:2 * M]
            for m, t in enumerate(steps):
                assert np.array_equal(t.cpu().numpy(), want[m]), f"chain_cc={cc} stored step {m}"
            assert np.array_equal(out.detach().cpu().numpy(), want[-1])
    finally:
        sfa.set_tuning("chain_cc", 0)
        sfa.set_tuning("chain_fused", 1)
    assert np.array_equal(got[2], want[-1]) and np.array_equal(got[1], want[-1])


LONG_CHAIN_SHAPES = [  # (B, N, M, L, C, residual): 2113 <= N <= 4160, one channel group, five rows per thread
    (2, 4097, 12, 14, 32, True), (2, 4096, 4, 13, 8, False), (1, 2113, 3, 12, 12, True), (2, 4160, 3, 13, 4, True), (2, 3000, 5, 20, 8, False),
]


@pytest.mark.parametrize("B,N,M,L,C,residual", LONG_CHAIN_SHAPES)
def test_fused_lds_chain_long_instance(gpu, B, N, M, L, C, residual):
    """chord_chain_rows_k with one channel group and five rows per thread (the LRA text task's N = 4096 + 1; knob chain_cc = 2
    forces it; automatic from 256 workgroups per launch on): bit-equal to the oracle and to the per-step kernels, with every
    step kept and with ping-pong storage."""
    import sparsefactorization_amd as sfa
    from sparsefactorization_amd import _lib
    W = _mk((M, B, N, L), 73, 0.4)
    V0 = _mk((B, N, C), 74)
    rows, cols = oc.chord_indices(N, L)
    want = oc.chain(np.stack([rows, cols]), W, V0, residual)
    Ws = [_t(W[m], gpu) for m in range(M)]
    sfa.set_tuning("chain_fused", 2)
    sfa.set_tuning("chain_cc", 2)
    try:
        assert "chord_chain_rows_k<f32,L=%d,G=1,R=5>" % L in _lib.describe_chain_fwd(B, N, L, C, M)
        with torch.no_grad():
            got = sfa.chord_chain(Ws, _t(V0, gpu), residual).cpu().numpy()
        Wg = [w.clone().requires_grad_(True) for w in Ws]
        out = sfa.chord_chain(Wg, _t(V0, gpu), residual)
        steps = [t for t in out.grad_fn.saved_tensors][1 + M:2 * M]
        for m, t in enumerate(steps):
            assert np.array_equal(t.cpu().numpy(), want[m]), f"stored step {m}"
        assert np.array_equal(out.detach().cpu().numpy(), want[-1])
    finally:
        sfa.set_tuning("chain_cc", 0)
        sfa.set_tuning("chain_fused", 1)
    assert np.array_equal(got, want[-1])
    with torch.no_grad():  # (few sequences: the automatic rule leaves these to the per-step kernels)
        assert np.array_equal(sfa.chord_chain(Ws, _t(V0, gpu), residual).cpu().numpy(), want[-1])


def test_fused_lds_chain_large_instance_rule(gpu):
    """Automatic: the large instance only when the launch keeps at least 256 workgroups (one per CU)."""
    from sparsefactorization_amd import _lib
    assert "rows_k<f32,L=12,G=2" in _lib.describe_chain_fwd(32, 2000, 12, 128, 11)       # ListOps: 32 x 16 workgroups
    assert "rows_k" not in _lib.describe_chain_fwd(40, 2048, 12, 8, 11)     # 40 workgroups: two per sequence stay
    assert "rows_k" not in _lib.describe_chain_fwd(32, 1024, 11, 32, 10)    # N <= 1056: the two-group instance of old
    assert "rows_k<f32,L=14,G=1,R=5>" in _lib.describe_chain_fwd(32, 4097, 14, 32, 12)   # the LRA text task: 32 x 8 workgroups
    assert "rows_k" not in _lib.describe_chain_fwd(8, 4097, 14, 32, 12)


BWD_CHAIN_SHAPES = [  # (B, N, M, L, C, residual, explicit offsets): N <= 1024, C = 4 or 8 — the one-launch backward chain
    (40, 128, 7, 8, 8, True, None), (3, 1024, 10, 11, 8, True, None), (2, 777, 4, 9, 4, False, None), (2, 64, 3, 7, 8, False, None),
    (1, 1000, 5, 20, 8, True, None), (2, 300, 9, 3, 4, True, None), (2, 256, 4, 6, 8, True, [0, 5, 255, 300, -1, 128]), (3, 1, 2, 2, 8, True, None),
]


@pytest.mark.parametrize("B,N,M,L,C,residual,offsets", BWD_CHAIN_SHAPES)
def test_backward_chain_in_one_launch(gpu, B, N, M, L, C, residual, offsets):
    """chord_chain_bwd_lds_k behind psf_chord_chain_bwd_f32 (what _ChordChain.backward runs for short sequences of narrow rows):
    dV0 and every dW_m bit-equal to the oracle's per-step backward (the residual terms summed left to right as the per-step
    path does), and dV0 bit-equal / dW within 1e-6 of the per-step kernels (knob chain_bwd_fused = 0)."""
    import sparsefactorization_amd as sfa
    from sparsefactorization_amd import _lib
    assert _lib.load().psf_chord_chain_bwd_supported(N, L, C, M) == 1
    W = _mk((M, B, N, L), 81, 0.4)
    V0 = _mk((B, N, C), 82)
    gout = _mk((B, N, C), 83)
    X = [V0]
    for m in range(M):
        nxt = oc.spmul_fwd(W[m], X[-1], offsets)
        X.append((nxt + V0).astype(np.float32) if residual else nxt)
    g, want_dW, terms = gout, [None] * M, []
    for m in range(M - 1, -1, -1):
        terms.append(g)
        want_dW[m], g = oc.spmul_bwd(g, W[m], X[m], offsets)
    want_dV0 = g
    if residual:
        acc = terms[0]
        for t in terms[1:] + [g]:
            acc = (acc + t).astype(np.float32)
        want_dV0 = acc
    got = {}
    for knob in (1, 0):
        sfa.set_tuning("chain_bwd_fused", knob)
        try:
            Wg = [_t(W[m], gpu).requires_grad_(True) for m in range(M)]
            Vg = _t(V0, gpu).requires_grad_(True)
            out = sfa.chord_chain(Wg, Vg, residual, offsets=offsets)
            assert np.array_equal(out.detach().cpu().numpy(), X[-1])
            out.backward(_t(gout, gpu))
            got[knob] = (Vg.grad.cpu().numpy(), [w.grad.cpu().numpy() for w in Wg])
        finally:
            sfa.set_tuning("chain_bwd_fused", 1)
    assert np.array_equal(got[1][0], want_dV0)
    for m in range(M):
        assert np.array_equal(got[1][1][m], want_dW[m]), f"dW_{m}"
    assert np.array_equal(got[0][0], got[1][0])
    for m in range(M):
        scale = max(np.abs(want_dW[m]).max(), 1e-30)
        assert np.abs(got[0][1][m] - got[1][1][m]).max() / scale <= 1e-6


def test_backward_chain_in_one_launch_rule_and_partial_gradients(gpu):
    """Beyond N = 1024 or C = 8 the library says PSF_E_UNSUPPORTED and the steps run; a chain in which only some W_m (or only V0)
    need gradients takes the per-step path too — same results as with the knob off."""
    import sparsefactorization_amd as sfa
    from sparsefactorization_amd import _lib
    lib = _lib.load()
    assert lib.psf_chord_chain_bwd_supported(1024, 11, 8, 10) == 1 and lib.psf_chord_chain_bwd_supported(1025, 12, 8, 10) == 0
    assert lib.psf_chord_chain_bwd_supported(128, 8, 16, 7) == 0 and lib.psf_chord_chain_bwd_supported(128, 21, 8, 7) == 0
    B, N, M, L, C = 2, 256, 3, 9, 8
    W = [_t(_mk((B, N, L), 84 + m, 0.4), gpu) for m in range(M)]
    V0 = _t(_mk((B, N, C), 88), gpu)
    res = {}
    for knob in (1, 0):
        sfa.set_tuning("chain_bwd_fused", knob)
        try:
            Wg = [w.clone().requires_grad_(m != 1) for m, w in enumerate(W)]
            out = sfa.chord_chain(Wg, V0, True)
            out.sum().backward()
            res[knob] = [None if w.grad is None else w.grad.clone() for w in Wg]
        finally:
            sfa.set_tuning("chain_bwd_fused", 1)
    assert res[1][1] is None and res[0][1] is None
    for m in (0, 2):
        assert torch.equal(res[1][m], res[0][m])


@pytest.mark.parametrize("B,N,M,L,C,residual", [(2, 2000, 5, 12, 16, True), (2, 1025, 4, 11, 32, False), (1, 4097, 3, 13, 8, True),
                                                 (2, 300, 33, 9, 8, True), (2, 300, 33, 9, 12, False), (2, 300, 33, 9, 12, True)])
def test_backward_chain_steps_issued_by_the_library(gpu, B, N, M, L, C, residual):
    """Shapes the one-launch kernel does not cover: psf_chord_chain_bwd_f32 issues the per-step kernels and the residual sum itself.
    Same kernels in the same order as the Python loop (knob chain_bwd_fused = 0): every gradient bit-equal. A residual chain
    of more than 31 steps is beyond psf_sum_tensors_f32 and comes back PSF_E_UNSUPPORTED: the loop runs, same result."""
    import sparsefactorization_amd as sfa
    from sparsefactorization_amd import _lib
    if not (N <= 1024 and C <= 8):
        assert _lib.load().psf_chord_chain_bwd_supported(N, L, C, M) == 0
    W = [_t(_mk((B, N, L), 90 + m, 0.2), gpu) for m in range(M)]
    V0 = _t(_mk((B, N, C), 89), gpu)
    gout = _t(_mk((B, N, C), 88), gpu)
    res = {}
    for knob in (1, 0):
        sfa.set_tuning("chain_bwd_fused", knob)
        try:
            Wg = [w.clone().requires_grad_(True) for w in W]
            Vg = V0.clone().requires_grad_(True)
            sfa.chord_chain(Wg, Vg, residual).backward(gout)
            res[knob] = [Vg.grad.clone()] + [w.grad.clone() for w in Wg]
        finally:
            sfa.set_tuning("chain_bwd_fused", 1)
    if N <= 1024 and C <= 8:   # (the one-launch kernel: dV0 the same bits, dW the oracle's bits — within 1e-6 of the step kernels')
        assert torch.equal(res[1][0], res[0][0])
        for a, b in zip(res[1][1:], res[0][1:]):
            assert float((a - b).abs().max() / b.abs().max().clamp_min(1e-30)) <= 1e-6
    else:
        for a, b in zip(res[1], res[0]):
            assert torch.equal(a, b)


def test_fused_lds_chain_custom_offsets_and_broadcast(gpu):
    import sparsefactorization_amd as sfa
    B, N, M, L = 3, 256, 4, 6
    W = _mk((M, B, N, L), 63, 0.5)
    off = [0, 5, 255, 300, -1, 128]
    eye = np.eye(N, dtype=np.float32)
    Ws = [_t(W[m], gpu) for m in range(M)]
    sfa.set_tuning("chain_fused", 2)  # 64 workgroups per sequence: beyond the default's limit
    try:
        got = sfa.chord_chain(Ws, _t(eye, gpu), False, offsets=off).cpu().numpy()
    finally:
        sfa.set_tuning("chain_fused", 1)
    X = np.broadcast_to(eye, (B, N, N)).copy()
    for m in range(M):
        X = oc.spmul_fwd(W[m], X, off)
    assert np.array_equal(got, X)


def test_attention_map_against_fixture(gpu):
    """W_M ... W_1 on eye(N): C = N = 1024, unbatched first operand (cfg4)."""
    import sparsefactorization_amd as sfa
    g = load_golden("lra_pathfinder_ckpt.npz")
    W = g["W"]
    N = W.shape[2]
    Ws = [_t(W[m], gpu) for m in range(W.shape[0])]
    Wf = sfa.chord_chain(Ws, torch.eye(N, device=gpu), False).cpu().numpy()
    assert rel_inf(Wf[0, ::16, :], g["Wfinal_rows"]) <= TOL
    assert rel_inf(Wf.sum(-1), g["Wfinal_rowsum"]) <= TOL


# ---------------------------------------------------------------------------------------------------
# full BASELINE size (cfg2): size-independent properties
# ---------------------------------------------------------------------------------------------------
def _cfg2(gpu, B=64):
    N, M, L, C = 16384, 14, 15, 8
    g = torch.Generator(device=gpu)
    Ws = []
    for i in range(M):
        g.manual_seed(1234 + i + 1)
        Ws.append(0.1 * torch.randn(B, N, L, device=gpu, generator=g))
    g.manual_seed(1234)
    V0 = torch.randn(B, N, C, device=gpu, generator=g)
    return Ws, V0


def test_full_size_properties(gpu):
    import sparsefactorization_amd as sfa
    Ws, V0 = _cfg2(gpu)
    out = sfa.chord_chain(Ws, V0, True)
    assert torch.isfinite(out).all()
    # determinism: no atomics anywhere -> bitwise repeatable
    assert torch.equal(out, sfa.chord_chain(Ws, V0, True))
    # fused chain == step-by-step
    V = V0
    for w in Ws:
        V = sfa.chord_spmm(w, V, V0)
    assert torch.equal(V, out)
    # generic kernel == window kernel, bit for bit
    sfa.set_tuning("fwd_variant", 1)
    try:
        assert torch.equal(sfa.chord_chain(Ws, V0, True), out)
    finally:
        sfa.set_tuning("fwd_variant", 0)
    # linearity in V (no residual): chain(a*X + Y) == a*chain(X) + chain(Y) up to rounding
    X, Y = V0, torch.roll(V0, 3, 0)
    a = 0.5
    lhs = sfa.chord_chain(Ws, a * X + Y, False)
    rhs = a * sfa.chord_chain(Ws, X, False) + sfa.chord_chain(Ws, Y, False)
    assert float((lhs - rhs).abs().max() / rhs.abs().max()) <= TOL
    # spot-check 3 batch elements of the full-size chain against the oracle: the first, one in the middle and the
    # LAST one (last XCD's range of the tile remap, tail of the zigzag walk)
    sel = [0, 37, 63]
    W_np = np.stack([w[sel].cpu().numpy() for w in Ws])
    rows, cols = oc.chord_indices(16384, 15)
    want = oc.chain(np.stack([rows, cols]), W_np, V0[sel].cpu().numpy(), True)[-1]
    assert np.array_equal(out[sel].cpu().numpy(), want)


@pytest.mark.parametrize("B,N,L,C", [(8, 16384, 15, 8), (4, 16385, 15, 8), (4, 4097, 13, 32), (4, 16384, 15, 32)])
def test_full_size_shift_equivariance_and_identity(gpu, B, N, L, C):
    """The pattern is circulant: rolling W and V along N rolls the output. W = e_0 (self link only) is identity. At
    BASELINE.json's sequence length and at the 2^k + 1 lengths a CLS-token column makes (aligned and general instances of the
    forward and backward kernels)."""
    import sparsefactorization_amd as sfa
    g = torch.Generator(device=gpu).manual_seed(5)
    W = torch.randn(B, N, L, device=gpu, generator=g)
    V = torch.randn(B, N, C, device=gpu, generator=g)
    out = sfa.chord_spmm(W, V)
    s = 4099
    out_s = sfa.chord_spmm(torch.roll(W, s, 1), torch.roll(V, s, 1))
    assert torch.equal(torch.roll(out, s, 1), out_s)
    Wid = torch.zeros(B, N, L, device=gpu)
    Wid[:, :, 0] = 1.0
    assert torch.equal(sfa.chord_spmm(Wid, V), V)
    # adjoint identity <dZ, W.V> == <W^T.dZ, V> ties dV to the forward kernel at full size
    dZ = torch.randn(B, N, C, device=gpu, generator=g)
    Vr = V.clone().requires_grad_(True)
    sfa.chord_spmm(W, Vr).backward(dZ)
    lhs = (dZ.double() * out.double()).sum()
    rhs = (Vr.grad.double() * V.double()).sum()
    assert abs(float(lhs - rhs)) <= 1e-6 * abs(float(lhs))
    # both gradients of one step (the fused backward kernels): <dW, W> == <dZ, W.V> as well (the step is bilinear), and the
    # gradients roll with their operands
    Wr, Vr = W.clone().requires_grad_(True), V.clone().requires_grad_(True)
    sfa.chord_spmm(Wr, Vr).backward(dZ)
    assert abs(float((Wr.grad.double() * W.double()).sum() - lhs)) <= 1e-5 * abs(float(lhs))
    Ws, Vs = torch.roll(W, s, 1).requires_grad_(True), torch.roll(V, s, 1).requires_grad_(True)
    sfa.chord_spmm(Ws, Vs).backward(torch.roll(dZ, s, 1))
    assert torch.equal(torch.roll(Vr.grad, s, 1), Vs.grad)
    assert float((torch.roll(Wr.grad, s, 1) - Ws.grad).abs().max() / Wr.grad.abs().max()) <= TOL


def test_errors(gpu):
    import sparsefactorization_amd as sfa
    W = torch.zeros(2, 16, 4, device=gpu)
    V = torch.zeros(2, 16, 8, device=gpu)
    with pytest.raises(RuntimeError):
        sfa.chord_spmm(W.cpu(), V.cpu())  # no CPU path
    with pytest.raises(TypeError):
        sfa.chord_spmm(W.half(), V.half())
    with pytest.raises(ValueError):
        sfa.chord_spmm(W, torch.zeros(3, 16, 8, device=gpu))
    idx = torch.tensor(sfa.get_chord_indices_assym(16, 4)).to(gpu)
    bad = idx.clone()
    bad[1, 5] = (bad[1, 5] + 1) % 16
    with pytest.raises(ValueError):
        sfa.spmm(bad, W.reshape(2, 64), 16, 16, V)
    with pytest.raises(ValueError):
        sfa.spmm(idx, W.reshape(2, 64), 16, 8, V)


@pytest.mark.parametrize("count", [1, 2, 7, 8, 9, 15, 16, 17, 23, 31, 32])
def test_sum_tensors_is_the_left_to_right_sum(gpu, count):
    """psf_sum_tensors_f32 (csrc/sum_tensors.hip): ((t0 + t1) + t2) + ... bit for bit, for every pass structure
    (16 sources in the first pass, 15 per later pass) — the residual gradient of the chain is summed with it."""
    from sparsefactorization_amd.chord import _sum_tensors
    g = torch.Generator(device=gpu).manual_seed(count)
    terms = [torch.randn(3, 1001, 8, device=gpu, generator=g) * (10.0 ** (i % 5 - 2)) for i in range(count)]
    want = terms[0].clone()
    for t in terms[1:]:
        want = want + t
    got = _sum_tensors(terms)
    assert torch.equal(got, want)
    assert all(torch.equal(t, u) for t, u in zip(terms, [x.clone() for x in terms]))  # sources untouched


def test_chain_backward_residual_gradient_one_pass_sum(gpu):
    """dV0 of a residual chain (summed once at the end) equals the step-by-step autograd through chord_spmm bit for bit."""
    import sparsefactorization_amd as sfa
    B, N, M, C = 2, 3000, 11, 8
    Ws = [_t(_mk((B, N, M + 1), 80 + m, 0.3), gpu).requires_grad_(True) for m in range(M)]
    V0a = _t(_mk((B, N, C), 99), gpu).requires_grad_(True)
    V0b = V0a.detach().clone().requires_grad_(True)
    dZ = _t(_mk((B, N, C), 98), gpu)
    sfa.chord_chain(Ws, V0a, True).backward(dZ)
    gW = [w.grad.clone() for w in Ws]
    for w in Ws:
        w.grad = None
    V = V0b
    for w in Ws:
        V = sfa.chord_spmm(w, V, V0b)
    V.backward(dZ)
    assert all(torch.equal(a, w.grad) for a, w in zip(gW, Ws))
    assert rel_inf(V0a.grad.cpu().numpy(), V0b.grad.cpu().numpy()) <= 1e-6  # autograd adds the M+1 terms in its own order


LIMIT_SHAPES = [
    # the ends of what the window kernels are compiled for: L = 4 (N = 8) and L = 20 (N = 2^19), the longest sequences with the
    # narrowest and with 32-channel rows, a batch of one, 64-bit element offsets between batch elements (B N L > 2^31 is out of
    # reach of a test; B N C 4 bytes > 2^31 is not: 2 x 2^19 x 512 x 4 = 2.1 GB)
    (3, 8, 4, 4), (1, 1 << 19, 20, 4), (1, 1 << 18, 19, 32), (2, 1 << 16, 17, 8), (1, (1 << 16) + 1, 17, 8), (2, 1 << 19, 20, 512),
]


@pytest.mark.parametrize("B,N,L,C", LIMIT_SHAPES)
def test_limit_shapes_forward_and_backward(gpu, B, N, L, C):
    """Forward step (+ residual) and both gradients at the ends of the compiled ranges, against the oracle (sampled rows for the
    2 GB case: the oracle is a scalar C loop): forward and dV bit for bit, dW to the parity bar."""
    import sparsefactorization_amd as sfa
    from sparsefactorization_amd.chord import _launch_bwd
    big = B * N * C > (1 << 27)
    g = torch.Generator(device=gpu).manual_seed(5)
    Wt = 0.2 * torch.randn(B, N, L, device=gpu, generator=g)
    Vt = torch.randn(B, N, C, device=gpu, generator=g)
    Rt = torch.randn(B, N, C, device=gpu, generator=g)
    out = sfa.chord_spmm(Wt, Vt, Rt)
    dZt = torch.randn(B, N, C, device=gpu, generator=g)
    gW = torch.full((B, N, L), float("nan"), device=gpu)
    gV = torch.full((B, N, C), float("nan"), device=gpu)
    _launch_bwd(dZt, Wt, Vt, gW, gV, B, N, L, C, N * C, None)
    assert torch.isfinite(out).all() and torch.isfinite(gW).all() and torch.isfinite(gV).all()
    if not big:
        W, V, R, dZ = (t.cpu().numpy() for t in (Wt, Vt, Rt, dZt))
        assert np.array_equal(out.cpu().numpy(), _oracle_fwd(W, V, R))
        dF, dV = oc.spmul_bwd(dZ, W, V)
        assert np.array_equal(gV.cpu().numpy(), dV)
        assert rel_inf(gW.cpu().numpy(), dF) <= TOL
        return
    # sampled rows of the last batch element (its rows lie beyond 2^31 bytes), in float64 from the definition
    off = [0] + [(1 << k) % N for k in range(L - 1)]
    b = B - 1
    rows = torch.tensor([0, 1, 255, 256, N // 2 - 1, N // 2, N - 257, N - 1], device=gpu)
    W64, V64, dZ64 = Wt[b].double(), Vt[b].double(), dZt[b].double()
    want = Rt[b, rows].double()
    wantW = torch.zeros(len(rows), L, dtype=torch.float64, device=gpu)
    wantV = torch.zeros(len(rows), C, dtype=torch.float64, device=gpu)
    for k, o in enumerate(off):
        src = (rows + o) % N
        want = want + W64[rows, k, None] * V64[src]
        wantW[:, k] = (dZ64[rows] * V64[src]).sum(-1)
        back = (rows - o) % N
        wantV = wantV + W64[back, k, None] * dZ64[back]
    for got, ref in ((out[b, rows], want), (gW[b, rows], wantW), (gV[b, rows], wantV)):
        assert float((got.double() - ref).abs().max() / ref.abs().max()) <= TOL


def _special(shape, seed, values, frac=0.02):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal(shape).astype(np.float32)
    flat = x.reshape(-1)
    n = max(4, int(frac * flat.size))
    flat[rng.choice(flat.size, n, replace=False)] = np.asarray(values, dtype=np.float32)[rng.integers(0, len(values), n)]
    return x


def _same_bits_or_both_nan(a, b):
    return bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))))


@pytest.mark.parametrize("B,N,L,C", [(2, 2048, 12, 8), (2, 1025, 11, 32), (2, 300, 9, 20), (1, 16384, 15, 8), (2, 2000, 12, 128),
                                     (3, 128, 8, 8)])
def test_special_values_propagate_as_in_the_oracle(gpu, B, N, L, C):
    """NaN, +-Inf, denormals, the smallest normal, +-0 and values near FLT_MAX scattered over W, V, the residual and dZ: the
    forward step and dV are the oracle's bits (a NaN where the oracle has one, the same denormal, the same sign of zero) on
    every kernel family; dW, whose row sums run in another order, has the oracle's NaN / +Inf / -Inf pattern exactly and meets
    the bar elsewhere as long as no partial sum can overflow (the second value set); with values near FLT_MAX in play an
    overflow in one summation order and not in the other is legitimate, and only elements finite on both sides are compared."""
    import sparsefactorization_amd as sfa
    from sparsefactorization_amd.chord import _launch_bwd
    tame = [np.nan, np.inf, -np.inf, 1e-42, -3e-45, 0.0, -0.0, 1.17549435e-38, 2e-38]
    for values, strict_dw in ((tame + [3e38, -3e38], False), (tame, True)):
        W, V, R, dZ = (_special((B, N, L), 1, values), _special((B, N, C), 2, values), _special((B, N, C), 3, values),
                       _special((B, N, C), 4, values))
        with np.errstate(all="ignore"):
            want = _oracle_fwd(W, V, R)
            dF, dV = oc.spmul_bwd(dZ, W, V)
        got = sfa.chord_spmm(_t(W, gpu), _t(V, gpu), _t(R, gpu)).cpu().numpy()
        assert _same_bits_or_both_nan(got, want)
        gW = torch.full((B, N, L), 7.0, device=gpu)
        gV = torch.full((B, N, C), 7.0, device=gpu)
        _launch_bwd(_t(dZ, gpu), _t(W, gpu), _t(V, gpu), gW, gV, B, N, L, C, N * C, None)
        assert _same_bits_or_both_nan(gV.cpu().numpy(), dV)
        g = gW.cpu().numpy()
        if strict_dw:
            assert np.array_equal(np.isnan(g), np.isnan(dF)) and np.array_equal(np.isposinf(g), np.isposinf(dF))
            assert np.array_equal(np.isneginf(g), np.isneginf(dF))
            fin = np.isfinite(dF)
            assert np.max(np.abs(g[fin] - dF[fin])) <= TOL * np.max(np.abs(dF[fin]))
        else:  # a partial sum that overflows in one order and not in the other: only rows without such values are comparable
            clean = (np.abs(np.nan_to_num(dZ, nan=0.0, posinf=0.0, neginf=0.0)).max(-1) < 1e30)[..., None] & np.isfinite(dF) & np.isfinite(g)
            assert np.max(np.abs(g[clean] - dF[clean])) <= TOL * max(1.0, float(np.max(np.abs(dF[clean]))))


def _random_shapes(n, seed):
    rng = np.random.default_rng(seed)
    shapes = []
    for _ in range(n):
        N = int(rng.choice([rng.integers(1, 64), rng.integers(64, 700), rng.integers(700, 6000)]))
        L = int(rng.integers(1, 25))
        C = int(rng.choice([rng.integers(1, 13), 4 * rng.integers(1, 12), 32 * rng.integers(1, 9), 4 * rng.integers(12, 70)]))
        B = int(rng.integers(1, 4))
        shapes.append((B, N, L, C))
    return shapes


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_random_shapes_forward_and_backward(gpu, seed):
    """Dispatcher sweep: 30 random (B, N, L, C) per seed — tiny and non-power-of-two N, L from 1 to 24 (inside and outside
    the compiled window kernels), C not a multiple of 4, multiples of 32 (chunk-looping dW), ragged tilings — forward and
    dV bit-exact against the oracle, dW <= 1e-5, with and without the residual."""
    import sparsefactorization_amd as sfa
    from sparsefactorization_amd.chord import _launch_bwd
    for (B, N, L, C) in _random_shapes(30, seed):
        W, V, R, dZ = _mk((B, N, L), 7 * seed + 1, 0.5), _mk((B, N, C), 7 * seed + 2), _mk((B, N, C), 7 * seed + 3), _mk((B, N, C), 7 * seed + 4)
        tag = f"B={B} N={N} L={L} C={C}"
        got = sfa.chord_spmm(_t(W, gpu), _t(V, gpu), _t(R, gpu)).cpu().numpy()
        assert np.array_equal(got, _oracle_fwd(W, V, R)), tag
        got = sfa.chord_spmm(_t(W, gpu), _t(V, gpu)).cpu().numpy()
        assert np.array_equal(got, _oracle_fwd(W, V)), tag
        gW = torch.full((B, N, L), float("nan"), device=gpu)
        gV = torch.full((B, N, C), float("nan"), device=gpu)
        _launch_bwd(_t(dZ, gpu), _t(W, gpu), _t(V, gpu), gW, gV, B, N, L, C, N * C, None)
        dF, dV = oc.spmul_bwd(dZ, W, V)
        assert np.array_equal(gV.cpu().numpy(), dV), tag
        assert rel_inf(gW.cpu().numpy(), dF) <= TOL, tag


FUSED_SHAPES = [(3, 16384, 15, 8), (2, 1024, 12, 32), (2, 2048, 12, 16), (5, 512, 9, 4), (2, 256, 8, 8), (1, 1536, 20, 16),
                (2, 128, 7, 32),
                # round 6: rows of 64 channels at every length and of 128 up to N = 4096 (LRA/psf_training_config.py:2-30: ListOps
                # N = 2000 x 128; BASELINE configs[2]: N = 2048 x 64) — aligned instance, and 128 channels beyond 4096 (two kernels)
                (2, 2048, 12, 64), (2, 2000, 12, 128), (1, 16384, 15, 64), (2, 4096, 13, 128), (1, 8192, 14, 128), (3, 64, 6, 64)]


@pytest.mark.parametrize("B,N,L,C", FUSED_SHAPES)
def test_fused_backward_step_kernel(gpu, B, N, L, C):
    """chord_bwd_fused_k (csrc/bwd_fused.h): dV and dW of a step from ONE kernel — what backward_host computes with two
    (spmul/spmul_cuda.cu:114-159) — vs the oracle: dV bit-exact, dW <= 1e-5; a broadcast V (v_batch_stride = 0), and the same
    numbers as the two-kernel path (bwd_fused = 0). Shapes the fused kernel does not take fall back silently (smallest N here)."""
    import sparsefactorization_amd as sfa
    from sparsefactorization_amd.chord import _launch_bwd
    W, V, dZ = _mk((B, N, L), 81), _mk((B, N, C), 82), _mk((B, N, C), 83)
    Wt, Vt, dZt = _t(W, gpu), _t(V, gpu), _t(dZ, gpu)
    dF, dV = oc.spmul_bwd(dZ, W, V)
    try:
        results = []
        for fused in (1, 0):
            sfa.set_tuning("bwd_fused", fused)
            gW = torch.full_like(Wt, float("nan"))
            gV = torch.full_like(Vt, float("nan"))
            _launch_bwd(dZt, Wt, Vt, gW, gV, B, N, L, C, N * C, None)
            assert np.array_equal(gV.cpu().numpy(), dV), fused
            assert rel_inf(gW.cpu().numpy(), dF) <= TOL, fused
            results.append(gW)
        # broadcast first operand (the attention-map chain's unbatched start, pathfinder_inference.py:57,75-81)
        sfa.set_tuning("bwd_fused", 1)
        dFb, dVb = oc.spmul_bwd(dZ, W, np.broadcast_to(V[:1], V.shape).copy())
        gW = torch.full_like(Wt, float("nan"))
        gV = torch.full_like(Vt, float("nan"))
        _launch_bwd(dZt, Wt, Vt[0].contiguous(), gW, gV, B, N, L, C, 0, None)
        assert np.array_equal(gV.cpu().numpy(), dVb) and rel_inf(gW.cpu().numpy(), dFb) <= TOL
    finally:
        sfa.set_tuning("bwd_fused", 1)


FUSED_EDGE_SHAPES = [(2, 4097, 13, 32), (2, 1025, 11, 32), (1, 16385, 15, 8), (3, 777, 9, 16), (2, 643, 10, 4), (2, 1024, 12, 32),
                     (2, 2049, 13, 64), (2, 2001, 12, 128), (2, 4097, 13, 64), (2, 1024, 11, 64), (3, 515, 9, 128)]  # round 6: wide rows


@pytest.mark.parametrize("B,N,L,C", FUSED_EDGE_SHAPES)
def test_fused_backward_step_general_instance(gpu, B, N, L, C):
    """chord_bwd_fused_edge_k (csrc/bwd_fused.h): the fused step for any sequence length (LRA's CLS-token column makes
    N = 2^k + 1, LRA/listops_training.py:65-72), for W / dW buffers that start anywhere (views into larger buffers, one
    float off a 16-byte boundary) and for far offsets that are no multiples of the tile (last case: aligned N, odd far
    offsets): dV bit for bit against the oracle, dW to the parity bar, and both equal to the two-kernel path's results."""
    import sparsefactorization_amd as sfa
    from sparsefactorization_amd.chord import _launch_bwd
    W, V, dZ = _mk((B, N, L), 95), _mk((B, N, C), 96), _mk((B, N, C), 97)
    off = None
    if N % 64 == 0:  # aligned length: make the far offsets odd so that only the general instance applies
        KN = 2
        t = 256 >> {4: 0, 8: 1, 16: 2, 32: 3, 64: 4, 128: 5}[C]
        while t > 1:
            KN, t = KN + 1, t >> 1
        KN = min(KN, L)
        off = [0] + [1 << k for k in range(KN - 1)] + [(1 << k) + 37 for k in range(KN - 1, L - 1)]
    dF, dV = oc.spmul_bwd(dZ, W, V, off)
    Wbig = torch.zeros(B * N * L + 1, device=gpu)
    Wbig[1:] = _t(W, gpu).reshape(-1)
    Wt = Wbig[1:].view(B, N, L)  # 4 bytes off a 16-byte boundary
    Vt, dZt = _t(V, gpu), _t(dZ, gpu)
    try:
        got = {}
        for fused in (1, 0):
            sfa.set_tuning("bwd_fused", fused)
            gWbig = torch.full((B * N * L + 3,), float("nan"), device=gpu)
            gW = gWbig[3:].view(B, N, L)  # 12 bytes off
            gV = torch.full_like(Vt, float("nan"))
            _launch_bwd(dZt, Wt, Vt, gW, gV, B, N, L, C, N * C, off)
            assert torch.isnan(gWbig[:3]).all(), "wrote in front of the dW buffer"
            assert np.array_equal(gV.cpu().numpy(), dV), fused
            assert rel_inf(gW.cpu().numpy(), dF) <= TOL, fused
            got[fused] = (gW.clone(), gV.clone())
        assert torch.equal(got[1][1], got[0][1])
    finally:
        sfa.set_tuning("bwd_fused", 1)



@pytest.mark.parametrize("B,N,L,C", [(3, 16384, 15, 8), (2, 8192, 14, 32), (2, 8192 + 256, 14, 16)])
def test_fused_backward_fronts_only_reorder_workgroups(gpu, B, N, L, C):
    """Knob "bwd_fronts" (round 6): an XCD walks the tiles of a batch element as 1, 2, 4 or 8 interleaved fronts. Whatever
    the walk, both gradients are the same bits, dV the oracle's (spmul_cuda.cu:75-84) and dW within 1e-5 (:102-111)."""
    import sparsefactorization_amd as sfa
    g = torch.Generator(device=gpu).manual_seed(N + C)
    W = 0.3 * torch.randn(B, N, L, device=gpu, generator=g)
    V = torch.randn(B, N, C, device=gpu, generator=g)
    dZ = torch.randn(B, N, C, device=gpu, generator=g)
    got = {}
    try:
        for fronts in (0, 1, 2, 4, 8):
            sfa.set_tuning("bwd_fronts", fronts)
            Wr, Vr = W.clone().requires_grad_(True), V.clone().requires_grad_(True)
            sfa.chord_spmm(Wr, Vr).backward(dZ)
            got[fronts] = (Wr.grad, Vr.grad)
    finally:
        sfa.set_tuning("bwd_fronts", 0)
    for fronts in (0, 2, 4, 8):
        assert torch.equal(got[fronts][0], got[1][0]) and torch.equal(got[fronts][1], got[1][1]), fronts
    want_dW, want_dV = oc.spmul_bwd(dZ.cpu().numpy(), W.cpu().numpy(), V.cpu().numpy())
    assert np.array_equal(got[0][1].cpu().numpy(), want_dV)
    assert rel_inf(got[0][0].cpu().numpy(), want_dW) <= TOL


@pytest.mark.parametrize("B,N,L,C,res", [(2, 16384, 15, 32, False), (3, 4096, 13, 16, True), (2, 2048, 12, 64, False),
                                         (2, 4097, 13, 32, True), (3, 1024, 11, 16, False)])
def test_forward_rows_per_thread_is_a_launch_choice_only(gpu, B, N, L, C, res):
    """Knob "fwd_rows" (round 6): rows of 16..64 channels run four rows per thread where the four-row tile divides N. Two or
    four rows per thread, forced or automatic, full tiles or the ragged 2^k + 1 length: the same bits, the oracle's
    (forward_kernel, spmul_cuda.cu:20-27; residual SyntheticExperiments/psf.py:187-188)."""
    import sparsefactorization_amd as sfa
    g = torch.Generator(device=gpu).manual_seed(N + C)
    W = 0.3 * torch.randn(B, N, L, device=gpu, generator=g)
    V = torch.randn(B, N, C, device=gpu, generator=g)
    R = torch.randn(B, N, C, device=gpu, generator=g) if res else None
    got = {}
    try:
        for rows in (0, 2, 4):
            sfa.set_tuning("fwd_rows", rows)
            got[rows] = sfa.chord_spmm(W, V, R)
    finally:
        sfa.set_tuning("fwd_rows", 0)
    assert torch.equal(got[2], got[0]) and torch.equal(got[4], got[0])
    want = oc.spmul_fwd(W.cpu().numpy(), V.cpu().numpy())
    if res:
        want = want + R.cpu().numpy()
    assert np.array_equal(got[0].cpu().numpy(), want)
