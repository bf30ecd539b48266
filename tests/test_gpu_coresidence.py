"""GPU: the chord kernels beside an MFMA kernel on a second stream of the same device.

Why this test exists (profiles/r04b_mixer_lds_wait.md, isa_lint.py R3): the one sporadically wrong result this library ever
produced (round 4, chord_fwd_mlp_k) was attributed — by inference, never by observation — to a packed-f32 instruction
consuming a just-released ds_read result "under a co-resident wave's MFMA phase". The chord window kernels carry that ISA
pattern thousands of times (21 sites in the headline instance chord_fwd_win_k<float,15,1,2,256,true,true,2>) and issue no
MFMA themselves, so whether they are safe hinged on an untested premise: that they never share a CU with an MFMA kernel.
Nothing in the boundary says so, and a PSFNet forward on two streams (producers on one, chain on the other) does exactly that.

So: the chain of SURVEY.md §8(a2) at BASELINE configs[1]'s full size (SyntheticExperiments/psf.py:172-188 semantics) runs on
stream A while the producer MLP kernel (x3_fwd_k: split-bf16 MFMA phases, psf_mlp_fwd_f32) runs on stream B, a fixed small
number of rounds, and every result must be bit-equal to the solo run — and the chain bit-equal to the CPU oracle on batch
elements 0 / 37 / 63, as tests/test_gpu_parity.py::test_full_size_properties checks the solo chain. The backward step
(chord_bwd_fused_k, 36 sites) gets the same treatment beside the MLP backward (mlp_bwd_x3p_k).

This is new coverage of a configuration, not a re-run of a known failure: a fixed launch count, one pass.
"""
import numpy as np
import pytest
import torch

from oracle import chord_oracle as oc

pytestmark = pytest.mark.gpu

ROUNDS = 6


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


def _producer(gpu, T, E=32, M=14, C=8):
    """The 15 MLPs of the synthetic PSFNet at cfg2 (SyntheticExperiments/psf.py:35-60,165,175): parameters and input rows."""
    g = torch.Generator(device=gpu).manual_seed(99)
    x2 = torch.randn(T, E, device=gpu, generator=g)
    params = []
    for out in [C] + [M + 1] * M:
        params += [torch.randn(32, E, device=gpu, generator=g) / E ** 0.5, 0.1 * torch.randn(32, device=gpu, generator=g),
                   torch.randn(out, 32, device=gpu, generator=g) / 32 ** 0.5, 0.1 * torch.randn(out, device=gpu, generator=g)]
    return x2, params


def _overlap_ms(e0, a0, a1, b0, b1):
    """Length of the intersection of the two streams' busy intervals, from events timed against a common origin."""
    sa, ea, sb, eb = e0.elapsed_time(a0), e0.elapsed_time(a1), e0.elapsed_time(b0), e0.elapsed_time(b1)
    return min(ea, eb) - max(sa, sb), (sa, ea, sb, eb)


def test_chain_beside_the_mfma_producer_on_a_second_stream(gpu):
    import sparsefactorization_amd as sfa
    from sparsefactorization_amd import fused_mlp
    Ws, V0 = _cfg2(gpu)
    x2, params = _producer(gpu, T=64 * 16384)
    solo = sfa.chord_chain(Ws, V0, True)
    ys_solo = fused_mlp._forward_raw(x2, params)
    torch.cuda.synchronize()

    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
    ev = lambda: torch.cuda.Event(enable_timing=True)
    bad = {sA: torch.zeros((), dtype=torch.int64, device=gpu), sB: torch.zeros((), dtype=torch.int64, device=gpu)}
    spans, last = [], None
    # Results are compared with the solo run ON the stream that made them and dropped at once (a count of differing elements
    # per stream, read after the last round): nothing synchronises between rounds, and from the second round on each stream's
    # allocator pool already holds the blocks — round 0 is the warm-up that fills the pools (hipMalloc stalls the host; its
    # overlap is not counted).
    for r in range(ROUNDS + 1):
        e0, a0, a1, b0, b1 = ev(), ev(), ev(), ev(), ev()
        e0.record()
        sA.wait_event(e0)
        sB.wait_event(e0)
        # alternate who goes first: the chain's workgroups move into CUs the producer is draining from, and the other way round
        for s in ((sB, sA) if r % 2 == 0 else (sA, sB)):
            with torch.cuda.stream(s):
                if s is sB:
                    b0.record()
                    for _ in range(2):
                        ys = fused_mlp._forward_raw(x2, params)
                        b1.record()
                        for y, y0 in zip(ys, ys_solo):
                            bad[sB] += (y != y0).sum()
                        del ys, y
                else:
                    a0.record()
                    for _ in range(3):  # three chains ~ two producer launches in time
                        last = sfa.chord_chain(Ws, V0, True)
                        a1.record()
                        bad[sA] += (last != solo).sum()
        if r:
            spans.append((e0, a0, a1, b0, b1))
        # both streams start the next round together (on the device: the host does not wait)
        da, db = torch.cuda.Event(), torch.cuda.Event()
        da.record(sA)
        db.record(sB)
        sA.wait_event(db)
        sB.wait_event(da)
    torch.cuda.synchronize()

    overlaps = [_overlap_ms(*s)[0] for s in spans]
    print("coresidence: overlap of the two streams' kernel intervals per round, ms:", [round(o, 3) for o in overlaps])
    assert sum(o > 0.1 for o in overlaps) >= ROUNDS // 2, \
        f"the two streams ran at the same time in too few rounds ({overlaps}): the test did not test co-residence"
    assert int(bad[sA]) == 0, f"{int(bad[sA])} chain elements differ from the solo run beside the MFMA producer"
    assert int(bad[sB]) == 0, f"{int(bad[sB])} producer outputs differ from the solo run beside the chord chain"
    sel = [0, 37, 63]
    W_np = np.stack([w[sel].cpu().numpy() for w in Ws])
    rows, cols = oc.chord_indices(16384, 15)
    want = oc.chain(np.stack([rows, cols]), W_np, V0[sel].cpu().numpy(), True)[-1]
    assert np.array_equal(solo[sel].cpu().numpy(), want)
    assert np.array_equal(last[sel].cpu().numpy(), want)


def test_backward_step_beside_the_mfma_mlp_backward_on_a_second_stream(gpu):
    """chord_bwd_fused_k (dV bit-exact vs the oracle, dW <= 1e-5: spmul_cuda.cu:75-84,102-111) while mlp_bwd_x3p_k runs."""
    import sparsefactorization_amd as sfa
    from sparsefactorization_amd import fused_mlp
    B, N, L, C = 40, 16384, 15, 8  # the Temporal-Order training shape (BASELINE configs[4])
    g = torch.Generator(device=gpu).manual_seed(7)
    W = 0.1 * torch.randn(B, N, L, device=gpu, generator=g)
    V = torch.randn(B, N, C, device=gpu, generator=g)
    dZ = torch.randn(B, N, C, device=gpu, generator=g)
    x2, params = _producer(gpu, T=B * N)
    gys = [torch.randn(x2.shape[0], p.shape[0], device=gpu, generator=g) for p in params[2::4]]

    def step():
        Wr, Vr = W.clone().requires_grad_(True), V.clone().requires_grad_(True)
        sfa.chord_spmm(Wr, Vr).backward(dZ)
        return Wr.grad, Vr.grad

    dW0, dV0 = step()
    dX0, grads0 = fused_mlp._backward_raw(x2, params, gys, True)
    torch.cuda.synchronize()
    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
    sA.wait_stream(torch.cuda.current_stream())
    sB.wait_stream(torch.cuda.current_stream())
    bad = {sA: torch.zeros((), dtype=torch.int64, device=gpu), sB: torch.zeros((), dtype=torch.int64, device=gpu)}
    for r in range(ROUNDS + 1):  # (round 0 fills the streams' allocator pools)
        for s in ((sB, sA) if r % 2 == 0 else (sA, sB)):
            with torch.cuda.stream(s):
                if s is sB:
                    dX, grads = fused_mlp._backward_raw(x2, params, gys, True)
                    bad[sB] += (dX != dX0).sum()
                    for a, b in zip(grads, grads0):
                        bad[sB] += (a != b).sum()
                    del dX, grads
                else:
                    for _ in range(8):  # ~ 8 x 45 us beside a ~ 0.75 ms MLP backward
                        dW, dV = step()
                        bad[sA] += (dV != dV0).sum() + (dW != dW0).sum()
                        del dW, dV
    torch.cuda.synchronize()
    assert int(bad[sA]) == 0, f"{int(bad[sA])} gradient elements of the chord step differ from the solo run"
    assert int(bad[sB]) == 0, f"{int(bad[sB])} gradient elements of the MLP backward differ from the solo run"
    sel = [0, 21, 39]
    want_dW, want_dV = oc.spmul_bwd(dZ[sel].cpu().numpy(), W[sel].cpu().numpy(), V[sel].cpu().numpy())
    assert np.array_equal(dV0[sel].cpu().numpy(), want_dV)
    assert float(np.abs(dW0[sel].cpu().numpy() - want_dW).max() / np.abs(want_dW).max()) <= 1e-5


def test_wide_row_kernels_beside_the_wide_mfma_producer(gpu):
    """The same for the LRA widths (reference ListOps: N = 2000, 128 channels, E = 512, hidden 128 —
    LRA/psf_training_config.py:2-30): the per-step forward kernel, the one-launch chain (chord_chain_lds_k) and the two-kernel
    backward step (chord_dv_win_k + chord_dw_chunk_k) on stream A while the wide producer GEMMs (x3_gemm_k, mlp_wide.hip:
    split-bf16 MFMA, LDS-DMA ring) run on stream B. Every result bit-equal to its solo run; the chain bit-equal to the oracle."""
    import sparsefactorization_amd as sfa
    from sparsefactorization_amd import fused_mlp
    from sparsefactorization_amd.psfnet import MLPBlock
    B, N, L, C, E, M = 8, 2000, 12, 128, 512, 11
    g = torch.Generator(device=gpu).manual_seed(21)
    Ws = [0.2 * torch.randn(B, N, L, device=gpu, generator=g) for _ in range(M)]
    V0 = torch.randn(B, N, C, device=gpu, generator=g)
    dZ = torch.randn(B, N, C, device=gpu, generator=g)
    torch.manual_seed(5)
    blocks = [MLPBlock([128, 'GELU'], E, C).to(gpu)] + [MLPBlock([128, 'GELU'], E, L).to(gpu) for _ in range(M)]
    x = torch.randn(16, N, E, device=gpu, generator=g)
    assert fused_mlp.wide_ok(x, blocks)

    def chord_work():
        with torch.no_grad():
            one_launch = sfa.chord_chain(Ws, V0, False)            # inference: chord_chain_lds_k
            sfa.set_tuning("chain_fused", 0)
            try:
                per_step = sfa.chord_chain(Ws, V0, False)          # what training runs: M per-step kernels
            finally:
                sfa.set_tuning("chain_fused", 1)
        Wr, Vr = Ws[0].clone().requires_grad_(True), V0.clone().requires_grad_(True)
        sfa.chord_spmm(Wr, Vr).backward(dZ)                        # wide rows: dV and dW as two kernels
        return one_launch, per_step, Wr.grad, Vr.grad

    def producer_work():
        with torch.no_grad():
            return fused_mlp.wide_apply(x, blocks)

    solo = chord_work()
    solo_y = producer_work()
    torch.cuda.synchronize()
    assert torch.equal(solo[0], solo[1])  # the one-launch chain and the per-step kernels: the same bits
    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
    sA.wait_stream(torch.cuda.current_stream())
    sB.wait_stream(torch.cuda.current_stream())
    bad = {sA: torch.zeros((), dtype=torch.int64, device=gpu), sB: torch.zeros((), dtype=torch.int64, device=gpu)}
    for r in range(ROUNDS + 1):
        for s in ((sB, sA) if r % 2 == 0 else (sA, sB)):
            with torch.cuda.stream(s):
                if s is sB:
                    for _ in range(2):
                        ys = producer_work()
                        for y, y0 in zip(ys, solo_y):
                            bad[sB] += (y != y0).sum()
                        del ys
                else:
                    for _ in range(4):
                        got = chord_work()
                        for a, b in zip(got, solo):
                            bad[sA] += (a != b).sum()
                        del got
        da, db = torch.cuda.Event(), torch.cuda.Event()
        da.record(sA)
        db.record(sB)
        sA.wait_event(db)
        sB.wait_event(da)
    torch.cuda.synchronize()
    assert int(bad[sA]) == 0, f"{int(bad[sA])} elements of the wide-row chord kernels differ from the solo run"
    assert int(bad[sB]) == 0, f"{int(bad[sB])} producer outputs differ from the solo run"
    rows, cols = oc.chord_indices(N, L)
    want = oc.chain(np.stack([rows, cols]), np.stack([w[:2].cpu().numpy() for w in Ws]), V0[:2].cpu().numpy(), False)[-1]
    assert np.array_equal(solo[0][:2].cpu().numpy(), want)
