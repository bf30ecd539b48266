"""isa_lint.py — the build-time checks on the emitted gfx950 assembly (sparsefactorization_amd/isa_lint.py), on small
hand-written streams: each rule fires on the pattern it guards and stays silent on the form the kernels ship."""
import os
import textwrap


from sparsefactorization_amd import build, isa_lint

HEAD = "\t.type\tk,@function\nk:\n"
TAIL = "\ts_endpgm\n.Lfunc_end0:\n"


def _lint(tmp_path, body):
    p = tmp_path / "k.s"
    p.write_text(HEAD + textwrap.dedent(body) + TAIL)
    errs, notes, n = isa_lint.lint_file(str(p))
    assert n == 1
    return errs, notes


def test_r1_counted_vmcnt_matches_the_loads_behind_the_last_dma(tmp_path):
    ok = """
        global_load_lds_dwordx4 v[2:3], off
        global_load_dwordx4 v[4:7], v1, s[2:3]
        global_load_dwordx4 v[8:11], v1, s[2:3] offset:64
        ;;#ASMSTART
        s_waitcnt vmcnt(2)
        ;;#ASMEND
        s_barrier
    """
    assert _lint(tmp_path, ok) == ([], [])
    merged = ok.replace("        global_load_dwordx4 v[8:11], v1, s[2:3] offset:64\n", "")
    errs, _ = _lint(tmp_path, merged)  # hipcc dropped / merged one of the counted loads: a DMA may still be in flight
    assert len(errs) == 1 and "only 1 counted" in errs[0]
    extra = ok.replace("        ;;#ASMSTART", "        global_store_dwordx4 v1, v[4:7], s[2:3]\n        ;;#ASMSTART")
    errs, notes = _lint(tmp_path, extra)  # one more counted operation: safe, waits longer than intended
    assert errs == [] and len(notes) == 1
    branch = ok.replace("        global_load_dwordx4 v[8:11]", "        s_cbranch_scc1 .LBB0_9\n        global_load_dwordx4 v[8:11]")
    errs, _ = _lint(tmp_path, branch)
    assert len(errs) == 1 and "straight-line" in errs[0]


def test_r1_leaves_dmas_written_as_inline_assembly_alone(tmp_path):
    body = """
        ;;#ASMSTART
        global_load_lds_dwordx4 v0, s[2:3]
        ;;#ASMEND
    .LBB0_1:
        ;;#ASMSTART
        s_waitcnt vmcnt(6)
        ;;#ASMEND
        s_barrier
    """
    assert _lint(tmp_path, body) == ([], [])  # x3_gemm.h: the source counts its own inline-assembly DMAs per iteration


def test_r0_and_r2(tmp_path):
    errs, _ = _lint(tmp_path, """
        ds_read_b128 v[2:5], v10
        ds_read_b128 v[6:9], v10 offset:16
        s_waitcnt lgkmcnt(1)
        v_add_f32_e32 v20, v6, v2
    """)
    assert len(errs) == 1 and "(R0)" in errs[0]  # v6 is still owned by the second read
    errs, _ = _lint(tmp_path, """
        s_load_dwordx2 s[4:5], s[0:1], 0x10
        ds_read_b128 v[2:5], v10
        ds_read_b128 v[6:9], v10 offset:16
        s_waitcnt lgkmcnt(1)
        v_add_f32_e32 v20, v2, v3
    """)
    assert any("scalar-memory" in e for e in errs)  # SMEM returns out of order: the count proves nothing


def test_r3_is_graded(tmp_path, monkeypatch):
    """R3 (isa_lint.py's docstring): counted everywhere; an error only in a kernel that issues MFMAs, behind a counted wait
    with other reads still in flight; a note-free count behind a full wait; PSF_ISA_LINT_R3=warn downgrades the error."""
    monkeypatch.delenv("PSF_ISA_LINT_R3", raising=False)
    pattern = """
        ds_read_b128 v[14:17], v113
        ds_read_b128 v[18:21], v113 offset:128
        s_waitcnt lgkmcnt(1)
        v_pk_mul_f32 v[14:15], v[2:3], v[14:15] op_sel_hi:[0,1]
    """
    mfma = "        v_mfma_f32_32x32x16_bf16 v[30:45], v[22:25], v[26:29], v[30:45]\n"

    def sites(body):
        p = tmp_path / "k.s"
        p.write_text(HEAD + textwrap.dedent(body) + TAIL)
        found = {}
        errs, notes, _ = isa_lint.lint_file(str(p), found, strict)
        return errs, notes, sum(found.values())

    strict = True  # (the units of the kernels the failure was seen in: fwd_mlp_step_*, mixer_lds)
    errs, _, n = sites(mfma + pattern)
    assert len(errs) == 1 and "(R3)" in errs[0] and n == 1
    strict = False  # every other unit: the same site is a loud note (589 of them in linear_wgrad.hip, results pinned since round 2)
    errs, notes, n = sites(mfma + pattern)
    assert errs == [] and len(notes) == 1 and notes[0].startswith("WARNING") and n == 1
    monkeypatch.setenv("PSF_ISA_LINT_R3", "error")
    assert len(sites(mfma + pattern)[0]) == 1
    monkeypatch.delenv("PSF_ISA_LINT_R3")
    strict = True
    assert sites(pattern) == ([], [], 1)  # no MFMA in the kernel: the forward window kernels' normal form — counted, not an error
    monkeypatch.setenv("PSF_ISA_LINT_R3", "warn")
    errs, notes, n = sites(mfma + pattern)
    assert errs == [] and len(notes) == 1 and notes[0].startswith("WARNING") and n == 1
    monkeypatch.delenv("PSF_ISA_LINT_R3")
    # the shipped form: a full wait, the empty behind_wait statements, hipcc's own (now redundant) counted waits in between.
    # The packed multiply is still the first vector instruction behind the wait that released its operand: the site is
    # COUNTED (the round-5 rule let the redundant wait hide it) but nothing was in flight, so it is not an error
    shipped = """
        ds_read_b128 v[14:17], v113
        ds_read_b128 v[18:21], v113 offset:128
        ;;#ASMSTART
        s_waitcnt lgkmcnt(0)
        ;;#ASMEND
        s_waitcnt lgkmcnt(1)
        ;;#ASMSTART
        ;;#ASMEND
        s_nop 0
        v_pk_mul_f32 v[14:15], v[2:3], v[14:15] op_sel_hi:[0,1]
    """
    assert sites(mfma + shipped) == ([], [], 1)
    # two vector instructions between the wait and the packed consumer: not a site at all
    distant = shipped.replace("        s_nop 0\n", "        v_mov_b32_e32 v40, v1\n        v_mov_b32_e32 v41, v1\n")
    assert sites(mfma + distant) == ([], [], 0)
    # a counted wait that releases one read, then a later wait that releases nothing: the watch survives it
    leaky = """
        ds_read_b128 v[14:17], v113
        ds_read_b128 v[18:21], v113 offset:128
        s_waitcnt lgkmcnt(1)
        s_waitcnt lgkmcnt(1)
        v_pk_mul_f32 v[14:15], v[2:3], v[14:15] op_sel_hi:[0,1]
    """
    errs, _, n = sites(mfma + leaky)
    assert len(errs) == 1 and n == 1


def test_r3_summary_line():
    assert "0 sites" in isa_lint.r3_summary("u", {})
    line = isa_lint.r3_summary("fwd_window_tgs1", {"a": 21, "b": 3})
    assert "24 sites in 2 kernels" in line and "21 in a" in line


def test_csrc_hash_covers_the_effective_flags(monkeypatch):
    """A diagnostic build (PSF_HIPCC_EXTRA, per-unit flags) must not carry the product's hash: _lib.load's staleness check and
    bench.py's counter matching both rely on it."""
    base = build.csrc_hash()
    monkeypatch.setattr(build, "HIPCC_FLAGS", [*build.HIPCC_FLAGS, "-DPSF_BWD_ABLATE_LAB"])
    assert build.csrc_hash() != base


def test_every_unit_with_a_hand_placed_wait_is_linted():
    """A source that contains a hand-written s_waitcnt must belong to a unit that build.py lints."""
    csrc = build.CSRC
    hand = set()
    import re
    for f in os.listdir(csrc):
        with open(os.path.join(csrc, f)) as fh:
            if re.search(r'asm volatile\("s_waitcnt|__builtin_amdgcn_s_waitcnt\(', fh.read()):  # (comments mention it too)
                hand.add(f)
    assert {"fwd_window.h", "psf_common.h", "x3_gemm.h", "mlp_bwd.hip"} <= hand
    assert all(lint for *_, lint in build._unit_table())  # round 6: every unit is linted (R3 sites counted everywhere)
    linted_sources = {os.path.basename(src) for _, src, _, lint in build._unit_table() if lint}
    includes = {"fwd_window.h": "fwd_window_inst.hip", "fwd_mlp_step.h": "fwd_mlp_step_inst.hip", "mixer_lds.h": "mixer_lds_inst.hip",
                "x3_gemm.h": "mlp_wide.hip", "psf_common.h": "fwd_mlp_step_inst.hip"}
    for f in hand:
        assert includes.get(f, f) in linted_sources, f"{f} carries a hand-placed wait but no linted unit compiles it"
