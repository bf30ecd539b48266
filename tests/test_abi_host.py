"""CPU: the C-ABI library loads and exports every symbol of include/psf_chord.h; host-only entry points and
argument validation (everything that returns before the first HIP call); Python host logic and module surface.
No compute is launched here.
"""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden


@pytest.fixture(scope="module")
def lib():
    from sparsefactorization_amd import _lib, build
    build.build()  # hipcc cross-compiles gfx950 without a GPU
    return _lib.load()


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "psf_chord.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(psf_[a-z0-9_]+)\s*\(", text)))


def test_exports_every_declared_symbol(lib):
    from sparsefactorization_amd import _lib
    declared = _declared_functions()
    assert len(declared) >= 14
    for name in declared:
        assert hasattr(lib, name), f"libpsf_chord.so does not export {name}"
    assert sorted(_lib.SIGNATURES) == declared  # the ctypes table binds exactly the header
    assert lib.psf_version() == 2
    assert b"gfx950" in lib.psf_build_info()


def test_chord_indices_host_routine_matches_reference(lib):
    g = load_golden("chord_indices.npz")
    for n, l in g["full_cases"]:
        n, l = int(n), int(l)
        rows = (ctypes.c_int64 * (n * l))()
        cols = (ctypes.c_int64 * (n * l))()
        assert lib.psf_chord_indices(n, l, rows, cols) == 0
        assert np.array_equal(np.frombuffer(rows, dtype=np.int64), g[f"rows_{n}_{l}"])
        assert np.array_equal(np.frombuffer(cols, dtype=np.int64), g[f"cols_{n}_{l}"])


def test_python_helper_returns_reference_types(lib):
    import sparsefactorization_amd as sfa
    g = load_golden("psfnet_adding_n128.npz")
    rows, cols = sfa.get_chord_indices_assym(128, 8)
    assert isinstance(rows, list) and isinstance(cols, list)
    assert np.array_equal(torch.tensor((rows, cols)).numpy(), g["chord_indicies"])


def test_offsets(lib):
    from sparsefactorization_amd import _lib
    assert _lib.chord_offsets(16384, 15) == [0] + [2 ** k for k in range(14)]
    assert _lib.chord_offsets(1024, 12)[-1] == 0          # 2^10 mod 1024: duplicate self link
    assert _lib.chord_offsets(100, 9) == [0, 1, 2, 4, 8, 16, 32, 64, 28]
    assert _lib.chord_offsets(1, 4) == [0, 0, 0, 0]
    assert _lib.chord_offsets(3, 64)[63] == pow(2, 62, 3)


def test_argument_validation_returns_codes_not_crashes(lib):
    f = lib.psf_chord_spmm_fwd_f32
    one = ctypes.c_void_p(16)
    two = ctypes.c_void_p(32)
    assert f(None, one, None, two, 1, 8, 4, 4, 32, None, None) == -1          # PSF_E_NULL
    assert f(one, one, None, two, 1, 0, 4, 4, 0, None, None) == -2            # PSF_E_SHAPE (N < 1)
    assert f(one, one, None, two, 1, 8, 65, 4, 32, None, None) == -2          # L > PSF_MAX_LINKS
    assert f(one, one, None, two, 1, 8, 4, 4, 7, None, None) == -2            # bad batch stride
    assert f(one, two, None, two, 1, 8, 4, 4, 32, None, None) == -3           # PSF_E_ALIAS
    assert f(ctypes.c_void_p(18), one, None, two, 1, 8, 4, 4, 32, None, None) == -4  # PSF_E_ALIGN
    assert f(one, one, None, two, 0, 8, 4, 4, 32, None, None) == 0            # empty batch: nothing to do
    assert f(one, two, None, two, 1, 8, 4, 4, 32, None, None) == -3 and b"alias" in lib.psf_last_error()
    # producer-side entry point: validation happens before any HIP call too
    assert lib.psf_linear_wgrad_workspace(1 << 20, 32, 15) > 0
    assert lib.psf_linear_wgrad_workspace(1 << 20, 129, 15) == -1
    assert lib.psf_linear_wgrad_f32(None, one, 64, 8, 8, two, None, two, 1 << 20, None) == -1
    assert b"non-NULL" in lib.psf_last_error()
    assert lib.psf_linear_wgrad_f32(one, one, 64, 8, 300, two, None, two, 1 << 20, None) == -2
    assert lib.psf_linear_wgrad_f32(one, one, 1 << 20, 32, 32, two, None, two, 16, None) == -2
    b = lib.psf_chord_spmm_bwd_f32
    assert b(None, one, one, two, two, 1, 8, 4, 4, 32, None, None) == -1
    assert b(one, None, one, None, two, 1, 8, 4, 4, 32, None, None) == -1     # dV needs W
    assert b(one, one, one, None, one, 1, 8, 4, 4, 32, None, None) == -3      # dV aliases dZ
    c = lib.psf_chord_chain_fwd_f32
    assert c(None, one, None, 2, 0, 1, 8, 4, 4, 32, None, None) == -1
    assert c(None, None, None, 0, 0, 1, 8, 4, 4, 32, None, None) == 0          # M == 0


C99_CONSUMER = r"""
/* A consumer of include/psf_chord.h that is not Python: what a maintainer of the reference would link where
   spmul/spmul_cuda.cu:163-166 binds forward_host / backward_host with pybind11. Host-only calls: no GPU needed. */
#include <stdio.h>
#include <string.h>
#include "psf_chord.h"

int main(void) {
  int64_t off[12];
  float w[4] = {0}, v[4] = {0};
  if (psf_version() != PSF_ABI_VERSION) return 1;
  if (psf_chord_offsets(1024, 12, off) != PSF_OK) return 2;
  if (off[0] != 0 || off[1] != 1 || off[11] != 0) return 3;          /* 2^10 mod 1024: the duplicate self link */
  if (psf_chord_spmm_fwd_f32(NULL, v, NULL, w, 1, 8, 4, 4, 32, NULL, NULL) != PSF_E_NULL) return 4;
  if (strstr(psf_last_error(), "NULL") == NULL) return 5;
  if (psf_chord_spmm_fwd_f32(w, v, NULL, v, 1, 8, 4, 4, 32, NULL, NULL) != PSF_E_ALIAS) return 6;
  if (psf_chord_spmm_fwd_f32(w, v, NULL, w + 1, 1, 0, 4, 4, 0, NULL, NULL) != PSF_E_SHAPE) return 7;
  printf("psf_version=%d offsets[11]=%lld\n", psf_version(), (long long)off[11]);
  return 0;
}
"""


def test_header_is_c99_and_a_plain_c_program_links_the_library(lib, tmp_path):
    """include/psf_chord.h is the boundary, and its consumer need not be Python: the header compiles as strict C99
    (-std=c99 -pedantic -Wall -Werror), a C program links libpsf_chord.so and gets the ABI version, the chord offsets of
    Pathfinder's shape (N = 1024, L = 12: offset 2^10 = 0 mod N, the duplicate self link the reference keeps —
    SyntheticExperiments/psf.py:7-32) and the argument-error codes, all before any HIP call."""
    import shutil
    import subprocess
    from sparsefactorization_amd import build
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    src = tmp_path / "consumer.c"
    src.write_text(C99_CONSUMER)
    exe = tmp_path / "consumer"
    libdir = os.path.dirname(build.LIB_PATH)
    cmd = [gcc, "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), str(src),
           "-L", libdir, "-l:libpsf_chord.so", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)]
    proc = subprocess.run(cmd, capture_output=True, text=True)
    assert proc.returncode == 0, proc.stderr
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, f"consumer exited {run.returncode}: {run.stdout} {run.stderr}"
    assert run.stdout.strip() == "psf_version=2 offsets[11]=0"


def test_entry_points_are_callable_from_two_threads_at_once(lib):
    """include/psf_chord.h, "Threads": host-only entry points and the validation paths of the device entry points hammered
    from two threads while a third flips a knob; every call must return what it returns single-threaded, and each thread
    keeps its own error string."""
    import threading
    one, two = ctypes.c_void_p(16), ctypes.c_void_p(32)
    want = [0] + [2 ** k for k in range(14)]
    errors = []

    def offsets_worker():
        buf = (ctypes.c_int64 * 15)()
        for _ in range(3000):
            if lib.psf_chord_offsets(16384, 15, buf) != 0 or list(buf) != want:
                errors.append("offsets")
            if lib.psf_describe_fwd(64, 16384, 15, 8, 4, ctypes.create_string_buffer(128), 128) != 0:
                errors.append("describe")

    def validation_worker():
        for _ in range(3000):
            if lib.psf_chord_spmm_fwd_f32(one, two, None, two, 1, 8, 4, 4, 32, None, None) != -3:   # PSF_E_ALIAS
                errors.append("alias code")
            if b"alias" not in lib.psf_last_error():
                errors.append("alias message")
            if lib.psf_chord_chain_fwd_f32(None, None, None, 0, 0, 1, 8, 4, 4, 32, None, None) != 0:
                errors.append("empty chain")

    def knob_worker():
        for i in range(3000):
            if lib.psf_set_tuning(b"xcd_remap", i & 1) != 0:
                errors.append("set_tuning")
            if lib.psf_chord_spmm_fwd_f32(None, one, None, two, 1, 8, 4, 4, 32, None, None) != -1 or b"non-NULL" not in lib.psf_last_error():
                errors.append("null message")
        lib.psf_set_tuning(b"xcd_remap", 1)

    threads = [threading.Thread(target=f) for f in (offsets_worker, validation_worker, knob_worker)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, sorted(set(errors))


def test_mixer_entry_point_states_its_limits_without_touching_the_gpu(lib):
    """psf_mixer_fwd_workspace returns -1 outside the fused path's limits (include/psf_chord.h); psf_mixer_fwd_f32 rejects
    NULL tables, short workspaces, misaligned and aliased buffers before any HIP call."""
    i32, vp = ctypes.c_int32, ctypes.c_void_p
    h = (i32 * 3)(32, 32, 32)
    assert lib.psf_mixer_fwd_workspace(16384, 32, 2, h, 8, 15) == 3 * 14080
    assert lib.psf_mixer_fwd_workspace(1024, 32, 2, (i32 * 3)(128, 128, 128), 32, 12) == 12 * 14080
    assert lib.psf_mixer_fwd_workspace(128, 32, 2, h, 8, 8) == 3 * 14080   # BASELINE configs[0]: the single-launch LDS kernel
    assert lib.psf_mixer_fwd_workspace(100, 32, 2, h, 8, 8) == -1      # neither: N < two tiles of 256 rows, not a multiple of 32
    assert lib.psf_mixer_fwd_workspace(256, 32, 2, h, 16, 8) == 3 * 14080  # C = 16: two 128-row tiles, per-step kernels
    assert lib.psf_mixer_fwd_workspace(128, 32, 2, h, 16, 8) == -1     # C = 16 and N < two tiles
    assert lib.psf_mixer_fwd_workspace(512, 32, 2, h, 8, 10) > 0       # exactly two tiles
    assert lib.psf_mixer_fwd_workspace(16384, 48, 2, h, 8, 15) == -1   # E > 32
    assert lib.psf_mixer_fwd_workspace(16384, 30, 2, h, 8, 15) == -1   # E not a multiple of 4
    assert lib.psf_mixer_fwd_workspace(16384, 32, 2, h, 64, 15) == -1  # C > 32
    assert lib.psf_mixer_fwd_workspace(16384, 32, 2, h, 8, 21) == -1   # L > 20
    assert lib.psf_mixer_fwd_workspace(16384, 32, 2, (i32 * 3)(32, 200, 32), 8, 15) == -1  # hidden > 128
    assert lib.psf_mixer_fwd_workspace(16384, 32, 2, None, 8, 15) == -1
    one, two, three = vp(16), vp(32), vp(48)
    tab = (vp * 3)(16, 16, 16)
    outs = (vp * 2)(32, 48)
    f = lib.psf_mixer_fwd_f32
    ws = 3 * 14080
    assert f(None, 1, 16384, 32, 2, tab, tab, tab, tab, h, 8, 15, 1, one, outs, three, ws, None) == -1
    assert f(one, 1, 100, 32, 2, tab, tab, tab, tab, h, 8, 8, 1, two, outs, three, ws, None) == -2 and b"fused path" in lib.psf_last_error()
    assert f(one, 1, 16384, 32, 2, tab, tab, tab, tab, h, 8, 15, 1, two, outs, three, ws - 16, None) == -2  # workspace too small
    assert f(vp(20), 1, 16384, 32, 2, tab, tab, tab, tab, h, 8, 15, 1, two, outs, three, ws, None) == -4    # X misaligned
    assert f(one, 1, 16384, 32, 2, tab, tab, tab, tab, h, 8, 15, 1, two, (vp * 2)(32, 32), three, ws, None) == -3  # out aliases V0
    assert f(one, 1, 16384, 32, 2, tab, tab, tab, tab, h, 8, 15, 1, two, (vp * 2)(64, 64), three, ws, None) == -3  # step in == out
    assert f(one, 0, 16384, 32, 2, tab, tab, tab, tab, h, 8, 15, 1, two, outs, three, ws, None) == 0        # empty batch
    # an input recipe (psf_mixer_input kinds 1, 2) is evaluated by the single-launch kernel only: on the per-step path the entry
    # says so, names the two row-writing entry points, and has launched nothing
    from sparsefactorization_amd import _lib
    rec = _lib.MixerInput(_lib.MIXER_IN_TOKENS, 6, 16, 32, None, None)
    g = lib.psf_mixer_fwd_in_f32
    assert g(ctypes.byref(rec), 1, 16384, 32, 2, tab, tab, tab, tab, h, 8, 15, 1, two, outs, three, ws, None) == -2
    assert b"psf_embed_tokens_f32" in lib.psf_last_error() and b"single-launch" in lib.psf_last_error()
    for gone in ("psf_chord_chain_fwd_far_f32", "psf_chord_spmm_bwd_far_f32", "psf_chord_bwd_far_first_link"):
        assert not hasattr(lib, gone), f"{gone} was removed in ABI version 2"


def test_producer_entry_points_validate_before_touching_the_gpu(lib):
    """psf_mlp_fwd_f32 / psf_mlp_bwd_f32 / psf_linear_wgrad_strided_f32: sizes outside the kernels' limits, NULL
    tables and short workspaces come back as PSF_E_* codes (no HIP call has been made at that point)."""
    i32, vp = ctypes.c_int32, ctypes.c_void_p
    one, two = vp(16), vp(32)

    def arr(vals):
        return (i32 * len(vals))(*vals)

    def ptrs(n, v=16):
        return (vp * n)(*([v] * n))

    h, O = arr([32, 32, 128]), arr([8, 15, 32])
    fwd_ws = lib.psf_mlp_fwd_workspace(32, 3, h, O)
    assert fwd_ws > 0 and fwd_ws % 16 == 0
    assert lib.psf_mlp_fwd_workspace(64, 3, h, O) > 0             # E = 64: f32-MFMA variant only
    assert lib.psf_mlp_fwd_workspace(68, 3, h, O) == -1            # E > 64
    assert lib.psf_mlp_fwd_workspace(30, 3, h, O) == -1            # E not a multiple of 4
    assert lib.psf_mlp_fwd_workspace(32, 33, h, O) == -1           # K > 32
    assert lib.psf_mlp_fwd_workspace(32, 3, arr([32, 129, 32]), O) == -1   # h > 128
    assert lib.psf_mlp_fwd_workspace(32, 3, h, arr([8, 33, 8])) == -1      # out > 32
    f = lib.psf_mlp_fwd_f32
    tabs = [ptrs(3) for _ in range(5)]
    assert f(None, 100, 32, 3, tabs[0], tabs[1], tabs[2], tabs[3], h, O, tabs[4], two, fwd_ws, None) == -1
    assert f(one, 0, 32, 3, tabs[0], tabs[1], tabs[2], tabs[3], h, O, tabs[4], two, fwd_ws, None) == -2       # T < 1
    assert f(vp(20), 100, 32, 3, tabs[0], tabs[1], tabs[2], tabs[3], h, O, tabs[4], two, fwd_ws, None) == -4  # X alignment
    assert f(one, 100, 32, 3, tabs[0], tabs[1], tabs[2], tabs[3], h, O, tabs[4], two, fwd_ws - 16, None) == -2
    assert b"workspace" in lib.psf_last_error()
    assert f(one, 100, 32, 3, (vp * 3)(16, None, 16), tabs[1], tabs[2], tabs[3], h, O, tabs[4], two, fwd_ws, None) == -1

    bwd_ws = lib.psf_mlp_bwd_workspace(100000, 32, 3, h, O)
    assert bwd_ws > 0
    assert lib.psf_mlp_bwd_workspace(100000, 64, 3, h, O) == -1    # the backward covers E <= 32
    assert lib.psf_mlp_bwd_workspace(0, 32, 3, h, O) == -1
    # the partial-sum buffer grows with T: one slot per workgroup of 4 waves x 1 or 2 tiles
    assert lib.psf_mlp_bwd_workspace(10 ** 6, 32, 3, h, O) > bwd_ws
    g = lib.psf_mlp_bwd_f32
    t = [ptrs(3) for _ in range(8)]
    assert g(one, 100000, 32, 3, t[0], t[1], t[2], h, O, t[3], two, t[4], t[5], t[6], t[7], two, bwd_ws - 16, None) == -2
    assert g(one, 100000, 32, 3, t[0], t[1], t[2], h, O, None, two, t[4], t[5], t[6], t[7], two, bwd_ws, None) == -1
    assert g(one, 100000, 32, 3, t[0], t[1], t[2], h, O, (vp * 3)(16, 16, None), two, t[4], t[5], t[6], t[7], two,
             bwd_ws, None) == -1
    assert b"NULL" in lib.psf_last_error()

    assert lib.psf_embed_tokens_bwd_workspace(65536, 225, 32) > 0
    assert lib.psf_embed_tokens_bwd_workspace(65536, 513, 32) == -1     # vocabulary beyond the LDS table
    assert lib.psf_embed_tokens_bwd_f32(one, one, 65536, 225, 32, two, two, 16, None) == -2   # short workspace
    assert lib.psf_embed_tokens_bwd_f32(None, one, 65536, 225, 32, two, two, 1 << 30, None) == -1
    assert lib.psf_embed_tokens_f32(one, one, None, two, 100, 10, 6, 30, None) == -2           # E % 4 != 0
    assert lib.psf_embed_tokens_f32(one, vp(20), None, two, 100, 10, 6, 32, None) == -4        # table alignment

    s = lib.psf_linear_wgrad_strided_f32
    assert s(one, 31, one, 15, 1 << 20, 32, 15, two, None, two, 1 << 30, None) == -2   # ldx < m
    assert s(one, 480, one, 14, 1 << 20, 32, 15, two, None, two, 1 << 30, None) == -2  # ldy < n
    assert b"strides" in lib.psf_last_error()


def test_tuning_knobs(lib):
    import sparsefactorization_amd as sfa
    assert sfa.get_tuning("mlp_variant") == 0
    sfa.set_tuning("mlp_variant", 3)
    assert sfa.get_tuning("mlp_variant") == 3
    sfa.set_tuning("mlp_variant", 0)
    with pytest.raises(sfa.PSFLibraryError):
        sfa.set_tuning("mlp_variant", 4)
    assert sfa.get_tuning("fwd_variant") == 0
    sfa.set_tuning("fwd_variant", 1)
    assert "generic" in sfa.describe_fwd(64, 16384, 15, 8)
    sfa.set_tuning("fwd_variant", 0)
    assert "win" in sfa.describe_fwd(64, 16384, 15, 8)
    with pytest.raises(sfa.PSFLibraryError):
        sfa.set_tuning("no_such_knob", 1)
    with pytest.raises(sfa.PSFLibraryError):
        sfa.set_tuning("fwd_variant", 99)


def test_no_cpu_fallback(lib):
    """CPU tensors must raise: the product has no CPU path (the oracle is test infrastructure only)."""
    import sparsefactorization_amd as sfa
    W, V = torch.zeros(1, 8, 4), torch.zeros(1, 8, 4)
    with pytest.raises(RuntimeError, match="HIP"):
        sfa.chord_spmm(W, V)
    with pytest.raises(RuntimeError, match="HIP"):
        sfa.chord_chain([W], V)
    idx = torch.tensor(sfa.get_chord_indices_assym(8, 4))
    with pytest.raises(RuntimeError, match="HIP"):
        sfa.spmm(idx, W.reshape(1, 32), 8, 8, V)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "sparsefactorization_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+\.*oracle", text, flags=re.M), f
                assert "liboracle" not in text and "chord_oracle" not in text, f


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from sparsefactorization_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libpsf_chord.so"))
    with pytest.raises(_lib.PSFLibraryError, match="not built"):
        _lib.load()


def test_offsets_from_index_validation():
    from sparsefactorization_amd.chord import offsets_from_index
    import sparsefactorization_amd as sfa
    idx = torch.tensor(sfa.get_chord_indices_assym(100, 9))
    assert offsets_from_index(idx, 100) == (0, 1, 2, 4, 8, 16, 32, 64, 28)
    bad = idx.clone()
    bad[0, 3] = 7
    with pytest.raises(ValueError, match="rows"):
        offsets_from_index(bad, 100)
    bad = idx.clone()
    bad[1, 20] += 1
    with pytest.raises(ValueError, match="cols"):
        offsets_from_index(bad, 100)
    with pytest.raises(ValueError):
        offsets_from_index(idx[:, :-1], 100)


# ---------------------------------------------------------------------------------------------------
# module surface: constructor signatures, attribute names, state_dict layout, shipped checkpoints
# ---------------------------------------------------------------------------------------------------
def _layouts():
    out = {}
    for line in load_golden("checkpoint_layouts.npz")["layouts"]:
        ck, key, shape = str(line).split("|")
        out.setdefault(ck, {})[key] = tuple(int(s) for s in shape.split("x")) if shape else ()
    return out


PATHFINDER = dict(vocab_size=225, embedding_size=32, n_vec=1024, n_W=11, Ws=[128, 'GELU'], V=[128, 'GELU'],
                  n_channels_V=32, n_class=2, pooling_type="FLATTEN", head=['linear'], use_cuda=False,
                  use_residuals=False, dropout1_p=0, dropout2_p=0, dropout3_p=0, init_embedding_weights=False,
                  use_pos_embedding=True, problem="pathfinder")
IMDB = dict(vocab_size=97, embedding_size=32, n_vec=4097, n_W=12, Ws=[128, 'GELU'], V=[128, 'GELU'],
            n_channels_V=32, n_class=2, pooling_type="CLS", head=['linear'], use_cuda=False, use_residuals=True,
            dropout1_p=0.4, dropout2_p=0, dropout3_p=0, init_embedding_weights=True, use_pos_embedding=False,
            problem="imdb")


@pytest.mark.parametrize("ckpt,cfg", [("pathfinder_epoch27.pt", PATHFINDER), ("imdb_epoch138.pt", IMDB)])
def test_state_dict_layout_matches_shipped_checkpoints(lib, ckpt, cfg):
    """LRA/psf_training_config.py:60-117 configs; layouts recorded from LRA/attention_maps/*.pt."""
    from sparsefactorization_amd.lra_psf import PSFNet
    net = PSFNet(**cfg)
    mine = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    assert mine == _layouts()[ckpt]
    assert "chord_indicies" not in mine
    assert net.chord_indicies.shape == (2, cfg["n_vec"] * (cfg["n_W"] + 1))
    assert net.n_links == cfg["n_W"] + 1


def test_synthetic_module_surface(lib):
    from sparsefactorization_amd.synthetic_psf import PSFNet, MLPBlock, MakeMLP, get_chord_indices_assym  # noqa: F401
    g = load_golden("psfnet_adding_n128.npz")
    torch.manual_seed(42)
    net = PSFNet(vocab_size=1, add_init_linear_layer=True, embedding_size=32, n_vec=128, n_W=7, Ws=[32, 'GELU'],
                 V=[32, 'GELU'], n_channels_V=8, n_class=1, pooling_type="FLATTEN", head=['linear'],
                 use_cuda=False, use_residuals=True, use_pos_embedding=False, problem="adding")
    want = {k[4:]: g[k] for k in g.files if k.startswith("sd::")}
    sd = net.state_dict()
    assert set(sd) == set(want)
    # same construction order as the reference => the same seed draws the same initial weights
    for k, v in want.items():
        assert np.array_equal(sd[k].numpy(), v), k
    assert np.array_equal(net.chord_indicies.numpy(), g["chord_indicies"])
    for attr in ("fs", "g", "final", "embedding", "pos_embedding", "init_linear", "n_W", "n_links", "n_vec"):
        assert hasattr(net, attr)
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in want.items()}, strict=True)


def test_attention_block_surface(lib):
    from sparsefactorization_amd.attention_block import PSFNet
    net = PSFNet(vocab_size=256, embedding_size=16, max_seq_len=100, use_cuda=False, use_residuals=False,
                 dropout1_p=0, dropout2_p=0, dropout3_p=0)
    assert net.n_W == 7 and net.n_links == 8
    keys = set(net.state_dict())
    assert {"embedding.weight", "apc_embedding.weight", "g.network.0.weight", "fs.6.network.2.bias"} <= keys


def test_genome_surface(lib):
    from sparsefactorization_amd.genome_psf import PSFNet
    cfg = dict(IMDB)
    cfg.pop("problem")
    cfg.update(n_vec=64, n_W=6)
    net = PSFNet(**cfg)
    assert net.embedding.padding_idx is None


def test_library_carries_the_hash_of_its_sources_and_a_stale_one_is_reported(monkeypatch):
    """psf_build_info ends in csrc=<sha256 of csrc/ + the header>; build.needs_build and _lib.load compare it with the
    tree, so an edited kernel source cannot silently run on an old binary."""
    import warnings
    from sparsefactorization_amd import _lib, build
    lib = _lib.load()
    assert ("csrc=" + build.csrc_hash()).encode() in lib.psf_build_info()
    assert build.built_hash() == build.csrc_hash() and not build.needs_build()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        _lib._warn_if_stale(lib)  # fresh: silent
    monkeypatch.setattr(build, "csrc_hash", lambda: "0" * 64)
    assert build.needs_build()
    with pytest.warns(RuntimeWarning, match="built from other sources"):
        _lib._warn_if_stale(lib)


def test_fused_mixer_handoff_is_per_thread_and_uncovered_blocks_raise():
    """fused_mixer hands what its eligibility check found to the forward that follows (round 5: a third fewer Python calls).
    That hand-off is per thread and keyed: another thread's check can never give this forward another model's layers, and
    blocks the path does not cover raise a ValueError naming the requirement instead of a TypeError."""
    import threading
    from sparsefactorization_amd import fused_mixer
    from sparsefactorization_amd.psfnet import MLPBlock
    g, fs = MLPBlock([32, 'GELU'], 32, 8), [MLPBlock([32, 'GELU'], 32, 12) for _ in range(3)]
    found = fused_mixer._block_pairs(32, g, fs)
    assert found is not None and found[0][0] == 3 and found[0][2:] == (8, 12) and len(found[1]) == 4
    fused_mixer._handoff.pending = (fused_mixer._key(32, g, fs), found[0], found[1])
    seen = []
    t = threading.Thread(target=lambda: seen.append(getattr(fused_mixer._handoff, "pending", None)))
    t.start()
    t.join()
    assert seen == [None]  # the other thread sees nothing of this thread's pending hand-off
    fused_mixer._handoff.pending = None
    odd = [MLPBlock([32, 'GELU'], 32, 12), MLPBlock([32, 'GELU'], 32, 13)]  # link MLPs that disagree on L
    assert fused_mixer._block_pairs(32, g, odd) is None
    with pytest.raises(ValueError, match="does not cover these blocks"):
        fused_mixer.mixer_forward(torch.zeros(1, 64, 32), g, odd, True)
    assert getattr(fused_mixer._handoff, "pending", None) is None


def test_interleaved_fronts_visit_every_tile_once():
    """Geom::ileave (csrc/psf_common.h: decode_block; knob "bwd_fronts"): position t of an XCD's walk through a batch element
    is row block (t mod 2^s) * tiles / 2^s + t / 2^s. Restated here: for every tile count the host accepts (2^s | tiles) the
    map is a bijection, consecutive positions are tiles / 2^s apart (the fronts) and each front advances one tile per 2^s
    positions — speed only, never correctness; the GPU test test_fused_backward_fronts_only_reorder_workgroups checks the bits."""
    for s in (1, 2, 3):
        for tiles in (1 << s, 2 << s, 6 << s, 64, 128, 512, 1000 << s):
            if tiles % (1 << s):
                continue
            walk = [(t & ((1 << s) - 1)) * (tiles >> s) + (t >> s) for t in range(tiles)]
            assert sorted(walk) == list(range(tiles))
            assert walk[1] - walk[0] == tiles >> s
            if tiles > (1 << s):
                assert walk[1 << s] == 1
