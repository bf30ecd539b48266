"""Builds libpsf_chord.so (hand-written HIP for gfx950 + the C ABI of include/psf_chord.h) in-tree with hipcc.

The analogue of the reference's spmul/setup.py (CUDAExtension('spmul_cuda', ['spmul_cuda.cu']), lines 4-13),
except that the product is a plain C-ABI shared object bound with ctypes — no torch headers, no pybind11.

    python -m sparsefactorization_amd.build [--force]

Translation units: psf_chord.hip (C ABI, dispatch, generic kernels) and fwd_window_inst.hip compiled once per
channel-group shift (-DPSF_TGS=0..6) — the window kernels are a few hundred template instances, so they are
compiled in parallel and linked into one library.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
import threading
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
OBJ_DIR = os.path.join(PKG_DIR, "build")
LIB_PATH = os.path.join(PKG_DIR, "libpsf_chord.so")
ARCH = "gfx950"
WIN_TGS = list(range(7))

HEADERS = ["psf_common.h", "fwd_kernels.h", "fwd_window.h", "fwd_window_launch.h", "bwd_kernels.h",
           "bwd_window.h", "bwd_dw_chunk.h", "bwd_window_launch.h", "fwd_chain_lds.h", "fwd_chain_lds_launch.h", "bwd_chain_lds.h", "fwd_mlp_step.h", "fwd_mlp_step_launch.h", "mixer_lds.h", "mixer_lds_launch.h", "mlp_x3_image.h", "mlp_fwd_x3.h", "mlp_x3_common.h", "mlp_planes.h", "x3_gemm.h",
           os.path.join("..", "..", "include", "psf_chord.h")]
SOURCES = ["psf_chord.hip", "fwd_window_inst.hip", "bwd_window_inst.hip", "linear_wgrad.hip",
           "fwd_chain_lds_inst.hip", "bwd_chain_lds_inst.hip", "fwd_mlp_step_inst.hip", "mixer_lds_inst.hip", "embed.hip", "flat_head.hip", "sum_tensors.hip", "adam.hip", "mlp_fwd.hip", "mlp_fwd_x3.hip", "mlp_bwd.hip", "mlp_wide.hip", "stream_mix.hip"]

# -ffp-contract=off: products and sums stay separate roundings (bitwise parity with the CPU oracle).
HIPCC_FLAGS = ["-O3", f"--offload-arch={ARCH}", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall",
               "-Wno-pass-failed", *os.environ.get("PSF_HIPCC_EXTRA", "").split()]  # PSF_HIPCC_EXTRA: diagnostic builds


def csrc_hash() -> str:
    """SHA-256 over the kernel sources (every file under csrc/ plus include/psf_chord.h, names and contents, sorted).
    Identifies the code a profile was collected on: profiles/*_pmc.json records it and bench.py withholds
    ``roofline.traffic`` when it no longer matches (no .git on the GPU box, so a content hash instead of a commit)."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hip")))
    files.append(os.path.join(PKG_DIR, "..", "include", "psf_chord.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    # the effective flag list of every unit (a diagnostic build with PSF_HIPCC_EXTRA / PSF_MLP_STEP_EXTRA, or one made with
    # other per-unit flags, must not carry the product hash: _lib.load's staleness check and bench.py's PMC match use it)
    h.update(" ".join(HIPCC_FLAGS).encode())
    for obj, _src, extra, _lint in _unit_table():
        h.update((os.path.basename(obj) + " " + " ".join(extra)).encode())
    return h.hexdigest()


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, PATH and /opt/rocm/bin/hipcc)")


def _units():
    """(object path, source, extra flags, lint) for every translation unit; psf_chord.o carries the source hash."""
    table = _unit_table()
    define = f'-DPSF_CSRC_HASH="{csrc_hash()}"'
    return [(o, s, [*e, define] if os.path.basename(o) == "psf_chord.o" else e, l) for o, s, e, l in table]


def _unit_table():
    """The units without the hash define (csrc_hash() folds their flags in). `lint`: the unit's gfx950 assembly is kept and
    checked by isa_lint.py — since round 6 EVERY unit (R0-R2 are errors everywhere; R3-pattern sites are counted everywhere,
    reported as warnings in their failing form, and errors only in R3_STRICT_UNITS: isa_lint.py's docstring)."""
    units = [(os.path.join(OBJ_DIR, "psf_chord.o"), os.path.join(CSRC, "psf_chord.hip"), []),
             (os.path.join(OBJ_DIR, "linear_wgrad.o"), os.path.join(CSRC, "linear_wgrad.hip"), []),
             # -fno-slp-vectorize: the LDS-resident chain's multiply-add loop is faster on scalar f32 instructions than on the
             # v_pk_* pairs hipcc builds from it (Pathfinder chain 41.4 -> 33.9 us, profiles/r04p_lib_ab_noslp.log; the
             # per-step window kernels are the other way round: 367 -> 386 us at cfg2)
             (os.path.join(OBJ_DIR, "fwd_chain_lds.o"), os.path.join(CSRC, "fwd_chain_lds_inst.hip"), ["-fno-slp-vectorize"]),
             (os.path.join(OBJ_DIR, "bwd_chain_lds.o"), os.path.join(CSRC, "bwd_chain_lds_inst.hip"), ["-fno-slp-vectorize"]),
             (os.path.join(OBJ_DIR, "embed.o"), os.path.join(CSRC, "embed.hip"), []),
             (os.path.join(OBJ_DIR, "flat_head.o"), os.path.join(CSRC, "flat_head.hip"), []),
             (os.path.join(OBJ_DIR, "sum_tensors.o"), os.path.join(CSRC, "sum_tensors.hip"), []),
             (os.path.join(OBJ_DIR, "adam.o"), os.path.join(CSRC, "adam.hip"), []),
             (os.path.join(OBJ_DIR, "stream_mix.o"), os.path.join(CSRC, "stream_mix.hip"), []),
             (os.path.join(OBJ_DIR, "mixer_lds.o"), os.path.join(CSRC, "mixer_lds_inst.hip"), []),
             (os.path.join(OBJ_DIR, "mlp_fwd.o"), os.path.join(CSRC, "mlp_fwd.hip"), []),
             (os.path.join(OBJ_DIR, "mlp_fwd_x3.o"), os.path.join(CSRC, "mlp_fwd_x3.hip"), []),
             # -fno-slp-vectorize: keeps hipcc from re-packing the scalar f32 arithmetic of the bf16-pipe kernel into
             # v_pk_* instructions, which are slow beside MFMAs (mlp_bwd.hip, gelu_and_grad1)
             (os.path.join(OBJ_DIR, "mlp_bwd.o"), os.path.join(CSRC, "mlp_bwd.hip"), ["-fno-slp-vectorize"]),
             (os.path.join(OBJ_DIR, "mlp_wide.o"), os.path.join(CSRC, "mlp_wide.hip"), ["-fno-slp-vectorize"])]
    for t in WIN_TGS:
        units.append((os.path.join(OBJ_DIR, f"fwd_window_tgs{t}.o"), os.path.join(CSRC, "fwd_window_inst.hip"),
                      [f"-DPSF_TGS={t}"]))
        units.append((os.path.join(OBJ_DIR, f"bwd_window_tgs{t}.o"), os.path.join(CSRC, "bwd_window_inst.hip"),
                      [f"-DPSF_TGS={t}"]))
    # the forward step that computes its own W tile (fwd_mlp_step_launch.h: kMlpStepTgsMax)
    for t in range(4):
        units.append((os.path.join(OBJ_DIR, f"fwd_mlp_step_tgs{t}.o"), os.path.join(CSRC, "fwd_mlp_step_inst.hip"),
                      [f"-DPSF_TGS={t}", *os.environ.get("PSF_MLP_STEP_EXTRA", "").split()]))
    # dV at 512 threads x 1 row per thread for narrow rows (bwd_window_launch.h: kDvMidThreads, kDvMidTgsMax)
    for t in range(6):
        units.append((os.path.join(OBJ_DIR, f"bwd_window_mid_tgs{t}.o"), os.path.join(CSRC, "bwd_window_inst.hip"),
                      [f"-DPSF_TGS={t}", "-DPSF_NT=512"]))
    # wide-row configuration (fwd_window_launch.h: kWideTgs, kWideThreads)
    wide = ["-DPSF_TGS=3", "-DPSF_NT=1024"]
    units.append((os.path.join(OBJ_DIR, "fwd_window_wide.o"), os.path.join(CSRC, "fwd_window_inst.hip"), wide))
    units.append((os.path.join(OBJ_DIR, "bwd_window_wide.o"), os.path.join(CSRC, "bwd_window_inst.hip"), wide))
    return [(o, s, e, True) for o, s, e in units]


# the kernels the round-4 wrong result was seen in (and their single-launch sibling): failing-form R3 sites are errors here
R3_STRICT_UNITS = ("fwd_mlp_step", "mixer_lds")
LINT_LOG = os.path.join(OBJ_DIR, "isa_lint.log")  # one line per unit: functions, errors, notes, R3-pattern sites


def _unit_weight(unit) -> int:
    """Rough compile cost of a unit (seconds on this container), for the order in which the pool starts them."""
    name = os.path.basename(unit[0])
    for prefix, w in (("fwd_mlp_step", 30), ("mlp_bwd", 28), ("fwd_chain_lds", 26), ("mlp_wide", 24), ("fwd_window", 16),
                      ("bwd_window_mid", 14), ("bwd_window", 12), ("mixer_lds", 10), ("mlp_fwd", 8), ("psf_chord", 8)):
        if name.startswith(prefix):
            return w
    return 3


def built_hash(path: str = LIB_PATH):
    """The ``csrc=<sha256>`` a library carries in its psf_build_info string (read from the file: no dlopen), or None."""
    import re
    try:
        with open(path, "rb") as fh:
            m = re.search(rb"csrc=([0-9a-f]{64})", fh.read())
    except OSError:
        return None
    return m.group(1).decode() if m else None


def needs_build() -> bool:
    """True when the library is missing or was built from other sources than the ones in csrc/ now (content hash, not
    timestamps: an edited header that no list names, or a checkout that restores old mtimes, cannot leave a stale .so)."""
    return not os.path.exists(LIB_PATH) or built_hash() != csrc_hash()


_LOG_LOCK = threading.Lock()


def _compile(unit, cc, verbose):
    obj, src, extra, lint = unit
    if lint:  # keep the unit's assembly (in a directory of its own: the temporaries are named after the source file)
        tmp = obj[:-2] + ".tmp.d"
        shutil.rmtree(tmp, ignore_errors=True)
        os.makedirs(tmp)
        real_obj, obj = obj, os.path.join(tmp, os.path.basename(obj))
        extra = [*extra, "-save-temps=obj"]
    cmd = [cc, *HIPCC_FLAGS, *extra, "-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError(f"hipcc failed on {os.path.basename(src)} {extra} ({proc.returncode}):\n{proc.stderr[-4000:]}")
    if lint:
        from . import isa_lint
        asm = [os.path.join(tmp, f) for f in os.listdir(tmp) if f.endswith(f"{ARCH}.s")]
        if len(asm) != 1:
            raise RuntimeError(f"isa_lint: expected one {ARCH} assembly file for {os.path.basename(real_obj)}, found {asm}")
        sites: dict = {}
        unit_name = os.path.basename(real_obj)[:-2]
        errs, notes, nfun = isa_lint.lint_file(asm[0], sites, unit_name.startswith(R3_STRICT_UNITS))
        warn = [w for w in notes if w.startswith("WARNING")]
        report = (f"isa_lint {unit_name}: {nfun} functions, {len(errs)} errors, {len(notes)} notes"
                  + (f" ({len(warn)} of them R3 warnings: failing-form sites)" if warn else "") + "\n"
                  + isa_lint.r3_summary(unit_name, sites) + "\n" + "".join(f"  note {w}\n" for w in notes[:4]))
        with _LOG_LOCK:
            with open(LINT_LOG, "a") as fh:
                fh.write(report)
        if verbose:
            print(report, end="", file=sys.stderr)
        if errs:
            raise RuntimeError(f"isa_lint rejects {os.path.basename(real_obj)} (kept: {asm[0]}):\n" + "\n".join(errs[:20]))
        os.replace(obj, real_obj)
        shutil.rmtree(tmp, ignore_errors=True)
        obj = real_obj
    return obj


def build(force: bool = False, verbose: bool = False, jobs: int | None = None) -> str:
    """Compile every HIP source for gfx950 into sparsefactorization_amd/libpsf_chord.so. Returns its path."""
    if not force and not needs_build():
        return LIB_PATH
    cc = hipcc()
    os.makedirs(OBJ_DIR, exist_ok=True)
    open(LINT_LOG, "w").close()
    units = _units()
    jobs = jobs or min(len(units), max(1, (os.cpu_count() or 2)))
    # longest units first (the pool takes them in order: a 25 s unit started last would be the build's tail)
    order = sorted(range(len(units)), key=lambda i: -_unit_weight(units[i]))
    with ThreadPoolExecutor(max_workers=jobs) as pool:
        done = list(pool.map(lambda i: _compile(units[i], cc, verbose), order))
    objs = [None] * len(units)
    for i, o in zip(order, done):
        objs[i] = o
    link = [cc, f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-o", LIB_PATH + ".tmp"]
    if verbose:
        print(" ".join(link), file=sys.stderr)
    proc = subprocess.run(link, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError(f"link failed ({proc.returncode}):\n{proc.stderr[-4000:]}")
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    path = build(force="--force" in sys.argv, verbose=True)
    print(path)
