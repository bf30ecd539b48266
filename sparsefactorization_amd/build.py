"""Builds libpsf_chord.so (hand-written HIP for gfx950 + the C ABI of include/psf_chord.h) in-tree with hipcc.

The analogue of the reference's spmul/setup.py (CUDAExtension('spmul_cuda', ['spmul_cuda.cu']), lines 4-13),
except that the product is a plain C-ABI shared object bound with ctypes — no torch headers, no pybind11.

    python -m sparsefactorization_amd.build [--force]
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libpsf_chord.so")
ARCH = "gfx950"

SOURCES = ["psf_chord.hip"]
HEADERS = ["psf_common.h", "fwd_kernels.h", "bwd_kernels.h", os.path.join("..", "..", "include", "psf_chord.h")]

# -ffp-contract=off: products and sums stay separate roundings (bitwise parity with the CPU oracle).
HIPCC_FLAGS = ["-O3", f"--offload-arch={ARCH}", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
               "-Wall", "-Wno-pass-failed"]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, PATH and /opt/rocm/bin/hipcc)")


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    lib_m = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > lib_m for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into sparsefactorization_amd/libpsf_chord.so. Returns its path."""
    if not force and not needs_build():
        return LIB_PATH
    cmd = [hipcc(), *HIPCC_FLAGS, *[os.path.join(CSRC, s) for s in SOURCES], "-o", LIB_PATH + ".tmp"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError(f"hipcc failed ({proc.returncode}):\n{proc.stderr[-4000:]}")
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    path = build(force="--force" in sys.argv, verbose=True)
    print(path)
