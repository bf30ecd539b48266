"""ctypes binding of libpsf_chord.so — one Python function per entry point of include/psf_chord.h.

There is no fallback: if the HIP library is missing or cannot be loaded, every operator raises.
"""
from __future__ import annotations

import ctypes
import os
import threading
from typing import Optional, Sequence

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG_DIR, "libpsf_chord.so")
ABI_VERSION = 2
MAX_LINKS = 64

c_i32, c_i64, c_vp = ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p
_I64P = ctypes.POINTER(c_i64)

# name -> argtypes (all return int unless noted); mirrors include/psf_chord.h exactly
_STEP = [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i32, c_i64, c_i64, _I64P, c_vp]
_BWD = [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i32, c_i64, c_i64, _I64P, c_vp]
_CHAIN = [ctypes.POINTER(c_vp), c_vp, ctypes.POINTER(c_vp), c_i32, c_i32, c_i64, c_i64, c_i32, c_i64, c_i64, _I64P, c_vp]
SIGNATURES = {
    "psf_version": ([], ctypes.c_int),
    "psf_last_error": ([], ctypes.c_char_p),
    "psf_build_info": ([], ctypes.c_char_p),
    "psf_device_info": ([ctypes.c_char_p, c_i32], ctypes.c_int),
    "psf_chord_offsets": ([c_i64, c_i32, _I64P], ctypes.c_int),
    "psf_chord_indices": ([c_i64, c_i32, _I64P, _I64P], ctypes.c_int),
    "psf_chord_spmm_fwd_f32": (_STEP, ctypes.c_int),
    "psf_chord_spmm_fwd_f64": (_STEP, ctypes.c_int),
    "psf_chord_spmm_bwd_f32": (_BWD, ctypes.c_int),
    "psf_chord_spmm_bwd_f64": (_BWD, ctypes.c_int),
    "psf_chord_chain_fwd_f32": (_CHAIN, ctypes.c_int),
    "psf_chord_chain_fwd_f64": (_CHAIN, ctypes.c_int),
    "psf_chord_chain_bwd_supported": ([c_i64, c_i32, c_i64, c_i32], ctypes.c_int),
    "psf_chord_chain_bwd_f32": ([c_vp, ctypes.POINTER(c_vp), c_vp, ctypes.POINTER(c_vp), ctypes.POINTER(c_vp), c_vp, ctypes.POINTER(c_vp), c_i32, c_i32,
                                 c_i64, c_i64, c_i32, c_i64, _I64P, c_vp], ctypes.c_int),
    "psf_linear_wgrad_workspace": ([c_i64, c_i32, c_i32], c_i64),
    "psf_linear_wgrad_f32": ([c_vp, c_vp, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp, c_i64, c_vp], ctypes.c_int),
    "psf_linear_wgrad_strided_f32": ([c_vp, c_i64, c_vp, c_i64, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp, c_i64, c_vp], ctypes.c_int),
    "psf_adam_step_f32": ([ctypes.POINTER(c_vp), ctypes.POINTER(c_vp), ctypes.POINTER(c_vp), ctypes.POINTER(c_vp), _I64P, c_i32,
                           ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, c_vp, c_vp], ctypes.c_int),
    "psf_sum_tensors_f32": ([ctypes.POINTER(c_vp), c_i32, c_i64, c_vp, c_vp], ctypes.c_int),
    "psf_embed_tokens_f32": ([c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i32, c_i32, c_vp], ctypes.c_int),
    "psf_affine_rows_f32": ([c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_vp], ctypes.c_int),
    "psf_embed_tokens_bwd_workspace": ([c_i64, c_i32, c_i32], c_i64),
    "psf_embed_tokens_bwd_f32": ([c_vp, c_vp, c_i64, c_i32, c_i32, c_vp, c_vp, c_i64, c_vp], ctypes.c_int),
    "psf_flat_head_workspace": ([c_i32, c_i64, c_i32], c_i64),
    "psf_flat_head_f32": ([c_vp, c_vp, c_vp, c_vp, c_i32, c_i64, c_i32, c_vp, c_i64, c_vp], ctypes.c_int),
    "psf_flat_head_bwd_f32": ([c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i64, c_i32, c_vp], ctypes.c_int),
    "psf_mlp_fwd_workspace": ([c_i32, c_i32, ctypes.POINTER(c_i32), ctypes.POINTER(c_i32)], c_i64),
    "psf_mlp_fwd_f32": ([c_vp, c_i64, c_i32, c_i32, ctypes.POINTER(c_vp), ctypes.POINTER(c_vp), ctypes.POINTER(c_vp),
                         ctypes.POINTER(c_vp), ctypes.POINTER(c_i32), ctypes.POINTER(c_i32), ctypes.POINTER(c_vp), c_vp,
                         c_i64, c_vp], ctypes.c_int),
    "psf_mixer_fwd_workspace": ([c_i64, c_i32, c_i32, ctypes.POINTER(c_i32), c_i64, c_i32], c_i64),
    "psf_mixer_fwd_plan": ([c_i64, c_i32, c_i32, ctypes.POINTER(c_i32), c_i64, c_i32], c_i32),
    "psf_mixer_fwd_f32": ([c_vp, c_i64, c_i64, c_i32, c_i32, ctypes.POINTER(c_vp), ctypes.POINTER(c_vp), ctypes.POINTER(c_vp),
                           ctypes.POINTER(c_vp), ctypes.POINTER(c_i32), c_i64, c_i32, c_i32, c_vp, ctypes.POINTER(c_vp), c_vp,
                           c_i64, c_vp], ctypes.c_int),
    "psf_mixer_fwd_in_f32": ([c_vp, c_i64, c_i64, c_i32, c_i32, ctypes.POINTER(c_vp), ctypes.POINTER(c_vp), ctypes.POINTER(c_vp),
                              ctypes.POINTER(c_vp), ctypes.POINTER(c_i32), c_i64, c_i32, c_i32, c_vp, ctypes.POINTER(c_vp), c_vp,
                              c_i64, c_vp], ctypes.c_int),
    "psf_mlp_bwd_workspace": ([c_i64, c_i32, c_i32, ctypes.POINTER(c_i32), ctypes.POINTER(c_i32)], c_i64),
    "psf_mlp_bwd_f32": ([c_vp, c_i64, c_i32, c_i32, ctypes.POINTER(c_vp), ctypes.POINTER(c_vp), ctypes.POINTER(c_vp),
                         ctypes.POINTER(c_i32), ctypes.POINTER(c_i32), ctypes.POINTER(c_vp), c_vp, ctypes.POINTER(c_vp),
                         ctypes.POINTER(c_vp), ctypes.POINTER(c_vp), ctypes.POINTER(c_vp), c_vp, c_i64, c_vp], ctypes.c_int),
    "psf_mlp_wide_saved_bytes": ([c_i64, c_i32, c_i32, ctypes.POINTER(c_i32), ctypes.POINTER(c_i32)], c_i64),
    "psf_mlp_wide_fwd_workspace": ([c_i64, c_i32, c_i32, ctypes.POINTER(c_i32), ctypes.POINTER(c_i32)], c_i64),
    "psf_mlp_wide_bwd_workspace": ([c_i64, c_i32, c_i32, ctypes.POINTER(c_i32), ctypes.POINTER(c_i32)], c_i64),
    "psf_mlp_wide_fwd_f32": ([c_vp, c_i64, c_i32, c_i32, ctypes.POINTER(c_vp), ctypes.POINTER(c_vp), ctypes.POINTER(c_vp),
                              ctypes.POINTER(c_vp), ctypes.POINTER(c_i32), ctypes.POINTER(c_i32), ctypes.POINTER(c_vp), c_vp,
                              c_i64, c_vp, c_i64, c_vp], ctypes.c_int),
    "psf_mlp_wide_bwd_f32": ([c_vp, c_i64, c_i64, c_i32, c_i32, ctypes.POINTER(c_vp), ctypes.POINTER(c_vp),
                              ctypes.POINTER(c_i32), ctypes.POINTER(c_i32), ctypes.POINTER(c_vp), c_vp, ctypes.POINTER(c_vp),
                              ctypes.POINTER(c_vp), ctypes.POINTER(c_vp), ctypes.POINTER(c_vp), c_vp, c_i64, c_vp], ctypes.c_int),
    "psf_stream_mix_f32": ([c_vp, c_vp, c_vp, c_vp, c_i64, c_vp], ctypes.c_int),
    "psf_stream_mix_bwd_f32": ([c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp], ctypes.c_int),
    "psf_set_tuning": ([ctypes.c_char_p, c_i32], ctypes.c_int),
    "psf_get_tuning": ([ctypes.c_char_p], ctypes.c_int),
    "psf_describe_fwd": ([c_i64, c_i64, c_i32, c_i64, c_i32, ctypes.c_char_p, c_i32], ctypes.c_int),
    "psf_describe_chain_fwd": ([c_i64, c_i64, c_i32, c_i64, c_i32, ctypes.c_char_p, c_i32], ctypes.c_int),
}


class MixerInput(ctypes.Structure):
    """``psf_mixer_input`` of include/psf_chord.h."""
    _fields_ = [("kind", c_i32), ("K", c_i32), ("src", c_vp), ("weight", c_vp), ("bias", c_vp), ("pos", c_vp)]


MIXER_IN_DATA, MIXER_IN_AFFINE, MIXER_IN_TOKENS = 0, 1, 2


class PSFLibraryError(RuntimeError):
    """libpsf_chord.so is missing, stale or failed a call."""


_lock = threading.Lock()
_lib: Optional[ctypes.CDLL] = None


def load() -> ctypes.CDLL:
    """Load the in-tree HIP library. Raises PSFLibraryError (never falls back) if that is impossible."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise PSFLibraryError(
                f"{LIB_PATH} not found: the HIP extension is not built. Run "
                "`python -m sparsefactorization_amd.build` (needs hipcc, cross-compiles gfx950 without a GPU). "
                "There is no CPU or PyTorch fallback for the chord-spmm path.")
        try:
            lib = ctypes.CDLL(LIB_PATH)
        except OSError as e:
            raise PSFLibraryError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (argtypes, restype) in SIGNATURES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError as e:
                raise PSFLibraryError(f"{LIB_PATH} does not export {name}; rebuild it") from e
            fn.argtypes = argtypes
            fn.restype = restype
        v = lib.psf_version()
        if v != ABI_VERSION:
            raise PSFLibraryError(f"{LIB_PATH} has ABI version {v}, this package needs {ABI_VERSION}; rebuild it")
        _warn_if_stale(lib)
        _lib = lib
    return _lib


def _warn_if_stale(lib) -> None:
    """A library built from other kernel sources than the ones in csrc/ (an edit without a rebuild) is loaded, but loudly."""
    import re
    import warnings
    csrc = os.path.join(PKG_DIR, "csrc")
    if not os.path.isdir(csrc):
        return  # a binary-only installation: nothing to compare with
    m = re.search(r"csrc=([0-9a-f]{64})", lib.psf_build_info().decode("utf-8", "replace"))
    from .build import csrc_hash
    want = csrc_hash()
    if m is None or m.group(1) != want:
        warnings.warn(f"{LIB_PATH} was built from other sources than sparsefactorization_amd/csrc "
                      f"(library {m.group(1)[:12] if m else 'unstamped'}, tree {want[:12]}): run "
                      "`python -m sparsefactorization_amd.build`", RuntimeWarning, stacklevel=3)


def last_error() -> str:
    return load().psf_last_error().decode("utf-8", "replace")


_torch_stream = None  # (raw getter or False, torch.cuda): filled on first use — this module does not import torch at load


def stream_ptr(dev) -> int:
    """The current HIP stream of ``dev`` as the integer the C ABI takes. torch.cuda.current_stream builds a Stream object on every
    call (3-5 us; a dozen calls per training step of a model whose step the host bounds); the raw getter returns the handle."""
    global _torch_stream
    if _torch_stream is None:
        import torch
        _torch_stream = (getattr(torch._C, "_cuda_getCurrentRawStream", False), torch.cuda)
    raw, cuda = _torch_stream
    if raw:
        idx = dev.index
        return raw(cuda.current_device() if idx is None else idx)
    return cuda.current_stream(dev).cuda_stream


PSF_E_UNSUPPORTED = -7  # psf_chord_chain_bwd_f32: no one-launch kernel for the shape (the caller runs the steps)


def check(rc: int, what: str) -> None:
    if rc != 0:
        kind = "invalid argument" if rc < 0 else "HIP error"
        raise PSFLibraryError(f"{what} failed ({kind} {rc}): {last_error()}")


def offsets_array(offsets: Optional[Sequence[int]]):
    """Host int64 array for the `offsets` parameter, or NULL for the chord pattern."""
    if offsets is None:
        return None
    arr = (c_i64 * len(offsets))(*[int(o) for o in offsets])
    return arr


def chord_offsets(N: int, L: int) -> list:
    buf = (c_i64 * L)()
    check(load().psf_chord_offsets(N, L, buf), "psf_chord_offsets")
    return list(buf)


def set_tuning(key: str, value: int) -> None:
    check(load().psf_set_tuning(key.encode(), value), f"psf_set_tuning({key})")


def get_tuning(key: str) -> int:
    v = load().psf_get_tuning(key.encode())
    if v < 0:
        raise PSFLibraryError(f"psf_get_tuning({key}): {last_error()}")
    return v


def describe_fwd(B: int, N: int, L: int, C: int, elem_bytes: int = 4) -> str:
    buf = ctypes.create_string_buffer(256)
    check(load().psf_describe_fwd(B, N, L, C, elem_bytes, buf, 256), "psf_describe_fwd")
    return buf.value.decode()


def describe_chain_fwd(B: int, N: int, L: int, C: int, M: int) -> str:
    buf = ctypes.create_string_buffer(256)
    check(load().psf_describe_chain_fwd(B, N, L, C, M, buf, 256), "psf_describe_chain_fwd")
    return buf.value.decode()


def build_info() -> str:
    return load().psf_build_info().decode()


def device_info() -> str:
    """The calling thread's current HIP device as the library sees it (PCI bus id, XCDs, CUs, name)."""
    buf = ctypes.create_string_buffer(256)
    check(load().psf_device_info(buf, 256), "psf_device_info")
    return buf.value.decode()
