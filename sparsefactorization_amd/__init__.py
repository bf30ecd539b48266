"""sparsefactorization_amd — PSF-Attn's chord-sparse batched matmul chain, hand-written for MI355X (gfx950).

One hot path of RuslanKhalitov/SparseFactorization, rebuilt MI355X-first: the product W_M ... W_1 V with
chord-structured sparse W_m that PSFNet.forward runs through torch_sparse.spmm
(SyntheticExperiments/psf.py:172-188). Everything numeric happens in libpsf_chord.so (HIP kernels behind the C
ABI of include/psf_chord.h); this package is the thin PyTorch-ROCm host side that keeps the reference's
Python surface. Importing the package does not need a GPU; calling an operator does, and needs the built
library — nothing falls back to the CPU.
"""
from ._lib import PSFLibraryError, build_info, describe_fwd, get_tuning, set_tuning
from .chord import chord_chain, chord_spmm, get_chord_indices_assym, offsets_from_index, spmm
from .spmul import SparseMultiply, get_offsets

__all__ = [
    "spmm", "chord_spmm", "chord_chain", "get_chord_indices_assym", "offsets_from_index",
    "SparseMultiply", "get_offsets", "PSFLibraryError", "build_info", "describe_fwd", "set_tuning", "get_tuning",
]
__version__ = "0.1.0"
