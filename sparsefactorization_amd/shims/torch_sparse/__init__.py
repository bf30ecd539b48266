"""Stand-in for the ``torch_sparse`` import of the reference: ``from torch_sparse import spmm``
(SyntheticExperiments/psf.py:5, LRA/psf.py:5, Genome_Clf/psf.py:5, attention_block.py:10).

Only ``spmm`` exists here, and only for the chord pattern PSFNet passes (get_chord_indices_assym, psf.py:7-32): any other
index list raises ``ValueError``; CPU tensors raise too (there is no CPU path). It is the lazy operator — the
reference's unmodified loop ``V = spmm(...); V = V + res_conn`` is recorded and runs as ONE chord-chain library call.
``SFA_SHIM_EAGER=1`` selects the eager per-step operator instead.
"""
import os as _os

if _os.environ.get("SFA_SHIM_EAGER"):
    from sparsefactorization_amd.chord import spmm  # noqa: F401
else:
    from sparsefactorization_amd.lazy import spmm  # noqa: F401

__all__ = ["spmm"]
__version__ = "0.6.11+sfa"  # the reference pins torch-sparse==0.6.11 (requirements.txt:146)
