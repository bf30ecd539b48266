"""Stand-in for the ``torch_sparse`` import of the reference: ``from torch_sparse import spmm``
(SyntheticExperiments/psf.py:5, LRA/psf.py:5, Genome_Clf/psf.py:5, attention_block.py:10).

``spmm`` is the operator, and only for the chord pattern PSFNet passes (get_chord_indices_assym, psf.py:7-32): any other
index list raises ``ValueError``; CPU tensors raise too (there is no CPU path). It is the lazy operator — the
reference's unmodified loop ``V = spmm(...); V = V + res_conn`` is recorded and runs as ONE chord-chain library call.
``SFA_SHIM_EAGER=1`` selects the eager per-step operator instead.

``spspmm`` is importable because LRA/attention_maps/pathfinder_inference.py:9 imports it (``from torch_sparse import spmm,
spspmm``) without ever calling it; calling it raises.
"""
import os as _os

if _os.environ.get("SFA_SHIM_EAGER"):
    from sparsefactorization_amd.chord import spmm  # noqa: F401
else:
    from sparsefactorization_amd.lazy import spmm  # noqa: F401



def spspmm(*args, **kwargs):
    """Sparse x sparse product of torch_sparse: imported by pathfinder_inference.py:9, never called by the reference and
    outside the chord path — not provided."""
    raise NotImplementedError("torch_sparse.spspmm is not part of the chord-spmm path; sparsefactorization_amd's shim only "
                              "provides spmm (the reference imports spspmm in pathfinder_inference.py:9 and never calls it)")


__all__ = ["spmm", "spspmm"]
__version__ = "0.6.11+sfa"  # the reference pins torch-sparse==0.6.11 (requirements.txt:146)
