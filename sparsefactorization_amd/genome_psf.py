"""Drop-in for ``Genome_Clf/psf.py``: ``from sparsefactorization_amd.genome_psf import PSFNet``."""
from .chord import get_chord_indices_assym, spmm  # noqa: F401
from .psfnet import MakeMLP, MLPBlock  # noqa: F401
from .psfnet import GenomePSFNet as PSFNet  # noqa: F401
