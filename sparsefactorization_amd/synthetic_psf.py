"""Drop-in for ``SyntheticExperiments/psf.py``: ``from sparsefactorization_amd.synthetic_psf import PSFNet``."""
from .chord import get_chord_indices_assym, spmm  # noqa: F401
from .psfnet import MakeMLP, MLPBlock  # noqa: F401
from .psfnet import SyntheticPSFNet as PSFNet  # noqa: F401
