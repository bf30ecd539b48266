"""Drop-in for ``LRA/psf.py`` (and ``LRA/attention_maps/psf.py``): ``from sparsefactorization_amd.lra_psf import PSFNet``."""
from .chord import get_chord_indices_assym, spmm  # noqa: F401
from .psfnet import ChangedPSF, MakeMLP, MLPBlock  # noqa: F401
from .psfnet import LRAPSFNet as PSFNet  # noqa: F401
