"""Stand-in for the ``torch_geometric`` import of the reference's training scripts, which use exactly one name of it:
``torch_geometric.data.DataLoader`` (SyntheticExperiments/psf_training.py:8,80-114; LRA/*_training.py)."""
from . import data  # noqa: F401

__all__ = ["data"]
__version__ = "1.7.2+sfa"  # the reference pins torch-geometric==1.7.2
