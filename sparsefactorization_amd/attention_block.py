"""Drop-in for ``attention_block.py``: ``from sparsefactorization_amd.attention_block import PSFNet``.

Unlike the reference file (attention_block.py:181-192) importing this module constructs nothing and prints nothing.
"""
from .chord import get_chord_indices_assym, spmm  # noqa: F401
from .psfnet import MakeMLP, MLPBlock  # noqa: F401
from .psfnet import AttentionBlockPSF as PSFNet  # noqa: F401
