"""Import shims that let the reference's scripts run unedited (see README.md in this directory)."""
import os
import sys

SHIM_DIR = os.path.dirname(os.path.abspath(__file__))


def install() -> str:
    """Append the shim directory to ``sys.path`` (append, not insert: a real torch_sparse / torch_geometric wins)."""
    if SHIM_DIR not in sys.path:
        sys.path.append(SHIM_DIR)
    return SHIM_DIR
