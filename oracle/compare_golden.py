#!/usr/bin/env python3
"""Compares a freshly generated set of fixtures (PSF_GOLDEN_OUT=<dir> python oracle/gen_golden.py) with tests/golden/:
every array of every .npz, bit for bit. TEST INFRASTRUCTURE.   python oracle/compare_golden.py <dir>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    fresh = sys.argv[1]
    committed = os.path.join(ROOT, "tests", "golden")
    bad = 0
    names = sorted(f for f in os.listdir(fresh) if f.endswith(".npz"))
    for name in names:
        a, b = np.load(os.path.join(fresh, name), allow_pickle=False), np.load(os.path.join(committed, name), allow_pickle=False)
        same = sorted(a.files) == sorted(b.files) and all(
            a[k].dtype == b[k].dtype and a[k].shape == b[k].shape and a[k].tobytes() == b[k].tobytes() for k in a.files)
        print(f"{name}: {'identical' if same else 'DIFFERENT'} ({len(a.files)} arrays)")
        bad += not same
    missing = sorted(set(f for f in os.listdir(committed) if f.endswith(".npz")) - set(names))
    if missing:
        print("not regenerated:", missing)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
