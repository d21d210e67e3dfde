#!/usr/bin/env python3
"""Writes tests/golden/function_vectors.npz from the current oracle (see vectors.py for what it holds).
Run from the repository root after a deliberate change of the oracle:  python tests/golden/make_vectors.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import vectors  # noqa: E402

if __name__ == "__main__":
    out = vectors.compute()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "function_vectors.npz")
    np.savez_compressed(path, **out)
    print(path, "%.1f KB" % (os.path.getsize(path) / 1e3), len(out), "arrays")
