"""Shared helpers for the test-suite (fixture loading, packed <-> list conversion)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_npz(name):
    with np.load(os.path.join(GOLDEN, name), allow_pickle=False) as f:
        return {k: f[k] for k in f.files}


def manifest_of(arrs, key="manifest"):
    return json.loads(str(arrs[key]))


def split_rows(packed, row_ptr):
    return [np.array(packed[row_ptr[i] : row_ptr[i + 1]]) for i in range(len(row_ptr) - 1)]


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    den = np.linalg.norm(b)
    return np.linalg.norm(a - b) / (den if den > 0 else 1.0)
