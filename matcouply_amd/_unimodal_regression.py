"""Host-callable unimodal regression with the names and return conventions of the reference's module
(`/root/reference/src/matcouply/_unimodal_regression.py`): `prefix_isotonic_regression` (:27-69),
`_compute_isotonic_from_index` (:73-81), `_get_best_unimodality_index` (:84-92), `_unimodal_regression` (:95-104) and
`unimodal_regression` (:107-141).  Plain NumPy, fp64 - the plugin surface for penalty authors and post-processing.
The solver never calls it: inside `cmf_aoadmm` the projection runs in the HIP kernel `k_slab_unimodal_v4`
(`csrc/generic.hip`), one lane per column.

Only unit weights are implemented (the reference's decomposition never passes any others, penalties.py:1014-1015).
"""
import numpy as np

from .penalties import _fit_of_prefix, _prefix_isotonic


def prefix_isotonic_regression(y, weights=None, non_negativity=False):
    """Isotonic (non-decreasing) regression of every prefix y[:k] in O(n) (Stout 2008).

    Returns ``(level_set, index_range), error``: `level_set[i]` / `index_range[i]` are the level and the start index of the
    block that ends at position i in the fit of the prefix y[:i+1]; `error[k]` is the squared error of the fit of y[:k]
    (k = 0..n)."""
    y = np.asarray(y, dtype=float)
    if weights is not None and not np.all(np.asarray(weights) == 1):
        raise NotImplementedError("matcouply_amd: only unit weights are supported")
    level, start, err = _prefix_isotonic(y, bool(non_negativity))
    return (level, start), err


def _compute_isotonic_from_index(end_index, level_set, index_range):
    """Fit of the prefix y[:end_index] from the per-position block records (chain of blocks walked from the end)."""
    if end_index is None:
        end_index = len(level_set)
    return _fit_of_prefix(int(end_index), np.asarray(level_set), np.asarray(index_range))


def _get_best_unimodality_index(error_left, error_right):
    """Split t minimising error_left[t] + error_right[n - t]; ties go to the smallest t (strict '<' scan from t = 0)."""
    n = len(error_left) - 1
    best_error, best_idx = error_right[n], 0
    for i in range(n + 1):
        e = error_left[i] + error_right[n - i]
        if e < best_error:
            best_error, best_idx = e, i
    return best_idx, best_error


def _unimodal_regression(y, non_negativity):
    """Unimodal fit of a vector and its squared error."""
    y = np.asarray(y, dtype=float)
    (lvl_l, st_l), err_l = prefix_isotonic_regression(y, non_negativity=non_negativity)
    (lvl_r, st_r), err_r = prefix_isotonic_regression(y[::-1], non_negativity=non_negativity)
    split, error = _get_best_unimodality_index(err_l, err_r)
    left = _compute_isotonic_from_index(split, lvl_l, st_l)
    right = _compute_isotonic_from_index(len(y) - split, lvl_r, st_r)
    return np.concatenate([left, right[::-1]]), error


def unimodal_regression(y, non_negativity=False):
    """Unimodal least-squares projection of a vector, or of every mode-0 fibre (column) of an array."""
    y = np.asarray(y, dtype=float)
    if y.ndim == 1:
        return _unimodal_regression(y, non_negativity=non_negativity)[0]
    flat = y.reshape(y.shape[0], -1)
    out = np.stack([_unimodal_regression(flat[:, c], non_negativity=non_negativity)[0] for c in range(flat.shape[1])], axis=1)
    return out.reshape(y.shape)
