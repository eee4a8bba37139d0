"""CPU: the host unimodal-regression module (names of the reference's `_unimodal_regression.py`) against scikit-learn's
isotonic regression, a brute-force unimodal fit, the oracle and the golden vectors of the reference (the strategy of the
reference's tests/test_unimodal_regression.py:21-95)."""
import numpy as np
import pytest
from sklearn.isotonic import IsotonicRegression

from matcouply_amd._unimodal_regression import (_compute_isotonic_from_index, _unimodal_regression, prefix_isotonic_regression,
                                                unimodal_regression)
from oracle import aoadmm_oracle as orc
from tests.helpers import load_npz


@pytest.mark.parametrize("non_negativity", [False, True])
@pytest.mark.parametrize("seed", range(4))
def test_prefix_isotonic_matches_sklearn_on_every_prefix(seed, non_negativity):
    rng = np.random.RandomState(seed)
    y = np.cumsum(rng.standard_normal(40)) * 0.3 + rng.standard_normal(40)
    (level, start), err = prefix_isotonic_regression(y, non_negativity=non_negativity)
    assert err[0] == 0
    for k in range(1, len(y) + 1):
        ref = IsotonicRegression(y_min=0 if non_negativity else None).fit_transform(np.arange(k), y[:k])
        got = _compute_isotonic_from_index(k, level, start)
        np.testing.assert_allclose(got, ref, atol=1e-10)
        np.testing.assert_allclose(err[k], np.sum((ref - y[:k]) ** 2), atol=1e-9)


@pytest.mark.parametrize("non_negativity", [False, True])
def test_unimodal_regression_is_the_best_split_of_two_isotonic_fits(non_negativity):
    rng = np.random.RandomState(3)
    y = np.exp(-0.5 * ((np.arange(30) - 11) / 4.0) ** 2) + 0.2 * rng.standard_normal(30)
    fit, error = _unimodal_regression(y, non_negativity)
    iso = lambda v, inc: IsotonicRegression(y_min=0 if non_negativity else None, increasing=inc).fit_transform(np.arange(len(v)), v) \
        if len(v) else np.zeros(0)
    brute = min(np.sum((np.concatenate([iso(y[:t], True), iso(y[t:], False)]) - y) ** 2) for t in range(len(y) + 1))
    np.testing.assert_allclose(error, brute, atol=1e-9)
    np.testing.assert_allclose(np.sum((fit - y) ** 2), brute, atol=1e-9)
    peak = int(np.argmax(fit))
    assert np.all(np.diff(fit[: peak + 1]) >= -1e-12) and np.all(np.diff(fit[peak:]) <= 1e-12)
    if non_negativity:
        assert fit.min() >= 0


def test_columns_of_arrays_and_the_oracle_agree():
    rng = np.random.RandomState(0)
    Y = rng.standard_normal((25, 3, 2))
    out = unimodal_regression(Y, non_negativity=True)
    assert out.shape == Y.shape
    for a in range(3):
        for b in range(2):
            np.testing.assert_allclose(out[:, a, b], unimodal_regression(Y[:, a, b], non_negativity=True), atol=0)
    flat = Y.reshape(25, -1)
    np.testing.assert_allclose(unimodal_regression(flat, True), orc.unimodal_columns(flat, True), atol=1e-12)


def test_golden_vectors_of_the_reference():
    arrs = load_npz("prox.npz")
    n_uni = int(arrs["n_uni"])
    assert n_uni > 0
    for ui in range(n_uni):
        y = arrs[f"uni{ui}_y"]
        for nn, key in ((False, "out"), (True, "out_nn")):
            np.testing.assert_allclose(unimodal_regression(y, non_negativity=nn), arrs[f"uni{ui}_{key}"], rtol=0, atol=1e-12)
