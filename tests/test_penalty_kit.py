"""The reusable penalty test kit (matcouply_amd.testing) applied to the built-in penalties - the same contract the
reference runs in tests/test_penalties.py.  CPU only (host-callable plugin methods)."""
import numpy as np
import pytest

from matcouply_amd import penalties as pen
from matcouply_amd.testing import (BaseTestFactorMatricesPenalty, BaseTestFactorMatrixPenalty, BaseTestRowVectorPenalty,
                                   MixinTestHardConstraint)

pytest_plugins = ["matcouply_amd.testing.fixtures"]


class TestNonNegativity(MixinTestHardConstraint, BaseTestRowVectorPenalty):
    PenaltyType = pen.NonNegativity

    def get_invariant_row(self, rng, n_columns):
        return rng.uniform(size=n_columns)

    def get_non_invariant_row(self, rng, n_columns):
        row = rng.uniform(size=n_columns)
        row[rng.randint(n_columns)] = -1.0
        return row


class TestBox(MixinTestHardConstraint, BaseTestRowVectorPenalty):
    PenaltyType = pen.Box
    penalty_default_kwargs = {"min_val": 0, "max_val": 1}

    def get_invariant_row(self, rng, n_columns):
        return rng.uniform(size=n_columns)

    def get_non_invariant_row(self, rng, n_columns):
        row = rng.uniform(size=n_columns)
        row[rng.randint(n_columns)] = 2.0
        return row


class TestL1Penalty(BaseTestRowVectorPenalty):
    PenaltyType = pen.L1Penalty
    penalty_default_kwargs = {"reg_strength": 1}

    def get_invariant_row(self, rng, n_columns):
        return np.zeros(n_columns)

    def get_non_invariant_row(self, rng, n_columns):
        return rng.uniform(1, 2, size=n_columns)

    def test_penalty(self, random_ragged_cmf):
        cmf, shapes, rank = random_ragged_cmf
        weights, (A, B_is, C) = cmf
        p = pen.L1Penalty(0.3)
        assert p.penalty(A) == pytest.approx(0.3 * np.abs(A).sum())
        assert p.penalty(B_is) == pytest.approx(0.3 * sum(np.abs(B).sum() for B in B_is))

    @pytest.mark.parametrize("non_negativity", [True, False])
    def test_small_entries_are_zeroed(self, non_negativity):
        p = pen.L1Penalty(1.0, non_negativity=non_negativity)
        out = p.factor_matrix_update(np.array([[0.05, -0.05, 3.0, -3.0]]), 10, None)
        assert out[0, 0] == 0 and out[0, 1] == 0 and out[0, 2] == pytest.approx(2.9)
        assert out[0, 3] == (0 if non_negativity else pytest.approx(-2.9))


class TestL2Ball(MixinTestHardConstraint, BaseTestFactorMatrixPenalty):
    PenaltyType = pen.L2Ball
    penalty_default_kwargs = {"norm_bound": 1}
    min_rows = 2

    def get_invariant_matrix(self, rng, shape):
        M = rng.uniform(-1, 1, size=shape)
        return M / (np.linalg.norm(M, axis=0, keepdims=True) * 1.5)

    def get_non_invariant_matrix(self, rng, shape):
        M = rng.uniform(-1, 1, size=shape)
        return 3 * M / np.linalg.norm(M, axis=0, keepdims=True)


class TestUnimodality(MixinTestHardConstraint, BaseTestFactorMatrixPenalty):
    PenaltyType = pen.Unimodality
    min_rows = 3

    def get_invariant_matrix(self, rng, shape):
        t = np.arange(shape[0])[:, None]
        peak = rng.randint(shape[0], size=shape[1])[None, :]
        return -np.abs(t - peak).astype(float)

    def get_non_invariant_matrix(self, rng, shape):
        M = np.ones(shape)
        M[1::2] = -1.0  # zig-zag
        return M * np.arange(1, shape[0] + 1)[:, None]


class TestParafac2(BaseTestFactorMatricesPenalty):
    PenaltyType = pen.Parafac2
    min_rows, min_columns, max_columns = 5, 2, 4

    def _auxes(self, rng, shapes):
        rank = shapes[0][1]
        P = [np.linalg.qr(rng.standard_normal((J, rank)))[0] for J, _ in shapes]
        return P, rng.standard_normal((rank, rank))

    def get_invariant_matrices(self, rng, shapes):
        P, D = self._auxes(rng, shapes)
        return [p @ D for p in P]

    def get_non_invariant_matrices(self, rng, shapes):
        return [rng.standard_normal(shape) for shape in shapes]

    def _update_all(self, matrices):
        rank = matrices[0].shape[1]
        rng = np.random.RandomState(0)
        penalty = self._make()
        aux = penalty.init_aux(matrices, rank, 1, rng)
        aux = penalty.factor_matrices_update(matrices, [10] * len(matrices), aux)
        for _ in range(20):  # the coordinate-descent projection is iterated to its fixed point
            aux = penalty.factor_matrices_update(matrices, [10] * len(matrices), aux)
        return penalty.auxes_as_matrices(aux)

    rtol, atol = 1e-5, 1e-8

    def test_penalty(self, random_ragged_cmf):
        cmf, shapes, rank = random_ragged_cmf
        assert pen.Parafac2().penalty(cmf[1][1]) == 0
        with pytest.raises(TypeError):
            pen.Parafac2().penalty(cmf[1][0])

    # PARAFAC2 parametrises its aux variable: the generic identity checks do not apply (reference test_penalties.py)
    def test_subtract_from_aux(self, random_matrices):
        with pytest.raises(TypeError):
            pen.Parafac2().subtract_from_aux(random_matrices[0], random_matrices[0])

    def test_aux_as_matrix(self, random_matrix):
        with pytest.raises(TypeError):
            pen.Parafac2().aux_as_matrix(random_matrix)

    def test_subtract_from_auxes(self, rng, random_matrices):
        shapes = [m.shape for m in random_matrices]
        P, D = self._auxes(rng, shapes)
        mats = [p @ D for p in P]
        for z in pen.Parafac2().subtract_from_auxes((P, D), mats):
            np.testing.assert_allclose(z, 0, atol=1e-12)

    def test_auxes_as_matrices(self, rng, random_matrices):
        shapes = [m.shape for m in random_matrices]
        P, D = self._auxes(rng, shapes)
        for got, p in zip(pen.Parafac2().auxes_as_matrices((P, D)), P):
            np.testing.assert_allclose(got, p @ D)

    def test_init_schemes(self):
        pytest.skip("PARAFAC2 auxiliary variables are (bases, coordinate matrix): covered in tests/test_host_api.py")

    test_given_init = test_rank_and_mode_validation = test_validating_given_init = test_input_validation_for_init = \
        lambda self, *a, **k: pytest.skip("PARAFAC2-specific initialisation is covered in tests/test_host_api.py")


class TestUnitSimplex(MixinTestHardConstraint, BaseTestFactorMatrixPenalty):
    PenaltyType = pen.UnitSimplex
    min_rows = 2

    def get_invariant_matrix(self, rng, shape):
        M = rng.uniform(0.1, 1.0, size=shape)
        return M / M.sum(axis=0, keepdims=True)

    def get_non_invariant_matrix(self, rng, shape):
        return rng.uniform(0.6, 1.0, size=shape) + 1.0  # every column sums to more than one

    def test_result_is_on_the_simplex(self, rng):
        x = self.PenaltyType().factor_matrix_update(rng.standard_normal((9, 4)) * 3, 1.0, None)
        assert x.min() >= 0 and np.allclose(x.sum(axis=0), 1.0)


_CHAIN = 2 * np.eye(6) - np.eye(6, k=1) - np.eye(6, k=-1)
_CHAIN[0, 0] = _CHAIN[-1, -1] = 1


class TestGeneralizedL2Penalty(BaseTestFactorMatrixPenalty):
    PenaltyType = pen.GeneralizedL2Penalty
    penalty_default_kwargs = {"norm_matrix": _CHAIN}
    min_rows = max_rows = 6  # the norm matrix fixes the number of rows

    def get_invariant_matrix(self, rng, shape):
        return np.ones(shape) * rng.uniform(-1, 1, size=(1, shape[1]))  # null space of the chain Laplacian

    def get_non_invariant_matrix(self, rng, shape):
        return rng.standard_normal(shape) + np.arange(shape[0])[:, None]

    def test_penalty(self, rng):
        x = rng.standard_normal((6, 3))
        p = self.PenaltyType(_CHAIN)
        assert p.penalty(x) == pytest.approx(np.sum(np.diff(x, axis=0) ** 2))  # the quadratic form of the chain graph
        assert p.penalty([x, 2 * x]) == pytest.approx(5 * p.penalty(x))


class TestTotalVariationPenalty(BaseTestFactorMatrixPenalty):
    PenaltyType = pen.TotalVariationPenalty
    penalty_default_kwargs = {"reg_strength": 1}
    min_rows = 3

    def get_invariant_matrix(self, rng, shape):
        return np.ones(shape) * rng.uniform(-1, 1, size=(1, shape[1]))  # constant columns: zero total variation

    def get_non_invariant_matrix(self, rng, shape):
        M = np.ones(shape)
        M[1::2] = -1.0
        return M * (1 + np.arange(shape[0]))[:, None]

    def test_penalty(self, rng):
        x = rng.standard_normal((7, 3))
        assert self.PenaltyType(0.5).penalty(x) == pytest.approx(0.5 * np.sum(np.abs(np.diff(x, axis=0))))
        assert self.PenaltyType(0.5, l1_strength=2).penalty(x) == pytest.approx(
            0.5 * np.sum(np.abs(np.diff(x, axis=0))) + 2 * np.sum(np.abs(x)))
        assert self.PenaltyType(1).penalty([x, x]) == pytest.approx(2 * self.PenaltyType(1).penalty(x))
