"""Base test classes for ADMM penalties.  The checks are those of the reference kit
(/root/reference/src/matcouply/testing/admm_penalty.py): initialisation of aux / dual variables for every scheme and
mode incl. validation errors, `subtract_from_aux(es)` / `aux(es)_as_matri(x/ces)` consistency, and for the prox:
an invariant point stays, a non-invariant point moves, the penalty never increases, hard constraints report 0."""
import numpy as np
import pytest

INITS = ["random_uniform", "random_standard_normal", "zeros"]


def assert_allclose(actual, desired, *args, **kwargs):
    np.testing.assert_allclose(np.asarray(actual), np.asarray(desired), *args, **kwargs)


class BaseTestADMMPenalty:
    PenaltyType = None
    penalty_default_kwargs = {}
    min_rows, max_rows = 1, 10
    min_columns, max_columns = 1, 10
    min_matrices, max_matrices = 1, 10
    rtol, atol = 1e-6, 1e-10

    # ---- fixtures of random inputs -------------------------------------------------------------------------
    @pytest.fixture
    def random_row(self, rng):
        return rng.standard_normal(rng.randint(self.min_columns, self.max_columns + 1))

    @pytest.fixture
    def random_matrix(self, rng):
        return rng.standard_normal((rng.randint(self.min_rows, self.max_rows + 1),
                                    rng.randint(self.min_columns, self.max_columns + 1)))

    @pytest.fixture
    def random_matrices(self, rng):
        shape = (rng.randint(self.min_rows, self.max_rows + 1), rng.randint(self.min_columns, self.max_columns + 1))
        return [rng.standard_normal(shape) for _ in range(rng.randint(self.min_matrices, self.max_matrices + 1))]

    def _make(self, **kw):
        return self.PenaltyType(**{**self.penalty_default_kwargs, **kw})

    # ---- initialisation --------------------------------------------------------------------------------------
    def _init(self, penalty, which, matrices, rank, mode, rng):
        return getattr(penalty, f"init_{which}")(matrices, rank, mode=mode, random_state=rng)

    def _check_shapes(self, var, matrices, shapes, rank, mode):
        if mode == 0:
            assert np.shape(var) == (len(shapes), rank)
        elif mode == 2:
            assert np.shape(var) == (shapes[0][1], rank)
        else:
            assert len(var) == len(shapes)
            for v, shape in zip(var, shapes):
                assert np.shape(v) == (shape[0], rank)

    @pytest.mark.parametrize("which", ["aux", "dual"])
    @pytest.mark.parametrize("other_init", INITS)
    @pytest.mark.parametrize("scheme", INITS)
    def test_init_schemes(self, rng, random_ragged_cmf, which, scheme, other_init):
        cmf, shapes, rank = random_ragged_cmf
        matrices = cmf.to_matrices()
        kw = {"aux_init": scheme, "dual_init": other_init} if which == "aux" else {"aux_init": other_init, "dual_init": scheme}
        penalty = self._make(**kw)
        for mode in range(3):
            var = self._init(penalty, which, matrices, rank, mode, rng)
            self._check_shapes(var, matrices, shapes, rank, mode)
            flat = np.concatenate([np.ravel(v) for v in var]) if mode == 1 else np.ravel(var)
            if scheme == "zeros":
                assert np.all(flat == 0)
            elif scheme == "random_uniform":
                assert np.all(flat >= 0) and np.all(flat < 1)
            elif flat.size > 40:
                assert np.any(flat < 0)  # standard normal draws are not all non-negative

    @pytest.mark.parametrize("which", ["aux", "dual"])
    def test_given_init(self, rng, random_ragged_cmf, which):
        cmf, shapes, rank = random_ragged_cmf
        weights, (A, B_is, C) = cmf
        matrices = cmf.to_matrices()
        for mode, given in ((0, A), (1, B_is), (2, C)):
            penalty = self._make(**{f"{which}_init": given})
            assert self._init(penalty, which, matrices, rank, mode, rng) is given

    @pytest.mark.parametrize("which", ["aux", "dual"])
    def test_rank_and_mode_validation(self, rng, random_ragged_cmf, which):
        cmf, shapes, rank = random_ragged_cmf
        matrices = cmf.to_matrices()
        penalty = self._make(aux_init="zeros", dual_init="zeros")
        for bad_rank in (float(rank), [rank], None):
            with pytest.raises(TypeError):
                self._init(penalty, which, matrices, bad_rank, 0, rng)
        for bad_mode, err in ((1.0, TypeError), (None, TypeError), (-1, ValueError), (3, ValueError)):
            with pytest.raises(err):
                self._init(penalty, which, matrices, rank, bad_mode, rng)

    @pytest.mark.parametrize("which", ["aux", "dual"])
    def test_validating_given_init(self, rng, random_ragged_cmf, which):
        cmf, shapes, rank = random_ragged_cmf
        weights, (A, B_is, C) = cmf
        matrices = cmf.to_matrices()
        I, K = len(shapes), shapes[0][1]
        key = f"{which}_init"
        cases = [
            (rng.random_sample((I + 1, rank)), 0, ValueError), (rng.random_sample((I, rank + 1)), 0, ValueError),
            (rng.random_sample((K + 1, rank)), 2, ValueError), (rng.random_sample((K, rank + 1)), 2, ValueError),
            ([rng.random_sample((J_i, rank + 1)) for J_i, _ in shapes], 1, ValueError),
            ([rng.random_sample((J_i + 1, rank)) for J_i, _ in shapes], 1, ValueError),
            (B_is + B_is, 1, ValueError),
            (B_is, 0, TypeError), (B_is, 2, TypeError),  # a list cannot initialise modes 0 / 2
            (A, 1, TypeError),                           # a matrix cannot initialise mode 1
        ]
        for given, mode, err in cases:
            with pytest.raises(err):
                self._init(self._make(**{key: given}), which, matrices, rank, mode, rng)

    @pytest.mark.parametrize("which", ["aux", "dual"])
    def test_input_validation_for_init(self, rng, random_ragged_cmf, which):
        cmf, shapes, rank = random_ragged_cmf
        matrices = cmf.to_matrices()
        key = f"{which}_init"
        for invalid in (None, 1, 1.1):
            for mode in range(3):
                with pytest.raises(TypeError):
                    self._init(self._make(**{key: invalid}), which, matrices, rank, mode, rng)
        for mode in range(3):
            with pytest.raises(ValueError):
                self._init(self._make(**{key: "invalid init name"}), which, matrices, rank, mode, rng)

    # ---- aux <-> matrix helpers -------------------------------------------------------------------------------
    def test_penalty(self, rng):
        raise NotImplementedError

    def test_subtract_from_aux(self, random_matrices):
        penalty = self._make()
        for matrix in random_matrices:
            assert np.all(np.asarray(penalty.subtract_from_aux(matrix, matrix)) == 0)

    def test_subtract_from_auxes(self, random_matrices):
        for zeros in self._make().subtract_from_auxes(random_matrices, random_matrices):
            assert np.all(np.asarray(zeros) == 0)

    def test_aux_as_matrix(self, random_matrix):
        np.testing.assert_array_equal(random_matrix, self._make().aux_as_matrix(random_matrix))

    def test_auxes_as_matrices(self, random_matrices):
        out = self._make().auxes_as_matrices(random_matrices)
        assert len(out) == len(random_matrices)
        for a, b in zip(random_matrices, out):
            np.testing.assert_array_equal(a, b)


class BaseTestFactorMatricesPenalty(BaseTestADMMPenalty):  # e.g. PARAFAC2
    def get_invariant_matrices(self, rng, shapes):
        raise NotImplementedError

    def get_non_invariant_matrices(self, rng, shapes):
        raise NotImplementedError

    def _random_shapes(self, rng):
        n_columns = rng.randint(self.min_columns, self.max_columns + 1)
        n_matrices = rng.randint(self.min_matrices, self.max_matrices + 1)
        return tuple((rng.randint(self.min_rows, self.max_rows + 1), n_columns) for _ in range(n_matrices))

    @pytest.fixture
    def invariant_matrices(self, rng):
        return self.get_invariant_matrices(rng, self._random_shapes(rng))

    @pytest.fixture
    def non_invariant_matrices(self, rng):
        return self.get_non_invariant_matrices(rng, self._random_shapes(rng))

    def _update_all(self, matrices):
        return self._make().factor_matrices_update(matrices, [10] * len(matrices), [None] * len(matrices))

    def test_factor_matrices_update_invariant_point(self, invariant_matrices):
        for before, after in zip(invariant_matrices, self._update_all(invariant_matrices)):
            assert_allclose(before, after, rtol=self.rtol, atol=self.atol)

    def test_factor_matrices_update_changes_input(self, non_invariant_matrices):
        for before, after in zip(non_invariant_matrices, self._update_all(non_invariant_matrices)):
            assert not np.allclose(after, before, rtol=self.rtol, atol=self.atol)

    def test_factor_matrices_update_reduces_penalty(self, random_matrices):
        penalty = self._make()
        assert penalty.penalty(self._update_all(random_matrices)) <= penalty.penalty(random_matrices)


class BaseTestFactorMatrixPenalty(BaseTestFactorMatricesPenalty):
    def get_invariant_matrix(self, rng, shape):
        raise NotImplementedError

    def get_non_invariant_matrix(self, rng, shape):
        raise NotImplementedError

    def get_invariant_matrices(self, rng, shapes):
        return [self.get_invariant_matrix(rng, shape) for shape in shapes]

    def get_non_invariant_matrices(self, rng, shapes):
        return [self.get_non_invariant_matrix(rng, shape) for shape in shapes]

    def _random_shape(self, rng):
        return rng.randint(self.min_rows, self.max_rows + 1), rng.randint(self.min_columns, self.max_columns + 1)

    @pytest.fixture
    def invariant_matrix(self, rng):
        return self.get_invariant_matrix(rng, self._random_shape(rng))

    @pytest.fixture
    def non_invariant_matrix(self, rng):
        return self.get_non_invariant_matrix(rng, self._random_shape(rng))

    def test_factor_matrix_update_invariant_point(self, invariant_matrix):
        assert_allclose(invariant_matrix, self._make().factor_matrix_update(invariant_matrix, 10, None),
                        rtol=self.rtol, atol=self.atol)

    def test_factor_matrix_update_changes_input(self, non_invariant_matrix):
        out = self._make().factor_matrix_update(non_invariant_matrix, 10, None)
        assert not np.allclose(out, non_invariant_matrix, rtol=self.rtol, atol=self.atol)

    def test_factor_matrix_update_reduces_penalty(self, random_matrix):
        penalty = self._make()
        assert penalty.penalty(penalty.factor_matrix_update(random_matrix, 10, None)) <= penalty.penalty(random_matrix)


class BaseTestRowVectorPenalty(BaseTestFactorMatrixPenalty):  # e.g. non-negativity
    def get_invariant_row(self, rng, n_columns):
        raise NotImplementedError

    def get_non_invariant_row(self, rng, n_columns):
        raise NotImplementedError

    def get_invariant_matrix(self, rng, shape):
        return np.stack([self.get_invariant_row(rng, shape[1]) for _ in range(shape[0])], axis=0)

    def get_non_invariant_matrix(self, rng, shape):
        return np.stack([self.get_non_invariant_row(rng, shape[1]) for _ in range(shape[0])], axis=0)

    @pytest.fixture
    def invariant_row(self, rng):
        return self.get_invariant_row(rng, rng.randint(self.min_columns, self.max_columns + 1))

    @pytest.fixture
    def non_invariant_row(self, rng):
        return self.get_non_invariant_row(rng, rng.randint(self.min_columns, self.max_columns + 1))

    def test_row_update_invariant_point(self, invariant_row):
        assert_allclose(invariant_row, self._make().factor_matrix_row_update(invariant_row, 10, None),
                        rtol=self.rtol, atol=self.atol)

    def test_row_update_changes_input(self, non_invariant_row):
        out = self._make().factor_matrix_row_update(non_invariant_row, 10, None)
        assert not np.allclose(out, non_invariant_row, rtol=self.rtol, atol=self.atol)

    def test_row_update_reduces_penalty(self, random_row):
        penalty = self._make()
        assert penalty.penalty(penalty.factor_matrix_row_update(random_row, 10, None)) <= penalty.penalty(random_row)


class MixinTestHardConstraint:
    def test_penalty(self, random_ragged_cmf):
        cmf, shapes, rank = random_ragged_cmf
        weights, (A, B_is, C) = cmf
        penalty = self.PenaltyType(**self.penalty_default_kwargs)
        assert penalty.penalty(A) == 0 and penalty.penalty(B_is) == 0 and penalty.penalty(C) == 0
