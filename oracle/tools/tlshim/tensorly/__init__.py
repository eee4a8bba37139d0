"""NumPy stand-in for the `tensorly` names MatCoupLy uses.

TEST/ORACLE INFRASTRUCTURE ONLY.  `tensorly` is an un-vendored, un-pinned third-party dependency of
the reference (`/root/reference/setup.cfg:26-29`) that is absent from this image.  This package is the
build's own code: it lets the *unmodified* reference be imported in the build container so that
`oracle/tools/gen_golden.py` can capture small input/output fixtures under `tests/golden/`.  It is
never imported by `matcouply_amd`, `bench.py` or any `-m gpu` test, and it never travels as part of the
product path.  Equivalence with real TensorLy's NumPy backend is argued, not verified.

Each name maps to the same-named NumPy function; the only non-trivial semantics are:
  * `index_update(t, idx, v)` mutates `t` IN PLACE and returns it
    (`/root/reference/src/matcouply/decomposition.py:191-195,207-211` rely on that);
  * `SVD_FUNS[name](M, n_eigenvecs=None)` = thin LAPACK SVD truncated to `n_eigenvecs`;
  * `check_random_state(None)` = NumPy's global legacy `RandomState`.
"""
import numpy as np

from . import _factorized_tensor, cp_tensor, decomposition, metrics, parafac2_tensor, random, tenalg, testing  # noqa: F401

tensor = np.array
shape = np.shape
dot = np.dot
matmul = np.matmul
transpose = np.transpose
sum = np.sum
zeros = np.zeros
zeros_like = np.zeros_like
ones = np.ones
copy = np.copy
abs = np.abs
trace = np.trace
stack = np.stack
sign = np.sign
diag = np.diag
concatenate = np.concatenate
sqrt = np.sqrt
reshape = np.reshape
min = np.min
max = np.max
all = np.all
solve = np.linalg.solve
float64 = np.float64
float32 = np.float32
int64 = np.int64
int32 = np.int32


def is_tensor(x):
    return isinstance(x, np.ndarray)


def to_numpy(x):
    return np.asarray(x)


def clip(x, a_min=None, a_max=None):
    return np.clip(x, a_min, a_max)


def eye(n, m=None, **context):
    return np.eye(n, m, **context)


def norm(x, order=2, axis=None):
    if order == 2:
        return np.sqrt(np.sum(np.abs(x) ** 2, axis=axis))
    if order == 1:
        return np.sum(np.abs(x), axis=axis)
    if order == "inf":
        return np.max(np.abs(x), axis=axis)
    return np.sum(np.abs(x) ** order, axis=axis) ** (1 / order)


def context(x):
    return {"dtype": x.dtype}


class _Index:
    def __getitem__(self, item):
        return item


index = _Index()


def index_update(t, idx, values):
    t[idx] = values
    return t


def check_random_state(seed):
    if seed is None:
        return np.random.mtrand._rand
    if isinstance(seed, (int, np.integer)):
        return np.random.RandomState(seed)
    if isinstance(seed, np.random.RandomState):
        return seed
    raise ValueError("Seed should be None, int or np.random.RandomState")


def get_backend():
    return "numpy"


def unfold(t, mode):
    return np.reshape(np.moveaxis(t, mode, 0), (t.shape[mode], -1))


def tensor_to_vec(t):
    return np.reshape(t, (-1,))


def _thin_svd(matrix, n_eigenvecs=None, **kwargs):
    U, s, Vh = np.linalg.svd(matrix, full_matrices=False)
    if n_eigenvecs is not None:
        U, s, Vh = U[:, :n_eigenvecs], s[:n_eigenvecs], Vh[:n_eigenvecs]
    return U, s, Vh


SVD_FUNS = {"truncated_svd": _thin_svd, "numpy_svd": _thin_svd, "randomized_svd": _thin_svd}

# `tl.parafac` appears in a docstring of the reference only; keep the attribute for completeness.
parafac = decomposition.parafac
