"""ADMM penalties - the plugin surface of the solver.

Mirrors the class tree and method names of /root/reference/src/matcouply/penalties.py:

    ADMMPenalty (:21-366) -> MatricesPenalty (:369-386) -> MatrixPenalty (:389-423) -> RowVectorPenalty (:426-463)
    HardConstraintMixin (:466-485); NonNegativity (:488-508), Box (:511-542), L1Penalty (:545-592),
    L2Ball (:844-925), Unimodality (:983-1015), Parafac2 (:1018-1324)

`cmf_aoadmm` calls only `init_aux`, `init_dual` and `penalty` on these objects.  For the classes above, the
proximal steps run in native HIP kernels (the object hands the engine a descriptor via `_native_descriptor`);
the `factor_matrix*_update` / `subtract_from_aux*` methods below are the user-facing restatement of the same
operators on NumPy arrays or torch tensors, so that code written against the reference (direct prox calls,
the reusable test kit, custom subclasses) keeps working.  A penalty WITHOUT a native descriptor (a user
subclass) is driven through the engine's step calls with its Python prox evaluated on device tensors; that is also how
`GeneralizedL2Penalty` (:595-747) and `UnitSimplex` (:928-980) run: their prox is a pair of matrix products / a sort +
cumulative sum, written below for NumPy arrays and torch (device) tensors alike.  `TotalVariationPenalty` (:750-841)
needs the GPL `condat_tv` package in the reference; here its prox is an own restatement of L. Condat's published direct
1-D TV algorithm (native kernel k_slab_tv; host version below), verified against the optimality conditions of the
problem since that package is not available to generate reference vectors.
"""
from abc import abstractmethod

import numpy as np

from . import _engine
from ._doc_utils import InheritableDocstrings
from ._utils import check_random_state, get_svd, is_tensor, is_torch, shape, to_numpy, torch


# ---- small backend helpers (NumPy arrays or torch tensors) --------------------------------------------------
def _clip(x, lo=None, hi=None):
    if is_torch(x):
        return torch.clamp(x, min=lo, max=hi)
    return np.clip(x, lo, hi)


def _abs(x):
    return x.abs() if is_torch(x) else np.abs(x)


def _sign(x):
    return torch.sign(x) if is_torch(x) else np.sign(x)


def _sum(x, axis=None):
    if is_torch(x):
        return x.sum() if axis is None else x.sum(dim=axis)
    return np.sum(x, axis=axis)


def _sqrt(x):
    return torch.sqrt(x) if is_torch(x) else np.sqrt(x)


def _eye(n, m, like=None):
    if like is not None and is_torch(like):
        return torch.eye(n, m, dtype=like.dtype, device=like.device)
    return np.eye(n, m)


class ADMMPenalty(metaclass=InheritableDocstrings):
    """Base class for all regularizers and constraints (penalties.py:21-366).

    Parameters
    ----------
    aux_init, dual_init : {"random_uniform", "random_standard_normal", "zeros", 2-D array, list of 2-D arrays}
    """

    def __init__(self, aux_init="random_uniform", dual_init="random_uniform"):
        self.aux_init = aux_init
        self.dual_init = dual_init

    # -- initialisation (penalties.py:36-261) ---------------------------------------------------------------
    def _init_var(self, init, matrices, rank, mode, random_state, what):
        random_state = check_random_state(random_state)
        if not isinstance(rank, int):
            raise TypeError("Rank must be int, not {}".format(type(rank)))
        if not isinstance(mode, int):
            raise TypeError("Mode must be int, not {}".format(type(mode)))
        elif mode not in [0, 1, 2]:
            raise ValueError("Mode must be 0, 1, or 2.")
        if not isinstance(init, str) and not is_tensor(init) and not isinstance(init, list):
            raise TypeError(
                "self.{}_init must be a tensor, a list of tensors or a string specifiying init method, not {}".format(
                    what, type(init)))
        if mode in {0, 2} and is_tensor(init):
            length_, rank_ = shape(init)
            I, K = len(matrices), shape(matrices[0])[1]
            if rank != rank_ or (mode == 0 and length_ != I):
                raise ValueError("Invalid shape for pre-specified auxiliary variable for mode 0"
                                 "\nShould have shape {}, but has shape {}".format((I, rank), (length_, rank_)))
            elif rank != rank_ or (mode == 2 and length_ != K):
                raise ValueError("Invalid shape for pre-specified auxiliary variable for mode 2"
                                 "\nShould have shape {}, but has shape {}".format((K, rank), (length_, rank_)))
            return init
        elif mode in {0, 2} and isinstance(init, list):
            raise TypeError("Cannot use list of matrices to initialize auxiliary matrices for mode 0 or 2.")
        elif mode == 1 and isinstance(init, list):
            shapes = ((shape(matrix)[0], rank) for matrix in matrices)
            if any(shape(v) != shp for v, shp in zip(init, shapes)):
                raise ValueError("Invalid shape for at least one of matrices in the auxiliary variable list for mode 1.")
            elif len(init) != len(matrices):
                raise ValueError("Different number of pre-specified auxiliary factor matrices for mode 1 "
                                 "than the number of coupled matrices.")
            return init
        elif mode == 1 and is_tensor(init):
            raise TypeError(
                "Cannot use a tensor (matrix) to initialize auxiliary matrices for mode 1. Must be a list instead.")

        if init == "random_uniform":
            draw = lambda size: random_state.uniform(size=size)
        elif init == "random_standard_normal":
            draw = lambda size: random_state.standard_normal(size=size)
        elif init == "zeros":
            draw = lambda size: np.zeros(size)
        else:
            raise ValueError("Unknown aux init: {}".format(init))
        if mode == 0:
            return draw((len(matrices), rank))
        elif mode == 1:
            return [draw((shape(matrix)[0], rank)) for matrix in matrices]
        return draw((shape(matrices[0])[1], rank))

    def init_aux(self, matrices, rank, mode, random_state=None):
        """Initialize the auxiliary variables (penalties.py:36-147)."""
        return self._init_var(self.aux_init, matrices, rank, mode, random_state, "aux")

    def init_dual(self, matrices, rank, mode, random_state=None):
        """Initialize the dual variables (penalties.py:149-261)."""
        return self._init_var(self.dual_init, matrices, rank, mode, random_state, "dual")

    @abstractmethod
    def penalty(self, x):  # pragma: nocover
        """Value of the penalty at `x` (a factor matrix, or the list of matrices of a multi-matrix mode); it is added to
        the regularised loss the stopping rule looks at.  Hard constraints return 0."""
        raise NotImplementedError

    # -- aux <-> matrix helpers (penalties.py:268-343) --------------------------------------------------------
    def subtract_from_auxes(self, auxes, duals):
        """(aux - dual) for each auxiliary- and dual-factor matrix for mode=1."""
        return [self.subtract_from_aux(aux, dual) for aux, dual in zip(auxes, duals)]

    def subtract_from_aux(self, aux, dual):
        """(aux - dual) for mode=0 and mode=2."""
        return aux - dual

    def aux_as_matrix(self, aux):
        return aux

    def auxes_as_matrices(self, auxes):
        return [self.aux_as_matrix(aux) for aux in auxes]

    # -- repr (penalties.py:345-366) ------------------------------------------------------------------------------
    def _auto_add_param_to_repr(self, param):
        if param.startswith("_"):
            return False
        elif param in {"aux_init", "dual_init"}:
            return False
        return True

    def __repr__(self):
        param_strings = [f"{key}={repr(value)}" for key, value in self.__dict__.items()
                         if self._auto_add_param_to_repr(key)]
        param_strings.append(f"aux_init='{self.aux_init}'" if isinstance(self.aux_init, str) else "aux_init=given_init")
        param_strings.append(
            f"dual_init='{self.dual_init}'" if isinstance(self.dual_init, str) else "dual_init=given_init")
        params = ", ".join(param_strings)
        return f"<'{self.__module__}.{type(self).__name__}' with {params})>"

    # -- bridge to the HIP engine ---------------------------------------------------------------------------------
    def _native_descriptor(self):
        """(kind, non_negativity, p0, p1) for penalties with a native kernel, else None (host-evaluated prox)."""
        return None


# methods whose behaviour the native kernels restate: a subclass that overrides any of them must be evaluated on the host
_NATIVE_CONTRACT = ("factor_matrices_update", "factor_matrix_update", "factor_matrix_row_update", "penalty",
                    "subtract_from_auxes", "subtract_from_aux", "aux_as_matrix", "auxes_as_matrices")


def native_descriptor_of(reg):
    """The native-kernel descriptor of a penalty object, or None when its proximal operator must be evaluated on the
    host: a user subclass of a built-in penalty that overrides the prox, the penalty value or the aux bookkeeping is NOT
    routed to the built-in kernel (the reference would call the override, decomposition.py:278-285)."""
    cls = type(reg)
    owner = next((k for k in cls.__mro__ if "_native_descriptor" in vars(k)), None)
    if owner is None:
        return None
    for name in _NATIVE_CONTRACT:
        if getattr(cls, name, None) is not getattr(owner, name, None):
            return None
    return reg._native_descriptor()


class MatricesPenalty(ADMMPenalty):
    """Penalties applied to a list of factor matrices simultaneously (penalties.py:369-386)."""

    @abstractmethod
    def factor_matrices_update(self, factor_matrices, feasibility_penalties, auxes):  # pragma: nocover
        """Proximal step for all matrices of the mode at once: returns the new auxiliary variables for the points
        `factor_matrices` (already shifted by the scaled duals), one feasibility penalty rho_i per matrix; `auxes` holds
        the previous auxiliary variables for operators that warm-start from them."""
        raise NotImplementedError


class MatrixPenalty(MatricesPenalty):
    """Penalties that can be applied to a single factor matrix at a time (penalties.py:389-423)."""

    def factor_matrices_update(self, factor_matrices, feasibility_penalties, auxes):
        return [self.factor_matrix_update(fm, feasibility_penalty, aux)
                for fm, feasibility_penalty, aux in zip(factor_matrices, feasibility_penalties, auxes)]

    @abstractmethod
    def factor_matrix_update(self, factor_matrix, feasibility_penalty, aux):  # pragma: nocover
        """Proximal step for one factor matrix: returns the new auxiliary matrix for the point `factor_matrix` (already
        shifted by the scaled dual) under the feasibility penalty rho; `aux` is the previous auxiliary matrix."""
        raise NotImplementedError


class RowVectorPenalty(MatrixPenalty):
    """Penalties that can be applied to one row of a factor matrix at a time (penalties.py:426-463)."""

    def factor_matrix_update(self, factor_matrix, feasibility_penalty, aux):
        out = factor_matrix.clone() if is_torch(factor_matrix) else np.zeros(shape(factor_matrix))
        for row, factor_matrix_row in enumerate(factor_matrix):
            out[row] = self.factor_matrix_row_update(factor_matrix_row, feasibility_penalty, aux[row])
        return out

    @abstractmethod
    def factor_matrix_row_update(self, factor_matrix_row, feasibility_penalty, aux_row):  # pragma: nocover
        """Proximal step for a single row of a factor matrix (the A-mode updates every row a_i with its own rho_i)."""
        raise NotImplementedError


class HardConstraintMixin:
    """Hard constraints report a penalty of 0 (penalties.py:466-485); inspect the feasibility gaps instead."""

    def penalty(self, x):
        return 0


class NonNegativity(HardConstraintMixin, RowVectorPenalty):
    r"""Impose non-negative values for the factor: max(x, 0) (penalties.py:488-508)."""

    def factor_matrix_row_update(self, factor_matrix_row, feasibility_penalty, aux_row):
        return _clip(factor_matrix_row, 0, float("inf"))

    def factor_matrix_update(self, factor_matrix, feasibility_penalty, aux):
        return _clip(factor_matrix, 0, float("inf"))

    def _native_descriptor(self):
        return (_engine.PEN_NN, False, 0.0, 0.0)


class Box(HardConstraintMixin, RowVectorPenalty):
    r"""Set minimum and maximum value for the factor: clip(x, min_val, max_val) (penalties.py:511-542)."""

    def __init__(self, min_val, max_val, aux_init="random_uniform", dual_init="random_uniform"):
        super().__init__(aux_init=aux_init, dual_init=dual_init)
        self.min_val = min_val
        self.max_val = max_val

    def factor_matrix_row_update(self, factor_matrix_row, feasibility_penalty, aux_row):
        return _clip(factor_matrix_row, self.min_val, self.max_val)

    def factor_matrix_update(self, factor_matrix, feasibility_penalty, aux):
        return _clip(factor_matrix, self.min_val, self.max_val)

    def _native_descriptor(self):
        lo = -float("inf") if self.min_val is None else float(self.min_val)
        hi = float("inf") if self.max_val is None else float(self.max_val)
        return (_engine.PEN_BOX, False, lo, hi)


class L1Penalty(RowVectorPenalty):
    r"""L1 (LASSO) regularization: soft thresholding by reg_strength / rho (penalties.py:545-592)."""

    def __init__(self, reg_strength, non_negativity=False, aux_init="random_uniform", dual_init="random_uniform"):
        super().__init__(aux_init=aux_init, dual_init=dual_init)
        if reg_strength < 0:
            raise ValueError("Regularization strength must be nonnegative.")
        self.reg_strength = reg_strength
        self.non_negativity = non_negativity

    def factor_matrix_row_update(self, factor_matrix_row, feasibility_penalty, aux_row):
        if self.non_negativity:
            return _clip(factor_matrix_row - self.reg_strength / feasibility_penalty, 0, float("inf"))
        sign = _sign(factor_matrix_row)
        return sign * _clip(_abs(factor_matrix_row) - self.reg_strength / feasibility_penalty, 0, float("inf"))

    def factor_matrix_update(self, factor_matrix, feasibility_penalty, aux):
        return self.factor_matrix_row_update(factor_matrix, feasibility_penalty, aux)

    def penalty(self, x):
        if is_tensor(x):
            return _sum(_abs(x)) * self.reg_strength
        return sum(_sum(_abs(xi)) for xi in x) * self.reg_strength

    def _native_descriptor(self):
        return (_engine.PEN_L1, bool(self.non_negativity), float(self.reg_strength), 0.0)


class L2Ball(HardConstraintMixin, MatrixPenalty):
    r"""Column L2 norms at most `norm_bound`, optionally with non-negativity (penalties.py:844-925)."""

    def __init__(self, norm_bound, non_negativity=False, aux_init="random_uniform", dual_init="random_uniform"):
        super().__init__(aux_init, dual_init)
        self.norm_bound = norm_bound
        self.non_negativity = non_negativity
        if norm_bound <= 0:
            raise ValueError("The norm bound must be positive.")

    def factor_matrix_update(self, factor_matrix, feasibility_penalty, aux):
        if self.non_negativity:
            factor_matrix = _clip(factor_matrix, 0, float("inf"))
        column_norms = _sqrt(_sum(factor_matrix ** 2, axis=0))
        column_norms = _clip(column_norms, self.norm_bound, float("inf"))
        return factor_matrix * self.norm_bound / column_norms

    def _native_descriptor(self):
        return (_engine.PEN_L2BALL, bool(self.non_negativity), float(self.norm_bound), 0.0)


def tv_denoise(x, lam):
    """argmin_y 1/2 ||y - x||^2 + lam sum_n |y_n - y_{n-1}| for a vector (L. Condat, "A Direct Algorithm for 1-D Total
    Variation Denoising", IEEE Signal Processing Letters 20(11), 2013), host NumPy version of the kernel k_slab_tv."""
    x = np.asarray(x, dtype=float)
    n = len(x)
    y = np.empty(n)
    if n == 0:
        return y
    if lam <= 0:
        return x.copy()
    k = k0 = km = kp = 0
    vmin, vmax, umin, umax = x[0] - lam, x[0] + lam, lam, -lam
    while True:
        if k == n - 1:
            if umin < 0.0:
                y[k0:km + 1] = vmin
                k0 = k = km = km + 1
                vmin, umin = x[k], lam
                umax = vmin + lam - vmax
            elif umax > 0.0:
                y[k0:kp + 1] = vmax
                k0 = k = kp = kp + 1
                vmax, umax = x[k], -lam
                umin = vmax - lam - vmin
            else:
                y[k0:k + 1] = vmin + umin / (k - k0 + 1)
                return y
        else:
            umin += x[k + 1] - vmin
            umax += x[k + 1] - vmax
            if umin < -lam:
                y[k0:km + 1] = vmin
                k0 = k = km = kp = km + 1
                vmin = x[k]
                vmax, umin, umax = vmin + 2.0 * lam, lam, -lam
            elif umax > lam:
                y[k0:kp + 1] = vmax
                k0 = k = km = kp = kp + 1
                vmax = x[k]
                vmin, umin, umax = vmax - 2.0 * lam, lam, -lam
            else:
                k += 1
                if umin >= lam:
                    km = k
                    vmin += (umin - lam) / (km - k0 + 1)
                    umin = lam
                if umax <= -lam:
                    kp = k
                    vmax += (umax + lam) / (kp - k0 + 1)
                    umax = -lam


class TotalVariationPenalty(MatrixPenalty):
    r"""Piecewise constant components: alpha * sum_n |x_n - x_{n-1}| (+ beta * sum_n |x_n|) on every column
    (penalties.py:750-841).  prox: TV denoising with 2 alpha / rho, then the L1 soft threshold with beta / rho."""

    def __init__(self, reg_strength, l1_strength=0, aux_init="random_uniform", dual_init="random_uniform"):
        if reg_strength <= 0:
            raise ValueError("The TV regularization strength must be positive.")
        if l1_strength < 0:
            raise ValueError("The L1 regularization strength must be non-negative.")
        super().__init__(aux_init, dual_init)
        self.reg_strength = reg_strength
        self.l1_strength = l1_strength

    def factor_matrix_update(self, factor_matrix, feasibility_penalty, aux):
        fm = factor_matrix.detach().cpu().numpy() if is_torch(factor_matrix) else np.asarray(factor_matrix)
        lam = self.reg_strength * 2 / feasibility_penalty
        X = np.stack([tv_denoise(fm[:, c], lam) for c in range(fm.shape[1])], axis=1)
        if self.l1_strength:
            X = np.sign(X) * np.clip(np.abs(X) - self.l1_strength / feasibility_penalty, 0, float("inf"))
        if is_torch(factor_matrix):
            return torch.as_tensor(X, dtype=factor_matrix.dtype, device=factor_matrix.device)
        return X

    def _penalty(self, x):
        if is_torch(x):
            value = self.reg_strength * torch.diff(x, dim=0).abs().sum()
        else:
            value = self.reg_strength * np.sum(np.abs(np.diff(x, axis=0)))
        if self.l1_strength:
            value = value + self.l1_strength * _sum(_abs(x))
        return value

    def penalty(self, x):
        if is_tensor(x):
            return self._penalty(x)
        return sum(self._penalty(xi) for xi in x)

    def _native_descriptor(self):
        return (_engine.PEN_TV, False, float(self.reg_strength), float(self.l1_strength))


class GeneralizedL2Penalty(MatrixPenalty):
    r"""Penalty x^T M x on every column, M symmetric positive semidefinite (penalties.py:595-747), e.g. a graph Laplacian.

    prox: (M + rho/2 I)^-1 (rho/2) x = U (S + rho/2 I)^-1 U^T (rho/2) x with M = U S U^T computed once.  No native kernel:
    the solver evaluates it on device tensors as two matrix products per call."""

    def __init__(self, norm_matrix, svd="truncated_svd", aux_init="random_uniform", dual_init="random_uniform",
                 validate=True):
        super().__init__(aux_init, dual_init)
        self.norm_matrix = norm_matrix
        self.svd = svd
        self.validate = validate
        M = norm_matrix.detach().cpu().numpy() if is_torch(norm_matrix) else np.asarray(norm_matrix)
        if validate and not np.all(M.T == M):
            raise ValueError("The norm matrix should be symmetric positive semidefinite")
        if validate and np.any(np.linalg.eigvals(M) < -1e-14):
            raise ValueError("The norm matrix should be symmetric positive semidefinite")
        self._U, self._s, _ = get_svd(svd)(np.asarray(M, dtype=np.float64))  # Vh ignored: the norm matrix is symmetric
        self._device_cache = {}

    @property
    def svd_fun(self):
        return get_svd(self.svd)

    def _factors_like(self, x):
        """(U, s, M) as arrays of the same kind / dtype / device as x"""
        if not is_torch(x):
            return self._U, self._s, np.asarray(self.norm_matrix.detach().cpu().numpy() if is_torch(self.norm_matrix)
                                                else self.norm_matrix)
        key = (x.device, x.dtype)
        if key not in self._device_cache:
            M = self.norm_matrix.detach().cpu().numpy() if is_torch(self.norm_matrix) else np.asarray(self.norm_matrix)
            self._device_cache[key] = tuple(torch.as_tensor(np.ascontiguousarray(a), dtype=x.dtype, device=x.device)
                                            for a in (self._U, self._s, M))
        return self._device_cache[key]

    def factor_matrix_update(self, factor_matrix, feasibility_penalty, aux):
        U, s, _ = self._factors_like(factor_matrix)
        s_aug = s + 0.5 * feasibility_penalty
        tmp = U.T @ (0.5 * feasibility_penalty * factor_matrix)
        return (U * (1 / s_aug)) @ tmp

    def _penalty(self, x):
        _, _, M = self._factors_like(x)
        return (x.T @ M @ x).trace() if is_torch(x) else np.trace(x.T @ M @ x)

    def penalty(self, x):
        if is_tensor(x):
            return self._penalty(x)
        return sum(self._penalty(xi) for xi in x)

    def _native_descriptor(self):
        return (_engine.PEN_GL2, False, 0.0, 0.0)

    def _native_matrix(self):
        """what the native kernel needs of the norm matrix: fp64 [U | s | U^T] (n * n + n + n * n) and n"""
        U = np.ascontiguousarray(np.asarray(self._U, dtype=np.float64))
        s = np.asarray(self._s, dtype=np.float64).ravel()
        return np.concatenate([U.ravel(), s, np.ascontiguousarray(U.T).ravel()]), U.shape[0]


class UnitSimplex(HardConstraintMixin, MatrixPenalty):
    """Component vectors non-negative and summing to one (penalties.py:928-980).

    The reference finds the Lagrange multiplier mu of sum(max(y - mu, 0)) = 1 per column by bisection; the sorted
    cumulative-sum formula used here is the exact root of the same equation and runs on the device for torch tensors."""

    def _compute_lagrange_multiplier(self, factor_matrix_column):
        """Multiplier mu of the constraint sum(max(y - mu, 0)) = 1 for one column (penalties.py:942-966), closed form"""
        y = to_numpy(factor_matrix_column).astype(float).ravel()
        u = -np.sort(-y)
        css = np.cumsum(u) - 1.0
        k = np.nonzero(u - css / np.arange(1, len(y) + 1) > 0)[0][-1]
        return float(css[k] / (k + 1.0))

    def factor_matrix_update(self, factor_matrix, feasibility_penalty, aux):
        y = factor_matrix
        n = y.shape[0]
        if is_torch(y):
            u, _ = torch.sort(y, dim=0, descending=True)
            css = torch.cumsum(u, dim=0) - 1.0
            idx = torch.arange(1, n + 1, dtype=y.dtype, device=y.device).unsqueeze(1)
            k = (u - css / idx > 0).to(torch.int64).cumsum(dim=0).argmax(dim=0)  # last index where the condition holds
            mu = css.gather(0, k.unsqueeze(0)).squeeze(0) / (k.to(y.dtype) + 1.0)
            return torch.clamp(y - mu.unsqueeze(0), min=0)
        y = np.asarray(y)
        u = -np.sort(-y, axis=0)
        css = np.cumsum(u, axis=0) - 1.0
        cond = u - css / np.arange(1, n + 1)[:, None] > 0
        k = n - 1 - np.argmax(cond[::-1], axis=0)
        mu = css[k, np.arange(y.shape[1])] / (k + 1.0)
        return np.clip(y - mu[None, :], 0, None)

    def _native_descriptor(self):
        return (_engine.PEN_SIMPLEX, False, 0.0, 0.0)


def _prefix_isotonic(y, non_negativity):
    """Prefix isotonic regression (Stout 2008): per prefix the last block (level, start) and the prefix SSE."""
    n = len(y)
    level, start, err = np.zeros(n), np.zeros(n, dtype=np.int64), np.zeros(n + 1)
    sy, sy2, sw = y.astype(float).copy(), (y * y).astype(float), np.ones(n)
    cum2 = np.cumsum(y * y)
    for i in range(n):
        level[i], start[i] = y[i], i
        while start[i] != 0 and level[i] <= level[start[i] - 1]:
            p = start[i] - 1
            sy[i] += sy[p]
            sy2[i] += sy2[p]
            sw[i] += sw[p]
            level[i] = sy[i] / sw[i]
            start[i] = start[p]
        if non_negativity and level[i] < 0:
            err[i + 1] = cum2[i]
        else:
            err[i + 1] = (sy2[i] - sy[i] ** 2 / sw[i]) + err[start[i]]
    if non_negativity:
        level[level < 0] = 0.0
    return level, start, err


def _fit_of_prefix(length, level, start):
    out = np.empty(length)
    idx = length - 1
    while idx >= 0:
        out[start[idx]: idx + 1] = level[idx]
        idx = start[idx] - 1
    return out


def unimodal_regression(y, non_negativity=False):
    """Unimodal least-squares projection of a vector / of the columns of a matrix, host NumPy version
    (reference: _unimodal_regression.py:107-141).  The solver itself uses the HIP kernel k_slab_unimodal."""
    y = np.asarray(y, dtype=float)
    if y.ndim > 1:
        flat = y.reshape(y.shape[0], -1)
        out = np.stack([unimodal_regression(flat[:, c], non_negativity) for c in range(flat.shape[1])], axis=1)
        return out.reshape(y.shape)
    n = len(y)
    lvl_l, st_l, err_l = _prefix_isotonic(y, non_negativity)
    lvl_r, st_r, err_r = _prefix_isotonic(y[::-1], non_negativity)
    best, split = err_r[n], 0
    for i in range(n + 1):
        e = err_l[i] + err_r[n - i]
        if e < best:
            best, split = e, i
    return np.concatenate([_fit_of_prefix(split, lvl_l, st_l), _fit_of_prefix(n - split, lvl_r, st_r)[::-1]])


class Unimodality(HardConstraintMixin, MatrixPenalty):
    r"""Unimodal (optionally non-negative) component vectors (penalties.py:983-1015)."""

    def __init__(self, non_negativity=False, aux_init="random_uniform", dual_init="random_uniform"):
        super().__init__(aux_init, dual_init)
        self.non_negativity = non_negativity

    def factor_matrix_update(self, factor_matrix, feasibility_penalty, aux):
        if is_torch(factor_matrix):
            out = unimodal_regression(factor_matrix.detach().cpu().numpy(), non_negativity=self.non_negativity)
            return torch.as_tensor(out, dtype=factor_matrix.dtype, device=factor_matrix.device)
        return unimodal_regression(factor_matrix, non_negativity=self.non_negativity)

    def _native_descriptor(self):
        return (_engine.PEN_UNIMODAL, bool(self.non_negativity), 0.0, 0.0)


class Parafac2(MatricesPenalty):
    r"""PARAFAC2 constraint B_i^T B_i = const on the uncoupled factor matrices (penalties.py:1018-1324).

    The auxiliary variable is the tuple ``(list of orthogonal basis matrices P_i, coordinate matrix Delta)`` with
    B_i = P_i Delta; one sweep of the coordinate-descent projection per ADMM iteration by default."""

    def __init__(self, svd="truncated_svd", n_iter=1, update_basis_matrices=True, update_coordinate_matrix=True,
                 aux_init="random_uniform", dual_init="random_uniform"):
        self.svd = svd
        self.aux_init = aux_init
        self.dual_init = dual_init
        self.update_basis_matrices = update_basis_matrices
        self.update_coordinate_matrix = update_coordinate_matrix
        self.n_iter = n_iter

    @property
    def svd_fun(self):
        return get_svd(self.svd)

    def init_aux(self, matrices, rank, mode, random_state=None):
        """(basis matrices = first `rank` columns of the identity, coordinate matrix per `aux_init`)
        (penalties.py:1111-1222)."""
        if not isinstance(self.aux_init, (str, tuple)):
            raise TypeError("Parafac2 auxiliary variables must be initialized using either a string"
                            " or a tuple (containing the orthogonal basis matrices and the coordinate matrix).")
        if not isinstance(rank, int):
            raise TypeError("Rank must be int, not {}".format(type(rank)))
        if not isinstance(mode, int):
            raise TypeError("Mode must be int, not {}".format(type(mode)))
        if mode != 1:
            raise ValueError("PARAFAC2 constraint can only be imposed with mode=1")
        if isinstance(self.aux_init, str):
            if self.aux_init == "random_uniform":
                coordinate_matrix = random_state.uniform(size=(rank, rank))
            elif self.aux_init == "random_standard_normal":
                coordinate_matrix = random_state.standard_normal(size=(rank, rank))
            elif self.aux_init == "zeros":
                coordinate_matrix = np.zeros((rank, rank))
            else:
                raise ValueError(f"Unknown aux init: {self.aux_init}")
            basis_matrices = [np.eye(shape(M)[0], rank) for M in matrices]
            return basis_matrices, coordinate_matrix
        basis_matrices, coordinate_matrix = self.aux_init
        if not isinstance(basis_matrices, list) or not is_tensor(coordinate_matrix):
            raise TypeError("If self.aux_init is a tuple, then its first element must be a list of basis matrices "
                            "and second element the coordinate matrix.")
        if not len(shape(coordinate_matrix)) == 2:
            raise ValueError("The coordinate matrix must have two modes, not {}".format(len(shape(coordinate_matrix))))
        if shape(coordinate_matrix)[0] != shape(coordinate_matrix)[1] or shape(coordinate_matrix)[0] != rank:
            raise ValueError("The coordinate matrix must be rank x rank, with rank={}, not {}".format(
                rank, shape(coordinate_matrix)))
        for matrix, basis_matrix in zip(matrices, basis_matrices):
            if not is_tensor(basis_matrix):
                raise TypeError("Each basis matrix must be a tensorly tensor")
            if not len(shape(basis_matrix)) == 2:
                raise ValueError("Each basis matrix must be tensor with two modes, not {}".format(
                    len(shape(basis_matrix))))
            if shape(matrix)[0] != shape(basis_matrix)[0] or shape(basis_matrix)[1] != rank:
                raise ValueError("The i-th basis matrix must have shape J_i x rank, where J_i is the number of "
                                 "rows in the i-th matrix.")
            cross_product = basis_matrix.T @ basis_matrix
            if not _sum((cross_product - _eye(rank, rank, like=basis_matrix)) ** 2) < 1e-8:
                raise ValueError("The basis matrices must be orthogonal")
        if len(basis_matrices) != len(matrices):
            raise ValueError("There must be as many basis matrices as there are matrices")
        return self.aux_init

    def factor_matrices_update(self, factor_matrices, feasibility_penalties, auxes):
        basis_matrices, coordinate_matrix = auxes
        R = shape(coordinate_matrix)[0]
        for _ in range(self.n_iter):
            if self.update_basis_matrices:
                basis_matrices = []
                for fm in factor_matrices:
                    if is_torch(fm):
                        U, s, Vh = torch.linalg.svd(fm @ coordinate_matrix.T, full_matrices=False)
                    else:
                        U, s, Vh = self.svd_fun(fm @ coordinate_matrix.T, n_eigenvecs=R)
                    basis_matrices.append(U[:, :R] @ Vh[:R])
            if self.update_coordinate_matrix:
                coordinate_matrix = 0
                for fm, basis_matrix, feasibility_penalty in zip(factor_matrices, basis_matrices, feasibility_penalties):
                    coordinate_matrix = coordinate_matrix + feasibility_penalty * basis_matrix.T @ fm
                coordinate_matrix = coordinate_matrix / sum(feasibility_penalties)
            if (not self.update_coordinate_matrix) or (not self.update_basis_matrices):
                break
        return basis_matrices, coordinate_matrix

    def subtract_from_aux(self, aux, dual):
        raise TypeError("The PARAFAC2 constraint cannot shift a single factor matrix.")

    def subtract_from_auxes(self, auxes, duals):
        P_is, coord_mat = auxes
        return [P_i @ coord_mat - dual for P_i, dual in zip(P_is, duals)]

    def aux_as_matrix(self, aux):
        raise TypeError("The PARAFAC2 constraint cannot convert a single aux to a matrix")

    def auxes_as_matrices(self, auxes):
        P_is, coord_mat = auxes
        return [P_i @ coord_mat for P_i in P_is]

    def penalty(self, x):
        if not isinstance(x, list):
            raise TypeError("Cannot compute PARAFAC2 penalty of other types than a list of tensors")
        return 0

    def _native_descriptor(self):
        if self.n_iter == 1 and self.update_basis_matrices and self.update_coordinate_matrix:
            return (_engine.PEN_PARAFAC2, False, 0.0, 0.0)
        return None  # frozen-basis / frozen-coordinate / multi-sweep variants are evaluated on the host
