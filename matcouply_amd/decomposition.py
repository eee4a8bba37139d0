"""`cmf_aoadmm` / `parafac2_aoadmm` on the MI355X engine.

Same signature, defaults, return types, error behaviour and stopping rules as
/root/reference/src/matcouply/decomposition.py (`cmf_aoadmm` :662-1100, `parafac2_aoadmm` :1103-1179,
`compute_feasibility_gaps` :351-417, `ADMMVars` :643-645, `DiagnosticMetrics` :648-659).  The host side only
parses arguments, draws the initial state with the reference's RNG order (:35-37, :904-905), keeps the loss /
stopping bookkeeping (:990-1053) and converts results; every per-mode ADMM update runs in the HIP library
(matcouply_amd/_engine.py -> libmatcouply_hip.so).  There is no CPU fallback.

Extensions (keyword-only, not in the reference): `group=` a torch.distributed process group - every rank passes
ITS OWN contiguous range of the I matrices; the replicated C-mode normal equations and the diagnostic sums are
all-reduced over the group (RCCL on MI355X, gloo in the CPU tests).
"""
from copy import copy
import os
from typing import NamedTuple, Optional

import numpy as np

from . import _engine, penalties
from ._utils import check_random_state, get_svd, is_iterable, is_tensor, is_torch, shape, to_numpy, torch
from .coupled_matrices import CoupledMatrixFactorization

__all__ = ["compute_feasibility_gaps", "ADMMVars", "DiagnosticMetrics", "cmf_aoadmm", "parafac2_aoadmm",
           "PackedMatrices", "partition_slabs"]

# TEST-ONLY seam.  The CPU test-suite (tests/oracle_engine.py) substitutes a checker engine here to exercise the host and
# multi-process logic without a GPU.  The substitution is honoured only under MATCOUPLY_AMD_TEST_ENGINE=1, which
# tests/conftest.py sets and the product never does: matcouply_amd has no CPU path of its own.
_ENGINE_FACTORY = None


# arithmetic="auto" (see cmf_aoadmm): problems of at most 2^20 elements always take the exact arithmetic (the library's own rule);
# up to _AUTO_EXACT_MAX_ELEMENTS a penalty-free mode whose normal equations have kappa = ||M||_F ||M^-1||_F above
# _AUTO_EXACT_KAPPA moves the run there; larger problems are only warned, above _AUTO_EXACT_WARN_KAPPA
_AUTO_EXACT_MAX_ELEMENTS = float(1 << 24)
_AUTO_EXACT_KAPPA = 1e3
_AUTO_EXACT_POLAR = 1e5  # ... or a PARAFAC2 polar factor (||sigma|| / sigma_min of Y_i Delta^T) above this
_AUTO_EXACT_WARN_KAPPA = 1e6
_AUTO_EXACT_PROBE_EVERY = 64
_AUTO_EXACT_TRIAL_ITERATIONS = 2


def _test_engine_factory():
    if _ENGINE_FACTORY is None:
        return None
    if os.environ.get("MATCOUPLY_AMD_TEST_ENGINE") != "1":
        raise RuntimeError("a substitute compute engine is installed outside the test-suite: matcouply_amd runs on the "
                           "HIP engine only (set MATCOUPLY_AMD_TEST_ENGINE=1 in a test harness to allow a checker)")
    return _ENGINE_FACTORY


def partition_slabs(n_rows, world_size, contiguous=True):
    """Split the I coupled matrices over `world_size` ranks so that every rank holds about the same number of ROWS
    (sum of J_i) - the per-rank work of every phase is proportional to its rows, not to its number of matrices.

    n_rows: J_i for every matrix (or anything with a `.shape[0]`).  Returns a list of `world_size` index arrays; rank k
    passes `[matrices[i] for i in parts[k]]` to `cmf_aoadmm(..., group=pg)` (the rows of A come back in that order).

    contiguous=True   consecutive ranges [lo_k, hi_k): the cut after slab i is placed where the prefix sum of rows is
                      closest to k / world_size of the total (keeps the matrices in order; imbalance <= max J_i / 2 rows);
    contiguous=False  longest-processing-time greedy (largest matrix first onto the least loaded rank; indices sorted
                      within a rank): within a fraction of a percent for ragged J_i such as BASELINE config 4.
    """
    J = np.asarray([getattr(m, "shape", (m,))[0] if not np.isscalar(m) else m for m in n_rows], dtype=np.int64)
    I, W = len(J), int(world_size)
    if W < 1:
        raise ValueError("world_size must be positive")
    if contiguous:
        prefix = np.concatenate([[0], np.cumsum(J)])
        total = prefix[-1]
        cuts = [0]
        for k in range(1, W):
            target = total * k / W
            i = int(np.searchsorted(prefix, target))
            if i > 0 and abs(prefix[i - 1] - target) <= abs(prefix[min(i, I)] - target):
                i -= 1
            cuts.append(min(max(i, cuts[-1]), I))
        cuts.append(I)
        return [np.arange(cuts[k], cuts[k + 1], dtype=np.int64) for k in range(W)]
    load = np.zeros(W, dtype=np.int64)
    parts = [[] for _ in range(W)]
    for i in np.argsort(-J, kind="stable"):
        k = int(np.argmin(load))
        parts[k].append(int(i))
        load[k] += J[i]
    return [np.array(sorted(p), dtype=np.int64) for p in parts]


class PackedMatrices:
    """The I coupled matrices already packed along rows in device memory: X [sum J_i, K] float32 (torch CUDA
    tensor) + row_ptr [I+1].  Behaves like the list of the I matrices (len / iteration / indexing give views),
    and lets callers that keep their data in HBM skip the host-side packing."""

    def __init__(self, X, row_ptr):
        self.X = X
        self.row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int64)
        if self.row_ptr[0] != 0 or self.row_ptr[-1] != X.shape[0]:
            raise ValueError("row_ptr must start at 0 and end at X.shape[0]")

    def __len__(self):
        return len(self.row_ptr) - 1

    def __getitem__(self, i):
        return self.X[self.row_ptr[i]: self.row_ptr[i + 1]]

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]


# ------------------------------------------------------------------------------------------------------------
# initialisation (decomposition.py:18-89)
# ------------------------------------------------------------------------------------------------------------
def _data_on_device(matrices):
    """the matrices already live in HBM: a PackedMatrices, or torch CUDA tensors"""
    if isinstance(matrices, PackedMatrices):
        return is_torch(matrices.X) and matrices.X.is_cuda
    try:
        return len(matrices) > 0 and all(is_torch(m) and m.is_cuda for m in matrices)
    except TypeError:
        return False


_DEVICE_SVD_MAX_K = 2048  # the device initialiser keeps K x K fp64 Gram matrices per matrix in flight (32 MB each at this K)


def _device_svd_applies(matrices, rank):
    """The device form of init="svd" serves the shapes its subspace iteration covers: rank > K, a matrix with fewer rows than
    components, or a K whose Gram matrices would not fit the initialiser's workspace take the host path, as the reference does
    (ADVICE r5).  (`svd=` only names the driver: every valid name is the same thin LAPACK SVD on the host, _utils.get_svd.)"""
    try:
        K = int(shape(matrices[0])[1])
        rows = [int(shape(m)[0]) for m in matrices] if not isinstance(matrices, PackedMatrices) else \
            [int(matrices.row_ptr[i + 1] - matrices.row_ptr[i]) for i in range(len(matrices))]
    except Exception:
        return False
    return rank <= K <= _DEVICE_SVD_MAX_K and min(rows) >= rank


def initialize_cmf(matrices, rank, init, svd_fun, random_state=None, init_params=None):
    random_state = check_random_state(random_state)
    if isinstance(init, (tuple, list, CoupledMatrixFactorization)):
        weights, (A, B_is, C) = init
        if weights is not None:
            scaled_A = weights * A
            return CoupledMatrixFactorization((None, (scaled_A, B_is, C)))
        return CoupledMatrixFactorization(init)
    if init == "random":
        I = len(matrices)
        K = shape(matrices[0])[1]
        A = random_state.uniform(size=(I, rank))
        C = random_state.uniform(size=(K, rank))
        B_is = [random_state.uniform(size=(shape(matrix)[0], rank)) for matrix in matrices]
        return CoupledMatrixFactorization((None, [A, B_is, C]))
    if (init == "svd" or init == "threshold_svd") and _data_on_device(matrices) and _device_svd_applies(matrices, rank):
        # data resident in HBM: the singular vectors on the device, no copy of X to the host (mcl_svd_init: fp64 subspace
        # iteration on the Gram matrices).  The vectors are LAPACK's up to their SIGNS (largest-magnitude entry positive here):
        # B_i and C come from independent decompositions, so the reference's trajectory from this initialiser is reproduced
        # by the host path below only - which host-resident data keeps taking.
        X, row_ptr = _pack(matrices, _device())
        B, C, info = _engine.svd_init(X, row_ptr, rank, threshold=(init == "threshold_svd"))
        info_min = int(info.min().item())
        if info_min <= -1000:
            import warnings

            warnings.warn("svd initialisation on the device: a matrix has numerical rank below `rank` (its trailing singular "
                          "vectors are not determined by the data); taking the host (LAPACK) path, which completes the basis",
                          RuntimeWarning)
        else:
            if info_min < 0:
                import warnings

                warnings.warn("svd initialisation on the device: the subspace iteration of some matrices had not settled after 400 "
                              "iterations (no gap behind the leading singular values); the vectors are approximate", RuntimeWarning)
            A = torch.ones((len(row_ptr) - 1, rank), dtype=torch.float32, device=X.device)
            B_is = [B[int(row_ptr[i]): int(row_ptr[i + 1])] for i in range(len(row_ptr) - 1)]
            return CoupledMatrixFactorization((None, [A, B_is, C]))
    if init == "svd" or init == "threshold_svd":
        # one-off set-up on the host (decomposition.py:42-53)
        from ._utils import to_numpy

        if svd_fun is None:
            svd_fun = get_svd("truncated_svd")
        mats = [np.asarray(to_numpy(m), dtype=np.float64) for m in matrices]
        A = np.ones((len(mats), rank))
        B_is = [svd_fun(m, n_eigenvecs=rank)[0] for m in mats]
        C = svd_fun(np.concatenate(mats, 0), n_eigenvecs=rank)[2].T
        if init == "threshold_svd":
            B_is = [np.clip(B_i, 0, float("inf")) for B_i in B_is]
            C = np.clip(C, 0, float("inf"))
        return CoupledMatrixFactorization((None, [A, B_is, np.ascontiguousarray(C)]))
    if init in ("parafac2_als", "cp_als", "parafac_als", "cp_hals", "parafac_hals"):
        raise NotImplementedError(
            f'init="{init}" delegates to TensorLy decompositions in the reference (decomposition.py:55-73) and is '
            "out of scope of this engine; pass an explicit (weights, (A, B_is, C)) tuple instead.")
    raise ValueError('Initialization method "{}" not recognized'.format(init))


def initialize_aux(matrices, rank, reg, random_state):
    A_aux_list = [A_reg.init_aux(matrices, rank, 0, random_state=random_state) for A_reg in reg[0]]
    B_aux_list = [B_reg.init_aux(matrices, rank, 1, random_state=random_state) for B_reg in reg[1]]
    C_aux_list = [C_reg.init_aux(matrices, rank, 2, random_state=random_state) for C_reg in reg[2]]
    return A_aux_list, B_aux_list, C_aux_list


def initialize_dual(matrices, rank, reg, random_state):
    A_dual_list = [A_reg.init_dual(matrices, rank, 0, random_state=random_state) for A_reg in reg[0]]
    B_dual_list = [B_reg.init_dual(matrices, rank, 1, random_state=random_state) for B_reg in reg[1]]
    C_dual_list = [C_reg.init_dual(matrices, rank, 2, random_state=random_state) for C_reg in reg[2]]
    return A_dual_list, B_dual_list, C_dual_list


# ------------------------------------------------------------------------------------------------------------
# public diagnostics helper (decomposition.py:347-417) on host/device arrays
# ------------------------------------------------------------------------------------------------------------
def _sq(x):
    return float((x.double() ** 2).sum()) if is_torch(x) else float(np.sum(np.asarray(x, dtype=np.float64) ** 2))


def _root_sum_squared_list(x_list):
    return np.sqrt(sum(_sq(x) for x in x_list))


def compute_feasibility_gaps(cmf, regs, A_aux_list, B_aux_list, C_aux_list):
    r"""Feasibility gaps ||aux - x|| / ||x|| per penalty and mode; the B-mode gap pools all B_i (:404-417)."""
    weights, (A, B_is, C) = cmf
    A_norm = np.sqrt(_sq(A))
    B_norm = _root_sum_squared_list(B_is)
    C_norm = np.sqrt(_sq(C))
    A_gaps = [np.sqrt(_sq(A_reg.subtract_from_aux(A_aux, A))) / A_norm for A_reg, A_aux in zip(regs[0], A_aux_list)]
    B_gaps = [_root_sum_squared_list(B_reg.subtract_from_auxes(B_is_aux, B_is)) / B_norm
              for B_reg, B_is_aux in zip(regs[1], B_aux_list)]
    C_gaps = [np.sqrt(_sq(C_reg.subtract_from_aux(C_aux, C))) / C_norm for C_reg, C_aux in zip(regs[2], C_aux_list)]
    return A_gaps, B_gaps, C_gaps


def _check_inner_convergence(factor_matrix, old_factor_matrix, cmf, reg_list, aux_list, mode, inner_tol):
    """Inner stopping rule on host objects (:92-116): relative change and every feasibility gap of the mode below
    `inner_tol`.  (Inside `cmf_aoadmm` the same test runs on device tensors between the step calls.)"""
    if not inner_tol or inner_tol < 0:
        return False
    if mode == 1:
        norm = _root_sum_squared_list(factor_matrix)
        change = _root_sum_squared_list([B_i - prev_B_i for B_i, prev_B_i in zip(factor_matrix, old_factor_matrix)])
    else:
        norm = np.sqrt(_sq(factor_matrix))
        change = np.sqrt(_sq(factor_matrix - old_factor_matrix))
    if change > inner_tol * norm:
        return False
    if len(reg_list) == 0:
        return True
    regs, auxes = [[], [], []], [[], [], []]
    regs[mode], auxes[mode] = reg_list, aux_list
    return max(compute_feasibility_gaps(cmf, regs, *auxes)[mode]) < inner_tol


def _cmf_reconstruction_error(matrices, cmf, norm_matrices=None, intermediate_A_calculations=None):
    """sqrt(max(0, ||X||^2 - 2 <X, M> + ||M||^2)) on host objects (:420-452); with the by-products of `admm_update_A`
    (`rhses`, `cross_products`) the two model terms need no pass over the matrices (:445-449).  fp64 accumulation.
    (Inside `cmf_aoadmm` the same quantity comes out of the A-phase kernels, `mcl_diagnostics`.)"""
    f64 = lambda x: to_numpy(x).astype(np.float64)
    norm_X_sq = sum(_sq(m) for m in matrices) if norm_matrices is None else float(norm_matrices) ** 2
    weights, (A, B_is, C) = cmf
    A = f64(A)
    if weights is not None:
        A = A * f64(weights)
    if intermediate_A_calculations is None:
        C64 = f64(C)
        CtC = C64.T @ C64
        inner_product = norm_cmf_sq = 0.0
        for i, B_i in enumerate(B_is):
            B_i = f64(B_i) * A[i]
            inner_product += float(np.sum((f64(matrices[i]) @ C64) * B_i))
            norm_cmf_sq += float(np.sum((B_i.T @ B_i) * CtC))
    else:
        A_rhses, cross_products = intermediate_A_calculations
        inner_product = sum(float(np.sum(f64(rhs_i) * a_i)) for rhs_i, a_i in zip(A_rhses, A))
        norm_cmf_sq = sum(float(a_i @ f64(cross_products[i]) @ a_i) for i, a_i in enumerate(A))
    return np.sqrt(max(0.0, norm_X_sq - 2 * inner_product + norm_cmf_sq))


def _compute_l2_penalty(cmf, l2_parameters):
    """sum over modes of l2_m / 2 ||factor_m||^2 (:617-627)"""
    weights, (A, B_is, C) = cmf
    l2reg = 0
    if l2_parameters[0]:
        l2reg += 0.5 * l2_parameters[0] * _sq(A)
    if l2_parameters[1]:
        l2reg += 0.5 * l2_parameters[1] * sum(_sq(B_i) for B_i in B_is)
    if l2_parameters[2]:
        l2reg += 0.5 * l2_parameters[2] * _sq(C)
    return l2reg


# ------------------------------------------------------------------------------------------------------------
# argument parsing (decomposition.py:455-614)
# ------------------------------------------------------------------------------------------------------------
def _listify(input_value, param_name):
    """One value per mode from a dict {mode: value}, a scalar (broadcast) or a length-3 iterable."""
    if hasattr(input_value, "get"):
        return [input_value.get(mode, None) for mode in range(3)]
    if not is_iterable(input_value):
        return [input_value, input_value, input_value]
    per_mode = list(input_value)
    if len(per_mode) != 3:
        raise ValueError(
            "All parameters must be a dictionary, non-iterable value or non-dictionary iterable of length 3."
            f" {param_name} is iterable of length {len(per_mode)}.")
    return per_mode


def _parse_all_penalties(non_negative, lower_bound, upper_bound, l2_norm_bound, unimodal, parafac2, l1_penalty,
                         tv_penalty, generalized_l2_penalty, svd, regs, dual_init, aux_init, verbose):
    if regs is None:
        regs = [[], [], []]
    elif is_iterable(regs):
        for modereg in regs:
            if not is_iterable(modereg):
                raise TypeError(
                    "regs should contain an iterable of iterables containting "
                    "matcouply.penalties.ADMMMPenalty instances at least one of the"
                    f"elements in regs were not iterable (regs={regs})")
            else:
                for reg in modereg:
                    if not isinstance(reg, penalties.ADMMPenalty):
                        raise TypeError(
                            "regs should contain an iterable of iterables containting "
                            "matcouply.penalties.ADMMMPenalty instances at least one of the"
                            f"elements in regs contained something other than an ADMMPenalty (regs={regs})")
    regs = [copy(reg_list) for reg_list in regs]  # avoid side effects on the input lists

    non_negative = _listify(non_negative, "non_negative")
    upper_bound = _listify(upper_bound, "upper_bound")
    lower_bound = _listify(lower_bound, "lower_bound")
    l2_norm_bound = _listify(l2_norm_bound, "l2_norm_bound")
    unimodal = _listify(unimodal, "unimodal")
    parafac2 = [False, True, False] if parafac2 else [False, False, False]
    l1_penalty = _listify(l1_penalty, "l1_penalty")
    generalized_l2_penalty = _listify(generalized_l2_penalty, "generalized_l2_penalty")
    tv_penalty = _listify(tv_penalty, "tv_penalty")

    for mode in range(3):
        parsed_regs = _parse_mode_penalties(
            non_negative=non_negative[mode], lower_bound=lower_bound[mode], upper_bound=upper_bound[mode],
            l2_norm_bound=l2_norm_bound[mode], unimodal=unimodal[mode], parafac2=parafac2[mode],
            l1_penalty=l1_penalty[mode], tv_penalty=tv_penalty[mode],
            generalized_l2_penalty=generalized_l2_penalty[mode], svd=svd, dual_init=dual_init, aux_init=aux_init)
        regs[mode] = parsed_regs + regs[mode]

    if verbose:
        print("All regularization penalties (including regs list):")
        for mode, reg in enumerate(regs):
            print(f"* Mode {mode}:")
            if len(reg) == 0:
                print("   - (no regularization added)")
            for single_reg in reg:
                print(f"   - {single_reg}")
    return regs


def _parse_mode_penalties(non_negative, lower_bound, upper_bound, l2_norm_bound, unimodal, parafac2, l1_penalty,
                          tv_penalty, generalized_l2_penalty, svd, dual_init, aux_init):
    """Keyword arguments of one mode -> its penalty list, in the reference's order Parafac2, Unimodality,
    GeneralizedL2, L2Ball, TV, L1, Box, NonNegativity (decomposition.py:571-612).  Table-driven: every row is
    (is the keyword set?, constructor, does the penalty absorb the non-negativity flag?); a plain NonNegativity is only
    appended when no earlier row absorbed the flag."""
    init = dict(aux_init=aux_init, dual_init=dual_init)
    l1 = l1_penalty or 0
    has_gl2 = generalized_l2_penalty is not None and generalized_l2_penalty is not False  # None, False or the norm matrix
    has_box = lower_bound is not None or upper_bound is not None

    def box():
        lo = -float("inf") if lower_bound is None else lower_bound
        return penalties.Box(max(lo, 0) if non_negative else lo, upper_bound, **init)

    rules = (
        (parafac2, lambda: penalties.Parafac2(svd=svd, **init), False),
        (unimodal, lambda: penalties.Unimodality(non_negativity=non_negative, **init), True),
        (has_gl2, lambda: penalties.GeneralizedL2Penalty(generalized_l2_penalty, svd=svd, **init), False),
        (l2_norm_bound, lambda: penalties.L2Ball(l2_norm_bound, non_negativity=non_negative, **init), True),
        (tv_penalty, lambda: penalties.TotalVariationPenalty(tv_penalty, l1_strength=l1, **init), False),
        # the L1 term rides inside the total-variation penalty when both are requested
        (l1 and not tv_penalty, lambda: penalties.L1Penalty(l1, non_negativity=non_negative, **init), True),
        (has_box, box, True),
    )
    regs, absorbed = [], False
    for requested, build, absorbs in rules:
        if requested:
            regs.append(build())
            absorbed = absorbed or absorbs
    if non_negative and not absorbed:
        regs.append(penalties.NonNegativity(**init))
    return regs


def _check_feasibility(feasibility_gaps, feasibility_tol):
    """True when every feasibility gap of every mode is below the tolerance (vacuously for no penalties)."""
    worst = max((gap for mode_gaps in feasibility_gaps for gap in mode_gaps), default=-float("inf"))
    return worst < feasibility_tol


class _StopRule:
    """The stopping test of the outer loop (decomposition.py:990-1053) as an object.  Quirks kept on purpose: the
    absolute criterion is only looked at when `tol` is set, and it tests the newest loss."""

    RELATIVE = "FEASIBILITY GAP CRITERION AND RELATIVE LOSS CRITERION SATISFIED"
    ABSOLUTE = "FEASIBILITY GAP CRITERION AND ABSOLUTE LOSS CRITERION SATISFIED"
    EXHAUSTED = "MAXIMUM NUMBER OF ITERATIONS REACHED"

    def __init__(self, tol, absolute_tol, feasibility_tol):
        self.tol, self.absolute_tol, self.feasibility_tol = tol, absolute_tol, feasibility_tol

    @property
    def active(self):
        return bool(self.tol or self.absolute_tol)

    def feasible(self, gaps):
        return self.feasibility_tol and _check_feasibility(gaps, self.feasibility_tol)

    def verdict(self, feasible, losses):
        """message of the criterion that fires on the last two losses, or None"""
        if not self.tol:
            return None
        relative = abs(losses[-2] - losses[-1]) < (self.tol * losses[-2])
        absolute = losses[-1] < self.absolute_tol
        if feasible and relative:
            return self.RELATIVE
        if feasible and absolute:
            return self.ABSOLUTE
        return None


class _Progress:
    """Console output of the outer loop (`verbose`: 0 / None silent, n > 0 every n-th iteration, -1 final message only)."""

    def __init__(self, verbose):
        self.verbose = verbose

    def _due(self, it):
        return bool(self.verbose) and self.verbose > 0 and it % self.verbose == 0

    @staticmethod
    def _gaps(gaps):
        print("Feasibility gaps for A: {}".format(gaps[0]))
        print("Feasibility gaps for the Bi-matrices: {}".format(gaps[1]))
        print("Feasibility gaps for C: {}".format(gaps[2]))

    def initial(self, gaps):
        if self.verbose and self.verbose > 0:
            self._gaps(gaps)

    def iteration(self, it, rec_error=None, loss=None, variation=None, gaps=None):
        if not self._due(it):
            return
        if gaps is None:
            print("Coupled matrix factorization iteration={}".format(it))
            return
        shown = ("NOT COMPUTED",) * 3 if rec_error is None else (rec_error, loss, variation)
        print("Coupled matrix factorization iteration={}, ".format(it)
              + "reconstruction error={}, ".format(shown[0])
              + ("regularized loss={}, " if rec_error is None else "regularized loss={} ").format(shown[1])
              + "regularized loss variation={}.".format(shown[2]))
        self._gaps(gaps)

    def converged(self, it, message):
        if self.verbose:
            print("converged in {} iterations: {}".format(it, message))

    def exhausted(self):
        if self.verbose:
            print("REACHED MAXIMUM NUMBER OF ITERATIONS")


class ADMMVars(NamedTuple):
    auxes: tuple  #: Length three tuple containing a list of auxiliary factor matrices for each mode
    duals: tuple  #: Length three tuple containing a list of dual variables for each mode


class DiagnosticMetrics(NamedTuple):
    rec_errors: list  #: reconstruction errors, one per iteration plus the initial one
    feasibility_gaps: list  #: feasibility gaps, one per iteration plus the initial one
    regularized_loss: list  #: regularized loss, one per iteration plus the initial one
    satisfied_stopping_condition: Optional[bool]  #: None if no tolerance is set
    satisfied_feasibility_condition: Optional[bool]  #: None if no tolerance is set
    n_iter: int  #: Number of iterations ran
    message: str  #: Convergence message


# ------------------------------------------------------------------------------------------------------------
# host <-> device plumbing
# ------------------------------------------------------------------------------------------------------------
def _device():
    if torch is None or not torch.cuda.is_available():
        raise _engine.EngineError(
            "no HIP device visible: matcouply_amd runs the AO-ADMM updates on an MI355X (gfx950); there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def _dtype():
    """float32 on the device; a substituted checker engine (tests) may ask for float64 state."""
    sub = _test_engine_factory()
    return getattr(sub, "dtype", torch.float32) if sub is not None else torch.float32


def _to_dev(x, device):
    if is_torch(x):
        return x.detach().to(device=device, dtype=_dtype()).contiguous().clone()
    return torch.as_tensor(np.ascontiguousarray(np.asarray(x)), device=device).to(_dtype()).contiguous()


def _pack(matrices, device):
    """-> (X [N, K] float32 on device, row_ptr int64 [I+1])"""
    if isinstance(matrices, PackedMatrices):
        X = matrices.X
        if not (is_torch(X) and X.is_cuda and X.dtype == torch.float32 and X.is_contiguous()):
            raise TypeError("PackedMatrices.X must be a contiguous float32 CUDA tensor")
        return X, matrices.row_ptr
    mats = list(matrices)
    K = shape(mats[0])[1]
    for m in mats:
        if len(shape(m)) != 2 or shape(m)[1] != K:
            raise ValueError("All matrices must be second order tensors with the same number of columns")
    rows = [shape(m)[0] for m in mats]
    row_ptr = np.concatenate([[0], np.cumsum(rows)]).astype(np.int64)
    if all(is_torch(m) for m in mats):
        X = torch.cat([m.detach().to(device=device, dtype=torch.float32) for m in mats], 0).contiguous()
    else:
        host = np.empty((int(row_ptr[-1]), K), dtype=np.float32)
        for i, m in enumerate(mats):
            host[row_ptr[i]: row_ptr[i + 1]] = m.detach().cpu().numpy() if is_torch(m) else np.asarray(m)
        X = torch.from_numpy(host).to(device)
    return X, row_ptr


def _pack_rows(list_of_matrices, device):
    if len(list_of_matrices) == 0:
        return torch.zeros((0, 0), dtype=torch.float32, device=device)
    if all(is_torch(m) for m in list_of_matrices):
        return torch.cat([m.detach().to(device=device, dtype=_dtype()) for m in list_of_matrices], 0).contiguous()
    return _to_dev(np.concatenate([m.detach().cpu().numpy() if is_torch(m) else np.asarray(m)
                                   for m in list_of_matrices], 0), device)


class _Out:
    """Converts device results back to the array type / dtype / device of the caller's matrices."""

    def __init__(self, matrices):
        first = matrices.X if isinstance(matrices, PackedMatrices) else matrices[0]
        self.torch_out = is_torch(first)
        self.dtype = first.dtype
        self.device = first.device if self.torch_out else None

    def __call__(self, t):
        if self.torch_out:
            return t.detach().to(device=self.device, dtype=self.dtype).clone()
        return t.detach().cpu().numpy().astype(self.dtype if np.issubdtype(self.dtype, np.floating) else np.float64)

    def split(self, t, row_ptr):
        full = self(t)
        return [full[row_ptr[i]: row_ptr[i + 1]] for i in range(len(row_ptr) - 1)]


def _aux_to_device(aux, device):
    """Move an auxiliary variable in any parametrisation (array, list of arrays, tuple of those) to device tensors."""
    if isinstance(aux, tuple):
        return tuple(_aux_to_device(a, device) for a in aux)
    if isinstance(aux, list):
        return [_aux_to_device(a, device) for a in aux]
    return _to_dev(aux, device)


def _aux_out(aux, out):
    if isinstance(aux, tuple):
        return tuple(_aux_out(a, out) for a in aux)
    if isinstance(aux, list):
        return [_aux_out(a, out) for a in aux]
    return out(aux)


def _default_engine_factory(**kw):
    return _engine.HipEngine(**kw)


_DIRECT_COMMS = {}


def _direct_comm(group, device=None):
    """the cached direct RCCL communicator of a process group on `device` (None: use torch.distributed), see _rccl.py.
    The cache entry is tied to the RESOLVED group object (the default group changes identity when a host destroys and
    re-creates it), its world size, this rank and the device: an entry whose group is gone is dropped, not reused."""
    import torch.distributed as dist

    from . import _rccl

    resolved = group if group is not None else dist.group.WORLD
    try:
        if device is None or getattr(device, "index", None) is None:  # the caller's current device, as DirectComm resolves it
            device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else device
        key = (id(resolved), dist.get_world_size(group), dist.get_rank(group), str(device))
    except Exception:
        return None
    try:  # entries of process groups that have been destroyed since (their communicators are abandoned with them)
        alive = dist.distributed_c10d._world.pg_map
        for k in [k for k, (g, _) in _DIRECT_COMMS.items() if g not in alive]:
            del _DIRECT_COMMS[k]
    except Exception:
        pass
    if key not in _DIRECT_COMMS:
        _DIRECT_COMMS[key] = (resolved, _rccl.DirectComm.try_create(group, device))  # the group is kept alive with its communicator
    return _DIRECT_COMMS[key][1]


# ------------------------------------------------------------------------------------------------------------
# the solver
# ------------------------------------------------------------------------------------------------------------
def cmf_aoadmm(
    matrices,
    rank,
    init="random",
    n_iter_max=1000,
    l2_penalty=None,
    tv_penalty=None,
    l1_penalty=None,
    non_negative=None,
    unimodal=None,
    generalized_l2_penalty=None,
    l2_norm_bound=None,
    lower_bound=None,
    upper_bound=None,
    parafac2=None,
    regs=None,
    feasibility_penalty_scale=1,
    constant_feasibility_penalty=False,
    aux_init="random_uniform",
    dual_init="random_uniform",
    svd="truncated_svd",
    init_params=None,
    random_state=None,
    tol=1e-8,
    absolute_tol=1e-10,
    feasibility_tol=1e-4,
    inner_tol=None,
    inner_n_iter_max=5,
    update_A=True,
    update_B_is=True,
    update_C=True,
    return_admm_vars=False,
    return_errors=False,
    verbose=False,
    *,
    group=None,
    gather_A=False,
    arithmetic="auto",
    _byproducts=None,
):
    r"""Fit a regularized coupled matrix factorization model with AO-ADMM on an MI355X.

    Parameters, defaults, return values (``cmf`` or ``(cmf[, ADMMVars][, DiagnosticMetrics])``) and stopping rules are
    those of ``matcouply.decomposition.cmf_aoadmm`` (reference decomposition.py:662-1100): regularization parameters may
    be ``None``, a scalar (all modes), a length-3 list, or a ``{mode: value}`` dict; mode 0 is A, mode 1 the B_i, mode 2 C.

    ``matrices`` is a list of I arrays (NumPy or torch, J_i x K) or a :class:`PackedMatrices` already in HBM.  The
    arithmetic runs in fp32 on the device (rank x rank systems and all reductions in fp64); results are returned in the
    array type and dtype of the input.  Not supported (out of scope, raise ``NotImplementedError``): TensorLy-ALS
    initialisations.  ``inner_tol`` > 0 is evaluated by the engine on the device (single-device runs; sharded runs: on a host-driven step path).  Penalties without a native kernel (user
    subclasses of ``matcouply_amd.penalties.ADMMPenalty``) are evaluated through their own Python methods on device
    tensors between the native solve and dual-update steps.

    Sharded runs (``group=``, keyword-only, not in the reference): every rank of the ``torch.distributed`` process group
    passes ITS matrices (see :func:`partition_slabs`) and gets back its rows of ``A``, its ``B_i`` and the replicated ``C``;
    ``arithmetic`` (keyword-only, not in the reference): ``"auto"`` - problems of at most 2^20 elements take every contraction
    as fp64 sums of exact products and their inner ADMM loops in fp64 (the reference's own regime: parity at the storage level
    of fp32), larger ones the fp32 matrix-core kernels; ``"exact"`` forces the former at any size (slower; for large problems
    with ill-conditioned penalty-free modes), ``"fast"`` the latter.
    ``gather_A=True`` all-gathers the rows of ``A`` once at the end (rank order, i.e. the original order for contiguous
    partitions): the factorization returned stays the rank's own, the whole ``A`` rides along as ``cmf.A_all`` with this
    rank's rows at ``cmf.rows_of_rank`` (the ADMM variables of mode 0 stay rank-local).  Per outer iteration
    the ranks exchange the fp64 normal equations ``[G | R]`` of the C-phase, the diagnostic sums while a stopping rule is
    active, one ``r*r + 1`` reduction per inner iteration with PARAFAC2 and a scalar MAX per phase with a constant
    feasibility penalty.  A ``TotalVariationPenalty`` on the ``B_i`` or on ``C`` works under ``group=`` (its value, summed on
    the host, travels with the diagnostic sums; the run then takes the host-driven loop), and so does a host-evaluated
    (user-defined / overridden) ``MatrixPenalty`` on the ``B_i`` - its prox acts on one matrix at a time.  NOT supported with
    ``group=`` (``NotImplementedError``): host-evaluated ``MatricesPenalty`` classes on mode 1 (they may couple matrices
    of different ranks).  Matrix penalties on mode 0 (L2 ball, unimodality, total variation along the rows of ``A``, a
    user's ``MatrixPenalty``) need ``constant_feasibility_penalty``, as in the reference; the L2 ball all-reduces its column
    norms, every other one is evaluated by each rank on the all-gathered ``A`` (I x rank floats).  None of these occurs in the
    BASELINE configurations.

    >>> import numpy as np, matcouply_amd
    >>> len(matcouply_amd.decomposition._listify({1: 0.5}, "l1_penalty"))
    3
    """
    if arithmetic not in ("auto", "exact", "fast"):
        raise ValueError(f'arithmetic must be "auto", "exact" or "fast", not {arithmetic!r}')
    random_state = check_random_state(random_state)
    svd_fun = get_svd(svd)
    cmf = initialize_cmf(matrices, rank, init, svd_fun=svd_fun, random_state=random_state, init_params=init_params)

    l2_penalty = _listify(l2_penalty, "l2_penalty")
    l2_penalty = [l2 if l2 is not None else 0 for l2 in l2_penalty]

    regs = _parse_all_penalties(
        non_negative=non_negative, lower_bound=lower_bound, upper_bound=upper_bound, l2_norm_bound=l2_norm_bound,
        unimodal=unimodal, parafac2=parafac2, l1_penalty=l1_penalty, tv_penalty=tv_penalty,
        generalized_l2_penalty=generalized_l2_penalty, svd=svd, regs=regs, dual_init=dual_init, aux_init=aux_init,
        verbose=verbose)
    if not update_A:
        regs[0] = []
    if not update_B_is:
        regs[1] = []
    if not update_C:
        regs[2] = []
    # inner_tol > 0 (early exit of the inner ADMM loops, decomposition.py:90-117) needs a convergence test after every
    # inner iteration.  Single-device runs: the ENGINE evaluates it (mcl_options.inner_tol: one launch per step, a device-side
    # flag the remaining inner launches test; native prox kernels stay native, no host synchronisation).  Sharded runs (the
    # norms must be all-reduced over the ranks) and the CPU checker keep the step path: native solves, every prox through
    # the penalty objects' own methods on device tensors, the test on the host
    check_inner = bool(inner_tol) and inner_tol > 0
    if isinstance(constant_feasibility_penalty, str) and constant_feasibility_penalty not in {"A", "B"}:
        raise ValueError(
            f"If `constant_feasibility_penalty` is a string, it must be 'A' or 'B', not {constant_feasibility_penalty}")
    constant_A = (constant_feasibility_penalty and not isinstance(constant_feasibility_penalty, str)
                  ) or constant_feasibility_penalty == "A"
    constant_B = (constant_feasibility_penalty and not isinstance(constant_feasibility_penalty, str)
                  ) or constant_feasibility_penalty == "B"

    # ---- initial ADMM state with the reference's draw order (aux of modes 0,1,2 then duals of modes 0,1,2) ----
    A_aux_list, B_aux_list, C_aux_list = initialize_aux(matrices, rank, regs, random_state=random_state)
    A_dual_list, B_dual_list, C_dual_list = initialize_dual(matrices, rank, regs, random_state=random_state)

    # ---- move everything to the device -----------------------------------------------------------------------
    sub = _test_engine_factory()
    factory = sub or _default_engine_factory
    device = _device() if sub is None else getattr(sub, "device", torch.device("cpu"))
    X, row_ptr = _pack(matrices, device) if sub is None else sub.pack(matrices)
    out = _Out(matrices)
    _, (A0, B0_is, C0) = cmf
    A, B, C = _to_dev(A0, device), _pack_rows(B0_is, device), _to_dev(C0, device)
    if B.shape != (X.shape[0], rank):
        raise ValueError("The B_i matrices of `init` do not match the shapes of `matrices`")

    native = [[], [], []]
    device_inner = check_inner and sub is None and group is None
    aux_lists, dual_lists = (A_aux_list, B_aux_list, C_aux_list), (A_dual_list, B_dual_list, C_dual_list)
    # host-evaluated penalties: the Python object keeps the auxiliary variable in ITS parametrisation (on the device),
    # the engine sees `aux_as_matrix` of it (reference penalties.py:311-343)
    ext_aux = {}
    for mode in range(3):
        for k, (reg, aux, dual) in enumerate(zip(regs[mode], aux_lists[mode], dual_lists[mode])):
            desc = None if (check_inner and not device_inner) else penalties.native_descriptor_of(reg)
            gl2_matrix = None
            if desc is not None and desc[0] == _engine.PEN_GL2:
                # every matrix of the mode must have as many rows as the norm matrix (anything else fails in the reference's
                # product too: left to the penalty's own method); a sharded A holds only this rank's rows of the I x I problem
                gl2_matrix, n_gl2 = reg._native_matrix()
                rows_m = [shape(m_)[0] for m_ in matrices] if mode == 1 else [len(matrices) if mode == 0 else shape(matrices[0])[1]]
                if any(rw != n_gl2 for rw in rows_m) or (group is not None and mode == 0):
                    desc = None
            dual_t = _pack_rows(dual, device) if mode == 1 else _to_dev(dual, device)
            if desc is None:
                if mode == 1:
                    obj = _aux_to_device(aux, device)
                    mat = _pack_rows(reg.auxes_as_matrices(obj), device)
                else:
                    obj = _aux_to_device(aux, device)
                    mat = _to_dev(reg.aux_as_matrix(obj), device)
                ext_aux[(mode, k)] = obj
                native[mode].append(_engine.NativeReg(_engine.PEN_EXTERNAL, mat, dual_t))
                continue
            kind, nonneg, p0, p1 = desc
            if kind == _engine.PEN_PARAFAC2:
                P_is, Delta = aux
                native[mode].append(_engine.NativeReg(kind, _pack_rows(P_is, device), dual_t,
                                                      aux2=_to_dev(Delta, device)))
            else:
                aux_t = _pack_rows(aux, device) if mode == 1 else _to_dev(aux, device)
                extra = {}
                if kind == _engine.PEN_GL2:
                    extra = dict(matrix=torch.as_tensor(gl2_matrix, dtype=torch.float64, device=device).contiguous(), matrix_rows=n_gl2)
                native[mode].append(_engine.NativeReg(kind, aux_t, dual_t, non_negativity=nonneg, p0=p0, p1=p1, **extra))

    world = 1
    dist = None
    if group is not None:
        import torch.distributed as dist  # noqa: F811

        world = dist.get_world_size(group)
        rank_id = dist.get_rank(group)
    else:
        rank_id = 0
    # the arithmetic of small problems (exact-products mode) is chosen by the size of the WHOLE problem:
    # every rank of a sharded run, and every rank layout of the same problem, then computes with the same kernels
    exact_products = {"auto": 0, "exact": 1, "fast": 2}[arithmetic]
    n_el_total = float(X.shape[0]) * float(X.shape[1])
    if world > 1 and arithmetic == "auto":
        n_el = torch.tensor([n_el_total], dtype=torch.float64, device=X.device)
        dist.all_reduce(n_el, group=group)
        n_el_total = float(n_el.item())
        exact_products = 1 if n_el_total <= float(1 << 20) else 2
    eng = factory(X=X, row_ptr=row_ptr, rank=rank, A=A, B=B, C=C, regs=native, l2_penalty=l2_penalty,
                  inner_n_iter_max=inner_n_iter_max, feasibility_penalty_scale=feasibility_penalty_scale,
                  constant_A=constant_A, constant_B=constant_B, exact_products=exact_products,
                  **(dict(inner_tol=inner_tol) if device_inner else {}))
    # the sharded code path (step calls with the reductions in between): taken with more than one rank - and, for
    # rehearsals of that path on a single-GPU box, with a one-rank group when MCL_FORCE_SHARDED_PATH=1
    sharded = world > 1 or (group is not None and os.environ.get("MCL_FORCE_SHARDED_PATH") == "1")
    needs_B_steps = sharded and (constant_B or any(r.kind == _engine.PEN_PARAFAC2 for r in native[1]))
    needs_A_steps = sharded and constant_A
    # mode 0 under sharding: the rows of A live on different ranks.  Row-separable penalties need nothing; an L2 ball
    # (the README case: l2_norm_bound on A with a constant feasibility penalty) needs the r column sums of squares of
    # A + U all-reduced in every inner iteration; other matrix penalties are not supported.
    sharded_ball_A = sharded and any(r.kind == _engine.PEN_L2BALL for r in native[0])
    # ... any other matrix penalty (unimodality, total variation along the rows of A, a user's MatrixPenalty) is evaluated by
    # EVERY rank on the all-gathered matrix A + U (I x r floats: small) through the penalty's own prox, each rank keeping its
    # rows - the reference's arithmetic, replicated; row-separable kinds stay rank-local
    def _local_on_A(nat, reg):
        return nat.kind in (_engine.PEN_NN, _engine.PEN_BOX, _engine.PEN_L1) or (
            nat.kind == _engine.PEN_EXTERNAL and isinstance(reg, penalties.RowVectorPenalty))

    gathered_A = [sharded and nat.kind != _engine.PEN_L2BALL and not _local_on_A(nat, reg)
                  for nat, reg in zip(native[0], regs[0])]
    if (sharded_ball_A or any(gathered_A)) and not constant_A:
        raise NotImplementedError("a matrix penalty on mode 0 needs constant_feasibility_penalty (as in the reference, "
                                  "which has no row update for it)")

    # RCCL groups: the collectives go straight onto the engine's stream through a communicator of the engine's own
    # (_rccl.DirectComm: no stream hand-over); any other backend (gloo in the CPU tests), or a failed self-test: torch's
    direct = _direct_comm(group, X.device) if (sharded and sub is None) else None

    def all_reduce(t, op="sum"):
        if not sharded:
            return
        if (direct is not None and is_torch(t) and t.is_cuda and t.device == direct.device
                and t.dtype in (torch.float32, torch.float64) and t.is_contiguous()):
            direct.all_reduce(t, op)
        elif sharded:
            dist.all_reduce(t, op=(dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM), group=group)

    has_ext = [any(r.kind == _engine.PEN_EXTERNAL for r in native[m]) for m in range(3)]
    needs_B_steps = needs_B_steps or has_ext[1]
    row_slices = [slice(int(row_ptr[i]), int(row_ptr[i + 1])) for i in range(len(row_ptr) - 1)]

    def host_prox_B(k):
        """user prox of penalty k on mode 1 (decomposition.py:276-285), evaluated on device tensors"""
        reg, nat = regs[1][k], native[1][k]
        rhos = [float(v) for v in eng.rho(1).cpu().numpy()]
        shifted = [eng.B[sl] + nat.dual[sl] for sl in row_slices]
        obj = reg.factor_matrices_update(shifted, rhos, ext_aux[(1, k)])
        ext_aux[(1, k)] = obj
        nat.aux.copy_(torch.cat([m.to(nat.aux.dtype) for m in reg.auxes_as_matrices(obj)], 0))
        nat.dual.copy_(eng.B - (nat.aux - nat.dual))

    # host-evaluated penalties on mode 1 that act on ALL matrices at once (a user's MatricesPenalty) under group=: like PARAFAC2
    # they may couple the B_i of different ranks, so every rank evaluates the prox on the all-gathered matrices and keeps its own
    gathered_B = [sharded and nat.kind == _engine.PEN_EXTERNAL and not isinstance(reg, penalties.MatrixPenalty)
                  for reg, nat in zip(regs[1], native[1])]
    b_layout = []  # row counts of every rank's matrices, in rank order (gathered once)

    def gather_matrices_B(t):
        """all ranks' matrices of a packed mode-1 array [rows of this rank, r], in rank order, as a list"""
        if not b_layout:
            mine = torch.tensor([sl.stop - sl.start for sl in row_slices], dtype=torch.int64, device=t.device)
            n_loc = torch.tensor([len(row_slices)], dtype=torch.int64, device=t.device)
            cnt = [torch.zeros_like(n_loc) for _ in range(world)]
            dist.all_gather(cnt, n_loc, group=group)
            cnt = [int(c.item()) for c in cnt]
            padded = torch.zeros((max(cnt),), dtype=torch.int64, device=t.device)
            padded[: len(row_slices)] = mine
            parts = [torch.empty_like(padded) for _ in range(world)]
            dist.all_gather(parts, padded, group=group)
            b_layout.extend([int(v) for v in p[:c].cpu().numpy()] for p, c in zip(parts, cnt))
        rows = [sum(js) for js in b_layout]
        padded = torch.zeros((max(rows), t.shape[1]), dtype=t.dtype, device=t.device)
        padded[: t.shape[0]] = t
        parts = [torch.empty_like(padded) for _ in range(world)]
        dist.all_gather(parts, padded, group=group)
        out = []
        for p, js in zip(parts, b_layout):
            o = 0
            for j in js:
                out.append(p[o:o + j])
                o += j
        return out

    def sharded_gathered_prox_B(k):
        """a MatricesPenalty on the sharded B_i (decomposition.py:276-285): its prox on ALL matrices, this rank's kept"""
        reg, nat = regs[1][k], native[1][k]
        obj = ext_aux[(1, k)]
        # the object the penalty keeps must be the list of matrices itself to be split over ranks by matrix
        if not (isinstance(obj, (list, tuple)) and len(obj) == len(row_slices)
                and all(is_torch(o) and tuple(o.shape) == (sl.stop - sl.start, nat.aux.shape[1]) for o, sl in zip(obj, row_slices))):
            raise NotImplementedError("a host-evaluated MatricesPenalty on mode 1 whose auxiliary variable is not the list of "
                                      "matrices itself is not supported with group=")
        shifted = gather_matrices_B(eng.B + nat.dual)
        auxes = gather_matrices_B(torch.cat([o.to(nat.aux.dtype) for o in obj], 0))
        # the feasibility penalties of all matrices (one per matrix, fp64), in the same order
        cnt = [len(js) for js in b_layout]
        padded = torch.zeros((max(cnt),), dtype=torch.float64, device=nat.aux.device)
        padded[: len(row_slices)] = eng.rho(1).to(device=nat.aux.device, dtype=torch.float64)
        parts = [torch.empty_like(padded) for _ in range(world)]
        dist.all_gather(parts, padded, group=group)
        rho_all = [float(v) for p, c in zip(parts, cnt) for v in p[:c].cpu().numpy()]
        full = reg.factor_matrices_update(shifted, rho_all, auxes)
        lo = sum(len(js) for js in b_layout[:rank_id])
        own = [m.clone() for m in full[lo:lo + len(row_slices)]]
        ext_aux[(1, k)] = own
        nat.aux.copy_(torch.cat([m.to(nat.aux.dtype) for m in reg.auxes_as_matrices(own)], 0))
        nat.dual.copy_(eng.B - (nat.aux - nat.dual))

    def host_prox_matrix(mode, k, F, rho_rows, constant):
        """user prox of penalty k on mode 0 / 2 (decomposition.py:197-213 / 333-338)"""
        reg, nat = regs[mode][k], native[mode][k]
        shifted = F + nat.dual
        obj = ext_aux.get((mode, k))
        if obj is None:  # native kind evaluated through its host method inside a host-driven mode
            obj = nat.aux.clone()
        if constant:
            obj = reg.factor_matrix_update(shifted, float(rho_rows[0]), obj)
        else:
            obj = obj.clone() if is_torch(obj) else obj
            for i in range(F.shape[0]):
                obj[i] = reg.factor_matrix_row_update(shifted[i], float(rho_rows[i]), obj[i])
        if (mode, k) in ext_aux:
            ext_aux[(mode, k)] = obj
        nat.aux.copy_(reg.aux_as_matrix(obj).to(nat.aux.dtype))
        nat.dual.copy_(F - (nat.aux - nat.dual))

    def inner_converged(F, F_old, mode):
        """decomposition.py:90-117 on device tensors: relative change and every feasibility gap of the mode below inner_tol"""
        if not check_inner:  # (host-driven modes - host-evaluated penalties - run the test here even when the engine has it too)
            return False
        # squared norms in fp64: [ |F|^2, |F - F_old|^2, |F - aux_k|^2 ... ].  With group= the rows of modes 0 and 1 live on
        # different ranks: the sums are all-reduced, so that every rank leaves the inner loop at the same iteration and the
        # test is the reference's (over ALL B_i); C is replicated and needs nothing
        F64 = F.double()
        sq = torch.stack([(F64 ** 2).sum(), ((F64 - F_old.double()) ** 2).sum()]
                         + [((F64 - nat.aux.double()) ** 2).sum() for nat in native[mode]])
        if sharded and mode != 2:
            all_reduce(sq)
        sq = sq.cpu().numpy()
        nrm = float(np.sqrt(sq[0]))
        if float(np.sqrt(sq[1])) > inner_tol * nrm:
            return False
        if not native[mode]:
            return True
        return max(float(np.sqrt(v)) / nrm for v in sq[2:]) < inner_tol

    def do_update_B():
        if not needs_B_steps:
            eng.update_B()
            return
        eng.B_begin()
        if constant_B:
            all_reduce(eng.B_rho_max(), "max")
        eng.B_factor()
        n_it = inner_n_iter_max if native[1] else min(1, inner_n_iter_max)
        for _ in range(n_it):
            B_old = eng.B.clone() if check_inner else None
            eng.B_solve()
            for k, reg in enumerate(native[1]):
                if reg.kind == _engine.PEN_EXTERNAL:
                    # a MatrixPenalty's prox acts on one matrix at a time: every rank evaluates it on its own matrices
                    if gathered_B[k]:
                        sharded_gathered_prox_B(k)
                    else:
                        host_prox_B(k)
                    continue
                eng.B_prox_local(k)
                if reg.kind == _engine.PEN_PARAFAC2:
                    all_reduce(eng.B_prox_reduce_buffer(k))
                eng.B_prox_finish(k)
            if check_inner:
                eng.B_end()  # the convergence test reads B / aux on the host side
                if inner_converged(eng.B, B_old, 1):
                    break
        eng.B_end()

    def do_update_C():
        gr = eng.update_C_local()
        all_reduce(gr)
        if not has_ext[2]:
            eng.update_C_finish()
            return
        eng.C_begin()
        n_it = inner_n_iter_max if native[2] else min(1, inner_n_iter_max)
        for _ in range(n_it):
            C_old = eng.C.clone() if check_inner else None
            eng.C_solve()
            rho_c = eng.rho(2).cpu().numpy()
            for k in range(len(native[2])):
                host_prox_matrix(2, k, eng.C, rho_c, True)
            if inner_converged(eng.C, C_old, 2):
                break
        eng.C_end()

    def sharded_ball_prox_A(k, rho_a):
        """L2-ball prox on the sharded A (penalties.py:920-925): column norms over ALL rows = all-reduced partial sums"""
        nat = native[0][k]
        y = eng.A + nat.dual
        if nat.non_negativity:
            y = torch.clamp(y, min=0)
        sq = (y.double() ** 2).sum(0)
        all_reduce(sq)
        nrm = torch.sqrt(sq).to(y.dtype)
        bound = torch.as_tensor(nat.p0, dtype=y.dtype, device=y.device)
        z = y * (bound / torch.maximum(nrm, bound))
        nat.dual.copy_(eng.A - (z - nat.dual))
        nat.aux.copy_(z)

    a_counts = []

    def gather_rows_A(t):
        """all ranks' rows of a mode-0 matrix, in rank order (uneven shares: padded for the collective, trimmed after)"""
        if not a_counts:
            n_loc = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
            cnt = [torch.zeros_like(n_loc) for _ in range(world)]
            dist.all_gather(cnt, n_loc, group=group)
            a_counts.extend(int(c.item()) for c in cnt)
        padded = torch.zeros((max(a_counts), t.shape[1]), dtype=t.dtype, device=t.device)
        padded[: t.shape[0]] = t
        parts = [torch.empty_like(padded) for _ in range(world)]
        dist.all_gather(parts, padded, group=group)
        return torch.cat([p[:c] for p, c in zip(parts, a_counts)], 0)

    def sharded_gathered_prox_A(k, rho_a):
        """matrix penalty on the sharded A (decomposition.py:203): the penalty's prox on the gathered A + U, own rows kept"""
        reg, nat = regs[0][k], native[0][k]
        obj = ext_aux.get((0, k))
        if obj is None:
            aux_rows = nat.aux  # native kind evaluated through its host method: its aux IS the matrix
        else:
            # host-evaluated penalty (a user's MatrixPenalty; any matrix penalty while inner_tol is set): the object it keeps
            # must be the matrix itself to be gathered by rows - another parametrisation cannot be split over ranks
            aux_rows = reg.aux_as_matrix(obj)
            if not (is_torch(obj) and is_torch(aux_rows) and tuple(obj.shape) == tuple(nat.aux.shape)
                    and tuple(aux_rows.shape) == tuple(nat.aux.shape)):
                raise NotImplementedError("a host-evaluated matrix penalty on mode 0 whose auxiliary variable is not the matrix "
                                          "itself (aux_as_matrix is not the identity) is not supported with group=")
        full = reg.factor_matrix_update(gather_rows_A(eng.A + nat.dual), float(rho_a[0]), gather_rows_A(aux_rows))
        lo = sum(a_counts[:rank_id])
        own = full[lo:lo + a_counts[rank_id]] if obj is not None else None
        z = reg.aux_as_matrix(full)[lo:lo + a_counts[rank_id]].to(nat.aux.dtype)
        if obj is not None:
            ext_aux[(0, k)] = own.clone()  # this rank's rows of the object the penalty returned (what return_admm_vars hands out)
        nat.dual.copy_(eng.A - (z - nat.dual))
        nat.aux.copy_(z)

    def do_update_A():
        if sharded_ball_A or any(gathered_A):
            eng.A_begin()
            all_reduce(eng.A_rho_max(), "max")
            eng.A_factor()
            rho_a = eng.rho(0).cpu().numpy()
            n_it = inner_n_iter_max if native[0] else min(1, inner_n_iter_max)
            for _ in range(n_it):
                A_old = eng.A.clone() if check_inner else None
                eng.A_solve()
                for k, nat in enumerate(native[0]):
                    if nat.kind == _engine.PEN_L2BALL:
                        sharded_ball_prox_A(k, rho_a)
                    elif gathered_A[k]:
                        sharded_gathered_prox_A(k, rho_a)
                    else:
                        host_prox_matrix(0, k, eng.A, rho_a, True)
                if inner_converged(eng.A, A_old, 0):
                    break
            eng.A_end()
            return
        if has_ext[0]:
            eng.A_begin()
            if constant_A:
                all_reduce(eng.A_rho_max(), "max")
            eng.A_factor()
            rho_a = eng.rho(0).cpu().numpy()
            n_it = inner_n_iter_max if native[0] else min(1, inner_n_iter_max)
            for _ in range(n_it):
                A_old = eng.A.clone() if check_inner else None
                eng.A_solve()
                for k in range(len(native[0])):
                    host_prox_matrix(0, k, eng.A, rho_a, constant_A)
                if inner_converged(eng.A, A_old, 0):
                    break
            eng.A_end()
            return
        if not needs_A_steps:
            eng.update_A()
            return
        eng.A_begin()
        all_reduce(eng.A_rho_max(), "max")
        eng.A_finish()

    def read_diag(vec):
        """MCL_DIAG vector (already all-reduced) -> (rec_error, (A_gaps, B_gaps, C_gaps), reg_penalty + l2)"""
        d = vec.detach().cpu().numpy() if is_torch(vec) else np.asarray(vec)
        xsq, inner, model = d[_engine.DIAG_X_SQ], d[_engine.DIAG_INNER], d[_engine.DIAG_MODEL_SQ]
        norm_matrices = np.sqrt(xsq)
        rec_error = np.sqrt(max(0.0, xsq - 2 * inner + model)) / norm_matrices
        gaps, reg_penalty = [], 0.0
        for mode in range(3):
            fnorm = np.sqrt(d[_engine.DIAG_NORM_SQ + mode])
            mode_gaps = []
            for k, reg in enumerate(regs[mode]):
                base = _engine.DIAG_REG + (mode * _engine.MCL_MAX_REGS + k) * 2
                mode_gaps.append(np.sqrt(d[base]) / fnorm)
                if isinstance(reg, penalties.L1Penalty):
                    reg_penalty += reg.reg_strength * d[base + 1]
                elif sharded and mode == 1 and native[mode][k].kind in (_engine.PEN_EXTERNAL, _engine.PEN_TV, _engine.PEN_GL2):
                    reg_penalty += d[base + 1]  # this rank's matrices only: summed over the ranks with the vector (diagnostics())
                elif native[mode][k].kind == _engine.PEN_GL2:
                    reg_penalty += float(eng.penalty_value(mode, k))  # trace(F^T M F), evaluated by the engine
                elif native[mode][k].kind in (_engine.PEN_EXTERNAL, _engine.PEN_TV) or (mode == 0 and sharded and gathered_A[k]):
                    # value computed on device tensors; a sharded A is gathered first (every rank adds the same, whole value)
                    factor = [eng.B[sl] for sl in row_slices] if mode == 1 else (
                        (gather_rows_A(eng.A) if sharded else eng.A) if mode == 0 else eng.C)
                    reg_penalty += float(reg.penalty(factor))
            gaps.append(mode_gaps)
            if l2_penalty[mode]:
                reg_penalty += 0.5 * l2_penalty[mode] * d[_engine.DIAG_NORM_SQ + mode]
        return rec_error, tuple(gaps), reg_penalty

    def read_diag_rows(rows):
        """read_diag for a whole ring of iterations at once (the same elementwise arithmetic in the same order, vectorised:
        the per-iteration Python of read_diag costs as much as 7 % of a config-3 iteration): (rec_errors, gaps per iteration,
        reg_penalties).  Only for stacks whose penalty values come out of the diagnostics vector (no host-evaluated value)."""
        d = np.asarray(rows, dtype=np.float64).reshape(-1, _engine.DIAG_LEN)
        xsq, inner, model = d[:, _engine.DIAG_X_SQ], d[:, _engine.DIAG_INNER], d[:, _engine.DIAG_MODEL_SQ]
        rec = np.sqrt(np.maximum(0.0, xsq - 2 * inner + model)) / np.sqrt(xsq)
        reg_pen = np.zeros(len(d))
        per_mode = []
        for mode in range(3):
            fnorm = np.sqrt(d[:, _engine.DIAG_NORM_SQ + mode])
            cols = []
            for k, reg in enumerate(regs[mode]):
                base = _engine.DIAG_REG + (mode * _engine.MCL_MAX_REGS + k) * 2
                cols.append(np.sqrt(d[:, base]) / fnorm)
                if isinstance(reg, penalties.L1Penalty):
                    reg_pen = reg_pen + reg.reg_strength * d[:, base + 1]
                elif native[mode][k].kind in (_engine.PEN_EXTERNAL, _engine.PEN_TV, _engine.PEN_GL2):
                    raise AssertionError("read_diag_rows: host-evaluated penalty value")
            per_mode.append(np.stack(cols, axis=1) if cols else np.zeros((len(d), 0)))
            if l2_penalty[mode]:
                reg_pen = reg_pen + 0.5 * l2_penalty[mode] * d[:, _engine.DIAG_NORM_SQ + mode]
        gaps = [tuple(list(per_mode[m][i]) for m in range(3)) for i in range(len(d))]
        return rec, gaps, reg_pen

    def diagnostics():
        vec = eng.diagnostics(include_replicated=(rank_id == 0))
        if sharded:
            # penalties on the B_i whose value is summed on the host (total variation, host-evaluated MatrixPenalty classes):
            # the value over THIS rank's matrices travels in the penalty-value slot of the vector and is summed over the
            # ranks with it
            for k, reg in enumerate(regs[1]):
                if isinstance(reg, penalties.L1Penalty):
                    continue  # read_diag takes reg_strength * sum|B| from the native slot, which is summed over the ranks already
                if native[1][k].kind == _engine.PEN_GL2:
                    vec[_engine.DIAG_REG + (_engine.MCL_MAX_REGS + k) * 2 + 1] = float(eng.penalty_value(1, k))
                elif gathered_B[k]:  # a value over ALL matrices (need not be a sum over them): counted once, by rank 0
                    value = float(reg.penalty(gather_matrices_B(eng.B)))
                    vec[_engine.DIAG_REG + (_engine.MCL_MAX_REGS + k) * 2 + 1] = value if rank_id == 0 else 0.0
                elif native[1][k].kind in (_engine.PEN_EXTERNAL, _engine.PEN_TV):
                    vec[_engine.DIAG_REG + (_engine.MCL_MAX_REGS + k) * 2 + 1] = float(reg.penalty([eng.B[sl] for sl in row_slices]))
        all_reduce(vec)
        return read_diag(vec)

    # ---- arithmetic="auto" above the small-problem limit: decided by CONDITIONING, not by size alone ------------------------
    # A mode without any penalty solves un-shifted normal equations (the reference: an fp64 SVD, decomposition.py:172, 252-256,
    # 319-321) and multiplies whatever the fp32 kernels left in its inputs (1e-8 .. 4e-7 relative) by the condition number of
    # its system.  On large problems those roundings average out over 1e5 .. 1e7 rows (BASELINE config 4: C to 8e-8 at
    # condition 600) and the exact arithmetic would cost passes over X; below _AUTO_EXACT_MAX_ELEMENTS elements neither
    # holds.  The condition numbers that matter are those AT THE START OF EACH PHASE (Gauss-Seidel: the A-phase of an
    # iteration solves systems built from the B_i and C of the same iteration - a random start has kappa ~30 where the first
    # A-phase meets 3e4), so they are measured there: a TRIAL of _AUTO_EXACT_TRIAL_ITERATIONS iterations runs under the engine's
    # condition monitor (mcl_condition_monitor: per phase of a penalty-free mode, kappa = ||M||_F ||M^-1||_F of its system, from
    # the factors, no pass over X), the initial state is restored, and with a kappa above _AUTO_EXACT_KAPPA the run takes the
    # exact arithmetic (fp64 sums of exact products, fp64 inner loops) from its first iteration - the result is that of one
    # arithmetic from start to end.  Long runs look again every _AUTO_EXACT_PROBE_EVERY iterations (one monitored iteration) and
    # switch forward.  Under `group=` the maxima are all-reduced (every rank switches, or none).  Larger problems keep the fast
    # kernels; a badly conditioned one is told about `arithmetic="exact"`.
    updated_modes = (update_A, update_B_is, update_C)
    free_modes = [m for m in range(3) if updated_modes[m] and len(regs[m]) == 0]
    all_native = not any(r.kind == _engine.PEN_EXTERNAL for m in range(3) for r in native[m])
    has_pf2 = update_B_is and any(r_.kind == _engine.PEN_PARAFAC2 for r_ in native[1]) and rank <= 32
    auto_candidate = (arithmetic == "auto" and (bool(free_modes) or (has_pf2 and n_el_total <= _AUTO_EXACT_MAX_ELEMENTS))
                      and n_el_total > float(1 << 20) and n_iter_max > 0 and all_native and hasattr(eng, "condition_monitor"))

    def plain_iterations(n):
        """n outer iterations without diagnostics, on whichever path this run takes"""
        if sharded:
            for _ in range(n):
                if update_B_is:
                    do_update_B()
                if update_C:
                    do_update_C()
                if update_A:
                    do_update_A()
        else:
            eng.iterate(n, update_A=update_A, update_B=update_B_is, update_C=update_C)

    def monitored_kappa(n):
        """worst kappa the penalty-free phases of the next n iterations meet (all ranks: the same number); the iterations RUN"""
        mon = eng.condition_monitor(True, update_A, update_B_is, update_C)
        try:
            plain_iterations(n)
        finally:
            eng.condition_monitor(False)
        if sharded:
            all_reduce(mon, "max")
        return kappa_of(mon)

    def kappa_of(mon):
        """monitor vector -> the number the rule looks at: the worst kappa of a penalty-free mode's system, or - on the same
        footing, _AUTO_EXACT_KAPPA / _AUTO_EXACT_POLAR apart - the worst conditioning of a PARAFAC2 polar factor (the Gram route of
        the fast kernels squares it: a factor of condition 1e5 sits at the edge of what fp32 statistics resolve)"""
        m = mon.cpu().numpy()
        return max(float(m[:3].max()), float(m[3]) * (_AUTO_EXACT_KAPPA / _AUTO_EXACT_POLAR))

    def decide_arithmetic(worst):
        nonlocal auto_candidate
        if n_el_total > _AUTO_EXACT_MAX_ELEMENTS:
            auto_candidate = False  # (large problems are looked at once, for the warning only)
            if worst > _AUTO_EXACT_WARN_KAPPA:
                import warnings

                warnings.warn(
                    f"cmf_aoadmm: a mode without penalties has normal equations of condition ~{worst:.1e}; the fp32 kernels this "
                    f"problem size takes by default carry about 1e-8 x that in the factors. Pass arithmetic=\"exact\" for the "
                    "reference's fp64 solve (slower: fp64 passes over the matrices).", RuntimeWarning, stacklevel=4)
        elif worst > _AUTO_EXACT_KAPPA:
            eng.set_exact(True)
            auto_candidate = False
            if verbose:
                print(f"matcouply_amd: penalty-free mode with condition ~{worst:.1e}: the exact arithmetic from here on")

    if auto_candidate:
        state = [eng.A, eng.B, eng.C] + [t for m in range(3) for r_ in native[m] for t in (r_.aux, r_.dual, r_.aux2) if t is not None]
        saved = [t.clone() for t in state]
        worst = monitored_kappa(min(_AUTO_EXACT_TRIAL_ITERATIONS, n_iter_max))
        for t, t0 in zip(state, saved):
            t.copy_(t0)
        del saved
        eng.invalidate()  # the factors changed behind the engine's back
        decide_arithmetic(worst)

    rec_errors, feasibility_gaps, losses = [], [], []
    rec_error, gaps0, reg0 = diagnostics()
    rec_errors.append(rec_error)
    losses.append(0.5 * rec_error ** 2 + reg0)
    feasibility_gaps.append(gaps0)
    progress = _Progress(verbose)
    progress.initial(gaps0)
    stop = _StopRule(tol, absolute_tol, feasibility_tol)

    satisfied_stopping_condition = False
    message = _StopRule.EXHAUSTED
    feasibility_criterion = None

    it = -1  # Needed if n_iter_max <= 0
    # penalty values that need a host call per iteration (the device-resident loops below do not apply)
    host_value = any(r.kind in (_engine.PEN_TV, _engine.PEN_EXTERNAL, _engine.PEN_GL2) for m in range(3) for r in native[m]) or any(gathered_A)
    fast_path = ((not (tol or absolute_tol)) and not sharded and not verbose and n_iter_max > 0 and not any(has_ext)
                 and not host_value)
    lazy_diag = (not (tol or absolute_tol)) and sharded and not verbose and n_iter_max > 0 and not host_value
    # stopping rule on the device (mcl_run): single device, every penalty native, silent.  (tol set with absolute_tol=None is
    # a TypeError in the reference's comparison - the host loop below raises it the same way.)
    device_stop = (bool(tol or absolute_tol) and not sharded and not verbose and n_iter_max > 0 and not any(has_ext)
                   and not host_value and not (tol and absolute_tol is None) and hasattr(eng, "run"))
    # ... and the same rule under sharding (mcl_gate_begin / mcl_verdict): the phases are stepped with their reductions, the
    # diagnostics vector is all-reduced and every rank evaluates the rule on the same bits.  Every rank must enqueue the same
    # number of iterations (their collectives pair up), so the loop runs in fixed chunks with one synchronisation each.
    sharded_stop = (bool(tol or absolute_tol) and sharded and not verbose and n_iter_max > 0 and not any(has_ext)
                    and not host_value and not sharded_ball_A and not (tol and absolute_tol is None)
                    and hasattr(eng, "gate_begin"))
    final_gaps_known = False
    if lazy_diag:
        # sharded, fixed iteration count: nothing depends on the diagnostics inside the loop, so their partial sums stay
        # on the device and are all-reduced once for all iterations (one collective per iteration remains: [G | R])
        ring = torch.zeros((n_iter_max, _engine.DIAG_LEN), dtype=torch.float64, device=device) if return_errors else None
        for it in range(n_iter_max):
            mon = (eng.condition_monitor(True, update_A, update_B_is, update_C)
                   if (auto_candidate and it > 0 and it % _AUTO_EXACT_PROBE_EVERY == 0) else None)
            if update_B_is:
                do_update_B()
            if update_C:
                do_update_C()
            if update_A:
                do_update_A()
            if ring is not None:  # the table reduction rides on the next iteration's C-phase reduction kernel
                eng.diagnostics_deferred(include_replicated=(rank_id == 0), out=ring[it])
            if mon is not None:
                eng.condition_monitor(False)
                all_reduce(mon, "max")
                decide_arithmetic(kappa_of(mon))
        if ring is not None:
            eng.flush_diagnostics()
            all_reduce(ring)
            rec, gaps, reg = read_diag_rows(ring.cpu().numpy())
            feasibility_gaps.extend(gaps)
            rec_errors.extend(rec)
            losses.extend(0.5 * rec ** 2 + reg)
    elif sharded_stop:
        weights = [[(reg.reg_strength if isinstance(reg, penalties.L1Penalty) else 0.0) for reg in regs[m]] for m in range(3)]
        eng.gate_begin(tol, absolute_tol, feasibility_tol, initial_loss=losses[-1], penalty_weight=weights,
                       evaluate_loss_always=return_errors)
        gate_closed = False
        try:
            chunk, done, code, stop_it = 8, 0, 0, -1
            ring = torch.zeros((n_iter_max if n_iter_max <= 4096 else 4096, _engine.DIAG_LEN), dtype=torch.float64, device=device)
            verdict = torch.zeros((ring.shape[0], 4), dtype=torch.float64, device=device)
            while done < n_iter_max and not code:
                n_now = min(chunk, n_iter_max - done)
                base = done % ring.shape[0]
                if base + n_now > ring.shape[0]:
                    base = 0
                for j in range(n_now):
                    if update_B_is:
                        do_update_B()
                    if update_C:
                        do_update_C()
                    if update_A:
                        do_update_A()
                    eng.diagnostics(include_replicated=(rank_id == 0), out=ring[base + j])
                    all_reduce(ring[base + j])
                    eng.verdict(ring[base + j], done + j, verdict[base + j])
                if is_torch(ring) and ring.is_cuda:
                    torch.cuda.synchronize(device)
                stopped, stop_it, code = eng.gate_status()
                n_ran = (stop_it - done + 1) if stopped else n_now
                ring_h, verdict_h = ring[base:base + n_ran].cpu().numpy(), verdict[base:base + n_ran].cpu().numpy()
                feasibility_gaps.extend(read_diag_rows(ring_h)[1])
                if len(verdict_h):
                    flags = verdict_h[:, 3].astype(np.int64)
                    feasibility_criterion = bool(flags[-1] & _engine.VERDICT_FEASIBLE) if feasibility_tol else feasibility_tol
                    evaluated = (flags & _engine.VERDICT_LOSS_EVALUATED) != 0
                    rec_errors.extend(verdict_h[evaluated, 0].tolist())
                    losses.extend(verdict_h[evaluated, 1].tolist())
                done += n_ran
                if not stopped:
                    code = 0
            eng.gate_end(bool(code))
            gate_closed = True
        finally:
            if not gate_closed:  # an exception inside the loop (a failed collective, a NotImplementedError of a step): never
                eng.gate_end(True)  # leave the context gated - its kernels would silently do nothing from then on
        it = done - 1
        if code:
            satisfied_stopping_condition = True
            message = _StopRule.RELATIVE if code == _engine.STOP_RELATIVE else _StopRule.ABSOLUTE
        final_gaps_known = True
    elif device_stop:
        # a stopping rule is active (the DEFAULT call: tol=1e-8, absolute_tol=1e-10, feasibility_tol=1e-4): the rule is
        # evaluated by a kernel at the end of every iteration (mcl_run), the host enqueues ahead of the verdicts and never
        # blocks on one; state-writing kernels behind a stopping iteration see the device-side flag and do nothing, so
        # the factors returned are exactly those of the stopping iteration.  Chunked so that the rings stay small.
        weights = [[(reg.reg_strength if isinstance(reg, penalties.L1Penalty) else 0.0) for reg in regs[m]] for m in range(3)]
        done, code, chunk = 0, 0, 4096

        def run_chunk(n_now):
            nonlocal done, feasibility_criterion
            n_ran, code_, ring_h, verdict_h = eng.run(
                n_now, tol, absolute_tol, feasibility_tol, initial_loss=losses[-1], penalty_weight=weights,
                evaluate_loss_always=return_errors, update_A=update_A, update_B=update_B_is, update_C=update_C)
            feasibility_gaps.extend(read_diag_rows(ring_h)[1])
            if len(verdict_h):
                flags = verdict_h[:, 3].astype(np.int64)
                feasibility_criterion = bool(flags[-1] & _engine.VERDICT_FEASIBLE) if feasibility_tol else feasibility_tol
                evaluated = (flags & _engine.VERDICT_LOSS_EVALUATED) != 0  # not on infeasible iterates unless errors are recorded (Q10)
                rec_errors.extend(verdict_h[evaluated, 0].tolist())
                losses.extend(verdict_h[evaluated, 1].tolist())
            done += n_ran
            return code_

        while done < n_iter_max and not code:
            n_now = min(_AUTO_EXACT_PROBE_EVERY if auto_candidate else chunk, n_iter_max - done)
            if auto_candidate and done > 0:  # the first iteration of every further chunk runs under the condition monitor
                mon = eng.condition_monitor(True, update_A, update_B_is, update_C)
                code = run_chunk(1)
                eng.condition_monitor(False)
                decide_arithmetic(kappa_of(mon))
                n_now -= 1
                if code or n_now == 0:
                    continue
            code = run_chunk(n_now)
        it = done - 1
        if code:
            satisfied_stopping_condition = True
            message = _StopRule.RELATIVE if code == _engine.STOP_RELATIVE else _StopRule.ABSOLUTE
        final_gaps_known = True
    elif fast_path:
        # fixed iteration count: the whole outer loop runs natively, diagnostics stay on the device until the end
        ring = None
        if return_errors:
            ring = torch.zeros((n_iter_max, _engine.DIAG_LEN), dtype=torch.float64, device=device)
        done = 0
        while done < n_iter_max:  # (one call, unless the conditioning of a penalty-free mode is being watched)
            n_now = min(_AUTO_EXACT_PROBE_EVERY, n_iter_max - done) if auto_candidate else n_iter_max - done
            mon = eng.condition_monitor(True, update_A, update_B_is, update_C) if (auto_candidate and done > 0) else None
            eng.iterate(1 if mon is not None else n_now, update_A=update_A, update_B=update_B_is, update_C=update_C,
                        diag_ring=(ring[done:] if ring is not None else None))
            if mon is not None:  # the first iteration of every further chunk ran under the monitor
                eng.condition_monitor(False)
                decide_arithmetic(kappa_of(mon))
                if n_now > 1:
                    eng.iterate(n_now - 1, update_A=update_A, update_B=update_B_is, update_C=update_C,
                                diag_ring=(ring[done + 1:] if ring is not None else None))
            done += n_now
        it = n_iter_max - 1
        if return_errors:
            rec, gaps, reg = read_diag_rows(ring.cpu().numpy())
            feasibility_gaps.extend(gaps)
            rec_errors.extend(rec)
            losses.extend(0.5 * rec ** 2 + reg)
    else:
        for it in range(n_iter_max):
            mon = (eng.condition_monitor(True, update_A, update_B_is, update_C)
                   if (auto_candidate and it > 0 and it % _AUTO_EXACT_PROBE_EVERY == 0) else None)
            if update_B_is:
                do_update_B()
            if update_C:
                do_update_C()
            if update_A:
                do_update_A()
            if mon is not None:
                eng.condition_monitor(False)
                all_reduce(mon, "max")
                decide_arithmetic(kappa_of(mon))

            if not (stop.active or return_errors):
                progress.iteration(it)
                continue
            rec_error, gaps, reg_pen = diagnostics()
            feasibility_gaps.append(gaps)
            if stop.active:
                feasibility_criterion = stop.feasible(gaps)
                if not feasibility_criterion and not return_errors:
                    progress.iteration(it, gaps=gaps)  # the loss is not evaluated on infeasible iterates (Q10)
                    continue
            rec_errors.append(rec_error)
            losses.append(0.5 * rec_error ** 2 + reg_pen)
            progress.iteration(it, rec_errors[-1], losses[-1], abs(losses[-2] - losses[-1]) / losses[-2], gaps)
            fired = stop.verdict(feasibility_criterion, losses)
            if fired is not None:
                satisfied_stopping_condition, message = True, fired
                progress.converged(it, message)
                break
        else:
            progress.exhausted()

    if feasibility_tol and return_errors and final_gaps_known:
        # the reference re-evaluates the gaps of the final state here: the state the last verdict was taken on
        feasibility_criterion = _check_feasibility(feasibility_gaps[-1], feasibility_tol)
    elif feasibility_tol and return_errors:
        _, final_gaps, _ = diagnostics()
        feasibility_criterion = _check_feasibility(final_gaps, feasibility_tol)
    elif not feasibility_tol:
        feasibility_criterion = None

    # ---- results back in the caller's array type ------------------------------------------------------------------
    cmf = CoupledMatrixFactorization((None, (out(eng.A), out.split(eng.B, row_ptr), out(eng.C))))
    if gather_A and group is not None:
        # the one all-gather of a sharded run: ranks hold different numbers of matrices, so the rows are
        # padded to the largest share for the collective and trimmed afterwards.  The factorization returned stays this
        # rank's (its rows of A, its B_i); the whole A rides along as `cmf.A_all`, this rank's rows being
        # `cmf.A_all[cmf.rows_of_rank[0]:cmf.rows_of_rank[1]]`.
        n_loc = torch.tensor([eng.A.shape[0]], dtype=torch.int64, device=eng.A.device)
        counts = [torch.zeros_like(n_loc) for _ in range(world)]
        dist.all_gather(counts, n_loc, group=group)
        counts = [int(c.item()) for c in counts]
        padded = torch.zeros((max(counts), eng.A.shape[1]), dtype=eng.A.dtype, device=eng.A.device)
        padded[: eng.A.shape[0]] = eng.A
        parts = [torch.empty_like(padded) for _ in range(world)]
        dist.all_gather(parts, padded, group=group)
        cmf.A_all = out(torch.cat([p[:c] for p, c in zip(parts, counts)], 0))
        lo = sum(counts[:rank_id])
        cmf.rows_of_rank = (lo, lo + counts[rank_id])
    result = [cmf]
    if return_admm_vars:
        auxes, duals = [[], [], []], [[], [], []]
        for mode in range(3):
            for k, reg in enumerate(native[mode]):
                if reg.kind == _engine.PEN_EXTERNAL:
                    auxes[mode].append(_aux_out(ext_aux[(mode, k)], out))
                    duals[mode].append(out.split(reg.dual, row_ptr) if mode == 1 else out(reg.dual))
                    continue
                if reg.kind == _engine.PEN_PARAFAC2:
                    auxes[mode].append((out.split(reg.aux, row_ptr), out(reg.aux2)))
                else:
                    auxes[mode].append(out.split(reg.aux, row_ptr) if mode == 1 else out(reg.aux))
                duals[mode].append(out.split(reg.dual, row_ptr) if mode == 1 else out(reg.dual))
        result.append(ADMMVars(auxes=tuple(auxes), duals=tuple(duals)))
    if return_errors:
        if not satisfied_stopping_condition and not (tol or absolute_tol):
            satisfied_stopping_condition = None
        result.append(DiagnosticMetrics(
            rec_errors=rec_errors, feasibility_gaps=feasibility_gaps, regularized_loss=losses,
            satisfied_stopping_condition=satisfied_stopping_condition,
            satisfied_feasibility_condition=feasibility_criterion, message=message, n_iter=it + 1))
    if _byproducts is not None and update_A:
        _byproducts["rhses"] = out(eng.rhses())
        _byproducts["cross_products"] = out(eng.cross_products())
    if hasattr(eng, "close"):
        eng.close()
    if len(result) == 1:
        return result[0]
    return tuple(result)


def _single_phase(mode, matrices, reg, cmf, aux_list, dual_list, l2_penalty, inner_n_iter_max, inner_tol,
                  feasibility_penalty_scale, constant_feasibility_penalty):
    """One call of one phase on the device, with the given ADMM variables as initial state."""
    import copy

    if len(reg) != len(aux_list) or len(reg) != len(dual_list):
        raise ValueError("reg, aux and dual lists must have the same length")
    stack = []
    for r_k, aux, dual in zip(reg, aux_list, dual_list):
        r_c = copy.copy(r_k)  # same penalty, initial state = the caller's variables
        r_c.aux_init, r_c.dual_init = aux, dual
        stack.append(r_c)
    weights, (A, B_is, C) = cmf
    regs = [[], [], []]
    regs[mode] = stack
    l2 = [0, 0, 0]
    l2[mode] = l2_penalty if l2_penalty else 0
    by = {}
    new_cmf, admm = cmf_aoadmm(
        matrices, A.shape[1], init=(None, (A, B_is, C)), regs=regs, l2_penalty=l2, n_iter_max=1, tol=None, absolute_tol=None,
        feasibility_tol=None, inner_tol=inner_tol, inner_n_iter_max=inner_n_iter_max,
        feasibility_penalty_scale=feasibility_penalty_scale,
        constant_feasibility_penalty=({0: "A", 1: "B"}.get(mode, False) if constant_feasibility_penalty else False),
        update_A=(mode == 0), update_B_is=(mode == 1), update_C=(mode == 2), return_admm_vars=True, _byproducts=by)
    out_cmf = (None, [new_cmf[1][0], new_cmf[1][1], new_cmf[1][2]])
    return out_cmf, list(admm.auxes[mode]), list(admm.duals[mode]), by


def admm_update_A(matrices, reg, cmf, A_aux_list, A_dual_list, l2_penalty, inner_n_iter_max, inner_tol,
                  feasibility_penalty_scale, constant_feasibility_penalty, svd_fun=None):
    """One A-phase (reference: decomposition.py:120-219, same positional arguments and return value
    ``(cmf, A_aux_list, A_dual_list, (rhses, cross_products))``) on the device.  `svd_fun` is accepted for compatibility:
    the r x r systems are symmetric positive definite and solved by Gauss-Jordan elimination in fp64 registers."""
    out_cmf, aux, dual, by = _single_phase(0, matrices, reg, cmf, A_aux_list, A_dual_list, l2_penalty, inner_n_iter_max,
                                           inner_tol, feasibility_penalty_scale, constant_feasibility_penalty)
    rhses, cross = by["rhses"], by["cross_products"]
    return out_cmf, aux, dual, ([rhses[i] for i in range(len(rhses))], [cross[i] for i in range(len(cross))])


def admm_update_B(matrices, reg, cmf, B_is_aux_list, B_is_dual_list, l2_penalty, inner_n_iter_max, inner_tol,
                  feasibility_penalty_scale, constant_feasibility_penalty, svd_fun=None):
    """One B-phase (reference: decomposition.py:222-292); returns ``(cmf, B_is_aux_list, B_is_dual_list)``."""
    out_cmf, aux, dual, _ = _single_phase(1, matrices, reg, cmf, B_is_aux_list, B_is_dual_list, l2_penalty,
                                          inner_n_iter_max, inner_tol, feasibility_penalty_scale,
                                          constant_feasibility_penalty)
    return out_cmf, aux, dual


def admm_update_C(matrices, reg, cmf, C_aux_list, C_dual_list, l2_penalty, inner_n_iter_max, inner_tol,
                  feasibility_penalty_scale, svd_fun=None):
    """One C-phase (reference: decomposition.py:295-344); returns ``(cmf, C_aux_list, C_dual_list)``."""
    out_cmf, aux, dual, _ = _single_phase(2, matrices, reg, cmf, C_aux_list, C_dual_list, l2_penalty, inner_n_iter_max,
                                          inner_tol, feasibility_penalty_scale, False)
    return out_cmf, aux, dual


def parafac2_aoadmm(
    matrices,
    rank,
    init="random",
    n_iter_max=1000,
    l2_penalty=0,
    tv_penalty=None,
    l1_penalty=None,
    non_negative=None,
    unimodal=None,
    generalized_l2_penalty=None,
    l2_norm_bound=None,
    lower_bound=None,
    upper_bound=None,
    regs=None,
    feasibility_penalty_scale=1,
    constant_feasibility_penalty=False,
    aux_init="random_uniform",
    dual_init="random_uniform",
    svd="truncated_svd",
    init_params=None,
    random_state=None,
    tol=1e-8,
    absolute_tol=1e-10,
    feasibility_tol=1e-4,
    inner_tol=None,
    inner_n_iter_max=5,
    update_A=True,
    update_B_is=True,
    update_C=True,
    return_errors=False,
    return_admm_vars=False,
    verbose=False,
    *,
    group=None,
    gather_A=False,
    arithmetic="auto",
):
    """Alias for cmf_aoadmm with the PARAFAC2 constraint on mode 1 (reference decomposition.py:1103-1179)."""
    return cmf_aoadmm(
        matrices=matrices, rank=rank, init=init, n_iter_max=n_iter_max, l2_penalty=l2_penalty, tv_penalty=tv_penalty,
        l1_penalty=l1_penalty, non_negative=non_negative, unimodal=unimodal,
        generalized_l2_penalty=generalized_l2_penalty, l2_norm_bound=l2_norm_bound, lower_bound=lower_bound,
        upper_bound=upper_bound, parafac2=True, regs=regs, feasibility_penalty_scale=feasibility_penalty_scale,
        constant_feasibility_penalty=constant_feasibility_penalty, aux_init=aux_init, dual_init=dual_init, svd=svd,
        init_params=init_params, random_state=random_state, tol=tol, absolute_tol=absolute_tol,
        feasibility_tol=feasibility_tol, inner_tol=inner_tol, inner_n_iter_max=inner_n_iter_max, update_A=update_A,
        update_B_is=update_B_is, update_C=update_C, return_errors=return_errors, return_admm_vars=return_admm_vars,
        verbose=verbose, group=group, gather_A=gather_A, arithmetic=arithmetic)
