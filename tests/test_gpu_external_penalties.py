"""-m gpu: penalties WITHOUT a native kernel (user subclasses of the plugin classes) are evaluated through their own
Python methods on device tensors between the library's solve steps (EXTERNAL path).  A user-written
copy of a built-in penalty must reproduce the native kernels' result."""
import numpy as np
import pytest

from tests.helpers import rel_err, split_rows

pytestmark = pytest.mark.gpu


def _problem():
    from oracle import aoadmm_oracle as orc

    J = np.array([33, 64, 7, 100, 18, 70])
    X, row_ptr = orc.synthetic_problem(len(J), J, 24, 5, seed=3, dtype=np.float64)
    rng = np.random.RandomState(4)
    I, N, K, r = len(J), X.shape[0], 24, 5
    state = dict(A=rng.uniform(size=(I, r)), B=rng.uniform(size=(N, r)), C=rng.uniform(size=(K, r)))
    for m, shp in ((0, (I, r)), (1, (N, r)), (2, (K, r))):
        state[f"aux{m}"], state[f"dual{m}"] = rng.uniform(size=shp), rng.uniform(size=shp)
    return X, row_ptr, r, state


def _run(make_reg, **kw):
    from matcouply_amd import decomposition as dec

    X, row_ptr, r, st = _problem()
    regs = []
    for m in range(3):
        aux = split_rows(st[f"aux{m}"], row_ptr) if m == 1 else st[f"aux{m}"].copy()
        dual = split_rows(st[f"dual{m}"], row_ptr) if m == 1 else st[f"dual{m}"].copy()
        regs.append([make_reg(m, aux, dual)])
    return dec.cmf_aoadmm(split_rows(X, row_ptr), r, init=(None, (st["A"].copy(), split_rows(st["B"], row_ptr), st["C"].copy())),
                          regs=regs, n_iter_max=4, tol=None, absolute_tol=None, return_errors=True, return_admm_vars=True, **kw)


def _close(a, b, tol=5e-6):
    (cmf_a, admm_a, diag_a), (cmf_b, admm_b, diag_b) = a, b
    assert rel_err(cmf_a[1][0], cmf_b[1][0]) < tol and rel_err(cmf_a[1][2], cmf_b[1][2]) < tol
    assert rel_err(np.concatenate(cmf_a[1][1]), np.concatenate(cmf_b[1][1])) < tol
    np.testing.assert_allclose(diag_a.rec_errors, diag_b.rec_errors, rtol=1e-5)
    np.testing.assert_allclose(diag_a.regularized_loss, diag_b.regularized_loss, rtol=1e-5)
    for m in range(3):
        za, zb = admm_a.auxes[m][0], admm_b.auxes[m][0]
        assert rel_err(np.concatenate(za) if m == 1 else za, np.concatenate(zb) if m == 1 else zb) < tol


def test_user_matrix_penalty_equals_native_nonnegativity():
    import torch
    from matcouply_amd import penalties as pen

    class UserNonNeg(pen.MatrixPenalty):  # no _native_descriptor -> host-evaluated
        def factor_matrix_update(self, factor_matrix, feasibility_penalty, aux):
            return torch.clamp(factor_matrix, min=0)

        def penalty(self, x):
            return 0

    for const in (True, "A"):
        native = _run(lambda m, a, d: pen.NonNegativity(aux_init=a, dual_init=d), constant_feasibility_penalty=const)
        user = _run(lambda m, a, d: UserNonNeg(aux_init=a, dual_init=d), constant_feasibility_penalty=const)
        _close(user, native)
    # like the reference, a matrix penalty on mode 0 without a constant feasibility penalty has no row update
    with pytest.raises(AttributeError):
        _run(lambda m, a, d: UserNonNeg(aux_init=a, dual_init=d), constant_feasibility_penalty="B")


def test_user_row_penalty_with_per_row_feasibility_penalties():
    import torch
    from matcouply_amd import penalties as pen

    class UserL1(pen.RowVectorPenalty):
        """soft thresholding written by a user; on mode 0 each row gets its own rho_i (decomposition.py:205-211)"""

        def __init__(self, strength, **kw):
            super().__init__(**kw)
            self.strength = strength

        def factor_matrix_row_update(self, row, feasibility_penalty, aux_row):
            return torch.sign(row) * torch.clamp(row.abs() - self.strength / feasibility_penalty, min=0)

        def factor_matrix_update(self, factor_matrix, feasibility_penalty, aux):
            return self.factor_matrix_row_update(factor_matrix, feasibility_penalty, aux)

        def penalty(self, x):
            if isinstance(x, list):
                return self.strength * sum(float(xi.abs().sum()) for xi in x)
            return self.strength * float(x.abs().sum())

    native = _run(lambda m, a, d: pen.L1Penalty(0.05, aux_init=a, dual_init=d))
    user = _run(lambda m, a, d: UserL1(0.05, aux_init=a, dual_init=d))
    _close(user, native)


def test_parafac2_variant_without_native_kernel_runs_on_the_host_path():
    """Parafac2(n_iter=2) has no fused native variant: the engine solves, the class's own sweep runs on device tensors."""
    from matcouply_amd import decomposition as dec
    from matcouply_amd import penalties as pen

    X, row_ptr, r, st = _problem()
    mats = split_rows(X, row_ptr)
    cmf, admm, diag = dec.cmf_aoadmm(mats, r, regs=[[], [pen.Parafac2(n_iter=2)], []], non_negative={0: True, 2: True},
                                     n_iter_max=15, tol=None, absolute_tol=None, return_errors=True, return_admm_vars=True,
                                     random_state=0)
    P_is, Delta = admm.auxes[1][0]
    for P in P_is:
        if P.shape[0] >= r:
            np.testing.assert_allclose(P.T @ P, np.eye(r), atol=1e-4)
    assert diag.rec_errors[-1] < diag.rec_errors[0] and np.isfinite(diag.regularized_loss).all()


def test_overridden_builtin_penalty_is_not_routed_to_the_native_kernel():
    """A subclass of NonNegativity whose prox clamps at 0.1 must behave like Box(0.1, None) - not like the built-in
    non-negativity kernel its parent class maps to (the reference calls the override, decomposition.py:278-285)."""
    import torch
    from matcouply_amd import penalties as pen

    class AtLeastTenth(pen.NonNegativity):
        def factor_matrix_row_update(self, factor_matrix_row, feasibility_penalty, aux_row):
            return torch.clamp(factor_matrix_row, min=0.1)

        def factor_matrix_update(self, factor_matrix, feasibility_penalty, aux):
            return torch.clamp(factor_matrix, min=0.1)

    user = _run(lambda m, a, d: AtLeastTenth(aux_init=a, dual_init=d))
    box = _run(lambda m, a, d: pen.Box(0.1, None, aux_init=a, dual_init=d))
    plain = _run(lambda m, a, d: pen.NonNegativity(aux_init=a, dual_init=d))
    _close(user, box)
    assert min(float(np.min(z)) for z in user[1].auxes[1][0]) >= 0.1 - 1e-7
    assert rel_err(user[0][1][2], plain[0][1][2]) > 1e-3  # and it is NOT the parent's kernel


def test_custom_penalty_written_against_the_reference_docs():
    """The pattern of the reference's custom-penalty example (examples/plot_custom_penalty.py:213-231, re-typed): a hard
    constraint built from HardConstraintMixin + MatrixPenalty with the docstring helper, imposing unimodality on all but
    the last component, on the B mode of a PARAFAC2 model - through the EXTERNAL step path of the engine."""
    import torch
    from matcouply_amd import decomposition as dec
    from matcouply_amd import penalties as pen
    from matcouply_amd._doc_utils import copy_ancestor_docstring
    from matcouply_amd._unimodal_regression import unimodal_regression
    from matcouply_amd.data import get_simple_simulated_data

    class UnimodalButLast(pen.HardConstraintMixin, pen.MatrixPenalty):
        def __init__(self, non_negativity=False, aux_init="random_uniform", dual_init="random_uniform"):
            super().__init__(aux_init, dual_init)
            self.non_negativity = non_negativity

        @copy_ancestor_docstring
        def factor_matrix_update(self, factor_matrix, feasibility_penalty, aux):
            host = factor_matrix.detach().cpu().numpy().astype(np.float64)
            host[:, :-1] = unimodal_regression(host[:, :-1], non_negativity=self.non_negativity)
            if self.non_negativity:
                host = np.clip(host, 0, None)
            return torch.as_tensor(host, dtype=factor_matrix.dtype, device=factor_matrix.device)

    assert UnimodalButLast.factor_matrix_update.__doc__ == pen.MatrixPenalty.factor_matrix_update.__doc__
    X, _ = get_simple_simulated_data(noise_level=0.2, random_state=1)
    cmf, admm, diag = dec.parafac2_aoadmm(X, 3, n_iter_max=30, non_negative={0: True, 2: True},
                                         regs=[[], [UnimodalButLast(non_negativity=True)], []], random_state=0,
                                         tol=None, absolute_tol=None, return_errors=True, return_admm_vars=True)
    assert np.isfinite(diag.regularized_loss).all() and diag.rec_errors[-1] < 0.5 * diag.rec_errors[0]
    k = [i for i, z in enumerate(admm.auxes[1]) if not isinstance(z, tuple)][0]  # the custom penalty (PARAFAC2 is a tuple)
    for Z in admm.auxes[1][k]:
        Z = np.asarray(Z)
        assert Z.min() >= 0
        for c in range(Z.shape[1] - 1):  # rises to one peak, then falls
            col = Z[:, c]
            p = int(np.argmax(col))
            assert np.all(np.diff(col[: p + 1]) >= -1e-6) and np.all(np.diff(col[p:]) <= 1e-6)


# ---- the EXTERNAL path against the REFERENCE's own trajectories -------------------------------------------------
def _user_classes():
    """Penalties a user would write against the plugin classes (penalties.py:369-463 of the reference): none of them has a
    native descriptor, so every proximal step below runs through these Python methods on device tensors."""
    import torch
    from matcouply_amd import penalties as pen

    class UserNonNeg(pen.RowVectorPenalty):
        def factor_matrix_row_update(self, row, feasibility_penalty, aux_row):
            return torch.clamp(row, min=0)

        def factor_matrix_update(self, factor_matrix, feasibility_penalty, aux):
            return torch.clamp(factor_matrix, min=0)

        def penalty(self, x):
            return 0

    class UserNonNegL1(pen.RowVectorPenalty):
        def __init__(self, strength, **kw):
            super().__init__(**kw)
            self.strength = strength

        def factor_matrix_row_update(self, row, feasibility_penalty, aux_row):
            return torch.clamp(row - self.strength / feasibility_penalty, min=0)

        def factor_matrix_update(self, factor_matrix, feasibility_penalty, aux):
            return torch.clamp(factor_matrix - self.strength / feasibility_penalty, min=0)

        def penalty(self, x):
            xs = x if isinstance(x, list) else [x]
            return self.strength * sum(float(xi.double().abs().sum()) for xi in xs)

    class UserBall(pen.HardConstraintMixin, pen.MatrixPenalty):
        def __init__(self, bound, **kw):
            super().__init__(**kw)
            self.bound = bound

        def factor_matrix_update(self, factor_matrix, feasibility_penalty, aux):
            norms = torch.linalg.norm(factor_matrix.double(), dim=0, keepdim=True)
            return (factor_matrix.double() * (self.bound / torch.clamp(norms, min=self.bound))).to(factor_matrix.dtype)

    class UserParafac2(pen.Parafac2):
        """the class's own Python sweep (an override of the prox takes the penalty off the native kernel)"""

        def factor_matrices_update(self, factor_matrices, feasibility_penalties, auxes):
            return super().factor_matrices_update(factor_matrices, feasibility_penalties, auxes)

    return dict(nn=UserNonNeg, l1=UserNonNegL1, l2ball=UserBall, parafac2=UserParafac2)


@pytest.mark.parametrize("fname", ["traj_c3_nn_l1C.npz", "traj_c4_pf2_ball.npz"])
def test_user_written_penalties_follow_the_reference_trajectory(fname):
    """The reference's 20-iteration trajectories (tests/golden/traj_*.npz, generated from the imported reference) with EVERY
    penalty replaced by a user-written class: factors, errors, losses and feasibility gaps at the bar of the native path."""
    import json

    from matcouply_amd import decomposition as dec
    from matcouply_amd import penalties as pen
    from tests.helpers import load_npz
    from tests.test_gpu_end_to_end import gap_error

    arrs = load_npz(fname)
    spec = json.loads(str(arrs["spec"]))
    c1 = load_npz("c1_data.npz")
    X, rp = c1["X"], c1["row_ptr"]
    user = _user_classes()
    regs = [[], [], []]
    for m in range(3):
        for s, d in enumerate(spec["regs"][m]):
            dual = split_rows(arrs[f"dual_in_m{m}_{s}"], rp) if m == 1 else arrs[f"dual_in_m{m}_{s}"].copy()
            if d["kind"] == "parafac2":
                reg = user["parafac2"](aux_init=(split_rows(arrs[f"aux_in_m{m}_{s}_P"], rp), arrs[f"aux_in_m{m}_{s}_Delta"].copy()),
                                       dual_init=dual)
            else:
                aux = split_rows(arrs[f"aux_in_m{m}_{s}"], rp) if m == 1 else arrs[f"aux_in_m{m}_{s}"].copy()
                if d["kind"] == "nn":
                    reg = user["nn"](aux_init=aux, dual_init=dual)
                elif d["kind"] == "l1":
                    assert d.get("non_negativity", False)
                    reg = user["l1"](d["reg_strength"], aux_init=aux, dual_init=dual)
                elif d["kind"] == "l2ball":
                    assert not d.get("non_negativity", False)
                    reg = user["l2ball"](d["norm_bound"], aux_init=aux, dual_init=dual)
                else:
                    raise AssertionError(d)
            assert pen.native_descriptor_of(reg) is None, reg  # host-evaluated: the engine only solves
            regs[m].append(reg)
    kw = spec["kwargs"]
    cmf, admm, diag = dec.cmf_aoadmm(
        split_rows(X, rp), arrs["A0"].shape[1], init=(None, (arrs["A0"].copy(), split_rows(arrs["B0"], rp), arrs["C0"].copy())),
        regs=regs, n_iter_max=spec["n_iter_max"], tol=None, absolute_tol=None, return_errors=True, return_admm_vars=True,
        l2_penalty=kw.get("l2_penalty", None), feasibility_penalty_scale=kw.get("feasibility_penalty_scale", 1),
        constant_feasibility_penalty=kw.get("constant_feasibility_penalty", False))
    e = {"A": rel_err(cmf[1][0], arrs["A"]), "B": rel_err(np.concatenate(cmf[1][1]), arrs["B"]), "C": rel_err(cmf[1][2], arrs["C"])}
    ref_gaps = [[arrs[f"gaps_m{m}"][it] for m in range(3)] for it in range(len(arrs["rec_errors"]))]
    e["gaps"] = gap_error(diag.feasibility_gaps, ref_gaps)
    print(fname, "user-written penalties, 20 it vs reference:", {k: f"{v:.1e}" for k, v in e.items()})
    np.testing.assert_allclose(diag.rec_errors, arrs["rec_errors"], rtol=1e-5)
    np.testing.assert_allclose(diag.regularized_loss, arrs["regularized_loss"], rtol=1e-5)
    assert max(e["A"], e["B"], e["C"]) < 1e-5 and e["gaps"] <= 1.0, e
