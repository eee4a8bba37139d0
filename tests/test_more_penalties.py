"""GeneralizedL2Penalty and UnitSimplex against golden vectors of the reference
(tests/golden/more_penalties.npz, oracle/tools/gen_golden.py --only more): the oracle's restatement, the product's
host prox methods on NumPy arrays and torch tensors, and (-m gpu) a 10-iteration trajectory through the solver, where
both penalties are evaluated on device tensors between the library's solve steps."""
import json

import numpy as np
import pytest

from oracle import aoadmm_oracle as orc
from tests.helpers import load_npz, rel_err, split_rows


def _descs(arrs):
    out = []
    for d in json.loads(str(arrs["manifest"])):
        d = dict(d)
        if d["kind"] == "gl2":
            d["norm_matrix"] = arrs[d["norm_matrix"]]
        out.append(d)
    return out


def _make(d, **kw):
    from matcouply_amd import penalties as pen

    return pen.GeneralizedL2Penalty(d["norm_matrix"], **kw) if d["kind"] == "gl2" else pen.UnitSimplex(**kw)


def test_oracle_prox_matches_reference():
    arrs = load_npz("more_penalties.npz")
    Y, row_ptr, rhos = arrs["Y"], arrs["row_ptr"], arrs["rhos"]
    J = int(row_ptr[1])
    for ci, d in enumerate(_descs(arrs)):
        tol = 1e-9 if d["kind"] == "simplex" else 1e-12  # the reference's bisection stops at ~1e-12 of the multiplier
        assert rel_err(orc.prox_matrix(d, Y[:J].copy(), 10.0), arrs[f"p{ci}_single_rho10"]) < tol, d["kind"]
        out = np.concatenate([orc.prox_matrix(d, Y[row_ptr[i]:row_ptr[i + 1]].copy(), rhos[i]) for i in range(len(rhos))])
        assert rel_err(out, arrs[f"p{ci}_list"]) < tol
        assert abs(orc.penalty_value(d, Y[:J]) - float(arrs[f"p{ci}_penalty"])) < 1e-10
        assert abs(orc.penalty_value(d, Y) - float(arrs[f"p{ci}_penalty_list"])) < 1e-10


@pytest.mark.parametrize("backend", ["numpy", "torch"])
def test_host_prox_matches_reference(backend):
    import torch

    arrs = load_npz("more_penalties.npz")
    Y, row_ptr, rhos = arrs["Y"], arrs["row_ptr"], arrs["rhos"]
    J = int(row_ptr[1])
    cast = (lambda a: torch.as_tensor(np.array(a))) if backend == "torch" else (lambda a: np.array(a))
    back = (lambda a: a.numpy()) if backend == "torch" else (lambda a: np.asarray(a))
    for ci, d in enumerate(_descs(arrs)):
        p = _make(d)
        tol = 1e-9 if d["kind"] == "simplex" else 1e-12
        assert rel_err(back(p.factor_matrix_update(cast(Y[:J]), 10.0, None)), arrs[f"p{ci}_single_rho10"]) < tol
        out = p.factor_matrices_update([cast(m) for m in split_rows(Y, row_ptr)], list(rhos), [None] * len(rhos))
        assert rel_err(np.concatenate([back(m) for m in out]), arrs[f"p{ci}_list"]) < tol
        assert abs(float(p.penalty(cast(Y[:J]))) - float(arrs[f"p{ci}_penalty"])) < 1e-10
        assert abs(float(p.penalty([cast(m) for m in split_rows(Y, row_ptr)])) - float(arrs[f"p{ci}_penalty_list"])) < 1e-10
        if d["kind"] == "simplex":
            x = back(p.factor_matrix_update(cast(Y[:J]), 1.0, None))
            assert x.min() >= 0 and np.allclose(x.sum(axis=0), 1.0, atol=1e-12)


def test_generalized_l2_validation():
    from matcouply_amd import penalties as pen

    M = np.array([[2.0, -1.0], [-1.0, 2.0]])
    pen.GeneralizedL2Penalty(M)
    with pytest.raises(ValueError):
        pen.GeneralizedL2Penalty(np.array([[1.0, 0.5], [0.0, 1.0]]))   # not symmetric
    with pytest.raises(ValueError):
        pen.GeneralizedL2Penalty(np.array([[1.0, 2.0], [2.0, 1.0]]))   # indefinite
    pen.GeneralizedL2Penalty(np.array([[1.0, 2.0], [2.0, 1.0]]), validate=False)
    # M = I is the ridge penalty: prox = rho / (rho + 2) x
    x = np.random.RandomState(0).standard_normal((4, 3))
    assert np.allclose(pen.GeneralizedL2Penalty(np.eye(4)).factor_matrix_update(x, 3.0, None), 3.0 / 5.0 * x)


def _traj_state(arrs):
    spec = json.loads(str(arrs["t_spec"]))
    regs = [[dict(d, norm_matrix=arrs[d["norm_matrix"]]) if d["kind"] == "gl2" else d for d in spec["regs"][m]]
            for m in range(3)]
    c1 = load_npz("c1_data.npz")
    aux = [[arrs[f"t_aux_in_m{m}_{s}"] for s in range(len(regs[m]))] for m in range(3)]
    dual = [[arrs[f"t_dual_in_m{m}_{s}"] for s in range(len(regs[m]))] for m in range(3)]
    return orc.OracleState(c1["X"], c1["row_ptr"], arrs["t_A0"], arrs["t_B0"], arrs["t_C0"], regs, aux, dual), spec, regs


def test_oracle_trajectory_matches_reference():
    arrs = load_npz("more_penalties.npz")
    st, spec, regs = _traj_state(arrs)
    res = orc.run(st, spec["n_iter_max"], tol=None, absolute_tol=None)
    assert rel_err(st.A, arrs["t_A"]) < 1e-8 and rel_err(st.B, arrs["t_B"]) < 1e-8 and rel_err(st.C, arrs["t_C"]) < 1e-8
    np.testing.assert_allclose(res["rec_errors"], arrs["t_rec_errors"], rtol=1e-9)
    np.testing.assert_allclose(res["losses"], arrs["t_regularized_loss"], rtol=1e-9)
    for m in range(3):
        for s in range(len(regs[m])):
            assert rel_err(st.aux[m][s], arrs[f"t_aux_m{m}_{s}"]) < 1e-8
            assert rel_err(st.dual[m][s], arrs[f"t_dual_m{m}_{s}"]) < 1e-8


@pytest.mark.gpu
def test_solver_trajectory_matches_reference():
    """GeneralizedL2Penalty on the B_i and UnitSimplex on C through cmf_aoadmm on the GPU (host-evaluated prox on device
    tensors between native solve steps) vs the reference's own 10-iteration trajectory."""
    from matcouply_amd import decomposition as dec, penalties as pen

    arrs = load_npz("more_penalties.npz")
    c1 = load_npz("c1_data.npz")
    X, row_ptr = c1["X"], c1["row_ptr"]
    spec = json.loads(str(arrs["t_spec"]))

    def mk(d, m, s):
        aux, dual = arrs[f"t_aux_in_m{m}_{s}"], arrs[f"t_dual_in_m{m}_{s}"]
        kw = dict(aux_init=split_rows(aux, row_ptr) if m == 1 else aux.copy(),
                  dual_init=split_rows(dual, row_ptr) if m == 1 else dual.copy())
        if d["kind"] == "gl2":
            return pen.GeneralizedL2Penalty(arrs[d["norm_matrix"]], **kw)
        if d["kind"] == "simplex":
            return pen.UnitSimplex(**kw)
        return pen.NonNegativity(**kw)

    regs = [[mk(d, m, s) for s, d in enumerate(spec["regs"][m])] for m in range(3)]
    cmf, admm, diag = dec.cmf_aoadmm(
        split_rows(X, row_ptr), spec["rank"], init=(None, (arrs["t_A0"].copy(), split_rows(arrs["t_B0"], row_ptr), arrs["t_C0"].copy())),
        regs=regs, n_iter_max=spec["n_iter_max"], tol=None, absolute_tol=None, return_errors=True, return_admm_vars=True)
    tol = 2e-5  # fp32 engine vs fp64 reference over 10 outer iterations
    assert rel_err(cmf[1][0], arrs["t_A"]) < tol and rel_err(cmf[1][2], arrs["t_C"]) < tol
    assert rel_err(np.concatenate(cmf[1][1]), arrs["t_B"]) < tol
    np.testing.assert_allclose(diag.rec_errors, arrs["t_rec_errors"], rtol=2e-5)
    np.testing.assert_allclose(diag.regularized_loss, arrs["t_regularized_loss"], rtol=4e-5)
    assert rel_err(admm.auxes[2][0], arrs["t_aux_m2_0"]) < tol
    assert np.allclose(np.sum(admm.auxes[2][0], axis=0), 1.0, atol=1e-5) and np.min(admm.auxes[2][0]) >= 0


@pytest.mark.gpu
def test_generalized_l2_keyword():
    from matcouply_amd import decomposition as dec
    from matcouply_amd import penalties as pen

    c1 = load_npz("c1_data.npz")
    mats = split_rows(c1["X"], c1["row_ptr"])
    K = mats[0].shape[1]
    M = 2 * np.eye(K) - np.eye(K, k=1) - np.eye(K, k=-1)
    cmf, diag = dec.cmf_aoadmm(mats, 3, generalized_l2_penalty={2: 0.1 * M}, non_negative={0: True}, n_iter_max=5,
                               tol=None, absolute_tol=None, return_errors=True, random_state=0)
    assert np.isfinite(diag.regularized_loss).all() and diag.rec_errors[-1] < diag.rec_errors[0]
    with pytest.raises(ValueError):
        pen.GeneralizedL2Penalty(np.ones((3, 4)))
