"""Pins the oracle (oracle/aoadmm_oracle.py + oracle/csrc/unimodal_oracle.c) against fixtures captured from
the unmodified reference (tests/golden/, made by oracle/tools/gen_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest

from oracle import aoadmm_oracle as orc
from tests.helpers import GOLDEN, load_npz, manifest_of, rel_err

TOL = 1e-10  # fp64 restatement vs fp64 reference (different but equivalent r x r solve)

PHASE_B = load_npz("phase_B.npz")
SHARED = {k: PHASE_B[k] for k in PHASE_B if not k.startswith("c") or k in ("C",)}


def _phase_state(mode, case):
    S = PHASE_B
    regs, aux, dual = [[], [], []], [[], [], []], [[], [], []]
    name = "ABC"[mode]
    for s, d in enumerate(case["regs"]):
        regs[mode].append(d)
        if d["kind"] == "parafac2":
            aux[mode].append((S["P0"].copy(), S["Delta0"].copy()))
        else:
            aux[mode].append(S[f"aux{name}{s}"].copy())
        dual[mode].append(S[f"dual{name}{s}"].copy())
    l2 = [0.0, 0.0, 0.0]
    l2[mode] = case["l2"]
    return orc.OracleState(S["X"], S["row_ptr"], S["A"], S["B"], S["C"], regs, aux, dual, l2=l2,
                           inner_n_iter_max=case["inner"], feasibility_penalty_scale=case["scale"],
                           constant_A=case["constant"], constant_B=case["constant"])


def _phase_cases(name):
    arrs = load_npz(f"phase_{name}.npz")
    return arrs, manifest_of(arrs)


@pytest.mark.parametrize("name,mode", [("B", 1), ("C", 2), ("A", 0)])
def test_phase_goldens(name, mode):
    arrs, manifest = _phase_cases(name)
    assert len(manifest) >= 16
    for ci, case in enumerate(manifest):
        st = _phase_state(mode, case)
        if mode == 1:
            st.update_B()
            factor = st.B
        elif mode == 2:
            st.update_C()
            factor = st.C
        else:
            rhs, Q = st.update_A()
            factor = st.A
            assert rel_err(rhs, arrs[f"c{ci}_rhses"]) < TOL, (ci, case)
            assert rel_err(Q, arrs[f"c{ci}_cross_products"]) < TOL, (ci, case)
        assert rel_err(factor, arrs[f"c{ci}_factor"]) < TOL, (name, ci, case)
        for s, d in enumerate(case["regs"]):
            if d["kind"] == "parafac2":
                P, Delta = st.aux[mode][s]
                assert rel_err(P, arrs[f"c{ci}_aux{s}_P"]) < 1e-8, (ci, case)
                assert rel_err(Delta, arrs[f"c{ci}_aux{s}_Delta"]) < 1e-8, (ci, case)
            else:
                assert rel_err(st.aux[mode][s], arrs[f"c{ci}_aux{s}"]) < 1e-8, (name, ci, case)
            assert rel_err(st.dual[mode][s], arrs[f"c{ci}_dual{s}"]) < 1e-8, (name, ci, case)


def test_prox_goldens():
    arrs = load_npz("prox.npz")
    descs = manifest_of(arrs)
    Y, row_ptr, rhos = arrs["Y"], arrs["row_ptr"], arrs["rhos"]
    J = int(row_ptr[1])
    for ci, d in enumerate(descs):
        got = orc.prox_matrix(d, Y[:J].copy(), 10.0)
        assert rel_err(got, arrs[f"p{ci}_single_rho10"]) < 1e-13, d
        out = np.concatenate([orc.prox_matrix(d, Y[row_ptr[i]:row_ptr[i + 1]].copy(), rhos[i])
                              for i in range(len(rhos))])
        assert rel_err(out, arrs[f"p{ci}_list"]) < 1e-13, d
        if f"p{ci}_row" in arrs:
            assert rel_err(orc.prox_elementwise(d, Y[0].copy(), 2.5), arrs[f"p{ci}_row"]) < 1e-13, d
        assert abs(orc.penalty_value(d, Y[:J]) - float(arrs[f"p{ci}_penalty"])) < 1e-12
        assert abs(orc.penalty_value(d, Y) - float(arrs[f"p{ci}_penalty_list"])) < 1e-12
    P1, D1 = orc.prox_parafac2(Y, row_ptr, rhos, (arrs["pf2_P0"], arrs["pf2_D0"]))
    assert rel_err(P1, arrs["pf2_P1"]) < 1e-10
    assert rel_err(D1, arrs["pf2_D1"]) < 1e-10
    assert rel_err(P1 @ D1 - Y, arrs["pf2_aux_minus_Y"]) < 1e-10


@pytest.mark.parametrize("force_python", [True, False])
def test_unimodal_goldens(force_python):
    if not force_python and orc._oracle_lib() is None:
        pytest.skip("oracle C library not built (run __graft_entry__.build())")
    arrs = load_npz("prox.npz")
    for ui in range(int(arrs["n_uni"])):
        y = arrs[f"uni{ui}_y"]
        for nn, key in ((False, "out"), (True, "out_nn")):
            got = orc.unimodal_columns(y, nn, force_python=force_python)
            np.testing.assert_allclose(got, arrs[f"uni{ui}_{key}"], rtol=0, atol=1e-13)


def test_unimodal_c_equals_python_random():
    if orc._oracle_lib() is None:
        pytest.skip("oracle C library not built")
    rng = np.random.RandomState(0)
    for n in (1, 2, 3, 17, 64):
        Y = rng.standard_normal((n, 5))
        Y[:, 1] = np.round(Y[:, 1])  # ties
        for nn in (False, True):
            np.testing.assert_array_equal(orc.unimodal_columns(Y, nn), orc.unimodal_columns(Y, nn, force_python=True))


def _traj_state(arrs, spec):
    c1 = load_npz("c1_data.npz")
    regs, aux, dual = spec["regs"], [[], [], []], [[], [], []]
    for m in range(3):
        for s, d in enumerate(regs[m]):
            if d["kind"] == "parafac2":
                aux[m].append((arrs[f"aux_in_m{m}_{s}_P"], arrs[f"aux_in_m{m}_{s}_Delta"]))
            else:
                aux[m].append(arrs[f"aux_in_m{m}_{s}"])
            dual[m].append(arrs[f"dual_in_m{m}_{s}"])
    kw = spec["kwargs"]
    const = kw.get("constant_feasibility_penalty", False)
    return orc.OracleState(c1["X"], c1["row_ptr"], arrs["A0"], arrs["B0"], arrs["C0"], regs, aux, dual,
                           l2=kw.get("l2_penalty", [0, 0, 0]), feasibility_penalty_scale=kw.get("feasibility_penalty_scale", 1.0),
                           constant_A=const, constant_B=const)


TRAJ = sorted(f for f in os.listdir(GOLDEN) if f.startswith("traj_") and "seeded" not in f)


@pytest.mark.parametrize("fname", TRAJ)
def test_trajectory_goldens(fname):
    arrs = load_npz(fname)
    spec = json.loads(str(arrs["spec"]))
    st = _traj_state(arrs, spec)
    res = orc.run(st, spec["n_iter_max"], tol=None, absolute_tol=None, return_errors=True)
    np.testing.assert_allclose(res["rec_errors"], arrs["rec_errors"], rtol=1e-8)
    np.testing.assert_allclose(res["losses"], arrs["regularized_loss"], rtol=1e-8)
    for m in range(3):
        got = np.array([it[m] for it in res["gaps"]]).reshape(len(res["gaps"]), -1)
        np.testing.assert_allclose(got, arrs[f"gaps_m{m}"], rtol=1e-6, atol=1e-12)
    assert rel_err(st.A, arrs["A"]) < 1e-8
    assert rel_err(st.B, arrs["B"]) < 1e-8
    assert rel_err(st.C, arrs["C"]) < 1e-8
    for m in range(3):
        for s, d in enumerate(spec["regs"][m]):
            if d["kind"] == "parafac2":
                assert rel_err(st.aux[m][s][0], arrs[f"aux_m{m}_{s}_P"]) < 1e-7
                assert rel_err(st.aux[m][s][1], arrs[f"aux_m{m}_{s}_Delta"]) < 1e-7
            else:
                assert rel_err(st.aux[m][s], arrs[f"aux_m{m}_{s}"]) < 1e-7
            assert rel_err(st.dual[m][s], arrs[f"dual_m{m}_{s}"]) < 1e-7


def test_stopping_matrix():
    data = load_npz("stopping_data.npz")
    with open(os.path.join(GOLDEN, "stopping.json")) as f:
        results = json.load(f)
    dec = lambda v: None if v is None else (float(v) if isinstance(v, str) else v)
    nn = {"kind": "nn"}
    checked = 0
    for res in results:
        case = {k: dec(v) for k, v in res["case"].items()}
        if "raises" in res or not case.get("return_errors", True):
            continue
        st = orc.OracleState(data["X"], data["row_ptr"], data["A0"], data["B0"], data["C0"], [[nn], [nn], [nn]],
                             [[data[f"aux{m}"]] for m in range(3)], [[data[f"dual{m}"]] for m in range(3)])
        out = orc.run(st, int(case["n_iter_max"]), tol=case["tol"], absolute_tol=case["absolute_tol"],
                      feasibility_tol=case["feasibility_tol"])
        assert out["message"] == res["message"], case
        assert out["n_iter"] == res["n_iter"], case
        assert len(out["rec_errors"]) == res["n_rec"] and len(out["losses"]) == res["n_loss"], case
        assert len(out["gaps"]) == res["n_gaps"], case
        assert out["satisfied_stopping_condition"] == res["satisfied_stopping_condition"], case
        feas = out["satisfied_feasibility_condition"]
        assert (None if feas is None else bool(feas)) == res["satisfied_feasibility_condition"], case
        np.testing.assert_allclose(out["rec_errors"][-1], res["last_rec"], rtol=1e-7)
        checked += 1
    assert checked >= 8


def test_fp32_mode_tracks_fp64():
    """The oracle's fp32 mode (same arithmetic, float32 storage) stays within 1e-5 of fp64 on a short trajectory."""
    arrs = load_npz("traj_c3_nn_l1C.npz")
    spec = json.loads(str(arrs["spec"]))
    st64 = _traj_state(arrs, spec)
    st32 = _traj_state(arrs, spec)
    st32 = orc.OracleState(st32.X, st32.row_ptr, st32.A, st32.B, st32.C, st32.regs, st32.aux, st32.dual, dtype=np.float32)
    for _ in range(5):
        for st in (st64, st32):
            st.update_B(); st.update_C(); st.update_A()
    assert rel_err(st32.B, st64.B) < 1e-5 and rel_err(st32.C, st64.C) < 1e-5 and rel_err(st32.A, st64.A) < 1e-5
