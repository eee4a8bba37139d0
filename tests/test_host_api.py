"""CPU tests of the product's HOST logic (matcouply_amd.decomposition / penalties / coupled_matrices / data) with the
oracle-backed checker engine substituted for the HIP engine: argument parsing, RNG draw order, stopping rules,
return types and error behaviour - pinned against fixtures captured from the reference."""
import json
import os

import numpy as np
import pytest

import matcouply_amd
from matcouply_amd import decomposition as dec
from matcouply_amd import penalties as pen
from matcouply_amd.coupled_matrices import CoupledMatrixFactorization, cmf_to_matrices
from matcouply_amd.data import get_simple_simulated_data
from matcouply_amd.random import random_coupled_matrices
from tests.helpers import GOLDEN, load_npz, manifest_of, rel_err, split_rows
from tests.oracle_engine import OracleEngineFactory


@pytest.fixture
def checker_engine(monkeypatch):
    monkeypatch.setattr(dec, "_ENGINE_FACTORY", OracleEngineFactory())


def make_penalty(d, aux_init="random_uniform", dual_init="random_uniform"):
    kw = dict(aux_init=aux_init, dual_init=dual_init)
    k = d["kind"]
    if k == "nn":
        return pen.NonNegativity(**kw)
    if k == "box":
        return pen.Box(d["min_val"], d["max_val"], **kw)
    if k == "l1":
        return pen.L1Penalty(d["reg_strength"], non_negativity=d.get("non_negativity", False), **kw)
    if k == "l2ball":
        return pen.L2Ball(d["norm_bound"], non_negativity=d.get("non_negativity", False), **kw)
    if k == "unimodal":
        return pen.Unimodality(non_negativity=d.get("non_negativity", False), **kw)
    if k == "parafac2":
        return pen.Parafac2(**kw)
    if k == "tv":
        return pen.TotalVariationPenalty(d["reg_strength"], l1_strength=d.get("l1_strength", 0.0), **kw)
    if k == "gl2":
        return pen.GeneralizedL2Penalty(d["norm_matrix"], **kw)
    if k == "simplex":
        return pen.UnitSimplex(**kw)
    raise ValueError(k)


def test_no_gpu_fails_loudly():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    X, _ = get_simple_simulated_data()
    with pytest.raises(matcouply_amd._engine.EngineError, match="no CPU fallback"):
        dec.cmf_aoadmm(X, 3, n_iter_max=1)


def test_bench_launcher_starts_its_own_ranks_and_reports_their_failure():
    """`python bench.py --gpus 2` without torch.distributed.run around it (the driver's command form) becomes the launcher:
    here, without a GPU, both ranks refuse loudly and the launcher's exit code is theirs (the GPU run: -m gpu suite)"""
    import subprocess
    import sys
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and out.stdout.strip() == ""
    assert "needs an MI355X" in out.stderr and "WORLD_SIZE" not in out.stderr.split("Traceback")[0][:200]


def test_simulated_data_matches_reference():
    c1 = load_npz("c1_data.npz")
    X, cmf = get_simple_simulated_data(noise_level=0.2, random_state=1)
    np.testing.assert_allclose(np.concatenate(X, 0), c1["X"], rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(cmf[1][0], c1["A_true"], rtol=1e-14)
    np.testing.assert_allclose(cmf[1][2], c1["C_true"], rtol=1e-14)


def test_listify_and_parse_order():
    assert dec._listify(1, "x") == [1, 1, 1]
    assert dec._listify({1: 2}, "x") == [None, 2, None]
    assert dec._listify([1, 2, 3], "x") == [1, 2, 3]
    with pytest.raises(ValueError):
        dec._listify([1, 2], "x")
    regs = dec._parse_mode_penalties(non_negative=True, lower_bound=None, upper_bound=None, l2_norm_bound=1.0,
                                     unimodal=True, parafac2=True, l1_penalty=0.1, tv_penalty=None,
                                     generalized_l2_penalty=None, svd="truncated_svd", dual_init="zeros", aux_init="zeros")
    assert [type(r).__name__ for r in regs] == ["Parafac2", "Unimodality", "L2Ball", "L1Penalty"]
    assert all(getattr(r, "non_negativity", True) for r in regs[1:])
    regs = dec._parse_mode_penalties(non_negative=True, lower_bound=-1, upper_bound=2, l2_norm_bound=None,
                                     unimodal=None, parafac2=False, l1_penalty=None, tv_penalty=None,
                                     generalized_l2_penalty=None, svd="truncated_svd", dual_init="zeros", aux_init="zeros")
    assert [type(r).__name__ for r in regs] == ["Box"] and regs[0].min_val == 0 and regs[0].max_val == 2
    regs = dec._parse_mode_penalties(non_negative=True, lower_bound=None, upper_bound=None, l2_norm_bound=None,
                                     unimodal=None, parafac2=False, l1_penalty=None, tv_penalty=None,
                                     generalized_l2_penalty=None, svd="truncated_svd", dual_init="zeros", aux_init="zeros")
    assert [type(r).__name__ for r in regs] == ["NonNegativity"]
    with pytest.raises(TypeError):
        dec._parse_all_penalties(None, None, None, None, None, None, None, None, None, "truncated_svd", [[1], [], []],
                                 "zeros", "zeros", False)


def test_penalty_validation_and_repr():
    with pytest.raises(ValueError):
        pen.L1Penalty(-1)
    with pytest.raises(ValueError):
        pen.L2Ball(0)
    p = pen.L1Penalty(0.5, non_negativity=True)
    assert repr(p) == ("<'matcouply_amd.penalties.L1Penalty' with reg_strength=0.5, non_negativity=True, "
                       "aux_init='random_uniform', dual_init='random_uniform')>")
    mats = [np.zeros((4, 6)), np.zeros((5, 6))]
    rs = np.random.RandomState(0)
    assert p.init_aux(mats, 3, 0, rs).shape == (2, 3)
    assert [a.shape for a in p.init_aux(mats, 3, 1, rs)] == [(4, 3), (5, 3)]
    assert p.init_dual(mats, 3, 2, rs).shape == (6, 3)
    with pytest.raises(ValueError):
        p.init_aux(mats, 3, 3, rs)
    with pytest.raises(TypeError):
        p.init_aux(mats, 3.0, 0, rs)
    with pytest.raises(ValueError):
        pen.NonNegativity(aux_init=np.zeros((3, 3))).init_aux(mats, 3, 0, rs)
    with pytest.raises(TypeError):
        pen.NonNegativity(aux_init=np.zeros((2, 3))).init_aux(mats, 3, 1, rs)
    with pytest.raises(ValueError):
        pen.Parafac2().init_aux(mats, 3, 0, rs)
    P, D = pen.Parafac2().init_aux(mats, 3, 1, rs)
    assert D.shape == (3, 3) and np.allclose(P[1], np.eye(5, 3))
    with pytest.raises(TypeError):
        pen.Parafac2().subtract_from_aux(None, None)


def test_host_prox_methods_match_reference():
    arrs = load_npz("prox.npz")
    descs = manifest_of(arrs)
    Y, row_ptr, rhos = arrs["Y"], arrs["row_ptr"], arrs["rhos"]
    J = int(row_ptr[1])
    for ci, d in enumerate(descs):
        p = make_penalty(d)
        assert rel_err(p.factor_matrix_update(Y[:J].copy(), 10.0, None), arrs[f"p{ci}_single_rho10"]) < 1e-13
        out = p.factor_matrices_update(split_rows(Y, row_ptr), list(rhos), [None] * len(rhos))
        assert rel_err(np.concatenate(out), arrs[f"p{ci}_list"]) < 1e-13
        if f"p{ci}_row" in arrs:
            assert rel_err(p.factor_matrix_row_update(Y[0].copy(), 2.5, None), arrs[f"p{ci}_row"]) < 1e-13
        assert abs(float(p.penalty(Y[:J])) - float(arrs[f"p{ci}_penalty"])) < 1e-12
        assert abs(float(p.penalty(split_rows(Y, row_ptr))) - float(arrs[f"p{ci}_penalty_list"])) < 1e-12
    p2 = pen.Parafac2()
    P1, D1 = p2.factor_matrices_update(split_rows(Y, row_ptr), list(rhos),
                                       (split_rows(arrs["pf2_P0"], row_ptr), arrs["pf2_D0"]))
    assert rel_err(np.concatenate(P1), arrs["pf2_P1"]) < 1e-10 and rel_err(D1, arrs["pf2_D1"]) < 1e-10
    shifted = p2.subtract_from_auxes((P1, D1), split_rows(Y, row_ptr))
    assert rel_err(np.concatenate(shifted), arrs["pf2_aux_minus_Y"]) < 1e-10
    for ui in range(int(arrs["n_uni"])):
        for nn, key in ((False, "out"), (True, "out_nn")):
            np.testing.assert_allclose(pen.unimodal_regression(arrs[f"uni{ui}_y"], nn), arrs[f"uni{ui}_{key}"], atol=1e-13)


def test_cmf_container():
    cmf = random_coupled_matrices([(5, 4), (6, 4), (7, 4)], rank=2, random_state=0)
    assert cmf.rank == 2 and cmf.shape == ((5, 4), (6, 4), (7, 4)) and len(cmf) == 2
    w, (A, B_is, C) = cmf
    mats = cmf_to_matrices(cmf)
    np.testing.assert_allclose(mats[1], (B_is[1] * (A[1] * w)) @ C.T)
    assert cmf.to_tensor().shape == (3, 7, 4)
    assert cmf.to_unfolded(2, pad=False).shape == (4, 18)
    assert cmf.to_vec(pad=False).shape == (72,)
    with pytest.raises(TypeError):
        CoupledMatrixFactorization((None, ([1, 2], B_is, C)))
    with pytest.raises(ValueError):
        CoupledMatrixFactorization((None, (A, B_is, C[:, :1])))
    with pytest.raises(ValueError):
        CoupledMatrixFactorization((None, (A[:2], B_is, C)))
    with pytest.raises(IndexError):
        cmf[2]
    with pytest.raises(ValueError):
        random_coupled_matrices([(5, 4), (6, 3)], rank=2)


def _regs_from_traj(arrs, spec, row_ptr):
    regs = [[], [], []]
    for m in range(3):
        for s, d in enumerate(spec["regs"][m]):
            dual = arrs[f"dual_in_m{m}_{s}"]
            dual_init = split_rows(dual, row_ptr) if m == 1 else dual.copy()
            if d["kind"] == "parafac2":
                aux_init = (split_rows(arrs[f"aux_in_m{m}_{s}_P"], row_ptr), arrs[f"aux_in_m{m}_{s}_Delta"].copy())
            else:
                aux = arrs[f"aux_in_m{m}_{s}"]
                aux_init = split_rows(aux, row_ptr) if m == 1 else aux.copy()
            regs[m].append(make_penalty(d, aux_init=aux_init, dual_init=dual_init))
    return regs


TRAJ = sorted(f for f in os.listdir(GOLDEN) if f.startswith("traj_") and "seeded" not in f)


@pytest.mark.parametrize("fname", TRAJ)
def test_cmf_aoadmm_trajectories_through_public_api(checker_engine, fname):
    arrs = load_npz(fname)
    spec = json.loads(str(arrs["spec"]))
    c1 = load_npz("c1_data.npz")
    row_ptr = c1["row_ptr"]
    matrices = split_rows(c1["X"], row_ptr)
    regs = _regs_from_traj(arrs, spec, row_ptr)
    cmf, admm_vars, diag = dec.cmf_aoadmm(
        matrices, spec["rank"], init=(None, (arrs["A0"], split_rows(arrs["B0"], row_ptr), arrs["C0"])), regs=regs,
        n_iter_max=spec["n_iter_max"], tol=None, absolute_tol=None, return_errors=True, return_admm_vars=True,
        **spec["kwargs"])
    assert isinstance(cmf, CoupledMatrixFactorization) and isinstance(admm_vars, dec.ADMMVars)
    assert diag.n_iter == spec["n_iter_max"] and diag.satisfied_stopping_condition is None
    assert diag.message == "MAXIMUM NUMBER OF ITERATIONS REACHED"
    np.testing.assert_allclose(diag.rec_errors, arrs["rec_errors"], rtol=1e-7)
    np.testing.assert_allclose(diag.regularized_loss, arrs["regularized_loss"], rtol=1e-7)
    for m in range(3):
        got = np.array([[float(g) for g in it[m]] for it in diag.feasibility_gaps]).reshape(len(diag.feasibility_gaps), -1)
        np.testing.assert_allclose(got, arrs[f"gaps_m{m}"], rtol=1e-5, atol=1e-12)
    assert rel_err(cmf[1][0], arrs["A"]) < 1e-7 and rel_err(cmf[1][2], arrs["C"]) < 1e-7
    assert rel_err(np.concatenate(cmf[1][1]), arrs["B"]) < 1e-7
    for m in range(3):
        for s, d in enumerate(spec["regs"][m]):
            aux, dual = admm_vars.auxes[m][s], admm_vars.duals[m][s]
            if d["kind"] == "parafac2":
                assert rel_err(np.concatenate(aux[0]), arrs[f"aux_m{m}_{s}_P"]) < 1e-6
                assert rel_err(aux[1], arrs[f"aux_m{m}_{s}_Delta"]) < 1e-6
            else:
                assert rel_err(np.concatenate(aux) if m == 1 else aux, arrs[f"aux_m{m}_{s}"]) < 1e-6
            assert rel_err(np.concatenate(dual) if m == 1 else dual, arrs[f"dual_m{m}_{s}"]) < 1e-6


def test_seeded_keyword_run_pins_rng_order(checker_engine):
    """init="random" + keyword penalties + random_state=0 must draw A, C, B_i, aux (modes 0,1,2), duals (0,1,2) in the
    reference's order (decomposition.py:35-37, 904-905)."""
    arrs = load_npz("traj_seeded_keywords.npz")
    c1 = load_npz("c1_data.npz")
    matrices = split_rows(c1["X"], c1["row_ptr"])
    cmf, admm_vars, diag = dec.cmf_aoadmm(
        matrices, 3, non_negative=True, l1_penalty={2: 0.1}, l2_norm_bound={1: 1.0}, parafac2=True, n_iter_max=10,
        tol=None, absolute_tol=None, return_errors=True, return_admm_vars=True, random_state=0)
    np.testing.assert_allclose(diag.rec_errors, arrs["rec_errors"], rtol=1e-7)
    np.testing.assert_allclose(diag.regularized_loss, arrs["regularized_loss"], rtol=1e-7)
    assert rel_err(cmf[1][0], arrs["A"]) < 1e-7 and rel_err(np.concatenate(cmf[1][1]), arrs["B"]) < 1e-7
    assert rel_err(admm_vars.auxes[1][0][1], arrs["aux_B0_Delta"]) < 1e-6
    assert rel_err(np.concatenate(admm_vars.duals[1][1]), arrs["dual_B1"]) < 1e-6


def test_vectorised_ring_readers_equal_the_per_iteration_host_loop(checker_engine, capsys):
    """The fixed-count call (mcl_iterate) and the call under a stopping rule (mcl_run) turn their diagnostics / verdict rings
    into rec_errors, losses and feasibility gaps in ONE vectorised pass; `verbose` takes the per-iteration host loop
    (read_diag + _StopRule).  Same lists, to the bit - values, lengths and the stopping iteration."""
    c1 = load_npz("c1_data.npz")
    matrices = split_rows(c1["X"], c1["row_ptr"])
    kw = dict(non_negative=True, l1_penalty={2: 0.1}, l2_norm_bound={1: 1.0}, l2_penalty={0: 0.05}, parafac2=True,
              return_errors=True, random_state=0)
    for tols in (dict(tol=None, absolute_tol=None, n_iter_max=12), dict(tol=1e-3, absolute_tol=1e-10, n_iter_max=60)):
        _, quiet = dec.cmf_aoadmm(matrices, 3, **kw, **tols)
        _, loud = dec.cmf_aoadmm(matrices, 3, verbose=1, **kw, **tols)
        capsys.readouterr()
        assert quiet.n_iter == loud.n_iter and quiet.message == loud.message
        assert list(quiet.rec_errors) == list(loud.rec_errors) and list(quiet.regularized_loss) == list(loud.regularized_loss)
        assert len(quiet.feasibility_gaps) == len(loud.feasibility_gaps)
        for ga, gb in zip(quiet.feasibility_gaps, loud.feasibility_gaps):
            assert [list(m) for m in ga] == [list(m) for m in gb]
        assert quiet.satisfied_feasibility_condition == loud.satisfied_feasibility_condition


README_KW = dict(non_negative=True, l1_penalty={2: 0.1}, l2_norm_bound=[1, 1, 0], parafac2=True, unimodal={1: True},
                 constant_feasibility_penalty=True, random_state=0)


def test_readme_example_of_the_reference(checker_engine):
    """The call of the reference's README (README.rst:66-91): L2 ball on A under a constant feasibility penalty, the full
    B stack (PARAFAC2, unimodality, L2 ball), L1 on C - 10 iterations against the reference's own trajectory."""
    arrs = load_npz("readme_example.npz")
    c1 = load_npz("c1_data.npz")
    cmf, admm_vars, diag = dec.cmf_aoadmm(split_rows(c1["X"], c1["row_ptr"]), 3, n_iter_max=10, tol=None, absolute_tol=None,
                                          return_errors=True, return_admm_vars=True, **README_KW)
    assert [len(a) for a in admm_vars.auxes] == list(arrs["n_regs"])
    np.testing.assert_allclose(diag.rec_errors, arrs["rec_errors"], rtol=1e-7)
    np.testing.assert_allclose(diag.regularized_loss, arrs["regularized_loss"], rtol=1e-7)
    for m in range(3):
        got = np.array([[float(g) for g in it[m]] for it in diag.feasibility_gaps]).reshape(len(diag.feasibility_gaps), -1)
        np.testing.assert_allclose(got, arrs[f"gaps_m{m}"], rtol=1e-6, atol=1e-12)
    assert rel_err(cmf[1][0], arrs["A"]) < 1e-7 and rel_err(np.concatenate(cmf[1][1]), arrs["B"]) < 1e-7
    assert rel_err(cmf[1][2], arrs["C"]) < 1e-7
    assert rel_err(admm_vars.auxes[0][0], arrs["aux_A0"]) < 1e-6 and rel_err(admm_vars.duals[0][0], arrs["dual_A0"]) < 1e-6
    assert rel_err(admm_vars.auxes[1][0][1], arrs["aux_B0_Delta"]) < 1e-6


def test_config1_known_answer(checker_engine):
    """BASELINE config 1: parafac2_aoadmm(non_negative=True, random_state=0) on the simulated data converges by the
    tolerance rules after the same number of iterations as the reference run in the build container."""
    with open(os.path.join(GOLDEN, "c1_known_answer.json")) as f:
        ref = json.load(f)
    X, _ = get_simple_simulated_data(noise_level=0.2, random_state=1)
    cmf, diag = dec.parafac2_aoadmm(X, 3, non_negative=True, random_state=0, return_errors=True)
    assert diag.n_iter == ref["n_iter"] and diag.message == ref["message"]
    np.testing.assert_allclose(diag.rec_errors[-1], ref["final_rec_error"], rtol=1e-6)
    np.testing.assert_allclose(diag.regularized_loss[-1], ref["final_loss"], rtol=1e-6)


def test_stopping_matrix_through_public_api(checker_engine):
    data = load_npz("stopping_data.npz")
    with open(os.path.join(GOLDEN, "stopping.json")) as f:
        results = json.load(f)
    row_ptr = data["row_ptr"]
    matrices = split_rows(data["X"], row_ptr)
    dec_ = lambda v: None if v is None else (float(v) if isinstance(v, str) else v)
    for res in results:
        case = {k: dec_(v) for k, v in res["case"].items()}
        return_errors = case.pop("return_errors", True)
        case["n_iter_max"] = int(case["n_iter_max"])
        regs = [[pen.NonNegativity(aux_init=(split_rows(data[f"aux{m}"], row_ptr) if m == 1 else data[f"aux{m}"].copy()),
                                   dual_init=(split_rows(data[f"dual{m}"], row_ptr) if m == 1 else data[f"dual{m}"].copy()))]
                for m in range(3)]
        kw = dict(init=(None, (data["A0"].copy(), split_rows(data["B0"], row_ptr), data["C0"].copy())), regs=regs,
                  return_errors=return_errors, **case)
        if "raises" in res:
            with pytest.raises(eval(res["raises"])):
                dec.cmf_aoadmm(matrices, 2, **kw)
            continue
        out = dec.cmf_aoadmm(matrices, 2, **kw)
        if not return_errors:
            assert isinstance(out, CoupledMatrixFactorization)
            np.testing.assert_allclose(float(np.sum(out[1][0])), res["A_sum"], rtol=1e-6)
            continue
        cmf, diag = out
        assert diag.message == res["message"], case
        assert diag.n_iter == res["n_iter"], case
        assert (len(diag.rec_errors), len(diag.regularized_loss), len(diag.feasibility_gaps)) == \
            (res["n_rec"], res["n_loss"], res["n_gaps"]), case
        assert diag.satisfied_stopping_condition == res["satisfied_stopping_condition"], case
        feas = diag.satisfied_feasibility_condition
        assert (None if feas is None else bool(feas)) == res["satisfied_feasibility_condition"], case
        np.testing.assert_allclose(diag.rec_errors[-1], res["last_rec"], rtol=1e-6)


def test_argument_errors(checker_engine):
    X, _ = get_simple_simulated_data()
    with pytest.raises(ValueError, match="must be 'A' or 'B'"):
        dec.cmf_aoadmm(X, 3, n_iter_max=1, constant_feasibility_penalty="C")
    with pytest.raises(ValueError, match="not recognized"):
        dec.cmf_aoadmm(X, 3, n_iter_max=1, init="nope")
    with pytest.raises(TypeError):
        dec.cmf_aoadmm(X, 3, n_iter_max=1, regs=[[1], [], []])
    with pytest.raises(ValueError):
        dec.cmf_aoadmm(X, 3, n_iter_max=1, l2_penalty=[1, 2])
    with pytest.raises(ValueError):
        dec.cmf_aoadmm(X, 3, n_iter_max=1, tv_penalty=-1.0)  # negative TV strength (penalties.py:812-813)
    # update_X=False freezes that mode and drops its penalties (decomposition.py:896-901)
    rs = np.random.RandomState(3)
    init = (None, (rs.uniform(size=(15, 3)), [rs.uniform(size=(50, 3)) for _ in range(15)], rs.uniform(size=(20, 3))))
    cmf, admm = dec.cmf_aoadmm(X, 3, n_iter_max=2, init=init, non_negative=True, update_C=False, tol=None,
                               absolute_tol=None, return_admm_vars=True)
    np.testing.assert_array_equal(cmf[1][2], init[1][2])
    assert admm.auxes[2] == [] and len(admm.auxes[0]) == 1


def test_cmf_from_cp_and_parafac2_tensors():
    """CoupledMatrixFactorization.from_CPTensor / from_Parafac2Tensor (coupled_matrices.py:101-172) on plain tuples"""
    from matcouply_amd.coupled_matrices import CoupledMatrixFactorization as CMF

    rs = np.random.RandomState(0)
    A, B, C = rs.uniform(size=(4, 3)), rs.uniform(size=(6, 3)), rs.uniform(size=(5, 3))
    T = np.einsum("ir,jr,kr->ijk", A, B, C)
    assert np.allclose(CMF.from_CPTensor((None, (A, B, C))).to_tensor(), T)
    c2 = CMF.from_CPTensor((np.ones(3), (A, B, C)), shapes=[(6, 5), (4, 5), (2, 5), (6, 5)])
    assert [m.shape for m in c2.to_matrices()] == [(6, 5), (4, 5), (2, 5), (6, 5)]
    assert np.allclose(c2.to_matrices()[1], T[1, :4])
    Ps = [np.linalg.qr(rs.standard_normal((7, 6)))[0] for _ in range(4)]
    c3 = CMF.from_Parafac2Tensor((None, (A, B, C), Ps))
    assert np.allclose(c3.to_matrices()[2], Ps[2] @ B @ np.diag(A[2]) @ C.T)
    for bad in (lambda: CMF.from_CPTensor((None, (A, B))), lambda: CMF.from_CPTensor((None, (A, B, C)), shapes=[(6, 5)]),
                lambda: CMF.from_CPTensor((None, (A, B, C)), shapes=[(6, 4)] * 4),
                lambda: CMF.from_CPTensor((None, (A, B, C)), shapes=[(9, 5)] * 4)):
        with pytest.raises(ValueError):
            bad()


def test_padded_tensor_utils_and_gated_datasets():
    """_utils.py:33-54 of the reference; the downloadable datasets raise a clear error instead of an AttributeError"""
    from matcouply_amd import data
    from matcouply_amd._utils import create_padded_tensor, get_padded_tensor_shape

    mats = [np.ones((2, 3)), 2 * np.ones((4, 3)), 3 * np.ones((1, 3))]
    assert get_padded_tensor_shape(mats) == (3, 4, 3)
    t = create_padded_tensor(mats)
    assert t.shape == (3, 4, 3) and t[0, 2:].sum() == 0 and t[1].sum() == 24 and t[2, 0, 0] == 3
    with pytest.raises(ValueError):
        create_padded_tensor([np.ones((2, 3)), np.ones((2, 4))])
    for fn in (data.get_bike_data, data.get_semiconductor_etch_raw_data, data.get_semiconductor_etch_machine_data):
        with pytest.raises(NotImplementedError):
            fn()


def test_host_diagnostic_helpers_of_the_reference():
    """_cmf_reconstruction_error (both forms, decomposition.py:420-452), _compute_l2_penalty (:617-627),
    _check_inner_convergence (:92-116) on host arrays, against direct evaluation and the phase goldens of the reference."""
    rng = np.random.RandomState(2)
    I, K, r = 4, 6, 3
    J = [5, 7, 4, 6]
    A, C = rng.uniform(0.1, 1, (I, r)), rng.uniform(size=(K, r))
    B = [rng.uniform(size=(j, r)) for j in J]
    mats = [(b * A[i]) @ C.T + 0.1 * rng.standard_normal((b.shape[0], K)) for i, b in enumerate(B)]
    cmf = (None, (A, B, C))
    direct = np.sqrt(sum(np.sum((m - (b * A[i]) @ C.T) ** 2) for i, (m, b) in enumerate(zip(mats, B))))
    np.testing.assert_allclose(dec._cmf_reconstruction_error(mats, cmf), direct, rtol=1e-10)
    norm = np.sqrt(sum(np.sum(m ** 2) for m in mats))
    np.testing.assert_allclose(dec._cmf_reconstruction_error(mats, cmf, norm), direct, rtol=1e-10)
    rhses = [np.diag(b.T @ m @ C) for b, m in zip(B, mats)]
    cross = [(b.T @ b) * (C.T @ C) for b in B]
    np.testing.assert_allclose(dec._cmf_reconstruction_error(mats, cmf, norm, (rhses, cross)), direct, rtol=1e-9)
    w = rng.uniform(0.5, 2, r)
    direct_w = np.sqrt(sum(np.sum((m - (b * (A[i] * w)) @ C.T) ** 2) for i, (m, b) in enumerate(zip(mats, B))))
    np.testing.assert_allclose(dec._cmf_reconstruction_error(mats, (w, (A, B, C))), direct_w, rtol=1e-10)
    l2 = dec._compute_l2_penalty(cmf, [0.5, None, 2.0])
    np.testing.assert_allclose(l2, 0.25 * np.sum(A ** 2) + np.sum(C ** 2))
    assert dec._compute_l2_penalty(cmf, [0, 0, 0]) == 0
    reg = pen.NonNegativity()
    assert dec._check_inner_convergence(A, A, cmf, [reg], [A.copy()], 0, 1e-3)
    assert not dec._check_inner_convergence(A, A + 1, cmf, [reg], [A.copy()], 0, 1e-3)
    assert not dec._check_inner_convergence(A, A, cmf, [reg], [A + 1.0], 0, 1e-3)
    assert not dec._check_inner_convergence(A, A, cmf, [reg], [A.copy()], 0, None)
    assert dec._check_inner_convergence(B, [b.copy() for b in B], cmf, [], [], 1, 1e-3)


def test_bench_counts_the_cores_it_may_use():
    """bench.py's cpu_baseline reports `cores` = the threads it actually uses: the affinity mask capped by the cgroup quota"""
    import bench

    n = bench.usable_cores()
    assert 1 <= n <= (os.cpu_count() or 1)
    assert n <= len(os.sched_getaffinity(0))


# ---- plugin surface: docstring helpers and subclass routing ------------------------------------------------------
def test_doc_utils_shim_inherits_docstrings():
    """the names user penalties import from the reference's matcouply._doc_utils (examples/plot_custom_penalty.py:213-231)"""
    from matcouply_amd._doc_utils import InheritableDocstrings, copy_ancestor_docstring

    class Mine(pen.HardConstraintMixin, pen.MatrixPenalty):
        @copy_ancestor_docstring
        def factor_matrix_update(self, factor_matrix, feasibility_penalty, aux):
            return factor_matrix

    assert type(Mine) is InheritableDocstrings
    assert Mine.factor_matrix_update.__doc__ == pen.MatrixPenalty.factor_matrix_update.__doc__
    assert Mine.factor_matrix_update.__doc__  # the ancestor documents the method
    with pytest.raises(RuntimeError, match="already has docstring"):
        @copy_ancestor_docstring
        def documented():
            """x"""
    with pytest.raises(RuntimeError, match="does not exist in superclass"):
        class Bad(pen.MatrixPenalty):
            @copy_ancestor_docstring
            def no_such_method(self):
                pass


def test_native_descriptor_only_for_unmodified_builtin_penalties():
    """a subclass that overrides the prox (or the penalty value) of a built-in penalty is evaluated on the host"""
    class Shifted(pen.NonNegativity):
        def factor_matrix_row_update(self, factor_matrix_row, feasibility_penalty, aux_row):
            return np.maximum(factor_matrix_row, 0.1)

    class Renamed(pen.NonNegativity):  # nothing of the contract overridden: still native
        pass

    class Valued(pen.L1Penalty):
        def penalty(self, x):
            return 0.0

    assert pen.native_descriptor_of(pen.NonNegativity()) == pen.NonNegativity()._native_descriptor()
    assert pen.native_descriptor_of(Renamed()) == pen.NonNegativity()._native_descriptor()
    assert pen.native_descriptor_of(Shifted()) is None
    assert pen.native_descriptor_of(Valued(0.1)) is None
    # round 5: UnitSimplex and GeneralizedL2Penalty have native kernels too (MCL_PEN_SIMPLEX / MCL_PEN_GL2) ...
    assert pen.native_descriptor_of(pen.UnitSimplex())[0] == 10 and pen.native_descriptor_of(pen.GeneralizedL2Penalty(np.eye(3)))[0] == 9

    class Bisected(pen.UnitSimplex):  # ... unless a subclass touches the contract
        def factor_matrix_update(self, factor_matrix, feasibility_penalty, aux):
            return factor_matrix

    assert pen.native_descriptor_of(Bisected()) is None


def test_substitute_engine_needs_the_test_switch(monkeypatch):
    """the checker-engine seam of decomposition.py is inert outside the test-suite"""
    monkeypatch.setattr(dec, "_ENGINE_FACTORY", OracleEngineFactory())
    monkeypatch.delenv("MATCOUPLY_AMD_TEST_ENGINE")
    X, _ = get_simple_simulated_data()
    with pytest.raises(RuntimeError, match="outside the test-suite"):
        dec.cmf_aoadmm(X, 3, n_iter_max=1)


def test_arithmetic_keyword_is_validated():
    X, _ = get_simple_simulated_data(random_state=0)
    with pytest.raises(ValueError, match="arithmetic"):
        dec.cmf_aoadmm(X, 3, n_iter_max=1, arithmetic="double")
