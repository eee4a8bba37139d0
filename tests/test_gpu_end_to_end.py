"""-m gpu: the public API (`cmf_aoadmm` -> ctypes -> libmatcouply_hip.so) on the MI355X vs
  (a) trajectories captured from the reference (tests/golden/traj_*.npz),
  (b) the oracle on seeded inputs at sizes it finishes in seconds (BASELINE configs 2-5, down-scaled where needed),
  (c) size-independent properties at the FULL size of BASELINE config 3.
Tolerance: a flat 1e-5 relative (Frobenius) on short fixed trajectories, penalty-free modes included, as BASELINE.json's
north_star states for the fp32 engine against the fp64 NumPy reference; the 20-iteration golden trajectories are held
to 3e-6 (5e-5 with a localisation proof where a discontinuous projection re-pools a column), rec_errors to 1e-5
throughout."""
import json
import os

import numpy as np
import pytest

from tests.helpers import GOLDEN, load_npz, rel_err, split_rows

pytestmark = pytest.mark.gpu
TRAJ = sorted(f for f in os.listdir(GOLDEN) if f.startswith("traj_") and "seeded" not in f)


def _penalty(d, aux_init, dual_init):
    from tests.test_host_api import make_penalty

    return make_penalty(d, aux_init=aux_init, dual_init=dual_init)


def _regs_from_state(st):
    """matcouply_amd penalties carrying an OracleState's aux/dual as explicit inits"""
    rp = st.row_ptr
    regs = [[], [], []]
    for m in range(3):
        for d, z, u in zip(st.regs[m], st.aux[m], st.dual[m]):
            if d["kind"] == "parafac2":
                aux = (split_rows(z[0], rp), z[1].copy())
            else:
                aux = split_rows(z, rp) if m == 1 else z.copy()
            dual = split_rows(u, rp) if m == 1 else u.copy()
            regs[m].append(_penalty(d, aux, dual))
    return regs


def _run_both(st, n_iter, **kw):
    """n_iter outer iterations on the GPU (public API) and in the oracle from the same explicit state"""
    from matcouply_amd import decomposition as dec
    from oracle import aoadmm_oracle as orc

    rp = st.row_ptr
    mats = split_rows(st.X, rp)
    cmf, admm, diag = dec.cmf_aoadmm(
        mats, st.A.shape[1], init=(None, (st.A.copy(), split_rows(st.B, rp), st.C.copy())), regs=_regs_from_state(st),
        n_iter_max=n_iter, tol=None, absolute_tol=None, return_errors=True, return_admm_vars=True,
        l2_penalty=list(st.l2), feasibility_penalty_scale=st.scale,
        constant_feasibility_penalty=(True if (st.constant_A and st.constant_B) else ("A" if st.constant_A else
                                                                                         ("B" if st.constant_B else False))),
        inner_n_iter_max=st.inner, **kw)
    orc.POLAR_COND["max"] = 1.0
    res = orc.run(st, n_iter, tol=None, absolute_tol=None)
    res["polar_cond"] = orc.POLAR_COND["max"]  # worst cond(Y_i Delta^T) on the oracle's trajectory
    return cmf, admm, diag, res


GAP_RTOL, GAP_ATOL = 1e-4, 2e-7


def gap_error(got_hist, ref_hist, rtol=GAP_RTOL):
    """Feasibility gaps ||aux - x|| / ||x|| (decomposition.py:351-417) of every iteration, penalty and mode against reference
    values: worst |got - ref| / (GAP_ATOL + GAP_RTOL |ref|), i.e. <= 1 passes.  A gap is a DIFFERENCE of two fp32-stored
    arrays over the norm of one of them: its absolute error sits at the storage level (6e-8 per element, relative to the
    factor) however small the gap itself is - hence the absolute floor beside the relative bar."""
    worst = 0.0
    assert len(got_hist) == len(ref_hist), (len(got_hist), len(ref_hist))
    for got_it, ref_it in zip(got_hist, ref_hist):
        for m in range(3):
            g, r = np.asarray([float(v) for v in got_it[m]]), np.asarray([float(v) for v in ref_it[m]])
            assert g.shape == r.shape, (m, g.shape, r.shape)
            if g.size:
                assert np.isfinite(g).all(), got_it
                worst = max(worst, float(np.max(np.abs(g - r) / (GAP_ATOL + rtol * np.abs(r)))))
    return worst


def _compare(cmf, admm, diag, st, res, tol, tol_rec=1e-5):
    errs = {"A": rel_err(cmf[1][0], st.A), "B": rel_err(np.concatenate(cmf[1][1]), st.B), "C": rel_err(cmf[1][2], st.C)}
    errs["rec"] = max(abs(a - b) / b for a, b in zip(diag.rec_errors, res["rec_errors"]))
    errs["loss"] = max(abs(a - b) / abs(b) for a, b in zip(diag.regularized_loss, res["losses"]))
    # the VALUES of the feasibility gaps the device's diagnostics tables produce, every iteration
    errs["gaps"] = gap_error(diag.feasibility_gaps, res["gaps"])
    for m in range(3):
        for k, d in enumerate(st.regs[m]):
            z, u = admm.auxes[m][k], admm.duals[m][k]
            if d["kind"] == "parafac2":
                # the orthogonal bases enter the iteration only through the auxiliary MATRIX P_i Delta (penalties.py:1303):
                # that product and Delta are held to the flat bar; P itself is a polar factor, whose forward error is
                # cond(Y_i Delta^T) x the float32 rounding of Y_i = B_i + U_i (1e-8 relative rms), whatever the engine
                P_ref, D_ref = st.aux[m][k]
                errs[f"PD{m}{k}"] = rel_err(np.concatenate(z[0]) @ np.asarray(z[1]), P_ref @ D_ref)
                errs[f"D{m}{k}"] = rel_err(z[1], D_ref)
                errs[f"P{m}{k}"] = rel_err(np.concatenate(z[0]), P_ref)
            else:
                errs[f"aux{m}{k}"] = rel_err(np.concatenate(z) if m == 1 else z, st.aux[m][k])
            # duals live on the scale of their factor and are ~0 where a constraint is inactive: measure their error
            # against max(||dual||, ||factor||) instead of dividing by a vanishing norm
            un = np.concatenate(u) if m == 1 else np.asarray(u)
            scale = max(np.linalg.norm(st.dual[m][k]), np.linalg.norm((st.A, st.B, st.C)[m]))
            errs[f"dual{m}{k}"] = np.linalg.norm(un - st.dual[m][k]) / scale
    # loss = rec^2 / 2 + penalties: its relative error is up to twice the rec error's
    p_tol = max(tol, 1e-8 * res.get("polar_cond", 1.0))
    bound = lambda k: 1.0 if k == "gaps" else (
        tol_rec if k == "rec" else (2 * tol_rec if k == "loss" else (p_tol if k[0] == "P" and k[1] != "D" else tol)))
    bad = {k: v for k, v in errs.items() if not (v < bound(k))}
    assert not bad, (bad, errs)
    return errs


@pytest.mark.parametrize("fname", TRAJ)
def test_golden_trajectories(fname, kernel_paths):
    from tests.test_oracle_golden import _traj_state

    arrs = load_npz(fname)
    spec = json.loads(str(arrs["spec"]))
    # 5 iterations against the (golden-pinned) oracle at 1e-5 ...
    st = _traj_state(arrs, spec)
    cmf, admm, diag, res = _run_both(st, 5)
    print(fname, "5 it:", {k: f"{v:.1e}" for k, v in _compare(cmf, admm, diag, st, res, 1e-5).items()})
    # ... and the full 20-iteration trajectory against the reference's own numbers
    st = _traj_state(arrs, spec)
    cmf, admm, diag, res = _run_both(st, spec["n_iter_max"])
    np.testing.assert_allclose(diag.rec_errors, arrs["rec_errors"], rtol=1e-5)
    np.testing.assert_allclose(diag.regularized_loss, arrs["regularized_loss"], rtol=1e-5)
    e = {"A": rel_err(cmf[1][0], arrs["A"]), "B": rel_err(np.concatenate(cmf[1][1]), arrs["B"]),
         "C": rel_err(cmf[1][2], arrs["C"])}
    # the reference's own feasibility gaps, all 21 read-outs of every penalty (gaps_m{m}: [n_iter + 1, n_regs of the mode])
    ref_gaps = [[arrs[f"gaps_m{m}"][it] if f"gaps_m{m}" in arrs else [] for m in range(3)] for it in range(len(arrs["rec_errors"]))]
    # (relative error of a gap = error of the factor / the gap: the full stack's free-running B sits at 1.2e-6 after 20 iterations
    # where the other six trajectories sit at 1e-7, its gaps at 0.04 .. 0.08 - held to 1e-3 there)
    e_gap = gap_error(diag.feasibility_gaps, ref_gaps, rtol=(1e-3 if "c5_full" in fname else GAP_RTOL))
    print(fname, "20 it vs reference:", {k: f"{v:.1e}" for k, v in e.items()}, f"gaps {e_gap:.2f} of the bar")
    if max(e.values()) < 3e-6:  # measured: <= 1e-7 on six of the seven trajectories, 1.2e-6 on the full stack (default path)
        assert e_gap <= 1.0, (fname, e_gap)
        return
    # Round 5: on the DEFAULT path (fp64 inner loops of small problems, csrc/wide.hip) every one of the seven free-running
    # trajectories stays inside that bar - the full stack included, whose re-pooled column the fp32 row arithmetic used to
    # tip.  What follows is the fast-kernels leg only.
    assert kernel_paths == "fast-kernels", (fname, e)
    # Above that the free-running trajectory has met a DISCONTINUITY of the reference's own map: the unimodal regression of
    # the full stack (traj_c5_full) pools a column differently when two candidate level sets are closer than the distance
    # the two trajectories have drifted apart by then (tools/traj_growth.py: B error 3e-6 at iteration 12, 2e-5 at 16, 98 % of
    # it in three of 45 (slab, column) pairs).  That is not held to a looser tolerance: it is PINNED as an equivalence -
    # from the engine's OWN state before every one of the 20 iterations, the reference arithmetic (oracle, fp64) takes the
    # step the engine took, to the flat 1e-5 bar on every factor and every ADMM variable.  Whatever split the engine chose,
    # the reference chooses on the same input; the trajectories differ only in which side of a tie their inputs fell.
    assert any(d["kind"] == "unimodal" for d in spec["regs"][1]), (fname, e)
    _assert_stepwise_equivalence(_traj_state(arrs, spec), spec["n_iter_max"], fname)


@pytest.mark.parametrize("fname", ["traj_c3_nn_l1C.npz", "traj_c4_pf2_ball.npz", "traj_c5_full.npz"])
def test_public_compute_feasibility_gaps(fname):
    """`compute_feasibility_gaps(cmf, regs, A_aux_list, B_aux_list, C_aux_list)` (reference decomposition.py:351-417) on the
    DEVICE tensors a run returns: equal to the gaps the engine's own diagnostics tables reported for that state, and - after
    the reference's 20 iterations - to the reference's last read-out."""
    import torch

    from matcouply_amd import decomposition as dec
    from tests.test_oracle_golden import _traj_state

    arrs = load_npz(fname)
    spec = json.loads(str(arrs["spec"]))
    st = _traj_state(arrs, spec)
    dev = torch.device("cuda", 0)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)
    rp = st.row_ptr
    regs = _regs_from_state(st)
    cmf, admm, diag = dec.cmf_aoadmm(
        [t(m) for m in split_rows(st.X, rp)], st.A.shape[1], init=(None, (t(st.A), [t(b) for b in split_rows(st.B, rp)], t(st.C))),
        regs=regs, n_iter_max=spec["n_iter_max"], tol=None, absolute_tol=None, return_errors=True, return_admm_vars=True,
        l2_penalty=list(st.l2), feasibility_penalty_scale=st.scale, constant_feasibility_penalty=bool(st.constant_A))
    assert cmf[1][0].is_cuda and all(b.is_cuda for b in cmf[1][1])  # device in, device out
    gaps = dec.compute_feasibility_gaps(cmf, regs, *admm.auxes)
    ref = [arrs[f"gaps_m{m}"][-1] for m in range(3)]
    e_self, e_ref = gap_error([gaps], [diag.feasibility_gaps[-1]]), gap_error([gaps], [ref])
    print(fname, "public gaps:", [[f"{float(g):.3e}" for g in m] for m in gaps], f"vs diagnostics {e_self:.2f}, vs reference {e_ref:.2f} of the bar")
    assert e_self <= 1.0
    if fname != "traj_c5_full.npz":  # (the full stack's free-running trajectory is held step-wise, see above)
        assert e_ref <= 1.0


def _state_from_gpu(st, cmf, admm):
    """a fresh OracleState holding the engine's factors and ADMM variables (fp32 values, exactly, as fp64)"""
    from oracle import aoadmm_oracle as orc

    aux, dual = [[], [], []], [[], [], []]
    for m in range(3):
        for d, z, u in zip(st.regs[m], admm.auxes[m], admm.duals[m]):
            if d["kind"] == "parafac2":
                aux[m].append((np.concatenate(z[0]), np.asarray(z[1])))
            else:
                aux[m].append(np.concatenate(z) if m == 1 else np.asarray(z))
            dual[m].append(np.concatenate(u) if m == 1 else np.asarray(u))
    return orc.OracleState(st.X, st.row_ptr, cmf[1][0], np.concatenate(cmf[1][1]), cmf[1][2], st.regs, aux, dual, l2=st.l2,
                           inner_n_iter_max=st.inner, feasibility_penalty_scale=st.scale, constant_A=st.constant_A,
                           constant_B=st.constant_B)


def _assert_stepwise_equivalence(st, n_iter, label):
    worst = {}
    for it in range(n_iter):
        cmf, admm, diag, res = _run_both(st, 1)        # engine and oracle, one iteration each, from the SAME state
        errs = _compare(cmf, admm, diag, st, res, 1e-5)  # `st` now holds the oracle's result
        for k, v in errs.items():
            worst[k] = max(worst.get(k, 0.0), v)
        st = _state_from_gpu(st, cmf, admm)              # ... and both continue from the ENGINE's state
    print(label, f"re-synchronised one-step parity over {n_iter} iterations, worst:", {k: f"{v:.1e}" for k, v in worst.items()})


def test_seeded_keyword_run_on_gpu():
    from matcouply_amd import decomposition as dec

    arrs = load_npz("traj_seeded_keywords.npz")
    c1 = load_npz("c1_data.npz")
    cmf, diag = dec.cmf_aoadmm(split_rows(c1["X"], c1["row_ptr"]), 3, non_negative=True, l1_penalty={2: 0.1},
                               l2_norm_bound={1: 1.0}, parafac2=True, n_iter_max=10, tol=None, absolute_tol=None,
                               return_errors=True, random_state=0)
    np.testing.assert_allclose(diag.rec_errors, arrs["rec_errors"], rtol=1e-5)
    assert rel_err(cmf[1][0], arrs["A"]) < 1e-5 and rel_err(np.concatenate(cmf[1][1]), arrs["B"]) < 1e-5


def test_readme_example_of_the_reference_on_gpu():
    """README.rst:66-91 of the reference through the public API on the device: L2 ball on A with a constant feasibility
    penalty, PARAFAC2 + unimodality + L2 ball on the B_i, L1 on C, seeded random initialisation."""
    from matcouply_amd import decomposition as dec

    arrs = load_npz("readme_example.npz")
    c1 = load_npz("c1_data.npz")
    cmf, diag = dec.cmf_aoadmm(split_rows(c1["X"], c1["row_ptr"]), 3, non_negative=True, l1_penalty={2: 0.1},
                               l2_norm_bound=[1, 1, 0], parafac2=True, unimodal={1: True}, constant_feasibility_penalty=True,
                               n_iter_max=10, tol=None, absolute_tol=None, return_errors=True, random_state=0)
    np.testing.assert_allclose(diag.rec_errors, arrs["rec_errors"], rtol=1e-5)
    np.testing.assert_allclose(diag.regularized_loss, arrs["regularized_loss"], rtol=1e-5)
    e = {"A": rel_err(cmf[1][0], arrs["A"]), "B": rel_err(np.concatenate(cmf[1][1]), arrs["B"]), "C": rel_err(cmf[1][2], arrs["C"])}
    print("README example, 10 it vs reference:", {k: f"{v:.1e}" for k, v in e.items()})
    assert max(e.values()) < 1e-5, e


def test_config1_converges_like_the_reference():
    from matcouply_amd import decomposition as dec
    from matcouply_amd.data import get_simple_simulated_data

    with open(os.path.join(GOLDEN, "c1_known_answer.json")) as f:
        ref = json.load(f)
    X, _ = get_simple_simulated_data(noise_level=0.2, random_state=1)
    cmf, diag = dec.parafac2_aoadmm(X, 3, non_negative=True, random_state=0, return_errors=True)
    assert diag.message == ref["message"]
    # The default tol=1e-8 on the relative loss change sits at the resolution of fp32 STATE, so rounds 1-4 only asked for the same
    # rule in the same regime (185 iterations against the reference's 219 was accepted as 100..400).  Since round 5 a problem of
    # this size runs its inner loops in fp64 (csrc/wide.hip) and the rule fires at the reference's own iteration: 219 measured;
    # held to +-3 (the fp32 storage of the state between phases is still there), the final error to 1e-7.
    assert abs(diag.n_iter - ref["n_iter"]) <= 3, (diag.n_iter, ref["n_iter"])
    np.testing.assert_allclose(diag.rec_errors[-1], ref["final_rec_error"], rtol=1e-7)
    np.testing.assert_allclose(diag.regularized_loss[-1], ref["final_loss"], rtol=1e-6)


SCALE_CASES = {
    # BASELINE configs at sizes the oracle finishes in seconds
    "c2_full": dict(I=256, J=256, K=128, r=8, regs=[[{"kind": "nn"}], [{"kind": "nn"}], [{"kind": "nn"}]]),
    "c3_quarter": dict(I=256, J=512, K=256, r=16,
                       regs=[[{"kind": "nn"}], [{"kind": "nn"}], [{"kind": "l1", "reg_strength": 0.1, "non_negativity": True}]]),
    # config 4 down-scaled (7 M elements): A and C carry no penalty and the A systems have condition up to 1.5e5 - since round 6
    # the default call moves such a mid-size problem to the exact arithmetic by itself (mcl_condition_probe); the fast kernels
    # forced onto it (what rounds 1-5 tested here: A at 4.0 / 6.6 / 9.4e-6 in three builds that differ in the association of
    # fp64 sums) keep their own case with the bar that margin warrants.  (Exact arithmetic: A at 3.0e-6 - the fp32 STORAGE of B
    # between the phases in front of systems of condition 1.5e5; every sum of that path runs in a fixed order.)
    "c4_ragged": dict(I=48, J="ragged", K=256, r=16, regs=[[], [{"kind": "parafac2"}, {"kind": "l2ball", "norm_bound": 1.0}], []],
                      tol=5e-6),
    "c4_ragged_fast": dict(I=48, J="ragged", K=256, r=16, regs=[[], [{"kind": "parafac2"}, {"kind": "l2ball", "norm_bound": 1.0}], []],
                           tol=3e-5, arithmetic="fast"),
    "c5_stack": dict(I=24, J=160, K=192, r=32,
                     regs=[[{"kind": "nn"}],
                           [{"kind": "parafac2"}, {"kind": "unimodal", "non_negativity": True},
                            {"kind": "l2ball", "norm_bound": 1.0, "non_negativity": True}],
                           [{"kind": "l1", "reg_strength": 0.1, "non_negativity": True}]]),
    # config 5's matrix dimensions (K = 1024, rank 32: K-sliced X^T pass, fragment-streaming X C pass) with few slabs
    "c5_dims": dict(I=6, J="c5dims", K=1024, r=32,
                    regs=[[{"kind": "nn"}], [{"kind": "nn"}], [{"kind": "l1", "reg_strength": 0.1, "non_negativity": True}]]),
    # ... and the same dimensions with config 5's FULL penalty stack (PARAFAC2 + unimodality + L2 ball on the B_i, NN on A,
    # L1 on C): the combination BASELINE config 5 actually is
    # (slab heights >= 3 r, as in config 5 itself - a PARAFAC2 slab barely taller than the rank has a nearly singular
    # Y_i Delta^T, cond 2e4 with J_i = 33: see the polar-factor note in _compare)
    "c5_dims_stack": dict(I=6, J="c5stack", K=1024, r=32,
                          regs=[[{"kind": "nn"}],
                                [{"kind": "parafac2"}, {"kind": "unimodal", "non_negativity": True},
                                 {"kind": "l2ball", "norm_bound": 1.0, "non_negativity": True}],
                                [{"kind": "l1", "reg_strength": 0.1, "non_negativity": True}]]),
    "k512": dict(I=10, J=300, K=512, r=16,
                 regs=[[{"kind": "nn"}], [{"kind": "parafac2"}, {"kind": "nn"}], [{"kind": "nn"}]]),
    "odd_shapes": dict(I=9, J="odd", K=37, r=5, regs=[[{"kind": "box", "min_val": 0.0, "max_val": 0.9}],
                                                      [{"kind": "l1", "reg_strength": 0.05}], [{"kind": "nn"}]]),
    "r64": dict(I=8, J=96, K=80, r=64, regs=[[{"kind": "nn"}], [{"kind": "nn"}], []]),
    # the ragged slabs of config 4 at FULL size (I = 1024, 590 K rows): more work units than waves, so the planner cuts
    # bsegs / segments at the waves' quotas (csrc/api.hip) - config 3's penalties through the sweep, config 4 itself through
    # the two X passes
    "c3_ragged_full": dict(I=1024, J="ragged", K=256, r=16,
                           regs=[[{"kind": "nn"}], [{"kind": "nn"}], [{"kind": "l1", "reg_strength": 0.1, "non_negativity": True}]]),
    "c4_full": dict(I=1024, J="ragged", K=256, r=16, regs=[[], [{"kind": "parafac2"}, {"kind": "l2ball", "norm_bound": 1.0}], []]),
}


@pytest.mark.parametrize("name", sorted(SCALE_CASES))
def test_scale_parity_vs_oracle(name):
    from oracle import aoadmm_oracle as orc

    cfg = SCALE_CASES[name]
    J = cfg["J"]
    if J == "ragged":
        J = np.random.RandomState(0).randint(128, 1025, cfg["I"])
    elif J == "c5dims":
        J = np.array([2048, 700, 33, 1024, 515, 64])
    elif J == "c5stack":
        J = np.array([2048, 700, 100, 1024, 515, 130])
    elif J == "odd":
        J = np.array([1, 3, 64, 65, 17, 130, 5, 63, 2])
    X, row_ptr = orc.synthetic_problem(cfg["I"], J, cfg["K"], cfg["r"], seed=0, dtype=np.float64)
    X = X.astype(np.float32).astype(np.float64)  # the engine stores X in fp32: give both sides identical data
    st = orc.random_state_for(X, row_ptr, cfg["r"], cfg["regs"], seed=1)
    # flat bar, penalty-free modes included: their un-shifted normal equations are built and solved in fp64
    # ([G | R], the per-slab Grams and right-hand sides carry fp64 across tiles)
    kw = {"arithmetic": cfg["arithmetic"]} if "arithmetic" in cfg else {}
    cmf, admm, diag, res = _run_both(st, 2 if name.endswith("_full") else 3, **kw)
    errs = _compare(cmf, admm, diag, st, res, cfg.get("tol", 1e-5), min(1e-5, cfg.get("tol", 1e-5) * 5))
    print(name, {k: f"{v:.1e}" for k, v in errs.items()})


def test_empty_and_degenerate_inputs():
    """ragged edge cases the reference's fixtures exercise: a slab shorter than the rank, single-row slabs, zero
    iterations, inner_n_iter_max = 1, frozen modes."""
    from matcouply_amd import decomposition as dec
    from oracle import aoadmm_oracle as orc

    X, row_ptr = orc.synthetic_problem(5, np.array([2, 1, 7, 3, 4]), 6, 3, seed=2, dtype=np.float64)
    X = X.astype(np.float32).astype(np.float64)
    nn = {"kind": "nn"}
    st = orc.random_state_for(X, row_ptr, 3, [[nn], [nn], [nn]], seed=3, inner_n_iter_max=1)
    cmf, admm, diag, res = _run_both(st, 2)
    _compare(cmf, admm, diag, st, res, 1e-5)
    st = orc.random_state_for(X, row_ptr, 3, [[nn], [nn], [nn]], seed=3)
    A0 = st.A.copy()
    cmf, admm, diag, res = _run_both(st, 0)
    assert diag.n_iter == 0 and len(diag.rec_errors) == 1 and rel_err(cmf[1][0], A0) < 1e-7
    np.testing.assert_allclose(diag.rec_errors[0], res["rec_errors"][0], rtol=1e-5)
    # frozen C: the reference drops mode-2 penalties and never touches C
    st = orc.random_state_for(X, row_ptr, 3, [[nn], [nn], []], seed=4)
    C0 = st.C.copy()
    mats = split_rows(st.X, row_ptr)
    cmf = dec.cmf_aoadmm(mats, 3, init=(None, (st.A.copy(), split_rows(st.B, row_ptr), st.C.copy())),
                         regs=_regs_from_state(st), n_iter_max=3, tol=None, absolute_tol=None, update_C=False)
    for _ in range(3):
        st.update_B(); st.update_A()
    assert rel_err(cmf[1][2], C0) < 1e-7 and rel_err(cmf[1][0], st.A) < 1e-5 and rel_err(np.concatenate(cmf[1][1]), st.B) < 1e-5


def test_full_size_config3_properties():
    """BASELINE config 3 at FULL size (I=1024, J=512, K=256, r=16): properties that need no CPU reference."""
    import torch
    import bench
    from matcouply_amd import decomposition as dec
    from matcouply_amd import penalties as pen

    cfg = bench.CONFIGS["c3"]
    dev = torch.device("cuda", 0)
    X, row_ptr, I = bench.make_shard(cfg, 0, 1, dev)
    packed = dec.PackedMatrices(X, row_ptr)

    def run(perm=None):
        Xp, rp = X, row_ptr
        if perm is not None:
            Xp = X.view(I, cfg["J"], cfg["K"])[torch.as_tensor(perm, device=dev)].reshape(-1, cfg["K"]).contiguous()
        return dec.cmf_aoadmm(dec.PackedMatrices(Xp, rp), cfg["r"], non_negative=True, l1_penalty={2: 0.1}, n_iter_max=6,
                              tol=None, absolute_tol=None, return_errors=True, return_admm_vars=True, random_state=0,
                              aux_init="zeros", dual_init="zeros", init=init)

    g = torch.Generator(device="cpu").manual_seed(0)
    A0 = torch.rand((I, cfg["r"]), generator=g).to(dev)
    C0 = torch.rand((cfg["K"], cfg["r"]), generator=g).to(dev)
    B0 = torch.rand((I, cfg["J"], cfg["r"]), generator=g).to(dev)
    init = (None, (A0, [B0[i] for i in range(I)], C0))
    cmf1, admm1, diag1 = run()
    cmf2, admm2, diag2 = run()
    # 1. bitwise determinism (fixed summation orders, no float atomics)
    assert torch.equal(cmf1[1][0], cmf2[1][0]) and torch.equal(cmf1[1][2], cmf2[1][2])
    assert all(torch.equal(a, b) for a, b in zip(cmf1[1][1], cmf2[1][1]))
    assert diag1.rec_errors == diag2.rec_errors
    # 2. the auxiliary variables satisfy their constraints exactly
    assert float(admm1.auxes[0][0].min()) >= 0 and float(admm1.auxes[2][0].min()) >= 0
    assert min(float(z.min()) for z in admm1.auxes[1][0]) >= 0
    # 3. the fast error formula (no pass over X) equals the explicit ||X - M|| / ||X|| computed in fp64
    A, B, C = cmf1[1][0].double(), torch.stack(cmf1[1][1]).double(), cmf1[1][2].double()
    M = torch.einsum("ijr,ir,kr->ijk", B, A, C).reshape(-1, cfg["K"])
    explicit = float(torch.linalg.norm(X.double() - M) / torch.linalg.norm(X.double()))
    np.testing.assert_allclose(diag1.rec_errors[-1], explicit, rtol=1e-4)
    # 4. the regularised loss decreases monotonically after the first iteration on this problem
    losses = diag1.regularized_loss
    assert all(b <= a * (1 + 1e-6) for a, b in zip(losses[2:], losses[3:])), losses
    # 5. permuting the slabs permutes A's rows / the B_i and leaves C and the errors unchanged (summation order only)
    perm = np.random.RandomState(0).permutation(I)
    A0p, B0p = A0[torch.as_tensor(perm, device=dev)], B0[torch.as_tensor(perm, device=dev)]
    init = (None, (A0p, [B0p[i] for i in range(I)], C0))
    cmf3, admm3, diag3 = run(perm)
    np.testing.assert_allclose(diag3.rec_errors, diag1.rec_errors, rtol=1e-5)
    assert rel_err(cmf3[1][2].cpu().numpy(), cmf1[1][2].cpu().numpy()) < 1e-5
    assert rel_err(cmf3[1][0].cpu().numpy(), cmf1[1][0].cpu().numpy()[perm]) < 1e-5


def test_full_size_config4_properties():
    """BASELINE config 4 at FULL size (I=1024 ragged J_i in [128, 1024], K=256, r=16, parafac2 + L2 ball on the B_i):
    size-independent properties - determinism, the constraints on the auxiliary variables, the PARAFAC2 structure, the fast
    error formula, and equality of the chained row passes with the two-pass-per-inner-iteration form of the same loop."""
    import torch
    import bench
    from matcouply_amd import decomposition as dec

    cfg = bench.CONFIGS["c4"]
    dev = torch.device("cuda", 0)
    X, row_ptr, I = bench.make_shard(cfg, 0, 1, dev)
    r = cfg["r"]

    def run():
        return dec.cmf_aoadmm(dec.PackedMatrices(X, row_ptr), r, parafac2=True, l2_norm_bound={1: 1.0}, n_iter_max=4,
                              tol=None, absolute_tol=None, return_errors=True, return_admm_vars=True, random_state=0)

    cmf1, admm1, diag1 = run()
    cmf2, admm2, diag2 = run()
    # 1. bitwise determinism
    assert torch.equal(cmf1[1][0], cmf2[1][0]) and torch.equal(cmf1[1][2], cmf2[1][2]) and diag1.rec_errors == diag2.rec_errors
    assert all(torch.equal(a, b) for a, b in zip(cmf1[1][1], cmf2[1][1]))
    # 2. constraints hold on the auxiliary variables: P_i^T P_i = I, shared cross product, column norms <= 1
    P_is, Delta = admm1.auxes[1][0]
    worst = max(float((P.double().T @ P.double() - torch.eye(r, device=dev, dtype=torch.float64)).abs().max()) for P in P_is[:64])
    assert worst < 1e-5, worst
    ball = admm1.auxes[1][1]
    assert max(float(torch.linalg.norm(z.double(), dim=0).max()) for z in ball) <= 1 + 1e-5
    assert all(bool(torch.isfinite(f).all()) for f in (cmf1[1][0], cmf1[1][2], Delta))
    # 3. fast error formula == explicit fp64 residual
    A, C = cmf1[1][0].double(), cmf1[1][2].double()
    num = sum(float(torch.linalg.norm(X[row_ptr[i]:row_ptr[i + 1]].double() - (cmf1[1][1][i].double() * A[i]) @ C.T) ** 2)
              for i in range(I))
    explicit = np.sqrt(num) / float(torch.linalg.norm(X.double()))
    np.testing.assert_allclose(diag1.rec_errors[-1], explicit, rtol=1e-4)
    # 4. the chained row passes (finish of inner iteration t + solve of t + 1 in one kernel), the statistics sums inside the
    #    Newton-Schulz kernel and the merged sum + Delta kernel are re-organisations with the same arithmetic in the same
    #    order: the un-chained form must give the same iterates
    def run_with(env_keys):
        saved = {k: os.environ.get(k) for k in env_keys}
        try:
            for k in env_keys:
                os.environ[k] = "1"
            return run()
        finally:
            for k, v in saved.items():
                os.environ.pop(k, None)
                if v is not None:
                    os.environ[k] = v

    cmf3, admm3, diag3 = run_with(("MCL_NO_PASS_CHAIN", "MCL_STATS_REDUCE", "MCL_NO_PF2_DELTA_FUSION"))
    # (the penalty-free A and C of this configuration amplify last-bit differences - the two forms contract their
    # multiply-adds differently - by the condition number of their normal equations: 5e-6 on the error after 4 iterations)
    np.testing.assert_allclose(diag3.rec_errors, diag1.rec_errors, rtol=2e-5)
    assert rel_err(cmf3[1][2].cpu().numpy(), cmf1[1][2].cpu().numpy()) < 2e-4
    assert rel_err(cmf3[1][0].cpu().numpy(), cmf1[1][0].cpu().numpy()) < 2e-4
    # 5. plain instead of minimax-scaled Newton-Schulz steps: another route to the same polar factors (1e-8 apart); A and C
    #    carry no penalty in this configuration, so their un-shifted normal equations amplify that difference
    cmf4, admm4, diag4 = run_with(("MCL_NS_PLAIN",))
    np.testing.assert_allclose(diag4.rec_errors, diag1.rec_errors, rtol=5e-5)
    assert rel_err(cmf4[1][2].cpu().numpy(), cmf1[1][2].cpu().numpy()) < 5e-4
    P4 = admm4.auxes[1][0][0]
    assert max(rel_err(a.cpu().numpy(), b.cpu().numpy()) for a, b in zip(P4[:32], P_is[:32])) < 5e-4


def test_config5_stack_properties_at_scale():
    """The full penalty stack of BASELINE config 5 (NN on A; PARAFAC2 + unimodality + L2 ball on the B_i; L1 on C) at
    I=1024, J=512, K=256, r=32 (config 5 itself needs 128 GB: tools/run_c5_full.py / bench.py --config c5): the constraints
    hold exactly on the auxiliary variables, runs are deterministic, and the two organisations of the unimodal kernel
    (one lane per column / sweeps split over waves) produce the same iterates."""
    import torch
    import bench
    from matcouply_amd import decomposition as dec

    cfg = bench.CONFIGS["c5s"]
    dev = torch.device("cuda", 0)
    X, row_ptr, I = bench.make_shard(cfg, 0, 1, dev)
    r = cfg["r"]

    def run(split=None):
        saved = os.environ.get("MCL_UNI_SPLIT")
        try:
            if split is not None:
                os.environ["MCL_UNI_SPLIT"] = split
            return dec.cmf_aoadmm(dec.PackedMatrices(X, row_ptr), r, non_negative=True, l1_penalty={2: 0.1},
                                  l2_norm_bound={1: 1.0}, unimodal={1: True}, parafac2=True, n_iter_max=3, tol=None,
                                  absolute_tol=None, return_errors=True, return_admm_vars=True, random_state=0)
        finally:
            os.environ.pop("MCL_UNI_SPLIT", None)
            if saved is not None:
                os.environ["MCL_UNI_SPLIT"] = saved

    cmf1, admm1, diag1 = run()
    cmf2, admm2, diag2 = run()
    assert torch.equal(cmf1[1][0], cmf2[1][0]) and torch.equal(cmf1[1][2], cmf2[1][2]) and diag1.rec_errors == diag2.rec_errors
    (P_is, Delta), uni, ball = admm1.auxes[1]
    eye = torch.eye(r, device=dev, dtype=torch.float64)
    assert max(float((P.double().T @ P.double() - eye).abs().max()) for P in P_is[:64]) < 1e-5
    assert max(float(torch.linalg.norm(z.double(), dim=0).max()) for z in ball) <= 1 + 1e-5
    assert min(float(z.min()) for z in ball) >= 0 and min(float(z.min()) for z in uni) >= 0
    assert float(admm1.auxes[0][0].min()) >= 0 and float(admm1.auxes[2][0].min()) >= 0
    # unimodal: once a column starts to decrease it never increases again
    U = torch.stack(uni[:128])
    dif = torch.sign(U[:, 1:] - U[:, :-1])
    dec_seen = torch.cummax((dif < 0).int(), dim=1).values
    assert int(((dif > 0) & (dec_seen == 1)).sum()) == 0
    assert all(np.isfinite(diag1.rec_errors))
    # both forms of the unimodal kernel (MCL_UNI_SPLIT=0: one lane per column; =1: sweeps split over waves)
    cmf_a, admm_a, diag_a = run("0")
    cmf_b, admm_b, diag_b = run("1")
    assert all(torch.equal(a, b) for a, b in zip(admm_a.auxes[1][1], admm_b.auxes[1][1]))
    assert diag_a.rec_errors == diag_b.rec_errors and torch.equal(cmf_a[1][2], cmf_b[1][2])
