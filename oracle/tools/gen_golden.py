#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE UNMODIFIED REFERENCE.

Run only in the build container (the reference is not present on the GPU box):

    python oracle/tools/gen_golden.py

The script puts `/root/reference/src` and the build's own NumPy stand-in for the absent `tensorly`
dependency (`oracle/tools/tlshim`) on `sys.path` *for this process only*, calls
the reference's functions, and writes inputs + expected outputs as fp64 `.npz` files.  Nothing from
the reference's source text is stored; fixtures are data only.

Fixture families:
  G1  phase_{A,B,C}.npz   one call of admm_update_{A,B,C} (decomposition.py:120-344) per case
  G2  traj_*.npz          20-iteration cmf_aoadmm trajectories on config-1 data for 5 penalty stacks
                          + one seeded `init="random"` run pinning the RNG draw order
  G3  prox.npz            each penalty's prox on standard-normal inputs; unimodal regression vectors
  G4  stopping.json       stopping-message / n_iter / list-length matrix (decomposition.py:990-1100)
  G5  more_penalties.npz  GeneralizedL2Penalty / UnitSimplex: prox vectors and a 10-iteration
                          trajectory with a graph-Laplacian penalty on the B_i and the unit simplex on C
                          (`python oracle/tools/gen_golden.py --only more` regenerates just this family)
  G7  converters_inits.npz  random_coupled_matrices (random.py:9-66), the dense converters cmf_to_matrices / tensor /
                          unfolded / vec (coupled_matrices.py:365-799) on its output, and init="svd" / "threshold_svd"
                          (decomposition.py:41-54): initial factors + a 5-iteration trajectory (`--only converters`)
  G6  readme_example.npz   the call of the reference's README (README.rst:66-91: non_negative, L1 on C, L2 balls on
                          A and B_i, PARAFAC2, unimodality, constant feasibility penalty, random_state=0), 10 iterations
                          (`--only readme`)

Penalties are described by neutral JSON descriptors ({"kind": "l1", "reg_strength": 0.1, ...}) so
that the fixtures do not depend on any class of the reference or of the product.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF_SRC = "/root/reference/src"
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(HERE, "tlshim"))
sys.path.insert(0, REF_SRC)

import matcouply  # noqa: E402  (the reference, over the stand-in)
from matcouply import decomposition as ref_dec  # noqa: E402
from matcouply import penalties as ref_pen  # noqa: E402
from matcouply._unimodal_regression import unimodal_regression as ref_unimodal  # noqa: E402
from matcouply._utils import get_svd  # noqa: E402
from matcouply.data import get_simple_simulated_data  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
SVD = get_svd("truncated_svd")


# ----------------------------------------------------------------------------------------------
# descriptor -> reference penalty instance
# ----------------------------------------------------------------------------------------------
def make_ref_penalty(desc, aux_init="random_uniform", dual_init="random_uniform"):
    kind = desc["kind"]
    kw = dict(aux_init=aux_init, dual_init=dual_init)
    if kind == "nn":
        return ref_pen.NonNegativity(**kw)
    if kind == "box":
        return ref_pen.Box(desc["min_val"], desc["max_val"], **kw)
    if kind == "l1":
        return ref_pen.L1Penalty(desc["reg_strength"], non_negativity=desc.get("non_negativity", False), **kw)
    if kind == "l2ball":
        return ref_pen.L2Ball(desc["norm_bound"], non_negativity=desc.get("non_negativity", False), **kw)
    if kind == "unimodal":
        return ref_pen.Unimodality(non_negativity=desc.get("non_negativity", False), **kw)
    if kind == "parafac2":
        return ref_pen.Parafac2(**kw)
    if kind == "gl2":
        return ref_pen.GeneralizedL2Penalty(np.asarray(desc["norm_matrix"], dtype=float), **kw)
    if kind == "simplex":
        return ref_pen.UnitSimplex(**kw)
    raise ValueError(kind)


def split_rows(packed, row_ptr):
    return [np.array(packed[row_ptr[i] : row_ptr[i + 1]]) for i in range(len(row_ptr) - 1)]


def pack_rows(mats):
    return np.concatenate([np.asarray(m) for m in mats], axis=0)


NN = {"kind": "nn"}
L1 = {"kind": "l1", "reg_strength": 0.1, "non_negativity": False}
L1NN = {"kind": "l1", "reg_strength": 0.1, "non_negativity": True}
BOX = {"kind": "box", "min_val": 0.1, "max_val": 0.6}
BALL = {"kind": "l2ball", "norm_bound": 0.5, "non_negativity": False}
BALLNN = {"kind": "l2ball", "norm_bound": 0.5, "non_negativity": True}
BALL1NN = {"kind": "l2ball", "norm_bound": 1.0, "non_negativity": True}
UNI = {"kind": "unimodal", "non_negativity": False}
UNINN = {"kind": "unimodal", "non_negativity": True}
PF2 = {"kind": "parafac2"}


# ----------------------------------------------------------------------------------------------
# G1: phase goldens
# ----------------------------------------------------------------------------------------------
def gen_phase():
    rng = np.random.RandomState(20241008)
    I, K, r = 6, 7, 3
    J = np.array([5, 12, 8, 9, 6, 11])
    row_ptr = np.concatenate([[0], np.cumsum(J)]).astype(np.int64)
    N = int(row_ptr[-1])
    A_true = rng.uniform(0.1, 1.1, size=(I, r))
    B_true = rng.uniform(size=(N, r))
    C_true = rng.uniform(size=(K, r))
    X = np.concatenate(
        [(B_true[row_ptr[i] : row_ptr[i + 1]] * A_true[i]) @ C_true.T for i in range(I)], axis=0
    ) + 0.05 * rng.standard_normal((N, K))
    A0 = rng.uniform(size=(I, r))
    B0 = rng.uniform(size=(N, r))
    C0 = rng.uniform(size=(K, r))
    n_slots = 3
    # aux/dual pools, one entry per reg slot; PARAFAC2 aux = (orthonormal P_i, Delta)
    pool = {}
    for s in range(n_slots):
        pool[f"auxA{s}"] = rng.uniform(size=(I, r))
        pool[f"dualA{s}"] = rng.uniform(size=(I, r))
        pool[f"auxB{s}"] = rng.uniform(size=(N, r))
        pool[f"dualB{s}"] = rng.uniform(size=(N, r))
        pool[f"auxC{s}"] = rng.uniform(size=(K, r))
        pool[f"dualC{s}"] = rng.uniform(size=(K, r))
    P0 = pack_rows([np.linalg.qr(rng.standard_normal((j, r)))[0] for j in J])
    Delta0 = rng.uniform(size=(r, r))

    shared = dict(X=X, row_ptr=row_ptr, A=A0, B=B0, C=C0, P0=P0, Delta0=Delta0, **pool)
    matrices = split_rows(X, row_ptr)

    def run_case(mode, descs, const, scale, l2, inner=5):
        regs = [make_ref_penalty(d) for d in descs]
        A, C = A0.copy(), C0.copy()
        B_is = split_rows(B0, row_ptr)
        cmf = (None, [A, B_is, C])
        out = {}
        if mode == 1:
            aux_list, dual_list = [], []
            for s, d in enumerate(descs):
                if d["kind"] == "parafac2":
                    aux_list.append((split_rows(P0, row_ptr), Delta0.copy()))
                else:
                    aux_list.append(split_rows(pool[f"auxB{s}"], row_ptr))
                dual_list.append(split_rows(pool[f"dualB{s}"], row_ptr))
            (_, (A_, B_, C_)), aux_o, dual_o = ref_dec.admm_update_B(
                matrices, regs, cmf, aux_list, dual_list, l2, inner, None, scale, const, SVD
            )
            out["factor"] = pack_rows(B_)
            for s, d in enumerate(descs):
                if d["kind"] == "parafac2":
                    out[f"aux{s}_P"] = pack_rows(aux_o[s][0])
                    out[f"aux{s}_Delta"] = np.asarray(aux_o[s][1])
                else:
                    out[f"aux{s}"] = pack_rows(aux_o[s])
                out[f"dual{s}"] = pack_rows(dual_o[s])
        elif mode == 2:
            aux_list = [pool[f"auxC{s}"].copy() for s in range(len(descs))]
            dual_list = [pool[f"dualC{s}"].copy() for s in range(len(descs))]
            (_, (A_, B_, C_)), aux_o, dual_o = ref_dec.admm_update_C(
                matrices, regs, cmf, aux_list, dual_list, l2, inner, None, scale, SVD
            )
            out["factor"] = np.asarray(C_)
            for s in range(len(descs)):
                out[f"aux{s}"] = np.asarray(aux_o[s])
                out[f"dual{s}"] = np.asarray(dual_o[s])
        else:
            aux_list = [pool[f"auxA{s}"].copy() for s in range(len(descs))]
            dual_list = [pool[f"dualA{s}"].copy() for s in range(len(descs))]
            (_, (A_, B_, C_)), aux_o, dual_o, (rhses, cross) = ref_dec.admm_update_A(
                matrices, regs, cmf, aux_list, dual_list, l2, inner, None, scale, const, SVD
            )
            out["factor"] = np.asarray(A_)
            out["rhses"] = np.stack(rhses)
            out["cross_products"] = np.stack(cross)
            for s in range(len(descs)):
                out[f"aux{s}"] = np.asarray(aux_o[s])
                out[f"dual{s}"] = np.asarray(dual_o[s])
        return out

    rowsep_sets = [[], [NN], [L1], [L1NN], [BOX]]
    matrix_sets = [[BALL], [BALLNN], [NN, BALL], [UNI], [UNINN]]
    pf2_sets = [[PF2], [PF2, NN], [PF2, UNINN, BALL1NN]]
    variations = [(0.5, 1.0), (2.0, 0.3)]

    for mode, name in ((1, "B"), (2, "C"), (0, "A")):
        cases = []
        if mode == 1:
            for descs in rowsep_sets + matrix_sets + pf2_sets:
                for const in (False, True):
                    cases.append((descs, const, 1.0, 0.0))
            for descs in ([NN], [PF2], [NN, BALL], []):
                for scale, l2 in variations:
                    cases.append((descs, False, scale, l2))
            cases.append(([NN], False, 1.0, 0.0, 1))  # single inner iteration
        elif mode == 2:
            for descs in rowsep_sets + matrix_sets:
                cases.append((descs, False, 1.0, 0.0))
            for descs in ([L1NN], [NN, BALL], []):
                for scale, l2 in variations:
                    cases.append((descs, False, scale, l2))
        else:
            for descs in rowsep_sets:
                for const in (False, True):
                    cases.append((descs, const, 1.0, 0.0))
            for descs in matrix_sets:
                cases.append((descs, True, 1.0, 0.0))  # matrix penalties on A need a constant rho
            for descs in ([NN], [L1], []):
                for scale, l2 in variations:
                    cases.append((descs, False, scale, l2))
            cases.append(([NN, BALL], True, 2.0, 0.3))
        arrays = dict(shared) if mode == 1 else {}
        manifest = []
        for ci, case in enumerate(cases):
            descs, const, scale, l2 = case[:4]
            inner = case[4] if len(case) > 4 else 5
            out = run_case(mode, descs, const, scale, l2, inner)
            manifest.append(dict(regs=descs, constant=const, scale=scale, l2=l2, inner=inner))
            for k, v in out.items():
                arrays[f"c{ci}_{k}"] = v
        arrays["manifest"] = np.array(json.dumps(manifest))
        np.savez_compressed(os.path.join(OUT, f"phase_{name}.npz"), **arrays)
        print(f"phase_{name}: {len(cases)} cases")


# ----------------------------------------------------------------------------------------------
# G2: trajectories on config-1 data
# ----------------------------------------------------------------------------------------------
TRAJ_STACKS = {
    # down-scaled penalty stacks of BASELINE configs 1-5 (per mode: list of descriptors)
    "c1_pf2_nn": dict(regs=[[NN], [PF2, NN], [NN]], kwargs=dict()),
    "c2_nn": dict(regs=[[NN], [NN], [NN]], kwargs=dict()),
    "c3_nn_l1C": dict(regs=[[NN], [NN], [L1NN]], kwargs=dict()),
    "c4_pf2_ball": dict(regs=[[], [PF2, {"kind": "l2ball", "norm_bound": 1.0, "non_negativity": False}], []], kwargs=dict()),
    "c5_full": dict(regs=[[NN], [PF2, UNINN, BALL1NN], [L1NN]], kwargs=dict()),
    "l2_const": dict(regs=[[NN], [NN], [NN]], kwargs=dict(l2_penalty=[0.1, 0.2, 0.3], constant_feasibility_penalty=True,
                                                         feasibility_penalty_scale=2.0)),
}


def gen_traj():
    matrices, true_cmf = get_simple_simulated_data(noise_level=0.2, random_state=1)
    I, (J, K), r = len(matrices), matrices[0].shape, 3
    row_ptr = np.arange(I + 1, dtype=np.int64) * J
    X = pack_rows(matrices)
    np.savez_compressed(
        os.path.join(OUT, "c1_data.npz"), X=X, row_ptr=row_ptr,
        A_true=true_cmf[1][0], B_true=pack_rows(true_cmf[1][1]), C_true=true_cmf[1][2],
    )
    rng = np.random.RandomState(7)
    A0, B0, C0 = rng.uniform(size=(I, r)), rng.uniform(size=(I * J, r)), rng.uniform(size=(K, r))
    shapes = {0: (I, r), 1: (I * J, r), 2: (K, r)}
    for name, spec in TRAJ_STACKS.items():
        arrays = dict(A0=A0, B0=B0, C0=C0)
        regs = [[], [], []]
        for mode in range(3):
            for s, d in enumerate(spec["regs"][mode]):
                dual = rng.uniform(size=shapes[mode])
                arrays[f"dual_in_m{mode}_{s}"] = dual
                dual_init = split_rows(dual, row_ptr) if mode == 1 else dual.copy()  # the reference mutates aux/dual inits in place
                if d["kind"] == "parafac2":
                    P = pack_rows([np.eye(J, r) for _ in range(I)])
                    Delta = rng.uniform(size=(r, r))
                    arrays[f"aux_in_m{mode}_{s}_P"] = P
                    arrays[f"aux_in_m{mode}_{s}_Delta"] = Delta
                    aux_init = (split_rows(P, row_ptr), Delta.copy())
                else:
                    aux = rng.uniform(size=shapes[mode])
                    arrays[f"aux_in_m{mode}_{s}"] = aux
                    aux_init = split_rows(aux, row_ptr) if mode == 1 else aux.copy()
                regs[mode].append(make_ref_penalty(d, aux_init=aux_init, dual_init=dual_init))
        cmf, admm_vars, diag = ref_dec.cmf_aoadmm(
            matrices, r, init=(None, (A0.copy(), split_rows(B0, row_ptr), C0.copy())), regs=regs,
            n_iter_max=20, tol=None, absolute_tol=None, return_errors=True, return_admm_vars=True,
            **spec["kwargs"],
        )
        arrays["A"] = cmf[1][0]
        arrays["B"] = pack_rows(cmf[1][1])
        arrays["C"] = cmf[1][2]
        arrays["rec_errors"] = np.array(diag.rec_errors)
        arrays["regularized_loss"] = np.array(diag.regularized_loss)
        for mode in range(3):
            gaps = np.array([[float(g) for g in it_gaps[mode]] for it_gaps in diag.feasibility_gaps])
            arrays[f"gaps_m{mode}"] = gaps.reshape(len(diag.feasibility_gaps), -1)
            for s, d in enumerate(spec["regs"][mode]):
                aux, dual = admm_vars.auxes[mode][s], admm_vars.duals[mode][s]
                if d["kind"] == "parafac2":
                    arrays[f"aux_m{mode}_{s}_P"] = pack_rows(aux[0])
                    arrays[f"aux_m{mode}_{s}_Delta"] = np.asarray(aux[1])
                else:
                    arrays[f"aux_m{mode}_{s}"] = pack_rows(aux) if mode == 1 else np.asarray(aux)
                arrays[f"dual_m{mode}_{s}"] = pack_rows(dual) if mode == 1 else np.asarray(dual)
        arrays["spec"] = np.array(json.dumps(dict(regs=spec["regs"], kwargs=spec["kwargs"], rank=r, n_iter_max=20)))
        np.savez_compressed(os.path.join(OUT, f"traj_{name}.npz"), **arrays)
        print(f"traj_{name}: rec {diag.rec_errors[0]:.6f} -> {diag.rec_errors[-1]:.6f}")

    # Seeded run through the keyword interface: pins the RNG draw order (SURVEY.md Appendix C, Q-RNG)
    cmf, admm_vars, diag = ref_dec.cmf_aoadmm(
        matrices, r, non_negative=True, l1_penalty={2: 0.1}, l2_norm_bound={1: 1.0}, parafac2=True,
        n_iter_max=10, tol=None, absolute_tol=None, return_errors=True, return_admm_vars=True, random_state=0,
    )
    arrays = dict(A=cmf[1][0], B=pack_rows(cmf[1][1]), C=cmf[1][2], rec_errors=np.array(diag.rec_errors),
                  regularized_loss=np.array(diag.regularized_loss))
    arrays["aux_B0_Delta"] = np.asarray(admm_vars.auxes[1][0][1])
    arrays["dual_B1"] = pack_rows(admm_vars.duals[1][1])
    arrays["spec"] = np.array(json.dumps(dict(
        kwargs=dict(non_negative=True, l1_penalty={"2": 0.1}, l2_norm_bound={"1": 1.0}, parafac2=True,
                    n_iter_max=10, random_state=0), rank=r)))
    np.savez_compressed(os.path.join(OUT, "traj_seeded_keywords.npz"), **arrays)
    print(f"traj_seeded_keywords: rec {diag.rec_errors[0]:.6f} -> {diag.rec_errors[-1]:.6f}")

    # config 1 exactly as BASELINE.json states it (converges by tolerance; behavioural pin only)
    cmf, diag = ref_dec.parafac2_aoadmm(matrices, r, non_negative=True, random_state=0, return_errors=True)
    with open(os.path.join(OUT, "c1_known_answer.json"), "w") as f:
        json.dump(dict(n_iter=int(diag.n_iter), message=diag.message, final_rec_error=float(diag.rec_errors[-1]),
                       final_loss=float(diag.regularized_loss[-1])), f, indent=1)
    print("c1 known answer:", diag.n_iter, diag.rec_errors[-1])


# ----------------------------------------------------------------------------------------------
# G3: prox goldens
# ----------------------------------------------------------------------------------------------
def gen_prox():
    rng = np.random.RandomState(3)
    n_mats, J, r = 5, 10, 3
    row_ptr = np.arange(n_mats + 1, dtype=np.int64) * J
    Y = rng.standard_normal((n_mats * J, r))
    rhos = rng.uniform(2, 3, size=n_mats)
    arrays = dict(Y=Y, row_ptr=row_ptr, rhos=rhos)
    manifest = []
    descs = [NN, L1, L1NN, BOX, BALL, BALLNN, UNI, UNINN,
             {"kind": "l1", "reg_strength": 5.0, "non_negativity": False},
             {"kind": "l2ball", "norm_bound": 100.0, "non_negativity": False}]
    for ci, d in enumerate(descs):
        pen = make_ref_penalty(d)
        out_const = pen.factor_matrix_update(Y[:J].copy(), 10.0, None)
        out_list = pen.factor_matrices_update(split_rows(Y, row_ptr), list(rhos), [None] * n_mats)
        arrays[f"p{ci}_single_rho10"] = np.asarray(out_const)
        arrays[f"p{ci}_list"] = pack_rows(out_list)
        if hasattr(pen, "factor_matrix_row_update"):
            arrays[f"p{ci}_row"] = np.asarray(pen.factor_matrix_row_update(Y[0].copy(), 2.5, None))
        arrays[f"p{ci}_penalty"] = np.array(float(pen.penalty(Y[:J])))
        arrays[f"p{ci}_penalty_list"] = np.array(float(pen.penalty(split_rows(Y, row_ptr))))
        manifest.append(d)
    # PARAFAC2 prox: one coordinate-descent sweep from random orthonormal P / random Delta
    P0 = pack_rows([np.linalg.qr(rng.standard_normal((J, r)))[0] for _ in range(n_mats)])
    D0 = rng.standard_normal((r, r))
    pen = ref_pen.Parafac2()
    P1, D1 = pen.factor_matrices_update(split_rows(Y, row_ptr), list(rhos), (split_rows(P0, row_ptr), D0))
    arrays.update(pf2_P0=P0, pf2_D0=D0, pf2_P1=pack_rows(P1), pf2_D1=np.asarray(D1))
    shifted = pen.subtract_from_auxes((P1, D1), split_rows(Y, row_ptr))
    arrays["pf2_aux_minus_Y"] = pack_rows(shifted)
    # unimodal regression vectors (the shapes tests/test_unimodal_regression.py exercises + ties/negatives)
    uni_inputs = [
        rng.standard_normal(25), np.arange(12.0), np.arange(12.0)[::-1].copy(),
        np.concatenate([np.arange(6.0), np.arange(6.0)[::-1]]) + 0.3 * rng.standard_normal(12),
        np.array([1.0, 1.0, 1.0, 1.0]), np.array([0.0, 2.0, 2.0, 0.0, 2.0, 2.0, 0.0]),
        -np.abs(rng.standard_normal(9)), np.array([3.0]), np.array([1.0, -1.0]),
        rng.uniform(size=50) + np.exp(-0.5 * ((np.arange(50) - 30) / 5.0) ** 2),
    ]
    for ui, y in enumerate(uni_inputs):
        arrays[f"uni{ui}_y"] = y
        arrays[f"uni{ui}_out"] = ref_unimodal(y.copy(), non_negativity=False)
        arrays[f"uni{ui}_out_nn"] = ref_unimodal(y.copy(), non_negativity=True)
    arrays["n_uni"] = np.array(len(uni_inputs))
    arrays["manifest"] = np.array(json.dumps(manifest))
    np.savez_compressed(os.path.join(OUT, "prox.npz"), **arrays)
    print("prox: done")


# ----------------------------------------------------------------------------------------------
# G4: stopping matrix
# ----------------------------------------------------------------------------------------------
def gen_stopping():
    rng = np.random.RandomState(11)
    I, J, K, r = 5, 8, 6, 2
    row_ptr = np.arange(I + 1, dtype=np.int64) * J
    A_t, B_t, C_t = rng.uniform(0.1, 1.1, (I, r)), rng.uniform(size=(I * J, r)), rng.uniform(size=(K, r))
    X = np.concatenate([(B_t[row_ptr[i] : row_ptr[i + 1]] * A_t[i]) @ C_t.T for i in range(I)], 0)
    X = X + 0.01 * rng.standard_normal(X.shape)
    matrices = split_rows(X, row_ptr)
    A0, B0, C0 = rng.uniform(size=(I, r)), rng.uniform(size=(I * J, r)), rng.uniform(size=(K, r))
    auxs = {m: rng.uniform(size=s) for m, s in ((0, (I, r)), (1, (I * J, r)), (2, (K, r)))}
    duals = {m: rng.uniform(size=s) for m, s in ((0, (I, r)), (1, (I * J, r)), (2, (K, r)))}
    np.savez_compressed(os.path.join(OUT, "stopping_data.npz"), X=X, row_ptr=row_ptr, A0=A0, B0=B0, C0=C0,
                        **{f"aux{m}": v for m, v in auxs.items()}, **{f"dual{m}": v for m, v in duals.items()})
    inf = float("inf")
    cases = [
        dict(tol=1e-8, absolute_tol=1e-10, feasibility_tol=1e-4, n_iter_max=1000),
        dict(tol=inf, absolute_tol=-inf, feasibility_tol=inf, n_iter_max=50),
        dict(tol=-inf, absolute_tol=inf, feasibility_tol=inf, n_iter_max=50),
        dict(tol=inf, absolute_tol=inf, feasibility_tol=-inf, n_iter_max=7),
        dict(tol=-inf, absolute_tol=-inf, feasibility_tol=inf, n_iter_max=7),
        dict(tol=None, absolute_tol=None, feasibility_tol=1e-4, n_iter_max=9),
        dict(tol=None, absolute_tol=1e-1, feasibility_tol=1e-4, n_iter_max=9),   # Q8: never stops
        dict(tol=1e-3, absolute_tol=None, feasibility_tol=None, n_iter_max=60),
        dict(tol=1e-8, absolute_tol=1e-10, feasibility_tol=1e-4, n_iter_max=0),
        dict(tol=1e-8, absolute_tol=1e-10, feasibility_tol=1e-4, n_iter_max=-3),
        dict(tol=1e-2, absolute_tol=1e-10, feasibility_tol=1e-1, n_iter_max=200, return_errors=False),
    ]
    results = []
    for case in cases:
        kw = dict(case)
        return_errors = kw.pop("return_errors", True)
        regs = [[ref_pen.NonNegativity(aux_init=(split_rows(auxs[m], row_ptr) if m == 1 else auxs[m].copy()),
                                       dual_init=(split_rows(duals[m], row_ptr) if m == 1 else duals[m].copy()))]
                for m in range(3)]
        enc = {k: (None if v is None else ("inf" if v == inf else ("-inf" if v == -inf else v))) for k, v in case.items()}
        try:
            out = ref_dec.cmf_aoadmm(matrices, r, init=(None, (A0.copy(), split_rows(B0, row_ptr), C0.copy())),
                                     regs=regs, return_errors=return_errors, **kw)
        except Exception as e:  # the reference's own error behaviour is part of the contract
            results.append(dict(case=enc, raises=type(e).__name__))
            continue
        enc = {k: (None if v is None else ("inf" if v == inf else ("-inf" if v == -inf else v))) for k, v in case.items()}
        if return_errors:
            cmf, diag = out
            results.append(dict(case=enc, message=diag.message, n_iter=int(diag.n_iter),
                                n_rec=len(diag.rec_errors), n_loss=len(diag.regularized_loss),
                                n_gaps=len(diag.feasibility_gaps),
                                satisfied_stopping_condition=diag.satisfied_stopping_condition,
                                satisfied_feasibility_condition=(None if diag.satisfied_feasibility_condition is None
                                                                 else bool(diag.satisfied_feasibility_condition)),
                                last_rec=float(diag.rec_errors[-1]), last_loss=float(diag.regularized_loss[-1])))
        else:
            cmf = out
            results.append(dict(case=enc, A_sum=float(np.sum(cmf[1][0])), C_sum=float(np.sum(cmf[1][2]))))
    with open(os.path.join(OUT, "stopping.json"), "w") as f:
        json.dump(results, f, indent=1)
    for res in results:
        print("stopping:", res.get("message"), res.get("n_iter"))


# ----------------------------------------------------------------------------------------------
# G5: GeneralizedL2Penalty / UnitSimplex
# ----------------------------------------------------------------------------------------------
def chain_laplacian(n, weight=1.0):
    M = 2 * np.eye(n) - np.eye(n, k=1) - np.eye(n, k=-1)
    M[0, 0] = M[-1, -1] = 1
    return weight * M


def gen_more_penalties():
    rng = np.random.RandomState(11)
    n_mats, J, r = 4, 10, 3
    row_ptr = np.arange(n_mats + 1, dtype=np.int64) * J
    Y = rng.standard_normal((n_mats * J, r))
    rhos = rng.uniform(2, 3, size=n_mats)
    G = rng.standard_normal((J, J))
    arrays = dict(Y=Y, row_ptr=row_ptr, rhos=rhos, M_chain=chain_laplacian(J), M_dense=G @ G.T / J)
    descs = [{"kind": "gl2", "norm_matrix": "M_chain"}, {"kind": "gl2", "norm_matrix": "M_dense"}, {"kind": "simplex"}]
    for ci, d in enumerate(descs):
        dd = dict(d)
        if d["kind"] == "gl2":
            dd["norm_matrix"] = arrays[d["norm_matrix"]]
        pen = make_ref_penalty(dd)
        arrays[f"p{ci}_single_rho10"] = np.asarray(pen.factor_matrix_update(Y[:J].copy(), 10.0, None))
        arrays[f"p{ci}_list"] = pack_rows(pen.factor_matrices_update(split_rows(Y, row_ptr), list(rhos), [None] * n_mats))
        arrays[f"p{ci}_penalty"] = np.array(float(pen.penalty(Y[:J])))
        arrays[f"p{ci}_penalty_list"] = np.array(float(pen.penalty(split_rows(Y, row_ptr))))
    arrays["manifest"] = np.array(json.dumps(descs))

    # trajectory on config-1 data: graph Laplacian (smoothness along the rows of every B_i) + NN on A + simplex on C
    matrices, _ = get_simple_simulated_data(noise_level=0.2, random_state=1)
    I, (Jd, K), rk = len(matrices), matrices[0].shape, 3
    rp = np.arange(I + 1, dtype=np.int64) * Jd
    A0, B0, C0 = rng.uniform(size=(I, rk)), rng.uniform(size=(I * Jd, rk)), rng.uniform(size=(K, rk))
    M_traj = chain_laplacian(Jd, 0.5)
    spec = [[NN], [{"kind": "gl2", "norm_matrix": "M_traj"}], [{"kind": "simplex"}]]
    shapes = {0: (I, rk), 1: (I * Jd, rk), 2: (K, rk)}
    regs = [[], [], []]
    arrays.update(t_A0=A0, t_B0=B0, t_C0=C0, M_traj=M_traj)
    for mode in range(3):
        for sidx, d in enumerate(spec[mode]):
            aux, dual = rng.uniform(size=shapes[mode]), rng.uniform(size=shapes[mode])
            arrays[f"t_aux_in_m{mode}_{sidx}"], arrays[f"t_dual_in_m{mode}_{sidx}"] = aux, dual
            dd = dict(d)
            if d["kind"] == "gl2":
                dd["norm_matrix"] = M_traj
            regs[mode].append(make_ref_penalty(
                dd, aux_init=split_rows(aux, rp) if mode == 1 else aux.copy(),
                dual_init=split_rows(dual, rp) if mode == 1 else dual.copy()))
    cmf, admm_vars, diag = ref_dec.cmf_aoadmm(
        matrices, rk, init=(None, (A0.copy(), split_rows(B0, rp), C0.copy())), regs=regs, n_iter_max=10, tol=None,
        absolute_tol=None, return_errors=True, return_admm_vars=True)
    arrays.update(t_A=cmf[1][0], t_B=pack_rows(cmf[1][1]), t_C=cmf[1][2], t_rec_errors=np.array(diag.rec_errors),
                  t_regularized_loss=np.array(diag.regularized_loss))
    for mode in range(3):
        for sidx in range(len(spec[mode])):
            aux, dual = admm_vars.auxes[mode][sidx], admm_vars.duals[mode][sidx]
            arrays[f"t_aux_m{mode}_{sidx}"] = pack_rows(aux) if mode == 1 else np.asarray(aux)
            arrays[f"t_dual_m{mode}_{sidx}"] = pack_rows(dual) if mode == 1 else np.asarray(dual)
    arrays["t_spec"] = np.array(json.dumps(dict(regs=spec, rank=rk, n_iter_max=10)))

    # early exit of the inner loops (inner_tol, decomposition.py:90-117): 8 outer iterations, up to 25 inner ones
    spec_it = [[NN], [NN, BALL1NN], [L1NN]]
    regs = [[], [], []]
    for mode in range(3):
        for sidx, d in enumerate(spec_it[mode]):
            aux, dual = rng.uniform(size=shapes[mode]), rng.uniform(size=shapes[mode])
            arrays[f"it_aux_in_m{mode}_{sidx}"], arrays[f"it_dual_in_m{mode}_{sidx}"] = aux, dual
            regs[mode].append(make_ref_penalty(
                d, aux_init=split_rows(aux, rp) if mode == 1 else aux.copy(),
                dual_init=split_rows(dual, rp) if mode == 1 else dual.copy()))
    cmf, admm_vars, diag = ref_dec.cmf_aoadmm(
        matrices, rk, init=(None, (A0.copy(), split_rows(B0, rp), C0.copy())), regs=regs, n_iter_max=8, tol=None,
        absolute_tol=None, inner_tol=1e-2, inner_n_iter_max=25, return_errors=True, return_admm_vars=True)
    arrays.update(it_A=cmf[1][0], it_B=pack_rows(cmf[1][1]), it_C=cmf[1][2], it_rec_errors=np.array(diag.rec_errors),
                  it_regularized_loss=np.array(diag.regularized_loss))
    for mode in range(3):
        for sidx in range(len(spec_it[mode])):
            aux, dual = admm_vars.auxes[mode][sidx], admm_vars.duals[mode][sidx]
            arrays[f"it_aux_m{mode}_{sidx}"] = pack_rows(aux) if mode == 1 else np.asarray(aux)
            arrays[f"it_dual_m{mode}_{sidx}"] = pack_rows(dual) if mode == 1 else np.asarray(dual)
    arrays["it_spec"] = np.array(json.dumps(dict(regs=spec_it, rank=rk, n_iter_max=8, inner_tol=1e-2, inner_n_iter_max=25)))
    np.savez_compressed(os.path.join(OUT, "more_penalties.npz"), **arrays)
    print(f"more_penalties: rec {diag.rec_errors[0]:.6f} -> {diag.rec_errors[-1]:.6f}; "
          f"column sums of C {np.sum(cmf[1][2], axis=0)}")


# ----------------------------------------------------------------------------------------------
# G6: the call of the reference's README (README.rst:66-91) as a 10-iteration trajectory
# ----------------------------------------------------------------------------------------------
def gen_readme():
    matrices, _ = get_simple_simulated_data(noise_level=0.2, random_state=1)
    r = 3
    kwargs = dict(non_negative=True, l1_penalty={2: 0.1}, l2_norm_bound=[1, 1, 0], parafac2=True, unimodal={1: True},
                  constant_feasibility_penalty=True, random_state=0)
    cmf, admm_vars, diag = ref_dec.cmf_aoadmm(matrices, r, n_iter_max=10, tol=None, absolute_tol=None, return_errors=True,
                                              return_admm_vars=True, **kwargs)
    arrays = dict(A=cmf[1][0], B=pack_rows(cmf[1][1]), C=cmf[1][2], rec_errors=np.array(diag.rec_errors),
                  regularized_loss=np.array(diag.regularized_loss))
    for mode in range(3):
        gaps = np.array([[float(g) for g in it_gaps[mode]] for it_gaps in diag.feasibility_gaps])
        arrays[f"gaps_m{mode}"] = gaps.reshape(len(diag.feasibility_gaps), -1)
    arrays["aux_A0"] = np.asarray(admm_vars.auxes[0][0])
    arrays["dual_A0"] = np.asarray(admm_vars.duals[0][0])
    arrays["aux_B0_Delta"] = np.asarray(admm_vars.auxes[1][0][1])
    arrays["n_regs"] = np.array([len(admm_vars.auxes[m]) for m in range(3)])
    arrays["spec"] = np.array(json.dumps(dict(
        kwargs=dict(non_negative=True, l1_penalty={"2": 0.1}, l2_norm_bound=[1, 1, 0], parafac2=True, unimodal={"1": True},
                    constant_feasibility_penalty=True, random_state=0, n_iter_max=10), rank=r)))
    np.savez_compressed(os.path.join(OUT, "readme_example.npz"), **arrays)
    print(f"readme_example: rec {diag.rec_errors[0]:.6f} -> {diag.rec_errors[-1]:.6f}; regs per mode {arrays['n_regs']}")


def gen_converters_inits():
    """random_coupled_matrices + dense converters + SVD-based initialisers of the reference, as data."""
    from matcouply import coupled_matrices as ref_cm
    from matcouply.random import random_coupled_matrices as ref_random

    arrays = {}
    shapes = [(7, 6), (4, 6), (9, 6), (5, 6)]
    rank = 3
    arrays["shapes"] = np.array(shapes)
    for tag, kw in (("norm", dict()), ("raw", dict(normalise_factors=False)), ("normB", dict(normalise_factors=False, normalise_B=True))):
        cmf = ref_random(shapes, rank, random_state=3, **kw)
        weights, (A, B_is, C) = cmf
        arrays[f"rc_{tag}_weights"], arrays[f"rc_{tag}_A"], arrays[f"rc_{tag}_C"] = np.asarray(weights), np.asarray(A), np.asarray(C)
        arrays[f"rc_{tag}_B"] = pack_rows([np.asarray(B) for B in B_is])
    cmf = ref_random(shapes, rank, random_state=3)  # the normalised one, with non-trivial weights
    arrays["cv_matrices"] = pack_rows([np.asarray(m) for m in ref_cm.cmf_to_matrices(cmf)])
    arrays["cv_matrix_2"] = np.asarray(ref_cm.cmf_to_matrix(cmf, 2))
    arrays["cv_tensor"] = np.asarray(ref_cm.cmf_to_tensor(cmf))
    for mode in range(3):
        arrays[f"cv_unfolded_{mode}"] = np.asarray(ref_cm.cmf_to_unfolded(cmf, mode))
    arrays["cv_unfolded_2_nopad"] = np.asarray(ref_cm.cmf_to_unfolded(cmf, 2, pad=False))
    arrays["cv_vec"] = np.asarray(ref_cm.cmf_to_vec(cmf))
    arrays["cv_vec_nopad"] = np.asarray(ref_cm.cmf_to_vec(cmf, pad=False))
    full = ref_random(shapes, rank, full=True, random_state=3)
    arrays["rc_full"] = pack_rows([np.asarray(m) for m in full])

    # SVD-based initialisers on config-1 data, and what the solver makes of them in 5 iterations
    X, _ = get_simple_simulated_data(noise_level=0.2, random_state=1)
    X = [np.asarray(x) for x in X]
    for init in ("svd", "threshold_svd"):
        cmf0 = ref_dec.initialize_cmf(X, 3, init, svd_fun=SVD, random_state=None, init_params=None)
        _, (A0, B0, C0) = cmf0
        arrays[f"init_{init}_A"], arrays[f"init_{init}_C"] = np.asarray(A0), np.asarray(C0)
        arrays[f"init_{init}_B"] = pack_rows([np.asarray(B) for B in B0])
        cmf, diag = ref_dec.cmf_aoadmm(X, 3, init=init, non_negative=True, n_iter_max=5, tol=None, absolute_tol=None,
                                       return_errors=True, random_state=0, aux_init="zeros", dual_init="zeros")
        _, (A, B_is, C) = cmf
        arrays[f"run_{init}_A"], arrays[f"run_{init}_C"] = np.asarray(A), np.asarray(C)
        arrays[f"run_{init}_B"] = pack_rows([np.asarray(B) for B in B_is])
        arrays[f"run_{init}_rec_errors"] = np.asarray(diag.rec_errors)
        print(f"init={init}: rec {diag.rec_errors[0]:.6f} -> {diag.rec_errors[-1]:.6f}")
    np.savez_compressed(os.path.join(OUT, "converters_inits.npz"), **arrays)
    print("converters_inits:", len(arrays), "arrays")


if __name__ == "__main__":
    print("reference version", matcouply.__version__)
    if "--only" in sys.argv and sys.argv[sys.argv.index("--only") + 1] == "converters":
        gen_converters_inits()
        sys.exit(0)
    if "--only" in sys.argv and sys.argv[sys.argv.index("--only") + 1] == "more":
        gen_more_penalties()
        sys.exit(0)
    if "--only" in sys.argv and sys.argv[sys.argv.index("--only") + 1] == "readme":
        gen_readme()
        sys.exit(0)
    gen_phase()
    gen_traj()
    gen_prox()
    gen_stopping()
    gen_more_penalties()
    gen_readme()
    gen_converters_inits()
    total = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print(f"total fixture bytes: {total}")
