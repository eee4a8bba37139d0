"""CPU, world sizes 2 / 4 / 8 over gloo: the product's sharded driver (`cmf_aoadmm(..., group=...)`: slabs split over ranks,
all-reduce of the C-mode normal equations / diagnostics / PARAFAC2 coordinate sums / constant-rho maxima) must
reproduce the single-process run.  The checker engine stands in for the HIP engine (no GPU here)."""
import json
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# keyword overrides of the runs below (set per test; the default is a fixed iteration count)
RUN_KW = dict(n_iter_max=6, tol=None, absolute_tol=None)
STOP_RULES = {
    # stops on the relative criterion somewhere inside the budget (every iterate counts as feasible)
    "relative": dict(n_iter_max=40, tol=2e-2, absolute_tol=1e-12, feasibility_tol=float("inf")),
    # never feasible: runs to the budget; the loss is evaluated all the same (return_errors)
    "budget": dict(n_iter_max=11, tol=1e-3, absolute_tol=1e-12, feasibility_tol=1e-30),
}

CASES = {
    "c3_nn_l1C": dict(non_negative=True, l1_penalty={2: 0.1}),
    "pf2_ball_constant": dict(parafac2=True, l2_norm_bound={1: 1.0}, non_negative={0: True},
                              constant_feasibility_penalty=True),
    # the stack of the reference's README (README.rst:66-91): L2 ball on A (column norms over rows that live on different
    # ranks), PARAFAC2 + unimodality + L2 ball on the B_i, L1 on C, constant feasibility penalty
    "readme_stack": dict(non_negative=True, l1_penalty={2: 0.1}, l2_norm_bound=[1, 1, 0], parafac2=True,
                         unimodal={1: True}, constant_feasibility_penalty=True),
    # total variation on the B_i and on C: the penalty value is summed on the host - under sharding this rank's share travels in
    # the diagnostics vector (mode 1) / is counted once (replicated mode 2)
    "tv_B_and_C": dict(),
    # inner_tol set: every penalty is evaluated on the host (PEN_EXTERNAL), the inner loops stop on the reference's test -
    # over ALL B_i, i.e. on all-reduced norms under sharding - and the L1 value on the B_i must not be scaled twice
    "inner_tol_l1B": dict(),
    # matrix penalties on A other than the L2 ball: unimodality down the rows of A and total variation along them couple rows
    # that live on different ranks - every rank evaluates the prox on the all-gathered A + U and keeps its rows
    "matrix_penalties_on_A": dict(constant_feasibility_penalty=True),
    # ... and the same with inner_tol set: the two penalties are then host-evaluated (PEN_EXTERNAL) and keep their auxiliary
    # variable in an object of their own, which must follow the gathered prox (it stayed at its initial value)
    "inner_tol_matrix_A": dict(constant_feasibility_penalty=True),
    # the kinds that became native in round 5 on the replicated mode: a GeneralizedL2 penalty (its value comes from the engine,
    # mcl_penalty_value, and is counted once) and the unit simplex on C
    "gl2_simplex_on_C": dict(),
    # a user's MatricesPenalty on the B_i whose prox and value couple ALL matrices (as PARAFAC2 does natively): under sharding
    # every rank evaluates it on the all-gathered matrices and keeps its own; the value is counted once
    "coupled_matrices_on_B": dict(),
}


def _coupled_penalty(aux, dual):
    """every B_i scaled by ONE factor: the stacked matrix projected on a Frobenius ball whose radius follows the feasibility
    penalties of all matrices; the value is the (non-additive) Frobenius norm of the stack.  Module level: spawned ranks import it"""
    from matcouply_amd import penalties as pen

    class SharedScale(pen.MatricesPenalty):
        def factor_matrices_update(self, factor_matrices, feasibility_penalties, auxes):
            total = float(sum((m ** 2).sum() for m in factor_matrices)) ** 0.5
            radius = 2.0 + 0.1 * float(np.mean(feasibility_penalties)) + 0.05 * len(factor_matrices)
            scale = min(1.0, radius / total)
            return [0.9 * scale * m + 0.1 * a for m, a in zip(factor_matrices, auxes)]

        def penalty(self, x):
            return 0.01 * float(sum((m ** 2).sum() for m in x)) ** 0.5

    return SharedScale(aux_init=aux, dual_init=dual)


J_ALL = [12, 11, 9, 10, 5, 6, 3, 4, 5, 3, 4, 6, 3, 4]  # ragged: the ranks' shares are uneven in slabs AND in rows


def _bounds(world):
    """contiguous slab ranges balanced by rows (the product's own partition_slabs)"""
    from matcouply_amd.decomposition import partition_slabs

    parts = partition_slabs(J_ALL, world)
    assert all(len(p) for p in parts)
    return [int(parts[0][0])] + [int(p[-1]) + 1 for p in parts]


def _problem():
    rng = np.random.RandomState(5)
    I, K, r = len(J_ALL), 9, 3
    J = J_ALL
    A, C = rng.uniform(0.1, 1.1, (I, r)), rng.uniform(size=(K, r))
    mats = [(rng.uniform(size=(j, r)) * A[i]) @ C.T + 0.05 * rng.standard_normal((j, K)) for i, j in enumerate(J)]
    init = (None, (rng.uniform(size=(I, r)), [rng.uniform(size=(j, r)) for j in J], rng.uniform(size=(K, r))))
    return mats, init, r


def _explicit_state(case, mats, r, seed=9):
    """explicit aux/dual so that the sharded and the single run start from identical states"""
    from matcouply_amd import penalties as pen

    rng = np.random.RandomState(seed)
    I, K = len(mats), mats[0].shape[1]
    kw = CASES[case]
    regs = [[], [], []]
    mk = lambda shp: rng.uniform(size=shp)
    if case == "c3_nn_l1C":
        regs[0] = [("nn", mk((I, r)), mk((I, r)))]
        regs[1] = [("nn", [mk((m.shape[0], r)) for m in mats], [mk((m.shape[0], r)) for m in mats])]
        regs[2] = [("l1nn", mk((K, r)), mk((K, r)))]
    elif case == "inner_tol_l1B":
        regs[0] = [("nn", mk((I, r)), mk((I, r)))]
        regs[1] = [("l1B", [mk((m.shape[0], r)) for m in mats], [mk((m.shape[0], r)) for m in mats])]
        regs[2] = [("l1nn", mk((K, r)), mk((K, r)))]
    elif case in ("matrix_penalties_on_A", "inner_tol_matrix_A"):
        regs[0] = [("uninn", mk((I, r)), mk((I, r))), ("tvA", mk((I, r)), mk((I, r)))]
        regs[1] = [("nn", [mk((m.shape[0], r)) for m in mats], [mk((m.shape[0], r)) for m in mats])]
        regs[2] = [("l1nn", mk((K, r)), mk((K, r)))]
    elif case == "gl2_simplex_on_C":
        regs[0] = [("nn", mk((I, r)), mk((I, r)))]
        regs[1] = [("nn", [mk((m.shape[0], r)) for m in mats], [mk((m.shape[0], r)) for m in mats])]
        regs[2] = [("gl2C", mk((K, r)), mk((K, r))), ("simplex", mk((K, r)), mk((K, r)))]
    elif case == "coupled_matrices_on_B":
        regs[0] = [("nn", mk((I, r)), mk((I, r)))]
        regs[1] = [("coupled", [mk((m.shape[0], r)) for m in mats], [mk((m.shape[0], r)) for m in mats]),
                   ("nn", [mk((m.shape[0], r)) for m in mats], [mk((m.shape[0], r)) for m in mats])]
        regs[2] = [("l1nn", mk((K, r)), mk((K, r)))]
    elif case == "tv_B_and_C":
        regs[0] = [("nn", mk((I, r)), mk((I, r)))]
        regs[1] = [("tv", [mk((m.shape[0], r)) for m in mats], [mk((m.shape[0], r)) for m in mats])]
        regs[2] = [("tvl1", mk((K, r)), mk((K, r)))]
    elif case == "readme_stack":
        regs[0] = [("ballnn", mk((I, r)), mk((I, r)))]
        regs[1] = [("pf2", ([np.eye(m.shape[0], r) for m in mats], mk((r, r))), [mk((m.shape[0], r)) for m in mats]),
                   ("uninn", [mk((m.shape[0], r)) for m in mats], [mk((m.shape[0], r)) for m in mats]),
                   ("ballnn", [mk((m.shape[0], r)) for m in mats], [mk((m.shape[0], r)) for m in mats])]
        regs[2] = [("l1nn", mk((K, r)), mk((K, r)))]
    else:
        regs[0] = [("nn", mk((I, r)), mk((I, r)))]
        regs[1] = [("pf2", ([np.eye(m.shape[0], r) for m in mats], mk((r, r))), [mk((m.shape[0], r)) for m in mats]),
                   ("ball", [mk((m.shape[0], r)) for m in mats], [mk((m.shape[0], r)) for m in mats])]
    return regs, kw


def _build(regs_spec, lo, hi):
    from matcouply_amd import penalties as pen

    out = [[], [], []]
    for m in range(3):
        for kind, aux, dual in regs_spec[m]:
            if m == 0:
                aux, dual = aux[lo:hi].copy(), dual[lo:hi].copy()
            elif m == 1:
                dual = [d.copy() for d in dual[lo:hi]]
                aux = ([p.copy() for p in aux[0][lo:hi]], aux[1].copy()) if kind == "pf2" else [a.copy() for a in aux[lo:hi]]
            else:
                aux, dual = aux.copy(), dual.copy()
            if kind == "nn":
                out[m].append(pen.NonNegativity(aux_init=aux, dual_init=dual))
            elif kind == "l1nn":
                out[m].append(pen.L1Penalty(0.1, non_negativity=True, aux_init=aux, dual_init=dual))
            elif kind == "l1B":
                out[m].append(pen.L1Penalty(0.3, aux_init=aux, dual_init=dual))
            elif kind == "pf2":
                out[m].append(pen.Parafac2(aux_init=aux, dual_init=dual))
            elif kind == "ball":
                out[m].append(pen.L2Ball(1.0, aux_init=aux, dual_init=dual))
            elif kind == "ballnn":
                out[m].append(pen.L2Ball(1.0, non_negativity=True, aux_init=aux, dual_init=dual))
            elif kind == "uninn":
                out[m].append(pen.Unimodality(non_negativity=True, aux_init=aux, dual_init=dual))
            elif kind == "tv":
                out[m].append(pen.TotalVariationPenalty(0.05, aux_init=aux, dual_init=dual))
            elif kind == "tvA":
                out[m].append(pen.TotalVariationPenalty(0.04, aux_init=aux, dual_init=dual))
            elif kind == "gl2C":
                n = aux.shape[0]
                out[m].append(pen.GeneralizedL2Penalty(0.3 * (2 * np.eye(n) - np.eye(n, k=1) - np.eye(n, k=-1)) + 0.05 * np.eye(n),
                                                       aux_init=aux, dual_init=dual))
            elif kind == "simplex":
                out[m].append(pen.UnitSimplex(aux_init=aux, dual_init=dual))
            elif kind == "coupled":
                out[m].append(_coupled_penalty(aux, dual))
            elif kind == "tvl1":
                out[m].append(pen.TotalVariationPenalty(0.03, l1_strength=0.02, aux_init=aux, dual_init=dual))
    return out


def _run(case, lo, hi, group):
    from matcouply_amd import decomposition as dec
    from tests.oracle_engine import OracleEngineFactory

    previous, dec._ENGINE_FACTORY = dec._ENGINE_FACTORY, OracleEngineFactory()
    try:
        return _run_with_checker(dec, case, lo, hi, group)
    finally:
        dec._ENGINE_FACTORY = previous


def _run_with_checker(dec, case, lo, hi, group):
    mats, init, r = _problem()
    regs_spec, kw = _explicit_state(case, mats, r)
    const = kw.get("constant_feasibility_penalty", False)
    w, (A0, B0, C0) = init
    cmf, admm, diag = dec.cmf_aoadmm(mats[lo:hi], r, init=(None, (A0[lo:hi].copy(), [b.copy() for b in B0[lo:hi]], C0.copy())),
                               regs=_build(regs_spec, lo, hi), return_errors=True, constant_feasibility_penalty=const, group=group,
                               **(json.loads(os.environ["MCL_TEST_RUN_KW"]) if os.environ.get("MCL_TEST_RUN_KW") else RUN_KW),
                               **(dict(inner_tol=3e-2, inner_n_iter_max=12) if case.startswith("inner_tol") else {}),
                               gather_A=group is not None, return_admm_vars=True)
    return cmf, diag, admm


def _worker(rank, world, port, case, q):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    bounds = _bounds(world)
    cmf, diag, admm = _run(case, bounds[rank], bounds[rank + 1], dist.group.WORLD)
    q.put((rank, cmf[1][0], np.concatenate(cmf[1][1]), cmf[1][2], diag.rec_errors, diag.regularized_loss,
           [[list(map(float, g)) for g in it] for it in diag.feasibility_gaps], np.asarray(cmf.A_all), cmf.rows_of_rank,
           (diag.n_iter, diag.message, diag.satisfied_stopping_condition, bool(diag.satisfied_feasibility_condition)),
           [np.asarray(a) for a in admm.auxes[0]], [np.asarray(u) for u in admm.duals[0]]))
    dist.destroy_process_group()


def _spawn(world, case, salt=0):
    import socket

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as sock:  # a free rendezvous port on the loopback interface
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = [ctx.Process(target=_worker, args=(rk, world, port, case, q)) for rk in range(world)]
    for p in procs:
        p.start()
    try:
        results = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    assert all(p.exitcode == 0 for p in procs)
    return results


# world size 2: every stack; 4 and 8 ranks (the driver's scaling points): the stacks with the most collectives per iteration
SHARDED_RUNS = [(2, c) for c in sorted(CASES)] + [(w, c) for w in (4, 8) for c in ("c3_nn_l1C", "readme_stack", "inner_tol_l1B")] \
    + [(4, "matrix_penalties_on_A"), (4, "coupled_matrices_on_B")]


@pytest.mark.parametrize("world,case", SHARDED_RUNS, ids=[f"{c}-x{w}" for w, c in SHARDED_RUNS])
def test_two_rank_sharded_run_equals_single_process(world, case):
    sys.path.insert(0, REPO)
    ref_cmf, ref_diag, ref_admm = _run(case, 0, len(J_ALL), None)
    results = _spawn(world, case)
    bounds = _bounds(world)
    # the ADMM variables of mode 0 (rank-local rows; return_admm_vars): the ranks' rows together are the single-process ones
    for k in range(len(ref_admm.auxes[0])):
        np.testing.assert_allclose(np.concatenate([res[10][k] for res in results]), np.asarray(ref_admm.auxes[0][k]), rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(np.concatenate([res[11][k] for res in results]), np.asarray(ref_admm.duals[0][k]), rtol=1e-9, atol=1e-12)
    assert len({hi - lo for lo, hi in zip(bounds, bounds[1:])}) > 1 or world == len(J_ALL)  # uneven shares
    A = np.concatenate([res[1] for res in results])
    # gather_A=True: every rank also holds the WHOLE A (one all-gather at the end), its own rows at rows_of_rank
    for rk, res in enumerate(results):
        np.testing.assert_array_equal(res[7], A)
        assert res[8] == (bounds[rk], bounds[rk + 1])
        np.testing.assert_array_equal(res[7][res[8][0]:res[8][1]], res[1])
    B = np.concatenate([res[2] for res in results])
    np.testing.assert_allclose(A, ref_cmf[1][0], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(B, np.concatenate(ref_cmf[1][1]), rtol=1e-9, atol=1e-12)
    for res in results:  # replicated quantities identical on every rank
        np.testing.assert_allclose(res[3], ref_cmf[1][2], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(res[4], ref_diag.rec_errors, rtol=1e-9)
        np.testing.assert_allclose(res[5], ref_diag.regularized_loss, rtol=1e-9)
        ref_gaps = [[list(map(float, g)) for g in it] for it in ref_diag.feasibility_gaps]
        for got_it, ref_it in zip(res[6], ref_gaps):
            for g, rg in zip(got_it, ref_it):
                np.testing.assert_allclose(g, rg, rtol=1e-8, atol=1e-12)
    for res in results[1:]:  # the replicated factor C: bit-identical on every rank
        np.testing.assert_array_equal(results[0][3], res[3])
    if case.startswith("inner_tol"):  # the inner loops did stop early somewhere (otherwise the case pins nothing)
        assert len(ref_diag.rec_errors) == RUN_KW["n_iter_max"] + 1


@pytest.mark.parametrize("world", [2, 8])
@pytest.mark.parametrize("rule", sorted(STOP_RULES))
def test_two_rank_sharded_run_with_a_stopping_rule(rule, world, monkeypatch):
    """the stopping rule under sharding (mcl_gate_begin / mcl_verdict, here restated by the checker engine): the phases are
    stepped with their reductions, the diagnostics vector is all-reduced, every rank evaluates the rule on the same bits and
    the loop runs in fixed chunks - same stopping iteration, message, lists and factors as the single-process call"""
    sys.path.insert(0, REPO)
    kw = STOP_RULES[rule]
    monkeypatch.setenv("MCL_TEST_RUN_KW", json.dumps(kw))  # (the workers are spawned processes: passed through the environment)
    case = "pf2_ball_constant"
    ref_cmf, ref_diag, _ = _run(case, 0, len(J_ALL), None)
    results = _spawn(world, case)
    if rule == "relative":
        assert ref_diag.message.startswith("FEASIBILITY GAP CRITERION AND RELATIVE") and 1 <= ref_diag.n_iter < kw["n_iter_max"]
    else:
        assert ref_diag.n_iter == kw["n_iter_max"] and not ref_diag.satisfied_stopping_condition
    want = (ref_diag.n_iter, ref_diag.message, ref_diag.satisfied_stopping_condition, bool(ref_diag.satisfied_feasibility_condition))
    for res in results:
        assert res[9] == want
        np.testing.assert_allclose(res[4], ref_diag.rec_errors, rtol=1e-9)
        np.testing.assert_allclose(res[5], ref_diag.regularized_loss, rtol=1e-9)
        np.testing.assert_allclose(res[3], ref_cmf[1][2], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(np.concatenate([res[1] for res in results]), ref_cmf[1][0], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(np.concatenate([res[2] for res in results]), np.concatenate(ref_cmf[1][1]), rtol=1e-9, atol=1e-12)


def test_partition_slabs_balances_rows():
    """slabs are split over ranks by their ROWS (sum of J_i), contiguous or greedy (LPT) for ragged data."""
    import numpy as np
    from matcouply_amd.decomposition import partition_slabs

    J = np.random.RandomState(0).randint(128, 1025, 1024)  # BASELINE config 4
    for world in (2, 4, 8):
        for contiguous in (True, False):
            parts = partition_slabs(J, world, contiguous=contiguous)
            assert sorted(np.concatenate(parts).tolist()) == list(range(1024))  # a partition
            rows = np.array([J[p].sum() for p in parts], dtype=np.float64)
            assert rows.max() / rows.mean() <= (1.02 if contiguous else 1.001), (world, contiguous, rows)
            if contiguous:
                assert all(np.all(np.diff(p) == 1) for p in parts)
    # by count it would be off by the spread of J_i; equal J_i degenerate to an even split
    assert [len(p) for p in partition_slabs([7] * 12, 4)] == [3, 3, 3, 3]
    assert [p.tolist() for p in partition_slabs([10, 1, 1, 1, 1, 10], 2)] == [[0, 1, 2], [3, 4, 5]]
    assert [p.tolist() for p in partition_slabs([9, 1, 1, 1, 1, 1, 1, 1, 1, 1], 2, contiguous=False)] == [[0], list(range(1, 10))]
    assert sum(len(p) for p in partition_slabs([5, 5, 5], 4)) == 3  # more ranks than matrices: some ranks stay empty
