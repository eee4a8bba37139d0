"""-m gpu: seeded random sweep of shapes / ranks / penalty stacks / options, two outer iterations on the GPU through
the public API against the oracle.  Modes without any penalty get an l2 term (their normal equations would otherwise be singular for
some draws); the 1e-5 bar applies to every case."""
import os

import numpy as np
import pytest

from matcouply_amd import _engine as _engine_mod
from tests.test_gpu_end_to_end import _compare, _run_both

pytestmark = pytest.mark.gpu

ROWSEP = [{"kind": "nn"}, {"kind": "l1", "reg_strength": 0.05}, {"kind": "l1", "reg_strength": 0.02, "non_negativity": True},
          {"kind": "box", "min_val": -0.2, "max_val": 0.8}]
MATRIX = [{"kind": "l2ball", "norm_bound": 0.7}, {"kind": "l2ball", "norm_bound": 1.5, "non_negativity": True},
          {"kind": "unimodal"}, {"kind": "unimodal", "non_negativity": True}]


def _draw_case(rng):
    r = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 12, 16, 17, 24, 32, 40]))
    K = int(rng.choice([r + 3, 33, 64, 100, 128, 200, 256, 300]))
    I = int(rng.randint(2, 9))
    J = rng.randint(max(r, 2), 150, size=I)
    if rng.rand() < 0.3:
        J[rng.randint(I)] = int(rng.choice([64, 65, 128, 256, 257]))
    const = bool(rng.rand() < 0.3)
    regs = [[], [], []]
    # mode 0: row-separable (matrix penalties need a constant rho)
    regs[0] = [dict(ROWSEP[rng.randint(len(ROWSEP))])] if rng.rand() < 0.8 else []
    if const and rng.rand() < 0.5:
        regs[0].append(dict(MATRIX[rng.randint(2)]))
    # mode 1: any mix, optionally PARAFAC2 first (as the reference's parser orders it)
    if rng.rand() < 0.4 and J.min() >= r:
        regs[1].append({"kind": "parafac2"})
    for _ in range(rng.randint(0, 3)):
        pool = ROWSEP + MATRIX
        regs[1].append(dict(pool[rng.randint(len(pool))]))
    regs[1] = regs[1][:3]
    # mode 2
    for _ in range(rng.randint(0, 3)):
        pool = ROWSEP + MATRIX
        regs[2].append(dict(pool[rng.randint(len(pool))]))
    l2 = [0.0 if regs[m] else float(rng.uniform(0.05, 0.5)) for m in range(3)]
    if rng.rand() < 0.3:
        l2 = [v + float(rng.uniform(0, 0.3)) for v in l2]
    return dict(I=I, J=J, K=K, r=r, regs=regs, l2=l2, const=const, scale=float(rng.choice([0.5, 1.0, 2.0])),
                inner=int(rng.choice([1, 3, 5])))


@pytest.mark.parametrize("seed", range(int(os.environ.get("MCL_FUZZ_SEEDS", 24))))  # MCL_FUZZ_SEEDS=300: extended sweep
def test_random_configuration(seed):
    """Flat 1e-5 on every drawn configuration, penalty-free modes included (factors, auxiliary matrices, duals, Delta;
    the PARAFAC2 bases P_i alone carry the polar factor's forward-error bound, see _compare)."""
    from oracle import aoadmm_oracle as orc

    case = _draw_case(np.random.RandomState(1000 + seed))
    X, row_ptr = orc.synthetic_problem(case["I"], case["J"], case["K"], case["r"], seed=seed, dtype=np.float64)
    X = X.astype(np.float32).astype(np.float64)
    st = orc.random_state_for(X, row_ptr, case["r"], case["regs"], seed=seed + 1, l2=case["l2"],
                              inner_n_iter_max=case["inner"], feasibility_penalty_scale=case["scale"],
                              constant_A=case["const"], constant_B=case["const"])
    cmf, admm, diag, res = _run_both(st, 2)
    errs = _compare(cmf, admm, diag, st, res, 1e-5, 1e-5)
    print(seed, {k: case[k] for k in ("I", "K", "r", "const", "inner")}, [[d["kind"] for d in m] for m in case["regs"]],
          f"worst {max(errs.values()):.1e} polar cond {res['polar_cond']:.0e}")


# The draws of the 2000-seed sweep (MCL_FUZZ_SEEDS=2000) that round 4 left outside the flat 1e-5 (1.0e-5 .. 8.5e-5: generic
# B stacks - L2 ball / PARAFAC2 / unimodality - in front of a penalty-free A system of condition >= 1e4, and rank-1 problems),
# pinned so that the driver's suite sees them (VERDICT r4 #1a).
RESIDUE_SEEDS = [84, 134, 193, 233, 240, 260, 381, 418, 484, 567, 695, 984, 1372, 1503, 1575, 1970, 1982]


@pytest.mark.parametrize("seed", RESIDUE_SEEDS)
def test_residue_of_the_extended_sweep(seed):
    test_random_configuration(seed)


# Round 5's 6000-draw sweep left four draws outside the flat bar (1.2e-5 .. 7.2e-5): pinned here so that the
# suite carries them - strict: the day one of them passes, the pin has to be turned into a plain test.
OUTSIDE_SEEDS = [2191, 2970, 3930, 4742]


@pytest.mark.parametrize("seed", OUTSIDE_SEEDS)
@pytest.mark.xfail(strict=True, reason="known residue of the 6000-draw sweep: conditioning of the comparison at fp32 state storage")
def test_known_outside_draws(seed):
    test_random_configuration(seed)


# ---- the MID-SIZE leg: 2^20 < elements of X <= 2^23 -------------------------------------------------------------------------
# The draws above have at most 0.62 M elements: all of them run the exact-products / fp64 path of small problems.  Between 2^20
# and 2^23 elements (BASELINE config 2 is exactly 2^23) the kernels of the BASELINE configurations are the DEFAULT - the MFMA
# contractions, the one-pass sweep, the chained row passes - unless the condition monitor (mcl_condition_monitor, a two-iteration
# trial of `cmf_aoadmm(arithmetic="auto")`) finds an ill-conditioned penalty-free mode or PARAFAC2 polar factor and moves the run
# to the exact arithmetic.  Nothing is forced here: no `arithmetic=` keyword, no MCL_EXACT.
def _draw_mid_case(rng):
    r = int(rng.choice([2, 3, 4, 5, 8, 12, 16, 24, 32]))
    K = int(rng.choice([64, 100, 128, 200, 256, 300, 512, 1024]))
    total = min(2.0 ** rng.uniform(20.05, 23.0), 64 * 800 * K)  # elements of X (at most 64 matrices of at most 1024 rows)
    n_rows = total / K
    I = int(rng.randint(max(4, int(np.ceil(n_rows / 800))), 65))
    J = rng.randint(max(r, 8), 1025, size=I).astype(np.float64)
    J = np.clip(np.round(J * n_rows / J.sum()), max(r, 2), 1024).astype(np.int64)
    for _ in range(4 * I):  # the clips can leave a draw just outside the range: nudge single matrices
        if J.sum() * K <= (1 << 20):
            J[np.argmin(J)] = 1024
        elif J.sum() * K > (1 << 23):
            J[np.argmax(J)] = max(r, 2, J.max() // 2)
        else:
            break
    const = bool(rng.rand() < 0.3)
    regs = [[], [], []]
    regs[0] = [dict(ROWSEP[rng.randint(len(ROWSEP))])] if rng.rand() < 0.7 else []
    if const and rng.rand() < 0.5:
        regs[0].append(dict(MATRIX[rng.randint(2)]))
    if rng.rand() < 0.4:
        regs[1].append({"kind": "parafac2"})
    pool = ROWSEP + MATRIX
    for _ in range(rng.randint(0, 3)):
        regs[1].append(dict(pool[rng.randint(len(pool))]))
    regs[1] = regs[1][:3]
    for _ in range(rng.randint(0, 3)):
        regs[2].append(dict(pool[rng.randint(len(pool))]))
    l2 = [0.0 if regs[m] else float(rng.uniform(0.05, 0.5)) for m in range(3)]
    if rng.rand() < 0.3:
        l2 = [v + float(rng.uniform(0, 0.3)) for v in l2]
    return dict(I=I, J=J, K=K, r=r, regs=regs, l2=l2, const=const, scale=float(rng.choice([0.5, 1.0, 2.0])),
                inner=int(rng.choice([1, 3, 5])))


def _mid_state(seed):
    from oracle import aoadmm_oracle as orc

    case = _draw_mid_case(np.random.RandomState(50000 + seed))
    X, row_ptr = orc.synthetic_problem(case["I"], case["J"], case["K"], case["r"], seed=seed, dtype=np.float64)
    X = X.astype(np.float32).astype(np.float64)
    assert (1 << 20) < X.size <= (1 << 23), X.size
    st = orc.random_state_for(X, row_ptr, case["r"], case["regs"], seed=seed + 1, l2=case["l2"],
                              inner_n_iter_max=case["inner"], feasibility_penalty_scale=case["scale"],
                              constant_A=case["const"], constant_B=case["const"])
    return case, st


# Round 6, 400 draws (MCL_FUZZ_MID_SEEDS=400 MCL_FUZZ_REPORT_ONLY=1): 398 inside the flat bar with the default call - round 5's
# kernels alone (no condition monitor): 25 of the 400 had a quantity above 1e-5; every draw the fast kernels leave outside has a penalty-free mode or an
# ill-conditioned PARAFAC2 polar factor.  The two left: 18 (A and B free, unimodality on C: A at 1.4e-5 in the exact arithmetic
# too - fp32 storage of B and C between the phases in front of A systems of condition ~1e5) and 327 (a PARAFAC2 basis P_i at
# 1.6e-5 where its bound max(1e-5, 1e-8 cond) is 1e-5: conditioning 1e3, just under the polar trigger).  18 lies in the suite's
# range: pinned, strictly.
MID_OUTSIDE = {18: "A at 1.4e-5 in either arithmetic: fp32 state storage in front of penalty-free A systems of condition ~1e5"}


_MID_ARITHMETIC = {}  # seed -> which arithmetic the default call chose (filled by the runs of this module)


def _mid_params(n):
    return [pytest.param(s, marks=pytest.mark.xfail(strict=True, reason=MID_OUTSIDE[s])) if s in MID_OUTSIDE and
            os.environ.get("MCL_FUZZ_REPORT_ONLY") is None else s for s in range(n)]


@pytest.mark.parametrize("seed", _mid_params(int(os.environ.get("MCL_FUZZ_MID_SEEDS", 48))))  # MCL_FUZZ_MID_SEEDS=400: extended sweep
def test_random_mid_size_configuration(seed, monkeypatch):
    """Flat 1e-5 after two outer iterations, default arithmetic, on problems of the size range where the fast kernels are
    the default (reference: decomposition.py:945-1053 through the oracle)."""
    switched = []
    orig = _engine_mod.HipEngine.set_exact
    monkeypatch.setattr(_engine_mod.HipEngine, "set_exact", lambda self, exact=True: (switched.append(bool(exact)), orig(self, exact))[1])
    case, st = _mid_state(seed)
    cmf, admm, diag, res = _run_both(st, 2)
    _MID_ARITHMETIC[seed] = "exact" if switched else "fast"
    label = f"mid {seed} [{_MID_ARITHMETIC[seed]}] " + str({k: case[k] for k in ("I", "K", "r", "const", "inner")}) + f" rows {int(case['J'].sum())} " + \
        str([[d["kind"] for d in m] for m in case["regs"]])
    try:
        errs = _compare(cmf, admm, diag, st, res, 1e-5, 1e-5)
    except AssertionError as exc:
        bad = exc.args[0][0] if exc.args and isinstance(exc.args[0], tuple) else str(exc)
        print(label, "OUTSIDE the bar:", {k: f"{v:.1e}" for k, v in bad.items()} if isinstance(bad, dict) else bad,
              f"polar cond {res['polar_cond']:.0e}")
        if os.environ.get("MCL_FUZZ_REPORT_ONLY") is None:
            raise
        return
    flat = max(v for k, v in errs.items() if k != "gaps" and not (k[0] == "P" and k[1] != "D"))
    print(label, f"inside: worst {flat:.1e} gaps {errs['gaps']:.2f} polar cond {res['polar_cond']:.0e}")


def test_mid_size_leg_runs_both_arithmetics():
    """The leg above must keep covering the kernels of the BASELINE configurations: the condition monitor moves the draws with
    an ill-conditioned penalty-free mode or PARAFAC2 polar factor to the exact arithmetic, the others stay on the fast
    kernels - of the suite's 48 draws at least a third each way (the small-problem fuzz never touched the
    benchmarked kernels)."""
    if len(_MID_ARITHMETIC) < 40:
        pytest.skip("needs the mid-size leg of this module to have run in the same session")
    n_fast = sum(v == "fast" for v in _MID_ARITHMETIC.values())
    print(f"mid-size leg: {n_fast} draws on the fast kernels, {len(_MID_ARITHMETIC) - n_fast} in the exact arithmetic")
    assert n_fast >= len(_MID_ARITHMETIC) // 3 and len(_MID_ARITHMETIC) - n_fast >= len(_MID_ARITHMETIC) // 6


@pytest.mark.parametrize("seed", [45, 135, 142, 237])
def test_ill_conditioned_penalty_free_modes(seed):
    """Draws whose penalty-free modes have normal equations of condition 2e3 .. 5e5 (tools/parity_probe.py fuzz:<seed>): the
    reference solves them with an fp64 SVD (decomposition.py:252-256, 319-321); fp32-accumulated [G | R] / X C (relative error
    1e-8 .. 4e-8 on a few hundred rows) put C or the reconstruction error at 1.2e-5 .. 2.3e-5.  Small problems therefore take
    every contraction as fp64 sums of exact products (csrc/contract.hip: mcl_exact_mode): inside the flat 1e-5, and better
    than the fast kernels on the same draw."""
    from oracle import aoadmm_oracle as orc

    def run(exact):
        old = os.environ.get("MCL_EXACT")
        os.environ.pop("MCL_EXACT", None)
        if not exact:
            os.environ["MCL_EXACT"] = "0"
        try:
            case = _draw_case(np.random.RandomState(1000 + seed))
            X, row_ptr = orc.synthetic_problem(case["I"], case["J"], case["K"], case["r"], seed=seed, dtype=np.float64)
            X = X.astype(np.float32).astype(np.float64)
            st = orc.random_state_for(X, row_ptr, case["r"], case["regs"], seed=seed + 1, l2=case["l2"],
                                      inner_n_iter_max=case["inner"], feasibility_penalty_scale=case["scale"],
                                      constant_A=case["const"], constant_B=case["const"])
            cmf, admm, diag, res = _run_both(st, 2)
            return _compare(cmf, admm, diag, st, res, 1.0, 1.0)  # errors only; the bars are applied below
        finally:
            os.environ.pop("MCL_EXACT", None)
            if old is not None:
                os.environ["MCL_EXACT"] = old

    exact, fast = run(True), run(False)
    worst = lambda e: max(v for k, v in e.items() if k != "gaps" and not (k[0] == "P" and k[1] != "D"))
    print(seed, f"exact products {worst(exact):.1e}   fast kernels {worst(fast):.1e}")
    assert worst(exact) < 1e-5, exact
    assert worst(fast) > worst(exact)


def test_exact_products_mode_is_not_a_performance_cliff():
    """The mode serves the sizes most users of the reference have (its own examples are 15 matrices of 50 x 20): it may cost a
    launch or two per iteration, not a multiple - its first form (one sequential sum per output) was 20x slower than the fast
    kernels at the size limit and nothing noticed.  Guard: at most 3x at the limit (measured 1.6x), 2x at BASELINE config 1's
    size (measured 1.1x) - the fastest of 7 repetitions of 200 iterations each, so that a neighbour on a shared GPU does not
    decide the outcome (a perf guard, not a parity test)."""
    import time

    import torch

    import bench

    dev = torch.device("cuda", 0)
    old = os.environ.get("MCL_EXACT")
    try:
        for (I, J, K, r), bound in (((16, 256, 256, 16), 3.0), ((15, 50, 20, 3), 2.0)):
            t = {}
            for exact in ("1", "0"):
                os.environ["MCL_EXACT"] = exact
                cfg = dict(bench.CONFIGS["c3"], I=I, J=J, K=K, r=r)
                X, row_ptr, I_loc = bench.make_shard(cfg, 0, 1, dev)
                eng = bench.make_engine(cfg, X, row_ptr, I_loc, 0, dev)
                assert (eng.kernel_variant(_engine_mod.VARIANT_EXACT_MODE) != "") == (exact == "1")
                eng.iterate(50)
                torch.cuda.synchronize()
                best = float("inf")
                for _ in range(7):
                    t0 = time.perf_counter()
                    eng.iterate(200)
                    torch.cuda.synchronize()
                    best = min(best, (time.perf_counter() - t0) / 200)
                t[exact] = best
                eng.close()
            print(f"I={I} J={J} K={K} r={r}: exact {1e6 * t['1']:.1f} us, fast {1e6 * t['0']:.1f} us per iteration")
            assert t["1"] < bound * t["0"], (I, J, K, r, t)
    finally:
        os.environ.pop("MCL_EXACT", None)
        if old is not None:
            os.environ["MCL_EXACT"] = old


@pytest.mark.parametrize("seed", [133, 859])
def test_ill_conditioned_polar_factor_takes_the_qr_route(seed):
    """PARAFAC2 at rank 40 from a random dual: in the first inner iteration cond(Y_i Delta^T) is 1e6 .. 2e7 in most slabs.  The
    Gram route of the polar factor squares that (its small eigenvalues are rounding; at 1e7 the pseudo-inverse threshold removes
    them: P off by 0.2, B by 1.6e-3 after the phase); k_pf2_algebra flags such slabs and k_pf2_polar_qr redoes them from
    Y_i Delta^T itself (Householder QR + one-sided Jacobi on R), as the reference's SVD does (penalties.py:1224-1250)."""
    import torch

    from oracle import aoadmm_oracle as orc
    from tests.helpers import engine_from_oracle_state, rel_err, to_np

    case = _draw_case(np.random.RandomState(1000 + seed))
    assert case["r"] == 40 and [d["kind"] for d in case["regs"][1]] == ["parafac2"]
    X, row_ptr = orc.synthetic_problem(case["I"], case["J"], case["K"], case["r"], seed=seed, dtype=np.float64)
    X = X.astype(np.float32).astype(np.float64)
    st = orc.random_state_for(X, row_ptr, case["r"], case["regs"], seed=seed + 1, l2=case["l2"],
                              inner_n_iter_max=case["inner"], feasibility_penalty_scale=case["scale"],
                              constant_A=case["const"], constant_B=case["const"])
    r32 = lambda a: np.asarray(a, np.float32).astype(np.float64)
    st.A, st.B, st.C = r32(st.A), r32(st.B), r32(st.C)
    st.aux[1][0] = (r32(st.aux[1][0][0]), r32(st.aux[1][0][1]))
    st.dual[1][0] = r32(st.dual[1][0])
    eng = engine_from_oracle_state(st)
    conds = []
    orig = orc.polar_factor

    def recording(M):
        sv = np.linalg.svd(M, compute_uv=False)
        conds.append(sv[0] / sv[-1])
        return orig(M)

    orc.polar_factor = recording
    try:
        st.update_B()
    finally:
        orc.polar_factor = orig
    eng.update_B()
    torch.cuda.synchronize()
    assert max(conds[: case["I"]]) > 1e6  # the first inner iteration is the ill-conditioned one
    errs = dict(B=rel_err(to_np(eng.B), st.B), Delta=rel_err(to_np(eng.regs[1][0].aux2), st.aux[1][0][1]),
                PDelta=rel_err(to_np(eng.regs[1][0].aux) @ to_np(eng.regs[1][0].aux2), st.aux[1][0][0] @ st.aux[1][0][1]))
    print(seed, f"cond up to {max(conds):.1e}", {k: f"{v:.1e}" for k, v in errs.items()})
    assert max(errs.values()) < 1e-5, errs
    eng.close()


def test_exact_arithmetic_can_be_forced_on_a_larger_problem():
    """arithmetic="exact" (mcl_options.exact_products = 1): the exact-products contractions and the fp64 inner loops at a size
    where the library would pick the fp32 matrix-core kernels (1.6 M elements of X) - for problems whose penalty-free modes are
    ill-conditioned (here: L2 ball on the B_i, A and C free but for a small ridge).  Closer to the oracle than the fast kernels,
    inside the flat 1e-5."""
    from oracle import aoadmm_oracle as orc

    I, K, r = 40, 128, 8
    J = np.random.RandomState(2).randint(200, 420, I)
    X, row_ptr = orc.synthetic_problem(I, J, K, r, seed=12, dtype=np.float64)
    X = X.astype(np.float32).astype(np.float64)
    assert X.size > (1 << 20)
    regs = [[], [{"kind": "l2ball", "norm_bound": 0.7}], []]
    worst = {}
    for arithmetic in ("exact", "fast", "auto"):
        st = orc.random_state_for(X, row_ptr, r, regs, seed=13, l2=[0.05, 0.0, 0.05])
        cmf, admm, diag, res = _run_both(st, 3, arithmetic=arithmetic)
        errs = _compare(cmf, admm, diag, st, res, 1.0, 1.0)
        worst[arithmetic] = max(v for k, v in errs.items() if k != "gaps")
    print(f"exact {worst['exact']:.1e}   fast {worst['fast']:.1e}   auto (the default) {worst['auto']:.1e}")
    assert worst["exact"] < 1e-5 and worst["exact"] < worst["fast"], worst
    # round 6: the DEFAULT call finds the ill-conditioned penalty-free modes itself (mcl_condition_probe) and is inside the bar
    assert worst["auto"] < 1e-5, worst


# Round 6: the PARAFAC2 Newton-Schulz kernel of rank <= 16 runs FOUR slabs per workgroup (one wave each; a fifth wave takes the
# L2-ball column sums of the four).  Slab counts that leave spare waves in the last workgroup, ranks below and at the tile width,
# slabs shorter than a tile and longer than sixteen tiles (the column sums' second batch), on the FAST kernels.
@pytest.mark.parametrize("I,r", [(1, 3), (2, 16), (3, 5), (5, 16), (6, 9), (7, 16), (9, 12)])
def test_four_slab_newton_schulz_workgroups(I, r):
    from oracle import aoadmm_oracle as orc

    rng = np.random.RandomState(700 + 31 * I + r)
    J = rng.randint(max(r, 20), 400, size=I)
    J[rng.randint(I)] = 1100  # more than sixteen 64-row tiles
    if I > 1:
        J[(int(np.argmax(J)) + 1) % I] = max(r, 17)  # a slab inside one tile
    K = 96
    regs = [[{"kind": "nn"}], [{"kind": "parafac2"}, {"kind": "l2ball", "norm_bound": 1.2}], [{"kind": "nn"}]]
    X, row_ptr = orc.synthetic_problem(I, J, K, r, seed=I + r, dtype=np.float64)
    X = X.astype(np.float32).astype(np.float64)
    st = orc.random_state_for(X, row_ptr, r, regs, seed=I + r + 1, l2=[0.0, 0.0, 0.0], inner_n_iter_max=5,
                              feasibility_penalty_scale=1.0, constant_A=False, constant_B=False)
    old = os.environ.get("MCL_EXACT")
    os.environ["MCL_EXACT"] = "0"  # the kernels of the BASELINE configurations (the problem is small enough for the exact mode)
    try:
        cmf, admm, diag, res = _run_both(st, 2)
    finally:
        os.environ.pop("MCL_EXACT", None)
        if old is not None:
            os.environ["MCL_EXACT"] = old
    # (1e-4: small problems on the fp32 kernels are what the exact-products mode exists for - DESIGN.md section 4; a slab or a
    # column sum taken from the wrong place shows as O(0.1))
    errs = _compare(cmf, admm, diag, st, res, 1e-4, 1e-4)
    print(I, r, [int(j) for j in J], "worst factor / variable error %.1e" % max(v for k, v in errs.items() if k != "gaps"),
          f"polar cond {res['polar_cond']:.0e}")
