"""-m gpu: the one-pass sweep (csrc/sweep.hip: B-phase fused with X_i^T B_i, X read once per outer iteration) against
the oracle and against the two-pass path, including the orders of phase calls in which its by-products must NOT be used.

Tolerance 1e-5 relative (BASELINE.json north_star) as everywhere else."""
import os

import numpy as np
import pytest

from matcouply_amd import _engine as _engine_mod
from tests.helpers import engine_from_oracle_state, rel_err, to_np

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("fast_kernels")]  # the subject is the sweep kernel

NN = {"kind": "nn"}
CASES = {
    # name: (J_i, K, rank, penalties per mode, constant feasibility penalty on (A, B))
    "k256_r16_nn": ([300, 64, 1100, 257, 80, 513], 256, 16, [[NN], [NN], [NN]], (False, False)),
    "k512_r16_nn": ([200, 90, 700, 333, 128], 512, 16, [[NN], [NN], [NN]], (False, False)),
    "k512_r12_l1_box": ([150, 400, 260, 96], 512, 12,
                        [[NN], [{"kind": "l1", "reg_strength": 0.02}, {"kind": "box", "min_val": -0.5, "max_val": 2.0}],
                         [{"kind": "l1", "reg_strength": 0.01, "non_negativity": True}]], (False, False)),
    "k256_r32_nn": ([150, 400, 260, 96], 256, 32, [[NN], [NN], [NN]], (False, False)),
    "k256_r20_l1nn": ([150, 400, 260, 96], 256, 20,
                      [[NN], [{"kind": "l1", "reg_strength": 0.05, "non_negativity": True}], [NN]], (False, False)),
    "k256_r8_constant": ([2100, 1024, 64], 256, 8, [[NN], [NN], [NN]], (True, True)),
    "k256_r16_constant_B": ([260, 100, 99, 515], 256, 16, [[NN], [NN], [NN]], (False, True)),
    # K that is not a multiple of 256: tile rows padded through zero C fragments
    "k128_r8_nn": ([256, 256, 100, 77], 128, 8, [[NN], [NN], [NN]], (False, False)),
    "k200_r16_box": ([150, 400, 260, 96], 200, 16, [[NN], [{"kind": "box", "min_val": 0.0, "max_val": 1.5}], [NN]], (False, False)),
    "k300_r12_nn": ([150, 400, 260, 96], 300, 12, [[NN], [NN], [NN]], (False, False)),
    "k36_r4_none": ([90, 200, 64], 36, 4, [[NN], [], [NN]], (False, False)),
    # K <= 128: the half-width kernels (two rows per wave load), with padded columns and a short second row pair
    "k100_r8_nn": ([150, 400, 260, 97], 100, 8, [[NN], [NN], [NN]], (False, False)),
    # more than 512 slabs with one partial each: the A-phase finish pairs two slabs per workgroup; the odd slab out
    "pairs_515_slabs": ([64] * 515, 64, 8, [[NN], [NN], [NN]], (False, False)),
    "k36_r4_nn": ([90, 200, 64, 33], 36, 4, [[NN], [{"kind": "l1", "reg_strength": 0.02, "non_negativity": True}], [NN]], (False, False)),
    "k508_r16_nn": ([130, 222, 97], 508, 16, [[NN], [NN], [NN]], (False, False)),
    # ranks that are not a multiple of 4: scalar row accesses of B / aux / dual
    "k256_r5_nn": ([150, 400, 260, 96], 256, 5, [[NN], [NN], [NN]], (False, False)),
    "k128_r3_l1": ([256, 100, 77, 300], 128, 3, [[NN], [{"kind": "l1", "reg_strength": 0.03}], [NN]], (False, False)),
    "k512_r7_box": ([130, 222, 97], 512, 7, [[NN], [{"kind": "box", "min_val": 0.0, "max_val": 1.2}, NN], [NN]], (False, False)),
    "k256_r21_nn": ([150, 400, 260], 256, 21, [[NN], [NN], [NN]], (False, False)),
    # bsegs shorter than one 16-row block, ragged tails, a slab of exactly one block
    "k256_ragged_tails": ([70, 130, 33, 257, 64, 16, 401, 15, 17, 300], 256, 16, [[NN], [NN], [NN]], (False, False)),
}


def _state(name, seed=2):
    from oracle import aoadmm_oracle as orc

    J, K, r, regs, (cA, cB) = CASES[name]
    J = np.array(J)
    X, row_ptr = orc.synthetic_problem(len(J), J, K, r, seed=1, dtype=np.float64)
    X = X.astype(np.float32).astype(np.float64)
    st = orc.random_state_for(X, row_ptr, r, regs, seed=seed)
    st.constant_A, st.constant_B = cA, cB
    return st


def _dual_err(got, want, factor):
    return np.linalg.norm(got - want) / max(np.linalg.norm(want), np.linalg.norm(factor))


@pytest.mark.parametrize("name", sorted(CASES))
def test_one_iteration_phase_by_phase(name):
    """B -> C -> A through the C ABI with the sweep active, every by-product against the oracle."""
    import copy
    import torch

    st = _state(name)
    ref = copy.deepcopy(st)
    eng = engine_from_oracle_state(st)
    r = st.A.shape[1]
    for it in range(2):
        eng.update_B()
        if st.regs[1]:  # a penalty-free B is a plain least-squares update: un-shifted systems, solved in fp64 (no sweep)
            assert eng.kernel_variant(_engine_mod.PROF_SWEEP).startswith("k_sweep<"), "the sweep did not run: " + repr(eng.kernel_variant(_engine_mod.PROF_SWEEP))
        ref.update_B()
        torch.cuda.synchronize()
        errs = {"B": rel_err(to_np(eng.B), ref.B)}
        for k in range(len(st.regs[1])):
            errs[f"auxB{k}"] = rel_err(to_np(eng.regs[1][k].aux), ref.aux[1][k])
            errs[f"dualB{k}"] = _dual_err(to_np(eng.regs[1][k].dual), ref.dual[1][k], ref.B)
        gr = to_np(eng.update_C_local())
        Ba = np.concatenate([ref.B[ref.row_ptr[i]:ref.row_ptr[i + 1]] * ref.A[i] for i in range(ref.A.shape[0])])
        errs["G"] = rel_err(gr[:r * r].reshape(r, r), Ba.T @ Ba)
        errs["R"] = rel_err(gr[r * r:].reshape(-1, r), ref.X.T @ Ba)
        eng.update_C_finish()
        ref.update_C()
        errs["C"] = rel_err(to_np(eng.C), ref.C)
        eng.update_A()
        ref.update_A()
        torch.cuda.synchronize()
        errs["A"] = rel_err(to_np(eng.A), ref.A)
        d = eng.diagnostics().cpu().numpy()
        from matcouply_amd import _engine as E

        rec = np.sqrt(max(0.0, d[E.DIAG_X_SQ] - 2 * d[E.DIAG_INNER] + d[E.DIAG_MODEL_SQ]) / d[E.DIAG_X_SQ])
        errs["rec"] = abs(rec - ref.rec_error_from_A_byproducts()) / ref.rec_error_from_A_byproducts()
        errs["normB"] = abs(d[E.DIAG_NORM_SQ + 1] - np.sum(ref.B ** 2)) / np.sum(ref.B ** 2)
        tol = 1e-5  # flat, penalty-free modes included (their normal equations are built and solved in fp64)
        bad = {k: v for k, v in errs.items() if not (v < tol)}
        assert not bad, (name, it, bad)
    eng.close()


@pytest.mark.parametrize("name", ["k256_r16_nn", "k512_r12_l1_box", "k256_ragged_tails", "k256_r8_constant", "k128_r8_nn",
                                  "k300_r12_nn", "k256_r5_nn", "k512_r7_box"])
def test_trajectory_vs_oracle(name):
    from tests.test_gpu_end_to_end import _compare, _run_both

    st = _state(name)
    cmf, admm, diag, res = _run_both(st, 8)
    _compare(cmf, admm, diag, st, res, 1e-5, tol_rec=1e-5)


@pytest.mark.parametrize("name", ["k256_r16_nn", "k512_r16_nn", "k256_r32_nn"])
def test_sweep_equals_two_pass(name, monkeypatch):
    """same problem with and without the sweep (MCL_NO_SWEEP=1 selects the two-pass kernels): fp32-rounding-level equality"""
    from tests.test_gpu_end_to_end import _run_both

    out = {}
    for mode in ("sweep", "two_pass"):
        if mode == "two_pass":
            monkeypatch.setenv("MCL_NO_SWEEP", "1")
        else:
            monkeypatch.delenv("MCL_NO_SWEEP", raising=False)
        out[mode] = _run_both(_state(name), 6)
    monkeypatch.delenv("MCL_NO_SWEEP", raising=False)
    (cs, _, ds, _), (ct, _, dt, _) = out["sweep"], out["two_pass"]
    assert rel_err(cs[1][0], ct[1][0]) < 1e-5 and rel_err(cs[1][2], ct[1][2]) < 1e-5
    assert rel_err(np.concatenate(cs[1][1]), np.concatenate(ct[1][1])) < 1e-5
    assert max(abs(a - b) / b for a, b in zip(ds.rec_errors, dt.rec_errors)) < 1e-5


@pytest.mark.parametrize("name,waves", [("k256_ragged_tails", 3), ("k256_r16_nn", 5), ("k512_r12_l1_box", 2)])
def test_balanced_wave_partition_with_cuts(name, waves, monkeypatch):
    """With fewer waves than work units the planner cuts bsegs / segments where a wave's quota of 16-row blocks ends
    (ragged slabs: csrc/api.hip, mcl_set_problem).  MCL_SWEEP_WAVES / MCL_XC_WAVES force that regime on a small problem:
    the sweep and the two-pass kernels must still agree with the oracle to 1e-5."""
    from tests.test_gpu_end_to_end import _compare, _run_both

    monkeypatch.setenv("MCL_SWEEP_WAVES", str(waves))
    monkeypatch.setenv("MCL_XC_WAVES", str(waves))
    st = _state(name)
    cmf, admm, diag, res = _run_both(st, 4)
    _compare(cmf, admm, diag, st, res, 1e-5, tol_rec=1e-5)
    monkeypatch.setenv("MCL_NO_SWEEP", "1")  # the X C / X^T passes over the cut segments
    st = _state(name)
    cmf, admm, diag, res = _run_both(st, 4)
    _compare(cmf, admm, diag, st, res, 1e-5, tol_rec=1e-5)


def test_wave_partition_is_balanced_on_ragged_slabs():
    """The X passes and the sweep run one wave per SIMD side by side: the longest wave is the kernel's duration.  On the
    ragged slabs of BASELINE config 4 (J_i in [128, 1024]) every wave must hold the same number of 16-row blocks (segments:
    exactly the quota; bsegs: at most 2 blocks above it), the units must tile every slab and no unit may exceed its cap."""
    import torch
    from matcouply_amd._engine import HipEngine, NativeReg, KIND

    J = np.random.RandomState(0).randint(128, 1025, 1024)
    row_ptr = np.concatenate([[0], np.cumsum(J)]).astype(np.int64)
    N, K, r = int(row_ptr[-1]), 256, 16
    dev = torch.device("cuda:0")
    X = torch.zeros((N, K), dtype=torch.float32, device=dev)
    f = lambda *shape: torch.rand(shape, dtype=torch.float32, device=dev)
    nn = lambda rows: NativeReg(KIND["nn"], f(rows, r), torch.zeros((rows, r), dtype=torch.float32, device=dev))
    eng = HipEngine(X, row_ptr, r, f(len(J), r), f(N, r), f(K, r), [[nn(len(J))], [nn(N)], [nn(K)]])
    ints = lambda which: eng.internal(which).view(torch.int32).cpu().numpy().astype(np.int64)
    E = _engine_mod
    for (i_row0, i_n, i_ptr, cap, tol) in ((E.BUF_SEG_ROW0, E.BUF_SEG_NROWS, E.BUF_WAVE_SEG_PTR, 256, 0),
                                           (E.BUF_BSEG_ROW0, E.BUF_BSEG_NROWS, E.BUF_WAVE_BSEG_PTR, 512, 2)):
        row0, n, ptr = ints(i_row0), ints(i_n), ints(i_ptr)
        assert n.min() >= 1 and n.max() <= cap
        assert np.array_equal(row0, np.concatenate([[0], np.cumsum(n)[:-1]])) and row0[-1] + n[-1] == N  # a tiling of the rows
        slab_of = np.searchsorted(row_ptr, row0, side="right") - 1
        assert np.all(row0 + n <= row_ptr[slab_of + 1])                       # no unit crosses a slab boundary
        assert ptr[0] == 0 and ptr[-1] == len(n) and np.all(np.diff(ptr) >= 1) and len(ptr) - 1 <= 1024
        blocks = np.add.reduceat((n + 15) // 16, ptr[:-1])
        quota = -(-int(((J + 15) // 16).sum()) // 1024)
        assert blocks.max() <= quota + tol, (blocks.max(), quota)
        assert blocks.max() <= 1.06 * blocks[:-1].mean()                        # was 1.32 with a fixed count of units per wave
    eng.close()


def test_by_products_are_not_reused_out_of_order():
    """The sweep weights its [G | R] partials with the a_i of the moment and its M_i belongs to the B it wrote: an A update
    before the C-phase, a second C-phase, or new factors must all fall back to passes over X with current operands."""
    import copy
    import torch

    st = _state("k256_r16_nn", seed=5)
    ref = copy.deepcopy(st)
    eng = engine_from_oracle_state(st)
    r = st.A.shape[1]

    def gr_ref():
        Ba = np.concatenate([ref.B[ref.row_ptr[i]:ref.row_ptr[i + 1]] * ref.A[i] for i in range(ref.A.shape[0])])
        return Ba.T @ Ba, ref.X.T @ Ba

    # B, then A (unusual order), then C: the partials of the sweep carry the OLD a_i
    eng.update_B(); ref.update_B()
    eng.update_A(); ref.update_A()
    gr = to_np(eng.update_C_local())
    G, R = gr_ref()
    assert rel_err(gr[:r * r].reshape(r, r), G) < 1e-5 and rel_err(gr[r * r:].reshape(-1, r), R) < 1e-5
    eng.update_C_finish(); ref.update_C()
    assert rel_err(to_np(eng.C), ref.C) < 1e-5
    # A again with the new C but the same B: M_i is still valid, the right-hand sides must use the NEW C
    eng.update_A(); ref.update_A()
    assert rel_err(to_np(eng.A), ref.A) < 1e-5
    # C twice in a row
    for _ in range(2):
        gr = to_np(eng.update_C_local())
        G, R = gr_ref()
        assert rel_err(gr[r * r:].reshape(-1, r), R) < 1e-5
        eng.update_C_finish(); ref.update_C()
    assert rel_err(to_np(eng.C), ref.C) < 1e-5
    # a full iteration afterwards is back on the sweep and still right
    eng.update_B(); ref.update_B()
    gr = to_np(eng.update_C_local())
    eng.update_C_finish(); ref.update_C()
    eng.update_A(); ref.update_A()
    torch.cuda.synchronize()
    assert max(rel_err(to_np(eng.A), ref.A), rel_err(to_np(eng.B), ref.B), rel_err(to_np(eng.C), ref.C)) < 1e-5
    eng.close()


def test_shapes_without_a_sweep_instantiation_keep_the_two_pass_path():
    from oracle import aoadmm_oracle as orc

    for J, K, r in (([300, 200], 130, 8), ([300, 200], 1024, 16), ([20, 30, 10, 25], 256, 16), ([300, 200], 516, 16),
                    ([300, 200], 512, 32), ([300, 200], 256, 40)):
        X, row_ptr = orc.synthetic_problem(len(J), np.array(J), K, r, seed=0, dtype=np.float64)
        st = orc.random_state_for(X, row_ptr, r, [[NN], [NN], [NN]], seed=1)
        eng = engine_from_oracle_state(st)
        eng.update_B()
        assert eng.kernel_variant(_engine_mod.PROF_SWEEP) == "", (J, K, r, eng.kernel_variant(_engine_mod.PROF_SWEEP))
        eng.close()
