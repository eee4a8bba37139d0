"""-m gpu: contract of the C ABI as include/matcouply_hip.h publishes it to hosts other than the in-repo Python driver."""
import ctypes

import numpy as np
import pytest

from tests.helpers import engine_from_oracle_state, rel_err, to_np

pytestmark = pytest.mark.gpu


def _state(regs, seed=3):
    from oracle import aoadmm_oracle as orc

    J = np.array([70, 33, 128, 65, 90])
    X, row_ptr = orc.synthetic_problem(len(J), J, 48, 6, seed=seed, dtype=np.float64)
    X = X.astype(np.float32).astype(np.float64)
    return orc.random_state_for(X, row_ptr, 6, regs, seed=seed + 1)


def test_workspace_size_query_leaves_an_installed_workspace_usable():
    """mcl_workspace_bytes() after mcl_set_workspace() - a size re-check a C host may well make - must not disturb
    the context: the next launches use the installed buffers and give the same answer as without the query."""
    import copy
    import torch

    nn = {"kind": "nn"}
    st = _state([[nn], [nn], [{"kind": "l1", "reg_strength": 0.05}]])
    ref = copy.deepcopy(st)
    eng = engine_from_oracle_state(st)
    eng.iterate(1)
    n1 = eng.lib.mcl_workspace_bytes(eng._h)
    assert n1 > 0
    eng.iterate(1)                                   # would dereference NULL workspace pointers if the query reset them
    assert eng.lib.mcl_workspace_bytes(eng._h) == n1
    eng.update_B()
    torch.cuda.synchronize()
    for _ in range(2):
        ref.update_B(); ref.update_C(); ref.update_A()
    ref.update_B()
    assert rel_err(to_np(eng.B), ref.B) < 1e-5 and rel_err(to_np(eng.C), ref.C) < 1e-5
    eng.close()


def test_incomplete_step_round_is_an_error():
    """step API on a fusable stack (PARAFAC2 + L2 ball): skipping a penalty between two solves must fail loudly instead
    of leaving stale aux / dual rows."""
    from matcouply_amd._engine import EngineError

    st = _state([[{"kind": "nn"}], [{"kind": "parafac2"}, {"kind": "l2ball", "norm_bound": 1.0}], [{"kind": "nn"}]])
    eng = engine_from_oracle_state(st)
    eng.B_begin(); eng.B_factor()
    eng.B_solve()
    eng.B_prox_local(0); eng.B_prox_finish(0)        # penalty 1 is never stepped
    with pytest.raises(EngineError, match="did not step every penalty"):
        eng.B_solve()
    # a complete round is accepted
    eng.B_begin(); eng.B_factor(); eng.B_solve()
    for k in range(2):
        eng.B_prox_local(k); eng.B_prox_finish(k)
    eng.B_end()
    eng.close()


def test_deferred_diagnostics_equal_immediate_ones():
    """mcl_diagnostics_deferred: the reduction rides on the next C-phase reduction kernel (sweep path) or is flushed by the
    next other entry point; the vector must describe the factors at the time of the call and equal mcl_diagnostics."""
    import torch
    from matcouply_amd._engine import DIAG_LEN

    nn = {"kind": "nn"}
    regs = [[nn], [nn], [{"kind": "l1", "reg_strength": 0.05, "non_negativity": True}]]
    from oracle import aoadmm_oracle as orc

    J = np.full(12, 96)
    X, row_ptr = orc.synthetic_problem(len(J), J, 64, 8, seed=5, dtype=np.float64)  # sweep-eligible shape
    X = X.astype(np.float32).astype(np.float64)

    def run(deferred, n=4):
        st = orc.random_state_for(X, row_ptr, 8, regs, seed=6)
        eng = engine_from_oracle_state(st)
        ring = torch.full((n, DIAG_LEN), float("nan"), dtype=torch.float64, device="cuda")
        for it in range(n):
            eng.update_B(); eng.update_C_local(); eng.update_C_finish(); eng.update_A()
            (eng.diagnostics_deferred if deferred else eng.diagnostics)(out=ring[it])
        assert eng.kernel_variant(3).startswith("k_sweep<")
        if deferred:
            torch.cuda.synchronize()
            assert bool(torch.isnan(ring[n - 1]).all())       # the last one is still pending ...
            eng.flush_diagnostics()                            # ... until flushed (or any other call)
        torch.cuda.synchronize()
        out = ring.cpu().numpy()
        eng.close()
        return out

    a, b = run(False), run(True)
    assert np.isfinite(b).all()
    np.testing.assert_allclose(b, a, rtol=1e-13, atol=1e-300)
    # iterate() defers between its iterations and flushes at the end
    st = orc.random_state_for(X, row_ptr, 8, regs, seed=6)
    eng = engine_from_oracle_state(st)
    ring = torch.zeros((4, DIAG_LEN), dtype=torch.float64, device="cuda")
    eng.iterate(4, diag_ring=ring)
    torch.cuda.synchronize()
    np.testing.assert_allclose(ring.cpu().numpy(), a, rtol=1e-13, atol=1e-300)
    eng.close()
