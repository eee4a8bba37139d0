"""-m gpu: contract of the C ABI as include/matcouply_hip.h publishes it to hosts other than the in-repo Python driver."""
import ctypes

import numpy as np
import pytest

from matcouply_amd import _engine as _engine_mod
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


def test_deferred_diagnostics_equal_immediate_ones(fast_kernels):
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
        assert eng.kernel_variant(_engine_mod.PROF_SWEEP).startswith("k_sweep<")
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


def _sweep_problem():
    from oracle import aoadmm_oracle as orc

    nn = {"kind": "nn"}
    regs = [[nn], [nn], [{"kind": "l1", "reg_strength": 0.05, "non_negativity": True}]]
    J = np.full(12, 96)
    X, row_ptr = orc.synthetic_problem(len(J), J, 64, 8, seed=5, dtype=np.float64)  # sweep-eligible shape
    return orc, regs, X.astype(np.float32).astype(np.float64), row_ptr


@pytest.mark.parametrize("how", ["set_workspace", "destroy", "set_factors"])
def test_deferred_diagnostics_survive_a_setter_or_destroy(how):
    """a pending deferred reduction must be issued by ANY other entry point - also by the setters
    that replace what it refers to and by mcl_destroy - while the old workspace is still the valid one."""
    import torch
    from matcouply_amd._engine import DIAG_LEN

    orc, regs, X, row_ptr = _sweep_problem()
    st = orc.random_state_for(X, row_ptr, 8, regs, seed=6)
    eng = engine_from_oracle_state(st)
    eng.update_B(); eng.update_C_local(); eng.update_C_finish(); eng.update_A()
    want = eng.diagnostics().cpu().numpy()
    out = torch.full((DIAG_LEN,), float("nan"), dtype=torch.float64, device="cuda")
    eng.diagnostics_deferred(out=out)
    torch.cuda.synchronize()
    assert bool(torch.isnan(out).all())  # really deferred
    if how == "set_workspace":
        nbytes = eng.lib.mcl_workspace_bytes(eng._h)
        ws2 = torch.empty(int(nbytes) + 256, dtype=torch.uint8, device="cuda")
        off = (-ws2.data_ptr()) % 256
        eng._check(eng.lib.mcl_set_workspace(eng._h, ws2.data_ptr() + off, nbytes))
        old, eng.workspace = eng.workspace, ws2
        old.zero_()  # the old workspace may be re-used by the host at once
    elif how == "set_factors":
        eng._check(eng.lib.mcl_set_factors(eng._h, eng.A.data_ptr(), eng.B.data_ptr(), eng.C.data_ptr()))
    else:
        eng.close()
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=1e-13, atol=1e-300)
    eng.close()


def test_a_deferral_crosses_at_most_one_sweep(fast_kernels):
    """two mcl_update_B in a row with a deferral pending - the second sweep would overwrite the mode-1
    table the deferral recorded; the vector must still describe the iterate at the time of the deferring call."""
    import torch
    from matcouply_amd._engine import DIAG_LEN

    orc, regs, X, row_ptr = _sweep_problem()
    st = orc.random_state_for(X, row_ptr, 8, regs, seed=6)
    eng = engine_from_oracle_state(st)
    eng.update_B(); eng.update_C_local(); eng.update_C_finish(); eng.update_A()
    assert eng.kernel_variant(_engine_mod.PROF_SWEEP).startswith("k_sweep<")
    want = eng.diagnostics().cpu().numpy()
    out = torch.full((DIAG_LEN,), float("nan"), dtype=torch.float64, device="cuda")
    eng.diagnostics_deferred(out=out)
    eng.update_B()
    eng.update_B()  # no C-phase in between
    eng.update_C_local(); eng.update_C_finish()
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=1e-13, atol=1e-300)
    eng.close()


def test_library_reports_the_header_abi_version():
    from matcouply_amd import _engine

    assert _engine.load_library().mcl_version() == _engine.MCL_ABI_VERSION == 410


def test_events_order_a_side_stream_collective():
    """mcl_record_event / mcl_wait_event: the C-phase reduction handed to ANOTHER stream (where a host's collective library
    runs) and back - the finish must see what the side stream wrote into [G | R]."""
    import torch

    orc, regs, X, row_ptr = _sweep_problem()
    results = []
    for side in (False, True):
        st = orc.random_state_for(X, row_ptr, 8, regs, seed=6)
        eng = engine_from_oracle_state(st)
        comm = torch.cuda.Stream()
        for _ in range(3):
            eng.update_B()
            gr = eng.update_C_local()
            if side:  # the "collective": a scaling by 1 + 2^-20 on the communication stream, in place
                e1, e2 = torch.cuda.Event(), torch.cuda.Event()
                e1.record()  # lazily creates the event; the engine then records it on ITS stream
                eng.record_event(e1)
                with torch.cuda.stream(comm):
                    comm.wait_event(e1)
                    torch.cuda._sleep(2000000)  # the side stream is slow: the finish has to wait for it
                    gr.mul_(1.0 + 2.0 ** -20)
                    e2.record(comm)
                eng.wait_event(e2)
            else:
                gr.mul_(1.0 + 2.0 ** -20)
            eng.update_C_finish(); eng.update_A()
        torch.cuda.synchronize()
        results.append(to_np(eng.C))
        eng.close()
    assert np.array_equal(results[0], results[1])


RELEASE_BUILD = r'''
import os, sys, json, warnings
sys.path.insert(0, os.environ["REPO"])
import numpy as np, torch
from matcouply_amd import _engine
_engine.LIB_PATH = os.environ["MCL_TEST_LIB"]          # the release build of this test, not the in-tree library
import bench
os.environ["MCL_NO_SWEEP"] = "1"                        # a kernel-form switch a debug build obeys
os.environ["MCL_RUN_WATCHDOG_S"] = "7"
cfg = dict(bench.CONFIGS["c2"], I=64)                 # 2 M elements of X: above the exact-products size, the sweep is planned
dev = torch.device("cuda", 0)
X, row_ptr, I_loc = bench.make_shard(cfg, 0, 1, dev)
with warnings.catch_warnings(record=True) as caught:
    warnings.simplefilter("always")
    eng = bench.make_engine(cfg, X, row_ptr, I_loc, 0, dev)
eng.iterate(2)
torch.cuda.synchronize()
print("RELEASE " + json.dumps(dict(active=eng.lib.mcl_active_switches(eng._h).decode(), sweep=eng.kernel_variant(_engine.PROF_SWEEP),
                                   warned=[str(w.message)[:60] for w in caught if "MCL_" in str(w.message)],
                                   lib=os.path.realpath(eng.lib._name))), flush=True)
'''


def test_release_build_ignores_the_environment_switches(tmp_path):
    """-DMCL_NO_ENV_SWITCHES (MCL_BUILD_DEFS of matcouply_amd/_build.py): the build a host embeds never consults the MCL_*
    environment - a kernel-form switch is ignored, mcl_active_switches() reports nothing, no warning is raised.  Only
    api.hip reads the environment, so the test compiles that one file and links it with the in-tree objects."""
    import json
    import os
    import subprocess
    import sys

    from matcouply_amd import _build

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = _build.build_library(defs=["-DMCL_NO_ENV_SWITCHES"], out_lib=str(tmp_path / "libmatcouply_hip_release.so"),
                               build_dir=str(tmp_path), only=["api.hip"], verbose=False)
    script = tmp_path / "release.py"
    script.write_text(RELEASE_BUILD)
    out = subprocess.run([sys.executable, str(script)], env=dict(os.environ, REPO=repo, MCL_TEST_LIB=lib), capture_output=True,
                         text=True, timeout=300)
    line = [l for l in out.stdout.splitlines() if l.startswith("RELEASE ")]
    assert line, out.stdout[-1500:] + out.stderr[-3000:]
    d = json.loads(line[0].split(" ", 1)[1])
    assert d["lib"] == os.path.realpath(lib)
    assert d["active"] == "" and d["warned"] == []
    assert d["sweep"].startswith("k_sweep<"), d  # MCL_NO_SWEEP=1 was not obeyed: the one-pass sweep is still planned
    # the in-tree (debug-capable) build does obey it - the switch is real
    dbg = subprocess.run([sys.executable, str(script)], env=dict(os.environ, REPO=repo, MCL_TEST_LIB=_build.LIB),
                         capture_output=True, text=True, timeout=300)
    d2 = json.loads([l for l in dbg.stdout.splitlines() if l.startswith("RELEASE ")][0].split(" ", 1)[1])
    assert "MCL_NO_SWEEP" in d2["active"] and d2["sweep"] == "" and d2["warned"]


def test_round5_entry_points_refuse_bad_arguments():
    """mcl_penalty_value / mcl_svd_init / mcl_set_options(inner_tol, exact_products) / GeneralizedL2 descriptors: errors are
    return codes with a message, never a fault"""
    import torch

    from matcouply_amd import _engine as E

    orc, regs, X, row_ptr = _sweep_problem()
    st = orc.random_state_for(X, row_ptr, 8, regs, seed=6)
    eng = engine_from_oracle_state(st)
    out = torch.zeros(1, dtype=torch.float64, device="cuda")
    assert eng.lib.mcl_penalty_value(eng._h, 0, 0, out.data_ptr()) != 0          # not a GeneralizedL2 penalty
    assert b"GeneralizedL2" in eng.lib.mcl_last_error(eng._h)
    assert eng.lib.mcl_penalty_value(eng._h, 5, 0, out.data_ptr()) != 0 and eng.lib.mcl_penalty_value(eng._h, 0, 9, out.data_ptr()) != 0
    opt = E.Options()
    opt.feasibility_penalty_scale, opt.inner_n_iter_max = 1.0, 5
    opt.exact_products = 3
    assert eng.lib.mcl_set_options(eng._h, ctypes.byref(opt)) != 0
    opt.exact_products, opt.inner_tol = 0, -1.0
    assert eng.lib.mcl_set_options(eng._h, ctypes.byref(opt)) != 0
    # a GeneralizedL2 descriptor without its matrix, or with a row count that is not the mode's
    d = (E.PenaltyDesc * 1)()
    d[0].kind, d[0].aux, d[0].dual = E.PEN_GL2, eng.regs[2][0].aux.data_ptr(), eng.regs[2][0].dual.data_ptr()
    assert eng.lib.mcl_set_penalties(eng._h, 2, 1, d) != 0
    mat = torch.zeros(2 * 9 + 3, dtype=torch.float64, device="cuda")
    d[0].matrix, d[0].matrix_rows = mat.data_ptr(), 3
    assert eng.lib.mcl_set_penalties(eng._h, 2, 1, d) != 0 and b"matrix_rows" in eng.lib.mcl_last_error(eng._h)
    eng.close()
    # the stateless svd initialiser
    Xd = torch.rand((40, 16), device="cuda")
    rp = np.array([0, 25, 40], dtype=np.int64)
    with pytest.raises(E.EngineError, match="rank exceeds"):
        E.svd_init(Xd, rp, 17)
    with pytest.raises(E.EngineError, match="fewer rows"):
        E.svd_init(Xd, np.array([0, 37, 40], dtype=np.int64), 5)
    B, C, info = E.svd_init(Xd, rp, 4)
    assert B.shape == (40, 4) and C.shape == (16, 4) and int(info.min()) > 0
