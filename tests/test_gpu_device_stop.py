"""-m gpu: the stopping rule evaluated ON THE DEVICE (mcl_run; the path behind every cmf_aoadmm call that has a tolerance
set, i.e. the DEFAULT call) against
  (a) the reference's stopping matrix (tests/golden/stopping.json: message, n_iter, list lengths incl. quirks Q8-Q10),
  (b) the host-evaluated rule on the same device (blocking read-back per iteration): same stopping iteration, and - the
      point of the gated kernels - BIT-identical factors and ADMM variables however far the host ran ahead."""
import json
import os

import numpy as np
import pytest

from tests.helpers import GOLDEN, engine_from_oracle_state, load_npz, rel_err, split_rows

pytestmark = pytest.mark.gpu


@pytest.fixture
def run_counter(monkeypatch):
    """counts the calls of HipEngine.run (= mcl_run) so that a test can assert which path a call took"""
    from matcouply_amd import _engine

    calls = []
    orig = _engine.HipEngine.run

    def counted(self, *a, **kw):
        out = orig(self, *a, **kw)
        calls.append(out[:2])
        return out

    monkeypatch.setattr(_engine.HipEngine, "run", counted)
    return calls


def _host_rule(monkeypatch):
    """the same call with the rule evaluated by the host loop (what round 2 did for every call with a tolerance)"""
    from matcouply_amd import _engine

    monkeypatch.delattr(_engine.HipEngine, "run")


def _matrix_cases():
    data = load_npz("stopping_data.npz")
    with open(os.path.join(GOLDEN, "stopping.json")) as f:
        results = json.load(f)
    return data, results


def _matrix_call(data, res):
    from matcouply_amd import penalties as pen

    row_ptr = data["row_ptr"]
    dec_ = lambda v: None if v is None else (float(v) if isinstance(v, str) else v)
    case = {k: dec_(v) for k, v in res["case"].items()}
    return_errors = case.pop("return_errors", True)
    case["n_iter_max"] = int(case["n_iter_max"])
    regs = [[pen.NonNegativity(aux_init=(split_rows(data[f"aux{m}"], row_ptr) if m == 1 else data[f"aux{m}"].copy()),
                               dual_init=(split_rows(data[f"dual{m}"], row_ptr) if m == 1 else data[f"dual{m}"].copy()))]
            for m in range(3)]
    kw = dict(init=(None, (data["A0"].copy(), split_rows(data["B0"], row_ptr), data["C0"].copy())), regs=regs,
              return_errors=return_errors, return_admm_vars=True, **case)
    return split_rows(data["X"], row_ptr), kw, return_errors


def _state_of(out):
    cmf, admm = out[0], out[1]
    parts = [cmf[1][0], np.concatenate(cmf[1][1]), cmf[1][2]]
    for m in range(3):
        for z, u in zip(admm.auxes[m], admm.duals[m]):
            if isinstance(z, tuple):  # PARAFAC2: (bases, coordinate matrix)
                parts += [np.concatenate(z[0]), np.asarray(z[1])]
            else:
                parts.append(np.concatenate(z) if m == 1 else z)
            parts.append(np.concatenate(u) if m == 1 else u)
    return [np.asarray(p) for p in parts]


def test_stopping_matrix_on_the_device(run_counter, monkeypatch):
    from matcouply_amd import decomposition as dec

    data, results = _matrix_cases()
    for res in results:
        mats, kw, return_errors = _matrix_call(data, res)
        n_before = len(run_counter)
        if "raises" in res:
            with pytest.raises(eval(res["raises"])):
                dec.cmf_aoadmm(mats, 2, **kw)
            continue
        out = dec.cmf_aoadmm(mats, 2, **kw)
        active = bool(kw["tol"] or kw["absolute_tol"]) and kw["n_iter_max"] > 0
        assert (len(run_counter) > n_before) == active, res["case"]  # every call with a tolerance takes the device path
        if not return_errors:
            # fp32 engine: the iteration the relative criterion fires on is not the reference's; the state must be sane
            assert np.isfinite(np.sum(out[0][1][0]))
            continue
        diag = out[2]
        fp32_robust = res["case"]["tol"] != 1e-08 or kw["n_iter_max"] <= 0  # tol = 1e-8 is below fp32 resolution
        if fp32_robust:
            assert diag.message == res["message"] and diag.n_iter == res["n_iter"], res["case"]
            assert (len(diag.rec_errors), len(diag.regularized_loss), len(diag.feasibility_gaps)) == \
                (res["n_rec"], res["n_loss"], res["n_gaps"]), res["case"]
            assert diag.satisfied_stopping_condition == res["satisfied_stopping_condition"], res["case"]
            feas = diag.satisfied_feasibility_condition
            assert (None if feas is None else bool(feas)) == res["satisfied_feasibility_condition"], res["case"]
            np.testing.assert_allclose(diag.rec_errors[-1], res["last_rec"], rtol=1e-5)
        else:
            assert len(diag.rec_errors) == len(diag.regularized_loss) == len(diag.feasibility_gaps) == diag.n_iter + 1
            assert 1 <= diag.n_iter <= res["n_iter"]


@pytest.mark.parametrize("case", ["default_tol", "loose_tol_no_errors", "absolute"])
def test_device_rule_stops_where_the_host_rule_stops(case, run_counter, monkeypatch):
    """same call, rule on the device vs rule on the host: same n_iter / message / lists, bit-identical final state"""
    from matcouply_amd import decomposition as dec

    data, results = _matrix_cases()
    mats, kw, _ = _matrix_call(data, results[0])
    if case == "default_tol":
        kw.update(tol=1e-8, absolute_tol=1e-10, feasibility_tol=1e-4, n_iter_max=400, return_errors=True)
    elif case == "loose_tol_no_errors":  # Q10: no loss on infeasible iterates, "previous loss" = previous computed one
        kw.update(tol=1e-2, absolute_tol=1e-10, feasibility_tol=5e-2, n_iter_max=200, return_errors=False)
    else:
        kw.update(tol=1e-12, absolute_tol=1e-3, feasibility_tol=1e-1, n_iter_max=300, return_errors=True)
    dev = dec.cmf_aoadmm(mats, 2, **kw)
    assert len(run_counter) >= 1
    _host_rule(monkeypatch)
    mats, kw2, _ = _matrix_call(data, results[0])
    kw2.update({k: kw[k] for k in ("tol", "absolute_tol", "feasibility_tol", "n_iter_max", "return_errors")})
    host = dec.cmf_aoadmm(mats, 2, **kw2)
    for a, b in zip(_state_of(dev), _state_of(host)):
        assert np.array_equal(a, b)
    if kw["return_errors"]:
        d, h = dev[2], host[2]
        assert (d.n_iter, d.message, d.satisfied_stopping_condition) == (h.n_iter, h.message, h.satisfied_stopping_condition)
        assert bool(d.satisfied_feasibility_condition) == bool(h.satisfied_feasibility_condition)
        np.testing.assert_allclose(d.rec_errors, h.rec_errors, rtol=1e-12)
        np.testing.assert_allclose(d.regularized_loss, h.regularized_loss, rtol=1e-12)
        if case == "absolute":
            assert d.message.startswith("FEASIBILITY GAP CRITERION AND ABSOLUTE") and d.n_iter < kw["n_iter_max"]


def _stack_state(stack, seed=11):
    from oracle import aoadmm_oracle as orc

    if stack == "sweep":  # row-separable stack on a sweep-eligible shape (the path of BASELINE configs 2 / 3)
        J, K, r = np.full(12, 96), 64, 8
        regs = [[{"kind": "nn"}], [{"kind": "nn"}], [{"kind": "l1", "reg_strength": 0.05, "non_negativity": True}]]
    elif stack == "pf2":  # BASELINE config 4's stack: chained generic passes, Newton-Schulz, penalty-free A and C
        J, K, r = np.array([70, 33, 128, 65, 90, 40]), 48, 4
        regs = [[], [{"kind": "parafac2"}, {"kind": "l2ball", "norm_bound": 1.0}], []]
    else:  # BASELINE config 5's full stack
        J, K, r = np.array([70, 33, 128, 65, 90, 40]), 48, 4
        regs = [[{"kind": "nn"}],
                [{"kind": "parafac2"}, {"kind": "unimodal", "non_negativity": True},
                 {"kind": "l2ball", "norm_bound": 1.0, "non_negativity": True}],
                [{"kind": "l1", "reg_strength": 0.1, "non_negativity": True}]]
    X, row_ptr = orc.synthetic_problem(len(J), J, K, r, seed=seed, dtype=np.float64)
    X = X.astype(np.float32).astype(np.float64)
    return orc.random_state_for(X, row_ptr, r, regs, seed=seed + 1), regs


def _engine_state(eng):
    import torch

    torch.cuda.synchronize()
    parts = [eng.A, eng.B, eng.C]
    for m in range(3):
        for reg in eng.regs[m]:
            parts += [reg.aux, reg.dual] + ([reg.aux2] if reg.aux2 is not None else [])
    return [p.detach().cpu().numpy().copy() for p in parts]


@pytest.mark.parametrize("stack", ["sweep", "pf2", "full"])
def test_run_ahead_leaves_the_state_of_the_stopping_iteration(stack, kernel_paths):
    """C ABI level: mcl_run with run-ahead 1 / 8 / 64 stops on the same iteration and leaves the same bits as stepping
    exactly that many iterations with mcl_iterate - the gated kernels of the iterations enqueued behind the verdict did
    nothing - for the sweep path, the PARAFAC2 chain and the full penalty stack (every kernel family that writes state)"""
    st, regs = _stack_state(stack)
    w = [[(d.get("reg_strength", 0.0) if d["kind"] == "l1" else 0.0) for d in regs[m]] for m in range(3)]
    eng0 = engine_from_oracle_state(st)
    init = eng0.diagnostics().cpu().numpy()
    loss0 = 0.5 * max(0.0, init[5] - 2 * init[3] + init[4]) / init[5]
    eng0.close()
    outs = {}
    for ahead in (1, 8, 64):
        eng = engine_from_oracle_state(st)
        n, code, ring, verdict = eng.run(60, tol=3e-2, absolute_tol=1e-12, feasibility_tol=float("inf"), initial_loss=loss0,
                                         penalty_weight=w, evaluate_loss_always=True, max_run_ahead=ahead)
        outs[ahead] = (n, code, ring, verdict, _engine_state(eng))
        eng.close()
    n, code = outs[1][:2]
    assert code == 1 and 1 <= n < 60, (n, code)  # a relative-criterion stop somewhere inside the budget
    for ahead in (8, 64):
        assert outs[ahead][:2] == (n, code)
        np.testing.assert_array_equal(outs[ahead][2], outs[1][2])
        np.testing.assert_array_equal(outs[ahead][3], outs[1][3])
        for a, b in zip(outs[ahead][4], outs[1][4]):
            assert np.array_equal(a, b)
    # the verdict rows: every iteration evaluated, only the last one carries the stop code, losses descend to the stop
    v = outs[8][3]
    assert (v[:-1, 3].astype(int) >> 2 == 0).all() and int(v[-1, 3]) >> 2 == 1
    prev = v[-2, 1] if n > 1 else loss0
    assert abs(prev - v[-1, 1]) < 3e-2 * prev
    # reference run: exactly n iterations without any rule
    eng = engine_from_oracle_state(st)
    eng.iterate(n)
    for a, b in zip(_engine_state(eng), outs[64][4]):
        assert np.array_equal(a, b)
    # a context that stopped early is still usable: one more iteration from the stopping state = n + 1 plain iterations
    eng.iterate(1)
    want = _engine_state(eng)
    eng.close()
    eng = engine_from_oracle_state(st)
    eng.run(60, tol=3e-2, absolute_tol=1e-12, feasibility_tol=float("inf"), initial_loss=loss0, penalty_weight=w,
            evaluate_loss_always=True, max_run_ahead=64)
    eng.iterate(1)
    for a, b in zip(_engine_state(eng), want):  # (the B-phase systems are rebuilt by another kernel: equal to rounding)
        assert rel_err(a, b) < 1e-6
    eng.close()


def test_run_needs_pinned_status_memory():
    import ctypes
    import torch
    from matcouply_amd import _engine

    st, _ = _stack_state("sweep")
    eng = engine_from_oracle_state(st)
    rule = _engine.StopRule()
    ring = torch.zeros((2, _engine.DIAG_LEN), dtype=torch.float64, device="cuda")
    verdict = torch.zeros((2, 4), dtype=torch.float64, device="cuda")
    pageable = torch.zeros(4, dtype=torch.int32)
    rc = eng.lib.mcl_run(eng._h, 2, 1, 1, 1, ctypes.byref(rule), ring.data_ptr(), verdict.data_ptr(), pageable.data_ptr())
    assert rc != 0 and b"pinned" in eng.lib.mcl_last_error(eng._h)
    eng.close()


WATCHDOG = r'''
import os, sys, json, time
sys.path.insert(0, os.environ["REPO"])
import torch
import bench
from matcouply_amd import _engine
cfg = dict(bench.CONFIGS["c2"], I=32)
dev = torch.device("cuda", 0)
X, row_ptr, I_loc = bench.make_shard(cfg, 0, 1, dev)
eng = bench.make_engine(cfg, X, row_ptr, I_loc, 0, dev)
eng.iterate(1)
torch.cuda.synchronize()
t0 = time.perf_counter(); torch.cuda._sleep(20000000); torch.cuda.synchronize(); per_cycle = (time.perf_counter() - t0) / 2e7
torch.cuda._sleep(int(float(os.environ["SLEEP_S"]) / per_cycle))   # the stream stays busy: no verdict can arrive in time
t0 = time.perf_counter()
err = None
try:
    eng.run(200, 1e-8, 1e-10, 1e-4, initial_loss=1.0, penalty_weight=[[0.0], [0.0], [0.0]], evaluate_loss_always=True)
except _engine.EngineError as e:
    err = str(e)
dt = time.perf_counter() - t0
later = None
try:
    eng.update_B()
except _engine.EngineError as e:
    later = str(e)
t1 = time.perf_counter()
torch.cuda.synchronize()   # the device itself is fine: the enqueued work drains
drain = time.perf_counter() - t1
print("WATCHDOG " + json.dumps(dict(err=err, seconds=dt, later=later, drain=drain)), flush=True)
'''


def test_run_watchdog_returns_without_waiting_for_the_stream(tmp_path):
    """the watchdog of mcl_run must END the call - not fall through to a stream synchronisation that blocks on the
    very stream it gave up on.  A long sleep kernel in front of the run (and verdict kernels that report into scratch,
    MCL_TEST_MUTE_VERDICT) keeps every verdict away; the call has to come back after MCL_RUN_WATCHDOG_S, well before the
    stream drains, with an error, and the context refuses every later call."""
    import subprocess
    import sys

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "watchdog.py"
    script.write_text(WATCHDOG)
    env = dict(os.environ, REPO=repo, MCL_TEST_MUTE_VERDICT="1", MCL_RUN_WATCHDOG_S="0.4", SLEEP_S="3.0")
    out = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=300)
    line = [l for l in out.stdout.splitlines() if l.startswith("WATCHDOG ")]
    assert line, out.stdout[-1500:] + out.stderr[-3000:]
    d = json.loads(line[0].split(" ", 1)[1])
    assert d["err"] and "MCL_RUN_WATCHDOG_S" in d["err"] and "WITHOUT synchronising" in d["err"], d
    assert 0.3 < d["seconds"] < 1.5, d            # ~0.4 s of watchdog, not the ~3 s the sleep kernel holds the stream
    assert d["drain"] > 0.5, d                     # ... which was indeed still busy when the call returned
    assert d["later"] and "failed state" in d["later"], d
