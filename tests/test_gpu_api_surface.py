"""-m gpu: the public API with the input kinds and corner shapes a drop-in user can throw at it."""
import os
import subprocess
import sys

import numpy as np
import pytest

from tests.helpers import rel_err, split_rows

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _data(I=5, J=20, K=12, r=3, seed=0):
    from oracle import aoadmm_oracle as orc

    X, row_ptr = orc.synthetic_problem(I, J, K, r, seed=seed, dtype=np.float64)
    return X, row_ptr, split_rows(X, row_ptr)


def test_input_kinds_give_the_same_answer():
    import torch
    from matcouply_amd import decomposition as dec

    X, row_ptr, mats = _data()
    kw = dict(non_negative=True, n_iter_max=5, tol=None, absolute_tol=None, random_state=3)
    ref = dec.cmf_aoadmm(mats, 3, **kw)
    assert isinstance(ref[1][0], np.ndarray) and ref[1][0].dtype == np.float64
    # float32 NumPy input -> float32 output
    out32 = dec.cmf_aoadmm([m.astype(np.float32) for m in mats], 3, **kw)
    assert out32[1][0].dtype == np.float32 and rel_err(out32[1][0], ref[1][0]) < 1e-5
    # 3-D array (equal shapes) iterates like a list of matrices
    out3d = dec.cmf_aoadmm(np.stack(mats), 3, **kw)
    assert rel_err(out3d[1][2], ref[1][2]) < 1e-6
    # torch CPU tensors -> torch CPU tensors; torch CUDA tensors -> CUDA tensors
    out_t = dec.cmf_aoadmm([torch.as_tensor(m) for m in mats], 3, **kw)
    assert isinstance(out_t[1][0], torch.Tensor) and not out_t[1][0].is_cuda and out_t[1][0].dtype == torch.float64
    assert rel_err(out_t[1][0].numpy(), ref[1][0]) < 1e-6
    out_c = dec.cmf_aoadmm([torch.as_tensor(m, dtype=torch.float32).cuda() for m in mats], 3, **kw)
    assert out_c[1][0].is_cuda and rel_err(out_c[1][0].cpu().numpy(), ref[1][0]) < 1e-5
    # data already packed in HBM
    packed = dec.PackedMatrices(torch.as_tensor(X, dtype=torch.float32).cuda(), row_ptr)
    assert len(packed) == 5 and packed[1].shape == (20, 12)
    out_p = dec.cmf_aoadmm(packed, 3, **kw)
    assert out_p[1][0].is_cuda and rel_err(out_p[1][0].cpu().numpy(), ref[1][0]) < 1e-5
    # weights of an explicit init are folded into A (decomposition.py:22-28)
    rs = np.random.RandomState(1)
    A0, B0, C0 = rs.uniform(size=(5, 3)), [rs.uniform(size=(20, 3)) for _ in range(5)], rs.uniform(size=(12, 3))
    w = np.array([2.0, 0.5, 1.5])
    a = dec.cmf_aoadmm(mats, 3, init=(w, (A0, B0, C0)), n_iter_max=2, tol=None, absolute_tol=None)
    b = dec.cmf_aoadmm(mats, 3, init=(None, (A0 * w, B0, C0)), n_iter_max=2, tol=None, absolute_tol=None)
    assert rel_err(a[1][0], b[1][0]) < 1e-7 and a[0] is None


@pytest.mark.parametrize("shape", [dict(I=1, J=9, K=5, r=2), dict(I=3, J=1, K=1, r=1), dict(I=2, J=70, K=3, r=3),
                                   dict(I=4, J=5, K=260, r=2), dict(I=2, J=3, K=7, r=6)])
def test_corner_shapes_against_oracle(shape):
    from oracle import aoadmm_oracle as orc
    from tests.test_gpu_end_to_end import _compare, _run_both

    X, row_ptr = orc.synthetic_problem(shape["I"], shape["J"], shape["K"], shape["r"], seed=1, dtype=np.float64)
    X = X.astype(np.float32).astype(np.float64)
    nn = {"kind": "nn"}
    st = orc.random_state_for(X, row_ptr, shape["r"], [[nn], [nn, {"kind": "l2ball", "norm_bound": 2.0}], [nn]], seed=2)
    cmf, admm, diag, res = _run_both(st, 3)
    _compare(cmf, admm, diag, st, res, 1e-5)


def test_svd_initialisations_run():
    from matcouply_amd import decomposition as dec

    _, _, mats = _data(I=4, J=15, K=10, r=3, seed=5)
    for init in ("svd", "threshold_svd"):
        cmf, diag = dec.cmf_aoadmm(mats, 3, init=init, non_negative=(init == "threshold_svd"), n_iter_max=30,
                                   tol=None, absolute_tol=None, return_errors=True)
        assert diag.rec_errors[-1] < 0.2 and np.isfinite(diag.regularized_loss).all()
    with pytest.raises(NotImplementedError):
        dec.cmf_aoadmm(mats, 3, init="parafac2_als", n_iter_max=1)


SHARDED = r'''
import os, sys, json
sys.path.insert(0, os.environ["REPO"])
import numpy as np, torch, torch.distributed as dist
from matcouply_amd import decomposition as dec, penalties as pen
from oracle import aoadmm_oracle as orc
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
J = np.array([40, 25, 64, 33, 90, 17])
X, row_ptr = orc.synthetic_problem(6, J, 24, 4, seed=0, dtype=np.float64)
mats = [X[row_ptr[i]:row_ptr[i+1]] for i in range(6)]
rs = np.random.RandomState(5)
A0, B0, C0 = rs.uniform(size=(6, 4)), [rs.uniform(size=(j, 4)) for j in J], rs.uniform(size=(24, 4))
mk = lambda shp: rs.uniform(size=shp)
auxA, dualA, auxC, dualC = mk((6, 4)), mk((6, 4)), mk((24, 4)), mk((24, 4))
P0 = [np.eye(j, 4) for j in J]; D0 = mk((4, 4)); dualP = [mk((j, 4)) for j in J]
auxL, dualL = [mk((j, 4)) for j in J], [mk((j, 4)) for j in J]
auxU, dualU = [mk((j, 4)) for j in J], [mk((j, 4)) for j in J]
README_STACK = os.environ.get("STACK") == "readme"  # README.rst:66-91 of the reference: L2 ball on A, unimodal B_i
TV_STACK = os.environ.get("STACK") == "tv"
EXT_STACK = os.environ.get("STACK") == "ext"
MATA_STACK = os.environ.get("STACK") == "matA"  # matrix penalties on the sharded A: evaluated on the all-gathered A + U
COUPLED_STACK = os.environ.get("STACK") == "coupledB"  # a user's MatricesPenalty that couples the B_i of ALL ranks
auxA2, dualA2 = mk((6, 4)), mk((6, 4))
class SharedScale(pen.MatricesPenalty):
    """one scale factor for all B_i (the stack projected on a Frobenius ball); the value is the norm of the stack: neither splits over ranks"""
    def factor_matrices_update(self, factor_matrices, feasibility_penalties, auxes):
        total = float(sum((m ** 2).sum() for m in factor_matrices)) ** 0.5
        scale = min(1.0, (2.0 + 0.1 * float(np.mean(feasibility_penalties))) / total)
        return [0.9 * scale * m + 0.1 * a for m, a in zip(factor_matrices, auxes)]
    def penalty(self, x):
        return 0.01 * float(sum((m ** 2).sum() for m in x)) ** 0.5
class Ridge(pen.MatrixPenalty):
    """alpha * ||x||^2: prox x / (1 + 2 alpha / rho)"""
    def __init__(self, alpha, aux_init="random_uniform", dual_init="random_uniform"):
        super().__init__(aux_init, dual_init)
        self.alpha = alpha
    def factor_matrix_update(self, factor_matrix, feasibility_penalty, aux):
        return factor_matrix / (1.0 + 2.0 * self.alpha / feasibility_penalty)
    def penalty(self, x):
        return sum(self.alpha * float((xi * xi).sum()) for xi in (x if isinstance(x, (list, tuple)) else [x]))
# STACK = "pf2_stop": the PARAFAC2 stack WITH a stopping rule - under sharding it is evaluated on the device too
# (mcl_gate_begin / mcl_verdict on the all-reduced vector; chunks of 8 iterations, gated kernels behind a hit)
STOP = dict(n_iter_max=60, tol=2e-2, absolute_tol=1e-12, feasibility_tol=float("inf")) if os.environ.get("STACK") == "pf2_stop" \
    else dict(n_iter_max=5, tol=None, absolute_tol=None)
def run(lo, hi, group):
    regs = [[pen.NonNegativity(aux_init=auxA[lo:hi].copy(), dual_init=dualA[lo:hi].copy())],
            [pen.Parafac2(aux_init=([p.copy() for p in P0[lo:hi]], D0.copy()), dual_init=[d.copy() for d in dualP[lo:hi]]),
             pen.L2Ball(1.0, aux_init=[a.copy() for a in auxL[lo:hi]], dual_init=[d.copy() for d in dualL[lo:hi]])],
            [pen.L1Penalty(0.05, non_negativity=True, aux_init=auxC.copy(), dual_init=dualC.copy())]]
    if TV_STACK:  # total variation on the B_i (value summed over the ranks with the diagnostics vector) and on the replicated C
        regs[1] = [pen.TotalVariationPenalty(0.05, aux_init=[a.copy() for a in auxL[lo:hi]], dual_init=[d.copy() for d in dualL[lo:hi]])]
        regs[2] = [pen.TotalVariationPenalty(0.03, l1_strength=0.02, aux_init=auxC.copy(), dual_init=dualC.copy())]
    if EXT_STACK:  # a user-defined MatrixPenalty on the B_i: prox and value evaluated on the host, per matrix, on every rank
        regs[1] = [Ridge(0.3, aux_init=[a.copy() for a in auxL[lo:hi]], dual_init=[d.copy() for d in dualL[lo:hi]])]
    if COUPLED_STACK:
        regs[1] = [SharedScale(aux_init=[a.copy() for a in auxL[lo:hi]], dual_init=[d.copy() for d in dualL[lo:hi]]),
                   pen.NonNegativity(aux_init=[a.copy() for a in auxU[lo:hi]], dual_init=[d.copy() for d in dualU[lo:hi]])]
    if MATA_STACK:
        regs[0] = [pen.Unimodality(non_negativity=True, aux_init=auxA[lo:hi].copy(), dual_init=dualA[lo:hi].copy()),
                   pen.TotalVariationPenalty(0.04, aux_init=auxA2[lo:hi].copy(), dual_init=dualA2[lo:hi].copy())]
    if README_STACK:
        regs[0] = [pen.L2Ball(1.0, non_negativity=True, aux_init=auxA[lo:hi].copy(), dual_init=dualA[lo:hi].copy())]
        regs[1].insert(1, pen.Unimodality(non_negativity=True, aux_init=[a.copy() for a in auxU[lo:hi]],
                                          dual_init=[d.copy() for d in dualU[lo:hi]]))
    return dec.cmf_aoadmm(mats[lo:hi], 4, init=(None, (A0[lo:hi].copy(), [b.copy() for b in B0[lo:hi]], C0.copy())), regs=regs,
                          return_errors=True, constant_feasibility_penalty=not (TV_STACK or EXT_STACK or COUPLED_STACK), group=group, **STOP)
bounds = [0, 2, 6]
cmf, diag = run(bounds[rank], bounds[rank + 1], dist.group.WORLD)
if rank == 0:
    ref_cmf, ref_diag = run(0, 6, None)
    err = dict(A=float(np.linalg.norm(cmf[1][0] - ref_cmf[1][0][:2]) / np.linalg.norm(ref_cmf[1][0][:2])),
               C=float(np.linalg.norm(cmf[1][2] - ref_cmf[1][2]) / np.linalg.norm(ref_cmf[1][2])),
               rec=float(max(abs(a - b) / b for a, b in zip(diag.rec_errors, ref_diag.rec_errors))),
               loss=float(max(abs(a - b) / b for a, b in zip(diag.regularized_loss, ref_diag.regularized_loss))))
    if STOP["tol"]:
        assert (diag.n_iter, diag.message) == (ref_diag.n_iter, ref_diag.message), (diag.n_iter, ref_diag.n_iter)
        assert 1 <= diag.n_iter < STOP["n_iter_max"] and diag.message.startswith("FEASIBILITY GAP CRITERION AND RELATIVE")
        assert len(diag.rec_errors) == diag.n_iter + 1
    print("SHARDED_RESULT " + json.dumps(err), flush=True)
dist.barrier()
dist.destroy_process_group()
'''


@pytest.mark.parametrize("stack", ["pf2", "readme", "pf2_stop", "tv", "ext", "matA", "coupledB"])
def test_two_ranks_sharing_the_gpu_equal_single_process(tmp_path, stack):
    """cmf_aoadmm(group=) with the REAL engine: 2 processes share cuda:0, collectives over gloo (RCCL refuses two ranks on one
    device); PARAFAC2 + constant feasibility penalty exercise every reduction of the step path; the "readme" stack adds the
    L2 ball on the sharded A (all-reduced column norms) and unimodality on the B_i."""
    script = tmp_path / "sharded.py"
    script.write_text(SHARDED)
    env = dict(os.environ, REPO=REPO, MASTER_ADDR="127.0.0.1", STACK=stack,
               MASTER_PORT=str(29600 + (os.getpid() + 7 * len(stack)) % 300))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", env["MASTER_PORT"], str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    line = [l for l in out.stdout.splitlines() if l.startswith("SHARDED_RESULT")]
    assert line, out.stdout[-2000:] + out.stderr[-3000:]
    import json

    err = json.loads(line[0].split(" ", 1)[1])
    assert max(err.values()) < 1e-5, err



@pytest.mark.parametrize("stack", ["pf2", "pf2_ball"])
def test_parafac2_polar_factor_routes_agree(stack):
    """PARAFAC2's polar factors (penalties.py:1224-1250) through both native routes - Newton-Schulz on the fp64 MFMA and the
    Jacobi eigen-solver that takes over for slabs it cannot handle - against the oracle: a slab with fewer rows than the
    rank is rank-deficient by construction (pseudo-inverse square root), and MCL_PF2_JACOBI=1 forces the solver everywhere."""
    import torch

    from oracle import aoadmm_oracle as orc
    from tests.helpers import engine_from_oracle_state, rel_err, to_np

    r, K = 6, 20
    J = np.array([3, 40, 17, 6, 64, 5, 130])  # slabs 0 and 5 have fewer rows than the rank
    X, row_ptr = orc.synthetic_problem(len(J), J, K, r, seed=4, dtype=np.float64)
    X = X.astype(np.float32).astype(np.float64)
    regs = [[{"kind": "nn"}], [{"kind": "parafac2"}] + ([{"kind": "l2ball", "norm_bound": 1.5}] if stack == "pf2_ball" else []),
            [{"kind": "nn"}]]
    want = orc.random_state_for(X, row_ptr, r, regs, seed=5)
    want.update_B()
    saved = os.environ.pop("MCL_PF2_JACOBI", None)
    try:
        for forced in (False, True):
            if forced:
                os.environ["MCL_PF2_JACOBI"] = "1"
            st = orc.random_state_for(X, row_ptr, r, regs, seed=5)
            eng = engine_from_oracle_state(st)
            eng.update_B()
            torch.cuda.synchronize()
            errs = dict(B=rel_err(to_np(eng.B), want.B), P=rel_err(to_np(eng.regs[1][0].aux), want.aux[1][0][0]),
                        Delta=rel_err(to_np(eng.regs[1][0].aux2), want.aux[1][0][1]),
                        dual=rel_err(to_np(eng.regs[1][0].dual), want.dual[1][0]))
            assert max(errs.values()) < 1e-5, (forced, errs)
            eng.close()
    finally:
        os.environ.pop("MCL_PF2_JACOBI", None)
        if saved is not None:
            os.environ["MCL_PF2_JACOBI"] = saved


def test_step_api_defers_and_merges_the_finish_pass(fast_kernels):
    """The B-phase through the step calls (what a multi-GPU host drives) must leave exactly the state of the single-call
    mcl_update_B: the library defers the fused prox + dual row pass of an inner iteration and merges it with the next
    mcl_B_solve; mcl_B_end (or any other entry point) issues it when no solve follows."""
    import torch

    from oracle import aoadmm_oracle as orc
    from tests.helpers import engine_from_oracle_state

    r, K = 8, 40
    J = np.array([70, 33, 129, 64, 18, 200])
    X, row_ptr = orc.synthetic_problem(len(J), J, K, r, seed=3, dtype=np.float64)
    X = X.astype(np.float32).astype(np.float64)
    regs = [[{"kind": "nn"}], [{"kind": "parafac2"}, {"kind": "unimodal", "non_negativity": True},
                               {"kind": "l2ball", "norm_bound": 1.2, "non_negativity": True}], [{"kind": "nn"}]]

    def state():
        return engine_from_oracle_state(orc.random_state_for(X, row_ptr, r, regs, seed=6))

    ref = state()
    ref.update_B()
    torch.cuda.synchronize()
    for explicit_end in (True, False):
        eng = state()
        eng.B_begin()
        eng.B_factor()
        for _ in range(5):
            eng.B_solve()
            for k in range(3):
                eng.B_prox_local(k)
                eng.B_prox_finish(k)
        if explicit_end:
            eng.B_end()
        else:
            eng.update_C_local()  # any other entry point issues the pending pass first
        torch.cuda.synchronize()
        assert torch.equal(eng.B, ref.B)
        for k in range(3):
            assert torch.equal(eng.regs[1][k].aux, ref.regs[1][k].aux), k
            assert torch.equal(eng.regs[1][k].dual, ref.regs[1][k].dual), k
        assert torch.equal(eng.regs[1][0].aux2, ref.regs[1][0].aux2)
        eng.close()
    ref.close()


@pytest.mark.parametrize("config", ["c3_8th", "c5_32nd"])
def test_bench_two_rank_rehearsal(config):
    """bench.py's multi-rank path (sharding, [G | R] all-reduce, PARAFAC2 step calls, max-over-ranks timing, one JSON line
    from rank 0) with two ranks sharing cuda:0 over gloo - RCCL refuses two ranks on one device, so this rehearses
    everything but the collective backend the driver's --gpus N runs use."""
    import json

    env = dict(os.environ, MCL_BENCH_SHARE_GPU="1", MCL_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    port = str(29700 + (os.getpid() + len(config)) % 200)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--config", config]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:] + out.stderr[-3000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["value"] > 0 and d["scaling"] == "strong"
    assert d["roofline"]["achieved"] > 0 and np.isfinite(d["final_rel_rec_error"])


RCCL_ONE_RANK = r'''
import os, sys, json
sys.path.insert(0, os.environ["REPO"])
import numpy as np, torch, torch.distributed as dist
from matcouply_amd import decomposition as dec
from oracle import aoadmm_oracle as orc
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))  # backend "nccl" IS RCCL on ROCm
J = np.array([40, 25, 64, 33, 90, 17])
X, row_ptr = orc.synthetic_problem(6, J, 24, 4, seed=0, dtype=np.float64)
mats = [X[row_ptr[i]:row_ptr[i+1]] for i in range(6)]
kw = dict(non_negative={0: True}, parafac2=True, l2_norm_bound={1: 1.0}, l1_penalty={2: 0.05}, n_iter_max=5, tol=None,
          absolute_tol=None, return_errors=True, constant_feasibility_penalty=True, random_state=0)
cmf_g, diag_g = dec.cmf_aoadmm(mats, 4, group=dist.group.WORLD, **kw)   # step path: every reduction goes through RCCL
direct = dec._direct_comm(dist.group.WORLD)                              # ... called directly on the engine's stream
print("RCCL_DIRECT " + json.dumps(dict(available=direct is not None, calls=(direct.calls if direct is not None else 0))), flush=True)
os.environ["MCL_NO_DIRECT_RCCL"] = "1"                                  # the same through torch.distributed's own stream
dec._DIRECT_COMMS.clear()
cmf_t, diag_t = dec.cmf_aoadmm(mats, 4, group=dist.group.WORLD, **kw)
assert dec._direct_comm(dist.group.WORLD) is None
print("RCCL_PATHS_EQUAL " + json.dumps(bool(np.array_equal(cmf_t[1][0], cmf_g[1][0]) and np.array_equal(cmf_t[1][2], cmf_g[1][2]))), flush=True)
os.environ.pop("MCL_FORCE_SHARDED_PATH")
cmf_1, diag_1 = dec.cmf_aoadmm(mats, 4, **kw)                           # single-call path
err = dict(A=float(np.linalg.norm(cmf_g[1][0] - cmf_1[1][0]) / np.linalg.norm(cmf_1[1][0])),
           C=float(np.linalg.norm(cmf_g[1][2] - cmf_1[1][2]) / np.linalg.norm(cmf_1[1][2])),
           rec=float(max(abs(a - b) / b for a, b in zip(diag_g.rec_errors, diag_1.rec_errors))))
print("RCCL_RESULT " + json.dumps(err), flush=True)
dist.barrier()
dist.destroy_process_group()
'''


def test_step_path_over_rccl_on_one_rank(tmp_path):
    """The collectives of the sharded path on the REAL backend: `init_process_group("nccl", device_id=...)` (= RCCL) with a
    world of one rank on this box's GPU - fp64 SUM all-reduce of [G | R] on a view of the library's workspace, fp32 MAX
    all-reduces of the feasibility penalties, the PARAFAC2 reduction per inner iteration, the fp64 diagnostics vector, a
    barrier.  (Two ranks need two devices: RCCL refuses to share one; that case runs over gloo above.)"""
    script = tmp_path / "rccl_one.py"
    script.write_text(RCCL_ONE_RANK)
    port = str(29300 + os.getpid() % 300)
    env = dict(os.environ, REPO=REPO, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY="0",
               MCL_FORCE_SHARDED_PATH="1")  # a one-rank group takes the step path with every reduction in it
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", port, str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    line = [l for l in out.stdout.splitlines() if l.startswith("RCCL_RESULT")]
    assert line, out.stdout[-2000:] + out.stderr[-3000:]
    import json

    err = json.loads(line[0].split(" ", 1)[1])
    assert max(err.values()) < 1e-5, err
    # the collectives went through the engine's own communicator (RCCL on the engine's stream), and give the bits of the
    # torch.distributed path
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("RCCL_DIRECT")][0].split(" ", 1)[1])
    assert d["available"] and d["calls"] > 20, d
    assert json.loads([l for l in out.stdout.splitlines() if l.startswith("RCCL_PATHS_EQUAL")][0].split(" ", 1)[1]) is True
