"""Soak of the round-5 paths through the PUBLIC call (GPU box): many small random configurations (the fp64 inner loops of the
exact-products mode, the device-side inner stopping test, the device svd initialiser, the native GeneralizedL2 / UnitSimplex
kinds), each run to 60 iterations - outputs finite, no growth of the device memory in use - then one long run of a small problem.
    python tools/soak_r5.py [n_cases=300]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from matcouply_amd import decomposition as dec, penalties as pen
from tests.test_gpu_fuzz_parity import _draw_case

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda", 0)


def build(d, shape_rows):
    k = d["kind"]
    if k == "nn":
        return pen.NonNegativity()
    if k == "l1":
        return pen.L1Penalty(d["reg_strength"], non_negativity=d.get("non_negativity", False))
    if k == "box":
        return pen.Box(d["min_val"], d["max_val"])
    if k == "l2ball":
        return pen.L2Ball(d["norm_bound"], non_negativity=d.get("non_negativity", False))
    if k == "unimodal":
        return pen.Unimodality(non_negativity=d.get("non_negativity", False))
    if k == "parafac2":
        return pen.Parafac2()
    raise ValueError(k)


def used():
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    return total - free


bad, t0 = 0, time.perf_counter()
base = None
for seed in range(n_cases):
    rng = np.random.RandomState(1000 + seed)
    case = _draw_case(rng)
    I, K, r = case["I"], case["K"], case["r"]
    g = np.random.RandomState(seed)
    A, C = g.uniform(0.1, 1.1, (I, r)), g.uniform(size=(K, r))
    mats = [((g.uniform(size=(int(j), r)) * A[i]) @ C.T + 0.05 * g.standard_normal((int(j), K))).astype(np.float32)
            for i, j in enumerate(case["J"])]
    regs = [[build(d, None) for d in case["regs"][m]] for m in range(3)]
    kw = {}
    flavour = seed % 5
    if flavour == 1:
        kw = dict(inner_tol=1e-3, inner_n_iter_max=8)
    elif flavour == 2 and min(case["J"]) >= r and K >= r and not any(d["kind"] == "parafac2" for d in case["regs"][1]):
        kw = dict(init="svd")
        mats = [torch.as_tensor(m, device=dev) for m in mats]
    elif flavour == 3 and not case["regs"][2]:
        n = K
        regs[2] = [pen.GeneralizedL2Penalty(0.3 * (2 * np.eye(n) - np.eye(n, k=1) - np.eye(n, k=-1)) + 0.05 * np.eye(n)), pen.UnitSimplex()]
    elif flavour == 4:
        kw = dict(tol=1e-6, absolute_tol=1e-9)
    const = case["const"]
    try:
        cmf, diag = dec.cmf_aoadmm(mats, r, regs=regs, n_iter_max=60, return_errors=True, random_state=seed,
                                   l2_penalty=list(case["l2"]), feasibility_penalty_scale=case["scale"],
                                   constant_feasibility_penalty=const, **({"tol": None, "absolute_tol": None} | kw))
    except Exception as e:  # noqa: BLE001
        bad += 1
        print(f"seed {seed}: {type(e).__name__}: {str(e)[:300]}", flush=True)
        continue
    w, (A_, B_, C_) = cmf
    fin = all(np.isfinite(np.asarray(x.cpu() if torch.is_tensor(x) else x)).all() for x in [A_, C_] + list(B_)) and np.isfinite(
        np.asarray(diag.rec_errors)).all() and np.isfinite(np.asarray(diag.regularized_loss)).all()
    if not fin:
        bad += 1
        print(f"seed {seed}: non-finite output ({case['regs']}, {kw})", flush=True)
    del cmf, diag, mats
    if seed == 20:
        torch.cuda.empty_cache()
        base = used()
    if seed % 50 == 49:
        torch.cuda.empty_cache()
        print(f"  {seed + 1} cases, {bad} bad, device memory in use {used() / 2**20:.0f} MiB (after case 20: {base / 2**20:.0f}), "
              f"{time.perf_counter() - t0:.0f} s", flush=True)
torch.cuda.empty_cache()
grow = used() - base
print(f"{n_cases} cases: {bad} bad; device memory in use grew by {grow / 2**20:.1f} MiB since case 20", flush=True)

# one long run of a small problem on the default path (fp64 inner loops; ring wrap-around of the diagnostics, no drift)
g = np.random.RandomState(7)
I, J, K, r = 16, 96, 64, 4
A, C = g.uniform(0.1, 1.1, (I, r)), g.uniform(size=(K, r))
mats = [((g.uniform(size=(J, r)) * A[i]) @ C.T + 0.05 * g.standard_normal((J, K))).astype(np.float32) for i in range(I)]
t1 = time.perf_counter()
cmf, diag = dec.cmf_aoadmm(mats, r, n_iter_max=30000, tol=None, absolute_tol=None, return_errors=True, random_state=0, non_negative=True,
                           l1_penalty={2: 0.01})
torch.cuda.synchronize()
re = np.asarray(diag.rec_errors)
print(f"long run: {diag.n_iter} iterations in {time.perf_counter() - t1:.1f} s; rec_error {re[0]:.4f} -> {re[-1]:.6f}; finite {np.isfinite(re).all()}; "
      f"largest increase between iterations {np.max(np.diff(re)):.2e}", flush=True)
sys.exit(1 if bad or grow > 64 * 2**20 else 0)
