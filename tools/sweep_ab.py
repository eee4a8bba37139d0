"""A/B check of the one-pass sweep (sweep.hip) against the two-pass path (MCL_NO_SWEEP=1) on the GPU."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from matcouply_amd import decomposition as dec, penalties as pen  # noqa: E402
from oracle import aoadmm_oracle as orc  # noqa: E402


def run(mats, r, regs_fn, n_iter, init, **kw):
    return dec.cmf_aoadmm(mats, r, init=init, regs=regs_fn(), n_iter_max=n_iter, tol=None, absolute_tol=None,
                          return_errors=True, return_admm_vars=True, random_state=7, **kw)


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


CASES = [
    dict(I=6, J=[300, 64, 1100, 257, 80, 513], K=256, r=16, regs="nn"),
    dict(I=5, J=[200, 90, 700, 333, 128], K=512, r=16, regs="nn"),
    dict(I=4, J=[150, 400, 260, 96], K=512, r=12, regs="l1box"),
    dict(I=4, J=[150, 400, 260, 96], K=256, r=32, regs="nn"),
    dict(I=4, J=[150, 400, 260, 96], K=256, r=20, regs="none"),
    dict(I=3, J=[2100, 1024, 64], K=512, r=8, regs="nn", constant=True),
]

for case in CASES:
    J = np.array(case["J"])
    X, row_ptr = orc.synthetic_problem(case["I"], J, case["K"], case["r"], seed=1, dtype=np.float64)
    mats = [X[row_ptr[i]:row_ptr[i + 1]].astype(np.float32) for i in range(case["I"])]
    rs = np.random.RandomState(0)
    r = case["r"]
    init = (None, (rs.uniform(size=(case["I"], r)), [rs.uniform(size=(j, r)) for j in J], rs.uniform(size=(case["K"], r))))

    def regs_fn():
        if case["regs"] == "nn":
            return [[pen.NonNegativity()], [pen.NonNegativity()], [pen.NonNegativity()]]
        if case["regs"] == "l1box":
            return [[pen.NonNegativity()], [pen.L1Penalty(0.02), pen.Box(-0.5, 2.0)], [pen.L1Penalty(0.01, non_negativity=True)]]
        return [[], [], []]

    kw = dict(constant_feasibility_penalty=True) if case.get("constant") else {}
    out = {}
    for mode in ("sweep", "twopass"):
        if mode == "twopass":
            os.environ["MCL_NO_SWEEP"] = "1"
        else:
            os.environ.pop("MCL_NO_SWEEP", None)
        out[mode] = run(mats, r, regs_fn, 6, init, **kw)
    os.environ.pop("MCL_NO_SWEEP", None)
    (cs, (auxs, duals), ds), (ct, (auxt, dualt), dt) = out["sweep"], out["twopass"]
    errs = dict(A=rel(cs[1][0], ct[1][0]), C=rel(cs[1][2], ct[1][2]),
                B=max(rel(a, b) for a, b in zip(cs[1][1], ct[1][1])),
                rec=max(abs(a - b) / b for a, b in zip(ds.rec_errors, dt.rec_errors)),
                loss=max(abs(a - b) / max(abs(b), 1e-30) for a, b in zip(ds.regularized_loss, dt.regularized_loss)))
    if auxs[1]:
        errs["auxB"] = max(rel(a, b) for a, b in zip(auxs[1][0], auxt[1][0]))
        errs["dualB"] = max(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), np.linalg.norm(f), 1e-30)
                            for a, b, f in zip(duals[1][0], dualt[1][0], ct[1][1]))
        fe = max(abs(a - b) / max(b, 1e-12) for a, b in zip(np.ravel([g[1] for g in ds.feasibility_gaps]),
                                                             np.ravel([g[1] for g in dt.feasibility_gaps])))
        errs["gapB"] = float(fe)
    print(case["K"], r, case["regs"], {k: f"{v:.2e}" for k, v in errs.items()}, "rec", f"{ds.rec_errors[-1]:.4f}", flush=True)
