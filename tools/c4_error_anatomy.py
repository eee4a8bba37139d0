"""Where the error of A lives in the down-scaled config-4 parity case (tests: c4_ragged): per-slab relative error of a_i after
3 iterations next to the condition number of that slab's A-phase system Q_i = B_i^T B_i o C^T C (oracle, fp64) and J_i.
GPU box: python tools/c4_error_anatomy.py [I]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import aoadmm_oracle as orc
from tests.test_gpu_end_to_end import _run_both

I = int(sys.argv[1]) if len(sys.argv) > 1 else 48
regs = [[], [{"kind": "parafac2"}, {"kind": "l2ball", "norm_bound": 1.0}], []]
J = np.random.RandomState(0).randint(128, 1025, I)
X, row_ptr = orc.synthetic_problem(I, J, 256, 16, seed=0, dtype=np.float64)
X = X.astype(np.float32).astype(np.float64)
st = orc.random_state_for(X, row_ptr, 16, regs, seed=1)
for n_it in (1, 2, 3):
    import copy
    s2 = copy.deepcopy(st)
    cmf, admm, diag, res = _run_both(s2, n_it)
    A_g, A_o = np.asarray(cmf[1][0]), s2.A
    per = np.linalg.norm(A_g - A_o, axis=1) / np.linalg.norm(A_o, axis=1)
    tot = np.linalg.norm(A_g - A_o) / np.linalg.norm(A_o)
    CtC = s2.C.T @ s2.C
    conds = []
    for i in range(I):
        Bi = s2.B[row_ptr[i]:row_ptr[i + 1]]
        conds.append(np.linalg.cond((Bi.T @ Bi) * CtC))
    conds = np.array(conds)
    share = np.sort((np.linalg.norm(A_g - A_o, axis=1) ** 2))[::-1]
    share = share / share.sum()
    worst = np.argsort(-per)[:5]
    print(f"after {n_it} iterations: ||dA||/||A|| = {tot:.2e}; B {np.linalg.norm(np.concatenate(cmf[1][1]) - s2.B) / np.linalg.norm(s2.B):.2e}; "
          f"share of the squared error in the worst 1 / 3 / 10 slabs: {share[0]:.2f} / {share[:3].sum():.2f} / {share[:10].sum():.2f}")
    print("   cond(Q_i): median %.1e max %.1e; worst slabs (slab, J_i, rel err a_i, cond Q_i): %s" % (
        np.median(conds), conds.max(), [(int(i), int(J[i]), float(f"{per[i]:.1e}"), float(f"{conds[i]:.1e}")) for i in worst]))
    print("   corr(log err, log cond) = %.2f" % np.corrcoef(np.log(per + 1e-300), np.log(conds))[0, 1])
