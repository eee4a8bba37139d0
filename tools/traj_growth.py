"""Error growth along a golden trajectory (GPU box): engine vs oracle after n = 1 .. N outer iterations from the golden's
initial state, next to the worst polar-factor condition number the oracle met.  usage: traj_growth.py traj_c5_full.npz"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import aoadmm_oracle as orc  # noqa: E402
from tests.helpers import load_npz, rel_err  # noqa: E402
from tests.test_gpu_end_to_end import _run_both  # noqa: E402
from tests.test_oracle_golden import _traj_state  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "traj_c5_full.npz"
arrs = load_npz(name)
spec = json.loads(str(arrs["spec"]))
for n in (1, 2, 3, 5, 8, 12, 16, 20):
    st = _traj_state(arrs, spec)
    cmf, admm, diag, res = _run_both(st, n)
    e = {"A": rel_err(cmf[1][0], st.A), "B": rel_err(np.concatenate(cmf[1][1]), st.B), "C": rel_err(cmf[1][2], st.C)}
    print(n, {k: f"{v:.1e}" for k, v in e.items()}, f"polar cond {res['polar_cond']:.1e}", f"rec {diag.rec_errors[-1]:.6f}", flush=True)

# where does the error of the last run live?  per (slab, column) share of ||B - B_ref||^2: a discontinuous projection
# (unimodal regression picks a split index) puts almost all of it into the few columns whose split differs
Bg, rp = np.concatenate(cmf[1][1]), st.row_ptr
err = np.array([[np.sum((Bg[rp[i]:rp[i + 1], c] - st.B[rp[i]:rp[i + 1], c]) ** 2) for c in range(st.B.shape[1])]
                for i in range(st.I)])
order = np.argsort(err.ravel())[::-1]
tot = err.sum()
print("share of the squared B error in the worst 1 / 3 / 10 of", err.size, "(slab, column) pairs:",
      [round(float(err.ravel()[order[:k]].sum() / tot), 3) for k in (1, 3, 10)])
for k_, d_ in enumerate(st.regs[1]):
    if d_["kind"] == "unimodal":
        Zg, Zr = np.concatenate(admm.auxes[1][k_]), st.aux[1][k_]
        flips = [(int(i), int(c)) for i in range(st.I) for c in range(Zr.shape[1])
                 if np.argmax(Zg[rp[i]:rp[i + 1], c]) != np.argmax(Zr[rp[i]:rp[i + 1], c])]
        print("unimodal aux: columns whose peak position differs from the reference's:", flips[:10], "of", st.I * Zr.shape[1])
