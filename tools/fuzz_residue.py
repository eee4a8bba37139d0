"""GPU: the fuzz draws that sit outside the flat 1e-5 (tests/test_gpu_fuzz_parity.py) under variants of the comparison:
   as-is      the oracle starts from the fp64 draw, the engine from its fp32 rounding (what the test does)
   same-init  both start from the fp32-representable state (identical inputs)
   python tools/fuzz_residue.py [seed ...]   (default: the pinned residue of the 2000-draw sweep)"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault("MATCOUPLY_AMD_TEST_ENGINE", "0")
from tests.test_gpu_end_to_end import _compare, _run_both  # noqa: E402
from tests.test_gpu_fuzz_parity import RESIDUE_SEEDS, _draw_case  # noqa: E402
from oracle import aoadmm_oracle as orc  # noqa: E402


def f32(x):
    return np.asarray(x, dtype=np.float64).astype(np.float32).astype(np.float64)


def state(seed, same_init):
    case = _draw_case(np.random.RandomState(1000 + seed))
    X, row_ptr = orc.synthetic_problem(case["I"], case["J"], case["K"], case["r"], seed=seed, dtype=np.float64)
    X = f32(X)
    st = orc.random_state_for(X, row_ptr, case["r"], case["regs"], seed=seed + 1, l2=case["l2"],
                              inner_n_iter_max=case["inner"], feasibility_penalty_scale=case["scale"],
                              constant_A=case["const"], constant_B=case["const"])
    if same_init:
        st.A, st.B, st.C = f32(st.A), f32(st.B), f32(st.C)
        for m in range(3):
            for k in range(len(st.aux[m])):
                z = st.aux[m][k]
                st.aux[m][k] = (f32(z[0]), f32(z[1])) if isinstance(z, tuple) else f32(z)
                st.dual[m][k] = f32(st.dual[m][k])
    return case, st


if __name__ == "__main__":
    seeds = [int(s) for s in sys.argv[1:]] or RESIDUE_SEEDS
    worst_of = lambda e: max((v, k) for k, v in e.items() if not (k[0] == "P" and k[1] != "D"))
    for seed in seeds:
        row = []
        for same in (False, True):
            case, st = state(seed, same)
            cmf, admm, diag, res = _run_both(st, 2)
            errs = _compare(cmf, admm, diag, st, res, 1.0, 1.0)
            w, k = worst_of(errs)
            row.append(f"{w:.1e} ({k})")
        print(f"seed {seed:5d} r={case['r']:2d} K={case['K']:3d} inner={case['inner']} B:{[d['kind'] for d in case['regs'][1]]}: "
              f"as-is {row[0]:18s} same-init {row[1]:18s}", flush=True)
