"""Per-phase parity probe (GPU box): for every outer iteration and phase the engine is rebuilt from the oracle's fp64
state (rounded to fp32), the phase runs on both sides and the relative errors of everything the phase writes are
printed next to the condition number of the phase's normal equations.  Localises WHERE a trajectory loses digits.
Test infrastructure: imports oracle/ (allowed for tools that check, never for the product path).

    python tools/parity_probe.py [case ...]      cases: names of tests.test_gpu_end_to_end.SCALE_CASES
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle import aoadmm_oracle as orc  # noqa: E402
from tests.helpers import engine_from_oracle_state, rel_err, to_np  # noqa: E402
from tests.test_gpu_end_to_end import SCALE_CASES  # noqa: E402


def case_state(name):
    if name.startswith("fuzz:"):  # a case of tests/test_gpu_fuzz_parity.py by seed
        from tests.test_gpu_fuzz_parity import _draw_case

        seed, _, nth = name[5:].partition(":")  # fuzz:<seed>[:<n-th draw of that seed's generator>]
        seed, rng = int(seed), np.random.RandomState(1000 + int(seed))
        for _ in range(int(nth or 0) + 1):
            case = _draw_case(rng)
        print({k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in case.items()})
        X, row_ptr = orc.synthetic_problem(case["I"], case["J"], case["K"], case["r"], seed=seed, dtype=np.float64)
        X = X.astype(np.float32).astype(np.float64)
        return orc.random_state_for(X, row_ptr, case["r"], case["regs"], seed=seed + 1, l2=case["l2"],
                                    inner_n_iter_max=case["inner"], feasibility_penalty_scale=case["scale"],
                                    constant_A=case["const"], constant_B=case["const"])
    cfg = SCALE_CASES[name]
    J = cfg["J"]
    if isinstance(J, str):
        J = {"ragged": np.random.RandomState(0).randint(128, 1025, cfg["I"]), "c5dims": np.array([2048, 700, 33, 1024, 515, 64]),
             "c5stack": np.array([2048, 700, 100, 1024, 515, 130]),
             "odd": np.array([1, 3, 64, 65, 17, 130, 5, 63, 2])}[J]
    X, row_ptr = orc.synthetic_problem(cfg["I"], J, cfg["K"], cfg["r"], seed=0, dtype=np.float64)
    X = X.astype(np.float32).astype(np.float64)
    return orc.random_state_for(X, row_ptr, cfg["r"], cfg["regs"], seed=1)


def f32(st):
    """round the oracle's state to what the engine will hold"""
    r = lambda a: np.asarray(a, np.float32).astype(np.float64)
    st.A, st.B, st.C = r(st.A), r(st.B), r(st.C)
    for m in range(3):
        st.aux[m] = [(r(z[0]), r(z[1])) if isinstance(z, tuple) else r(z) for z in st.aux[m]]
        st.dual[m] = [r(u) for u in st.dual[m]]


def probe(name, n_iter=3):
    st = case_state(name)
    print(f"== {name}: I={st.I} rows={st.X.shape[0]} K={st.X.shape[1]} r={st.A.shape[1]}")
    for it in range(n_iter):
        for phase in "BCA":
            f32(st)
            eng = engine_from_oracle_state(st)
            if phase == "B":
                st.update_B(); eng.update_B()
                e = {"B": rel_err(to_np(eng.B), st.B)}
                for k_, d_ in enumerate(st.regs[1]):
                    if d_["kind"] == "parafac2":
                        Y = st.B + st.dual[1][k_]
                        D_ = st.aux[1][k_][1]
                        e["cond_pf2"] = float(max(np.linalg.cond(Y[st.row_ptr[i]:st.row_ptr[i + 1]] @ D_.T) for i in range(st.I)))
                CtC = st.C.T @ st.C
                L = CtC[None] * st.A[:, :, None] * st.A[:, None, :] + (st.rho_B * len(st.regs[1]) + st.l2[1])[:, None, None] * np.eye(st.A.shape[1])
                e["cond_max"] = float(np.max(np.linalg.cond(L)))
            elif phase == "C":
                G, R = st.local_C_normal_equations()
                gr = to_np(eng.update_C_local())
                r_ = st.A.shape[1]
                e = {"G": rel_err(gr[: r_ * r_].reshape(r_, r_), G), "R": rel_err(gr[r_ * r_:].reshape(-1, r_), R)}
                st.update_C(); eng.update_C_finish()
                e["C"] = rel_err(to_np(eng.C), st.C)
                e["cond"] = float(np.linalg.cond(G + (st.rho_C * len(st.regs[2]) + st.l2[2]) * np.eye(r_)))
            else:
                rhs, Q = st.update_A(); eng.update_A()
                e = {"A": rel_err(to_np(eng.A), st.A), "rhs": rel_err(to_np(eng.rhses()), rhs), "Q": rel_err(to_np(eng.cross_products()), Q)}
                Lc = Q + (st.rho_A * len(st.regs[0]) + st.l2[0])[:, None, None] * np.eye(st.A.shape[1])
                e["cond_max"] = float(np.max(np.linalg.cond(Lc)))
                d = to_np(eng.diagnostics())
                rec = np.sqrt(max(0.0, d[5] - 2 * d[3] + d[4])) / np.sqrt(d[5])
                e["rec"] = abs(rec - st.rec_error_from_A_byproducts()) / st.rec_error_from_A_byproducts()
            for m, tag in ((1, "B"), (2, "C"), (0, "A")):
                if tag != phase:
                    continue
                for k, d_ in enumerate(st.regs[m]):
                    z = eng.regs[m][k]
                    if d_["kind"] == "parafac2":
                        e[f"P{k}"] = rel_err(to_np(z.aux), st.aux[m][k][0])
                        e[f"D{k}"] = rel_err(to_np(z.aux2), st.aux[m][k][1])
                    else:
                        e[f"aux{k}"] = rel_err(to_np(z.aux), st.aux[m][k])
            print(f"  it {it} {phase}: " + "  ".join(f"{k} {v:.1e}" for k, v in e.items()), flush=True)
            eng.close()


if __name__ == "__main__":
    for name in (sys.argv[1:] or ["c4_ragged", "r64", "c2_full"]):
        probe(name)
