"""CPU study (NumPy): which fp32 roundings of the PARAFAC2 + L2-ball B-phase (BASELINE config 4's stack) carry the per-phase
error of B against the all-fp64 arithmetic of the reference.  Emulates the engine's inner loop with a switch per rounding
site; prints the relative error of B / the auxiliary matrix P Delta / the dual after ONE B-phase (5 inner iterations) from
identical fp32-representable inputs.  Usage: python tools/pf2_rounding_study.py [I]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import aoadmm_oracle as orc  # noqa: E402

f32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)


def mm32(a, b):
    """product with fp32 inputs and fp32 accumulation (what an fp32 MFMA chain does, up to the order of the sums)"""
    return (np.asarray(a, dtype=np.float32) @ np.asarray(b, dtype=np.float32)).astype(np.float64)


def polar_T(S, Delta):
    """T = Delta^T (Delta S Delta^T)^-1/2 in fp64 (the engine's Newton-Schulz result)"""
    G = Delta @ S @ Delta.T
    w, V = np.linalg.eigh(0.5 * (G + G.T))
    return Delta.T @ (V * (1.0 / np.sqrt(w))) @ V.T


def b_phase(st, flags):
    X, A, C, rp = st.X, st.A, st.C, st.row_ptr
    r, I = A.shape[1], st.I
    rnd = lambda name, a: f32(a) if flags.get(name, True) else a
    mm = lambda name, a, b: mm32(a, b) if flags.get(name, True) else a @ b
    CtC = C.T @ C
    XC = np.concatenate([mm("xc_acc", X[rp[i]:rp[i + 1]], C) for i in range(I)], 0)
    XC = rnd("xc_store", XC)
    n = 2
    B = st.B.copy()
    P, Delta = st.aux[1][0][0].copy(), st.aux[1][0][1].copy()
    Upf, Zl2, Ul2 = st.dual[1][0].copy(), st.aux[1][1].copy(), st.dual[1][1].copy()
    bound = st.regs[1][1]["norm_bound"]
    rho = np.empty(I)
    Linv = []
    for i in range(I):
        L = CtC * np.outer(A[i], A[i])
        rho[i] = 0.5 * np.trace(L) * st.scale
        Linv.append(rnd("linv", np.linalg.inv(L + rho[i] * n * np.eye(r))))
    Zpf = rnd("z_store", mm("z_acc", P, Delta))
    for _ in range(st.inner):
        accD, Bn, Pn = np.zeros((r, r)), np.empty_like(B), np.empty_like(B)
        for i in range(I):
            s, e = rp[i], rp[i + 1]
            V = rnd("v", rho[i] * ((Zpf[s:e] - Upf[s:e]) + (Zl2[s:e] - Ul2[s:e])) + XC[s:e] * A[i])
            Bn[s:e] = rnd("st_B", mm("solve_acc", V, Linv[i]))  # B is stored in fp32
            Y = rnd("y_sum", Bn[s:e] + Upf[s:e])
            T = rnd("t_store", polar_T(Y.T @ Y, Delta))
            Pn[s:e] = rnd("st_P", mm("p_acc", Y, T))
            accD += rho[i] * (Pn[s:e].T @ Y)
        B, P = Bn, Pn
        Delta = rnd("st_Delta", accD / rho.sum())
        Zpf = rnd("z_store", mm("z_acc", P, Delta))
        Upf = rnd("st_Upf", B - (Zpf - Upf))
        for i in range(I):
            s, e = rp[i], rp[i + 1]
            Y = B[s:e] + Ul2[s:e]
            nrm = np.sqrt((Y ** 2).sum(0))
            Zl2[s:e] = rnd("st_Zl2", Y * (bound / np.maximum(nrm, bound)))
        Ul2 = rnd("st_Ul2", B - (Zl2 - Ul2))
    return B, Zpf, Upf


def main():
    I = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    J = np.random.RandomState(0).randint(128, 1025, I)
    regs = [[], [{"kind": "parafac2"}, {"kind": "l2ball", "norm_bound": 1.0}], []]
    X, rp = orc.synthetic_problem(I, J, 256, 16, seed=0, dtype=np.float64)
    st = orc.random_state_for(f32(X), rp, 16, regs, seed=1)
    for name in ("A", "B", "C"):
        setattr(st, name, f32(getattr(st, name)))
    st.aux[1][0] = (f32(st.aux[1][0][0]), f32(st.aux[1][0][1]))
    st.dual[1][0], st.aux[1][1], st.dual[1][1] = f32(st.dual[1][0]), f32(st.aux[1][1]), f32(st.dual[1][1])
    import copy

    ref = copy.deepcopy(st)
    ref.update_B()
    Zref = ref.aux[1][0][0] @ ref.aux[1][0][1]
    sites = ["xc_acc", "xc_store", "linv", "v", "solve_acc", "y_sum", "t_store", "p_acc", "z_acc", "z_store"]
    err = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)

    def report(label, flags):
        B, Z, U = b_phase(st, flags)
        print(f"{label:42s} B {err(B, ref.B):.2e}   P Delta {err(Z, Zref):.2e}   dual {err(U, ref.dual[1][0]):.2e}", flush=True)

    report("every site fp32 (the engine today)", {})
    report("storage roundings only (all sites exact)", {k: False for k in sites})
    for k in sites:
        report(f"all fp32 except {k}", {k: False})
    for k in sites:
        report(f"only {k} fp32", {s: (s == k) for s in sites})
    report("exact: y_sum t_store p_acc", {"y_sum": False, "t_store": False, "p_acc": False})
    report("exact: y_sum t_store p_acc solve_acc v linv", {k: False for k in ("y_sum", "t_store", "p_acc", "solve_acc", "v", "linv")})
    report("exact: all but xc_acc xc_store", {k: False for k in sites if not k.startswith("xc")})
    # composition of the storage floor: every compute site exact, ONE stored array kept exact as well
    stores = ["st_B", "st_P", "st_Delta", "st_Upf", "st_Zl2", "st_Ul2"]
    exact = {k: False for k in sites}
    for k in stores:
        report(f"compute exact, storage fp32 except {k}", dict(exact, **{k: False}))
    report("compute exact, B and U_pf2 stored exactly", dict(exact, st_B=False, st_Upf=False))
    report("compute exact, every array stored exactly", dict(exact, **{k: False for k in stores}))


if __name__ == "__main__":
    main()
