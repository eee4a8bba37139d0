"""Shared helpers for the test-suite (fixture loading, packed <-> list conversion)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_npz(name):
    with np.load(os.path.join(GOLDEN, name), allow_pickle=False) as f:
        return {k: f[k] for k in f.files}


def manifest_of(arrs, key="manifest"):
    return json.loads(str(arrs[key]))


def split_rows(packed, row_ptr):
    return [np.array(packed[row_ptr[i] : row_ptr[i + 1]]) for i in range(len(row_ptr) - 1)]


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    den = np.linalg.norm(b)
    return np.linalg.norm(a - b) / (den if den > 0 else 1.0)


# ---- GPU helpers (only used by -m gpu tests) ---------------------------------------------------------
from matcouply_amd._engine import KIND  # noqa: E402  (short kind names -> enum mcl_penalty_kind)


def native_regs(descs, aux, dual, device):
    """descriptor dicts + numpy aux/dual (PARAFAC2 aux = (P, Delta)) -> list of NativeReg on `device`."""
    import torch
    from matcouply_amd._engine import NativeReg

    out = []
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=device)
    for d, z, u in zip(descs, aux, dual):
        kind = KIND[d["kind"]]
        if d["kind"] == "parafac2":
            out.append(NativeReg(kind, t(z[0]), t(u), aux2=t(z[1])))
        elif d["kind"] == "box":
            lo = -np.inf if d["min_val"] is None else d["min_val"]
            hi = np.inf if d["max_val"] is None else d["max_val"]
            out.append(NativeReg(kind, t(z), t(u), p0=lo, p1=hi))
        elif d["kind"] == "l1":
            out.append(NativeReg(kind, t(z), t(u), non_negativity=d.get("non_negativity", False), p0=d["reg_strength"]))
        elif d["kind"] == "l2ball":
            out.append(NativeReg(kind, t(z), t(u), non_negativity=d.get("non_negativity", False), p0=d["norm_bound"]))
        elif d["kind"] == "tv":
            out.append(NativeReg(kind, t(z), t(u), p0=d["reg_strength"], p1=d.get("l1_strength", 0.0)))
        elif d["kind"] == "unimodal":
            out.append(NativeReg(kind, t(z), t(u), non_negativity=d.get("non_negativity", False)))
        else:
            out.append(NativeReg(kind, t(z), t(u)))
    return out


def engine_from_oracle_state(st, device="cuda:0"):
    """Build a HipEngine holding fp32 copies of an OracleState's problem, factors and ADMM variables."""
    import torch
    from matcouply_amd._engine import HipEngine

    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=device)
    X, A, B, C = t(st.X), t(st.A), t(st.B), t(st.C)
    regs = [native_regs(st.regs[m], st.aux[m], st.dual[m], device) for m in range(3)]
    eng = HipEngine(X, st.row_ptr, st.A.shape[1], A, B, C, regs, l2_penalty=st.l2, inner_n_iter_max=st.inner,
                    feasibility_penalty_scale=st.scale, constant_A=st.constant_A, constant_B=st.constant_B)
    return eng


def to_np(t):
    return t.detach().cpu().numpy().astype(np.float64)
