"""CHECKER engine for CPU tests: the engine interface of matcouply_amd._engine.HipEngine implemented with the oracle's
NumPy arithmetic on CPU torch tensors.  Test infrastructure only - it lets `-m "not gpu"` tests drive the PRODUCT's
host logic (argument parsing, RNG order, stopping rules, multi-rank collectives over gloo) without a GPU.  The product
never imports this module; `matcouply_amd.decomposition._ENGINE_FACTORY` is None outside these tests."""
import numpy as np
import torch

from matcouply_amd import _engine
from oracle import aoadmm_oracle as orc

KIND_NAME = {1: "nn", 2: "box", 3: "l1", 4: "l2ball", 5: "unimodal", 6: "parafac2", 7: "external", 8: "tv", 9: "gl2", 10: "simplex"}


def _desc(reg):
    name = KIND_NAME[reg.kind]
    d = {"kind": name, "non_negativity": reg.non_negativity}
    if name == "box":
        d.update(min_val=reg.p0, max_val=reg.p1)
    elif name == "l1":
        d["reg_strength"] = reg.p0
    elif name == "l2ball":
        d["norm_bound"] = reg.p0
    elif name == "tv":
        d.update(reg_strength=reg.p0, l1_strength=reg.p1)
    elif name == "gl2":  # the descriptor carries [U | s | U^T] of the norm matrix M = U diag(s) U^T
        n = reg.matrix_rows
        m = reg.matrix.numpy()
        U, s = m[: n * n].reshape(n, n), m[n * n: n * n + n]
        d["norm_matrix"] = (U * s) @ U.T
    return d


class OracleEngine:
    def __init__(self, X, row_ptr, rank, A, B, C, regs, l2_penalty=(0, 0, 0), inner_n_iter_max=5,
                 feasibility_penalty_scale=1.0, constant_A=False, constant_B=False, exact_products=0, inner_tol=0.0):
        assert not inner_tol  # the checker runs the inner stopping test on the host path (decomposition.py keeps it there)
        self.exact_products = int(exact_products)  # the HIP engine's arithmetic hint under group= (the checker is fp64 anyway)
        self.A, self.B, self.C, self.regs = A, B, C, regs
        self.Xn = X.numpy()
        self.row_ptr = np.asarray(row_ptr, dtype=np.int64)
        self.I, self.r = len(self.row_ptr) - 1, rank
        self.K = X.shape[1]
        self.l2 = [float(v or 0.0) for v in l2_penalty]
        self.inner, self.scale = int(inner_n_iter_max), float(feasibility_penalty_scale)
        self.constant_A, self.constant_B = bool(constant_A), bool(constant_B)
        self.descs = [[_desc(r) for r in regs[m]] for m in range(3)]
        self.slab_of_row = np.repeat(np.arange(self.I), np.diff(self.row_ptr))
        self.norm_X_sq = float(np.sum(self.Xn.astype(np.float64) ** 2))
        self.by_products = None
        self.GR = torch.zeros(rank * rank + self.K * rank, dtype=torch.float64)
        self.rho_max_B = torch.zeros(1, dtype=torch.float64)
        self.rho_max_A = torch.zeros(1, dtype=torch.float64)
        self.pf2_red = torch.zeros(rank * rank + 1, dtype=torch.float64)

    # numpy views of the shared torch state
    def _n(self, t):
        return t.numpy()

    def _aux_matrix(self, mode, k):
        reg = self.regs[mode][k]
        if reg.kind == _engine.PEN_PARAFAC2:
            return self._n(reg.aux) @ self._n(reg.aux2)
        return self._n(reg.aux)

    # ---- B-phase in steps -----------------------------------------------------------------------------------
    def B_begin(self):
        A, C, X, rp = self._n(self.A), self._n(self.C), self.Xn, self.row_ptr
        CtC = C.T @ C
        self.rhs_B = (X @ C) * A[self.slab_of_row]
        self.L_B = CtC[None] * A[:, :, None] * A[:, None, :]
        self.rho_B = 0.5 * np.trace(self.L_B, axis1=1, axis2=2) * self.scale if self.I else np.zeros(0)
        self.rho_max_B[0] = float(self.rho_B.max()) if self.I else 0.0

    def B_rho_max(self):
        return self.rho_max_B

    def B_factor(self):
        n = len(self.regs[1])
        if self.constant_B:
            self.rho_B = np.full_like(self.rho_B, float(self.rho_max_B[0]))
        L = self.L_B + (self.rho_B * n + self.l2[1])[:, None, None] * np.eye(self.r)
        self.Linv_B = np.linalg.inv(L) if self.I else L

    def B_solve(self):
        T = self.rhs_B.copy()
        rho_row = self.rho_B[self.slab_of_row][:, None]
        for k, reg in enumerate(self.regs[1]):
            T += rho_row * (self._aux_matrix(1, k) - self._n(reg.dual))
        B = self._n(self.B)
        for i in range(self.I):
            s, e = self.row_ptr[i], self.row_ptr[i + 1]
            B[s:e] = T[s:e] @ self.Linv_B[i]
        self.by_products = None

    def B_prox_local(self, k):
        reg, d = self.regs[1][k], self.descs[1][k]
        B, U, rp = self._n(self.B), self._n(reg.dual), self.row_ptr
        Y = B + U
        rho_row = self.rho_B[self.slab_of_row][:, None]
        if d["kind"] in orc.ROW_SEPARABLE:
            Z = orc.prox_elementwise(d, Y, rho_row)
        elif d["kind"] == "parafac2":
            Delta = self._n(reg.aux2)
            P = self._n(reg.aux)
            acc = np.zeros((self.r, self.r))
            for i in range(self.I):
                s, e = rp[i], rp[i + 1]
                P[s:e] = orc.polar_factor(Y[s:e] @ Delta.T)
                acc += self.rho_B[i] * (P[s:e].T @ Y[s:e])
            self.pf2_red[: self.r * self.r] = torch.from_numpy(acc.reshape(-1))
            self.pf2_red[self.r * self.r] = float(np.sum(self.rho_B))
            return
        else:
            Z = np.empty_like(Y)
            for i in range(self.I):
                Z[rp[i]: rp[i + 1]] = orc.prox_matrix(d, Y[rp[i]: rp[i + 1]], self.rho_B[i])
        self._n(reg.aux)[...] = Z
        U[...] = B - (Z - U)

    def B_prox_reduce_buffer(self, k):
        return self.pf2_red if self.regs[1][k].kind == _engine.PEN_PARAFAC2 else None

    def B_prox_finish(self, k):
        reg = self.regs[1][k]
        if reg.kind != _engine.PEN_PARAFAC2:
            return
        r2 = self.r * self.r
        self._n(reg.aux2)[...] = (self.pf2_red[:r2] / self.pf2_red[r2]).numpy().reshape(self.r, self.r)
        B, U = self._n(self.B), self._n(reg.dual)
        U[...] = B - (self._aux_matrix(1, k) - U)

    def B_end(self):
        pass  # the checker engine defers nothing

    def update_B(self):
        self.B_begin()
        self.B_factor()
        n_it = self.inner if self.regs[1] else min(1, self.inner)
        for _ in range(n_it):
            self.B_solve()
            for k in range(len(self.regs[1])):
                self.B_prox_local(k)
                self.B_prox_finish(k)

    # ---- C-phase ----------------------------------------------------------------------------------------------
    def update_C_local(self):
        Ba = self._n(self.B) * self._n(self.A)[self.slab_of_row]
        G, R = Ba.T @ Ba, self.Xn.T @ Ba
        self.GR[...] = torch.from_numpy(np.concatenate([G.reshape(-1), R.reshape(-1)]))
        return self.GR

    def update_C_finish(self):
        r, n = self.r, len(self.regs[2])
        GR = self.GR.numpy()
        G, R = GR[: r * r].reshape(r, r), GR[r * r:].reshape(self.K, r)
        rho = 0.5 * np.trace(G) * self.scale
        Linv = np.linalg.inv(G + (rho * n + self.l2[2]) * np.eye(r))
        C = self._n(self.C)
        n_it = self.inner if n else min(1, self.inner)
        for _ in range(n_it):
            T = R.copy()
            for k, reg in enumerate(self.regs[2]):
                T += rho * (self._n(reg.aux) - self._n(reg.dual))
            C[...] = T @ Linv
            for k, reg in enumerate(self.regs[2]):
                Z = orc.prox_matrix(self.descs[2][k], C + self._n(reg.dual), rho)
                self._n(reg.dual)[...] = C - (Z - self._n(reg.dual))
                self._n(reg.aux)[...] = Z
        self.by_products = None

    # ---- A-phase ----------------------------------------------------------------------------------------------
    def A_begin(self):
        B, C, rp = self._n(self.B), self._n(self.C), self.row_ptr
        CtC = C.T @ C
        XC = self.Xn @ C
        self.rhs_A = np.add.reduceat(B * XC, rp[:-1], axis=0) if self.I else np.zeros((0, self.r))
        self.Q = np.stack([(B[rp[i]: rp[i + 1]].T @ B[rp[i]: rp[i + 1]]) * CtC for i in range(self.I)]) \
            if self.I else np.zeros((0, self.r, self.r))
        self.rho_A = 0.5 * np.trace(self.Q, axis1=1, axis2=2) * self.scale if self.I else np.zeros(0)
        self.rho_max_A[0] = float(self.rho_A.max()) if self.I else 0.0

    def A_rho_max(self):
        return self.rho_max_A

    def A_finish(self):
        n = len(self.regs[0])
        rho = np.full_like(self.rho_A, float(self.rho_max_A[0])) if self.constant_A else self.rho_A
        Linv = np.linalg.inv(self.Q + (rho * n + self.l2[0])[:, None, None] * np.eye(self.r)) if self.I else self.Q
        A = self._n(self.A)
        n_it = self.inner if n else min(1, self.inner)
        for _ in range(n_it):
            T = self.rhs_A.copy()
            for reg in self.regs[0]:
                T += rho[:, None] * (self._n(reg.aux) - self._n(reg.dual))
            A[...] = np.einsum("ik,ikj->ij", T, Linv)
            for k, reg in enumerate(self.regs[0]):
                Y = A + self._n(reg.dual)
                d = self.descs[0][k]
                Z = orc.prox_matrix(d, Y, rho[0]) if self.constant_A else orc.prox_elementwise(d, Y, rho[:, None])
                self._n(reg.dual)[...] = A - (Z - self._n(reg.dual))
                self._n(reg.aux)[...] = Z
        self.by_products = (self.rhs_A, self.Q)

    # step form of the A inner loop (host-evaluated / cross-rank prox in between), as HipEngine exposes it
    def A_factor(self):
        n = len(self.regs[0])
        self.rho_A_used = np.full_like(self.rho_A, float(self.rho_max_A[0])) if self.constant_A else self.rho_A
        self.Linv_A = np.linalg.inv(self.Q + (self.rho_A_used * n + self.l2[0])[:, None, None] * np.eye(self.r)) \
            if self.I else self.Q

    def A_solve(self):
        T = self.rhs_A.copy()
        for reg in self.regs[0]:
            T += self.rho_A_used[:, None] * (self._n(reg.aux) - self._n(reg.dual))
        self._n(self.A)[...] = np.einsum("ik,ikj->ij", T, self.Linv_A)

    def A_end(self):
        self.by_products = (self.rhs_A, self.Q)

    def rho(self, mode):
        if mode == 0:
            return torch.as_tensor(np.asarray(self.rho_A_used, dtype=np.float64))
        if mode == 1:
            return torch.as_tensor(np.asarray(self.rho_B, dtype=np.float64))
        return torch.as_tensor(np.asarray([self.rho_C], dtype=np.float64))

    # step form of the C inner loop (host-evaluated prox in between): mcl_C_begin / mcl_C_solve / mcl_C_end
    def C_begin(self):
        r, n = self.r, len(self.regs[2])
        GR = self.GR.numpy()
        G, self.R_C = GR[: r * r].reshape(r, r), GR[r * r:].reshape(self.K, r).copy()
        self.rho_C = 0.5 * np.trace(G) * self.scale
        self.Linv_C = np.linalg.inv(G + (self.rho_C * n + self.l2[2]) * np.eye(r))

    def C_solve(self):
        T = self.R_C.copy()
        for reg in self.regs[2]:
            T += self.rho_C * (self._n(reg.aux) - self._n(reg.dual))
        self._n(self.C)[...] = T @ self.Linv_C
        self.by_products = None

    def C_end(self):
        pass

    def update_A(self):
        self.A_begin()
        self.A_finish()

    # ---- diagnostics ------------------------------------------------------------------------------------------
    def diagnostics(self, include_replicated=True, out=None):
        if self.by_products is None:
            self.A_begin()
            self.by_products = (self.rhs_A, self.Q)
        rhs, Q = self.by_products
        A, B, C = self._n(self.A), self._n(self.B), self._n(self.C)
        d = np.zeros(_engine.DIAG_LEN)
        d[_engine.DIAG_NORM_SQ + 0], d[_engine.DIAG_NORM_SQ + 1] = np.sum(A * A), np.sum(B * B)
        d[_engine.DIAG_INNER] = np.sum(rhs * A)
        d[_engine.DIAG_MODEL_SQ] = np.einsum("ik,ikj,ij->", A, Q, A)
        d[_engine.DIAG_X_SQ] = self.norm_X_sq
        modes = [(0, A), (1, B)] + ([(2, C)] if include_replicated else [])
        if include_replicated:
            d[_engine.DIAG_NORM_SQ + 2] = np.sum(C * C)
        for m, F in modes:
            for k in range(len(self.regs[m])):
                base = _engine.DIAG_REG + (m * _engine.MCL_MAX_REGS + k) * 2
                d[base] = np.sum((self._aux_matrix(m, k) - F) ** 2)
                d[base + 1] = np.sum(np.abs(F))
        vec = torch.from_numpy(d)
        if out is not None:
            out[...] = vec
            return out
        return vec

    def iterate(self, n_iter, update_A=True, update_B=True, update_C=True, diag_ring=None):
        for it in range(n_iter):
            if update_B:
                self.update_B()
            if update_C:
                self.update_C_local()
                self.update_C_finish()
            if update_A:
                self.update_A()
            if diag_ring is not None:
                self.diagnostics(out=diag_ring[it])

    def run(self, n_iter_max, tol, absolute_tol, feasibility_tol, initial_loss, penalty_weight, evaluate_loss_always,
            update_A=True, update_B=True, update_C=True, max_run_ahead=0):
        """mcl_run of the C ABI (include/matcouply_hip.h) restated on the host: same verdict arithmetic, same rings"""
        tol, absolute_tol, feasibility_tol = float(tol or 0.0), float(absolute_tol or 0.0), float(feasibility_tol or 0.0)
        prev, rows, verdicts, code = float(initial_loss), [], [], 0
        for it in range(int(n_iter_max)):
            self.iterate(1, update_A=update_A, update_B=update_B, update_C=update_C)
            d = self.diagnostics().numpy().copy()
            worst = -np.inf
            for m in range(3):
                fn = np.sqrt(d[_engine.DIAG_NORM_SQ + m])
                for k in range(len(self.regs[m])):
                    with np.errstate(invalid="ignore", divide="ignore"):
                        gap = np.sqrt(d[_engine.DIAG_REG + (m * _engine.MCL_MAX_REGS + k) * 2]) / fn
                    worst = gap if (gap > worst or gap != gap) else worst
            feasible = bool(feasibility_tol != 0.0 and worst < feasibility_tol)
            rec = loss = 0.0
            computed = feasible or bool(evaluate_loss_always)
            if computed:
                xsq, inner, model = d[_engine.DIAG_X_SQ], d[_engine.DIAG_INNER], d[_engine.DIAG_MODEL_SQ]
                rec = np.sqrt(max(0.0, xsq - 2.0 * inner + model)) / np.sqrt(xsq)
                reg = 0.0
                for m in range(3):
                    for k in range(len(self.regs[m])):
                        w = penalty_weight[m][k] if k < len(penalty_weight[m]) else 0.0
                        if w:
                            reg += w * d[_engine.DIAG_REG + (m * _engine.MCL_MAX_REGS + k) * 2 + 1]
                    if self.l2[m]:
                        reg += 0.5 * self.l2[m] * d[_engine.DIAG_NORM_SQ + m]
                loss = 0.5 * (rec * rec) + reg
                if tol != 0.0:
                    with np.errstate(invalid="ignore"):
                        rel = abs(prev - loss) < tol * prev
                    if feasible and rel:
                        code = _engine.STOP_RELATIVE
                    elif feasible and loss < absolute_tol:
                        code = _engine.STOP_ABSOLUTE
                prev = loss
            rows.append(d)
            verdicts.append([rec, loss, worst, float(int(feasible) | (int(computed) << 1) | (code << 2))])
            if code:
                break
        n = len(rows)
        return (n, code, np.asarray(rows).reshape(n, _engine.DIAG_LEN), np.asarray(verdicts, dtype=np.float64).reshape(n, 4))

    # ---- mcl_gate_begin / mcl_verdict / mcl_gate_end restated: after a hit every later state-writing call is a no-op ---------
    def gate_begin(self, tol, absolute_tol, feasibility_tol, initial_loss, penalty_weight, evaluate_loss_always):
        self._gate = dict(tol=float(tol or 0.0), abs=float(absolute_tol or 0.0), feas=float(feasibility_tol or 0.0),
                          prev=float(initial_loss), w=penalty_weight, always=bool(evaluate_loss_always), stopped=False,
                          stop_it=-1, code=0)
        if not getattr(type(self), "_gated", False):
            # wrap every state-writing method once: with the gate closed they do nothing (the kernels' MCL_GATE)
            def gated(fn):
                def inner(self, *a, **kw):
                    g = getattr(self, "_gate", None)
                    if g is not None and g["stopped"]:
                        return self.GR if fn.__name__ == "update_C_local" else None
                    return fn(self, *a, **kw)
                inner.__name__ = fn.__name__
                return inner
            for name in ("B_begin", "B_factor", "B_solve", "B_prox_local", "B_prox_finish", "B_end", "update_B", "update_C_local",
                         "update_C_finish", "A_begin", "A_finish", "A_factor", "A_solve", "A_end", "update_A", "C_begin", "C_solve", "C_end"):
                setattr(type(self), name, gated(getattr(type(self), name)))
            type(self)._gated = True

    def verdict(self, vec, iteration, row):
        g = self._gate
        if g["stopped"]:
            return
        d = vec.numpy()
        worst = -np.inf
        for m in range(3):
            fn = np.sqrt(d[_engine.DIAG_NORM_SQ + m])
            for k in range(len(self.regs[m])):
                with np.errstate(invalid="ignore", divide="ignore"):
                    gap = np.sqrt(d[_engine.DIAG_REG + (m * _engine.MCL_MAX_REGS + k) * 2]) / fn
                worst = gap if (gap > worst or gap != gap) else worst
        feasible = bool(g["feas"] != 0.0 and worst < g["feas"])
        rec = loss = 0.0
        code = 0
        computed = feasible or g["always"]
        if computed:
            xsq, inner, model = d[_engine.DIAG_X_SQ], d[_engine.DIAG_INNER], d[_engine.DIAG_MODEL_SQ]
            rec = np.sqrt(max(0.0, xsq - 2.0 * inner + model)) / np.sqrt(xsq)
            reg = 0.0
            for m in range(3):
                for k in range(len(self.regs[m])):
                    w = g["w"][m][k] if k < len(g["w"][m]) else 0.0
                    if w:
                        reg += w * d[_engine.DIAG_REG + (m * _engine.MCL_MAX_REGS + k) * 2 + 1]
                if self.l2[m]:
                    reg += 0.5 * self.l2[m] * d[_engine.DIAG_NORM_SQ + m]
            loss = 0.5 * (rec * rec) + reg
            if g["tol"] != 0.0:
                with np.errstate(invalid="ignore"):
                    rel = abs(g["prev"] - loss) < g["tol"] * g["prev"]
                if feasible and rel:
                    code = _engine.STOP_RELATIVE
                elif feasible and loss < g["abs"]:
                    code = _engine.STOP_ABSOLUTE
            g["prev"] = loss
        row[...] = torch.tensor([rec, loss, worst, float(int(feasible) | (int(computed) << 1) | (code << 2))], dtype=torch.float64)
        if code:
            g.update(stopped=True, stop_it=int(iteration), code=code)

    def gate_status(self):
        g = self._gate
        return g["stopped"], g["stop_it"], g["code"]

    def gate_end(self, stopped_early):
        self._gate = None

    def penalty_value(self, mode, k):
        F = self._n((self.A, self.B, self.C)[mode])
        M = self.descs[mode][k]["norm_matrix"]
        mats = [F[self.row_ptr[i]: self.row_ptr[i + 1]] for i in range(self.I)] if mode == 1 else [F]
        return torch.tensor([sum(float(np.trace(x.T @ M @ x)) for x in mats)], dtype=torch.float64)

    def diagnostics_deferred(self, include_replicated=True, out=None):
        return self.diagnostics(include_replicated=include_replicated, out=out)  # the checker has nothing to defer

    def flush_diagnostics(self):
        pass

    def close(self):
        pass


class OracleEngineFactory:
    """Install with `matcouply_amd.decomposition._ENGINE_FACTORY = OracleEngineFactory()` (tests only)."""
    device = torch.device("cpu")
    dtype = torch.float64

    def pack(self, matrices):
        mats = [np.asarray(m, dtype=np.float64) for m in matrices]
        row_ptr = np.concatenate([[0], np.cumsum([m.shape[0] for m in mats])]).astype(np.int64)
        X = torch.from_numpy(np.concatenate(mats, 0) if mats else np.zeros((0, 0)))
        return X, row_ptr

    def __call__(self, **kw):
        return OracleEngine(**kw)
