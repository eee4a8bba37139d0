"""ctypes binding of libmatcouply_hip.so (the C ABI in include/matcouply_hip.h).

PyTorch is used for plumbing only: device memory (`torch.empty(..., device="cuda")`), the current HIP stream
and `torch.distributed`.  All arithmetic of the AO-ADMM hot path runs in the HIP library; there is NO CPU or
eager-PyTorch fallback - if the library is missing or no MI355X is visible, construction raises.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmatcouply_hip.so")

MCL_ABI_VERSION = 410  # include/matcouply_hip.h; checked against mcl_version() when the library is loaded
MCL_MAX_REGS = 4
MCL_MAX_RANK = 64
DIAG_NORM_SQ, DIAG_INNER, DIAG_MODEL_SQ, DIAG_X_SQ, DIAG_REG = 0, 3, 4, 5, 8
DIAG_LEN = 8 + 3 * MCL_MAX_REGS * 2

PEN_NN, PEN_BOX, PEN_L1, PEN_L2BALL, PEN_UNIMODAL, PEN_PARAFAC2, PEN_EXTERNAL, PEN_TV, PEN_GL2, PEN_SIMPLEX = 1, 2, 3, 4, 5, 6, 7, 8, 9, 10
# enum mcl_buffer_id / enum mcl_profile_slot / MCL_VARIANT_EXACT_MODE of include/matcouply_hip.h
(BUF_RHSES, BUF_CROSS_PRODUCTS, BUF_XC, BUF_RHO_B, BUF_RHO_A, BUF_RHO_C, BUF_CTC, BUF_LINV_B, BUF_PF2_STATUS, BUF_PF2_ACC,
 BUF_PF2_GRAM, BUF_SWEEP_CYCLES, BUF_SEG_ROW0, BUF_SEG_NROWS, BUF_WAVE_SEG_PTR, BUF_BSEG_ROW0, BUF_BSEG_NROWS,
 BUF_WAVE_BSEG_PTR, BUF_NS_STAMPS, BUF_BSEG_PART) = range(20)
(PROF_XC, PROF_XT, PROF_ROWS_FUSED, PROF_SWEEP, PROF_REDUCE, PROF_C_FINISH, PROF_A_FINISH, PROF_ROWS_CHAIN, PROF_UNIMODAL,
 PROF_PF2, PROF_DIAG, PROF_OTHER, PROF_SLOTS) = range(13)
PROF_ROLE = {PROF_XC: "X C pass", PROF_XT: "X^T (B o a) pass", PROF_ROWS_FUSED: "fused B-phase rows",
             PROF_SWEEP: "one-pass sweep (X C -> B-phase -> X^T B)", PROF_REDUCE: "[G | R] reduction", PROF_C_FINISH: "C-phase finish",
             PROF_A_FINISH: "A-phase finish", PROF_ROWS_CHAIN: "chained B row pass", PROF_UNIMODAL: "unimodal regressions",
             PROF_PF2: "PARAFAC2 per-slab algebra", PROF_DIAG: "diagnostics reduction", PROF_OTHER: "other launches"}
VARIANT_EXACT_MODE = 100
# short names of the native kinds (descriptor dicts of bench.py / the test helpers -> enum mcl_penalty_kind)
KIND = {"nn": PEN_NN, "box": PEN_BOX, "l1": PEN_L1, "l2ball": PEN_L2BALL, "unimodal": PEN_UNIMODAL,
        "parafac2": PEN_PARAFAC2, "tv": PEN_TV, "gl2": PEN_GL2, "simplex": PEN_SIMPLEX}

# every symbol include/matcouply_hip.h declares (checked by tests/test_cabi_symbols.py)
EXPORTED_SYMBOLS = [
    "mcl_create", "mcl_destroy", "mcl_last_error", "mcl_version", "mcl_set_problem", "mcl_set_options",
    "mcl_set_factors", "mcl_set_penalties", "mcl_workspace_bytes", "mcl_set_workspace", "mcl_update_B",
    "mcl_update_C_local", "mcl_c_normal_equations", "mcl_update_C_finish", "mcl_update_A", "mcl_diagnostics",
    "mcl_diagnostics_deferred", "mcl_flush_diagnostics", "mcl_penalty_value", "mcl_condition_probe", "mcl_condition_monitor",
    "mcl_iterate", "mcl_run", "mcl_gate_begin", "mcl_verdict", "mcl_gate_end", "mcl_B_begin", "mcl_B_rho_max", "mcl_B_factor", "mcl_B_solve", "mcl_B_prox_local",
    "mcl_B_prox_reduce_buffer", "mcl_B_prox_finish", "mcl_B_end", "mcl_A_begin", "mcl_A_rho_max", "mcl_A_finish",
    "mcl_A_factor", "mcl_A_solve", "mcl_A_end", "mcl_C_begin", "mcl_C_solve", "mcl_C_end",
    "mcl_internal_buffer", "mcl_kernel_variant", "mcl_profile_enable", "mcl_profile_set_stride", "mcl_profile_read", "mcl_profile_launches", "mcl_profile_overhead_us",
    "mcl_reload_switches", "mcl_active_switches", "mcl_record_event", "mcl_wait_event", "mcl_cmf_to_packed",
    "mcl_svd_init_workspace_bytes", "mcl_svd_init", "mcl_svd_init_last_error", "mcl_read_bandwidth",
]


class PenaltyDesc(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int32), ("non_negativity", ctypes.c_int32), ("p0", ctypes.c_double),
                ("p1", ctypes.c_double), ("aux", ctypes.c_void_p), ("dual", ctypes.c_void_p),
                ("aux2", ctypes.c_void_p), ("matrix", ctypes.c_void_p), ("matrix_rows", ctypes.c_int64)]


class Options(ctypes.Structure):
    _fields_ = [("feasibility_penalty_scale", ctypes.c_double), ("l2_penalty", ctypes.c_double * 3),
                ("inner_tol", ctypes.c_double), ("inner_n_iter_max", ctypes.c_int32), ("constant_A", ctypes.c_int32), ("constant_B", ctypes.c_int32),
                ("exact_products", ctypes.c_int32)]


class StopRule(ctypes.Structure):
    """mcl_stop_rule of include/matcouply_hip.h (a tolerance of 0 = not set, like None / 0 in the reference)"""
    _fields_ = [("tol", ctypes.c_double), ("absolute_tol", ctypes.c_double), ("feasibility_tol", ctypes.c_double),
                ("initial_loss", ctypes.c_double), ("penalty_weight", (ctypes.c_double * MCL_MAX_REGS) * 3),
                ("evaluate_loss_always", ctypes.c_int32), ("max_run_ahead", ctypes.c_int32)]


STOP_RELATIVE, STOP_ABSOLUTE = 1, 2
VERDICT_FEASIBLE, VERDICT_LOSS_EVALUATED = 1, 2  # flag bits of a verdict row; the stop code sits above them (>> 2)

_lib = None


def load_library():
    """Load libmatcouply_hip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP engine has not been built. Run `python -c 'import __graft_entry__ as g; "
            "g.build()'` (or `python matcouply_amd/_build.py`). matcouply_amd has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    P, I32, I64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
    sig = {
        "mcl_create": (ctypes.c_int, [ctypes.POINTER(P), ctypes.c_int, P]),
        "mcl_destroy": (None, [P]),
        "mcl_last_error": (ctypes.c_char_p, [P]),
        "mcl_version": (ctypes.c_int, []),
        "mcl_set_problem": (ctypes.c_int, [P, P, ctypes.POINTER(I64), I64, I64, I32]),
        "mcl_set_options": (ctypes.c_int, [P, ctypes.POINTER(Options)]),
        "mcl_set_factors": (ctypes.c_int, [P, P, P, P]),
        "mcl_set_penalties": (ctypes.c_int, [P, I32, I32, ctypes.POINTER(PenaltyDesc)]),
        "mcl_workspace_bytes": (I64, [P]),
        "mcl_set_workspace": (ctypes.c_int, [P, P, I64]),
        "mcl_update_B": (ctypes.c_int, [P]),
        "mcl_update_C_local": (ctypes.c_int, [P]),
        "mcl_c_normal_equations": (P, [P, ctypes.POINTER(I64)]),
        "mcl_update_C_finish": (ctypes.c_int, [P]),
        "mcl_update_A": (ctypes.c_int, [P]),
        "mcl_diagnostics": (ctypes.c_int, [P, P, I32]),
        "mcl_diagnostics_deferred": (ctypes.c_int, [P, P, I32]),
        "mcl_flush_diagnostics": (ctypes.c_int, [P]),
        "mcl_penalty_value": (ctypes.c_int, [P, I32, I32, P]),
        "mcl_condition_probe": (ctypes.c_int, [P, I32, P]),
        "mcl_condition_monitor": (ctypes.c_int, [P, P, I32]),
        "mcl_iterate": (ctypes.c_int, [P, I32, I32, I32, I32, P]),
        "mcl_run": (ctypes.c_int, [P, I32, I32, I32, I32, ctypes.POINTER(StopRule), P, P, P]),
        "mcl_gate_begin": (ctypes.c_int, [P, ctypes.POINTER(StopRule), P]),
        "mcl_verdict": (ctypes.c_int, [P, P, I32, P]),
        "mcl_gate_end": (ctypes.c_int, [P, I32]),
        "mcl_B_begin": (ctypes.c_int, [P]),
        "mcl_B_rho_max": (P, [P]),
        "mcl_B_factor": (ctypes.c_int, [P]),
        "mcl_B_solve": (ctypes.c_int, [P]),
        "mcl_B_prox_local": (ctypes.c_int, [P, I32]),
        "mcl_B_prox_reduce_buffer": (P, [P, I32, ctypes.POINTER(I64)]),
        "mcl_B_prox_finish": (ctypes.c_int, [P, I32]),
        "mcl_B_end": (ctypes.c_int, [P]),
        "mcl_A_begin": (ctypes.c_int, [P]),
        "mcl_A_rho_max": (P, [P]),
        "mcl_A_finish": (ctypes.c_int, [P]),
        "mcl_A_factor": (ctypes.c_int, [P]),
        "mcl_A_solve": (ctypes.c_int, [P]),
        "mcl_A_end": (ctypes.c_int, [P]),
        "mcl_C_begin": (ctypes.c_int, [P]),
        "mcl_C_solve": (ctypes.c_int, [P]),
        "mcl_C_end": (ctypes.c_int, [P]),
        "mcl_internal_buffer": (P, [P, I32, ctypes.POINTER(I64)]),
        "mcl_kernel_variant": (ctypes.c_char_p, [P, I32]),
        "mcl_profile_enable": (ctypes.c_int, [P, I32]),
        "mcl_profile_set_stride": (ctypes.c_int, [P, I32]),
        "mcl_profile_read": (ctypes.c_int, [P, I32, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(I32)]),
        "mcl_profile_launches": (I64, [P, I32]),
        "mcl_profile_overhead_us": (ctypes.c_double, [P]),
        "mcl_reload_switches": (ctypes.c_int, [P]),
        "mcl_active_switches": (ctypes.c_char_p, [P]),
        "mcl_record_event": (ctypes.c_int, [P, P]),
        "mcl_wait_event": (ctypes.c_int, [P, P]),
        "mcl_cmf_to_packed": (ctypes.c_int, [P, P, P, P, P, I64, I64, I32, P, P]),
        "mcl_svd_init_workspace_bytes": (I64, [ctypes.POINTER(I64), I64, I64, I32]),
        "mcl_svd_init": (ctypes.c_int, [P, ctypes.POINTER(I64), I64, I64, I32, I32, P, P, P, I64, P, P]),
        "mcl_svd_init_last_error": (ctypes.c_char_p, []),
        "mcl_read_bandwidth": (ctypes.c_int, [P, I64, I32, P, P, ctypes.POINTER(ctypes.c_double)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.mcl_version() != MCL_ABI_VERSION:
        raise RuntimeError(f"{LIB_PATH} reports ABI version {lib.mcl_version()}, this binding was written against "
                           f"{MCL_ABI_VERSION} (include/matcouply_hip.h): rebuild the library (`python matcouply_amd/_build.py --force`)")
    _lib = lib
    return lib


class EngineError(RuntimeError):
    pass


def cmf_to_packed(A, B, C, row_ptr, weights=None):
    """Dense reconstruction on the device: packed [sum J_i, K] float32 tensor with rows row_ptr[i] .. row_ptr[i+1] equal to
    (B_i diag(weights o a_i)) C^T.  A [I, r], B packed [N, r], C [K, r]: contiguous float32 CUDA tensors."""
    import torch

    lib = load_library()
    for name, t in (("A", A), ("B", B), ("C", C)) + ((("weights", weights),) if weights is not None else ()):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise EngineError(f"{name} must be a contiguous float32 CUDA tensor")
    row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int64)
    N, K, r = int(B.shape[0]), int(C.shape[0]), int(C.shape[1])
    if int(row_ptr[-1]) != N or A.shape != (len(row_ptr) - 1, r) or B.shape[1] != r:
        raise EngineError("factor shapes do not match")
    counts = torch.as_tensor(np.diff(row_ptr), device=B.device)
    slab = torch.repeat_interleave(torch.arange(len(row_ptr) - 1, device=B.device, dtype=torch.int32), counts)
    out = torch.empty((N, K), dtype=torch.float32, device=B.device)
    with torch.cuda.device(B.device):
        stream = torch.cuda.current_stream(B.device).cuda_stream
        rc = lib.mcl_cmf_to_packed(A.data_ptr(), B.data_ptr(), C.data_ptr(), weights.data_ptr() if weights is not None else None,
                                   slab.data_ptr(), N, K, r, out.data_ptr(), ctypes.c_void_p(stream))
    if rc != 0:
        raise EngineError("mcl_cmf_to_packed failed")
    return out


def read_bandwidth(buf, repeats=10):
    """GB/s a pure streaming read of the CUDA tensor `buf` reaches on this box (mcl_read_bandwidth; synchronises)"""
    import torch

    lib = load_library()
    scratch = torch.zeros(1, dtype=torch.float32, device=buf.device)
    out = ctypes.c_double()
    with torch.cuda.device(buf.device):
        stream = torch.cuda.current_stream(buf.device).cuda_stream
        rc = lib.mcl_read_bandwidth(buf.data_ptr(), buf.numel() * buf.element_size(), int(repeats), scratch.data_ptr(),
                                    ctypes.c_void_p(stream), ctypes.byref(out))
    if rc != 0:
        raise EngineError("mcl_read_bandwidth failed")
    return out.value


def svd_init(X, row_ptr, rank, threshold=False):
    """init="svd" / "threshold_svd" on the device (mcl_svd_init): X packed [sum J_i, K] float32 CUDA tensor -> (B packed
    [sum J_i, rank], C [K, rank], info int32 [I + 1]: subspace iterations per matrix, the stack last).  Singular vectors with
    the entry of largest magnitude positive (LAPACK's vectors up to sign)."""
    import torch

    lib = load_library()
    if not (X.is_cuda and X.dtype == torch.float32 and X.is_contiguous()):
        raise EngineError("X must be a contiguous float32 CUDA tensor")
    row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int64)
    I, K, N = len(row_ptr) - 1, int(X.shape[1]), int(X.shape[0])
    rp = row_ptr.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))
    nbytes = lib.mcl_svd_init_workspace_bytes(rp, I, K, int(rank))
    if nbytes < 0:
        raise EngineError("mcl_svd_init_workspace_bytes: bad arguments")
    ws = torch.empty(int(nbytes) + 256, dtype=torch.uint8, device=X.device)
    off = (-ws.data_ptr()) % 256
    B = torch.empty((N, int(rank)), dtype=torch.float32, device=X.device)
    C = torch.empty((K, int(rank)), dtype=torch.float32, device=X.device)
    info = torch.zeros(I + 1, dtype=torch.int32, device=X.device)
    with torch.cuda.device(X.device):
        stream = torch.cuda.current_stream(X.device).cuda_stream
        rc = lib.mcl_svd_init(X.data_ptr(), rp, I, K, int(rank), int(bool(threshold)), B.data_ptr(), C.data_ptr(), ws.data_ptr() + off,
                              nbytes, info.data_ptr(), ctypes.c_void_p(stream))
    if rc != 0:
        raise EngineError(lib.mcl_svd_init_last_error().decode())
    return B, C, info


class NativeReg:
    """One penalty as the engine sees it: kind + parameters + device tensors of its ADMM variables."""

    def __init__(self, kind, aux, dual, aux2=None, non_negativity=False, p0=0.0, p1=0.0, matrix=None, matrix_rows=0):
        self.kind, self.aux, self.dual, self.aux2 = kind, aux, dual, aux2
        self.non_negativity, self.p0, self.p1 = bool(non_negativity), float(p0), float(p1)
        # GeneralizedL2: fp64 tensor [n * n + n + n * n] = U (eigenvectors of the norm matrix in its columns), s, U^T
        self.matrix, self.matrix_rows = matrix, int(matrix_rows)


class HipEngine:
    """Owns one `mcl_context` bound to torch's current HIP stream on `device`.

    X: float32 CUDA tensor [sum J_i, K] (packed slabs), row_ptr: int64 array [I+1],
    A/B/C: float32 CUDA tensors (updated in place), regs: [[NativeReg]*n0, [..]*n1, [..]*n2].
    """

    def __init__(self, X, row_ptr, rank, A, B, C, regs, l2_penalty=(0.0, 0.0, 0.0), inner_n_iter_max=5,
                 feasibility_penalty_scale=1.0, constant_A=False, constant_B=False, exact_products=0, inner_tol=0.0):
        import torch

        self._torch = torch
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise EngineError("no HIP device visible: matcouply_amd needs an MI355X (gfx950); there is no CPU fallback")
        for name, t in (("X", X), ("A", A), ("B", B), ("C", C)):
            if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
                raise EngineError(f"{name} must be a contiguous float32 CUDA tensor")
        self.device = X.device
        self.X, self.A, self.B, self.C = X, A, B, C
        self.row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int64)
        self.I, self.K, self.N, self.r = len(self.row_ptr) - 1, int(X.shape[1]), int(X.shape[0]), int(rank)
        if A.shape != (self.I, self.r) or B.shape != (self.N, self.r) or C.shape != (self.K, self.r):
            raise EngineError("factor shapes do not match the problem")
        self.regs = regs
        self._h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            rc = self.lib.mcl_create(ctypes.byref(self._h), self.device.index or 0, ctypes.c_void_p(stream))
        if rc != 0:
            raise EngineError(self.lib.mcl_last_error(None).decode())
        self._check(self.lib.mcl_set_problem(self._h, X.data_ptr(), self.row_ptr.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)),
                                             self.I, self.K, self.r))
        opt = Options()
        opt.feasibility_penalty_scale = float(feasibility_penalty_scale)
        for m in range(3):
            opt.l2_penalty[m] = float(l2_penalty[m] or 0.0)
        opt.inner_n_iter_max = int(inner_n_iter_max)
        opt.constant_A, opt.constant_B = int(bool(constant_A)), int(bool(constant_B))
        opt.inner_tol = float(inner_tol or 0.0)  # > 0: the inner stopping test of decomposition.py:90-117, evaluated on the device
        opt.exact_products = int(exact_products)  # 0: by this context's size; 1 / 2: forced on / off (sharded hosts: by the WHOLE problem)
        self._opt = opt
        self._check(self.lib.mcl_set_options(self._h, ctypes.byref(opt)))
        self._check(self.lib.mcl_set_factors(self._h, A.data_ptr(), B.data_ptr(), C.data_ptr()))
        for mode in range(3):
            n = len(regs[mode])
            arr = (PenaltyDesc * max(n, 1))()
            for k, reg in enumerate(regs[mode]):
                for t in (reg.aux, reg.dual) + ((reg.aux2,) if reg.aux2 is not None else ()):
                    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
                        raise EngineError("aux/dual variables must be contiguous float32 CUDA tensors")
                arr[k].kind, arr[k].non_negativity = reg.kind, int(reg.non_negativity)
                arr[k].p0, arr[k].p1 = reg.p0, reg.p1
                arr[k].aux, arr[k].dual = reg.aux.data_ptr(), reg.dual.data_ptr()
                arr[k].aux2 = reg.aux2.data_ptr() if reg.aux2 is not None else None
                if reg.kind == PEN_GL2:
                    nm = reg.matrix_rows
                    if not (reg.matrix is not None and reg.matrix.is_cuda and reg.matrix.dtype == torch.float64
                            and reg.matrix.is_contiguous() and reg.matrix.numel() == 2 * nm * nm + nm):
                        raise EngineError("GeneralizedL2: `matrix` must be a contiguous float64 CUDA tensor [U | s | U^T]")
                    arr[k].matrix, arr[k].matrix_rows = reg.matrix.data_ptr(), nm
                else:
                    arr[k].matrix, arr[k].matrix_rows = None, 0
            self._check(self.lib.mcl_set_penalties(self._h, mode, n, arr))
        nbytes = self.lib.mcl_workspace_bytes(self._h)
        if nbytes < 0:
            raise EngineError("mcl_workspace_bytes failed")
        active = self.lib.mcl_active_switches(self._h).decode()
        if active:
            import warnings

            warnings.warn(f"libmatcouply_hip: MCL_* switches in the environment ({active}) select non-default kernel forms; the "
                          "parity statements of DESIGN.md hold for a clean environment", RuntimeWarning, stacklevel=3)
        self.workspace = torch.empty(int(nbytes) + 256, dtype=torch.uint8, device=self.device)
        self._ws_off = (-self.workspace.data_ptr()) % 256
        self._check(self.lib.mcl_set_workspace(self._h, self.workspace.data_ptr() + self._ws_off, nbytes))

    # -- plumbing -----------------------------------------------------------------------------------
    def _check(self, rc):
        if rc != 0:
            raise EngineError(self.lib.mcl_last_error(self._h).decode())

    def _view(self, ptr, count, dtype):
        """torch view of `count` elements at device address `ptr` inside the workspace."""
        torch = self._torch
        off = ptr - self.workspace.data_ptr()
        nb = count * torch.empty(0, dtype=dtype).element_size()
        assert 0 <= off and off + nb <= self.workspace.numel()
        return self.workspace[off : off + nb].view(dtype)

    def close(self):
        if self._h:
            self.lib.mcl_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- phases ---------------------------------------------------------------------------------------
    def update_B(self):
        self._check(self.lib.mcl_update_B(self._h))

    def update_C_local(self):
        """Returns the fp64 device tensor [G | R] (r*r + K*r) to be all-reduced over ranks."""
        self._check(self.lib.mcl_update_C_local(self._h))
        return self.c_normal_equations()

    def c_normal_equations(self):
        n = ctypes.c_int64()
        p = self.lib.mcl_c_normal_equations(self._h, ctypes.byref(n))
        return self._view(p, n.value, self._torch.float64)

    def update_C_finish(self):
        self._check(self.lib.mcl_update_C_finish(self._h))

    def update_A(self):
        self._check(self.lib.mcl_update_A(self._h))

    def diagnostics(self, include_replicated=True, out=None):
        torch = self._torch
        if out is None:
            out = torch.empty(DIAG_LEN, dtype=torch.float64, device=self.device)
        self._check(self.lib.mcl_diagnostics(self._h, out.data_ptr(), int(include_replicated)))
        return out

    def diagnostics_deferred(self, include_replicated=True, out=None):
        """diagnostics() whose reduction may ride on the next C-phase reduction kernel; `out` is complete once any other
        call of this engine (or flush_diagnostics()) has been made"""
        torch = self._torch
        if out is None:
            out = torch.empty(DIAG_LEN, dtype=torch.float64, device=self.device)
        self._check(self.lib.mcl_diagnostics_deferred(self._h, out.data_ptr(), int(include_replicated)))
        return out

    def flush_diagnostics(self):
        self._check(self.lib.mcl_flush_diagnostics(self._h))

    def penalty_value(self, mode, k):
        """value of penalty k of `mode` on the current factor (kinds whose value is not in the diagnostics vector:
        GeneralizedL2 - trace(F^T M F) summed over the mode's matrices); a 1-element float64 device tensor"""
        out = self._torch.empty(1, dtype=self._torch.float64, device=self.device)
        self._check(self.lib.mcl_penalty_value(self._h, int(mode), int(k), out.data_ptr()))
        return out

    def condition_probe(self, update_A=True, update_B=True, update_C=True):
        """mcl_condition_probe: kappa = ||M||_F ||M^-1||_F of the normal equations every PENALTY-FREE mode among the updated
        ones would solve from the current factors (modes 0 / 1: the worst matrix; 0.0 for modes with penalties); a float64
        device tensor of 3 - reading it synchronises"""
        out = self._torch.zeros(3, dtype=self._torch.float64, device=self.device)
        mask = int(bool(update_A)) | int(bool(update_B)) << 1 | int(bool(update_C)) << 2
        self._check(self.lib.mcl_condition_probe(self._h, mask, out.data_ptr()))
        return out

    def condition_monitor(self, on, update_A=True, update_B=True, update_C=True):
        """mcl_condition_monitor: while on, every phase of a penalty-free mode among the updated ones first measures the kappa of
        the system it is about to solve, and every PARAFAC2 inner iteration the conditioning of its polar factors; returns the
        float64 device tensor of 4 that collects the running maxima (modes 0, 1, 2, polar factors) (on) / None"""
        if not on:
            self._check(self.lib.mcl_condition_monitor(self._h, None, 0))
            self._monitor = None
            return None
        self._monitor = self._torch.zeros(4, dtype=self._torch.float64, device=self.device)
        mask = int(bool(update_A)) | int(bool(update_B)) << 1 | int(bool(update_C)) << 2
        self._check(self.lib.mcl_condition_monitor(self._h, self._monitor.data_ptr(), mask))
        return self._monitor

    def invalidate(self):
        """the caller has written the factor / ADMM tensors behind the engine's back: every cached by-product is forgotten
        (mcl_set_factors with the same pointers)"""
        self._check(self.lib.mcl_set_factors(self._h, self.A.data_ptr(), self.B.data_ptr(), self.C.data_ptr()))

    def set_exact(self, exact=True):
        """Switch the context's arithmetic (mcl_options.exact_products = 1 / 2) between two outer iterations: the workspace is
        planned and installed anew (a larger one is allocated when the mode needs it); factors and ADMM variables live in the
        caller's tensors and are untouched, every cached by-product is recomputed by the next phase."""
        want = 1 if exact else 2
        if int(self._opt.exact_products) == want:
            return
        self._opt.exact_products = want
        self._check(self.lib.mcl_set_options(self._h, ctypes.byref(self._opt)))
        nbytes = self.lib.mcl_workspace_bytes(self._h)
        if nbytes < 0:
            raise EngineError("mcl_workspace_bytes failed")
        if int(nbytes) + 256 > self.workspace.numel():
            self._torch.cuda.synchronize(self.device)  # kernels of the old installation may still be reading the old buffer
            self.workspace = self._torch.empty(int(nbytes) + 256, dtype=self._torch.uint8, device=self.device)
            self._ws_off = (-self.workspace.data_ptr()) % 256
        self._check(self.lib.mcl_set_workspace(self._h, self.workspace.data_ptr() + self._ws_off, nbytes))

    def iterate(self, n_iter, update_A=True, update_B=True, update_C=True, diag_ring=None):
        ptr = diag_ring.data_ptr() if diag_ring is not None else None
        self._check(self.lib.mcl_iterate(self._h, int(n_iter), int(update_A), int(update_B), int(update_C), ptr))

    def _stop_rule(self, tol, absolute_tol, feasibility_tol, initial_loss, penalty_weight, evaluate_loss_always, max_run_ahead=0):
        rule = StopRule()
        rule.tol, rule.absolute_tol = float(tol or 0.0), float(absolute_tol or 0.0)
        rule.feasibility_tol, rule.initial_loss = float(feasibility_tol or 0.0), float(initial_loss)
        for m in range(3):
            for k in range(MCL_MAX_REGS):
                rule.penalty_weight[m][k] = float(penalty_weight[m][k]) if k < len(penalty_weight[m]) else 0.0
        rule.evaluate_loss_always, rule.max_run_ahead = int(bool(evaluate_loss_always)), int(max_run_ahead)
        return rule

    def _pinned_status(self):
        if getattr(self, "_status", None) is None:
            self._status = self._torch.zeros(4, dtype=self._torch.int32).pin_memory()  # written by the verdict kernel
        return self._status

    # the pieces of run() for a host that drives the iterations itself (sharded loop: reductions between the calls)
    def gate_begin(self, tol, absolute_tol, feasibility_tol, initial_loss, penalty_weight, evaluate_loss_always):
        rule = self._stop_rule(tol, absolute_tol, feasibility_tol, initial_loss, penalty_weight, evaluate_loss_always)
        self._check(self.lib.mcl_gate_begin(self._h, ctypes.byref(rule), self._pinned_status().data_ptr()))

    def verdict(self, vec, iteration, row):
        """the stopping test on the (all-reduced) diagnostics vector `vec` of iteration `iteration`; writes `row` (4 doubles)"""
        self._check(self.lib.mcl_verdict(self._h, vec.data_ptr(), int(iteration), row.data_ptr()))

    def gate_status(self):
        """(stopped, stop_iteration, code) - meaningful once the stream has been synchronised"""
        s = self._pinned_status()
        return bool(int(s[0])), int(s[1]), int(s[2])

    def gate_end(self, stopped_early):
        self._check(self.lib.mcl_gate_end(self._h, int(bool(stopped_early))))

    def run(self, n_iter_max, tol, absolute_tol, feasibility_tol, initial_loss, penalty_weight, evaluate_loss_always,
            update_A=True, update_B=True, update_C=True, max_run_ahead=0):
        """Up to `n_iter_max` outer iterations with the reference's stopping rule evaluated ON THE DEVICE (mcl_run).
        Tolerances: None / 0 = not set.  Returns (n_iter, code, diag [n_iter, DIAG_LEN], verdict [n_iter, 4]) with the
        two rings as NumPy arrays; code 0 = iteration budget exhausted, STOP_RELATIVE / STOP_ABSOLUTE otherwise."""
        torch = self._torch
        n = int(n_iter_max)
        rule = self._stop_rule(tol, absolute_tol, feasibility_tol, initial_loss, penalty_weight, evaluate_loss_always, max_run_ahead)
        ring = torch.zeros((max(n, 1), DIAG_LEN), dtype=torch.float64, device=self.device)
        verdict = torch.zeros((max(n, 1), 4), dtype=torch.float64, device=self.device)
        status = self._pinned_status()
        self._check(self.lib.mcl_run(self._h, n, int(update_A), int(update_B), int(update_C), ctypes.byref(rule),
                                     ring.data_ptr(), verdict.data_ptr(), status.data_ptr()))
        stopped, stop_it, code, _ = (int(v) for v in status)  # mcl_run has synchronised the stream
        n_done = stop_it + 1 if stopped else max(n, 0)
        return n_done, (code if stopped else 0), ring[:n_done].cpu().numpy(), verdict[:n_done].cpu().numpy()

    # -- step calls -----------------------------------------------------------------------------------
    def B_begin(self):
        self._check(self.lib.mcl_B_begin(self._h))

    def B_rho_max(self):
        return self._view(self.lib.mcl_B_rho_max(self._h), 1, self._torch.float32)

    def B_factor(self):
        self._check(self.lib.mcl_B_factor(self._h))

    def B_solve(self):
        self._check(self.lib.mcl_B_solve(self._h))

    def B_prox_local(self, k):
        self._check(self.lib.mcl_B_prox_local(self._h, k))

    def B_prox_reduce_buffer(self, k):
        n = ctypes.c_int64()
        p = self.lib.mcl_B_prox_reduce_buffer(self._h, k, ctypes.byref(n))
        return None if not p else self._view(p, n.value, self._torch.float32)

    def B_prox_finish(self, k):
        self._check(self.lib.mcl_B_prox_finish(self._h, k))

    def B_end(self):
        """issue the deferred prox + dual row pass of a fused stack (before reading B / aux / dual tensors directly)"""
        self._check(self.lib.mcl_B_end(self._h))

    def A_begin(self):
        self._check(self.lib.mcl_A_begin(self._h))

    def A_rho_max(self):
        return self._view(self.lib.mcl_A_rho_max(self._h), 1, self._torch.float32)

    def A_finish(self):
        self._check(self.lib.mcl_A_finish(self._h))

    def A_factor(self):
        self._check(self.lib.mcl_A_factor(self._h))

    def A_solve(self):
        self._check(self.lib.mcl_A_solve(self._h))

    def A_end(self):
        self._check(self.lib.mcl_A_end(self._h))

    def C_begin(self):
        self._check(self.lib.mcl_C_begin(self._h))

    def C_solve(self):
        self._check(self.lib.mcl_C_solve(self._h))

    def C_end(self):
        self._check(self.lib.mcl_C_end(self._h))

    def record_event(self, event):
        """record a torch.cuda.Event on the engine's stream (for a host that reduces on a communication stream of its own)"""
        self._check(self.lib.mcl_record_event(self._h, ctypes.c_void_p(event.cuda_event)))

    def wait_event(self, event):
        self._check(self.lib.mcl_wait_event(self._h, ctypes.c_void_p(event.cuda_event)))

    def rho(self, mode):
        """device fp32 feasibility penalties of the current phase: mode 0 -> [I], 1 -> [I], 2 -> [1]"""
        return self.internal({0: BUF_RHO_A, 1: BUF_RHO_B, 2: BUF_RHO_C}[mode])

    # -- introspection ------------------------------------------------------------------------------------
    def internal(self, which):
        n = ctypes.c_int64()
        p = self.lib.mcl_internal_buffer(self._h, which, ctypes.byref(n))
        return self._view(p, n.value, self._torch.float32)

    def rhses(self):
        return self.internal(BUF_RHSES).view(self.I, self.r)

    def cross_products(self):
        return self.internal(BUF_CROSS_PRODUCTS).view(self.I, self.r, self.r)

    def profile_enable(self, capacity, stride=1):
        """HIP-event pairs around up to `capacity` launches per kernel slot, every `stride`-th launch only."""
        self._check(self.lib.mcl_profile_enable(self._h, int(capacity)))
        self._check(self.lib.mcl_profile_set_stride(self._h, int(stride)))

    def profile_read(self, which):
        """(total_ms, launches timed) of launch site `which` (a PROF_* constant = enum mcl_profile_slot); synchronises."""
        tot, n = ctypes.c_double(), ctypes.c_int32()
        self._check(self.lib.mcl_profile_read(self._h, which, ctypes.byref(tot), ctypes.byref(n)))
        return tot.value, n.value

    def profile_launches(self, which):
        """launches launch site `which` has seen since profile_enable (timed or not)"""
        return int(self.lib.mcl_profile_launches(self._h, which))

    def profile_overhead_us(self):
        """elapsed time of an empty event pair on the engine's stream (calibrated by profile_enable)"""
        return float(self.lib.mcl_profile_overhead_us(self._h))

    def reload_switches(self):
        """re-read the MCL_* environment switches (they are otherwise read once, when the context is created)"""
        self._check(self.lib.mcl_reload_switches(self._h))

    def kernel_variant(self, which):
        return self.lib.mcl_kernel_variant(self._h, which).decode()
