/*
 * matcouply_hip.h - C ABI of the MI355X-native AO-ADMM engine (libmatcouply_hip.so).
 *
 * The reference (MarieRoald/matcouply) is pure Python and has no FFI: its "operator interface" for the
 * hot path is the set of Python functions in src/matcouply/decomposition.py.  Each entry point below
 * replaces one of them; the Python front end `matcouply_amd` (and any other host)
 * binds these symbols with ctypes.  Plain pointers and sizes only - no torch/NumPy types cross the ABI.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on error; mcl_last_error() gives the message;
 *   - the caller owns EVERY buffer (data, factors, aux/dual variables, workspace); the library never
 *     allocates device memory and never synchronises the stream, except where stated;
 *   - all device work is enqueued on the HIP stream given to mcl_create();
 *   - one context per (device, stream); a context is not thread-safe;
 *   - all matrices are row-major fp32; the I coupled matrices X_i (J_i x K) and every B-mode variable are
 *     PACKED along rows: X[sum J_i, K], B[sum J_i, r], slab i = rows row_ptr[i] .. row_ptr[i+1].
 */
#ifndef MATCOUPLY_HIP_H
#define MATCOUPLY_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mcl_context mcl_context;

/* ABI version = what mcl_version() of a matching library returns.  History: 100 first release (fp32 [G | R]);
 * 200 mcl_c_normal_equations returns fp64 (the buffer a multi-GPU host all-reduces changed its element size);
 * 300 mcl_run and the stop-rule structs, the communication-buffer and event entry points;
 * 400 this header: named indices (enum mcl_buffer_id, enum mcl_profile_slot, MCL_VARIANT_EXACT_MODE - the exact-mode query
 *     moved from index 4 to 100), twelve profile slots, mcl_profile_launches, a failed state after mcl_run's watchdog,
 *     mcl_options.inner_tol / exact_products, native GeneralizedL2 / UnitSimplex kinds (mcl_penalty_desc grew two fields),
 *     mcl_penalty_value, mcl_svd_init;
 * 410 mcl_condition_probe, mcl_condition_monitor, mcl_read_bandwidth (nothing else changed).
 * A host MUST compare mcl_version() with the MCL_ABI_VERSION it was built against before any other call. */
#define MCL_ABI_VERSION 410

#define MCL_MAX_REGS 4   /* penalties per mode */
#define MCL_MAX_RANK 64

/* Proximal operators with a native kernel (reference: src/matcouply/penalties.py). */
enum mcl_penalty_kind {
    MCL_PEN_NN = 1,       /* NonNegativity   penalties.py:488-508  */
    MCL_PEN_BOX = 2,      /* Box             penalties.py:511-542  */
    MCL_PEN_L1 = 3,       /* L1Penalty       penalties.py:545-592  */
    MCL_PEN_L2BALL = 4,   /* L2Ball          penalties.py:844-925  */
    MCL_PEN_UNIMODAL = 5, /* Unimodality     penalties.py:983-1015 */
    MCL_PEN_PARAFAC2 = 6, /* Parafac2        penalties.py:1018-1324 (mode 1 only) */
    MCL_PEN_EXTERNAL = 7, /* user prox evaluated by the host between mcl_*_solve and mcl_*_dual */
    MCL_PEN_TV = 8,       /* TotalVariationPenalty  penalties.py:750-841: p0 = TV strength, p1 = L1 strength; the prox
                             (condat_tv.tv_denoise_matrix in the reference) is L. Condat's direct 1-D TV algorithm */
    MCL_PEN_GL2 = 9,      /* GeneralizedL2Penalty   penalties.py:595-747: x^T M x per column, M = U diag(s) U^T given through
                             `matrix` (eigenvectors, then eigenvalues); every matrix of the mode has matrix_rows rows */
    MCL_PEN_SIMPLEX = 10  /* UnitSimplex            penalties.py:928-980: columns non-negative and summing to one */
};

typedef struct {
    int32_t kind;           /* enum mcl_penalty_kind */
    int32_t non_negativity; /* L1 / L2Ball / Unimodality: also clip at 0 */
    double p0;              /* Box: min_val; L1: reg_strength; L2Ball: norm_bound */
    double p1;              /* Box: max_val */
    float *aux;             /* [rows, r] auxiliary variable; PARAFAC2: packed orthogonal bases P_i */
    float *dual;            /* [rows, r] scaled dual variable */
    float *aux2;            /* PARAFAC2: coordinate matrix Delta [r, r]; otherwise NULL */
    const double *matrix;   /* GeneralizedL2: device fp64 [n * n + n]: U (row-major, columns = eigenvectors of M), then the n
                               eigenvalues s (M = U diag(s) U^T, penalties.py:720); otherwise NULL */
    int64_t matrix_rows;    /* GeneralizedL2: n = rows of every factor matrix of the mode */
} mcl_penalty_desc;

typedef struct {
    double feasibility_penalty_scale; /* decomposition.py:678 */
    double l2_penalty[3];             /* per mode; None -> 0 (decomposition.py:876-877) */
    double inner_tol;                 /* decomposition.py:90-117: > 0 ends the inner ADMM loop of a phase early - after an inner
                                         iteration with ||x - x_old|| <= inner_tol ||x|| and every feasibility gap of the mode
                                         below inner_tol.  Evaluated ON THE DEVICE (a flag the remaining inner launches test);
                                         the phases then take one launch per step instead of their fused kernels.  0: not set */
    int32_t inner_n_iter_max;         /* decomposition.py:689 */
    int32_t constant_A;               /* constant_feasibility_penalty for mode 0 (decomposition.py:937-939) */
    int32_t constant_B;               /* ... for mode 1 (decomposition.py:940-942) */
    int32_t exact_products;           /* 0: the library decides by THIS context's size (exact-products mode up to 2^20 elements
                                         of X); 1 / 2: force the mode on / off - a host that shards one problem over several
                                         contexts decides by the size of the WHOLE problem, so that every rank (and every
                                         rank layout) computes with the same arithmetic.  Looked at by mcl_set_workspace. */
} mcl_options;

/* Layout of the fp64 vector written by mcl_diagnostics(). */
#define MCL_DIAG_NORM_SQ 0     /* [3] ||A||^2, ||B||^2, ||C||^2 */
#define MCL_DIAG_INNER 3       /* <X, M>  = sum_i rhs_i . a_i            (decomposition.py:448) */
#define MCL_DIAG_MODEL_SQ 4    /* ||M||^2 = sum_i a_i^T Q_i a_i          (decomposition.py:449) */
#define MCL_DIAG_X_SQ 5        /* ||X||^2 of this context's slabs        (decomposition.py:906) */
#define MCL_DIAG_REG 8         /* + (mode*MCL_MAX_REGS + k)*2 : {||aux - factor||^2, sum |factor|} */
#define MCL_DIAG_LEN (8 + 3 * MCL_MAX_REGS * 2)

/* ---- life cycle ------------------------------------------------------------------------------------ */
int mcl_create(mcl_context **out, int device, void *hip_stream);
void mcl_destroy(mcl_context *ctx);
const char *mcl_last_error(const mcl_context *ctx); /* ctx may be NULL: last mcl_create() failure */
int mcl_version(void);

/* ---- problem definition (replaces the `matrices`, `rank` arguments of cmf_aoadmm, decomposition.py:662) */
/* X: device [row_ptr[I], K]; row_ptr: HOST int64[I+1], non-decreasing, row_ptr[0] = 0. */
int mcl_set_problem(mcl_context *ctx, const float *X, const int64_t *row_ptr, int64_t I, int64_t K, int32_t rank);
int mcl_set_options(mcl_context *ctx, const mcl_options *opt);
/* Factors of the CMF (decomposition.py:133,235,307): A [I, r], B packed [sum J_i, r], C [K, r]; updated in place. */
int mcl_set_factors(mcl_context *ctx, float *A, float *B, float *C);
/* Penalty list of one mode, in the order of decomposition.py:571-612; descs are copied. */
int mcl_set_penalties(mcl_context *ctx, int32_t mode, int32_t n, const mcl_penalty_desc *descs);
/* Scratch memory: call after the four setters above; bytes needed for the current problem. */
int64_t mcl_workspace_bytes(mcl_context *ctx);
int mcl_set_workspace(mcl_context *ctx, void *workspace, int64_t bytes);

/* ---- phase calls: one per reference function -------------------------------------------------------- */
/* admm_update_B (decomposition.py:222-292) */
int mcl_update_B(mcl_context *ctx);
/* admm_update_C (decomposition.py:295-344), split at the cross-slab reduction so that a multi-GPU host can
 * all-reduce the normal equations:  local -> [all-reduce over ranks of mcl_c_normal_equations()] -> finish. */
int mcl_update_C_local(mcl_context *ctx);
double *mcl_c_normal_equations(mcl_context *ctx, int64_t *count); /* device [G (r x r) | R (K x r)], fp64 */
int mcl_update_C_finish(mcl_context *ctx);
/* admm_update_A (decomposition.py:120-219); also leaves (rhses, cross_products) for the fast error formula. */
int mcl_update_A(mcl_context *ctx);
/* compute_feasibility_gaps + _cmf_reconstruction_error + penalty sums (decomposition.py:351-452, 916-921):
 * writes MCL_DIAG_LEN fp64 partial sums of THIS context's slabs to device memory `out`.
 * include_replicated = 0 leaves the C-mode entries (identical on every rank) at zero so that a SUM all-reduce
 * over ranks is exact.  If no A-phase by-products are current, they are recomputed with a pass over X
 * (decomposition.py:430-444). */
int mcl_diagnostics(mcl_context *ctx, double *out, int32_t include_replicated);
/* mcl_diagnostics whose table reduction MAY be issued later: when the next calls on this context are mcl_update_B and
 * mcl_update_C_local on the one-pass path, it rides on a spare workgroup of the C-phase reduction kernel instead of a launch
 * of its own (the per-iteration pattern B -> C_local -> [all-reduce] -> C_finish -> A -> diagnostics of a fixed-count loop).
 * `out` is written at the latest when ANY other entry point of the context, or mcl_flush_diagnostics(), has been called;
 * the factors / tables it reports are those at the time of THIS call.  Same values as mcl_diagnostics up to the association
 * of the fp64 sums. */
int mcl_diagnostics_deferred(mcl_context *ctx, double *out, int32_t include_replicated);
int mcl_flush_diagnostics(mcl_context *ctx);
/* Value of penalty k of `mode` on the current factor, for the kinds whose value is not a column of the diagnostics vector:
 * GeneralizedL2 - sum over the mode's matrices of trace(F^T M F) (penalties.py:737-745).  One fp64 to device memory `out`. */
int mcl_penalty_value(mcl_context *ctx, int32_t mode, int32_t k, double *out);
/* Condition estimates of the r x r normal equations the PENALTY-FREE modes among `mode_mask` (bit m = mode m) would solve from
 * the CURRENT factors - the systems the reference solves with an fp64 SVD (decomposition.py:155-172 mode 0, :240-256 mode 1,
 * :312-321 mode 2; l2_penalty included): out[m] = ||M||_F ||M^-1||_F (between cond_2 and r cond_2; modes 0 / 1: the largest
 * over the matrices; 1e300: singular), 0 for a mode that has penalties or is not in the mask.  Computed from the factors alone
 * (fp64 Gram matrices of the B_i, no pass over X), three fp64 to DEVICE memory `out`, enqueued like everything else.  The
 * fp32 kernels leave 1e-8 .. 4e-7 relative in what such a solve sees and the solve multiplies it by this number: a host that
 * wants the reference's results on an ill-conditioned penalty-free mode switches to mcl_options.exact_products = 1 when the
 * estimate is large (matcouply_amd does so between 2^20 and 2^24 elements of X above 1e3, and warns beyond). */
int mcl_condition_probe(mcl_context *ctx, int32_t mode_mask, double *out);
/* The same estimate taken WHERE IT MATTERS: the phases are Gauss-Seidel - the A-phase of an iteration solves systems built
 * from the B_i and C of the same iteration - so the numbers that count are those at the start of each phase.  While a monitor is
 * installed (out != NULL: device fp64[4], zeroed by the caller), every phase of a penalty-free mode in `mode_mask` (mcl_update_B,
 * mcl_update_C_local, mcl_update_A / mcl_A_begin) first launches the probe for ITS system and keeps the running maximum in
 * out[mode]; out[3] collects the worst conditioning of the PARAFAC2 polar factors (~ ||sigma|| / sigma_min of Y_i Delta^T,
 * penalties.py:1233-1235; 1e8 for a matrix the Newton-Schulz route had to hand on) of every inner iteration, for ranks <= 32;
 * out = NULL removes it.  Two small launches per monitored phase: meant for a short trial (matcouply_amd runs two
 * iterations under the monitor, restores the initial state and then chooses the arithmetic) and for an occasional check. */
int mcl_condition_monitor(mcl_context *ctx, double *out, int32_t mode_mask);
/* n outer iterations B -> C -> A on ONE device (decomposition.py:945-988); if diag_ring != NULL,
 * MCL_DIAG_LEN doubles are appended per iteration (device memory, n * MCL_DIAG_LEN doubles). */
int mcl_iterate(mcl_context *ctx, int32_t n_iter, int32_t update_A, int32_t update_B, int32_t update_C,
                double *diag_ring);

/* The outer loop WITH its stopping rule on one device (decomposition.py:945-1053): the reduction of the diagnostics and
 * the stopping test (feasibility gaps :996, relative / absolute loss criterion :1037-1053, quirks Q8-Q10 of SURVEY.md
 * Appendix C kept) run in a kernel at the end of every iteration; the host enqueues up to max_run_ahead iterations
 * ahead of the verdicts it has seen and never waits for one.  Every kernel that writes a factor or an ADMM variable tests
 * the device-side stop flag first, so when the call returns the factors and ADMM variables are EXACTLY those of the
 * stopping iteration, however far the host had run ahead.
 *   rule          Python truthiness is kept: a tolerance of 0 means "not set" (None / 0 / False of the reference);
 *   diag_ring     device fp64 [n_iter_max, MCL_DIAG_LEN]: row `it` = the mcl_diagnostics vector after iteration `it`;
 *   verdict_ring  device fp64 [n_iter_max, 4]: {relative reconstruction error, regularised loss, worst feasibility gap,
 *                 flags: bit 0 feasible, bit 1 error / loss evaluated (Q10), bits 2.. stop code}; rows beyond the stopping
 *                 iteration are not written;
 *   status        int32[4] in PINNED host memory (hipHostMalloc / a registered buffer): written by the device.
 * Unlike every other entry point this one synchronises the stream before it returns (it reports the verdict).
 * Watchdog: a device that reports no verdict for MCL_RUN_WATCHDOG_S seconds (default 120) ends the call with an error and
 * WITHOUT that synchronisation (a wedged stream would block it forever).  The context is then FAILED: every later entry
 * point but mcl_last_error / mcl_destroy returns an error, and the caller must keep the workspace, both rings and `status`
 * allocated until it has synchronised or reset the device itself (kernels that refer to them may still be enqueued). */
typedef struct {
    double tol;                 /* relative loss criterion, decomposition.py:1039; 0 = not set */
    double absolute_tol;        /* decomposition.py:1040; only looked at when tol is set (Q8); tests the newest loss (Q9) */
    double feasibility_tol;     /* decomposition.py:996; 0 = not set: the feasibility criterion then never holds */
    double initial_loss;        /* losses[0]: regularised loss of the initial state (decomposition.py:916-921) */
    double penalty_weight[3][MCL_MAX_REGS]; /* loss += weight * sum |factor|: reg_strength of an L1Penalty, else 0 */
    int32_t evaluate_loss_always; /* return_errors of the reference: the loss is evaluated on infeasible iterates too (Q10) */
    int32_t max_run_ahead;      /* iterations the host may enqueue beyond the newest verdict it has seen; <= 0: 8 */
} mcl_stop_rule;
#define MCL_STOP_RELATIVE 1     /* "FEASIBILITY GAP CRITERION AND RELATIVE LOSS CRITERION SATISFIED" */
#define MCL_STOP_ABSOLUTE 2     /* "FEASIBILITY GAP CRITERION AND ABSOLUTE LOSS CRITERION SATISFIED" */
typedef struct {
    volatile int32_t stopped;         /* 1 once a criterion has fired */
    volatile int32_t stop_iteration;  /* 0-based outer iteration it fired on (n_iter of the reference = this + 1) */
    volatile int32_t code;            /* MCL_STOP_RELATIVE / MCL_STOP_ABSOLUTE */
    volatile int32_t progress;        /* iterations whose verdict has been evaluated so far */
} mcl_run_status;
int mcl_run(mcl_context *ctx, int32_t n_iter_max, int32_t update_A, int32_t update_B, int32_t update_C,
            const mcl_stop_rule *rule, double *diag_ring, double *verdict_ring, mcl_run_status *status);

/* The same rule for a host that drives the iterations itself - the sharded loop, whose diagnostics vector only exists after an
 * all-reduce over the ranks:  mcl_gate_begin -> per iteration { phase / step calls with their reductions, mcl_diagnostics(ctx,
 * vec, rank == 0) -> [SUM all-reduce of vec] -> mcl_verdict(ctx, vec, it, row) } -> (synchronise, read `status`) -> mcl_gate_end.
 * Between begin and end every state-writing kernel tests the stop flag, so iterations enqueued behind a hit do nothing
 * (their collectives still run: every rank must enqueue the SAME number of iterations before it looks at `status` - e.g.
 * fixed chunks with a synchronisation in between; the ranks reach identical verdicts because they evaluate identical bits).
 * mcl_gate_end(ctx, 1) after an early stop makes the context forget its cached by-products. */
int mcl_gate_begin(mcl_context *ctx, const mcl_stop_rule *rule, mcl_run_status *status);
int mcl_verdict(mcl_context *ctx, const double *diag_vec, int32_t iteration, double *verdict_row);
int mcl_gate_end(mcl_context *ctx, int32_t stopped_early);

/* ---- step calls (constant feasibility penalty / PARAFAC2 / EXTERNAL penalties on several devices) ---- */
/* B-phase prologue: CtC, rhs_i, rho_i; returns pointer to this rank's max rho (device fp32[1]) for a MAX all-reduce */
int mcl_B_begin(mcl_context *ctx);
float *mcl_B_rho_max(mcl_context *ctx);
int mcl_B_factor(mcl_context *ctx);          /* build and invert the r x r systems (decomposition.py:252-256) */
int mcl_B_solve(mcl_context *ctx);           /* one normal-equation solve (decomposition.py:266-273) */
int mcl_B_prox_local(mcl_context *ctx, int32_t k);  /* prox of penalty k; PARAFAC2: bases + local Delta sums */
float *mcl_B_prox_reduce_buffer(mcl_context *ctx, int32_t k, int64_t *count); /* PARAFAC2: [sum rho P^T Y | sum rho] */
int mcl_B_prox_finish(mcl_context *ctx, int32_t k); /* PARAFAC2: Delta; all: dual update (decomposition.py:282-285) */
/* Contract of one inner iteration: mcl_B_solve, then for EVERY penalty k in order: mcl_B_prox_local(k) [all-reduce of
 * mcl_B_prox_reduce_buffer(k) for PARAFAC2] mcl_B_prox_finish(k).  For stacks of row-separable kinds, L2 balls and
 * PARAFAC2 the library defers the aux / dual row updates of all penalties to ONE pass issued by the last
 * mcl_B_prox_finish of the round, so the rows are only final once the whole stack has been stepped. */
/* That pass is issued by the NEXT entry point: merged with the solve when it is mcl_B_solve (one row pass per inner
 * iteration), on its own otherwise.  mcl_B_end issues it explicitly - call it before reading B / aux / dual buffers
 * directly (not through the library) after the last inner iteration. */
int mcl_B_end(mcl_context *ctx);
int mcl_A_begin(mcl_context *ctx);           /* X C, rhs_i, Q_i, rho_i (decomposition.py:136-162) */
float *mcl_A_rho_max(mcl_context *ctx);
int mcl_A_finish(mcl_context *ctx);          /* systems, inner ADMM loop, by-products (decomposition.py:163-219) */
/* Host-evaluated (MCL_PEN_EXTERNAL) penalties on modes 0 / 2: the library does the linear algebra, the host the prox.
 *   mode 0:  mcl_A_begin -> [MAX all-reduce] -> mcl_A_factor -> { mcl_A_solve -> host prox/dual }* -> mcl_A_end
 *   mode 2:  mcl_update_C_local -> [all-reduce] -> mcl_C_begin -> { mcl_C_solve -> host prox/dual }* -> mcl_C_end
 *   mode 1:  mcl_B_begin -> mcl_B_factor -> { mcl_B_solve -> per penalty: native mcl_B_prox_* or host prox/dual }*
 * The aux buffer of an EXTERNAL penalty holds the auxiliary variable AS A MATRIX (aux_as_matrix of the reference). */
int mcl_A_factor(mcl_context *ctx);          /* Q_i, rho_i, L_i^-1 (decomposition.py:162-172) */
int mcl_A_solve(mcl_context *ctx);           /* one row solve for all rows (decomposition.py:184-195) */
int mcl_A_end(mcl_context *ctx);             /* by-products for the fast error formula (decomposition.py:219, 445-449) */
int mcl_C_begin(mcl_context *ctx);           /* rho, L^-1 from the reduced normal equations (decomposition.py:319-321) */
int mcl_C_solve(mcl_context *ctx);           /* decomposition.py:328-331 */
int mcl_C_end(mcl_context *ctx);             /* invalidates everything derived from C */

/* ---- ordering against other streams (a host that runs its collectives on a communication stream of its own) ------------ */
/* The library enqueues everything on the stream of mcl_create().  A host whose collective library wants its own stream
 * chains it with two events per collective:  mcl_update_C_local -> mcl_record_event(ctx, e1) -> [comm stream: wait e1,
 * all-reduce mcl_c_normal_equations() in place, record e2] -> mcl_wait_event(ctx, e2) -> mcl_update_C_finish.
 * (A host that can enqueue its collective on the engine's stream itself - RCCL takes any stream - needs neither: that
 * is what matcouply_amd/_rccl.py does, and it saves the ~8 us of the hand-over.)  `hip_event`: a hipEvent_t. */
int mcl_record_event(mcl_context *ctx, void *hip_event);
int mcl_wait_event(mcl_context *ctx, void *hip_event);

/* ---- the step after the solver: dense reconstruction (replaces cmf_to_matrices, coupled_matrices.py:365-497) ---------- */
/* M_i = (B_i diag(weights o a_i)) C^T for all matrices, packed along rows like X.  Stateless (no context): A [I, r],
 * B packed [N, r], C [K, r], weights [r] or NULL, slab_of_row int32 [N] (matrix index of every packed row), out [N, K] -
 * all device pointers; enqueued on hip_stream. */
int mcl_cmf_to_packed(const float *A, const float *B, const float *C, const float *weights, const int32_t *slab_of_row,
                      int64_t N, int64_t K, int32_t rank, float *out, void *hip_stream);

/* ---- before the solver: init="svd" / "threshold_svd" (decomposition.py:41-54) for data resident in HBM ---------------------- */
/* B_i = the leading `rank` left singular vectors of X_i (packed [N, rank]), C = the leading right singular vectors of the
 * stacked matrices [K, rank]; threshold != 0 clips negative entries (threshold_svd).  Stateless; X, B, C, workspace, info:
 * device pointers; row_ptr: HOST int64[I+1].  fp64 subspace iteration on the K x K Gram matrices (csrc/svdinit.hip).  The
 * entry of largest magnitude of every column of B_i and of C is positive: a singular vector's sign is the driver's choice,
 * so the vectors equal LAPACK's up to these signs.  info: int32[I+1] - iterations used per matrix (last: the stack), negative
 * when the Ritz values had not settled after 400.  Synchronises the stream once (the upload of row_ptr). */
int64_t mcl_svd_init_workspace_bytes(const int64_t *row_ptr, int64_t I, int64_t K, int32_t rank);
int mcl_svd_init(const float *X, const int64_t *row_ptr, int64_t I, int64_t K, int32_t rank, int32_t threshold, float *B, float *C,
                 void *workspace, int64_t workspace_bytes, int32_t *info, void *hip_stream);
const char *mcl_svd_init_last_error(void);

/* ---- introspection for tests / profiling ------------------------------------------------------------- */
/* device pointers to internal by-products / planner tables (the int32 tables: read the bits) */
enum mcl_buffer_id {
    MCL_BUF_RHSES = 0,          /* rhses [I, r] of the A-phase (decomposition.py:147-152) */
    MCL_BUF_CROSS_PRODUCTS = 1, /* cross_products [I, r, r] (decomposition.py:153-161) */
    MCL_BUF_XC = 2,             /* X C [sum J_i, r] */
    MCL_BUF_RHO_B = 3,          /* [I] */
    MCL_BUF_RHO_A = 4,          /* [I] */
    MCL_BUF_RHO_C = 5,          /* [1] */
    MCL_BUF_CTC = 6,            /* C^T C [r, r] */
    MCL_BUF_LINV_B = 7,         /* inverses of the B-phase systems [I, r, r] */
    MCL_BUF_PF2_STATUS = 8,     /* int32 [I]: which polar-factor kernel took the slab */
    MCL_BUF_PF2_ACC = 9,        /* fp64 bits */
    MCL_BUF_PF2_GRAM = 10,      /* fp64 bits [I, r, r] */
    MCL_BUF_SWEEP_CYCLES = 11,  /* int64 bits (MCL_SWEEP_DBG & 32) */
    MCL_BUF_SEG_ROW0 = 12,      /* int32: first row of every segment of the X passes */
    MCL_BUF_SEG_NROWS = 13,     /* int32: its length */
    MCL_BUF_WAVE_SEG_PTR = 14,  /* int32: first segment of every wave */
    MCL_BUF_BSEG_ROW0 = 15,     /* the same three tables for the bsegs of the one-pass sweep */
    MCL_BUF_BSEG_NROWS = 16,
    MCL_BUF_WAVE_BSEG_PTR = 17,
    MCL_BUF_NS_STAMPS = 18,     /* instrumented builds (-DMCL_NS_STAMPS) only */
    MCL_BUF_BSEG_PART = 19      /* int32 per bseg: bits 0-27 the partial image the sweep adds the bseg to */
};
float *mcl_internal_buffer(mcl_context *ctx, int32_t which /* enum mcl_buffer_id */, int64_t *count);
/* The hot launch sites of the library, for HIP-event timing (bench.py's roofline block) and for asking which kernel form a
 * problem gets.  A slot covers every launch of its role; roles that a configuration does not use stay empty. */
enum mcl_profile_slot {
    MCL_PROF_XC = 0,         /* X C pass (k_contract_xc_*) */
    MCL_PROF_XT = 1,         /* X^T (B o a) pass (k_contract_xt) */
    MCL_PROF_ROWS_FUSED = 2, /* fused inner ADMM loop of row-separable stacks (k_rows_fused) */
    MCL_PROF_SWEEP = 3,      /* one-pass sweep: X C -> B-phase -> X^T B in a single pass over X (k_sweep) */
    MCL_PROF_REDUCE = 4,     /* [G | R] from the partials (k_reduce_frag / k_reduce_partials / k_exact_gr) */
    MCL_PROF_C_FINISH = 5,   /* C-phase finish (k_C_finish_* / k_C_prepare + row kernels) */
    MCL_PROF_A_FINISH = 6,   /* A-phase finish (k_A_finish*; k_CA_finish when it also holds the C-phase finish) */
    MCL_PROF_ROWS_CHAIN = 7, /* one chained row pass of a generic B stack (k_rows_solve_stats / _finish_solve_stats / _finish_fused) */
    MCL_PROF_UNIMODAL = 8,   /* unimodal regressions (k_slab_unimodal_*) */
    MCL_PROF_PF2 = 9,        /* per-slab PARAFAC2 algebra of one inner iteration (polar factors + coordinate matrix) */
    MCL_PROF_DIAG = 10,      /* reduction of the diagnostics tables (k_diag_final / k_diag_verdict) */
    MCL_PROF_OTHER = 11,     /* every other launch of a phase (statistics reductions, system builds, ...) */
    MCL_PROF_SLOTS = 12
};
/* name of the kernel variant chosen for the current problem.  which = an mcl_profile_slot: the kernel last launched in that
 * role ("" when the problem has not used the role yet; MCL_PROF_SWEEP stays empty when the problem is not eligible for the
 * one-pass sweep); MCL_VARIANT_EXACT_MODE: non-empty when the problem runs in the exact-products mode (at most 2^20 elements
 * of X: every contraction as fp64 sums of exact products) */
#define MCL_VARIANT_EXACT_MODE 100
const char *mcl_kernel_variant(mcl_context *ctx, int32_t which);
/* HIP-event timing of the launch sites on the context's stream.
 * mcl_profile_enable(ctx, capacity): record up to `capacity` launches per slot (0 disables and frees);
 * mcl_profile_read(ctx, slot, &total_ms, &count): synchronises the recorded events of `slot`, returns their summed
 * duration and number, and resets the slot; mcl_profile_launches(ctx, slot): launches the slot has SEEN since
 * mcl_profile_enable / mcl_profile_set_stride (timed or not) - launches per step = this / steps. */
int mcl_profile_enable(mcl_context *ctx, int32_t capacity);
/* record only every `stride`-th launch of a slot (default 1): an event pair opens ~5 us dispatch gaps on either side of
 * the kernel, so a timed loop samples its launches instead of bracketing all of them */
int mcl_profile_set_stride(mcl_context *ctx, int32_t stride);
int mcl_profile_read(mcl_context *ctx, int32_t which /* enum mcl_profile_slot */, double *total_ms, int32_t *count);
int64_t mcl_profile_launches(mcl_context *ctx, int32_t which /* enum mcl_profile_slot */);
/* what an event pair adds to the launch it brackets (microseconds), calibrated by mcl_profile_enable on the context's stream:
 * a pair around one tiny operation minus that operation's marginal cost inside a pair around two of them - the part of every
 * timed launch that is the command processor's marker handling, not the kernel */
double mcl_profile_overhead_us(mcl_context *ctx);
/* The MCL_* environment switches (A/B experiments, debug paths; tools/README.md) are read once in mcl_create();
 * this re-reads them for an existing context (tests that compare kernel forms on one problem).
 * PRODUCTION NOTE: the parity statements of this library hold for a CLEAN environment - several
 * switches select other kernel forms or other summation orders (MCL_SEG_ROWS, MCL_XC_WAVES, MCL_NO_ROWS64, MCL_NO_SWEEP, ...)
 * and move results at the 1e-6 level.  mcl_active_switches() returns the space-separated names of the MCL_* switches the
 * context found in its environment ("" = clean); a host should refuse, or at least log, a non-empty answer
 * (matcouply_amd.cmf_aoadmm issues a RuntimeWarning).  A library built with -DMCL_NO_ENV_SWITCHES
 * (MCL_BUILD_DEFS=-DMCL_NO_ENV_SWITCHES python -c "import __graft_entry__ as g; g.build()") never consults the
 * environment: every switch keeps its default and mcl_active_switches() is always "". */
/* What the box's memory system delivers to a pure streaming READ of `bytes` of device memory at `buf` (>= 1 MiB; take more than
 * the 256 MB last-level cache for an HBM figure): the better of two access geometries, `repeats` timed launches each, GB/s to
 * HOST `gbps`.  Stateless measurement helper (bench.py: roofline.frac_achievable); `scratch`: one device float; synchronises. */
int mcl_read_bandwidth(const void *buf, int64_t bytes, int32_t repeats, float *scratch, void *hip_stream, double *gbps);
int mcl_reload_switches(mcl_context *ctx);
const char *mcl_active_switches(const mcl_context *ctx);

#ifdef __cplusplus
}
#endif
#endif /* MATCOUPLY_HIP_H */
