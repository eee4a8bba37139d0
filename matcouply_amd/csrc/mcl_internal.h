// Internal declarations shared by the translation units of libmatcouply_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/matcouply_hip.h"

#define DIAG_COLS (2 + MCL_MAX_REGS)
#define MCL_SEG_ROWS 256

#define MCL_CHECK_HIP(ctx, expr)                                                                      \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess) {                                                                       \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(_e);                           \
            return 1;                                                                                 \
        }                                                                                             \
    } while (0)

// Descriptor of one mode's penalty list as the kernels see it (passed by value).
struct RegSet {
    int n;
    int kind[MCL_MAX_REGS];
    int nonneg[MCL_MAX_REGS];
    float p0[MCL_MAX_REGS];
    float p1[MCL_MAX_REGS];
    float *aux[MCL_MAX_REGS];
    float *dual[MCL_MAX_REGS];
    float *aux2[MCL_MAX_REGS];
    double p0d[MCL_MAX_REGS];  // the same parameters before their rounding to fp32 (the fp64 inner loops of wide.hip)
    double p1d[MCL_MAX_REGS];
    const double *mat[MCL_MAX_REGS];  // GeneralizedL2: U [n, n] then s [n] (fp64, device)
    int mat_rows[MCL_MAX_REGS];
    const int *gate;  // stop flag of a gated run (mcl_run with a stopping rule), else NULL: see MCL_GATE
};

// Gated runs (mcl_run with a stopping rule): the host enqueues outer iterations AHEAD of the device-side verdict.  Every
// kernel that writes a factor or an ADMM variable starts with this test of the stop flag the verdict kernel sets, so the
// iterations enqueued behind the stopping one leave the state exactly as the stopping iteration left it (kernels that only
// write scratch run on unchanged inputs; the host invalidates every cached by-product after an early stop).
#define MCL_GATE(p)                                   \
    do {                                              \
        if ((p) != nullptr && *(p) != 0) return;      \
    } while (0)

// The per-row / per-tile fp64 diagnostic tables of the three modes and what else the MCL_DIAG_LEN vector is made of
// (k_diag_final, and the spare workgroup of k_reduce_frag that takes over a DEFERRED reduction: admm.hip / sweep.hip).
struct DiagTables {
    const double *tab[3];
    int rows[3];
    int nreg[3];
    const double *e1;
    int I;
    const double *xsq;
};

// Row tiling of a packed [rows, r] matrix: every wave tile holds <= 64 rows of ONE slab.
struct TileMap {
    int n_tiles = 0;
    int *slab = nullptr;   // device int32[n_tiles]
    int *row0 = nullptr;   // device int32[n_tiles]
    int *nrows = nullptr;  // device int32[n_tiles]
};

// One mode's factor as the row kernels see it (generic.hip: view_of; passed by value)
struct ModeView {
    const int *tile_slab, *tile_row0, *tile_nrows;
    int n_tiles;
    const int *ext;        // slab extents (row_ptr)
    int n_slabs;
    const float *rho;      // [n_slabs]
    float *F;              // factor [rows, r]
    const int *gate;       // stop flag of a gated run (mcl_run with a stopping rule), else NULL: see MCL_GATE
};

// MCL_* environment switches (A/B experiments, debug paths): read ONCE per context in mcl_create(), never on a launch
// path - a getenv() is a linear scan of the environment and is not safe against a concurrent setenv().
struct mcl_switches {
    bool no_sweep = false, no_pass_chain = false, no_pf2_delta_fusion = false, ns_plain = false, pf2_jacobi = false;
    bool no_stack_fusion = false, no_solve_stats = false, no_next_b = false, no_fused_gram = false, no_fused_c = false;
    bool a_finish_cols = false, xc_norow = false, uni_noprune = false, stats_reduce = false;
    bool no_rows64 = false, no_uni_coop = false, no_wide = false, no_row_prefetch = false, no_xc_lds = false;
    bool no_a_fusion = false, no_a_wide = false, no_bseg_groups = false, no_x_nt = false, no_sweep_half = false, no_multi_c = false, no_diag_defer = false, xc_depth1 = false;
    int x_nt_mb = 0, seg_rows = 0, bseg_rows = 0, xc_waves = 0, xt_waves = 0, sweep_waves = 0;  // 0: default
    int uni_wpb = 0, xc_lds_depth = 0, xc_dbg = 0, xt_dbg = 0, xt_depth = 0, sweep_dbg = 0, reduce_el = 0, uni_split = -1;
    int exact = -1;  // MCL_EXACT: 1 / 0 force the exact-products mode on / off (default -1: by problem size, mcl_exact_mode)
    long run_spins = 2000;          // mcl_run: polite spins of a wait before it starts to sleep (MCL_RUN_SPINS)
    bool test_mute_verdict = false;  // MCL_TEST_MUTE_VERDICT: mcl_run's verdict kernels report into scratch (watchdog test)
    double run_watchdog_s = 120.0;  // mcl_run: seconds without a verdict from the device before the wait gives up (MCL_RUN_WATCHDOG_S)
};

struct mcl_context {
    int device = 0;
    mcl_switches sw;
    std::string active_switches;  // names of the MCL_* switches found in the environment when they were last read
    hipStream_t stream = nullptr;
    std::string err;
    bool failed = false;      // set when mcl_run's watchdog gave up: every later entry point refuses the context
    std::string failed_why;

    // problem
    const float *X = nullptr;
    std::vector<int64_t> row_ptr;  // host
    int64_t I = 0, K = 0, N = 0;
    int64_t max_slab_rows = 0;  // longest matrix (rows)
    int r = 0;
    int RP = 0;  // rank padded to 4/8/16/32/64 (register tiles)
    int NB = 0;  // number of 16-wide MFMA column blocks = ceil(r/16)
    mcl_options opt{};
    float *A = nullptr, *B = nullptr, *C = nullptr;
    RegSet regs[3]{};
    bool has_problem = false, has_factors = false, has_workspace = false;

    // host-side tile maps (uploaded into the workspace)
    std::vector<int> h_slab_of_row, h_tile_slab, h_tile_row0, h_tile_nrows;
    std::vector<int> h_ctile_slab, h_ctile_row0, h_ctile_nrows;  // C-mode: one slab of K rows
    std::vector<int> h_atile_slab, h_atile_row0, h_atile_nrows;  // A-mode: one slab of I rows
    std::vector<int> h_seg_slab, h_seg_row0, h_seg_nrows;
    std::vector<int> h_bseg_slab, h_bseg_row0, h_bseg_nrows, h_slab_bseg_ptr;  // work units of the one-pass sweep

    // workspace carve-up (device pointers)
    char *ws = nullptr;
    int64_t ws_bytes = 0;
    int *slab_of_row = nullptr;
    int *row_ptr_dev = nullptr;  // int32[I+1]
    TileMap tilesB, tilesC, tilesA;
    TileMap segs;  // <= MCL_SEG_ROWS-row segments of one slab each: work units of the X^T (B o a) pass
    TileMap bsegs;  // block segments (<= 1024 rows of one slab) of k_sweep (sweep.hip)
    int *slab_bseg_ptr = nullptr;  // int32[I+1] first bseg of every slab
    float *Mpart = nullptr;        // [n_bsegs, K * 16 NB]  per-bseg X^T B in C-fragment order
    double *part_btb = nullptr;    // [n_bsegs, r, r]       per-bseg B^T B (fp64 image of the fp32 accumulators)
    float *CfragS = nullptr;       // the sweep's view of the fragment image of C (aliases Cfrag: one shared image)
    float *GRpart = nullptr;       // [n_bsegs, (16 NB)^2 + 16 NB]  per-bseg a-weighted Gram and the a_i it was weighted with
    int n_grpart = 0;
    long long *sweep_cycles = nullptr;  // [n_blocks, 4 waves, 6] per-section cycle counts (MCL_SWEEP_DBG & 32)
    bool grpart_valid = false;     // GRpart was weighted with the current A (and mseg_valid)
    bool sweep_planned = false;    // the workspace holds the sweep buffers
    // exact-products mode: fp64 shadow state of the inner loops of modes 1 / 2 (wide.hip): factor, auxiliary and dual
    // variables for the length of a phase, the PARAFAC2 coordinate matrix
    double *wF[3] = {nullptr, nullptr, nullptr};
    double *wZ[3][MCL_MAX_REGS] = {}, *wU[3][MCL_MAX_REGS] = {};
    double *wD = nullptr;
    double *LinvA64 = nullptr, *rhsA64 = nullptr, *Q64 = nullptr;  // fp64 copies of the A-phase systems / right-hand sides / cross products (mode 0 of wide.hip, k_A_e1)
    bool a64_valid = false;  // ... and they belong to the systems mcl_launch_A_finish(fused_inner = false) built last
    double *exact_part = nullptr;  // exact-products mode: [G | R] per 256-row chunk (fp64), summed in a fixed order
    bool exact = false;            // exact-products mode (small problems): X C, [G | R] and the A-phase tables from fp64 sums of exact products
    bool mseg_valid = false;       // Mpart / part_btb correspond to the current B
    bool seg_from_sweep = false;   // k_A_finish sums seg_rhs / part_btb over bsegs instead of segments
    float *XC = nullptr;        // [N, r]   X C  (cached between the A-phase and the next B-phase)
    float *Cfrag = nullptr;     // C in MFMA-fragment order for the X C kernel
    float *CtC = nullptr;       // [r, r]
    double *CtC64 = nullptr;    // [r, r] the same product before its rounding to fp32 (systems of the A- and B-phase)
    float *rhoB = nullptr;      // [I]
    float *LinvB = nullptr;     // [I, r, r]
    double *LinvB64 = nullptr;  // [I, r, r] fp64 copy, only when mode 1 has no penalty (fp64 solve of the un-shifted systems)
    double *XC64 = nullptr;     // [N, r]  X C in fp64 for that solve (same condition)
    float *rho_max = nullptr;   // [2]  (0: B-phase, 1: A-phase)
    double *partials = nullptr; // [n_part, K*r + r*r]  per-block fp64 partials of the X^T (B o a) pass
    double *GR = nullptr;       // [r*r + K*r]  fp64 normal equations of the C-phase (all-reduced by a multi-GPU host)
    float *GRf = nullptr;       // fp32 image of R for the fp32 row kernels (written by k_C_prepare)
    double *LinvC64 = nullptr;  // [r, r] fp64 copy of the C-phase inverse (penalty-free C: fp64 solve)
    float *rhoC = nullptr;      // [1]
    float *LinvC = nullptr;     // [r, r]
    double *seg_rhs = nullptr;  // [max(n_segs, n_bsegs, I), r]  fp64 per-segment (per-bseg / per-slab) partial rhs_i
    double *seg_btb = nullptr;  // [max(n_segs, I), r, r]        fp64 per-segment (per-slab) partial B_i^T B_i
    int *slab_seg_ptr = nullptr;  // int32[I+1] first segment of every slab
    // partials of the sweep: normally one per bseg; when every wave holds one short bseg, the four bsegs of a workgroup
    // that belong to the same slab share ONE (bit 30 of bseg_part: k_sweep<.., GRP>)
    int *bseg_part = nullptr;       // int32[n_bsegs] partial index (| 1 << 30: grouped)
    int *slab_part_ptr = nullptr;   // int32[I+1] first partial of every slab
    std::vector<int> h_bseg_part, h_slab_part_ptr;
    int n_parts = 0;
    bool x_streams = false;         // X is larger than the last-level cache (256 MB): the X kernels load it non-temporally
    int sweep_kc = 4;               // 64-column chunks of the sweep's tiles / partial images (mcl_sweep_KC)
    int *wave_bseg_ptr = nullptr;  // int32[n_bseg_waves+1] first bseg of every wave of the sweep
    std::vector<int> h_wave_bseg_ptr;
    int n_bseg_waves = 0;
    int *wave_seg_ptr = nullptr;  // int32[n_seg_waves+1] first segment of every wave of the X passes (balanced by blocks)
    std::vector<int> h_wave_seg_ptr;
    int n_seg_waves = 0;
    std::vector<int> h_slab_seg_ptr;
    bool xc_with_gram = false;  // request: the next X C launch also produces seg_rhs / seg_btb
    bool xc_did_gram = false;   // the last X C launch produced them
    bool use_seg_gram = false;  // k_A_finish sums seg_rhs / seg_btb instead of reading rhsA / BtB
    float *rhsA = nullptr;      // [I, r]
    float *BtB = nullptr;       // [I, r, r]  -> overwritten by Q_i = BtB_i o CtC (cross_products)
    float *rhoA = nullptr;      // [I]
    float *LinvA = nullptr;     // [I, r, r]
    double *e1 = nullptr;       // [I, 2]   per-slab <X_i, M_i>, ||M_i||^2
    double *diagA_row = nullptr;   // [I, DIAG_COLS]        per-row sums of mode 0 (A-phase kernels)
    double *diagA_tile = nullptr;  // [tilesA, DIAG_COLS]
    double *diagB_tile = nullptr;  // [tilesB, DIAG_COLS]   per-tile ||F||^2, sum|F|, ||Z_k - F||^2
    double *diagC_tile = nullptr;  // [tilesC, DIAG_COLS]
    double *diag_sums = nullptr;   // [3*DIAG_COLS + 2]
    double *xsq_part = nullptr;    // [1024]
    int *ext_A = nullptr, *ext_C = nullptr;  // int32[2] slab extents {0, I} / {0, K} for the single-slab modes
    double *x_sq = nullptr;     // [1]
    double *cond_monitor = nullptr;  // mcl_condition_monitor: device fp64[4] running maxima (modes 0-2, PARAFAC2 polar factors), NULL = off
    int cond_monitor_mask = 0;
    float *row_sink = nullptr;  // [64 waves, 64 lanes, 4] where the software-pipelined row passes (rowchain.hip) send the stores of lanes outside the matrix
    double *cond_part = nullptr;  // [256, r*r + 2] mcl_condition_probe: per group of matrices the a-weighted Gram sum and the worst kappa of modes 0 / 1
    // deferred diagnostics (mcl_diagnostics_deferred): the reduction of the tables rides on a spare workgroup of the NEXT
    // C-phase reduction kernel instead of a launch of its own; the sweep alternates between two B tables so that the
    // tables of iteration t stay intact while the sweep of t + 1 writes its own
    double *diagB_bufs[2] = {nullptr, nullptr};
    int diagB_parity = 0;
    // gated runs (mcl_run): device state of the stopping rule
    int *mute_status = nullptr;        // int32[4]: where a muted run's verdict kernels report (MCL_TEST_MUTE_VERDICT)
    int *inner_gate = nullptr;         // int32[2]: stop flag of the inner loop of the current phase (mcl_options.inner_tol)
    double *inner_part = nullptr;      // [max tiles / rows (x 16 in the exact-products mode)] per-workgroup ||x - x_old||^2 of the last solve
    double *wide_tab = nullptr;        // exact-products mode: [16 max tiles, DIAG_COLS] the sums of the test from the fp64 state (wide.hip)
    int *gate = nullptr;               // int32[4]: {stopped, stop_it, code, ticket of the verdict launch}
    const int *gate_active = nullptr;  // == gate while a gated run is enqueueing (copied into ModeView / RegSet), else NULL
    double *stop_state = nullptr;      // fp64[4]: {last computed loss, ...}
    double h_stop_init[1] = {0.0};     // host source of the asynchronous upload of the initial loss (must outlive it)
    mcl_stop_rule run_rule{};          // mcl_gate_begin .. mcl_gate_end: the rule mcl_verdict evaluates
    int *run_status_dev = nullptr;     // ... and the device view of the host's pinned status words
    bool diag_pending = false;
    bool diag_crossed_sweep = false;  // the pending deferral has already survived one sweep (it may not survive a second)
    DiagTables diag_pending_T{};
    double *diag_pending_out = nullptr;
    int diag_pending_incl = 1;
    // launch fusion of the A-phase finish (admm.hip: AFuse)
    bool a_rhs_from_M = false;       // k_A_finish_rows forms rhs_i from the sweep's M_bseg itself
    bool a_rhs_wide = false;         // ... with one workgroup (four waves) per slab: k_A_finish_rows_wide
    bool a_rhs_pairs = false;        // ... or two slabs per workgroup (one system + one streaming wave each)

    double *gl2_T = nullptr;    // [max rows, r] fp64: U^T Y of the GeneralizedL2 prox / U^T F of its value
    double *colsq = nullptr;    // [max(I,1), r]   per-slab column sums of squares (L2Ball)
    double *uni_f64 = nullptr;  // unimodal regression scratch: 10 fp64 arrays of (rows + slabs) * r
    int uni_attr_set[5] = {0, 0, 0, 0, 0};  // dynamic-LDS attribute of the multi-wave unimodal kernels on this context's device
    float *uni_sink = nullptr;  // two floats per lane of the unimodal kernels: where predicated-off stores of the emit loops go
    double *pf2_S = nullptr;    // [I, r, r]  Y_i^T Y_i (fp64)
    float *pf2_T = nullptr;     // [I, r, r]  P_i = Y_i T_i
    double *pf2_T64 = nullptr;  // [I, r, r]  the same before its rounding (fp64 row passes: rows64)
    bool rows64 = false;        // rank <= 16 and a PARAFAC2 member on mode 1: the fused B-mode row passes compute in fp64
    double *pf2_acc = nullptr;  // [I, r*r + 1] per-slab rho_i P_i^T Y_i | rho_i
    float *pf2_xmin = nullptr;  // [I] 1 / ||(G_i / tr)^-1/2||_F of the last Newton-Schulz run: starting estimate of the next
    int *pf2_status = nullptr;  // [I] <= 0: Newton-Schulz converged, 1: redo with the Jacobi kernel, 2: redo from Y Delta^T (k_pf2_polar_qr)
    double *pf2_qr = nullptr;   // [N * r] fp64: Y_i Delta^T of the slabs k_pf2_polar_qr takes
    float *pf2_red = nullptr;   // [r*r + 1]  sum over this context's slabs (all-reduced by a multi-GPU host)
    std::vector<int> h_row_ptr32, h_ext;
    bool e1_from_raw_gram = false;  // BtB buffer holds B_i^T B_i (true) or Q_i = B_i^T B_i o CtC (false)
    int n_part = 0;

    // validity of cached by-products
    bool xc_valid = false;      // XC == X @ C for the current C
    bool ctc_valid = false;     // CtC == C^T C for the current C
    int ctc_parts = 0;          // > 0: ... as that many partial blocks in CtCpart, not yet folded (k_C_finish_multi)
    double *CtCpart = nullptr;  // [16, 16, 16] fp64
    bool cfrag_valid = false;   // Cfrag is the fragment image of the current C
    bool e1_valid = false;      // e1/rhsA/BtB consistent with the current factors (A-phase just ran)
    bool diag_valid[3] = {false, false, false};  // per-mode diag tables consistent with factors/aux
    bool diagA_from_rows = false;
    bool xsq_valid = false;
    bool b_systems_valid = false;  // rhoB/LinvB already hold the systems of the coming B-phase (built by A-finish)
    int diag_rows[3] = {0, 0, 0};  // rows currently valid in diagA_row / diagB_tile / diagC_tile
    bool b_begun = false;
    bool stack_fused = false;  // generic inner loop: statistics kernels only, then one fused prox + dual row pass
    bool step_fuse = false, step_stats = false;  // the same two decisions for the current mcl_B_solve .. mcl_B_prox_* round
    unsigned step_done_mask = 0;                 // penalties finished in this round
    bool b_finish_pending = false;  // step API: the fused prox + dual pass of the last inner iteration has not been issued yet
    bool pf2_delta_fused = false;  // single-process inner loop: k_pf2_sum_delta instead of k_pf2_sum / (all-reduce) / k_pf2_delta
    bool stats_in_solve = false;   // ... and the B-mode statistics (PARAFAC2 Gram, L2-ball column norms) already came out of
                                   // the solve pass (k_rows_solve_stats + k_stats_reduce)
    std::vector<int> h_slab_tile_ptr;  // first B tile of every slab
    int *slab_tile_ptr = nullptr;      // int32[I+1]
    double *stat_gram = nullptr;       // [tilesB, (16 NB)^2]  per-tile Y^T Y, Y = B + U_pf2 (fp64 MFMA result order)
    double *stat_colsq = nullptr;      // [tilesB, MCL_MAX_REGS, r]  per-tile column sums of squares of (B + U_k)

    std::string variant[MCL_PROF_SLOTS];  // kernel last launched in every role (enum mcl_profile_slot)

    // optional HIP-event timing of the launch sites (enum mcl_profile_slot)
    int prof_capacity = 0;
    std::vector<hipEvent_t> prof_ev[MCL_PROF_SLOTS];
    int prof_used[MCL_PROF_SLOTS] = {};
    int prof_stride = 1;                  // record every prof_stride-th launch of a slot (an event pair costs ~10 us of
    int prof_seen[MCL_PROF_SLOTS] = {};   // dispatch gaps around the kernel, so timed loops sample instead)
    int64_t prof_launches[MCL_PROF_SLOTS] = {};  // launches seen per slot since mcl_profile_enable / _set_stride
    double prof_overhead_ms = 0.0;        // elapsed time of an EMPTY event pair on this stream (mcl_profile_enable)
    bool prof_nested = false;             // a ProfScope is open: launch sites inside it belong to its slot
};

// RAII helper: records start/stop events around a launch site when profiling is enabled.  Scopes do not nest: a launcher
// that calls another launcher (e.g. the C-phase finish falling back to the row kernels) keeps everything in ITS slot.
struct ProfScope {
    mcl_context *c;
    int slot;
    bool on, outer;
    ProfScope(mcl_context *ctx, int s) : c(ctx), slot(s), on(false), outer(!ctx->prof_nested) {
        if (!outer || c->prof_capacity <= 0) return;
        c->prof_nested = true;
        c->prof_launches[slot] += 1;
        if (c->prof_used[slot] < c->prof_capacity && (c->prof_seen[slot]++ % c->prof_stride) == 0) {
            on = true;
            (void)hipEventRecord(c->prof_ev[slot][2 * c->prof_used[slot]], c->stream);
        }
    }
    ~ProfScope() {
        if (on) {
            (void)hipEventRecord(c->prof_ev[slot][2 * c->prof_used[slot] + 1], c->stream);
            c->prof_used[slot] += 1;
        }
        if (outer && c->prof_capacity > 0) c->prof_nested = false;
    }
};

// Chunk count (64 columns each) of the X C kernel: 2 or 4 when the C fragments fit in registers (KCT template
// argument), otherwise a multiple of 4 (runtime loop; Cfrag is zero-padded).
static inline int mcl_xc_chunks(const mcl_context *c, int *kct) {
    const int raw = (int)((c->K + 63) / 64);
    const int budget = 4 / c->NB;  // chunks whose fragments fit in 64 VGPRs
    int kc, t;
    if (raw <= 2 && budget >= 2) kc = 2, t = 2;
    else if (raw <= 4 && budget >= 4) kc = 4, t = 4;
    else kc = (raw + 3) & ~3, t = 0;
    if (kct) *kct = t;
    return kc;
}

static inline int mcl_pad_rank(int r) { return r <= 4 ? 4 : r <= 8 ? 8 : r <= 16 ? 16 : r <= 32 ? 32 : 64; }

// ---- launchers implemented in contract.hip ---------------------------------------------------------
int mcl_launch_build_cfrag(mcl_context *c);
int mcl_launch_contract_xc(mcl_context *c);                      // XC = X @ C
int mcl_launch_contract_xt(mcl_context *c);                      // partials of [G | R]
int mcl_launch_reduce_partials(mcl_context *c);                  // GR = sum of partials
int mcl_launch_slab_gram(mcl_context *c);                        // rhsA, BtB from B and XC
int mcl_contract_n_partials(const mcl_context *c);

// ---- launchers implemented in sweep.hip --------------------------------------------------------------
int mcl_sweep_KS(const mcl_context *c);          // 256-column super-chunks per tile row
int mcl_sweep_KC(const mcl_context *c);          // 64-column chunks per tile row (2: the half-width kernels for K <= 128)
// chunks (64 columns each) of the shared fragment image of C: enough for the X C kernels and, when planned, the sweep
static inline int mcl_cfrag_chunks(const mcl_context *c) {
    const int xc = mcl_xc_chunks(c, nullptr);
    return c->sweep_planned ? std::max(xc, mcl_sweep_KC(c)) : xc;
}
bool mcl_sweep_shape_ok(const mcl_context *c);   // shape has a k_sweep instantiation (decides the workspace plan)
bool mcl_sweep_eligible(const mcl_context *c);   // ... and the current penalties / options / pointers allow it
int mcl_launch_sweep(mcl_context *c);            // B-phase + per-bseg X^T B, B^T B in one pass over X
int mcl_launch_reduce_weighted(mcl_context *c);  // GR = sum of the sweep blocks' a-weighted partials
int mcl_launch_A_rhs_from_M(mcl_context *c);     // seg_rhs[bseg] = coldot(M_bseg, C)

// ---- launchers implemented in admm.hip ---------------------------------------------------------------
int mcl_launch_ctc(mcl_context *c);
int mcl_launch_ctc_fold(mcl_context *c);
int mcl_launch_B_rho(mcl_context *c);
int mcl_launch_B_systems(mcl_context *c);
int mcl_launch_B_solve_f64(mcl_context *c);                      // penalty-free B: B_i = ((X_i C) o a_i) L_i^-1 in fp64
int mcl_launch_rows_fused(mcl_context *c, int mode);             // fused inner ADMM loop, row-separable penalties
int mcl_launch_rows_solve(mcl_context *c, int mode, double *change_part = nullptr);
int mcl_launch_inner_check(mcl_context *c, int mode, bool begin);  // generic.hip: the inner stopping test on the device
int mcl_launch_rows_prox(mcl_context *c, int mode, int k);       // generic prox step of penalty k (local part)
int mcl_launch_rows_prox_finish(mcl_context *c, int mode, int k);
int mcl_launch_C_prepare(mcl_context *c);
int mcl_launch_C_solve_f64(mcl_context *c);                      // penalty-free C: C = R G^-1 in fp64
int mcl_launch_C_finish_fused(mcl_context *c);
int mcl_launch_A_rho(mcl_context *c);
int mcl_launch_A_finish(mcl_context *c, bool fused_inner);
int mcl_launch_A_rows_solve(mcl_context *c, double *change_part = nullptr);
int mcl_launch_A_e1(mcl_context *c, bool btb_is_q);
int mcl_launch_rows_diag(mcl_context *c, int mode);
int mcl_launch_diag_final(mcl_context *c, double *out, int include_replicated, bool a_from_rows);
DiagTables mcl_diag_tables(const mcl_context *c, bool a_from_rows);  // the tables as they stand now
int mcl_launch_diag_tables(mcl_context *c, const DiagTables &T, double *out, int include_replicated);
int mcl_launch_diag_verdict(mcl_context *c, double *out, const mcl_stop_rule *rule, int it, double *verdict_row,
                            int *status_dev);  // table reduction + the stopping test of mcl_run
int mcl_launch_verdict(mcl_context *c, const double *vec, const mcl_stop_rule *rule, int it, double *verdict_row,
                       int *status_dev);  // the stopping test on an already reduced vector (sharded loop)
int mcl_launch_x_sq(mcl_context *c);
bool mcl_mode_is_row_separable(const mcl_context *c, int mode);
bool mcl_stack_can_fuse(const mcl_context *c, int mode);          // generic.hip
bool mcl_rows64(const mcl_context *c);                            // generic.hip
int mcl_launch_rows_finish_fused(mcl_context *c, int mode, bool want_diag);  // generic.hip
bool mcl_stats_can_ride_in_solve(const mcl_context *c, int mode);  // generic.hip
bool mcl_stats_reduce_in_algebra(const mcl_context *c);            // generic.hip
int mcl_launch_rows_finish_solve_stats(mcl_context *c);           // generic.hip: finish of iteration t + solve / stats of t + 1
int mcl_launch_rows_solve_stats(mcl_context *c);                  // generic.hip: B solve + per-tile statistics + reduce
bool mcl_wide_applies(const mcl_context *c, int mode);          // wide.hip: the fp64 inner loop of small problems takes the mode
int mcl_wide_phase(mcl_context *c, int mode);                    // wide.hip
bool mcl_exact_mode(const mcl_context *c);                      // contract.hip
int mcl_launch_exact_xc(mcl_context *c);                         // contract.hip: XC64 (+ its fp32 image) = X C, exact products
int mcl_launch_exact_gr(mcl_context *c);                         // contract.hip: GR = [G | R] of this rank's slabs, exact products
int mcl_launch_unimodal(mcl_context *c, const int *ext, int n_slabs, float *F, const RegSet &rs, int mode, int k);  // unimodal.hip
int mcl_launch_gl2_value(mcl_context *c, int mode, int k, double *out);  // generic.hip: sum over slabs of trace(F^T M F)
int mcl_try_contract_xc_lds(mcl_context *c, int gram);  // xclds.hip: X C with the fragment image of C resident in LDS
int mcl_try_rows_chain_mid(mcl_context *c, const ModeView &mv, const float *rhs, bool vec, bool rows64);  // rowchain.hip
int mcl_try_rows_chain_first(mcl_context *c, const ModeView &mv, const float *rhs, bool vec, bool rows64);
int mcl_try_rows_chain_last(mcl_context *c, const ModeView &mv, bool vec, bool rows64, double *diag, int want_diag);
int mcl_launch_pf2_cond_track(mcl_context *c);                         // cond.hip: monitor slot 3 <- worst polar-factor conditioning of the last PARAFAC2 inner iteration
int64_t mcl_cond_part_doubles(const mcl_context *c);                    // cond.hip
int mcl_launch_cond_probe(mcl_context *c, int want, double *out, bool accumulate = false);  // cond.hip: kappa of the penalty-free modes' systems -> out[3]
