// C ABI of libmatcouply_hip.so: context, workspace carve-up and the orchestration of the kernels per phase.
// One function per reference entry point (see include/matcouply_hip.h for the file:line each one replaces).
#include <time.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>

#include "mcl_internal.h"

int mcl_launch_generic_prox_local(mcl_context *c, int mode, int k);   // generic.hip
int mcl_launch_generic_prox_finish(mcl_context *c, int mode, int k);  // generic.hip
int mcl_rows_fused_dispatch(mcl_context *c, int mode, double *diag);  // admm.hip

static std::string g_create_error;

// the smallest kernel there is: what mcl_profile_enable calibrates the cost of an event pair on
__global__ void k_prof_nop(int *p) {
    if (threadIdx.x == 0) p[0] = 0;
}

namespace {

struct Bump {
    char *base;
    int64_t off = 0;
    template <typename T>
    T *take(int64_t count) {
        off = (off + 255) & ~int64_t(255);
        T *p = base ? reinterpret_cast<T *>(base + off) : nullptr;
        off += (int64_t)sizeof(T) * std::max<int64_t>(count, 1);
        return p;
    }
};

bool has_kind(const mcl_context *c, int kind) {
    for (int m = 0; m < 3; ++m)
        for (int k = 0; k < c->regs[m].n; ++k)
            if (c->regs[m].kind[k] == kind) return true;
    return false;
}

int xc_chunks(const mcl_context *c) { return mcl_xc_chunks(c, nullptr); }

// Assign every workspace pointer; returns the number of bytes needed.
int64_t plan(mcl_context *c, char *base) {
    Bump b{base};
    const int64_t I = c->I, K = c->K, N = c->N, r = c->r;
    const int64_t E = K * r + r * r;
    c->slab_of_row = b.take<int>(N);
    c->row_ptr_dev = b.take<int>(I + 1);
    auto tm = [&](TileMap &t, size_t n) {
        t.n_tiles = (int)n;
        t.slab = b.take<int>((int64_t)n);
        t.row0 = b.take<int>((int64_t)n);
        t.nrows = b.take<int>((int64_t)n);
    };
    tm(c->tilesB, c->h_tile_slab.size());
    tm(c->tilesC, c->h_ctile_slab.size());
    tm(c->tilesA, c->h_atile_slab.size());
    tm(c->segs, c->h_seg_slab.size());
    c->exact = mcl_exact_mode(c);
    c->sweep_planned = !c->exact && mcl_sweep_shape_ok(c);
    if (c->sweep_planned) {
        tm(c->bsegs, c->h_bseg_slab.size());
        c->slab_bseg_ptr = b.take<int>(I + 1);
        c->wave_bseg_ptr = b.take<int>((int64_t)c->h_wave_bseg_ptr.size());
        c->bseg_part = b.take<int>((int64_t)c->h_bseg_part.size());
        c->slab_part_ptr = b.take<int>(I + 1);
        c->Mpart = b.take<float>((int64_t)c->bsegs.n_tiles * mcl_sweep_KC(c) * 64 * 16 * c->NB);
        c->part_btb = b.take<double>((int64_t)c->bsegs.n_tiles * r * r);
        c->GRpart = b.take<float>((int64_t)c->bsegs.n_tiles * (256 * c->NB * c->NB + 16 * c->NB));
        c->CfragS = nullptr;  // aliases Cfrag (below)
        c->sweep_cycles = b.take<long long>((int64_t)2048 * 6);
    } else {
        c->bsegs = TileMap{};
        c->slab_bseg_ptr = nullptr, c->wave_bseg_ptr = nullptr, c->bseg_part = nullptr, c->slab_part_ptr = nullptr, c->Mpart = nullptr, c->part_btb = nullptr, c->GRpart = nullptr, c->CfragS = nullptr;
    }
    c->ext_A = b.take<int>(2);
    c->ext_C = b.take<int>(2);
    c->XC = b.take<float>(N * r);
    // ONE fragment image of C serves the X C kernels (first xc_chunks chunks) and the sweep (4 KS chunks): the chunk
    // index is the slowest one, so the shorter view is a prefix of the longer; every builder fills mcl_cfrag_chunks().
    c->Cfrag = b.take<float>((int64_t)mcl_cfrag_chunks(c) * 4 * c->NB * 256);
    if (c->sweep_planned) c->CfragS = c->Cfrag;
    c->CtC = b.take<float>(r * r);
    c->CtC64 = b.take<double>(r * r);
    c->CtCpart = b.take<double>(16 * 256);
    c->rhoB = b.take<float>(I);
    c->LinvB = b.take<float>(I * r * r);
    c->rows64 = c->NB == 1 && has_kind(c, MCL_PEN_PARAFAC2) && !c->sw.no_rows64;
    c->LinvB64 = (c->regs[1].n == 0 || c->rows64 || c->exact) ? b.take<double>(I * r * r) : nullptr;
    c->XC64 = (c->regs[1].n == 0 || c->exact) ? b.take<double>(N * r) : nullptr;
    c->rho_max = b.take<float>(2);
    c->partials = b.take<double>((int64_t)mcl_contract_n_partials(c) * E);
    c->GR = b.take<double>(E);
    c->exact_part = c->exact ? b.take<double>(std::max<int64_t>(1, (N + 255) / 256) * E) : nullptr;
    for (int m = 0; m < 3; ++m) {  // fp64 state of the inner loops of modes 1 / 2 in the exact-products mode (wide.hip)
        const int64_t rows_m = (m == 1) ? N : (m == 2 ? K : I);
        const bool wide = c->exact && c->regs[m].n > 0;
        c->wF[m] = wide ? b.take<double>(rows_m * r) : nullptr;
        for (int k = 0; k < MCL_MAX_REGS; ++k) {
            c->wZ[m][k] = (wide && k < c->regs[m].n) ? b.take<double>(rows_m * r) : nullptr;
            c->wU[m][k] = (wide && k < c->regs[m].n) ? b.take<double>(rows_m * r) : nullptr;
        }
    }
    c->wD = c->exact ? b.take<double>(r * r) : nullptr;
    c->LinvA64 = c->exact ? b.take<double>(I * r * r) : nullptr;
    c->rhsA64 = c->exact ? b.take<double>(I * r) : nullptr;
    c->Q64 = c->exact ? b.take<double>(I * r * r) : nullptr;
    c->GRf = b.take<float>(E);
    c->LinvC64 = b.take<double>(r * r);
    c->rhoC = b.take<float>(1);
    c->LinvC = b.take<float>(r * r);
    c->seg_rhs = b.take<double>(std::max<int64_t>(std::max(c->segs.n_tiles, c->bsegs.n_tiles), I) * r);
    c->seg_btb = b.take<double>(std::max<int64_t>(c->segs.n_tiles, I) * r * r);
    c->slab_seg_ptr = b.take<int>(I + 1);
    c->wave_seg_ptr = b.take<int>((int64_t)c->h_wave_seg_ptr.size());
    c->rhsA = b.take<float>(I * r);
    c->BtB = b.take<float>(I * r * r);
    c->rhoA = b.take<float>(I);
    c->LinvA = b.take<float>(I * r * r);
    c->e1 = b.take<double>(2 * I);
    c->diagA_row = b.take<double>(I * DIAG_COLS);
    c->diagA_tile = b.take<double>((int64_t)c->tilesA.n_tiles * DIAG_COLS);
    c->diagB_bufs[0] = b.take<double>((int64_t)c->tilesB.n_tiles * DIAG_COLS);
    c->diagB_bufs[1] = b.take<double>((int64_t)c->tilesB.n_tiles * DIAG_COLS);
    c->diagB_tile = c->diagB_bufs[c->diagB_parity];
    c->diagC_tile = b.take<double>((int64_t)c->tilesC.n_tiles * DIAG_COLS);
    c->diag_sums = b.take<double>(3 * DIAG_COLS + 2);
    c->xsq_part = b.take<double>(1024);
    c->x_sq = b.take<double>(1);
    c->cond_part = b.take<double>(mcl_cond_part_doubles(c));
    c->row_sink = b.take<float>(64 * 64 * 4);
    c->inner_gate = b.take<int>(2);
    {
        const int64_t max_tiles = std::max<int64_t>(std::max<int64_t>(c->tilesB.n_tiles, c->tilesC.n_tiles), std::max<int64_t>(c->tilesA.n_tiles, 1));
        c->inner_part = b.take<double>(std::max<int64_t>((c->exact ? 16 : 1) * max_tiles, std::max<int64_t>(I, 1)));
        c->wide_tab = c->exact ? b.take<double>(16 * max_tiles * DIAG_COLS) : nullptr;
    }
    c->gate = b.take<int>(4);
    c->mute_status = b.take<int>(4);
    c->stop_state = b.take<double>(4);

    // generic (non row-separable) path scratch
    const int64_t maxrows = std::max<int64_t>(N, std::max<int64_t>(I, K));
    c->colsq = b.take<double>(std::max<int64_t>(I, 1) * r * MCL_MAX_REGS);  // one table per penalty slot (fused stack)
    c->gl2_T = has_kind(c, MCL_PEN_GL2) ? b.take<double>(maxrows * r) : nullptr;
    if (has_kind(c, MCL_PEN_UNIMODAL)) {
        c->uni_f64 = b.take<double>(10 * (maxrows + std::max<int64_t>(I, 1)) * r);
        c->uni_sink = b.take<float>(2 * 64 * ((std::max<int64_t>(I, 1) * r + 63) / 64 + 3));  // (+3: the spare waves of the last four-wave workgroup)
    } else {
        c->uni_f64 = nullptr;
        c->uni_sink = nullptr;
    }
    // per-tile statistics of the B solve pass (fused generic stacks)
    c->slab_tile_ptr = b.take<int>(I + 1);
    c->stat_gram = has_kind(c, MCL_PEN_PARAFAC2) ? b.take<double>((int64_t)c->tilesB.n_tiles * 256 * c->NB * c->NB) : nullptr;
    c->stat_colsq = has_kind(c, MCL_PEN_L2BALL) ? b.take<double>((int64_t)c->tilesB.n_tiles * MCL_MAX_REGS * r) : nullptr;
    if (has_kind(c, MCL_PEN_PARAFAC2)) {
        c->pf2_S = b.take<double>(I * r * r);
        c->pf2_T = b.take<float>(I * r * r);
        c->pf2_T64 = (c->rows64 || c->exact) ? b.take<double>(I * r * r) : nullptr;
        c->pf2_acc = b.take<double>(I * (r * r + 1));
        c->pf2_red = b.take<float>(r * r + 1);
        c->pf2_status = b.take<int>(I);
        c->pf2_qr = b.take<double>(N * r);  // Y_i Delta^T of the slabs k_pf2_polar_qr takes
#if defined(MCL_NS_STAMPS) || defined(MCL_UNI_STAMPS)
        c->pf2_xmin = b.take<float>(I + 16 * I + 64);  // + 8 int64 stamps per slab (tools/ns_stamps.py)
#else
        c->pf2_xmin = b.take<float>(I);  // zeroed with the workspace: "no estimate yet"
#endif
    } else {
        c->pf2_status = nullptr;
        c->pf2_qr = nullptr;
        c->pf2_xmin = nullptr;
        c->pf2_S = nullptr, c->pf2_T = nullptr, c->pf2_acc = nullptr, c->pf2_red = nullptr, c->pf2_T64 = nullptr;
    }
    return (b.off + 255) & ~int64_t(255);
}

// one polite spin of a wait loop, whatever the host CPU is
static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    asm volatile("yield" ::: "memory");
#else
    asm volatile("" ::: "memory");
#endif
}

void read_switches(mcl_switches &w) {
#ifdef MCL_NO_ENV_SWITCHES  // release build (MCL_BUILD_DEFS=-DMCL_NO_ENV_SWITCHES): the environment is not consulted
    w = mcl_switches{};
    return;
#endif
    auto flag = [](const char *name) { return getenv(name) != nullptr; };
    auto num = [](const char *name, int dflt) {
        const char *e = getenv(name);
        return e ? atoi(e) : dflt;
    };
    w.no_sweep = flag("MCL_NO_SWEEP"), w.no_pass_chain = flag("MCL_NO_PASS_CHAIN");
    w.no_pf2_delta_fusion = flag("MCL_NO_PF2_DELTA_FUSION"), w.ns_plain = flag("MCL_NS_PLAIN");
    w.pf2_jacobi = flag("MCL_PF2_JACOBI"), w.no_stack_fusion = flag("MCL_NO_STACK_FUSION");
    w.no_solve_stats = flag("MCL_NO_SOLVE_STATS"), w.no_next_b = flag("MCL_NO_NEXT_B");
    w.no_fused_gram = flag("MCL_NO_FUSED_GRAM"), w.no_fused_c = flag("MCL_NO_FUSED_C");
    w.a_finish_cols = flag("MCL_A_FINISH_COLS"), w.xc_norow = flag("MCL_XC_NOROW");
    w.uni_noprune = flag("MCL_UNI_NOPRUNE"), w.stats_reduce = flag("MCL_STATS_REDUCE");
    w.no_rows64 = flag("MCL_NO_ROWS64"), w.no_uni_coop = flag("MCL_NO_UNI_COOP");
    w.no_a_fusion = flag("MCL_NO_A_FUSION"), w.no_a_wide = flag("MCL_NO_A_WIDE"), w.no_bseg_groups = flag("MCL_NO_BSEG_GROUPS"), w.no_sweep_half = flag("MCL_NO_SWEEP_HALF"), w.no_x_nt = flag("MCL_NO_X_NT"), w.x_nt_mb = num("MCL_X_NT_MB", 0), w.no_multi_c = flag("MCL_NO_MULTI_C"), w.no_diag_defer = flag("MCL_NO_DIAG_DEFER"), w.xc_depth1 = flag("MCL_XC_DEPTH1");
    w.seg_rows = num("MCL_SEG_ROWS", 0), w.bseg_rows = num("MCL_BSEG_ROWS", 0);
    w.xc_waves = num("MCL_XC_WAVES", 0), w.xt_waves = num("MCL_XT_WAVES", 0), w.sweep_waves = num("MCL_SWEEP_WAVES", 0);
    w.xc_dbg = num("MCL_XC_DBG", 0), w.xt_dbg = num("MCL_XT_DBG", 0), w.xt_depth = num("MCL_XT_DEPTH", 0);
    w.sweep_dbg = num("MCL_SWEEP_DBG", 0), w.reduce_el = num("MCL_REDUCE_EL", 0), w.uni_split = num("MCL_UNI_SPLIT", -1);
    w.exact = num("MCL_EXACT", -1);
    w.no_wide = flag("MCL_NO_WIDE");
    w.no_row_prefetch = flag("MCL_NO_ROW_PREFETCH");
    w.no_xc_lds = flag("MCL_NO_XC_LDS");
    w.uni_wpb = num("MCL_UNI_WPB", 0);
    w.xc_lds_depth = num("MCL_XC_LDS_DEPTH", 0);
    w.test_mute_verdict = flag("MCL_TEST_MUTE_VERDICT");  // test hook of the mcl_run watchdog, not a kernel form
    if (const char *e = getenv("MCL_RUN_SPINS")) w.run_spins = atol(e);  // operating parameter of mcl_run's wait (see there)
    if (const char *e = getenv("MCL_RUN_WATCHDOG_S")) {  // an operating parameter, not a kernel form (not listed by mcl_active_switches)
        const double v = atof(e);
        if (v > 0) w.run_watchdog_s = v;
    }
}

// names of the MCL_* switches present in the environment (what mcl_active_switches reports)
std::string switches_in_env() {
    static const char *names[] = {
        "MCL_NO_SWEEP", "MCL_NO_PASS_CHAIN", "MCL_NO_PF2_DELTA_FUSION", "MCL_NS_PLAIN", "MCL_PF2_JACOBI", "MCL_NO_STACK_FUSION",
        "MCL_NO_SOLVE_STATS", "MCL_NO_NEXT_B", "MCL_NO_FUSED_GRAM", "MCL_NO_FUSED_C", "MCL_A_FINISH_COLS", "MCL_XC_NOROW",
        "MCL_UNI_NOPRUNE", "MCL_STATS_REDUCE", "MCL_NO_ROWS64", "MCL_NO_UNI_COOP", "MCL_NO_A_FUSION", "MCL_NO_A_WIDE", "MCL_NO_BSEG_GROUPS",
        "MCL_NO_SWEEP_HALF", "MCL_NO_X_NT", "MCL_X_NT_MB", "MCL_NO_MULTI_C", "MCL_NO_DIAG_DEFER", "MCL_XC_DEPTH1", "MCL_SEG_ROWS",
        "MCL_BSEG_ROWS", "MCL_XC_WAVES", "MCL_XT_WAVES", "MCL_SWEEP_WAVES", "MCL_XC_DBG", "MCL_XT_DBG", "MCL_XT_DEPTH",
        "MCL_SWEEP_DBG", "MCL_REDUCE_EL", "MCL_UNI_SPLIT", "MCL_EXACT", "MCL_NO_WIDE", "MCL_NO_ROW_PREFETCH", "MCL_NO_XC_LDS", "MCL_XC_LDS_DEPTH", "MCL_UNI_WPB"};
    std::string out;
#ifdef MCL_NO_ENV_SWITCHES
    return out;
#endif
    for (const char *n : names)
        if (getenv(n) != nullptr) out += (out.empty() ? "" : " ") + std::string(n);
    return out;
}

int fail(mcl_context *c, const std::string &msg) {
    c->err = msg;
    return 1;
}

int ready_noflush(mcl_context *c) {
    if (c->failed) return fail(c, "the context is in a failed state (" + c->failed_why + "): destroy it");
    if (!c->has_problem) return fail(c, "mcl_set_problem has not been called");
    if (!c->has_factors) return fail(c, "mcl_set_factors has not been called");
    if (!c->has_workspace) return fail(c, "mcl_set_workspace has not been called");
    return 0;
}

// The step API defers the prox + dual row pass of a fused stack (see mcl_B_prox_finish): every entry point but
// mcl_B_solve - which merges it with its own pass - issues it first, so callers never observe the deferral.
int flush_B_finish(mcl_context *c) {
    if (!c->b_finish_pending) return 0;
    c->b_finish_pending = false;
    if (int rc = mcl_launch_rows_finish_fused(c, 1, true)) return rc;  // also leaves the mode's diagnostics table
    c->diag_valid[1] = true;
    return 0;
}

// A deferred diagnostics reduction that nothing has picked up yet is issued as a launch of its own: every entry point
// but the two a deferral is meant to cross (mcl_update_B on the sweep path, mcl_update_C_local) starts with this.
int flush_diag(mcl_context *c) {
    if (!c->diag_pending) return 0;
    c->diag_pending = false;
    c->diag_crossed_sweep = false;
    return mcl_launch_diag_tables(c, c->diag_pending_T, c->diag_pending_out, c->diag_pending_incl);
}

// Entry points that REPLACE what a deferred reduction refers to (workspace, problem, factors, penalty buffers) or end the
// context: the reduction is issued first, while the tables it recorded are still the old, valid ones - the header's
// contract is that `out` is written when any other entry point has been called.
int settle_deferred(mcl_context *c) {
    if (!c->diag_pending) return 0;
    if (c->has_problem && c->has_factors && c->has_workspace) return flush_diag(c);
    c->diag_pending = false;  // cannot happen (a deferral needs a complete context); never reduce through stale pointers
    c->diag_crossed_sweep = false;
    return 0;
}

int ready(mcl_context *c) {
    if (int rc = ready_noflush(c)) return rc;
    if (int rc = flush_diag(c)) return rc;
    return flush_B_finish(c);
}

// Step API contract (matcouply_hip.h): after mcl_B_solve EVERY penalty of a fusable stack must be stepped before the
// next solve / the end of the phase - the single prox + dual row pass is only issued once the stack is complete.  A host
// that skipped an index would otherwise leave stale aux / dual rows without any error.
int round_complete(mcl_context *c, const char *who) {
    if (!c->step_fuse) return 0;
    c->step_fuse = c->step_stats = false;  // report once; the host may restart the phase with mcl_B_begin
    return fail(c, std::string(who) + ": the previous inner iteration did not step every penalty of mode 1 "
                   "(mcl_B_prox_local / mcl_B_prox_finish for k = 0 .. n-1): its aux / dual rows were not updated");
}

int ensure_ctc(mcl_context *c) {
    if (!c->ctc_valid) {
        if (int rc = mcl_launch_ctc(c)) return rc;
        c->ctc_valid = true;
        c->ctc_parts = 0;
    } else if (c->ctc_parts > 0) {  // current, but still in the partial blocks of k_C_finish_multi
        if (int rc = mcl_launch_ctc_fold(c)) return rc;
    }
    return 0;
}

// mcl_condition_monitor: before a penalty-free mode's phase, the condition estimate of the system it is about to solve (from the
// factors as they are NOW: the phases are Gauss-Seidel, the A-phase of an iteration sees the B and C of the same iteration)
int monitor_condition(mcl_context *c, int mode) {
    if (!c->cond_monitor || !(c->cond_monitor_mask >> mode & 1) || c->regs[mode].n != 0) return 0;
    if (int rc = ensure_ctc(c)) return rc;
    return mcl_launch_cond_probe(c, 1 << mode, c->cond_monitor, true);
}

int ensure_xc(mcl_context *c) {
    if (!c->xc_valid && c->exact) {  // exact-products mode: fp64 sums of exact products, rounded once for the fp32 image
        if (int rc = mcl_launch_exact_xc(c)) return rc;
        c->xc_did_gram = false;
        c->xc_valid = true;
    }
    if (!c->xc_valid) {
        if (!c->cfrag_valid) {
            if (int rc = mcl_launch_build_cfrag(c)) return rc;
            c->cfrag_valid = true;
        }
        if (int rc = mcl_launch_contract_xc(c)) return rc;
        c->xc_valid = true;
    }
    return 0;
}

// fragment image of the current C for the sweep kernels
int ensure_cfrag_sweep(mcl_context *c) {  // CfragS aliases Cfrag: one image serves the X C kernels and the sweep
    if (!c->cfrag_valid) {
        if (int rc = mcl_launch_build_cfrag(c)) return rc;
        c->cfrag_valid = true;
    }
    return 0;
}

// generic inner loop of mode m: solve, then per penalty prox (+ reduction) and dual update
// The inner loop WITH the reference's inner stopping test (decomposition.py:90-117, mcl_options.inner_tol), on the device:
// one launch per step; after every inner iteration a one-workgroup kernel takes ||x - x_old||^2 (per-tile sums of the solve
// pass), ||x||^2 and the feasibility gaps (the per-tile diagnostics table) and sets the phase's stop flag - the launches of
// the remaining inner iterations, already enqueued, test it first and do nothing (the MCL_GATE pattern of mcl_run).
int checked_inner_loop(mcl_context *c, int mode) {
    const int n_it = c->opt.inner_n_iter_max;
    if (int rc = mcl_launch_inner_check(c, mode, true)) return rc;  // flag <- the run's stop flag (0 outside a gated run)
    const int *run_gate = c->gate_active, *reg_gate = c->regs[mode].gate;
    c->gate_active = c->inner_gate, c->regs[mode].gate = c->inner_gate;
    int rc = 0;
    for (int it = 0; it < n_it && rc == 0; ++it) {
        rc = (mode == 0) ? mcl_launch_A_rows_solve(c, c->inner_part) : mcl_launch_rows_solve(c, mode, c->inner_part);
        c->stack_fused = c->stats_in_solve = false;
        c->pf2_delta_fused = !c->sw.no_pf2_delta_fusion;
        for (int k = 0; k < c->regs[mode].n && rc == 0; ++k) {
            rc = mcl_launch_generic_prox_local(c, mode, k);
            if (rc == 0) rc = mcl_launch_generic_prox_finish(c, mode, k);
        }
        c->pf2_delta_fused = false;
        if (rc == 0) rc = mcl_launch_rows_diag(c, mode);
        if (rc == 0) rc = mcl_launch_inner_check(c, mode, false);
    }
    c->gate_active = run_gate, c->regs[mode].gate = reg_gate;
    c->diag_valid[mode] = rc == 0 && n_it > 0 && mode != 0;  // (mode 0: mcl_launch_A_e1 follows and owns its tables)
    return rc;
}

int generic_inner_loop(mcl_context *c, int mode) {
    if (c->opt.inner_tol > 0.0 && c->regs[mode].n > 0) return checked_inner_loop(c, mode);
    const int n_it = (c->regs[mode].n == 0) ? std::min(1, (int)c->opt.inner_n_iter_max) : c->opt.inner_n_iter_max;
    // single-process run of the whole stack: per-slab statistics first (Gram / polar factor / Delta, column norms),
    // then ONE row pass for every prox + dual step (the step API used by multi-GPU hosts keeps one pass per penalty)
    const bool fuse = mcl_stack_can_fuse(c, mode);
    const bool stats = fuse && mcl_stats_can_ride_in_solve(c, mode);
    // ... and with the statistics riding in the solve, the finish pass of iteration t also does the solve of t + 1
    int n_l2 = 0;
    for (int k = 0; k < c->regs[mode].n; ++k) n_l2 += c->regs[mode].kind[k] == MCL_PEN_L2BALL;
    const bool chain = stats && n_l2 <= 1 && !c->sw.no_pass_chain;  // the chained kernel carries one L2-ball slot
    for (int it = 0; it < n_it; ++it) {
        if (mode == 0) {
            if (int rc = mcl_launch_A_rows_solve(c)) return rc;
        } else if (stats) {
            if (!chain || it == 0)
                if (int rc = mcl_launch_rows_solve_stats(c)) return rc;
        } else {
            if (int rc = mcl_launch_rows_solve(c, mode)) return rc;
        }
        c->stack_fused = fuse;
        c->stats_in_solve = stats;
        c->pf2_delta_fused = !c->sw.no_pf2_delta_fusion;
        int rc = 0;
        // (Round 6, measured and dropped: the PARAFAC2 algebra of an inner iteration does not depend on the unimodal regressions of
        // the same iteration, and a launch of the regressions leaves 13-16 % of the device's wave slots idle - two rounds of long
        // waves, tools/uni_stamps.py.  Run on a low-priority stream of the library's own beside the regressions, the algebra's
        // waves do not wait for those gaps: the regressions take 10.3 instead of 7.6 ms, config 5 106 instead of 96 ms per iteration.)
        for (int k = 0; k < c->regs[mode].n && rc == 0; ++k) {
            rc = mcl_launch_generic_prox_local(c, mode, k);
            if (rc == 0) rc = mcl_launch_generic_prox_finish(c, mode, k);
        }
        c->stack_fused = c->stats_in_solve = c->pf2_delta_fused = false;
        if (rc) return rc;
        if (fuse) {
            if (chain && it + 1 < n_it) {
                if (int rc2 = mcl_launch_rows_finish_solve_stats(c)) return rc2;
            } else if (int rc2 = mcl_launch_rows_finish_fused(c, mode, it == n_it - 1)) {  // diagnostics: last pass only
                return rc2;
            }
        }
    }
    c->diag_valid[mode] = fuse && n_it > 0;  // the fused finish pass leaves the mode's diagnostics table current
    return 0;
}

}  // namespace

extern "C" {

int mcl_version(void) { return MCL_ABI_VERSION; }

const char *mcl_last_error(const mcl_context *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int mcl_create(mcl_context **out, int device, void *hip_stream) {
    if (!out) return 1;
    *out = nullptr;
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev <= 0) {
        g_create_error = std::string("no HIP device available: ") + hipGetErrorString(e);
        return 1;
    }
    if (device < 0 || device >= n_dev) {
        g_create_error = "device index out of range";
        return 1;
    }
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) {
        g_create_error = std::string("hipGetDeviceProperties: ") + hipGetErrorString(e);
        return 1;
    }
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        g_create_error = std::string("libmatcouply_hip is built for gfx950 (MI355X) only; device is ") + prop.gcnArchName;
        return 1;
    }
    e = hipSetDevice(device);
    if (e != hipSuccess) {
        g_create_error = std::string("hipSetDevice: ") + hipGetErrorString(e);
        return 1;
    }
    mcl_context *c = new mcl_context();
    c->device = device;
    c->stream = reinterpret_cast<hipStream_t>(hip_stream);
    c->opt.feasibility_penalty_scale = 1.0;
    c->opt.inner_n_iter_max = 5;
    read_switches(c->sw);
    c->active_switches = switches_in_env();
    *out = c;
    return 0;
}

void mcl_destroy(mcl_context *ctx) {
    if (!ctx) return;
    if (!ctx->failed) (void)settle_deferred(ctx);  // a deferred diagnostics vector is still delivered
    for (int s = 0; s < MCL_PROF_SLOTS; ++s)
        for (hipEvent_t e : ctx->prof_ev[s]) (void)hipEventDestroy(e);
    delete ctx;
}

int mcl_set_problem(mcl_context *c, const float *X, const int64_t *row_ptr, int64_t I, int64_t K, int32_t rank) {
    if (!c) return 1;
    if (I < 0 || K < 1) return fail(c, "mcl_set_problem: need I >= 0 and K >= 1");
    if (rank < 1 || rank > MCL_MAX_RANK) return fail(c, "mcl_set_problem: rank must be in [1, 64]");
    if (int rc = settle_deferred(c)) return rc;
    c->b_finish_pending = false;  // a deferred row pass of the step API refers to the buffers being replaced
    if (!row_ptr || row_ptr[0] != 0) return fail(c, "mcl_set_problem: row_ptr[0] must be 0");
    for (int64_t i = 0; i < I; ++i)
        if (row_ptr[i + 1] < row_ptr[i]) return fail(c, "mcl_set_problem: row_ptr must be non-decreasing");
    const int64_t N = row_ptr[I];
    if (N >= (int64_t(1) << 31) - 64) return fail(c, "mcl_set_problem: more than 2^31 packed rows are not supported");
    if (N > 0 && !X) return fail(c, "mcl_set_problem: X is NULL");
    c->X = X;
    c->row_ptr.assign(row_ptr, row_ptr + I + 1);
    c->I = I, c->K = K, c->N = N, c->r = rank;
    c->max_slab_rows = 0;
    for (int64_t i = 0; i < I; ++i) c->max_slab_rows = std::max<int64_t>(c->max_slab_rows, row_ptr[i + 1] - row_ptr[i]);
    c->RP = mcl_pad_rank(rank);
    c->NB = (rank + 15) / 16;
    if (c->NB == 3) c->NB = 4;
    c->sweep_kc = (K <= 128 && c->NB == 1 && !c->sw.no_sweep_half) ? 2 : 4 * (int)((K + 255) / 256);
    c->x_streams = !c->sw.no_x_nt && (double)N * (double)K * 4.0 > (c->sw.x_nt_mb > 0 ? c->sw.x_nt_mb : 256) * 1048576.0;
    c->h_slab_of_row.resize((size_t)N);
    c->h_tile_slab.clear(), c->h_tile_row0.clear(), c->h_tile_nrows.clear();
    c->h_seg_slab.clear(), c->h_seg_row0.clear(), c->h_seg_nrows.clear();
    c->h_slab_seg_ptr.assign((size_t)I + 1, 0);
    c->h_slab_tile_ptr.assign((size_t)I + 1, 0);
    // segment = work unit of the two X passes: <= MCL_SEG_ROWS rows of one slab (256 rows - 256 KB at K = 256 - amortise the
    // per-segment prologue, and the fp32 accumulation chains of the passes end with their segment).  Every wave of the
    // passes owns a contiguous range of segments (wave_seg_ptr) holding the same number of 16-row blocks: the passes
    // run one wave per SIMD side by side, so the longest wave IS the kernel's duration - with ragged slabs a fixed number
    // of segments per wave left the longest wave 32 % above the mean (config 4: 48 blocks vs 36.4).  Segments are
    // therefore also cut where a wave's quota ends.  Up to 1024 waves (256 CUs x 4: measured best); small problems (e.g.
    // the per-rank shard of an 8-GPU run) keep >= 512 waves of >= 1 block, as long as there are 8 blocks for each of
    // them (measured optimum on a 64 K-row shard: 512 waves of 128 rows).
    int64_t seg_rows = MCL_SEG_ROWS;
    if (c->sw.seg_rows > 0) seg_rows = std::max(16, (c->sw.seg_rows / 16) * 16);
    int64_t total_units = 0;  // 16-row blocks, a slab's last partial block counts as one
    for (int64_t i = 0; i < I; ++i) total_units += (row_ptr[i + 1] - row_ptr[i] + 15) / 16;
    int64_t target_waves = 1024;
    if (c->sw.xc_waves > 0) target_waves = c->sw.xc_waves;
    else if (c->sw.xt_waves > 0) target_waves = c->sw.xt_waves;
    int64_t n_waves = std::min<int64_t>(target_waves, std::max<int64_t>(std::min<int64_t>(512, total_units), total_units / 8));
    n_waves = std::max<int64_t>(n_waves, 1);
    const int64_t quota = std::max<int64_t>(1, (total_units + n_waves - 1) / n_waves);
    c->h_wave_seg_ptr.assign(1, 0);
    int64_t used = 0;
    for (int64_t i = 0; i < I; ++i) {
        c->h_slab_seg_ptr[(size_t)i] = (int)c->h_seg_slab.size();
        c->h_slab_tile_ptr[(size_t)i] = (int)c->h_tile_slab.size();
        for (int64_t j = row_ptr[i]; j < row_ptr[i + 1];) {
            const int64_t take = std::min<int64_t>(std::min<int64_t>(seg_rows, row_ptr[i + 1] - j), (quota - used) * 16);
            c->h_seg_slab.push_back((int)i);
            c->h_seg_row0.push_back((int)j);
            c->h_seg_nrows.push_back((int)take);
            j += take;
            used += (take + 15) / 16;
            if (used >= quota) {  // the wave is full
                c->h_wave_seg_ptr.push_back((int)c->h_seg_slab.size());
                used = 0;
            }
        }
        for (int64_t j = row_ptr[i]; j < row_ptr[i + 1]; ++j) c->h_slab_of_row[(size_t)j] = (int)i;
        for (int64_t j = row_ptr[i]; j < row_ptr[i + 1]; j += 64) {
            c->h_tile_slab.push_back((int)i);
            c->h_tile_row0.push_back((int)j);
            c->h_tile_nrows.push_back((int)std::min<int64_t>(64, row_ptr[i + 1] - j));
        }
    }
    if (used > 0) c->h_wave_seg_ptr.push_back((int)c->h_seg_slab.size());
    c->n_seg_waves = (int)c->h_wave_seg_ptr.size() - 1;
    c->h_slab_seg_ptr[(size_t)I] = (int)c->h_seg_slab.size();
    c->h_slab_tile_ptr[(size_t)I] = (int)c->h_tile_slab.size();
    // bsegs of the one-pass sweep: <= 512 rows of one slab per WAVE, shorter on small problems so that there are still
    // >= ~1024 of them (one per SIMD)
    int64_t bseg_rows = 512;
    while (bseg_rows > 64 && N / bseg_rows < 1024) bseg_rows /= 2;
    if (c->sw.bseg_rows > 0) bseg_rows = std::max(64, (c->sw.bseg_rows / 64) * 64);
    // The waves of the sweep (<= 1024, one per SIMD) own contiguous bseg ranges with (nearly) the same number of 16-row
    // blocks, like the segments above; a bseg is cut at a wave's quota only when the overshoot would exceed 2 blocks
    // (every cut costs a 2 x 16 KB partial at K = 256).  Problems with at most one bseg per wave are left alone.
    int64_t n_uncut = 0;
    for (int64_t i = 0; i < I; ++i) n_uncut += (row_ptr[i + 1] - row_ptr[i] + bseg_rows - 1) / bseg_rows;
    const int64_t sweep_waves = c->sw.sweep_waves > 0 ? c->sw.sweep_waves : 1024;
    const bool one_each = n_uncut <= sweep_waves;
    const int64_t bquota = one_each ? 0 : (total_units + sweep_waves - 1) / sweep_waves, btol = 2;
    c->h_bseg_slab.clear(), c->h_bseg_row0.clear(), c->h_bseg_nrows.clear();
    c->h_slab_bseg_ptr.assign((size_t)I + 1, 0);
    c->h_wave_bseg_ptr.assign(1, 0);
    int64_t bused = 0;
    for (int64_t i = 0; i < I; ++i) {
        c->h_slab_bseg_ptr[(size_t)i] = (int)c->h_bseg_slab.size();
        for (int64_t j = row_ptr[i]; j < row_ptr[i + 1];) {
            int64_t take = std::min<int64_t>(bseg_rows, row_ptr[i + 1] - j);
            if (!one_each && bused + (take + 15) / 16 > bquota + btol) take = (bquota - bused) * 16;  // >= 16: bused < bquota
            c->h_bseg_slab.push_back((int)i);
            c->h_bseg_row0.push_back((int)j);
            c->h_bseg_nrows.push_back((int)take);
            j += take;
            bused += (take + 15) / 16;
            if (one_each || bused >= bquota) {
                c->h_wave_bseg_ptr.push_back((int)c->h_bseg_slab.size());
                bused = 0;
            }
        }
    }
    if (bused > 0) c->h_wave_bseg_ptr.push_back((int)c->h_bseg_slab.size());
    c->n_bseg_waves = (int)c->h_wave_bseg_ptr.size() - 1;
    c->h_slab_bseg_ptr[(size_t)I] = (int)c->h_bseg_slab.size();
    // Partials of the sweep (M = X^T B, B^T B, the a-weighted Gram: one 16 KB image at K = 256).  One bseg per wave and
    // short bsegs (small problems: per-rank shards, config 2): the four waves of a workgroup whose bsegs lie in the same
    // slab add their accumulators in LDS and write ONE partial (k_sweep<.., GRP>: K <= 256, rank <= 16).
    {
        const int nbs = (int)c->h_bseg_slab.size();
        const bool can_group = one_each && K <= 256 && c->NB == 1 && bseg_rows < 512 && !c->sw.no_bseg_groups &&
                               c->sw.sweep_dbg == 0;
        c->h_bseg_part.assign((size_t)nbs, 0);
        int parts = 0;
        // encoding of bseg_part: bits 0-27 the partial, bit 28 "this workgroup runs the cooperative flush", bits 29-30 log2 of
        // the group size of the wave (groups: all four bsegs of the workgroup, or an aligned pair of them, in one slab)
        for (int b0 = 0; b0 < nbs; b0 += 4) {
            const bool full = can_group && b0 + 3 < nbs;
            auto same = [&](int x, int y) { return c->h_bseg_slab[(size_t)x] == c->h_bseg_slab[(size_t)y]; };
            const bool g4 = full && same(b0, b0 + 3);
            const bool p0 = full && !g4 && same(b0, b0 + 1), p1 = full && !g4 && same(b0 + 2, b0 + 3);
            const int coop = (g4 || p0 || p1) ? (1 << 28) : 0;
            if (g4) {
                for (int b = b0; b < b0 + 4; ++b) c->h_bseg_part[(size_t)b] = parts | coop | (2 << 29);
                ++parts;
            } else {
                for (int h = 0; h < 2; ++h) {
                    const bool pair = h == 0 ? p0 : p1;
                    for (int b = b0 + 2 * h; b < std::min(b0 + 2 * h + 2, nbs); ++b)
                        c->h_bseg_part[(size_t)b] = pair ? (parts | coop | (1 << 29)) : ((parts++) | coop);
                    if (pair) ++parts;
                }
            }
        }
        c->n_parts = parts;
        c->h_slab_part_ptr.assign((size_t)I + 1, parts);
        for (int64_t i = I - 1; i >= 0; --i) {
            const int b = c->h_slab_bseg_ptr[(size_t)i];
            c->h_slab_part_ptr[(size_t)i] = (b < c->h_slab_bseg_ptr[(size_t)i + 1]) ? (c->h_bseg_part[(size_t)b] & 0x0fffffff)
                                                                                    : c->h_slab_part_ptr[(size_t)i + 1];
        }
    }
    auto single = [](int64_t rows, std::vector<int> &s, std::vector<int> &r0, std::vector<int> &nr) {
        s.clear(), r0.clear(), nr.clear();
        for (int64_t j = 0; j < rows; j += 64) {
            s.push_back(0);
            r0.push_back((int)j);
            nr.push_back((int)std::min<int64_t>(64, rows - j));
        }
    };
    single(K, c->h_ctile_slab, c->h_ctile_row0, c->h_ctile_nrows);
    single(I, c->h_atile_slab, c->h_atile_row0, c->h_atile_nrows);
    c->has_problem = true;
    c->has_workspace = false;
    c->xc_valid = c->ctc_valid = c->e1_valid = c->xsq_valid = false;
    c->mseg_valid = c->grpart_valid = false;
    c->diag_valid[0] = c->diag_valid[1] = c->diag_valid[2] = false;
    return 0;
}

int mcl_set_options(mcl_context *c, const mcl_options *opt) {
    if (!c || !opt) return 1;
    if (opt->inner_n_iter_max < 0) return fail(c, "mcl_set_options: inner_n_iter_max must be >= 0");
    if (opt->exact_products < 0 || opt->exact_products > 2) return fail(c, "mcl_set_options: exact_products must be 0, 1 or 2");
    if (!(opt->inner_tol >= 0.0)) return fail(c, "mcl_set_options: inner_tol must be >= 0 (0: not set)");
    if (c->has_workspace && opt->exact_products != c->opt.exact_products)
        c->has_workspace = false;  // the carve-up depends on the mode: the workspace has to be installed again
    c->opt = *opt;
    c->b_systems_valid = false;
    return 0;
}

int mcl_set_factors(mcl_context *c, float *A, float *B, float *C) {
    if (!c) return 1;
    if (!c->has_problem) return fail(c, "mcl_set_factors: call mcl_set_problem first");
    if ((c->I > 0 && !A) || (c->N > 0 && !B) || !C) return fail(c, "mcl_set_factors: NULL factor pointer");
    if (int rc = settle_deferred(c)) return rc;
    c->b_finish_pending = false;  // a deferred row pass of the step API refers to the buffers being replaced
    c->A = A, c->B = B, c->C = C;
    c->has_factors = true;
    c->b_systems_valid = false;
    c->cfrag_valid = false;
    c->xc_valid = c->ctc_valid = c->e1_valid = false;
    c->mseg_valid = c->grpart_valid = false;
    c->diag_valid[0] = c->diag_valid[1] = c->diag_valid[2] = false;
    return 0;
}

int mcl_set_penalties(mcl_context *c, int32_t mode, int32_t n, const mcl_penalty_desc *descs) {
    if (!c) return 1;
    if (mode < 0 || mode > 2) return fail(c, "mcl_set_penalties: mode must be 0, 1 or 2");
    if (n < 0 || n > MCL_MAX_REGS) return fail(c, "mcl_set_penalties: at most 4 penalties per mode");
    if (int rc = settle_deferred(c)) return rc;
    c->b_finish_pending = false;  // a deferred row pass of the step API refers to the buffers being replaced
    RegSet rs{};
    rs.n = n;
    for (int k = 0; k < n; ++k) {
        const mcl_penalty_desc &d = descs[k];
        if (d.kind < MCL_PEN_NN || d.kind > MCL_PEN_SIMPLEX) return fail(c, "mcl_set_penalties: unknown penalty kind");
        if (d.kind == MCL_PEN_GL2) {
            if (!d.matrix || d.matrix_rows < 1) return fail(c, "mcl_set_penalties: GeneralizedL2 needs its eigen-decomposition (matrix, matrix_rows)");
            if (!c->has_problem) return fail(c, "mcl_set_penalties: call mcl_set_problem first");
            bool rows_ok = (mode == 0) ? d.matrix_rows == c->I : (mode == 2 ? d.matrix_rows == c->K : true);
            for (int64_t i = 0; mode == 1 && i < c->I; ++i) rows_ok = rows_ok && (c->row_ptr[i + 1] - c->row_ptr[i] == d.matrix_rows);
            if (!rows_ok) return fail(c, "mcl_set_penalties: GeneralizedL2: every matrix of the mode must have matrix_rows rows");
        }
        if (d.kind == MCL_PEN_TV && (d.p0 <= 0 || d.p1 < 0))
            return fail(c, "mcl_set_penalties: TV strength must be positive and its L1 strength non-negative");
        if (d.kind == MCL_PEN_PARAFAC2 && mode != 1)
            return fail(c, "mcl_set_penalties: PARAFAC2 constraint can only be imposed with mode=1");
        if (d.kind == MCL_PEN_PARAFAC2 && !d.aux2) return fail(c, "mcl_set_penalties: PARAFAC2 needs the coordinate matrix");
        if (d.kind == MCL_PEN_L1 && d.p0 < 0) return fail(c, "mcl_set_penalties: L1 strength must be non-negative");
        if (d.kind == MCL_PEN_L2BALL && d.p0 <= 0) return fail(c, "mcl_set_penalties: L2 ball bound must be positive");
        if (!d.aux || !d.dual) return fail(c, "mcl_set_penalties: NULL aux/dual pointer");
        rs.kind[k] = d.kind;
        rs.nonneg[k] = d.non_negativity;
        rs.p0[k] = (float)d.p0;
        rs.p1[k] = (float)d.p1;
        rs.p0d[k] = d.p0;
        rs.p1d[k] = d.p1;
        rs.mat[k] = d.kind == MCL_PEN_GL2 ? d.matrix : nullptr;
        rs.mat_rows[k] = d.kind == MCL_PEN_GL2 ? (int)d.matrix_rows : 0;
        rs.aux[k] = d.aux;
        rs.dual[k] = d.dual;
        rs.aux2[k] = d.aux2;
    }
    c->regs[mode] = rs;
    c->b_systems_valid = false;
    c->has_workspace = false;  // scratch requirements may have changed
    c->diag_valid[mode] = false;
    c->e1_valid = false;
    return 0;
}

int64_t mcl_workspace_bytes(mcl_context *c) {
    if (!c || !c->has_problem) return -1;
    const int64_t need = plan(c, nullptr);
    if (c->has_workspace) plan(c, c->ws);  // a size query must leave an installed workspace as it was
    return need;
}

int mcl_set_workspace(mcl_context *c, void *workspace, int64_t bytes) {
    if (!c) return 1;
    if (!c->has_problem) return fail(c, "mcl_set_workspace: call mcl_set_problem first");
    if (int rc = settle_deferred(c)) return rc;  // reduces the tables of the OLD workspace before it is re-planned / zeroed
    const bool had = c->has_workspace;
    c->has_workspace = false;  // size the request with the mode a fresh installation would get
    const int64_t need = plan(c, nullptr);
    c->has_workspace = had;
    if (!workspace || bytes < need) {
        if (c->has_workspace) plan(c, c->ws);  // a refused call leaves the installed workspace as it was
        return fail(c, "mcl_set_workspace: workspace too small");
    }
    c->b_finish_pending = false;  // a deferred row pass of the step API refers to the buffers being replaced
    if (reinterpret_cast<uintptr_t>(workspace) & 255) return fail(c, "mcl_set_workspace: workspace must be 256-byte aligned");
    c->ws = static_cast<char *>(workspace);
    c->ws_bytes = bytes;
    c->has_workspace = false;  // the arithmetic mode (mcl_exact_mode) is decided anew for this installation ...
    plan(c, c->ws);            // ... and stays what it is until the next one
    hipStream_t s = c->stream;
    MCL_CHECK_HIP(c, hipMemsetAsync(c->ws, 0, (size_t)need, s));
    auto up = [&](int *dst, const std::vector<int> &src) -> hipError_t {
        if (src.empty()) return hipSuccess;
        return hipMemcpyAsync(dst, src.data(), src.size() * sizeof(int), hipMemcpyHostToDevice, s);
    };
    MCL_CHECK_HIP(c, up(c->slab_of_row, c->h_slab_of_row));
    c->h_row_ptr32.assign(c->row_ptr.begin(), c->row_ptr.end());
    MCL_CHECK_HIP(c, up(c->row_ptr_dev, c->h_row_ptr32));
    MCL_CHECK_HIP(c, up(c->tilesB.slab, c->h_tile_slab));
    MCL_CHECK_HIP(c, up(c->tilesB.row0, c->h_tile_row0));
    MCL_CHECK_HIP(c, up(c->tilesB.nrows, c->h_tile_nrows));
    MCL_CHECK_HIP(c, up(c->tilesC.slab, c->h_ctile_slab));
    MCL_CHECK_HIP(c, up(c->tilesC.row0, c->h_ctile_row0));
    MCL_CHECK_HIP(c, up(c->tilesC.nrows, c->h_ctile_nrows));
    MCL_CHECK_HIP(c, up(c->tilesA.slab, c->h_atile_slab));
    MCL_CHECK_HIP(c, up(c->tilesA.row0, c->h_atile_row0));
    MCL_CHECK_HIP(c, up(c->tilesA.nrows, c->h_atile_nrows));
    MCL_CHECK_HIP(c, up(c->segs.slab, c->h_seg_slab));
    MCL_CHECK_HIP(c, up(c->segs.row0, c->h_seg_row0));
    MCL_CHECK_HIP(c, up(c->segs.nrows, c->h_seg_nrows));
    MCL_CHECK_HIP(c, up(c->slab_seg_ptr, c->h_slab_seg_ptr));
    MCL_CHECK_HIP(c, up(c->wave_seg_ptr, c->h_wave_seg_ptr));
    MCL_CHECK_HIP(c, up(c->slab_tile_ptr, c->h_slab_tile_ptr));
    if (c->sweep_planned) {
        MCL_CHECK_HIP(c, up(c->bsegs.slab, c->h_bseg_slab));
        MCL_CHECK_HIP(c, up(c->bsegs.row0, c->h_bseg_row0));
        MCL_CHECK_HIP(c, up(c->bsegs.nrows, c->h_bseg_nrows));
        MCL_CHECK_HIP(c, up(c->slab_bseg_ptr, c->h_slab_bseg_ptr));
        MCL_CHECK_HIP(c, up(c->wave_bseg_ptr, c->h_wave_bseg_ptr));
        MCL_CHECK_HIP(c, up(c->bseg_part, c->h_bseg_part));
        MCL_CHECK_HIP(c, up(c->slab_part_ptr, c->h_slab_part_ptr));
    }
    c->h_ext = {0, (int)c->I, 0, (int)c->K};
    MCL_CHECK_HIP(c, hipMemcpyAsync(c->ext_A, c->h_ext.data(), 2 * sizeof(int), hipMemcpyHostToDevice, s));
    MCL_CHECK_HIP(c, hipMemcpyAsync(c->ext_C, c->h_ext.data() + 2, 2 * sizeof(int), hipMemcpyHostToDevice, s));
    // the host vectors must outlive the async copies: they are members of the context
    c->has_workspace = true;
    c->cfrag_valid = false;
    c->mseg_valid = c->grpart_valid = false;
    c->xc_valid = c->ctc_valid = c->e1_valid = c->xsq_valid = false;
    c->diag_valid[0] = c->diag_valid[1] = c->diag_valid[2] = false;
    return 0;
}

// ---- B-phase -------------------------------------------------------------------------------------------
int mcl_B_begin(mcl_context *c) {
    if (int rc = ready(c)) return rc;
    if (int rc = round_complete(c, "mcl_B_begin")) return rc;
    if (c->cond_monitor && c->regs[1].n == 0)  // (a stack WITH penalties comes here from mcl_update_B, which has looked already)
        if (int rc = monitor_condition(c, 1)) return rc;
    if (int rc = ensure_ctc(c)) return rc;
    if (int rc = ensure_xc(c)) return rc;
    if (c->opt.constant_B)
        if (int rc = mcl_launch_B_rho(c)) return rc;
    return 0;
}

float *mcl_B_rho_max(mcl_context *c) { return c ? c->rho_max : nullptr; }

static int B_factor_impl(mcl_context *c) {
    if (c->b_systems_valid) {  // built by the preceding A-finish from the same a_i and CtC
        c->b_systems_valid = false;
        return 0;
    }
    return mcl_launch_B_systems(c);
}

int mcl_B_factor(mcl_context *c) {
    if (int rc = ready(c)) return rc;
    return B_factor_impl(c);
}

// the finish pass of inner iteration t can be merged with the solve of t + 1 (k_rows_finish_solve_stats)
static bool step_can_chain(const mcl_context *c) {
    if (!c->step_fuse || !c->step_stats || c->sw.no_pass_chain) return false;
    int n_l2 = 0;
    for (int k = 0; k < c->regs[1].n; ++k) n_l2 += c->regs[1].kind[k] == MCL_PEN_L2BALL;
    return n_l2 <= 1;
}

int mcl_B_solve(mcl_context *c) {
    if (int rc = ready_noflush(c)) return rc;
    if (int rc = round_complete(c, "mcl_B_solve")) return rc;
    c->e1_valid = false;
    c->mseg_valid = c->grpart_valid = false;
    c->diag_valid[1] = false;
    // fusable stacks (see generic_inner_loop): statistics out of the solve pass, ONE prox + dual row pass once the
    // last penalty of the stack has been finished - the per-penalty calls below then only run the per-slab algebra
    c->step_fuse = mcl_stack_can_fuse(c, 1);
    c->step_stats = c->step_fuse && mcl_stats_can_ride_in_solve(c, 1);
    c->step_done_mask = 0;
    if (c->b_finish_pending) {  // previous inner iteration's prox + dual pass and this solve in one kernel
        c->b_finish_pending = false;
        if (step_can_chain(c)) return mcl_launch_rows_finish_solve_stats(c);
        if (int rc = mcl_launch_rows_finish_fused(c, 1, false)) return rc;
    }
    if (c->step_stats) return mcl_launch_rows_solve_stats(c);
    if (c->regs[1].n == 0) return mcl_launch_B_solve_f64(c);
    return mcl_launch_rows_solve(c, 1);
}

int mcl_B_end(mcl_context *c) {
    if (int rc = ready(c)) return rc;
    return round_complete(c, "mcl_B_end");
}

int mcl_B_prox_local(mcl_context *c, int32_t k) {
    if (int rc = ready(c)) return rc;
    if (k < 0 || k >= c->regs[1].n) return fail(c, "mcl_B_prox_local: penalty index out of range");
    c->stack_fused = c->step_fuse, c->stats_in_solve = c->step_stats;
    const int rc = mcl_launch_generic_prox_local(c, 1, k);
    c->stack_fused = c->stats_in_solve = false;
    return rc;
}

float *mcl_B_prox_reduce_buffer(mcl_context *c, int32_t k, int64_t *count) {
    if (!c || k < 0 || k >= c->regs[1].n || c->regs[1].kind[k] != MCL_PEN_PARAFAC2) {
        if (count) *count = 0;
        return nullptr;
    }
    if (count) *count = (int64_t)c->r * c->r + 1;
    return c->pf2_red;
}

int mcl_B_prox_finish(mcl_context *c, int32_t k) {
    if (int rc = ready(c)) return rc;
    if (k < 0 || k >= c->regs[1].n) return fail(c, "mcl_B_prox_finish: penalty index out of range");
    c->stack_fused = c->step_fuse, c->stats_in_solve = c->step_stats;
    int rc = mcl_launch_generic_prox_finish(c, 1, k);
    c->stack_fused = c->stats_in_solve = false;
    if (rc == 0 && c->step_fuse) {
        c->step_done_mask |= 1u << k;
        if (c->step_done_mask == (1u << c->regs[1].n) - 1u) {  // whole stack stepped: the fused row pass
            if (step_can_chain(c)) c->b_finish_pending = true;  // issued by the next entry point (merged into mcl_B_solve)
            else rc = mcl_launch_rows_finish_fused(c, 1, false);
            c->step_fuse = c->step_stats = false;
        }
    }
    return rc;
}

int mcl_update_B(mcl_context *c) {
    if (c && c->diag_pending && !c->diag_crossed_sweep && ready_noflush(c) == 0 && !c->b_finish_pending &&
        mcl_sweep_eligible(c)) {
        // a deferred diagnostics reduction survives ONE sweep (which writes the OTHER mode-1 table) and rides on the
        // C-phase reduction that follows it; a second sweep would flip the table parity back and overwrite the table the
        // deferral recorded, so a deferral that has already crossed a sweep is issued first (the else branch)
        c->diag_crossed_sweep = true;
    } else if (int rc = ready(c)) {
        return rc;
    }
    if (c->cond_monitor)
        if (int rc = monitor_condition(c, 1)) return rc;
    if (mcl_sweep_eligible(c)) {
        // one pass over X: B-phase fused with the per-bseg X^T B / B^T B that the C- and A-phases need (sweep.hip)
        if (int rc = ensure_ctc(c)) return rc;
        if (int rc = ensure_cfrag_sweep(c)) return rc;
        if (c->opt.constant_B)
            if (int rc = mcl_launch_B_rho(c)) return rc;
        if (int rc = B_factor_impl(c)) return rc;
        c->e1_valid = false;
        const int rc = mcl_launch_sweep(c);
        if (rc > 0) return rc;
        if (rc == 0) {
            c->mseg_valid = true;
            c->grpart_valid = true;
            c->diag_valid[1] = true;
            return 0;
        }
        c->sweep_planned = false;  // the device refused the kernel's LDS size: two-pass path from now on
        if (int rc2 = flush_diag(c)) return rc2;
    }
    if (c->regs[1].n == 0) {
        // plain least-squares update of B: un-shifted (or only l2-shifted) systems - X C, the inverse and the product in fp64
        if (int rc = ensure_ctc(c)) return rc;
        if (c->opt.constant_B)
            if (int rc = mcl_launch_B_rho(c)) return rc;
        if (int rc = B_factor_impl(c)) return rc;
        c->e1_valid = false;
        if (c->opt.inner_n_iter_max <= 0) return 0;
        c->mseg_valid = c->grpart_valid = false;
        c->diag_valid[1] = false;
        return mcl_launch_B_solve_f64(c);
    }
    if (int rc = mcl_B_begin(c)) return rc;
    if (int rc = mcl_B_factor(c)) return rc;
    c->e1_valid = false;
    if (c->opt.inner_n_iter_max <= 0) return 0;
    c->mseg_valid = c->grpart_valid = false;
    if (mcl_wide_applies(c, 1)) {  // small problem: the whole inner loop in fp64 (wide.hip)
        c->diag_valid[1] = false;  // (set again by the row-separable form, which leaves the table itself)
        return mcl_wide_phase(c, 1);
    }
    if (mcl_mode_is_row_separable(c, 1) && !(c->opt.inner_tol > 0.0)) {
        const int rc = mcl_launch_rows_fused(c, 1);
        if (rc == 0) {
            c->diag_valid[1] = true;
            return 0;
        }
        if (rc > 0) return rc;
    }
    return generic_inner_loop(c, 1);
}

// ---- C-phase -------------------------------------------------------------------------------------------
int mcl_update_C_local(mcl_context *c) {
    if (c && c->cond_monitor && ready_noflush(c) == 0 && !c->b_finish_pending)
        if (int rc = monitor_condition(c, 2)) return rc;
    if (c && c->diag_pending && ready_noflush(c) == 0 && !c->b_finish_pending && c->mseg_valid && c->grpart_valid)
        return mcl_launch_reduce_weighted(c);  // ... with the deferred diagnostics reduction on its spare workgroup
    if (int rc = ready(c)) return rc;
    if (c->exact) return mcl_launch_exact_gr(c);
    if (c->mseg_valid && c->grpart_valid) return mcl_launch_reduce_weighted(c);
    if (int rc = mcl_launch_contract_xt(c)) return rc;
    return mcl_launch_reduce_partials(c);
}

double *mcl_c_normal_equations(mcl_context *c, int64_t *count) {
    if (!c || !c->has_workspace) {
        if (count) *count = 0;
        return nullptr;
    }
    if (count) *count = c->K * c->r + (int64_t)c->r * c->r;
    return c->GR;
}

int mcl_update_C_finish(mcl_context *c) {
    if (int rc = ready(c)) return rc;
    ProfScope prof_(c, MCL_PROF_C_FINISH);
    if (mcl_wide_applies(c, 2)) {  // small problem: the whole inner loop in fp64 (wide.hip)
        if (int rc = mcl_launch_C_prepare(c)) return rc;
        c->xc_valid = c->ctc_valid = c->e1_valid = false;
        c->ctc_parts = 0;
        c->cfrag_valid = false;
        c->b_systems_valid = false;
        c->diag_valid[2] = false;
        c->variant[MCL_PROF_C_FINISH] = "k_C_prepare + fp64 inner loop (wide.hip)";
        return mcl_wide_phase(c, 2);
    }
    const bool checked_C = c->opt.inner_tol > 0.0 && c->regs[2].n > 0;  // inner stopping test: one launch per step
    if (c->opt.inner_n_iter_max > 0 && mcl_mode_is_row_separable(c, 2) && !checked_C) {
        // everything from the system solve to CtC / C fragments in one single-workgroup launch
        const int rc = mcl_launch_C_finish_fused(c);
        if (rc == 0) {
            c->xc_valid = c->e1_valid = false;
            c->b_systems_valid = false;
            c->ctc_valid = true;
            c->cfrag_valid = true;
            c->diag_valid[2] = true;
            return 0;
        }
        if (rc > 0) return rc;
    }
    if (int rc = mcl_launch_C_prepare(c)) return rc;
    if (c->opt.inner_n_iter_max <= 0) return 0;
    c->xc_valid = c->ctc_valid = c->e1_valid = false;
    c->cfrag_valid = false;
    c->b_systems_valid = false;
    if (c->regs[2].n == 0) {  // un-shifted normal equations: the solve runs in fp64
        c->diag_valid[2] = false;
        return mcl_launch_C_solve_f64(c);
    }
    if (mcl_mode_is_row_separable(c, 2) && !checked_C) {
        const int rc = mcl_launch_rows_fused(c, 2);
        if (rc == 0) {
            c->diag_valid[2] = true;
            return 0;
        }
        if (rc > 0) return rc;
    }
    return generic_inner_loop(c, 2);
}

// ---- A-phase -------------------------------------------------------------------------------------------
int mcl_A_begin(mcl_context *c) {
    if (int rc = ready(c)) return rc;
    if (c->cond_monitor)
        if (int rc = monitor_condition(c, 0)) return rc;
    // C^T C still in the partial blocks of k_C_finish_multi: the rows kernels of the A-phase finish sum them themselves (and
    // write the totals); anything else - the constant-rho pre-pass, the column-layout finish - gets them folded first
    const bool rows_finish = !(c->RP == 64 || c->RP == 4 || c->sw.a_finish_cols) && !c->opt.constant_A;
    if (!(c->ctc_valid && c->ctc_parts > 0 && rows_finish))
        if (int rc = ensure_ctc(c)) return rc;
    // When X C has to be recomputed anyway (C changed), the per-slab reductions ride in its epilogue; the
    // constant-rho pre-pass (k_A_rho) needs the assembled per-slab Gram, so it keeps the separate kernel.
    c->use_seg_gram = false;
    c->seg_from_sweep = false;
    c->a_rhs_from_M = false;
    c->a_rhs_wide = false;
    c->a_rhs_pairs = false;
    if (c->mseg_valid && !c->opt.constant_A && mcl_mode_is_row_separable(c, 0)) {
        // the sweep left M_i = X_i^T B_i per bseg: rhs_i = coldot(M_i, C), no pass over X
        if (int rc = ensure_cfrag_sweep(c)) return rc;
        // rank 5..32: the finish kernel forms rhs_i from the sweep's M partials itself (one launch less); otherwise a
        // separate pass.  Up to 8 partials per slab a workgroup per slab streams them on three waves while the fourth inverts
        // the system (k_A_finish_rows_wide; one wave alone took 27 us for 8 partials, against 5 + 13 us apart); one partial
        // per slab on big problems: the finishing wave streams it (16 KB at K = 256) in front of its Gauss-Jordan; more
        // than 8: the separate kernel
        const bool fusable = (c->RP == 8 || c->RP == 16 || c->RP == 32) && !c->sw.a_finish_cols && !c->sw.no_a_fusion;
        // (with one partial per slab only while all the workgroups are resident at once - two per CU: beyond 512 slabs a
        // second round of workgroups would cost more than the overlap returns, config 3: +4 us)
        c->a_rhs_wide = fusable && c->n_parts <= 8 * c->I && (c->n_parts > c->I || c->I <= 512) && !c->sw.no_a_wide;
        // ... up to 1024 slabs two slabs share a workgroup: one system wave and one streaming wave each
        c->a_rhs_pairs = fusable && !c->a_rhs_wide && c->n_parts <= c->I && c->I <= 1024 && !c->sw.no_a_wide;
        if (c->a_rhs_pairs) c->a_rhs_wide = true;
        c->a_rhs_from_M = fusable && (c->n_parts <= c->I || c->a_rhs_wide);
        if (!c->a_rhs_from_M)
            if (int rc = mcl_launch_A_rhs_from_M(c)) return rc;
        c->use_seg_gram = true;
        c->seg_from_sweep = true;
        c->e1_valid = false;
        return 0;
    }
    if (!c->xc_valid && !c->opt.constant_A && mcl_mode_is_row_separable(c, 0) && !c->sw.no_fused_gram) {
        c->xc_with_gram = true;
        const int rc = ensure_xc(c);
        c->xc_with_gram = false;
        if (rc) return rc;
        c->use_seg_gram = c->xc_did_gram;
    } else {
        if (int rc = ensure_xc(c)) return rc;
    }
    if (!c->use_seg_gram)
        if (int rc = mcl_launch_slab_gram(c)) return rc;
    c->e1_valid = false;
    if (c->opt.constant_A)
        if (int rc = mcl_launch_A_rho(c)) return rc;
    return 0;
}

float *mcl_A_rho_max(mcl_context *c) { return c ? c->rho_max + 1 : nullptr; }

int mcl_A_finish(mcl_context *c) {
    if (int rc = ready(c)) return rc;
    if (c->opt.inner_n_iter_max <= 0) return 0;
    ProfScope prof_(c, MCL_PROF_A_FINISH);
    c->grpart_valid = false;  // the sweep's [G | R] partials were weighted with the previous a_i
    if (mcl_mode_is_row_separable(c, 0) && c->opt.inner_tol > 0.0 && c->regs[0].n > 0) {
        // inner stopping test: the systems, then one launch per step (k_A_rows_solve, per-row prox, table, test)
        if (int rc = mcl_launch_A_finish(c, false)) return rc;
        if (int rc = generic_inner_loop(c, 0)) return rc;
        if (int rc = mcl_launch_A_e1(c, true)) return rc;
    } else if (mcl_mode_is_row_separable(c, 0)) {
        if (int rc = mcl_launch_A_finish(c, true)) return rc;
    } else {
        for (int k = 0; k < c->regs[0].n; ++k)
            if (c->regs[0].kind[k] == MCL_PEN_EXTERNAL)
                return fail(c, "mode 0 has host-evaluated penalties: use mcl_A_factor / mcl_A_solve / mcl_A_end");
        if (!c->opt.constant_A)
            return fail(c, "matrix penalties on mode 0 need constant_feasibility_penalty (the reference raises "
                           "AttributeError: no factor_matrix_row_update)");
        if (int rc = mcl_launch_A_finish(c, false)) return rc;
        if (mcl_wide_applies(c, 0)) {  // small problem: the whole inner loop in fp64 (wide.hip)
            if (int rc = mcl_wide_phase(c, 0)) return rc;
        } else if (int rc = generic_inner_loop(c, 0)) {
            return rc;
        }
        if (int rc = mcl_launch_A_e1(c, true)) return rc;
    }
    c->e1_valid = true;
    c->e1_from_raw_gram = false;  // the BtB buffer now holds Q_i (cross_products)
    c->diag_valid[0] = true;
    return 0;
}

int mcl_update_A(mcl_context *c) {
    if (int rc = mcl_A_begin(c)) return rc;
    return mcl_A_finish(c);
}

int mcl_A_factor(mcl_context *c) {
    if (int rc = ready(c)) return rc;
    if (c->use_seg_gram) return fail(c, "internal: mcl_A_factor needs the assembled per-slab Gram");
    c->b_systems_valid = false;
    return mcl_launch_A_finish(c, false);
}

int mcl_A_solve(mcl_context *c) {
    if (int rc = ready(c)) return rc;
    c->grpart_valid = false;
    c->e1_valid = false;
    c->diag_valid[0] = false;
    c->b_systems_valid = false;
    return mcl_launch_A_rows_solve(c);
}

int mcl_A_end(mcl_context *c) {
    if (int rc = ready(c)) return rc;
    if (int rc = mcl_launch_A_e1(c, true)) return rc;
    c->e1_valid = true;
    c->e1_from_raw_gram = false;
    c->diag_valid[0] = true;
    return 0;
}

int mcl_C_begin(mcl_context *c) {
    if (int rc = ready(c)) return rc;
    return mcl_launch_C_prepare(c);
}

int mcl_C_solve(mcl_context *c) {
    if (int rc = ready(c)) return rc;
    c->xc_valid = c->ctc_valid = c->e1_valid = false;
    c->cfrag_valid = false;
    c->b_systems_valid = false;
    c->diag_valid[2] = false;
    return mcl_launch_rows_solve(c, 2);
}

int mcl_C_end(mcl_context *c) {
    if (int rc = ready(c)) return rc;
    c->xc_valid = c->ctc_valid = c->e1_valid = false;
    c->cfrag_valid = false;
    c->b_systems_valid = false;
    c->diag_valid[2] = false;
    return 0;
}

// ---- diagnostics ---------------------------------------------------------------------------------------
// every table the diagnostics vector is reduced from is made current (what mcl_diagnostics does before its reduction)
static int mcl_prepare_diag_tables(mcl_context *c) {
    if (!c->xsq_valid) {
        if (int rc = mcl_launch_x_sq(c)) return rc;
        c->xsq_valid = true;
    }
    if (!c->diag_valid[1])
        if (int rc = mcl_launch_rows_diag(c, 1)) return rc;
    if (!c->diag_valid[2])
        if (int rc = mcl_launch_rows_diag(c, 2)) return rc;
    c->diag_valid[1] = c->diag_valid[2] = true;
    if (!c->e1_valid) {
        // no current A-phase by-products: full formula with a pass over X (decomposition.py:430-444)
        if (int rc = ensure_ctc(c)) return rc;
        if (int rc = ensure_xc(c)) return rc;
        if (int rc = mcl_launch_slab_gram(c)) return rc;
        if (int rc = mcl_launch_A_e1(c, false)) return rc;
        c->e1_valid = true;
        c->e1_from_raw_gram = true;
    } else if (!c->diag_valid[0]) {
        if (int rc = mcl_launch_A_e1(c, !c->e1_from_raw_gram)) return rc;
    }
    c->diag_valid[0] = true;
    return 0;
}

int mcl_diagnostics(mcl_context *c, double *out, int32_t include_replicated) {
    if (int rc = ready(c)) return rc;
    if (!out) return fail(c, "mcl_diagnostics: out is NULL");
    if (int rc = mcl_prepare_diag_tables(c)) return rc;
    return mcl_launch_diag_final(c, out, include_replicated, true);
}

int mcl_diagnostics_deferred(mcl_context *c, double *out, int32_t include_replicated) {
    if (!c) return 1;
    if (!out) return fail(c, "mcl_diagnostics_deferred: out is NULL");
    if (int rc = ready(c)) return rc;  // also issues an older deferred reduction
    // deferrable: every table is current and ||X||^2 is known, i.e. mcl_diagnostics() would only launch its reduction
    if (!(c->xsq_valid && c->e1_valid && c->diag_valid[0] && c->diag_valid[1] && c->diag_valid[2]) || c->sw.no_diag_defer)
        return mcl_diagnostics(c, out, include_replicated);
    c->diag_pending_T = mcl_diag_tables(c, true);
    c->diag_pending_out = out, c->diag_pending_incl = include_replicated ? 1 : 0;
    c->diag_pending = true;
    return 0;
}

int mcl_flush_diagnostics(mcl_context *c) {
    if (!c) return 1;
    return flush_diag(c);
}

int mcl_penalty_value(mcl_context *c, int32_t mode, int32_t k, double *out) {
    if (!c) return 1;
    if (int rc = ready(c)) return rc;
    if (mode < 0 || mode > 2 || k < 0 || k >= c->regs[mode].n || !out) return fail(c, "mcl_penalty_value: bad arguments");
    if (c->regs[mode].kind[k] != MCL_PEN_GL2) return fail(c, "mcl_penalty_value: only GeneralizedL2 penalties have a separate value");
    return mcl_launch_gl2_value(c, mode, k, out);
}

int mcl_condition_probe(mcl_context *c, int32_t mode_mask, double *out) {
    if (!c) return 1;
    if (!out) return fail(c, "mcl_condition_probe: out is NULL");
    if (int rc = ready(c)) return rc;
    int want = 0;
    for (int m = 0; m < 3; ++m)
        if ((mode_mask >> m & 1) && c->regs[m].n == 0) want |= 1 << m;  // a mode WITH penalties solves shifted, well-conditioned systems
    if (int rc = ensure_ctc(c)) return rc;
    return mcl_launch_cond_probe(c, want, out);
}

int mcl_condition_monitor(mcl_context *c, double *out, int32_t mode_mask) {
    if (!c) return 1;
    c->cond_monitor = out;
    c->cond_monitor_mask = out ? mode_mask : 0;
    return 0;
}

int mcl_iterate(mcl_context *c, int32_t n_iter, int32_t update_A, int32_t update_B, int32_t update_C,
                double *diag_ring) {
    if (int rc = ready(c)) return rc;
    for (int it = 0; it < n_iter; ++it) {
        if (update_B)
            if (int rc = mcl_update_B(c)) return rc;
        if (update_C) {
            if (int rc = mcl_update_C_local(c)) return rc;
            if (int rc = mcl_update_C_finish(c)) return rc;
        }
        if (update_A)
            if (int rc = mcl_update_A(c)) return rc;
        if (diag_ring) {
            // between two iterations the reduction of the diagnostics tables is deferred: it rides on the next C-phase
            // reduction kernel when that is the sweep path's, and is issued on its own otherwise
            double *slot = diag_ring + (int64_t)it * MCL_DIAG_LEN;
            const int rc = (it + 1 < n_iter && update_B && update_C) ? mcl_diagnostics_deferred(c, slot, 1)
                                                                      : mcl_diagnostics(c, slot, 1);
            if (rc) return rc;
        }
    }
    return flush_diag(c);
}

// Everything the context caches about the factors (validity flags of by-products, pending deferrals): forgotten.  After a
// gated run that stopped early the scratch buffers hold a mixture of the stopping iteration's by-products and recomputed
// ones, the host-side flags describe iterations that did not happen - the factors and ADMM variables are exact.
static void forget_byproducts(mcl_context *c) {
    c->b_finish_pending = false;
    c->diag_pending = c->diag_crossed_sweep = false;
    c->step_fuse = c->step_stats = false;
    c->b_systems_valid = false;
    c->cfrag_valid = false;
    c->xc_valid = c->ctc_valid = c->e1_valid = false;
    c->ctc_parts = 0;
    c->mseg_valid = c->grpart_valid = false;
    c->diag_valid[0] = c->diag_valid[1] = c->diag_valid[2] = false;
}

int mcl_run(mcl_context *c, int32_t n_iter_max, int32_t update_A, int32_t update_B, int32_t update_C,
            const mcl_stop_rule *rule, double *diag_ring, double *verdict_ring, mcl_run_status *status) {
    if (!c) return 1;
    if (!rule || !status) return fail(c, "mcl_run: rule and status must not be NULL (fixed iteration counts: mcl_iterate)");
    if (n_iter_max > 0 && (!diag_ring || !verdict_ring)) return fail(c, "mcl_run: diag_ring / verdict_ring is NULL");
    if (int rc = ready(c)) return rc;
    if (has_kind(c, MCL_PEN_GL2))  //  its value trace(F^T M F) is no column of the diagnostics vector
        return fail(c, "mcl_run: a GeneralizedL2 penalty's value is not part of the diagnostics vector, so the device-side stopping "
                       "rule would evaluate the loss without it (decomposition.py:1016-1023): drive the iterations with mcl_iterate / "
                       "the phase calls and add mcl_penalty_value to the loss on the host");
    int *status_dev = nullptr;
    if (hipHostGetDevicePointer(reinterpret_cast<void **>(&status_dev), (void *)status, 0) != hipSuccess || !status_dev) {
        (void)hipGetLastError();
        return fail(c, "mcl_run: status must live in pinned (page-locked, device-mapped) host memory");
    }
    status->stopped = 0, status->stop_iteration = -1, status->code = 0, status->progress = 0;
    if (n_iter_max <= 0) return 0;
    const int ahead = rule->max_run_ahead > 0 ? rule->max_run_ahead : 8;
    c->h_stop_init[0] = rule->initial_loss;
    MCL_CHECK_HIP(c, hipMemsetAsync(c->gate, 0, 4 * sizeof(int), c->stream));
    MCL_CHECK_HIP(c, hipMemcpyAsync(c->stop_state, c->h_stop_init, sizeof(double), hipMemcpyHostToDevice, c->stream));
    auto set_gate = [&](const int *g) {
        c->gate_active = g;
        for (int m = 0; m < 3; ++m) c->regs[m].gate = g;
    };
    set_gate(c->gate);
    int rc = 0, enqueued = 0;
    bool wedged = false;  // the watchdog gave up: the stream may never drain, nothing below may wait for it
    if (c->sw.test_mute_verdict) status_dev = c->mute_status;  // test hook: the verdicts never reach the host
    // the status words are written by the device (mapped pinned memory): every read in the wait loop is a volatile read
    const volatile mcl_run_status *seen = status;
    for (int it = 0; it < n_iter_max && rc == 0; ++it) {
        // bounded run-ahead: wait (without synchronising the stream) until the device is at most `ahead` verdicts behind.
        // A short spin first (the common wait is a fraction of an iteration), then sleeps that back off to 50 us - the queue
        // holds `ahead` iterations of work, so the host core is free while the device catches up; the stream is polled about
        // once per millisecond of waiting, and a device that reports NO progress for MCL_RUN_WATCHDOG_S seconds (120; one
        // outer iteration of the largest BASELINE configuration takes 0.12 s) ends the call with an error instead of a hang.
        long spins = 0;
        int last_progress = seen->progress;
        auto t_progress = std::chrono::steady_clock::now(), t_query = t_progress;
        long sleep_ns = 2000;
        while (!seen->stopped && it - seen->progress >= ahead) {
            if (++spins <= c->sw.run_spins) {
                cpu_relax();
                continue;
            }
            const timespec ts{0, sleep_ns};
            nanosleep(&ts, nullptr);
            sleep_ns = std::min<long>(sleep_ns * 2, 50000);
            const auto now = std::chrono::steady_clock::now();
            if (seen->progress != last_progress) last_progress = seen->progress, t_progress = now;
            if (now - t_query >= std::chrono::milliseconds(1)) {  // is the stream still working?
                t_query = now;
                const hipError_t q = hipStreamQuery(c->stream);
                if (q != hipErrorNotReady && it - seen->progress >= ahead && !seen->stopped) {
                    rc = fail(c, q == hipSuccess ? "mcl_run: the stream drained without the verdict kernel reporting progress"
                                                 : std::string("mcl_run: ") + hipGetErrorString(q));
                    break;
                }
            }
            if (now - t_progress >= std::chrono::duration<double>(c->sw.run_watchdog_s)) {
                char secs[32];
                snprintf(secs, sizeof secs, "%g", c->sw.run_watchdog_s);
                rc = fail(c, std::string("mcl_run: no verdict from the device for ") + secs +
                                 " s (MCL_RUN_WATCHDOG_S): giving up the wait WITHOUT synchronising the stream; the context is "
                                 "failed - keep the workspace, the rings and `status` allocated until the device has been "
                                 "synchronised or reset, then destroy the context");
                wedged = true;
                break;
            }
        }
        if (rc || seen->stopped) break;
        if (update_B) rc = mcl_update_B(c);
        if (rc == 0 && update_C) {
            rc = mcl_update_C_local(c);
            if (rc == 0) rc = mcl_update_C_finish(c);
        }
        if (rc == 0 && update_A) rc = mcl_update_A(c);
        if (rc == 0) rc = mcl_prepare_diag_tables(c);
        if (rc == 0)
            rc = mcl_launch_diag_verdict(c, diag_ring + (int64_t)it * MCL_DIAG_LEN, rule, it,
                                         verdict_ring + (int64_t)it * 4, status_dev);
        enqueued = it + 1;
    }
    set_gate(nullptr);
    if (wedged) {
        // a device that has reported nothing for the whole watchdog period may never drain its stream: a synchronisation
        // here is the very hang the watchdog exists to end.  The enqueued kernels still refer to the caller's buffers.
        c->failed = true;
        c->failed_why = "mcl_run gave up waiting for the device";
        forget_byproducts(c);
        return rc;
    }
    const hipError_t e = hipStreamSynchronize(c->stream);  // the call reports the verdict: the one entry point that waits
    if (e != hipSuccess && rc == 0) rc = fail(c, std::string("mcl_run: hipStreamSynchronize: ") + hipGetErrorString(e));
    if (status->stopped && enqueued > status->stop_iteration + 1) forget_byproducts(c);
    return rc;
}

int mcl_record_event(mcl_context *c, void *hip_event) {
    if (!c || !hip_event) return 1;
    if (int rc = ready(c)) return rc;  // pending deferred work belongs in front of the event
    MCL_CHECK_HIP(c, hipEventRecord(reinterpret_cast<hipEvent_t>(hip_event), c->stream));
    return 0;
}

int mcl_wait_event(mcl_context *c, void *hip_event) {
    if (!c || !hip_event) return 1;
    MCL_CHECK_HIP(c, hipStreamWaitEvent(c->stream, reinterpret_cast<hipEvent_t>(hip_event), 0));
    return 0;
}

// ---- the pieces of mcl_run for a host that drives the iterations itself (the sharded loop: reductions between the calls) ----
int mcl_gate_begin(mcl_context *c, const mcl_stop_rule *rule, mcl_run_status *status) {
    if (!c) return 1;
    if (!rule || !status) return fail(c, "mcl_gate_begin: rule and status must not be NULL");
    if (int rc = ready(c)) return rc;
    if (has_kind(c, MCL_PEN_GL2))
        return fail(c, "mcl_gate_begin: a GeneralizedL2 penalty's value is not part of the diagnostics vector (see mcl_run): write it "
                       "into the penalty-value slot of the vector on the host before mcl_verdict is NOT supported either - evaluate "
                       "the rule on the host");
    int *status_dev = nullptr;
    if (hipHostGetDevicePointer(reinterpret_cast<void **>(&status_dev), (void *)status, 0) != hipSuccess || !status_dev) {
        (void)hipGetLastError();
        return fail(c, "mcl_gate_begin: status must live in pinned (page-locked, device-mapped) host memory");
    }
    status->stopped = 0, status->stop_iteration = -1, status->code = 0, status->progress = 0;
    c->run_rule = *rule;
    c->run_status_dev = status_dev;
    c->h_stop_init[0] = rule->initial_loss;
    MCL_CHECK_HIP(c, hipMemsetAsync(c->gate, 0, 4 * sizeof(int), c->stream));
    MCL_CHECK_HIP(c, hipMemcpyAsync(c->stop_state, c->h_stop_init, sizeof(double), hipMemcpyHostToDevice, c->stream));
    c->gate_active = c->gate;
    for (int m = 0; m < 3; ++m) c->regs[m].gate = c->gate;
    return 0;
}

int mcl_verdict(mcl_context *c, const double *diag_vec, int32_t iteration, double *verdict_row) {
    if (!c) return 1;
    if (!c->gate_active || !c->run_status_dev) return fail(c, "mcl_verdict: call mcl_gate_begin first");
    if (!diag_vec || !verdict_row) return fail(c, "mcl_verdict: NULL vector");
    return mcl_launch_verdict(c, diag_vec, &c->run_rule, iteration, verdict_row, c->run_status_dev);
}

int mcl_gate_end(mcl_context *c, int32_t stopped_early) {
    if (!c) return 1;
    c->gate_active = nullptr;
    c->run_status_dev = nullptr;
    for (int m = 0; m < 3; ++m) c->regs[m].gate = nullptr;
    if (stopped_early) forget_byproducts(c);
    return 0;
}

float *mcl_internal_buffer(mcl_context *c, int32_t which, int64_t *count) {
    if (!c || !c->has_workspace) return nullptr;
    float *p = nullptr;
    int64_t n = 0;
    switch (which) {
        case MCL_BUF_RHSES: p = c->rhsA, n = c->I * c->r; break;
        case MCL_BUF_CROSS_PRODUCTS: p = c->BtB, n = c->I * c->r * c->r; break;
        case MCL_BUF_XC: p = c->XC, n = c->N * c->r; break;
        case MCL_BUF_RHO_B: p = c->rhoB, n = c->I; break;
        case MCL_BUF_RHO_A: p = c->rhoA, n = c->I; break;
        case MCL_BUF_RHO_C: p = c->rhoC, n = 1; break;
        case MCL_BUF_CTC:
            if (c->ctc_valid && c->ctc_parts > 0) (void)mcl_launch_ctc_fold(c);
            p = c->CtC, n = (int64_t)c->r * c->r;
            break;
        case MCL_BUF_LINV_B: p = c->LinvB, n = c->I * c->r * c->r; break;
        case MCL_BUF_PF2_ACC: p = reinterpret_cast<float *>(c->pf2_acc), n = c->pf2_acc ? 2 * c->I * ((int64_t)c->r * c->r + 1) : 0; break;
        case MCL_BUF_PF2_GRAM: p = reinterpret_cast<float *>(c->pf2_S), n = c->pf2_S ? 2 * c->I * (int64_t)c->r * c->r : 0; break;
        case MCL_BUF_SWEEP_CYCLES: p = reinterpret_cast<float *>(c->sweep_cycles), n = c->sweep_cycles ? (int64_t)2048 * 6 * 2 : 0; break;
        case MCL_BUF_PF2_STATUS: p = reinterpret_cast<float *>(c->pf2_status), n = c->pf2_status ? c->I : 0; break;  // int32 bits
#if defined(MCL_NS_STAMPS) || defined(MCL_UNI_STAMPS)
        case MCL_BUF_NS_STAMPS: p = c->pf2_xmin, n = c->pf2_xmin ? 17 * c->I : 0; break;
#endif
        // the planner's work-unit tables (int32 bits): segments of the X passes, bsegs of the sweep, and the first
        // unit of every wave (mcl_set_problem)
        case MCL_BUF_SEG_ROW0: p = reinterpret_cast<float *>(c->segs.row0), n = c->segs.n_tiles; break;
        case MCL_BUF_SEG_NROWS: p = reinterpret_cast<float *>(c->segs.nrows), n = c->segs.n_tiles; break;
        case MCL_BUF_WAVE_SEG_PTR: p = reinterpret_cast<float *>(c->wave_seg_ptr), n = c->n_seg_waves + 1; break;
        case MCL_BUF_BSEG_ROW0: p = reinterpret_cast<float *>(c->bsegs.row0), n = c->bsegs.n_tiles; break;
        case MCL_BUF_BSEG_NROWS: p = reinterpret_cast<float *>(c->bsegs.nrows), n = c->bsegs.n_tiles; break;
        case MCL_BUF_BSEG_PART: p = reinterpret_cast<float *>(c->bseg_part), n = c->bseg_part ? c->bsegs.n_tiles : 0; break;
        case MCL_BUF_WAVE_BSEG_PTR: p = reinterpret_cast<float *>(c->wave_bseg_ptr), n = c->wave_bseg_ptr ? c->n_bseg_waves + 1 : 0; break;
        default: break;
    }
    if (count) *count = n;
    return p;
}

int mcl_profile_enable(mcl_context *c, int32_t capacity) {
    if (!c) return 1;
    for (int s = 0; s < MCL_PROF_SLOTS; ++s) {
        for (hipEvent_t e : c->prof_ev[s]) (void)hipEventDestroy(e);
        c->prof_ev[s].clear();
        c->prof_used[s] = 0;
        c->prof_seen[s] = 0;
        c->prof_launches[s] = 0;
    }
    c->prof_capacity = 0;
    c->prof_nested = false;
    if (capacity <= 0) return 0;
    for (int s = 0; s < MCL_PROF_SLOTS; ++s) {
        c->prof_ev[s].resize((size_t)2 * capacity);
        for (auto &e : c->prof_ev[s]) MCL_CHECK_HIP(c, hipEventCreate(&e));
    }
    c->prof_capacity = capacity;
    // What an event pair adds to the kernel between its events (the command processor's marker -> dispatch and completion ->
    // marker handling), calibrated on the smallest kernel there is (one store to a scratch word): a pair around ONE of them reads
    // T1, around TWO back to back T2 - the second one's marginal cost T2 - T1 is what such a kernel takes inside a stream of
    // kernels, so the pair itself costs T1 - (T2 - T1).  (An EMPTY pair reads ~4.5 us and over-corrects: part of it overlaps
    // with the dispatch of the kernel it brackets.)
    c->prof_overhead_ms = 0.0;
    if (c->has_workspace) {
        hipEvent_t ev[2];
        MCL_CHECK_HIP(c, hipEventCreate(&ev[0]));
        MCL_CHECK_HIP(c, hipEventCreate(&ev[1]));
        float best[2] = {1e30f, 1e30f};
        for (int n = 1; n <= 2; ++n)
            for (int t = 0; t < 12; ++t) {
                MCL_CHECK_HIP(c, hipEventRecord(ev[0], c->stream));
                for (int k = 0; k < n; ++k) hipLaunchKernelGGL(k_prof_nop, dim3(1), dim3(64), 0, c->stream, c->mute_status);
                MCL_CHECK_HIP(c, hipEventRecord(ev[1], c->stream));
                MCL_CHECK_HIP(c, hipEventSynchronize(ev[1]));
                float ms = 0.f;
                MCL_CHECK_HIP(c, hipEventElapsedTime(&ms, ev[0], ev[1]));
                if (t >= 2 && ms < best[n - 1]) best[n - 1] = ms;
            }
        (void)hipEventDestroy(ev[0]);
        (void)hipEventDestroy(ev[1]);
        if (best[0] < 1e29f && best[1] < 1e29f) c->prof_overhead_ms = std::max(0.0, 2.0 * (double)best[0] - (double)best[1]);
    }
    return 0;
}

double mcl_profile_overhead_us(mcl_context *c) { return c ? 1e3 * c->prof_overhead_ms : 0.0; }

int mcl_profile_set_stride(mcl_context *c, int32_t stride) {
    if (!c || stride < 1) return 1;
    c->prof_stride = stride;
    for (int s = 0; s < MCL_PROF_SLOTS; ++s) c->prof_seen[s] = 0, c->prof_launches[s] = 0;
    return 0;
}

int mcl_profile_read(mcl_context *c, int32_t which, double *total_ms, int32_t *count) {
    if (!c || which < 0 || which >= MCL_PROF_SLOTS || !total_ms || !count) return 1;
    double tot = 0.0;
    const int n = c->prof_used[which];
    for (int i = 0; i < n; ++i) {
        MCL_CHECK_HIP(c, hipEventSynchronize(c->prof_ev[which][2 * i + 1]));
        float ms = 0.f;
        MCL_CHECK_HIP(c, hipEventElapsedTime(&ms, c->prof_ev[which][2 * i], c->prof_ev[which][2 * i + 1]));
        tot += ms;
    }
    *total_ms = tot;
    *count = n;
    c->prof_used[which] = 0;
    return 0;
}

int64_t mcl_profile_launches(mcl_context *c, int32_t which) {
    if (!c || which < 0 || which >= MCL_PROF_SLOTS) return -1;
    return c->prof_launches[which];
}

int mcl_reload_switches(mcl_context *c) {
    if (!c) return 1;
    read_switches(c->sw);
    c->active_switches = switches_in_env();
    return 0;
}

const char *mcl_active_switches(const mcl_context *c) { return c ? c->active_switches.c_str() : ""; }

const char *mcl_kernel_variant(mcl_context *c, int32_t which) {
    if (!c) return "";
    if (which == MCL_VARIANT_EXACT_MODE) return c->exact ? "exact products (fp64 sums; small problem)" : "";
    if (which < 0 || which >= MCL_PROF_SLOTS) return "";
    return c->variant[which].c_str();
}

}  // extern "C"
