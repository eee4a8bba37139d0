// fp64 inner ADMM loops for SMALL problems (the exact-products mode, contract.hip: mcl_exact_mode).
//
// The fast row kernels run the inner loop (decomposition.py:259-285 / 326-338) on the fp32 matrix core and - for stacks with
// slab-wise penalties - round the factor and every ADMM variable to fp32 after EVERY inner iteration (one row pass each).
// Per phase that leaves 4e-7 .. 1e-6 in B / C (tools/parity_probe.py) where the storage level is 3e-8, and a following
// penalty-free system of condition 1e4 .. 1e5 turns it into the 1e-5 .. 8.5e-5 of the extended fuzz sweep's residue
// (tests/test_gpu_fuzz_parity.py: RESIDUE_SEEDS).  On the problems of the exact-products mode (at most 2^20 elements of X:
// the reference's own regime) time is not the constraint, so modes 1 and 2 take their whole inner loop in fp64 there:
//   * right-hand sides from the fp64 contractions (XC64, the fp64 [G | R]), systems from the fp64 inverses;
//   * the factor and every auxiliary / dual variable live in fp64 shadow arrays for the length of the phase and are
//     rounded to the caller's fp32 buffers ONCE per value written (the shadows are loaded from those buffers at the start
//     of the phase: the caller owns the state between phases);
//   * penalty parameters in double (RegSet::p0d / p1d), thresholds divided in double.
// Row-separable stacks (NonNegativity / Box / L1, penalties.py:488-592) take ONE kernel for the whole loop, like the fast
// path; stacks with an L2 ball (penalties.py:920-925), unimodality (:1014-1015), total variation (:750-841) or PARAFAC2
// (:1224-1250, 1280-1281) take one kernel per step.  Layout of every kernel: a wave walks the rows of a tile (<= 64 rows
// of ONE slab), lane c owns column c (rank <= 64); r x r operands live in LDS.
#include <algorithm>

#include "mcl_internal.h"
#include "rows_mfma.h"

int mcl_launch_pf2_jacobi_wide(mcl_context *c, int k, const double *F64, const double *U64, const double *D64);  // generic.hip
int mcl_launch_gl2_wide(mcl_context *c, int mode, int k, const double *F64, double *Z64, double *D64);             // generic.hip
int mcl_launch_simplex_wide(mcl_context *c, int mode, int k, const double *F64, const double *D64, double *Z64);  // generic.hip
int mcl_launch_inner_check_raw(mcl_context *c, bool begin, const double *change_part, int n_change, const double *tab, int n_rows,
                               int n_regs);  // generic.hip

namespace {

struct WideRows {  // the rows of one mode, as tiles of <= 64 rows of one slab
    const int *tile_slab, *tile_row0, *tile_nrows;
    int n_tiles;
    const int *ext;  // slab extents
    int n_slabs;
    const float *rho;  // [n_slabs]
    const int *gate;
};

__device__ __forceinline__ double prox_rowsep_d(int kind, int nonneg, double p0, double p1, double thr, double y) {
    switch (kind) {
        case MCL_PEN_NN: return fmax(y, 0.0);
        case MCL_PEN_BOX: return fmin(fmax(y, p0), p1);
        case MCL_PEN_L1:
            if (nonneg) return fmax(y - thr, 0.0);
            return copysign(fmax(fabs(y) - thr, 0.0), y);
        default: return y;
    }
}

// The rows one wave (= one workgroup) works on: RW x G rows of ONE tile (<= 64 rows of one slab), G = 64 / RPW lane groups of
// RPW = 16 / 32 / 64 >= r lanes; lane (grp, c) owns column c of rows jbase + q G + grp, q < RW.  Workgroups per tile:
// ceil(64 / (RW G)) - a 4096-row factor of rank 16 is spread over 256 waves.
constexpr int WIDE_RW = 4;
struct RowGroup {
    int c, grp, G, RPW, slab, nrows, jbase, r;
    long row0;
    bool act;
    // one wave per workgroup (workgroup b = tile b / subs, sub-group b % subs), or - whole_tile - one workgroup per tile with
    // one wave per sub-group (the row-separable kernel: its workgroup also sums the tile's diagnostics)
    __device__ RowGroup(const WideRows &W, int r_, bool whole_tile = false) : r(r_) {
        RPW = r <= 16 ? 16 : (r <= 32 ? 32 : 64);
        G = 64 / RPW;
        const int per = WIDE_RW * G, subs = (64 + per - 1) / per;
        const int tile = whole_tile ? (int)blockIdx.x : (int)blockIdx.x / subs;
        const int sub = whole_tile ? (int)(threadIdx.x >> 6) : (int)blockIdx.x - tile * subs;
        const int ln = threadIdx.x & 63;
        c = ln % RPW, grp = ln / RPW;
        act = c < r;
        slab = W.tile_slab[tile], row0 = W.tile_row0[tile], nrows = W.tile_nrows[tile];
        jbase = sub * per;
    }
    __device__ bool any() const { return jbase < nrows; }  // (workgroup-uniform)
    __device__ long idx(int q, bool &ok) const {
        const int j = jbase + q * G + grp;
        ok = act && j < nrows;
        return (row0 + min(j, nrows - 1)) * r + (act ? c : 0);
    }
};
static int wide_subs(int r) {  // waves a 64-row tile is spread over
    const int G = 64 / (r <= 16 ? 16 : (r <= 32 ? 32 : 64)), per = WIDE_RW * G;
    return (64 + per - 1) / per;
}
static unsigned wide_grid(const WideRows &W, int r) {
    const int G = 64 / (r <= 16 ? 16 : (r <= 32 ? 32 : 64)), per = WIDE_RW * G;
    return (unsigned)(W.n_tiles * ((64 + per - 1) / per));
}

// f_c = sum_d t_d M[d][c] for the wave's RW x G rows at once: t of the rows through the wave's LDS strip (broadcast reads
// within a lane group), M in LDS
template <int RW>
__device__ __forceinline__ void row_times_matrix(const double (&t)[RW], double (&f)[RW], double *ts, const double *Ms, const RowGroup &R) {
    const int base = R.grp * R.RPW;
#pragma unroll
    for (int q = 0; q < RW; ++q) ts[q * 64 + base + R.c] = t[q];
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the strip is written (one wave: program order suffices beyond that)
#pragma unroll
    for (int q = 0; q < RW; ++q) f[q] = 0.0;
    for (int d = 0; d < R.r; ++d) {
        const double m = R.act ? Ms[d * R.r + R.c] : 0.0;
#pragma unroll
        for (int q = 0; q < RW; ++q) f[q] = fma(ts[q * 64 + base + d], m, f[q]);
    }
    __builtin_amdgcn_wave_barrier();
}

// ---------------------------------------------------------------------------------------------------------
// The whole inner loop of a row-separable stack (decomposition.py:259-285 / 326-338), one wave per tile, four rows in
// flight per wave.  rhs64: X C (mode 1: scaled by a_i here) or R (mode 2), fp64; Linv64: [n_slabs, r, r].
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_wide_rowsep(WideRows W, const double *__restrict__ rhs64, const float *__restrict__ Arows,
                                                      const double *__restrict__ Linv64, float *__restrict__ F, RegSet regs,
                                                      int r, int inner, double *__restrict__ diag_tile) {
    // one workgroup per tile, one wave per sub-group of its rows; diag_tile: the tile's row of the mode's diagnostics table
    // (||F||^2, sum |F|, ||Z_k - F||^2 of the rounded values the kernel stores - what k_rows_diag would read back)
    MCL_GATE(W.gate);
    constexpr int RW = WIDE_RW;
    extern __shared__ double wsm[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, n_waves = blockDim.x >> 6;
    double *Ls = wsm, *ts = wsm + r * r + wave * (RW * 64), *dsm = wsm + r * r + n_waves * (RW * 64);  // dsm: [n_waves][DIAG_COLS]
    const RowGroup R(W, r, true);
    for (int e = threadIdx.x; e < r * r; e += blockDim.x) Ls[e] = Linv64[(long)R.slab * r * r + e];
    __syncthreads();
    const double rho = (double)W.rho[R.slab];
    const double a_c = (Arows != nullptr && R.act) ? (double)Arows[(long)R.slab * r + R.c] : 1.0;
    const int n = regs.n;
    double thr[MCL_MAX_REGS];
#pragma unroll
    for (int k = 0; k < MCL_MAX_REGS; ++k) thr[k] = (k < n) ? regs.p0d[k] / rho : 0.0;
    double rhs[RW], z[MCL_MAX_REGS][RW], u[MCL_MAX_REGS][RW], f[RW], t[RW];
    bool ok[RW];
    long idx[RW];
#pragma unroll
    for (int q = 0; q < RW; ++q) {
        idx[q] = R.idx(q, ok[q]);
        rhs[q] = ok[q] ? rhs64[idx[q]] * a_c : 0.0;
#pragma unroll
        for (int k = 0; k < MCL_MAX_REGS; ++k) {
            z[k][q] = (k < n && ok[q]) ? (double)regs.aux[k][idx[q]] : 0.0;
            u[k][q] = (k < n && ok[q]) ? (double)regs.dual[k][idx[q]] : 0.0;
        }
        f[q] = 0.0;
    }
    if (R.any()) {  // (wave-uniform: a wave past the tile's end only takes part in the barriers)
        for (int it = 0; it < inner; ++it) {
#pragma unroll
            for (int q = 0; q < RW; ++q) {
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < MCL_MAX_REGS; ++k)
                    if (k < n) s += z[k][q] - u[k][q];
                t[q] = fma(rho, s, rhs[q]);
            }
            row_times_matrix<RW>(t, f, ts, Ls, R);
#pragma unroll
            for (int k = 0; k < MCL_MAX_REGS; ++k)
                if (k < n) {
#pragma unroll
                    for (int q = 0; q < RW; ++q) {
                        const double zn = prox_rowsep_d(regs.kind[k], regs.nonneg[k], regs.p0d[k], regs.p1d[k], thr[k], f[q] + u[k][q]);
                        u[k][q] = f[q] - (zn - u[k][q]);
                        z[k][q] = zn;
                    }
                }
        }
    }
    double nf = 0.0, na = 0.0, gap[MCL_MAX_REGS];
#pragma unroll
    for (int k = 0; k < MCL_MAX_REGS; ++k) gap[k] = 0.0;
#pragma unroll
    for (int q = 0; q < RW; ++q)
        if (ok[q]) {
            const float ff = (float)f[q];
            F[idx[q]] = ff;
            nf = fma((double)ff, (double)ff, nf);
            na += fabs((double)ff);
#pragma unroll
            for (int k = 0; k < MCL_MAX_REGS; ++k)
                if (k < n) {
                    const float zf = (float)z[k][q];
                    regs.aux[k][idx[q]] = zf;
                    regs.dual[k][idx[q]] = (float)u[k][q];
                    const double dlt = (double)zf - (double)ff;
                    gap[k] = fma(dlt, dlt, gap[k]);
                }
        }
    nf = wave_sum(nf), na = wave_sum(na);
#pragma unroll
    for (int k = 0; k < MCL_MAX_REGS; ++k) gap[k] = wave_sum(gap[k]);
    if (lane == 0) {
        double *o = dsm + wave * DIAG_COLS;
        o[0] = nf, o[1] = na;
#pragma unroll
        for (int k = 0; k < MCL_MAX_REGS; ++k) o[2 + k] = gap[k];
    }
    __syncthreads();
    if ((int)threadIdx.x < DIAG_COLS) {
        double tsum = 0.0;
        for (int wv = 0; wv < n_waves; ++wv) tsum += dsm[wv * DIAG_COLS + threadIdx.x];  // fixed order
        diag_tile[(long)blockIdx.x * DIAG_COLS + threadIdx.x] = tsum;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Generic stacks: the state of the phase in fp64 shadow arrays.  Every kernel that writes a shadow value also writes its
// rounding into the caller's fp32 buffer, so the buffers are current whenever the phase ends.
// ---------------------------------------------------------------------------------------------------------
struct WideLoad {
    int n;
    const float *src[2 * MCL_MAX_REGS + 1];
    double *dst[2 * MCL_MAX_REGS + 1];
    long count[2 * MCL_MAX_REGS + 1];
};

__global__ __launch_bounds__(256) void k_wide_load(WideLoad L) {
    for (int a = 0; a < L.n; ++a)
        for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < L.count[a]; e += (long)gridDim.x * 256) L.dst[a][e] = (double)L.src[a][e];
}

struct WideState {
    double *F;                 // [rows, r]
    double *Z[MCL_MAX_REGS];   // auxiliary variables (PARAFAC2: the bases P)
    double *U[MCL_MAX_REGS];
    double *D;                 // PARAFAC2: coordinate matrix [r, r]
    int kpf2;
};

// F = (rhs (o a) + rho sum_k (Z_k - U_k)) L^-1, Z_k = P Delta for the PARAFAC2 member  (decomposition.py:266-273 / 328-331)
__global__ __launch_bounds__(64) void k_wide_solve(WideRows W, WideState S, const double *__restrict__ rhs64,
                                                   const float *__restrict__ Arows, const double *__restrict__ Linv64,
                                                   float *__restrict__ F32, int n, int r, double *__restrict__ change_part) {
    // change_part != nullptr (inner stopping test): per workgroup ||f_new - f_old||^2, f_old = the shadow factor before this solve
    MCL_GATE(W.gate);
    constexpr int RW = WIDE_RW;
    extern __shared__ double wsm[];
    double *Ls = wsm, *Ds = Ls + r * r, *ts = Ds + (S.kpf2 >= 0 ? r * r : 0);
    const RowGroup R(W, r);
    if (!R.any()) {
        // a workgroup past the end of its tile: k_inner_check still sums EVERY entry of the table, which modes 0 / 1 / 2 and
        // the non-wide checked loop share - a stale value of another phase would inflate this one's change term
        if (change_part != nullptr && threadIdx.x == 0) change_part[blockIdx.x] = 0.0;
        return;
    }
    for (int e = threadIdx.x; e < r * r; e += 64) {
        Ls[e] = Linv64[(long)R.slab * r * r + e];
        if (S.kpf2 >= 0) Ds[e] = S.D[e];
    }
    __syncthreads();
    const double rho = (double)W.rho[R.slab];
    const double a_c = (Arows != nullptr && R.act) ? (double)Arows[(long)R.slab * r + R.c] : 1.0;
    double t[RW], f[RW], sacc[RW];
    bool ok[RW];
    long idx[RW];
#pragma unroll
    for (int q = 0; q < RW; ++q) {
        idx[q] = R.idx(q, ok[q]);
        sacc[q] = 0.0;
    }
    for (int k = 0; k < n; ++k) {
        double z[RW];
#pragma unroll
        for (int q = 0; q < RW; ++q) z[q] = ok[q] ? S.Z[k][idx[q]] : 0.0;
        if (k == S.kpf2) {
            double pd[RW];
            row_times_matrix<RW>(z, pd, ts, Ds, R);
#pragma unroll
            for (int q = 0; q < RW; ++q) z[q] = pd[q];
        }
#pragma unroll
        for (int q = 0; q < RW; ++q) sacc[q] += z[q] - (ok[q] ? S.U[k][idx[q]] : 0.0);
    }
#pragma unroll
    for (int q = 0; q < RW; ++q) t[q] = ok[q] ? fma(rho, sacc[q], rhs64[idx[q]] * a_c) : 0.0;
    row_times_matrix<RW>(t, f, ts, Ls, R);
    double chg = 0.0;
#pragma unroll
    for (int q = 0; q < RW; ++q)
        if (ok[q]) {
            if (change_part != nullptr) {
                const double d = f[q] - S.F[idx[q]];
                chg = fma(d, d, chg);
            }
            S.F[idx[q]] = f[q];
            F32[idx[q]] = (float)f[q];
        }
    if (change_part != nullptr) {
        chg = wave_sum(chg);
        if (threadIdx.x == 0) change_part[blockIdx.x] = chg;
    }
}

// mode 0 (constant feasibility penalty, matrix penalties on A: decomposition.py:184-195): every row has its own system -
// one wave per row of A, lane c owns column c
__global__ __launch_bounds__(64) void k_wide_A_solve(WideState S, const double *__restrict__ rhsA64, const float *__restrict__ rho_max,
                                                     const double *__restrict__ LinvA64, float *__restrict__ A32, int n, int r,
                                                     const int *__restrict__ gate, double *__restrict__ change_part) {
    MCL_GATE(gate);
    __shared__ double tS[64];
    const int i = blockIdx.x, c = threadIdx.x;
    const bool act = c < r;
    const long e = (long)i * r + (act ? c : 0);
    const double rho = (double)rho_max[1];
    double s = 0.0;
    for (int k = 0; k < n; ++k) s += act ? S.Z[k][e] - S.U[k][e] : 0.0;
    tS[c] = act ? fma(rho, s, rhsA64[e]) : 0.0;
    __syncthreads();
    double a = 0.0;
    for (int d = 0; d < r; ++d) a = fma(tS[d], act ? LinvA64[((long)i * r + d) * r + c] : 0.0, a);
    if (change_part != nullptr) {
        const double d = act ? a - S.F[e] : 0.0;
        const double sq = wave_sum(d * d);
        if (c == 0) change_part[i] = sq;
    }
    if (act) {
        S.F[e] = a;
        A32[e] = (float)a;
    }
}

// The sums of the inner stopping test (decomposition.py:100-116) from the fp64 state, one table row per workgroup in the
// layout of the diagnostics tables (column 0: ||F||^2, columns 2 + k: ||Z_k - F||^2, Z_k = P Delta for the PARAFAC2 member):
// k_inner_check (generic.hip) takes it from there
__global__ __launch_bounds__(64) void k_wide_sums(WideRows W, WideState S, int n, int r, double *__restrict__ tab) {
    MCL_GATE(W.gate);
    constexpr int RW = WIDE_RW;
    extern __shared__ double wsm[];
    double *Ds = wsm, *ts = wsm + (S.kpf2 >= 0 ? r * r : 0);
    const RowGroup R(W, r);
    double *row = tab + (long)blockIdx.x * DIAG_COLS;
    if (!R.any()) {
        if (threadIdx.x < DIAG_COLS) row[threadIdx.x] = 0.0;
        return;
    }
    if (S.kpf2 >= 0)
        for (int e = threadIdx.x; e < r * r; e += 64) Ds[e] = S.D[e];
    __syncthreads();
    double f[RW], nf = 0.0, gap[MCL_MAX_REGS];
    bool ok[RW];
    long idx[RW];
#pragma unroll
    for (int q = 0; q < RW; ++q) {
        idx[q] = R.idx(q, ok[q]);
        f[q] = ok[q] ? S.F[idx[q]] : 0.0;
        nf = fma(f[q], f[q], nf);
    }
#pragma unroll
    for (int k = 0; k < MCL_MAX_REGS; ++k) gap[k] = 0.0;
    for (int k = 0; k < n; ++k) {
        double z[RW];
#pragma unroll
        for (int q = 0; q < RW; ++q) z[q] = ok[q] ? S.Z[k][idx[q]] : 0.0;
        if (k == S.kpf2) {
            double pd[RW];
            row_times_matrix<RW>(z, pd, ts, Ds, R);
#pragma unroll
            for (int q = 0; q < RW; ++q) z[q] = pd[q];
        }
        double g = 0.0;
#pragma unroll
        for (int q = 0; q < RW; ++q)
            if (ok[q]) g = fma(z[q] - f[q], z[q] - f[q], g);
#pragma unroll
        for (int kk = 0; kk < MCL_MAX_REGS; ++kk)
            if (kk == k) gap[kk] = g;
    }
    nf = wave_sum(nf);
#pragma unroll
    for (int k = 0; k < MCL_MAX_REGS; ++k) gap[k] = wave_sum(gap[k]);
    if (threadIdx.x == 0) {
        row[0] = nf, row[1] = 0.0;
#pragma unroll
        for (int k = 0; k < MCL_MAX_REGS; ++k) row[2 + k] = gap[k];
    }
}

// row-separable member of a generic stack: prox + dual step of penalty k, elementwise
__global__ __launch_bounds__(256) void k_wide_prox_rowsep(WideRows W, WideState S, RegSet regs, int k, int r, long rows,
                                                          const int *__restrict__ slab_of_row) {
    MCL_GATE(W.gate);
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < rows * r; e += (long)gridDim.x * 256) {
        const long j = e / r;
        const double rho = (double)W.rho[slab_of_row ? slab_of_row[j] : 0];
        const double f = S.F[e], u = S.U[k][e];
        const double z = prox_rowsep_d(regs.kind[k], regs.nonneg[k], regs.p0d[k], regs.p1d[k], regs.p0d[k] / rho, f + u);
        const double un = f - (z - u);
        S.Z[k][e] = z, S.U[k][e] = un;
        regs.aux[k][e] = (float)z, regs.dual[k][e] = (float)un;
    }
}

// plain dual step U = F - (Z - U) of penalty k (after a prox that wrote Z: unimodality, total variation)
__global__ __launch_bounds__(256) void k_wide_dual(WideRows W, WideState S, RegSet regs, int k, long count) {
    MCL_GATE(W.gate);
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < count; e += (long)gridDim.x * 256) {
        const double un = S.F[e] - (S.Z[k][e] - S.U[k][e]);
        S.U[k][e] = un;
        regs.dual[k][e] = (float)un;
    }
}

// L2 ball (penalties.py:920-925): column sums of squares of Y = F + U (clipped at 0 first when non-negative) per slab,
// rows in ascending order per thread and the thread partials in a fixed order
__global__ __launch_bounds__(256) void k_wide_colsq(WideRows W, WideState S, int k, int nonneg, int r, double *__restrict__ colsq) {
    MCL_GATE(W.gate);
    __shared__ double sm[256];
    const int slab = blockIdx.x;
    const int s = W.ext[slab], e = W.ext[slab + 1];
    const int col = threadIdx.x & 63, rl = threadIdx.x >> 6;
    double acc = 0.0;
    if (col < r) {
        // eight rows' loads in flight per step (round 6: one dependent load pair per row made this kernel 40 us on the 576-row
        // matrices of a mid-size problem - the exact arithmetic serves up to 2^24 elements since the condition monitor); the
        // squares are added in the order of the rows, as before
        constexpr int UB = 8;
        for (long j0 = (long)s + rl; j0 < e; j0 += 4 * UB) {
            double yv[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const long j = min(j0 + 4 * u, (long)e - 1);
                yv[u] = S.F[j * r + col] + S.U[k][j * r + col];
            }
#pragma unroll
            for (int u = 0; u < UB; ++u)
                if (j0 + 4 * u < e) {
                    double y = yv[u];
                    if (nonneg) y = fmax(y, 0.0);
                    acc = fma(y, y, acc);
                }
        }
    }
    sm[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < 64 && col < r) colsq[(long)slab * r + col] = (sm[col] + sm[64 + col]) + (sm[128 + col] + sm[192 + col]);
}

__global__ __launch_bounds__(256) void k_wide_l2ball(WideRows W, WideState S, RegSet regs, int k, int r, long rows,
                                                     const int *__restrict__ slab_of_row, const double *__restrict__ colsq) {
    MCL_GATE(W.gate);
    const double bound = regs.p0d[k];
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < rows * r; e += (long)gridDim.x * 256) {
        const long j = e / r;
        const int col = (int)(e - j * r), slab = slab_of_row ? slab_of_row[j] : 0;
        const double f = S.F[e], u = S.U[k][e];
        double y = f + u;
        if (regs.nonneg[k]) y = fmax(y, 0.0);
        const double z = y / fmax(sqrt(colsq[(long)slab * r + col]), bound) * bound;
        const double un = f - (z - u);
        S.Z[k][e] = z, S.U[k][e] = un;
        regs.aux[k][e] = (float)z, regs.dual[k][e] = (float)un;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Unimodal regression of every column of a slab (penalties.py:1014-1015 -> _unimodal_regression.py:27-104), one lane per
// (slab, column), everything in fp64: two prefix-isotonic sweeps (Stout's pooling with per-position block records: level,
// start, sum, sum of squares), the split with the smallest total error (first one on ties, scanned from the left), the fits
// read off the records.  Scratch: ten per-position arrays of the lane (column-interleaved like the factor itself).
// ---------------------------------------------------------------------------------------------------------
struct UniWide {
    double *lv[2], *st[2], *sy[2], *s2[2], *er[2];  // [0]: forward sweep, [1]: sweep over the reversed column; er has n + 1 entries
};

__global__ __launch_bounds__(64) void k_wide_unimodal(WideRows W, WideState S, RegSet regs, int k, int r, UniWide Q) {
    MCL_GATE(W.gate);
    const long t = (long)blockIdx.x * 64 + threadIdx.x;
    if (t >= (long)W.n_slabs * r) return;
    const int slab = (int)(t / r), col = (int)(t - (long)slab * r);
    const long s = W.ext[slab];
    const int n = W.ext[slab + 1] - W.ext[slab];
    if (n <= 0) return;
    const bool nn = regs.nonneg[k] != 0;
    // position i of the column <-> element (s + i) * r + col; the error arrays carry one more entry per slab: shifted by slab
    auto at = [&](long i) { return (s + i) * r + col; };
    auto ate = [&](long i) { return (s + slab + i) * r + col; };
    for (int dir = 0; dir < 2; ++dir) {
        double *lv = Q.lv[dir], *stt = Q.st[dir], *sy = Q.sy[dir], *s2 = Q.s2[dir], *er = Q.er[dir];
        double cum2 = 0.0;
        er[ate(0)] = 0.0;
        for (int i = 0; i < n; ++i) {
            const long src = at(dir == 0 ? i : n - 1 - i);
            const double y = S.F[src] + S.U[k][src];
            cum2 += y * y;
            double level = y, sumy = y, sum2 = y * y;
            int start = i;
            while (start != 0 && level <= lv[at(start - 1)]) {
                const int p = start - 1;
                sumy += sy[at(p)];
                sum2 += s2[at(p)];
                start = (int)stt[at(p)];
                level = sumy / (double)(i - start + 1);
            }
            lv[at(i)] = level, stt[at(i)] = (double)start, sy[at(i)] = sumy, s2[at(i)] = sum2;
            if (nn && level < 0.0) er[ate(i + 1)] = cum2;
            else er[ate(i + 1)] = (sum2 - sumy * sumy / (double)(i - start + 1)) + er[ate(start)];
        }
    }
    int split = 0;
    double best = Q.er[1][ate(n)];
    for (int i = 0; i <= n; ++i) {
        const double e = Q.er[0][ate(i)] + Q.er[1][ate(n - i)];
        if (e < best) best = e, split = i;
    }
    // fit of the prefix y[:split] from the forward records, of the suffix from the reversed ones
    for (int idx = split - 1; idx >= 0;) {
        const int start = (int)Q.st[0][at(idx)];
        double level = Q.lv[0][at(idx)];
        if (nn && level < 0.0) level = 0.0;
        for (int i = start; i <= idx; ++i) {
            S.Z[k][at(i)] = level;
            regs.aux[k][at(i)] = (float)level;
        }
        idx = start - 1;
    }
    for (int idx = n - split - 1; idx >= 0;) {
        const int start = (int)Q.st[1][at(idx)];
        double level = Q.lv[1][at(idx)];
        if (nn && level < 0.0) level = 0.0;
        for (int i = start; i <= idx; ++i) {  // reversed position i <-> position n - 1 - i
            S.Z[k][at(n - 1 - i)] = level;
            regs.aux[k][at(n - 1 - i)] = (float)level;
        }
        idx = start - 1;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Total variation (penalties.py:750-841; L. Condat's direct algorithm as in generic.hip: k_slab_tv) on the fp64 state
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_wide_tv(WideRows W, WideState S, RegSet regs, int kreg, int r) {
    MCL_GATE(W.gate);
    const long t = (long)blockIdx.x * 64 + threadIdx.x;
    if (t >= (long)W.n_slabs * r) return;
    const int slab = (int)(t / r), col = (int)(t - (long)slab * r);
    const long s = W.ext[slab];
    const int n = W.ext[slab + 1] - W.ext[slab];
    if (n <= 0) return;
    const double rho = (double)W.rho[slab];
    const double lam = 2.0 * regs.p0d[kreg] / rho;
    const double l1 = regs.p1d[kreg] / rho;
    auto in = [&](int k) -> double { return S.F[(s + k) * r + col] + S.U[kreg][(s + k) * r + col]; };
    auto emit = [&](int a, int b, double v) {
        double z = v;
        if (l1 > 0.0) z = copysign(fmax(fabs(z) - l1, 0.0), z);
        for (int k = a; k <= b; ++k) {
            S.Z[kreg][(s + k) * r + col] = z;
            regs.aux[kreg][(s + k) * r + col] = (float)z;
        }
    };
    int k = 0, k0 = 0, km = 0, kp = 0;
    double vmin = in(0) - lam, vmax = in(0) + lam, umin = lam, umax = -lam;
    for (;;) {
        if (k == n - 1) {
            if (umin < 0.0) {
                emit(k0, km, vmin);
                k0 = km + 1;
                k = km = k0;
                vmin = in(k);
                umin = lam;
                umax = vmin + lam - vmax;
            } else if (umax > 0.0) {
                emit(k0, kp, vmax);
                k0 = kp + 1;
                k = kp = k0;
                vmax = in(k);
                umax = -lam;
                umin = vmax - lam - vmin;
            } else {
                vmin += umin / (double)(k - k0 + 1);
                emit(k0, k, vmin);
                return;
            }
        } else {
            const double xn = in(k + 1);
            umin += xn - vmin;
            umax += xn - vmax;
            if (umin < -lam) {
                emit(k0, km, vmin);
                k0 = km + 1;
                k = km = kp = k0;
                vmin = in(k);
                vmax = vmin + 2.0 * lam;
                umin = lam, umax = -lam;
            } else if (umax > lam) {
                emit(k0, kp, vmax);
                k0 = kp + 1;
                k = km = kp = k0;
                vmax = in(k);
                vmin = vmax - 2.0 * lam;
                umin = lam, umax = -lam;
            } else {
                ++k;
                if (umin >= lam) {
                    km = k;
                    vmin += (umin - lam) / (double)(km - k0 + 1);
                    umin = lam;
                }
                if (umax <= -lam) {
                    kp = k;
                    vmax += (umax + lam) / (double)(kp - k0 + 1);
                    umax = -lam;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// PARAFAC2 (penalties.py:1224-1250, 1280-1281) on the fp64 state: S_i = Y_i^T Y_i, Y = F + U; the polar factors through the
// Jacobi / QR kernels of generic.hip (fp64 throughout; mcl_launch_pf2_jacobi_wide); P_i = Y_i T_i; Delta; dual step
// ---------------------------------------------------------------------------------------------------------
constexpr int WIDE_GRAM_ROWS = 64;
__global__ __launch_bounds__(256) void k_wide_gram(WideRows W, WideState S, int k, int r, double *__restrict__ Sout) {
    MCL_GATE(W.gate);
    extern __shared__ double wsm[];  // WIDE_GRAM_ROWS rows x r of Y
    const int slab = blockIdx.x;
    const long s = W.ext[slab];
    const int n = W.ext[slab + 1] - W.ext[slab];
    const int n2 = r * r;
    constexpr int PER = 16;  // (a, b) pairs per thread: r <= 64
    double acc[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) acc[q] = 0.0;
    for (int j0 = 0; j0 < n; j0 += WIDE_GRAM_ROWS) {  // (64 rows per stage since round 6: a quarter of the barriers and load round trips; same order of the sums)
        const int nr = min(WIDE_GRAM_ROWS, n - j0);
        __syncthreads();
        for (int e = threadIdx.x; e < nr * r; e += 256) wsm[e] = S.F[(s + j0) * r + e] + S.U[k][(s + j0) * r + e];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int e = threadIdx.x + 256 * q;
            if (e < n2) {
                const int a = e / r, b = e - a * r;
                double v = acc[q];
                for (int j = 0; j < nr; ++j) v = fma(wsm[j * r + a], wsm[j * r + b], v);
                acc[q] = v;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int e = threadIdx.x + 256 * q;
        if (e < n2) Sout[(long)slab * n2 + e] = acc[q];
    }
}

// P = Y T_slab (T in fp64), one wave per tile
__global__ __launch_bounds__(64) void k_wide_pf2_apply(WideRows W, WideState S, int k, int r, const double *__restrict__ T64,
                                                       float *__restrict__ P32) {
    MCL_GATE(W.gate);
    constexpr int RW = WIDE_RW;
    extern __shared__ double wsm[];
    double *Ts = wsm, *ts = wsm + r * r;
    const RowGroup R(W, r);
    if (!R.any()) return;
    for (int e = threadIdx.x; e < r * r; e += 64) Ts[e] = T64[(long)R.slab * r * r + e];
    __syncthreads();
    double y[RW], p[RW];
    bool ok[RW];
    long idx[RW];
#pragma unroll
    for (int q = 0; q < RW; ++q) {
        idx[q] = R.idx(q, ok[q]);
        y[q] = ok[q] ? S.F[idx[q]] + S.U[k][idx[q]] : 0.0;
    }
    row_times_matrix<RW>(y, p, ts, Ts, R);
#pragma unroll
    for (int q = 0; q < RW; ++q)
        if (ok[q]) {
            S.Z[k][idx[q]] = p[q];
            P32[idx[q]] = (float)p[q];
        }
}

// Delta = sum_i rho_i P_i^T Y_i / sum_i rho_i from the per-slab accumulators of the polar-factor kernels (fixed order)
__global__ __launch_bounds__(256) void k_wide_pf2_delta(const double *__restrict__ acc, int n_slabs, int n2, double *__restrict__ D64,
                                                        float *__restrict__ D32, float *__restrict__ red, const int *__restrict__ gate) {
    MCL_GATE(gate);
    __shared__ double sm[2][4];
    const int e = blockIdx.x, n_el = n2 + 1;
    double s = 0.0, w = 0.0;
    for (int i = threadIdx.x; i < n_slabs; i += 256) {
        s += acc[(long)i * n_el + e];
        w += acc[(long)i * n_el + n2];
    }
    s = wave_sum(s);
    w = wave_sum(w);
    if ((threadIdx.x & 63) == 0) sm[0][threadIdx.x >> 6] = s, sm[1][threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double re = (sm[0][0] + sm[0][1]) + (sm[0][2] + sm[0][3]);
        const double rw = (sm[1][0] + sm[1][1]) + (sm[1][2] + sm[1][3]);
        red[e] = (float)re;
        if (e == 0) red[n2] = (float)rw;
        const double d = re / rw;
        D64[e] = d;
        D32[e] = (float)d;
    }
}

// dual step of the PARAFAC2 member: U = F - (P Delta - U), one wave per tile
__global__ __launch_bounds__(64) void k_wide_pf2_dual(WideRows W, WideState S, RegSet regs, int k, int r) {
    MCL_GATE(W.gate);
    constexpr int RW = WIDE_RW;
    extern __shared__ double wsm[];
    double *Ds = wsm, *ts = wsm + r * r;
    const RowGroup R(W, r);
    if (!R.any()) return;
    for (int e = threadIdx.x; e < r * r; e += 64) Ds[e] = S.D[e];
    __syncthreads();
    double p[RW], z[RW];
    bool ok[RW];
    long idx[RW];
#pragma unroll
    for (int q = 0; q < RW; ++q) {
        idx[q] = R.idx(q, ok[q]);
        p[q] = ok[q] ? S.Z[k][idx[q]] : 0.0;
    }
    row_times_matrix<RW>(p, z, ts, Ds, R);
#pragma unroll
    for (int q = 0; q < RW; ++q)
        if (ok[q]) {
            const double un = S.F[idx[q]] - (z[q] - S.U[k][idx[q]]);
            S.U[k][idx[q]] = un;
            regs.dual[k][idx[q]] = (float)un;
        }
}

WideRows rows_of(const mcl_context *c, int mode) {
    WideRows W{};
    const TileMap &tm = (mode == 1) ? c->tilesB : (mode == 2 ? c->tilesC : c->tilesA);
    W.tile_slab = tm.slab, W.tile_row0 = tm.row0, W.tile_nrows = tm.nrows, W.n_tiles = tm.n_tiles;
    if (mode == 1) W.ext = c->row_ptr_dev, W.n_slabs = (int)c->I, W.rho = c->rhoB;
    else if (mode == 2) W.ext = c->ext_C, W.n_slabs = 1, W.rho = c->rhoC;
    else W.ext = c->ext_A, W.n_slabs = 1, W.rho = c->rho_max + 1;  // mode 0: one "slab" of I rows, the constant rho
    W.gate = c->gate_active;
    return W;
}

inline unsigned blocks_for(long n) { return (unsigned)std::min<long>(std::max<long>((n + 255) / 256, 1), 4096); }

}  // namespace

// A problem in the exact-products mode, every penalty of the mode native: the fp64 inner loop applies to modes 1 and 2, and to
// mode 0 where it runs an un-fused loop at all (matrix penalties on A under a constant feasibility penalty; row-separable
// stacks on A stay in k_A_finish*, whose inner loop is fp64 already)
bool mcl_wide_applies(const mcl_context *c, int mode) {
    if (!c->exact || c->sw.no_wide || mode < 0 || mode > 2) return false;
    const RegSet &rs = c->regs[mode];
    if (rs.n == 0 || c->opt.inner_n_iter_max <= 0 || c->wF[mode] == nullptr) return false;
    if (c->opt.inner_tol > 0.0 && c->wide_tab == nullptr) return false;
    if (mode == 0 && (!c->opt.constant_A || c->LinvA64 == nullptr)) return false;
    for (int k = 0; k < rs.n; ++k)
        if (rs.kind[k] == MCL_PEN_EXTERNAL) return false;
    return true;
}

// The inner loop of mode 1 (after mcl_B_begin / mcl_B_factor: XC64, rho_i, the fp64 inverses) or mode 2 (after
// mcl_launch_C_prepare: rho, the fp64 inverse; R is the fp64 [G | R] itself).
int mcl_wide_phase(mcl_context *c, int mode) {
    const RegSet &rs = c->regs[mode];
    const int r = c->r, n = rs.n, n_it = c->opt.inner_n_iter_max;
    const long rows = (mode == 1) ? (long)c->N : (mode == 2 ? (long)c->K : (long)c->I);
    if (rows == 0) return 0;
    WideRows W = rows_of(c, mode);
    const double *rhs64 = (mode == 1) ? c->XC64 : (mode == 2 ? c->GR + (long)r * r : c->rhsA64);
    const float *Arows = (mode == 1) ? c->A : nullptr;
    const double *Linv64 = (mode == 1) ? c->LinvB64 : (mode == 2 ? c->LinvC64 : c->LinvA64);
    float *F32 = (mode == 1) ? c->B : (mode == 2 ? c->C : c->A);
    const int *slab_of_row = (mode == 1) ? c->slab_of_row : nullptr;
    bool rowsep = true;
    int kpf2 = -1;
    for (int k = 0; k < n; ++k) {
        const int kind = rs.kind[k];
        if (kind != MCL_PEN_NN && kind != MCL_PEN_BOX && kind != MCL_PEN_L1) rowsep = false;
        if (kind == MCL_PEN_PARAFAC2) kpf2 = k;
    }
    ProfScope prof(c, MCL_PROF_ROWS_FUSED);
    const size_t sm_one = sizeof(double) * (size_t)(r * r + 4 * 64);
    const bool checked = c->opt.inner_tol > 0.0;  // the inner stopping test (decomposition.py:90-117): one kernel per step
    if (rowsep && mode != 0 && !checked) {
        c->variant[MCL_PROF_ROWS_FUSED] = "k_wide_rowsep (fp64 inner loop)";
        const int subs = wide_subs(r);
        const size_t sm_rs = sizeof(double) * (size_t)(r * r + subs * (WIDE_RW * 64) + subs * DIAG_COLS);
        if (sm_rs > 65536)
            MCL_CHECK_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_wide_rowsep), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm_rs));
        double *diag = (mode == 1) ? c->diagB_tile : c->diagC_tile;
        hipLaunchKernelGGL(k_wide_rowsep, dim3((unsigned)W.n_tiles), dim3(64 * subs), sm_rs, c->stream, W, rhs64, Arows, Linv64, F32, rs, r,
                           n_it, diag);
        MCL_CHECK_HIP(c, hipGetLastError());
        c->diag_rows[mode] = W.n_tiles;
        c->diag_valid[mode] = true;  // the kernel left the mode's diagnostics table
        return 0;
    }
    c->variant[MCL_PROF_ROWS_FUSED] = "k_wide_* (fp64 state, one kernel per step)";
    WideState S{};
    S.F = c->wF[mode], S.D = c->wD, S.kpf2 = kpf2;
    WideLoad L{};
    for (int k = 0; k < n; ++k) {
        S.Z[k] = c->wZ[mode][k], S.U[k] = c->wU[mode][k];
        L.src[L.n] = rs.aux[k], L.dst[L.n] = S.Z[k], L.count[L.n++] = rows * r;
        L.src[L.n] = rs.dual[k], L.dst[L.n] = S.U[k], L.count[L.n++] = rows * r;
    }
    if (kpf2 >= 0) L.src[L.n] = rs.aux2[kpf2], L.dst[L.n] = S.D, L.count[L.n++] = (long)r * r;
    const int *run_gate = c->gate_active, *reg_gate = c->regs[mode].gate;
    RegSet rs_gated = rs;  // (the kernels below take the penalty list by value)
    if (checked) {
        // the factor too (||f_new - f_old|| of the first iteration), and the phase's own stop flag instead of the run's
        L.src[L.n] = F32, L.dst[L.n] = S.F, L.count[L.n++] = rows * r;
        if (int rc = mcl_launch_inner_check_raw(c, true, nullptr, 0, nullptr, 0, 0)) return rc;
        c->gate_active = c->inner_gate, c->regs[mode].gate = c->inner_gate;
        W.gate = c->inner_gate, rs_gated.gate = c->inner_gate;
    }
    struct GateRestore {  // every return path below hands the run's gate back
        mcl_context *c;
        int mode;
        const int *run_gate, *reg_gate;
        ~GateRestore() { c->gate_active = run_gate, c->regs[mode].gate = reg_gate; }
    } restore{c, mode, run_gate, reg_gate};
    const RegSet &rsk = rs_gated;
    hipLaunchKernelGGL(k_wide_load, dim3(blocks_for(rows * r)), dim3(256), 0, c->stream, L);
    const size_t sm_two = sizeof(double) * (size_t)(2 * r * r + 4 * 64);
    if (sm_two > 65536) {
        MCL_CHECK_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_wide_solve), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm_two));
    }
    const long n_cols = (long)W.n_slabs * r;
    UniWide Q{};
    if (c->uni_f64 != nullptr) {
        const int64_t maxrows = std::max<int64_t>(c->N, std::max<int64_t>(c->I, c->K));
        const int64_t n1 = (maxrows + std::max<int64_t>(c->I, 1)) * c->r;
        double *d = c->uni_f64;
        for (int dir = 0; dir < 2; ++dir)
            Q.lv[dir] = d + (5 * dir + 0) * n1, Q.st[dir] = d + (5 * dir + 1) * n1, Q.sy[dir] = d + (5 * dir + 2) * n1,
            Q.s2[dir] = d + (5 * dir + 3) * n1, Q.er[dir] = d + (5 * dir + 4) * n1;
    }
    for (int it = 0; it < n_it; ++it) {
        if (mode == 0)
            hipLaunchKernelGGL(k_wide_A_solve, dim3((unsigned)rows), dim3(64), 0, c->stream, S, rhs64, (const float *)c->rho_max, Linv64, F32,
                               n, r, c->gate_active, checked ? c->inner_part : nullptr);
        else
            hipLaunchKernelGGL(k_wide_solve, dim3(wide_grid(W, r)), dim3(64), kpf2 >= 0 ? sm_two : sm_one, c->stream, W, S, rhs64, Arows,
                               Linv64, F32, n, r, checked ? c->inner_part : nullptr);
        for (int k = 0; k < n; ++k) {
            switch (rs.kind[k]) {
                case MCL_PEN_NN:
                case MCL_PEN_BOX:
                case MCL_PEN_L1:
                    hipLaunchKernelGGL(k_wide_prox_rowsep, dim3(blocks_for(rows * r)), dim3(256), 0, c->stream, W, S, rsk, k, r, rows, slab_of_row);
                    break;
                case MCL_PEN_L2BALL:
                    hipLaunchKernelGGL(k_wide_colsq, dim3((unsigned)W.n_slabs), dim3(256), 0, c->stream, W, S, k, rs.nonneg[k], r, c->colsq);
                    hipLaunchKernelGGL(k_wide_l2ball, dim3(blocks_for(rows * r)), dim3(256), 0, c->stream, W, S, rsk, k, r, rows, slab_of_row,
                                       (const double *)c->colsq);
                    break;
                case MCL_PEN_UNIMODAL:
                    if (c->uni_f64 == nullptr) {
                        c->err = "internal: unimodal scratch missing";
                        return 1;
                    }
                    hipLaunchKernelGGL(k_wide_unimodal, dim3((unsigned)((n_cols + 63) / 64)), dim3(64), 0, c->stream, W, S, rsk, k, r, Q);
                    hipLaunchKernelGGL(k_wide_dual, dim3(blocks_for(rows * r)), dim3(256), 0, c->stream, W, S, rsk, k, rows * r);
                    break;
                case MCL_PEN_TV:
                    hipLaunchKernelGGL(k_wide_tv, dim3((unsigned)((n_cols + 63) / 64)), dim3(64), 0, c->stream, W, S, rsk, k, r);
                    hipLaunchKernelGGL(k_wide_dual, dim3(blocks_for(rows * r)), dim3(256), 0, c->stream, W, S, rsk, k, rows * r);
                    break;
                case MCL_PEN_GL2:
                    if (int rc = mcl_launch_gl2_wide(c, mode, k, S.F, S.Z[k], S.U[k])) return rc;
                    break;
                case MCL_PEN_SIMPLEX:
                    if (int rc = mcl_launch_simplex_wide(c, mode, k, S.F, S.U[k], S.Z[k])) return rc;
                    hipLaunchKernelGGL(k_wide_dual, dim3(blocks_for(rows * r)), dim3(256), 0, c->stream, W, S, rsk, k, rows * r);
                    break;
                case MCL_PEN_PARAFAC2: {
                    if (mode != 1) {
                        c->err = "PARAFAC2 constraint can only be imposed with mode=1";
                        return 1;
                    }
                    hipLaunchKernelGGL(k_wide_gram, dim3((unsigned)c->I), dim3(256), sizeof(double) * WIDE_GRAM_ROWS * r, c->stream, W, S, k, r, c->pf2_S);
                    if (int rc = mcl_launch_pf2_jacobi_wide(c, k, S.F, S.U[k], S.D)) return rc;
                    hipLaunchKernelGGL(k_wide_pf2_apply, dim3(wide_grid(W, r)), dim3(64), sm_one, c->stream, W, S, k, r,
                                       (const double *)c->pf2_T64, rs.aux[k]);
                    hipLaunchKernelGGL(k_wide_pf2_delta, dim3((unsigned)(r * r)), dim3(256), 0, c->stream, (const double *)c->pf2_acc,
                                       (int)c->I, r * r, S.D, rs.aux2[k], c->pf2_red, c->gate_active);
                    hipLaunchKernelGGL(k_wide_pf2_dual, dim3(wide_grid(W, r)), dim3(64), sm_one, c->stream, W, S, rsk, k, r);
                    break;
                }
                default:
                    c->err = "penalty kind has no native prox (EXTERNAL penalties are evaluated by the host)";
                    return 1;
            }
        }
        if (checked) {
            hipLaunchKernelGGL(k_wide_sums, dim3(wide_grid(W, r)), dim3(64), sm_one, c->stream, W, S, n, r, c->wide_tab);
            if (int rc = mcl_launch_inner_check_raw(c, false, c->inner_part, mode == 0 ? (int)rows : (int)wide_grid(W, r), c->wide_tab,
                                                    (int)wide_grid(W, r), n))
                return rc;
        }
    }
    MCL_CHECK_HIP(c, hipGetLastError());
    return 0;
}
