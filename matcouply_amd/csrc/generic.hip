// Un-fused inner ADMM loop (one kernel per step) for penalties that couple the rows of a slab or the slabs:
//   L2Ball       penalties.py:920-925    column norms over the rows of one slab
//   Unimodality  penalties.py:1014-1015, _unimodal_regression.py:27-104   per-column prefix isotonic regression (fp64)
//   Parafac2     penalties.py:1224-1250, 1280-1281   polar factors per slab + cross-slab coordinate matrix
// plus the row-separable ones when they share a mode with the above.  Used for all three modes: A and C are
// treated as a single slab (tile maps tilesA / tilesC, extents ext_A / ext_C).
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "mcl_internal.h"
#include "rows_mfma.h"

static __device__ __forceinline__ double wave_sum_d(double v) { return wave_sum(v); }  // DPP + readlane (rows_mfma.h)

static __device__ __forceinline__ float prox_elem_g(int kind, int nonneg, float p0, float p1, float thr, float y) {
    switch (kind) {
        case MCL_PEN_NN: return fmaxf(y, 0.f);
        case MCL_PEN_BOX: return fminf(fmaxf(y, p0), p1);
        case MCL_PEN_L1:
            if (nonneg) return fmaxf(y - thr, 0.f);
            return copysignf(fmaxf(fabsf(y) - thr, 0.f), y);
        default: return y;
    }
}

static __device__ __forceinline__ double prox_elem_g(int kind, int nonneg, double p0, double p1, double thr, double y) {
    switch (kind) {
        case MCL_PEN_NN: return fmax(y, 0.0);
        case MCL_PEN_BOX: return fmin(fmax(y, p0), p1);
        case MCL_PEN_L1:
            if (nonneg) return fmax(y - thr, 0.0);
            return copysign(fmax(fabs(y) - thr, 0.0), y);
        default: return y;
    }
}

static ModeView view_of(mcl_context *c, int mode) {
    ModeView v{};
    v.gate = c->gate_active;
    const TileMap &tm = (mode == 1) ? c->tilesB : (mode == 2 ? c->tilesC : c->tilesA);
    v.tile_slab = tm.slab, v.tile_row0 = tm.row0, v.tile_nrows = tm.nrows, v.n_tiles = tm.n_tiles;
    if (mode == 1) {
        v.ext = c->row_ptr_dev, v.n_slabs = (int)c->I, v.rho = c->rhoB, v.F = c->B;
    } else if (mode == 2) {
        v.ext = c->ext_C, v.n_slabs = 1, v.rho = c->rhoC, v.F = c->C;
    } else {
        v.ext = c->ext_A, v.n_slabs = 1, v.rho = c->rho_max + 1, v.F = c->A;  // constant rho only
    }
    return v;
}

// ---------------------------------------------------------------------------------------------------------
// All row kernels below use the tile / fragment layout of rows_mfma.h (coalesced 16-B accesses); products with a
// wave-uniform r x r matrix (L^-1, Delta, T_i) run on the fp32 MFMA.
// ---------------------------------------------------------------------------------------------------------
#define TILE_PROLOGUE()                                                                                      \
    MCL_GATE(mv.gate);                                                                                       \
    const int lane = threadIdx.x & 63;                                                                       \
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);                                                    \
    if (tile >= mv.n_tiles) return;                                                                          \
    const int slab = __builtin_amdgcn_readfirstlane(mv.tile_slab[tile]);                                     \
    const long row0 = __builtin_amdgcn_readfirstlane(mv.tile_row0[tile]);                                    \
    const int nrows = __builtin_amdgcn_readfirstlane(mv.tile_nrows[tile]);                                   \
    const int row16 = lane & 15, g = lane >> 4;                                                              \
    (void)slab; (void)row16; (void)g

#define FOR_ROW_BLOCKS()                                                                                     \
    _Pragma("unroll") for (int rb = 0; rb < 4; ++rb)                                                         \
        if (16 * rb < nrows)

// solve:  F = (rhs (o a) + rho sum_k (Z_k - U_k)) L^-1   (decomposition.py:266-273 / 328-331)
template <int NBR, bool VEC>
__global__ __launch_bounds__(256) void k_rows_solve(ModeView mv, const float *__restrict__ rhs_src,
                                                    const float *__restrict__ Arows, const float *__restrict__ Linv,
                                                    RegSet regs, int r, double *__restrict__ change_part) {
    // change_part != nullptr: per tile ||f_new - f_old||^2 (the inner stopping test, decomposition.py:100-107)
    TILE_PROLOGUE();
    double chg = 0.0;
    const float rho = mv.rho[slab];
    RowMat<NBR> L, D;
    L.load(Linv + (long)slab * r * r, r, lane);
    int kpf2 = -1;
    for (int k = 0; k < regs.n; ++k)
        if (regs.kind[k] == MCL_PEN_PARAFAC2) kpf2 = k;
    if (kpf2 >= 0) D.load(regs.aux2[kpf2], r, lane);
    float av[NBR][4];
#pragma unroll
    for (int h = 0; h < NBR; ++h)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int col = 16 * h + 4 * g + v;
            av[h][v] = (Arows != nullptr && col < r) ? Arows[(long)slab * r + col] : 1.f;
        }
    FOR_ROW_BLOCKS() {
        const bool ok = 16 * rb + row16 < nrows;
        const long j = row0 + 16 * rb + (ok ? row16 : 0);
        f32x4 t[NBR], f[NBR];
#pragma unroll
        for (int h = 0; h < NBR; ++h) {
            t[h] = row_ld4<VEC>(rhs_src, j, 16 * h + 4 * g, ok, r);
#pragma unroll
            for (int v = 0; v < 4; ++v) t[h][v] *= av[h][v];
        }
        for (int k = 0; k < regs.n; ++k) {
            f32x4 z[NBR];
#pragma unroll
            for (int h = 0; h < NBR; ++h) z[h] = row_ld4<VEC>(regs.aux[k], j, 16 * h + 4 * g, ok, r);
            if (k == kpf2) {
                f32x4 pz[NBR];
                D.apply(z, pz);
#pragma unroll
                for (int h = 0; h < NBR; ++h) z[h] = pz[h];
            }
#pragma unroll
            for (int h = 0; h < NBR; ++h) {
                const f32x4 u = row_ld4<VEC>(regs.dual[k], j, 16 * h + 4 * g, ok, r);
#pragma unroll
                for (int v = 0; v < 4; ++v) t[h][v] = fmaf(rho, z[h][v] - u[v], t[h][v]);
            }
        }
        L.apply(t, f);
        if (change_part != nullptr) {
#pragma unroll
            for (int h = 0; h < NBR; ++h) {
                const f32x4 fo = row_ld4<VEC>((const float *)mv.F, j, 16 * h + 4 * g, ok, r);
#pragma unroll
                for (int v = 0; v < 4; ++v)
                    if (ok && 16 * h + 4 * g + v < r) {
                        const double d = (double)f[h][v] - (double)fo[v];
                        chg = fma(d, d, chg);
                    }
            }
        }
#pragma unroll
        for (int h = 0; h < NBR; ++h) row_st4<VEC>(mv.F, j, 16 * h + 4 * g, ok, r, f[h]);
    }
    if (change_part != nullptr) {
        chg = wave_sum_d(chg);
        if (lane == 0) change_part[tile] = chg;
    }
}

// The same solve with the per-slab statistics of the slab-wise penalties riding in its epilogue (mode 1, fused stacks):
//   PARAFAC2:  per-tile Gram Y^T Y, Y = F + U   - B_new rows go ROW -> COL layout through 4 selector MFMAs (exact), then
//              the fp64 MFMA accumulates exact fp32 x fp32 products exactly like k_pf2_gram
//   L2 ball :  per-tile column sums of squares of (F + U) (optionally clamped at 0), fp64
// k_stats_reduce sums the tiles of every slab in fixed order.  Saves the two extra reads of F and the duals that
// k_pf2_gram and k_slab_colsq need.
// R64 (rank <= 16 with a PARAFAC2 member, see mcl_rows64): the same pass with every product and sum in fp64 registers
// around the fp32 loads and stores - L^-1 from its fp64 copy, the statistics of Y = F + U from the exact fp64 sum of the
// two STORED (fp32) values.  tools/pf2_rounding_study.py: the fp32 accumulation chains of the r x r products, the fp32
// image of T_i and the rounded sum F + U carry two thirds of the B-phase error of BASELINE config 4 against the fp64
// reference (7e-7 per phase, amplified to 1e-5 in A by the penalty-free A / C systems of that configuration).
template <int NBR, bool VEC, bool R64 = false>
__global__ __launch_bounds__(256) void k_rows_solve_stats(ModeView mv, const float *__restrict__ rhs_src,
                                                          const float *__restrict__ Arows, const float *__restrict__ Linv,
                                                          RegSet regs, int r, double *__restrict__ stat_gram,
                                                          double *__restrict__ stat_colsq,
                                                          const double *__restrict__ Linv64 = nullptr) {
    typedef double f64x4s __attribute__((ext_vector_type(4)));
    typedef RowArith<R64> RA;
    __shared__ double ytile[R64 ? 4 * 16 * 17 : 1];  // R64: one padded 16 x 16 fp64 tile per wave (row -> column layout of Y)
    TILE_PROLOGUE();
    const float rho = mv.rho[slab];
    typename RA::template Mat<NBR> L, D;
    if constexpr (R64) L.load(Linv64 + (long)slab * r * r, r, lane);
    else L.load(Linv + (long)slab * r * r, r, lane);
    int kpf2 = -1;
    for (int k = 0; k < regs.n; ++k)
        if (regs.kind[k] == MCL_PEN_PARAFAC2) kpf2 = k;
    if (kpf2 >= 0) D.load(regs.aux2[kpf2], r, lane);
    float av[NBR][4];
#pragma unroll
    for (int h = 0; h < NBR; ++h)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int col = 16 * h + 4 * g + v;
            av[h][v] = (Arows != nullptr && col < r) ? Arows[(long)slab * r + col] : 1.f;
        }
    typename std::conditional<R64, double, float>::type bsel[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) bsel[v] = (row16 == 4 * g + v) ? 1.f : 0.f;
    f64x4s accS[NBR][NBR];
#pragma unroll
    for (int a = 0; a < NBR; ++a)
#pragma unroll
        for (int b = 0; b < NBR; ++b) accS[a][b] = f64x4s{0.0, 0.0, 0.0, 0.0};
    double csq[MCL_MAX_REGS][NBR][4];
#pragma unroll
    for (int k = 0; k < MCL_MAX_REGS; ++k)
#pragma unroll
        for (int h = 0; h < NBR; ++h)
#pragma unroll
            for (int v = 0; v < 4; ++v) csq[k][h][v] = 0.0;
    FOR_ROW_BLOCKS() {
        const bool ok = 16 * rb + row16 < nrows;
        const long j = row0 + 16 * rb + (ok ? row16 : 0);
        f32x4 t[NBR], f[NBR], ukeep[MCL_MAX_REGS][NBR];
#pragma unroll
        for (int h = 0; h < NBR; ++h) {
            t[h] = (row_ld4<VEC>(rhs_src, j, 16 * h + 4 * g, ok, r));
#pragma unroll
            for (int v = 0; v < 4; ++v) t[h][v] *= av[h][v];
        }
#pragma unroll
        for (int k = 0; k < MCL_MAX_REGS; ++k) {
            if (k < regs.n) {
                f32x4 z[NBR];
#pragma unroll
                for (int h = 0; h < NBR; ++h) z[h] = (row_ld4<VEC>(regs.aux[k], j, 16 * h + 4 * g, ok, r));
                if (k == kpf2) {
                    f32x4 pz[NBR];
                    D.apply(z, pz);
#pragma unroll
                    for (int h = 0; h < NBR; ++h) z[h] = pz[h];
                }
#pragma unroll
                for (int h = 0; h < NBR; ++h) {
                    ukeep[k][h] = (row_ld4<VEC>(regs.dual[k], j, 16 * h + 4 * g, ok, r));
#pragma unroll
                    for (int v = 0; v < 4; ++v) t[h][v] = fmaf(rho, z[h][v] - ukeep[k][h][v], t[h][v]);
                }
            }
        }
        L.apply(t, f);
#pragma unroll
        for (int h = 0; h < NBR; ++h) row_st4<VEC>(mv.F, j, 16 * h + 4 * g, ok, r, f[h]);
        // ---- statistics of the new rows (padding rows / columns contribute zeros: row_ld4 returned zeros for them and
        //      L.apply leaves padding columns at zero)
#pragma unroll
        for (int k = 0; k < MCL_MAX_REGS; ++k) {
            if (k < regs.n) {
                if (regs.kind[k] == MCL_PEN_L2BALL) {
#pragma unroll
                    for (int h = 0; h < NBR; ++h)
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            float y = f[h][v] + ukeep[k][h][v];
                            if (regs.nonneg[k]) y = fmaxf(y, 0.f);
                            if (ok) csq[k][h][v] += (double)y * (double)y;
                        }
                } else if (k == kpf2) {
                    double yt[NBR][4];
#pragma unroll
                    for (int nb = 0; nb < NBR; ++nb) {
                        if constexpr (R64) {  // exact fp64 sum, transposed through the wave's LDS tile (the fp64 matrix pipe
                            // is the scarce unit of these passes: 44 TFLOP/s at best, tools/mfma64_rate.hip):
                            // lane (q, i16) reg w = Y[q + 4w][16nb + i16]
                            double *yl = ytile + (threadIdx.x >> 6) * (16 * 17);
#pragma unroll
                            for (int v = 0; v < 4; ++v)
                                yl[row16 * 17 + 4 * g + v] = ok ? (double)f[nb][v] + (double)ukeep[k][nb][v] : 0.0;
#pragma unroll
                            for (int w = 0; w < 4; ++w) yt[nb][w] = yl[(g + 4 * w) * 17 + row16];
                        } else {
                            f32x4 tr = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                            for (int v = 0; v < 4; ++v) {
                                const float y = ok ? f[nb][v] + ukeep[k][nb][v] : 0.f;
                                tr = MFMA16(y, bsel[v], tr);  // COL layout: lane (q, i16) reg w = Y[4q + w][16nb + i16]
                            }
#pragma unroll
                            for (int w = 0; w < 4; ++w) yt[nb][w] = (double)tr[w];
                        }
                    }
#pragma unroll
                    for (int w = 0; w < 4; ++w)
#pragma unroll
                        for (int a = 0; a < NBR; ++a)
#pragma unroll
                            for (int b = 0; b < NBR; ++b)
                                accS[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(yt[a][w], yt[b][w], accS[a][b], 0, 0, 0);
                }
            }
        }
    }
    constexpr int W = 16 * NBR;
    if (kpf2 >= 0) {  // D layout of the f64 MFMA: col = l & 15, row = (l >> 4) + 4 reg
        double *out = stat_gram + (long)tile * W * W;
#pragma unroll
        for (int a = 0; a < NBR; ++a)
#pragma unroll
            for (int b = 0; b < NBR; ++b)
#pragma unroll
                for (int v = 0; v < 4; ++v) out[(16 * a + g + 4 * v) * W + 16 * b + row16] = accS[a][b][v];
    }
#pragma unroll
    for (int k = 0; k < MCL_MAX_REGS; ++k) {
        if (k < regs.n && regs.kind[k] == MCL_PEN_L2BALL) {
#pragma unroll
            for (int h = 0; h < NBR; ++h)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    double sq = csq[k][h][v];
                    sq += __shfl_xor(sq, 1);
                    sq += __shfl_xor(sq, 2);
                    sq += __shfl_xor(sq, 4);
                    sq += __shfl_xor(sq, 8);
                    const int col = 16 * h + 4 * g + v;
                    if (row16 == 0 && col < r) stat_colsq[((long)tile * MCL_MAX_REGS + k) * r + col] = sq;
                }
        }
    }
}

// per-slab sums of the per-tile statistics, fixed order: S[slab] (natural r x r layout) and colsq[k][slab]
__global__ __launch_bounds__(256) void k_stats_reduce(const int *__restrict__ slab_tile_ptr, const double *__restrict__ stat_gram,
                                                      const double *__restrict__ stat_colsq, RegSet regs, int r, int W,
                                                      int n_slabs, double *__restrict__ S, double *__restrict__ colsq) {
    MCL_GATE(regs.gate);
    const int slab = blockIdx.x;
    const int t0 = slab_tile_ptr[slab], t1 = slab_tile_ptr[slab + 1];
    int kpf2 = -1;
    for (int k = 0; k < regs.n; ++k)
        if (regs.kind[k] == MCL_PEN_PARAFAC2) kpf2 = k;
    if (kpf2 >= 0) {
        for (int e = threadIdx.x; e < r * r; e += 256) {
            const int a = e / r, b = e - a * r;
            double s0 = 0.0, s1 = 0.0;
            int t = t0;
            for (; t + 1 < t1; t += 2) {
                s0 += stat_gram[(long)t * W * W + a * W + b];
                s1 += stat_gram[(long)(t + 1) * W * W + a * W + b];
            }
            if (t < t1) s0 += stat_gram[(long)t * W * W + a * W + b];
            S[((long)slab * r + a) * r + b] = s0 + s1;
        }
    }
    for (int k = 0; k < regs.n; ++k) {
        if (regs.kind[k] != MCL_PEN_L2BALL) continue;
        for (int col = threadIdx.x; col < r; col += 256) {
            double s = 0.0;
            for (int t = t0; t < t1; ++t) s += stat_colsq[((long)t * MCL_MAX_REGS + k) * r + col];
            colsq[((long)k * n_slabs + slab) * r + col] = s;
        }
    }
}

// mode 0: every row has its own system (decomposition.py:184-195); one wave per row
__global__ __launch_bounds__(64) void k_A_rows_solve(const float *__restrict__ rhsA, const float *__restrict__ rhoA,
                                                     const float *__restrict__ LinvA, float *__restrict__ A,
                                                     RegSet regs, int r, double *__restrict__ change_part) {
    MCL_GATE(regs.gate);
    __shared__ float tS[MCL_MAX_RANK];
    const int i = blockIdx.x, lane = threadIdx.x;
    const bool act = lane < r;
    const int c = act ? lane : 0;
    const float rho = rhoA[i];
    float t = rhsA[(long)i * r + c];
    for (int k = 0; k < regs.n; ++k) t = fmaf(rho, regs.aux[k][(long)i * r + c] - regs.dual[k][(long)i * r + c], t);
    if (act) tS[c] = t;
    __syncthreads();
    float a = 0.f;
    for (int d = 0; d < r; ++d) a = fmaf(tS[d], LinvA[((long)i * r + d) * r + c], a);
    if (change_part != nullptr) {  // ||a_new - a_old||^2 of the row (the inner stopping test)
        const double d = act ? (double)a - (double)A[(long)i * r + c] : 0.0;
        const double s = wave_sum_d(d * d);
        if (lane == 0) change_part[i] = s;
    }
    if (act) A[(long)i * r + c] = a;
}

// row-separable prox + dual step of penalty k on the rows of A when every row has ITS OWN feasibility penalty rho_i
// (decomposition.py:197-213 with factor_matrix_row_update): the threshold of an L1 penalty is reg_strength / rho_i
__global__ __launch_bounds__(256) void k_A_rows_prox(const float *__restrict__ rhoA, float *__restrict__ A, RegSet regs, int k, int I,
                                                     int r) {
    MCL_GATE(regs.gate);
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)I * r) return;
    const float rho = rhoA[e / r];
    const float f = A[e], u = regs.dual[k][e];
    const float z = prox_elem_g(regs.kind[k], regs.nonneg[k], regs.p0[k], regs.p1[k], regs.p0[k] / rho, f + u);
    regs.aux[k][e] = z;
    regs.dual[k][e] = f - (z - u);
}

// The inner stopping test of a phase (decomposition.py:90-117) on the device.  begin: flag <- the run's stop flag (a gated run
// that has stopped stays stopped).  Otherwise: change = sum of the solve pass's per-tile ||x - x_old||^2, norm and gaps from the
// per-tile diagnostics table (column 0: ||x||^2, columns 2 + k: ||aux_k - x||^2); converged = change <= tol * norm and
// every gap < tol (all on square roots, as the reference compares norms) -> flag = 1.  One workgroup, fixed summation order.
__global__ __launch_bounds__(256) void k_inner_check(int begin, const int *__restrict__ run_gate, int *__restrict__ flag,
                                                     const double *__restrict__ change_part, int n_change,
                                                     const double *__restrict__ diag_tab, int n_rows, int n_regs, double tol) {
    __shared__ double sm[256];
    if (begin) {
        if (threadIdx.x == 0) flag[0] = (run_gate != nullptr) ? run_gate[0] : 0;
        return;
    }
    if (flag[0] != 0) return;
    auto block_sum = [&](double v) {
        sm[threadIdx.x] = v;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
            __syncthreads();
        }
        const double t = sm[0];
        __syncthreads();
        return t;
    };
    double v = 0.0;
    for (int e = threadIdx.x; e < n_change; e += 256) v += change_part[e];
    const double change = block_sum(v);
    v = 0.0;
    for (int e = threadIdx.x; e < n_rows; e += 256) v += diag_tab[(long)e * DIAG_COLS];
    const double norm = block_sum(v);
    bool ok = !(sqrt(change) > tol * sqrt(norm));
    for (int k = 0; k < n_regs; ++k) {
        v = 0.0;
        for (int e = threadIdx.x; e < n_rows; e += 256) v += diag_tab[(long)e * DIAG_COLS + 2 + k];
        const double gap = sqrt(block_sum(v)) / sqrt(norm);
        ok = ok && (gap < tol);
    }
    if (threadIdx.x == 0 && ok) flag[0] = 1;
}

// row-separable prox + dual update of penalty k (generic path)
template <int NBR, bool VEC>
__global__ __launch_bounds__(256) void k_rows_prox_rowsep(ModeView mv, RegSet regs, int k, int r) {
    TILE_PROLOGUE();
    const float rho = mv.rho[slab];
    const float thr = regs.p0[k] / rho;
    FOR_ROW_BLOCKS() {
        const bool ok = 16 * rb + row16 < nrows;
        const long j = row0 + 16 * rb + (ok ? row16 : 0);
#pragma unroll
        for (int h = 0; h < NBR; ++h) {
            const int col = 16 * h + 4 * g;
            const f32x4 f = row_ld4<VEC>(mv.F, j, col, ok, r);
            f32x4 u = row_ld4<VEC>(regs.dual[k], j, col, ok, r), z;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                z[v] = prox_elem_g(regs.kind[k], regs.nonneg[k], regs.p0[k], regs.p1[k], thr, f[v] + u[v]);
                u[v] = f[v] - (z[v] - u[v]);
            }
            row_st4<VEC>(regs.aux[k], j, col, ok, r, z);
            row_st4<VEC>(regs.dual[k], j, col, ok, r, u);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// L2 ball: column sums of squares per slab (fp64), then scale + dual
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_slab_colsq(const int *__restrict__ ext, const float *__restrict__ F,
                                                    const float *__restrict__ U, int nonneg, int r, int RP,
                                                    double *__restrict__ colsq) {
    __shared__ double sm[256];
    const int slab = blockIdx.x;
    const int s = ext[slab], e = ext[slab + 1];
    const int col = threadIdx.x % RP, rl = threadIdx.x / RP, nrl = 256 / RP;
    double acc = 0.0;
    if (col < r) {
        for (long j = (long)s + rl; j < e; j += nrl) {
            float y = F[j * r + col] + U[j * r + col];
            if (nonneg) y = fmaxf(y, 0.f);
            acc += (double)y * (double)y;
        }
    }
    sm[threadIdx.x] = acc;
    __syncthreads();
    if ((int)threadIdx.x < RP && col < r) {
        double t = 0.0;
        for (int q = 0; q < nrl; ++q) t += sm[q * RP + col];
        colsq[(long)slab * r + col] = t;
    }
}

template <int NBR, bool VEC>
__global__ __launch_bounds__(256) void k_rows_l2ball(ModeView mv, RegSet regs, int k, int r,
                                                     const double *__restrict__ colsq) {
    TILE_PROLOGUE();
    const float bound = regs.p0[k];
    float scale[NBR][4];
#pragma unroll
    for (int h = 0; h < NBR; ++h)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int col = 16 * h + 4 * g + v;
            const float nrm = (col < r) ? (float)sqrt(colsq[(long)slab * r + col]) : 1.f;
            scale[h][v] = bound / fmaxf(nrm, bound);
        }
    FOR_ROW_BLOCKS() {
        const bool ok = 16 * rb + row16 < nrows;
        const long j = row0 + 16 * rb + (ok ? row16 : 0);
#pragma unroll
        for (int h = 0; h < NBR; ++h) {
            const int col = 16 * h + 4 * g;
            const f32x4 f = row_ld4<VEC>(mv.F, j, col, ok, r);
            f32x4 u = row_ld4<VEC>(regs.dual[k], j, col, ok, r), z;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                float y = f[v] + u[v];
                if (regs.nonneg[k]) y = fmaxf(y, 0.f);
                z[v] = y * scale[h][v];
                u[v] = f[v] - (z[v] - u[v]);
            }
            row_st4<VEC>(regs.aux[k], j, col, ok, r, z);
            row_st4<VEC>(regs.dual[k], j, col, ok, r, u);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Total variation (penalties.py:750-841): per column of a slab, aux = argmin 1/2 ||y - (F + U)||^2 + lam sum |y_n - y_{n-1}|
// with lam = 2 alpha / rho, then the L1 soft threshold with l1 / rho.  The reference calls condat_tv (GPL); this is
// L. Condat's direct algorithm (IEEE SPL 20(11), 2013) restated from the paper: one pass with occasional restarts at
// the last position where a bound was active; fp64 like the unimodal regression (its decisions are discontinuous).
// One lane per (slab, column): the r lanes of a slab read / write consecutive addresses.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_slab_tv(const int *__restrict__ ext, int n_slabs, const float *__restrict__ F,
                                                const float *__restrict__ rho_arr, RegSet regs, int kreg, int r) {
    MCL_GATE(regs.gate);
    const long t = (long)blockIdx.x * 64 + threadIdx.x;
    if (t >= (long)n_slabs * r) return;
    const int slab = (int)(t / r), col = (int)(t - (long)slab * r);
    const long s = ext[slab];
    const int n = ext[slab + 1] - ext[slab];
    if (n <= 0) return;
    const float *U = regs.dual[kreg];
    float *Z = regs.aux[kreg];
    const double rho = (double)rho_arr[slab];
    const double lam = 2.0 * (double)regs.p0[kreg] / rho;
    const float l1 = (float)((double)regs.p1[kreg] / rho);
    auto in = [&](int k) -> double { return (double)(F[(s + k) * r + col] + U[(s + k) * r + col]); };
    auto emit = [&](int a, int b, double v) {  // y[a..b] = v, then soft threshold
        float z = (float)v;
        if (l1 > 0.f) z = copysignf(fmaxf(fabsf(z) - l1, 0.f), z);
        for (int k = a; k <= b; ++k) Z[(s + k) * r + col] = z;
    };
    int k = 0, k0 = 0, km = 0, kp = 0;
    double x0 = in(0);
    double vmin = x0 - lam, vmax = x0 + lam, umin = lam, umax = -lam;
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
// GeneralizedL2Penalty (penalties.py:595-747): prox(Y) = U (S + rho/2 I)^-1 U^T (rho/2) Y with M = U S U^T, per slab.
// Two passes of ONE routine  out[a][c] = sum_b Mat[b][a] In[b][c]  (Mat = U for the projection onto the eigenvectors, Mat = U^T
// for the way back: both walk Mat along its rows, coalesced), fp64 accumulation in the order of b.  Workgroup = 64 values of
// `a` x 4 column phases; the rows of In are staged through LDS 16 at a time.  T: fp64 scratch [rows, r].
// TIN = float (the caller's buffers) or double (the fp64 state of wide.hip).
// ---------------------------------------------------------------------------------------------------------
template <typename TIN, bool BACK>
__global__ __launch_bounds__(256) void k_gl2_pass(const int *__restrict__ ext, const double *__restrict__ Mat,
                                                  const double *__restrict__ eig, int n, int r, const float *__restrict__ rho_arr,
                                                  const TIN *__restrict__ F, const TIN *__restrict__ D, double *__restrict__ T,
                                                  float *__restrict__ Z32, float *__restrict__ D32, double *__restrict__ Z64,
                                                  double *__restrict__ D64, const int *__restrict__ gate) {
    // !BACK: T[e][c] = (rho / 2) sum_j U[j][e] (F + D)[j][c]        (D = nullptr: F alone, unscaled - the penalty value)
    //  BACK: Z[j][c] = sum_e U^T[e][j] T[e][c] / (s_e + rho / 2), then the dual step D = F - (Z - D)
    MCL_GATE(gate);
    __shared__ double ins[16][64];
    // (the slab index is folded into blockIdx.x: gridDim.y stops at 65535, a GeneralizedL2 penalty on the B_i may see more matrices)
    const int tps = (n + 63) >> 6;
    const int slab = (int)(blockIdx.x / (unsigned)tps);
    const long s0 = ext[slab];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int a = (int)(blockIdx.x - (unsigned)slab * (unsigned)tps) * 64 + tx;
    const double half_rho = 0.5 * (double)rho_arr[slab];
    double acc[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.0;
    for (int b0 = 0; b0 < n; b0 += 16) {
        const int nb = min(16, n - b0);
        __syncthreads();
        for (int e = threadIdx.x; e < nb * r; e += 256) {
            const int bb = e / r, c = e - bb * r;
            const long idx = (s0 + b0 + bb) * r + c;
            double v;
            if (BACK) v = T[idx] / (eig[b0 + bb] + half_rho);
            else v = (double)F[idx] + (D != nullptr ? (double)D[idx] : 0.0);
            ins[bb][c] = v;
        }
        __syncthreads();
        if (a < n)
            for (int bb = 0; bb < nb; ++bb) {
                const double m = Mat[(long)(b0 + bb) * n + a];
#pragma unroll
                for (int q = 0; q < 16; ++q)
                    if (ty + 4 * q < r) acc[q] = fma(m, ins[bb][ty + 4 * q], acc[q]);
            }
    }
    if (a >= n) return;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int c = ty + 4 * q;
        if (c >= r) continue;
        const long idx = (s0 + a) * r + c;
        if (!BACK) {
            T[idx] = (D != nullptr ? half_rho : 1.0) * acc[q];
        } else {
            const double z = acc[q], f = (double)F[idx], u = (double)D[idx];
            const double un = f - (z - u);
            Z32[idx] = (float)z, D32[idx] = (float)un;
            if (Z64 != nullptr) Z64[idx] = z, D64[idx] = un;
        }
    }
}

// sum over all rows e of every slab of s_e sum_c T[e][c]^2 (T = U^T F): trace(F^T M F), one workgroup, fixed order
__global__ __launch_bounds__(256) void k_gl2_value(const double *__restrict__ T, const double *__restrict__ eig, long rows, int n,
                                                   int r, double *__restrict__ out) {
    __shared__ double sm[256];
    double acc = 0.0;
    for (long e = threadIdx.x; e < rows; e += 256) {
        double s = 0.0;
        for (int c = 0; c < r; ++c) s = fma(T[e * r + c], T[e * r + c], s);
        acc = fma(eig[e % n], s, acc);
    }
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sm[0];
}

// ---------------------------------------------------------------------------------------------------------
// UnitSimplex (penalties.py:928-980): per column of a slab the multiplier mu of sum_j max(y_j - mu, 0) = 1 (the reference:
// scipy's bisection on the same bracket), aux = max(y - mu, 0).  One wave per (slab, column): 60 bisection steps narrow the
// bracket to the linear piece of the root, where mu = (sum of the active entries - 1) / their number exactly.
// ---------------------------------------------------------------------------------------------------------
template <typename TIN>
__global__ __launch_bounds__(256) void k_slab_simplex(const int *__restrict__ ext, int n_slabs, int r, const TIN *__restrict__ F,
                                                      const TIN *__restrict__ D, float *__restrict__ Z32, double *__restrict__ Z64,
                                                      const int *__restrict__ gate) {
    MCL_GATE(gate);
    const int lane = threadIdx.x & 63;
    const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= (long)n_slabs * r) return;
    const int slab = (int)(w / r), col = (int)(w - (long)slab * r);
    const long s0 = ext[slab];
    const int n = ext[slab + 1] - ext[slab];
    if (n <= 0) return;
    auto y_at = [&](int j) { return (double)F[(s0 + j) * r + col] + (double)D[(s0 + j) * r + col]; };
    double mn = 1e300, mx = -1e300;
    for (int j = lane; j < n; j += 64) {
        const double y = y_at(j);
        mn = fmin(mn, y), mx = fmax(mx, y);
    }
    for (int o = 32; o > 0; o >>= 1) mn = fmin(mn, __shfl_xor(mn, o)), mx = fmax(mx, __shfl_xor(mx, o));
    double lo = mn - 1.0 - 1e-5, hi = mx + 1e-5;  // f(lo) > 0 > f(hi) = -1 (the reference's bracket, penalties.py:950-960)
    lo = fmin(0.9 * lo, 1.1 * lo), hi = fmax(0.9 * hi, 1.1 * hi);
    for (int it = 0; it < 60; ++it) {
        const double mid = 0.5 * (lo + hi);
        double s = 0.0;
        for (int j = lane; j < n; j += 64) s += fmax(y_at(j) - mid, 0.0);
        s = wave_sum_d(s) - 1.0;
        if (s > 0.0) lo = mid;
        else hi = mid;
    }
    // on [lo, hi] the active set is constant (unless a breakpoint sits within 2^-60 of the root): the root of the linear piece
    double sa = 0.0, cnt = 0.0;
    for (int j = lane; j < n; j += 64) {
        const double y = y_at(j);
        if (y > hi) sa += y, cnt += 1.0;
    }
    sa = wave_sum_d(sa), cnt = wave_sum_d(cnt);
    double mu = 0.5 * (lo + hi);
    if (cnt > 0.0) {
        const double m2 = (sa - 1.0) / cnt;
        if (m2 >= lo && m2 <= hi) mu = m2;
    }
    for (int j = lane; j < n; j += 64) {
        const double z = fmax(y_at(j) - mu, 0.0);
        Z32[(s0 + j) * r + col] = (float)z;
        if (Z64 != nullptr) Z64[(s0 + j) * r + col] = z;
    }
}

// ---------------------------------------------------------------------------------------------------------
// PARAFAC2 prox (mode 1):  Y_i = B_i + U_i,  P_i = polar(Y_i Delta^T),  Delta <- sum rho_i P_i^T Y_i / sum rho_i
// Gram route in fp64:  S_i = Y_i^T Y_i,  G_i = Delta S_i Delta^T = V L V^T,  W_i = V L^-1/2 V^T,
//                      T_i = Delta^T W_i,  P_i = Y_i T_i,  P_i^T Y_i = T_i^T S_i.
// ---------------------------------------------------------------------------------------------------------
typedef double f64x4 __attribute__((ext_vector_type(4)));

// S_i = Y_i^T Y_i on the fp64 MFMA (v_mfma_f64_16x16x4_f64): one workgroup per slab, lane (rsub = l>>4, c16 = l&15)
// reads Y[4g + rsub][16nb + c16] (256 contiguous bytes per wave-load for r = 16); fp32 x fp32 products are exact in
// fp64, so only the final rounding of the sums remains.  D layout of the f64 form: col = l&15, row = (l>>4) + 4 reg.
template <int NB>
__global__ __launch_bounds__(256) void k_pf2_gram(const int *__restrict__ ext, const float *__restrict__ F,
                                                  const float *__restrict__ U, int r, double *__restrict__ S) {
    constexpr int W = 16 * NB;
    __shared__ double sm[W * W];
    const int slab = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rsub = lane >> 4, c16 = lane & 15;
    const int s = ext[slab], e = ext[slab + 1];
    const int n_groups = (e - s + 3) >> 2;
    f64x4 acc[NB][NB];
#pragma unroll
    for (int a = 0; a < NB; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[a][b] = f64x4{0.0, 0.0, 0.0, 0.0};
    for (int gq = wave; gq < n_groups; gq += 4) {
        const long j = (long)s + 4 * gq + rsub;
        double y[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int col = 16 * nb + c16;
            // the EXACT sum of the two stored values (an fp32 sum would round Y by 6e-8: times cond(Y Delta^T) in the polar factor)
            y[nb] = (j < e && col < r) ? (double)F[j * r + col] + (double)U[j * r + col] : 0.0;
        }
#pragma unroll
        for (int a = 0; a < NB; ++a)
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(y[a], y[b], acc[a][b], 0, 0, 0);
    }
    for (int wv = 0; wv < 4; ++wv) {
        if (wave == wv) {
#pragma unroll
            for (int a = 0; a < NB; ++a)
#pragma unroll
                for (int b = 0; b < NB; ++b)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int idx = (16 * a + rsub + 4 * v) * W + 16 * b + c16;
                        sm[idx] = (wv == 0) ? acc[a][b][v] : sm[idx] + acc[a][b][v];
                    }
        }
        __syncthreads();
    }
    for (int t = threadIdx.x; t < W * W; t += 256) {
        const int a = t / W, b = t - a * W;
        if (a < r && b < r) S[((long)slab * r + a) * r + b] = sm[t];
    }
}

// Synchronisation of ONE wave with itself around its LDS traffic.  The LDS operations of a wave are executed in the order
// they were issued, so a value written by one lane is there for the next read of another lane of the same wave: what is needed is
// that the compiler keeps that order.  (A workgroup barrier would also do in a one-wave workgroup - where the compiler drops it -
// but the Newton-Schulz kernel runs FOUR independent slabs per workgroup at rank <= 16, each wave on its own way through the
// fallbacks: a real barrier there would wait for waves that never come.)
static __device__ __forceinline__ void wave_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One slab through the Jacobi route (one wave; `smd` = 4 r^2 + r doubles of LDS, `Ssrc` = the slab's S in any address
// space).  Called by the stand-alone kernel below and by k_pf2_algebra_ns for the slabs it cannot handle.
static __device__ void pf2_jacobi_slab(double *smd, const double *Ssrc, const float *__restrict__ Delta, double rh, int r,
                                       int slab, int lane, float *__restrict__ T, double *__restrict__ acc_out,
                                       double *__restrict__ T64 = nullptr, int *__restrict__ qr_flag = nullptr,
                                       bool full_rank_expected = false, const double *__restrict__ Delta64 = nullptr) {
    double *Sm = smd, *G = smd + r * r, *V = G + r * r, *lam = V + r * r, *D = lam + r;  // D: Delta in fp64 [r*r]
    const int n2 = r * r;
    for (int e = lane; e < n2; e += 64) {
        Sm[e] = Ssrc[e];
        D[e] = Delta64 != nullptr ? Delta64[e] : (double)Delta[e];  // (the fp64 state of wide.hip, or the caller's fp32 matrix)
        V[e] = ((e / r) == (e % r)) ? 1.0 : 0.0;
    }
    wave_sync_lds();
    // G = D S D^T  (via tmp = S D^T stored in V's place is not possible: V must stay I) -> two-step through lam-free loop
    for (int e = lane; e < n2; e += 64) {
        const int a = e / r, b = e - a * r;
        double sum = 0.0;
        for (int k = 0; k < r; ++k) {
            double inner = 0.0;
            for (int l = 0; l < r; ++l) inner += Sm[k * r + l] * D[b * r + l];
            sum += D[a * r + k] * inner;
        }
        G[e] = sum;
    }
    wave_sync_lds();
    // Jacobi eigen-decomposition of the symmetric G (fp64); V accumulates the eigenvectors.  Round 6: PARALLEL ordering - in
    // each of the n - 1 steps of a sweep the n / 2 disjoint index pairs of a round-robin tournament are rotated at once
    // (Brent-Luk), where rounds 2-5 walked the r (r - 1) / 2 pairs of a cyclic sweep one after the other: every rotation is two
    // barriers and a division / square-root chain (~1000 cycles), so a rank-16 matrix took ~1 M cycles - 436 us per call on the
    // 48-matrix problems of the exact arithmetic, 75 % of their iteration (rocprof, gpurun_out/r6).  A lane keeps ONE pair for the
    // whole step (its rotation computed redundantly by the lanes that share it, from the same LDS values: identical bits), applies
    // it to its rows of G J and V J, then to its columns of J^T (G J).
    {
        const int n = r + (r & 1);      // an odd rank plays with a bye (index r)
        const int npairs = n >> 1;
        int P = 1;
        while (P < npairs) P <<= 1;     // lanes per row group: pair i = lane % P (P <= 32 for rank <= 64)
        const int pi = lane % P, kstep = 64 / P, k0 = lane / P;
        for (int sweep = 0; sweep < 30; ++sweep) {
            double off = 0.0, dg = 0.0;
            for (int e = lane; e < n2; e += 64) {
                const int a = e / r, b = e - a * r;
                if (a == b) dg += G[e] * G[e];
                else off += G[e] * G[e];
            }
            off = wave_sum_d(off);
            dg = wave_sum_d(dg);
            if (off <= 1e-30 * dg) break;
            for (int st = 0; st < n - 1; ++st) {
                // the pair of this lane in step st: (n - 1, st) for i = 0, ((st + i) mod (n - 1), (st - i) mod (n - 1)) otherwise
                int p = -1, q = -1;
                if (pi < npairs) {
                    int a = pi == 0 ? n - 1 : (st + pi) % (n - 1), b = pi == 0 ? st : (st - pi + (n - 1)) % (n - 1);
                    p = min(a, b), q = max(a, b);
                    if (q >= r) p = -1;  // the bye
                }
                wave_sync_lds();
                double cs = 1.0, sn = 0.0, dpp = 0.0, dqq = 0.0;
                bool rot = false;
                if (p >= 0) {
                    const double apq = G[p * r + q], app = G[p * r + p], aqq = G[q * r + q];
                    if (fabs(apq) > 1e-300) {
                        const double tau = (aqq - app) / (2.0 * apq);
                        const double tt = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                        cs = 1.0 / sqrt(1.0 + tt * tt), sn = tt * cs;
                        dpp = app - tt * apq, dqq = aqq + tt * apq;
                        rot = true;
                    }
                }
                wave_sync_lds();
                if (rot)  // columns p, q of G and V:  X <- X J
                    for (int k = k0; k < r; k += kstep) {
                        const double gkp = G[k * r + p], gkq = G[k * r + q], vkp = V[k * r + p], vkq = V[k * r + q];
                        G[k * r + p] = cs * gkp - sn * gkq, G[k * r + q] = sn * gkp + cs * gkq;
                        V[k * r + p] = cs * vkp - sn * vkq, V[k * r + q] = sn * vkp + cs * vkq;
                    }
                wave_sync_lds();
                if (rot)  // rows p, q of G:  G <- J^T G
                    for (int k = k0; k < r; k += kstep) {
                        const double gpk = G[p * r + k], gqk = G[q * r + k];
                        G[p * r + k] = cs * gpk - sn * gqk, G[q * r + k] = sn * gpk + cs * gqk;
                    }
                wave_sync_lds();
                if (rot && k0 == 0) {  // the rotated 2 x 2 block in closed form (exact zero off the diagonal)
                    G[p * r + p] = dpp, G[q * r + q] = dqq;
                    G[p * r + q] = 0.0, G[q * r + p] = 0.0;
                }
            }
            wave_sync_lds();
        }
    }
    wave_sync_lds();
    double lmax = 0.0, lmin = 1e300;
    for (int k = 0; k < r; ++k) lmax = fmax(lmax, G[k * r + k]), lmin = fmin(lmin, G[k * r + k]);
    // G = (Y Delta^T)^T (Y Delta^T) squares the condition number: beyond cond(Y Delta^T) ~ 1e5 its small eigenvalues are
    // rounding (at 1e7 the pseudo-inverse threshold below removes them and the factor comes out rank-deficient: errors of
    // 0.2 in P, tools/parity_probe.py fuzz:133).  A slab with at least r rows is flagged for k_pf2_polar_qr, which works on
    // Y Delta^T itself; what is written below is then overwritten.
    if (qr_flag != nullptr && lane == 0) qr_flag[slab] = (full_rank_expected && !(lmin > 1e-10 * lmax)) ? 2 : 0;
    if (lane < r) {
        const double l = G[lane * r + lane];
        lam[lane] = (l > 1e-14 * lmax && l > 0.0) ? 1.0 / sqrt(l) : 0.0;  // pseudo-inverse square root
    }
    wave_sync_lds();
    // W = V diag(lam) V^T -> G
    for (int e = lane; e < n2; e += 64) {
        const int a = e / r, b = e - a * r;
        double sum = 0.0;
        for (int k = 0; k < r; ++k) sum += V[a * r + k] * lam[k] * V[b * r + k];
        G[e] = sum;
    }
    wave_sync_lds();
    // T = D^T W -> V
    for (int e = lane; e < n2; e += 64) {
        const int a = e / r, b = e - a * r;
        double sum = 0.0;
        for (int k = 0; k < r; ++k) sum += D[k * r + a] * G[k * r + b];
        V[e] = sum;
    }
    wave_sync_lds();
    for (int e = lane; e < n2; e += 64) {
        const int a = e / r, b = e - a * r;
        double sum = 0.0;
        for (int k = 0; k < r; ++k) sum += V[k * r + a] * Sm[k * r + b];
        acc_out[(long)slab * (n2 + 1) + e] = rh * sum;
        T[(long)slab * n2 + e] = (float)V[e];
        if (T64 != nullptr) T64[(long)slab * n2 + e] = V[e];
    }
    if (lane == 0) acc_out[(long)slab * (n2 + 1) + n2] = rh;
}

__global__ __launch_bounds__(64) void k_pf2_algebra(const double *__restrict__ S, const float *__restrict__ Delta,
                                                    const float *__restrict__ rho, int r, float *__restrict__ T,
                                                    double *__restrict__ acc_out, int *__restrict__ status, int use_status,
                                                    double *__restrict__ T64, const int *__restrict__ ext,
                                                    const double *__restrict__ Delta64 = nullptr) {
    extern __shared__ double smd[];
    // use_status: entries <= 0 were done by the Newton-Schulz kernel (-iterations); this kernel leaves 2 (k_pf2_polar_qr
    // takes the slab) or 0 in the entry of every slab it handles
    if (use_status && status[blockIdx.x] <= 0) return;
    const int slab = blockIdx.x;
    pf2_jacobi_slab(smd, S + (long)slab * r * r, Delta, (double)rho[slab], r, slab, threadIdx.x, T, acc_out, T64, status,
                    ext[slab + 1] - ext[slab] >= r, Delta64);
}

// ---------------------------------------------------------------------------------------------------------
// Polar factor of an ill-conditioned slab without squaring its condition number (the slabs k_pf2_algebra flags with 2):
//   A = Y_i Delta^T (J_i x r, fp64, Y_i = F + U exactly)  ->  Householder QR in place  ->  R (r x r)
//   ->  one-sided Jacobi on the columns of R (Hestenes): R V = U Sigma  ->  (A^T A)^-1/2 = V Sigma^-1 V^T = W
//   ->  T_i = Delta^T W,  acc_i = rho_i T_i^T S   (what pf2_jacobi_slab writes, decomposition of penalties.py:1224-1250's SVD)
// R inherits the singular values of A to eps * cond relative accuracy (the Gram route: eps * cond^2), the one-sided Jacobi
// keeps it.  One workgroup per flagged slab; rare by construction (cond >= 1e5: typically the first inner iteration from a
// random dual), so simplicity beats speed: the reflections are applied column-parallel, the rotations by one wave.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pf2_polar_qr(const int *__restrict__ ext, const float *__restrict__ F,
                                                      const float *__restrict__ U, const float *__restrict__ Delta,
                                                      const float *__restrict__ rho, int r, const double *__restrict__ S,
                                                      const int *__restrict__ status, double *__restrict__ Aws,
                                                      float *__restrict__ T, double *__restrict__ acc_out,
                                                      double *__restrict__ T64, const int *__restrict__ gate,
                                                      const double *__restrict__ F64 = nullptr, const double *__restrict__ U64 = nullptr,
                                                      const double *__restrict__ Delta64 = nullptr) {
    MCL_GATE(gate);
    const int slab = blockIdx.x;
    if (status[slab] != 2) return;
    const int s = ext[slab], n = ext[slab + 1] - s;
    if (n < r) return;  // (never flagged)
    extern __shared__ double sq[];
    const int n2 = r * r;
    double *D = sq, *R = D + n2, *V = R + n2, *red = V + n2, *lam = red + 4 * 64;  // red: 4 row chunks x 64 columns
    __shared__ double sh_alpha, sh_beta, sh_vkk;
    const int tid = threadIdx.x, lane = tid & 63, chunk = tid >> 6;
    for (int e = tid; e < n2; e += 256) D[e] = Delta64 != nullptr ? Delta64[e] : (double)Delta[e];
    __syncthreads();
    double *A = Aws + (long)s * r;
    for (long idx = tid; idx < (long)n * r; idx += 256) {
        const long j = idx / r;
        const int c = (int)(idx - j * r);
        double acc = 0.0;
        if (F64 != nullptr) {  // the fp64 state of wide.hip
            const double *f = F64 + ((long)s + j) * r, *u = U64 + ((long)s + j) * r;
            for (int k = 0; k < r; ++k) acc = fma(f[k] + u[k], D[c * r + k], acc);
        } else {
            const float *f = F + ((long)s + j) * r, *u = U + ((long)s + j) * r;
            for (int k = 0; k < r; ++k) acc = fma((double)f[k] + (double)u[k], D[c * r + k], acc);
        }
        A[idx] = acc;
    }
    __syncthreads();
    for (int k = 0; k < r; ++k) {
        // ||A[k:n, k]||^2
        double p = 0.0;
        for (int j = k + tid; j < n; j += 256) p = fma(A[(long)j * r + k], A[(long)j * r + k], p);
        p = wave_sum_d(p);
        if (lane == 0) red[chunk] = p;
        __syncthreads();
        if (tid == 0) {
            const double nrm2 = (red[0] + red[1]) + (red[2] + red[3]);
            const double xk = A[(long)k * r + k];
            const double alpha = -copysign(sqrt(nrm2), xk);
            const double vkk = xk - alpha;
            const double vtv = (nrm2 - xk * xk) + vkk * vkk;
            sh_alpha = alpha, sh_vkk = vkk, sh_beta = vtv > 0.0 ? 2.0 / vtv : 0.0;
        }
        __syncthreads();
        const double vkk = sh_vkk, beta = sh_beta;
        // w_c = beta v^T A[k:n, c] for the columns c > k: thread (chunk, lane) sums rows k + chunk, k + chunk + 4, ... of column k + 1 + lane
        for (int c0 = k + 1; c0 < r; c0 += 64) {
            const int c = c0 + lane;
            double dot = 0.0;
            if (c < r)
                for (int j = k + chunk; j < n; j += 4) {
                    const double vj = (j == k) ? vkk : A[(long)j * r + k];
                    dot = fma(vj, A[(long)j * r + c], dot);
                }
            red[chunk * 64 + lane] = dot;
            __syncthreads();
            if (c < r) {
                const double w = beta * ((red[lane] + red[64 + lane]) + (red[128 + lane] + red[192 + lane]));
                for (int j = k + chunk; j < n; j += 4) {
                    const double vj = (j == k) ? vkk : A[(long)j * r + k];
                    A[(long)j * r + c] = fma(-vj, w, A[(long)j * r + c]);
                }
            }
            __syncthreads();
        }
        if (tid == 0) A[(long)k * r + k] = sh_alpha;  // row k of A now holds row k of R
        __syncthreads();
    }
    for (int e = tid; e < n2; e += 256) {
        const int a = e / r, b = e - a * r;
        R[e] = (b >= a) ? A[(long)a * r + b] : 0.0;
        V[e] = (a == b) ? 1.0 : 0.0;
    }
    __syncthreads();
    if (chunk == 0) {  // one wave: lane i owns row i of R and of V (r <= 64)
        const bool act = lane < r;
        const int i = act ? lane : 0;
        for (int sweep = 0; sweep < 40; ++sweep) {
            int rotated = 0;
            for (int pc = 0; pc < r - 1; ++pc)
                for (int qc = pc + 1; qc < r; ++qc) {
                    const double rp = act ? R[i * r + pc] : 0.0, rq = act ? R[i * r + qc] : 0.0;
                    const double al = wave_sum_d(rp * rp), be = wave_sum_d(rq * rq), ga = wave_sum_d(rp * rq);
                    if (fabs(ga) > 1e-15 * sqrt(al * be) && ga != 0.0) {  // wave-uniform
                        const double zeta = (be - al) / (2.0 * ga);
                        const double tt = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                        const double cs = 1.0 / sqrt(1.0 + tt * tt), sn = cs * tt;
                        if (act) {
                            R[i * r + pc] = cs * rp - sn * rq;
                            R[i * r + qc] = sn * rp + cs * rq;
                            const double vp = V[i * r + pc], vq = V[i * r + qc];
                            V[i * r + pc] = cs * vp - sn * vq;
                            V[i * r + qc] = sn * vp + cs * vq;
                        }
                        rotated = 1;
                    }
                }
            if (!rotated) break;
        }
        double smax = 0.0;
        for (int c = 0; c < r; ++c) {
            const double rc = act ? R[i * r + c] : 0.0;
            const double sg = sqrt(wave_sum_d(rc * rc));
            if (lane == 0) lam[c] = sg;
            smax = fmax(smax, sg);
        }
        if (act) lam[lane] = (lam[lane] > 1e-14 * smax && lam[lane] > 0.0) ? 1.0 / lam[lane] : 0.0;  // pseudo-inverse
    }
    __syncthreads();
    // W = V diag(lam) V^T -> R;  T = D^T W -> V's place is still needed for W, so T goes to the red-free part of A's first rows?  no:
    // r x r results fit the LDS arrays in turn: W -> R, then T -> D is NOT possible (T needs D): T -> A (global scratch, n >= r rows)
    for (int e = tid; e < n2; e += 256) {
        const int a = e / r, b = e - a * r;
        double sum = 0.0;
        for (int k = 0; k < r; ++k) sum = fma(V[a * r + k] * lam[k], V[b * r + k], sum);
        R[e] = sum;
    }
    __syncthreads();
    for (int e = tid; e < n2; e += 256) {
        const int a = e / r, b = e - a * r;
        double sum = 0.0;
        for (int k = 0; k < r; ++k) sum = fma(D[k * r + a], R[k * r + b], sum);
        V[e] = sum;  // T = Delta^T W
    }
    __syncthreads();
    const double rh = (double)rho[slab];
    const double *Sm = S + (long)slab * n2;
    for (int e = tid; e < n2; e += 256) {
        const int a = e / r, b = e - a * r;
        double sum = 0.0;
        for (int k = 0; k < r; ++k) sum = fma(V[k * r + a], Sm[k * r + b], sum);
        acc_out[(long)slab * (n2 + 1) + e] = rh * sum;
        T[(long)slab * n2 + e] = (float)V[e];
        if (T64 != nullptr) T64[(long)slab * n2 + e] = V[e];
    }
    if (tid == 0) acc_out[(long)slab * (n2 + 1) + n2] = rh;
}

// ---------------------------------------------------------------------------------------------------------
// PARAFAC2 r x r algebra on the fp64 MFMA (one wave per slab), replacing the Jacobi eigen-solver in the common case:
//   G = Delta S Delta^T,  s = tr G,  Newton-Schulz:  Y_0 = G/s, Z_0 = I,  T = (3I - Z Y)/2,  Y <- Y T,  Z <- T Z
//   -> Z = (G/s)^-1/2,  W = Z / sqrt(s),  T_i = Delta^T W,  acc_i = rho_i T_i^T S.
// The D layout of v_mfma_f64_16x16x4_f64 (lane l, reg v: row = (l>>4) + 4v, col = l&15) is directly the B-operand
// layout (B[k = (l>>4) + 4 step][n = l&15], step = v) and, read as an A operand (A[i = l&15][k = (l>>4) + 4 step]),
// supplies the TRANSPOSE: two accumulator-layout matrices M1, M2 give M1^T M2 with no data movement.  The iterates
// are symmetric only up to rounding, and treating them as exactly symmetric destabilises Newton-Schulz for
// cond(G) > ~1e3; so every iterate is carried together with its transpose (Y, Yt, Z, Zt) and both are advanced with
// the products that are available (6 small matmuls per iteration) - stable up to cond(G) ~ 1e11 (tools/ check).
// status[slab] = 1 if the iteration did not converge (rank-deficient / extremely ill-conditioned Y_i Delta^T, or
// J_i < r): those slabs are redone by k_pf2_algebra (Jacobi, pseudo-inverse square root).
// ---------------------------------------------------------------------------------------------------------
template <int NB>
struct SymTiles {
    f64x4 t[NB][NB];
};

// C = A^T B for A and B in D layout; result in D layout
template <int NB>
static __device__ __forceinline__ void mm_t(const SymTiles<NB> &A, const SymTiles<NB> &B, SymTiles<NB> &C) {
#pragma unroll
    for (int a = 0; a < NB; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kb = 0; kb < NB; ++kb)
#pragma unroll
                for (int st = 0; st < 4; ++st)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A.t[kb][a][st], B.t[kb][b][st], acc, 0, 0, 0);
            C.t[a][b] = acc;
        }
}

// At = A^T for a matrix in D layout, through a wave-private LDS area of (16 NB) x (16 NB + 1) doubles (the kernel runs ONE wave
// per workgroup; LDS operations of a wave complete in order, so consecutive transposes need no barrier).  Round 3: the
// Newton-Schulz iteration used to carry every iterate WITH its transpose and advance both by matrix products (6 per step);
// the fp64 matrix pipe is what bounds the kernel (one v_mfma_f64_16x16x4_f64 per 143 cycles and wave, 44 TFLOP/s for the
// whole part: tools/mfma64_rate.hip), so the transposes now cost 8 NB^2 LDS operations instead of 4 NB^3 MFMAs each.
template <int NB>
static __device__ __forceinline__ void tr_lds(const SymTiles<NB> &A, SymTiles<NB> &At, double *W, int q, int c16) {
    constexpr int LD = 16 * NB + 1;
#pragma unroll
    for (int a = 0; a < NB; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int v = 0; v < 4; ++v) W[(16 * a + q + 4 * v) * LD + 16 * b + c16] = A.t[a][b][v];
#pragma unroll
    for (int a = 0; a < NB; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int v = 0; v < 4; ++v) At.t[a][b][v] = W[(16 * b + c16) * LD + 16 * a + q + 4 * v];
}

// TILES: the per-tile statistics of the solve pass (k_rows_solve_stats / k_rows_finish_solve_stats) are summed HERE, in
// k_stats_reduce's order - S_i goes to LDS (and to S for the Jacobi fallback), the L2-ball column sums to colsq - which
// saves the separate reduction launch in front of this kernel in every inner iteration.
struct TileStats {
    const int *slab_tile_ptr;
    const double *stat_gram, *stat_colsq;
    double *colsq;
    int W, n_slabs;
};
// -DMCL_NS_STAMPS (tools/ns_stamps.py; never in the shipped library): s_memtime stamps of the kernel's sections per slab -
// [0] entry, [1] statistics summed, [2] G formed, [3] iteration done, [4] exit, [5] steps - in the scratch of pf2_acc's tail
#ifdef MCL_NS_STAMPS
#define NS_STAMP(i) do { if (lane == 0) stamps[(long)slab * 8 + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define NS_STAMP(i) do { } while (0)
#endif
// Slabs per workgroup.  Rank <= 16 (config 4: 1024 slabs = one per SIMD, the kernel lasts as long as its slowest slab): FOUR, one
// wave each.  The waves of a workgroup are placed on the four SIMDs of its CU in turn, whereas the SIMD of a one-wave workgroup is
// the dispatcher's choice - and with 4 such workgroups per CU it put two Newton-Schulz waves on one SIMD for 140 of 1024 slabs
// (in-kernel HW_ID stamps, tools/ns_stamps.py: 70 SIMDs with two waves, 70 idle), where two fp64-MFMA chains run 1.3-1.5 x
// slower each (2520 against 1900 cycles per step; tools/ns_valu_probe.hip: 1355 -> 1980 for the bare step).
template <int NB>
struct NsShape {
    static constexpr int SPW = NB == 1 ? 4 : 1;
};
template <int NB, bool TILES>
__global__ __launch_bounds__(64 * NsShape<NB>::SPW + ((TILES && NB == 1) ? 64 : 0)) __attribute__((amdgpu_waves_per_eu(2))) void k_pf2_algebra_ns(double *__restrict__ S, const float *__restrict__ Delta,
                                                       const float *__restrict__ rho, const int *__restrict__ ext, int r,
                                                       float *__restrict__ T, double *__restrict__ acc_out,
                                                       int *__restrict__ status, TileStats ts, RegSet regs,
                                                       float *__restrict__ xmin_est, int scaled, double *__restrict__ T64) {
    MCL_GATE(regs.gate);
    // S_i while the system is formed (TILES), then the scratch of the LDS transposes (the epilogue re-reads S_i from memory)
    constexpr int SPW = NsShape<NB>::SPW;
    __shared__ double Ssm_w[SPW][(16 * NB) * (16 * NB + 1)];
    __shared__ float Dsm_w[SPW][256 * NB * NB];
    // rank <= 16: the Jacobi route of a slab this iteration cannot handle runs right here (8 KB of LDS; saves the launch of
    // the stand-alone kernel in every inner iteration); rank 32 would need 33 KB and lose a wave per CU, so those slabs are
    // left to k_pf2_algebra (status = 1)
    constexpr bool INK = NB == 1;
    __shared__ double Jsm_w[INK ? SPW : 1][INK ? 4 * 256 + 16 : 1];
    const int wave = SPW == 1 ? 0 : (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;
    // (the column-sum wave has its own slabs below)
    const int slab = SPW == 1 ? (int)blockIdx.x : min((int)blockIdx.x * SPW + (wave < SPW ? wave : 0), ts.n_slabs - 1);
    auto &Ssm = Ssm_w[wave < SPW ? wave : 0];
    // scratch of the LDS transposes.  Rank <= 16: an area of its own, so that S_i stays in LDS for the epilogue's T^T S (its re-read
    // from memory was a round trip at the end of every slab's chain); rank 32 has no LDS to spare (8 workgroups per CU) and uses
    // S_i's area once U1 has consumed it
    __shared__ double Wsm_w[NB == 1 ? SPW : 1][NB == 1 ? 16 * 17 : 1];
    double *Wtr;
    if constexpr (NB == 1) Wtr = Wsm_w[wave < SPW ? wave : 0];
    else Wtr = Ssm;
    auto &Dsm = Dsm_w[wave < SPW ? wave : 0];
    auto &Jsm = Jsm_w[(INK && wave < SPW) ? wave : 0];
    const int q = lane >> 4, c16 = lane & 15;
    const int n2 = r * r;
    const double *Ss = S + (long)slab * n2;
    // The L2-ball column sums of the slab's tiles, in k_stats_reduce's order: two more memory round trips.  Rank <= 16 (one
    // slab per SIMD at config 4: the chain's latency is the kernel's duration): a SECOND wave takes them and is done; rank 32
    // (register-bound at two waves per SIMD: a second wave would halve the resident slabs) keeps them in front of the chain.
    // (round 6: that wave is the FIFTH of the workgroup and takes the column sums of its four slabs, 16 lanes each)
    constexpr bool COLSQ_WAVE = TILES && NB == 1;
    if (TILES && (COLSQ_WAVE ? wave == SPW : true)) {
        const int cslab = COLSQ_WAVE ? blockIdx.x * SPW + (lane >> 4) : slab;
        const bool cs_ok = cslab < ts.n_slabs;
        const int t0 = cs_ok ? ts.slab_tile_ptr[cslab] : 0, t1 = cs_ok ? ts.slab_tile_ptr[cslab + 1] : 0;
        for (int k = 0; k < regs.n; ++k) {
            if (regs.kind[k] != MCL_PEN_L2BALL) continue;
            for (int col = COLSQ_WAVE ? (lane & 15) : lane; col < r; col += (COLSQ_WAVE ? 16 : 64)) {
                double sq = 0.0;
                for (int tb = t0; tb < t1; tb += 8) {  // ascending order, 8 loads in flight
                    double v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = ts.stat_colsq[((long)min(tb + u, t1 - 1) * MCL_MAX_REGS + k) * r + col];
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (tb + u < t1) sq += v[u];
                }
                if (cs_ok) ts.colsq[((long)k * ts.n_slabs + cslab) * r + col] = sq;
            }
        }
        if (COLSQ_WAVE) return;
    }
    if (SPW > 1 && (int)blockIdx.x * SPW + wave >= ts.n_slabs) return;  // the last workgroup's spare waves (nothing below is a workgroup barrier)
#ifdef MCL_NS_STAMPS
    long long *stamps = reinterpret_cast<long long *>(xmin_est + ts.n_slabs);  // the plan reserves 8 int64 per slab behind pf2_xmin
#endif
    NS_STAMP(0);
#ifdef MCL_NS_STAMPS  // where the wave runs: HW_REG_HW_ID (wave, SIMD, CU, SH, SE) and HW_REG_XCC_ID
    if (lane == 0) {
        stamps[(long)slab * 8 + 6] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 4);
        stamps[(long)slab * 8 + 7] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 20);
    }
#endif
    // everything this wave needs besides the statistics is requested up front, so its latency overlaps the tile loop:
    // the slab's extent and weight, and Delta (staged in LDS; every product below reads it from there)
    const int slab_rows = ext[slab + 1] - ext[slab];
    const double rh = (double)rho[slab];
    {
        float dv[4 * NB * NB];
#pragma unroll
        for (int m = 0; m < 4 * NB * NB; ++m) dv[m] = Delta[min(lane + 64 * m, n2 - 1)];
#pragma unroll
        for (int m = 0; m < 4 * NB * NB; ++m)
            if (lane + 64 * m < n2) Dsm[lane + 64 * m] = dv[m];
    }
    if (TILES) {
        const int t0 = ts.slab_tile_ptr[slab], t1 = ts.slab_tile_ptr[slab + 1];
        const long WW = (long)ts.W * ts.W;
        // the order of k_stats_reduce (even tile offsets -> s0, odd -> s1, ascending), tiles fetched TB at a time with
        // independent clamped loads (one memory latency per batch instead of one per tile)
        constexpr int EPL = 4 * NB * NB, TB = NB == 1 ? 16 : 2;  // elements per lane (n2 <= 256 NB^2); register budget
        // (rank <= 16: a slab of <= 1024 rows is ONE batch - one memory round trip instead of two in front of the chain)
        double s0[EPL], s1[EPL];
        long off[EPL];
#pragma unroll
        for (int m = 0; m < EPL; ++m) {
            const int e = min(lane + 64 * m, n2 - 1);
            const int a = e / r, b = e - a * r;
            off[m] = (long)a * ts.W + b;
            s0[m] = 0.0, s1[m] = 0.0;
        }
        for (int tb = t0; tb < t1; tb += TB) {
            double v[TB][EPL];
#pragma unroll
            for (int u = 0; u < TB; ++u)
#pragma unroll
                for (int m = 0; m < EPL; ++m) v[u][m] = ts.stat_gram[min(tb + u, t1 - 1) * WW + off[m]];
#pragma unroll
            for (int u = 0; u < TB; ++u)
                if (tb + u < t1) {
#pragma unroll
                    for (int m = 0; m < EPL; ++m) {
                        if (((tb + u - t0) & 1) == 0) s0[m] += v[u][m];
                        else s1[m] += v[u][m];
                    }
                }
        }
#pragma unroll
        for (int m = 0; m < EPL; ++m) {
            const int e = lane + 64 * m;
            if (e < n2) {
                Ssm[e] = s0[m] + s1[m];
                S[(long)slab * n2 + e] = s0[m] + s1[m];
            }
        }
    }
    if constexpr (SPW == 1) __syncthreads();  // (a one-wave workgroup: the compiler drops the barrier instruction)
    else wave_sync_lds();
    NS_STAMP(1);
    auto Sat = [&](int i, int j) -> double {
        if (TILES) return (i < r && j < r) ? Ssm[i * r + j] : 0.0;
        return (i < r && j < r) ? Ss[i * r + j] : 0.0;
    };
    auto Dat = [&](int i, int j) -> double { return (i < r && j < r) ? (double)Dsm[i * r + j] : 0.0; };
    const double *Sany = TILES ? static_cast<const double *>(Ssm) : Ss;
    if (slab_rows < r) {  // fewer rows than columns: rank-deficient by construction
        if (lane == 0) status[slab] = 1;
        if (INK) pf2_jacobi_slab(Jsm, Sany, Delta, rh, r, slab, lane, T, acc_out, T64);
        return;
    }
    // U1 = S Delta^T   (A = S, symmetric: A[i][k] = S[k][i];  B[k][n] = Delta[n][k])
    SymTiles<NB> U1, G;
#pragma unroll
    for (int a = 0; a < NB; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kb = 0; kb < NB; ++kb)
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    const int k = 16 * kb + q + 4 * st;
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Sat(k, 16 * a + c16), Dat(16 * b + c16, k), acc, 0, 0, 0);
                }
            U1.t[a][b] = acc;
        }
    // G = Delta U1 (A[i][k] = Delta[i][k];  B = U1 in D layout)  and  Gt = G^T = U1^T Delta^T (A = U1 as A-operand,
    // B[k][n] = Delta[n][k])
    SymTiles<NB> Gt;
    double tr = 0.0;
#pragma unroll
    for (int a = 0; a < NB; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kb = 0; kb < NB; ++kb)
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    const int k = 16 * kb + q + 4 * st;
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Dat(16 * a + c16, k), U1.t[kb][b][st], acc, 0, 0, 0);
                }
            G.t[a][b] = acc;
            if (a == b) {
#pragma unroll
                for (int v = 0; v < 4; ++v)
                    if (q + 4 * v == c16 && 16 * a + c16 < r) tr += acc[v];
            }
        }
    tr = wave_sum_d(tr);
    tr_lds<NB>(G, Gt, Wtr, q, c16);  // G^T (S_i has been consumed by U1: its LDS copy is free; the fallbacks below read memory)
    if (!(tr > 0.0)) {
        if (lane == 0) status[slab] = 1;
        if (INK) pf2_jacobi_slab(Jsm, Ss, Delta, rh, r, slab, lane, T, acc_out, T64, status, true);  // may flag the slab (2)
        return;
    }
    NS_STAMP(2);
    // Newton-Schulz for the inverse square root of G / tr (padding rows/cols >= r carry the identity)
    SymTiles<NB> Y, Yt, Z, Zt, P, Pt, Tm, Tmt, N1, N2, N3, N4;
    const double inv_s = 1.0 / tr;
#pragma unroll
    for (int a = 0; a < NB; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int row = 16 * a + q + 4 * v, col = 16 * b + c16;
                const double id = (row == col) ? 1.0 : 0.0;
                const bool in = (row < r && col < r);
                Y.t[a][b][v] = in ? G.t[a][b][v] * inv_s : id;
                Yt.t[a][b][v] = in ? Gt.t[a][b][v] * inv_s : id;
                Z.t[a][b][v] = id;
                Zt.t[a][b][v] = id;
            }
    bool converged = false;
    double prev = 1e300;
    int it_used = 0;
    // Scaled iteration: with x = sqrt(eig(Z Y)) in [lo, hi] the step T = a I - b Z Y maps x to x (a - b x^2); (a, b) are
    // the minimax coefficients of that cubic on [lo, hi] (equi-oscillation at lo, sqrt(s/3), hi; (1.5, 0.5) when lo = hi = 1),
    // which roughly doubles the growth of the small eigenvalues per step (2.6 x instead of 1.5 x) and so halves the
    // iteration count.  hi = 1 holds by the trace scaling; lo starts from the estimate 1 / ||Z||_F left by the previous
    // call for this slab (the matrices change slowly between inner iterations; 1e-2 when there is none).  An eigenvalue
    // below the assumed lo still grows by a every step (the cubic is increasing there) and nothing exceeds 1 + e, so a wrong
    // estimate costs iterations, not convergence.
    // Round 3: the interval is PREDICTED ([1 - e, 1 + e] after every step, e from the coefficients) instead of measured: the
    // coefficient chain - three square roots and a division, formerly in fp64 behind the wave reduction of ||I - Z Y||^2,
    // 1.4 k of a step's 3.0 k cycles - is a handful of fp32 instructions that do not depend on this step's product, and
    // the residual is only reduced once the prediction says the end is near (e < 0.01).  Below e = 1e-3 the step is the
    // plain Newton-Schulz one with EXACT coefficients (a - b = 1 is what fixes x = 1; fp32 coefficients would leave the
    // fixed point at 1 + 1e-7), where the minimax scaling has nothing left to gain.
    float flo = 1.f, fhi = 1.f;
    if (scaled) {
        const float est = xmin_est[slab];
        flo = est > 0.f ? fminf(0.5f * est, 1.f) : 1e-2f;
    }
    // (the interval, the residual and everything decided from them are the same in all lanes: kept in scalar registers, so that the
    // branches of a step are scalar branches and not the exec-mask sequences of a divergent one)
    // (rank <= 16 only: the rank-32 kernel sits at its 256-register budget and the extra scalar traffic costs it spills)
    auto uni_f = [](float v) { return NB == 1 ? __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))) : v; };
    flo = uni_f(flo);
    for (int it = 0; it < 100; ++it) {
        it_used = it;
        double ca = 1.5, cb = 0.5;
        const bool watch = !scaled || (1.f - flo) < 1e-2f;  // the end is near: measure the residual from here on
        if (scaled && (1.f - flo) >= 1e-3f) {
            // (the hardware's own square root and reciprocal - one instruction each, 1 ulp - where the library forms spent ~45
            // instructions per step on correct rounding and denormals: these coefficients only steer the speed of convergence)
            // (rank <= 16, where the step's latency is the kernel's; the rank-32 kernel keeps the forms its register allocation was
            // tuned with)
            const float ss = fhi * fhi + fhi * flo + flo * flo;
            const float sq = NB == 1 ? __builtin_amdgcn_sqrtf(ss * (1.f / 3.f)) : __builtin_sqrtf(ss * (1.f / 3.f));
            const float den = (2.f / 3.f) * ss * sq + flo * fhi * (flo + fhi);
            const float fb = NB == 1 ? 2.f * __builtin_amdgcn_rcpf(den) : 2.f / den;
            const float fa = fb * ss;
            const float e = (2.f / 3.f) * fa * sq - 1.f;
            ca = (double)fa, cb = (double)fb;
            flo = uni_f(1.f - e), fhi = uni_f(1.f + e);
        } else if (scaled) {
            const float e = 1.f - flo;
            flo = uni_f(1.f - 1.5f * e * e), fhi = 1.f;  // plain Newton-Schulz: x -> x (3 - x^2) / 2 <= 1, error 1.5 e^2
        }
        mm_t<NB>(Zt, Y, P);             // P  = Z Y
        tr_lds<NB>(P, Pt, Wtr, q, c16);  // Pt = P^T
        double res = 1.0;
        if (watch) {
            res = 0.0;
#pragma unroll
            for (int a = 0; a < NB; ++a)
#pragma unroll
                for (int b = 0; b < NB; ++b)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int row = 16 * a + q + 4 * v, col = 16 * b + c16;
                        const double id = (row == col) ? 1.0 : 0.0;
                        const double d = id - P.t[a][b][v];
                        res += d * d;
                    }
            res = wave_sum_d(res);
            if constexpr (NB == 1)
                res = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(res)), __builtin_amdgcn_readfirstlane(__double2loint(res)));
        }
#pragma unroll
        for (int a = 0; a < NB; ++a)
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int row = 16 * a + q + 4 * v, col = 16 * b + c16;
                    const double id = (row == col) ? 1.0 : 0.0;
                    Tm.t[a][b][v] = ca * id - cb * P.t[a][b][v];
                    Tmt.t[a][b][v] = ca * id - cb * Pt.t[a][b][v];
                }
        // converged: ||I - Z Y||_F < 1e-12 sqrt(r), or stagnation at the fp64 round-off floor of an ill-conditioned G
        // (the floor grows with cond(G); 1e-8 in ||I - ZY|| still leaves W accurate far beyond the fp32 data)
        if (watch) {
            if (res < 1e-24 * r || (res < 1e-16 && res > 0.25 * prev)) {
                converged = true;
                break;
            }
            prev = res;
            // a lower bound that was too optimistic (an eigenvalue below the assumed lo): the prediction is ahead of the
            // iterate - fall back to measuring, i.e. keep the predicted interval no tighter than the certified one
            // (|1 - x^2| <= ||I - Z Y||_F for every eigenvalue x^2 of Z Y, so x >= 1 - ||I - Z Y||_F)
            if (scaled) flo = uni_f(fminf(flo, res < 1.0 ? 1.f - (NB == 1 ? __builtin_amdgcn_sqrtf((float)res) : __builtin_sqrtf((float)res)) : 0.1f));
        }
        mm_t<NB>(Yt, Tm, N1);            // Y  <- Y T
        mm_t<NB>(Tmt, Z, N3);            // Z  <- T Z
        tr_lds<NB>(N1, N2, Wtr, q, c16);  // Yt <- Y^T
        tr_lds<NB>(N3, N4, Wtr, q, c16);  // Zt <- Z^T
        Y = N1;
        Yt = N2;
        Z = N3;
        Zt = N4;
    }
    if (!converged) {
        if (lane == 0) status[slab] = 1;
        if (INK) pf2_jacobi_slab(Jsm, Ss, Delta, rh, r, slab, lane, T, acc_out, T64, status, true);  // (the LDS copy of S is gone: from memory); may flag the slab (2)
        return;
    }
    NS_STAMP(3);
#ifdef MCL_NS_STAMPS
    if (lane == 0) stamps[(long)slab * 8 + 5] = it_used;
#endif
    if (lane == 0) status[slab] = -it_used;  // <= 0: converged (number of Newton-Schulz iterations, for diagnostics)
    if (scaled) {  // 1 / ||Z||_F <= 1 / ||Z||_2 = sqrt(eig_min(G / tr)): the next call's starting estimate for this slab
        double zn = 0.0;
#pragma unroll
        for (int a = 0; a < NB; ++a)
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int v = 0; v < 4; ++v) zn += Z.t[a][b][v] * Z.t[a][b][v];
        zn = wave_sum_d(zn);
        if (lane == 0) xmin_est[slab] = (float)(1.0 / sqrt(zn));
    }
    // S_i for the epilogue: from memory (this wave wrote it above when it summed the tiles; its LDS copy has been the scratch of
    // the transposes since)
    auto Smem = [&](int i, int j) -> double {
        if constexpr (NB == 1 && TILES) return (i < r && j < r) ? Ssm[i * r + j] : 0.0;  // (still in LDS: see Wtr)
        else return (i < r && j < r) ? Ss[i * r + j] : 0.0;
    };
    const double wscale = 1.0 / sqrt(tr);  // W = Z / sqrt(tr)
    // T = Delta^T W  (A[i][k] = Delta[k][i];  B = W in D layout), then acc = rho T^T S (A = T^T: A[i][k] = T[k][i] = D layout of T)
    SymTiles<NB> Tt;
#pragma unroll
    for (int a = 0; a < NB; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kb = 0; kb < NB; ++kb)
#pragma unroll
                for (int st = 0; st < 4; ++st)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Dat(16 * kb + q + 4 * st, 16 * a + c16), Z.t[kb][b][st] * wscale, acc, 0, 0, 0);
            Tt.t[a][b] = acc;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int row = 16 * a + q + 4 * v, col = 16 * b + c16;
                if (row < r && col < r) {
                    T[(long)slab * n2 + row * r + col] = (float)acc[v];
                    if (T64 != nullptr) T64[(long)slab * n2 + row * r + col] = acc[v];  // fp64 row passes (mcl_rows64)
                }
            }
        }
#pragma unroll
    for (int a = 0; a < NB; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kb = 0; kb < NB; ++kb)
#pragma unroll
                for (int st = 0; st < 4; ++st)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Tt.t[kb][a][st], Smem(16 * kb + q + 4 * st, 16 * b + c16), acc, 0, 0, 0);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int row = 16 * a + q + 4 * v, col = 16 * b + c16;
                if (row < r && col < r) acc_out[(long)slab * (n2 + 1) + row * r + col] = rh * acc[v];
            }
        }
    if (lane == 0) acc_out[(long)slab * (n2 + 1) + n2] = rh;
    NS_STAMP(4);
}

// P = Y T_slab
template <int NBR, bool VEC>
__global__ __launch_bounds__(256) void k_pf2_apply(ModeView mv, const float *__restrict__ U, const float *__restrict__ T,
                                                   float *__restrict__ P, int r) {
    TILE_PROLOGUE();
    RowMat<NBR> Ts;
    Ts.load(T + (long)slab * r * r, r, lane);
    FOR_ROW_BLOCKS() {
        const bool ok = 16 * rb + row16 < nrows;
        const long j = row0 + 16 * rb + (ok ? row16 : 0);
        // Y = F + U as an unevaluated sum y + yl of two floats (Knuth's two-sum: exact), P = y T + yl T: the polar factor of the
        // SAME matrix whose Gram k_pf2_gram formed, not of its fp32 rounding
        f32x4 y[NBR], yl[NBR], p[NBR], pl[NBR];
#pragma unroll
        for (int h = 0; h < NBR; ++h) {
            const f32x4 f = row_ld4<VEC>(mv.F, j, 16 * h + 4 * g, ok, r);
            const f32x4 u = row_ld4<VEC>(U, j, 16 * h + 4 * g, ok, r);
            y[h] = f + u;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float bb = y[h][v] - f[v];
                yl[h][v] = (f[v] - (y[h][v] - bb)) + (u[v] - bb);
            }
        }
        Ts.apply(y, p);
        Ts.apply(yl, pl);
#pragma unroll
        for (int h = 0; h < NBR; ++h) row_st4<VEC>(P, j, 16 * h + 4 * g, ok, r, p[h] + pl[h]);
    }
}

// red[e] = sum_i acc[i][e]: one workgroup per element, fixed summation order
__global__ __launch_bounds__(256) void k_pf2_sum(const double *__restrict__ acc, int n_slabs, int n_el,
                                                 float *__restrict__ red) {
    __shared__ double sm[4];
    const int e = blockIdx.x;
    double s = 0.0;
    for (int i = threadIdx.x; i < n_slabs; i += 256) s += acc[(long)i * n_el + e];
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) red[e] = (float)((sm[0] + sm[1]) + (sm[2] + sm[3]));
}

// k_pf2_sum + k_pf2_delta in one launch (single-process runs: no all-reduce between them).  Workgroup e sums element e
// and the weight total with the same order and roundings as the two kernels.
__global__ __launch_bounds__(256) void k_pf2_sum_delta(const double *__restrict__ acc, int n_slabs, int n2,
                                                       float *__restrict__ red, float *__restrict__ Delta,
                                                       const int *__restrict__ gate) {
    MCL_GATE(gate);
    __shared__ double sm[2][4];
    const int e = blockIdx.x, n_el = n2 + 1;
    double s = 0.0, w = 0.0;
    for (int i = threadIdx.x; i < n_slabs; i += 256) {
        s += acc[(long)i * n_el + e];
        w += acc[(long)i * n_el + n2];
    }
    s = wave_sum_d(s);
    w = wave_sum_d(w);
    if ((threadIdx.x & 63) == 0) sm[0][threadIdx.x >> 6] = s, sm[1][threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float re = (float)((sm[0][0] + sm[0][1]) + (sm[0][2] + sm[0][3]));
        const float rw = (float)((sm[1][0] + sm[1][1]) + (sm[1][2] + sm[1][3]));
        red[e] = re;
        if (e == 0) red[n2] = rw;
        Delta[e] = re / rw;
    }
}
__global__ void k_pf2_delta(const float *__restrict__ red, int r, float *__restrict__ Delta, const int *__restrict__ gate) {
    MCL_GATE(gate);
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < r * r) Delta[e] = red[e] / red[r * r];
}

// dual update of the PARAFAC2 penalty: U = F - (P Delta - U)
template <int NBR, bool VEC>
__global__ __launch_bounds__(256) void k_rows_pf2_dual(ModeView mv, RegSet regs, int k, int r) {
    TILE_PROLOGUE();
    RowMat<NBR> D;
    D.load(regs.aux2[k], r, lane);
    FOR_ROW_BLOCKS() {
        const bool ok = 16 * rb + row16 < nrows;
        const long j = row0 + 16 * rb + (ok ? row16 : 0);
        f32x4 p[NBR], z[NBR];
#pragma unroll
        for (int h = 0; h < NBR; ++h) p[h] = row_ld4<VEC>(regs.aux[k], j, 16 * h + 4 * g, ok, r);
        D.apply(p, z);
#pragma unroll
        for (int h = 0; h < NBR; ++h) {
            const int col = 16 * h + 4 * g;
            const f32x4 f = row_ld4<VEC>(mv.F, j, col, ok, r);
            f32x4 u = row_ld4<VEC>(regs.dual[k], j, col, ok, r);
#pragma unroll
            for (int v = 0; v < 4; ++v) u[v] = f[v] - (z[h][v] - u[v]);
            row_st4<VEC>(regs.dual[k], j, col, ok, r, u);
        }
    }
}

// plain dual update of penalty k: U = F - (Z - U)   (decomposition.py:282-285)
template <int NBR, bool VEC>
__global__ __launch_bounds__(256) void k_rows_dual(ModeView mv, RegSet regs, int k, int r) {
    TILE_PROLOGUE();
    FOR_ROW_BLOCKS() {
        const bool ok = 16 * rb + row16 < nrows;
        const long j = row0 + 16 * rb + (ok ? row16 : 0);
#pragma unroll
        for (int h = 0; h < NBR; ++h) {
            const int col = 16 * h + 4 * g;
            const f32x4 f = row_ld4<VEC>(mv.F, j, col, ok, r);
            const f32x4 z = row_ld4<VEC>(regs.aux[k], j, col, ok, r);
            f32x4 u = row_ld4<VEC>(regs.dual[k], j, col, ok, r);
#pragma unroll
            for (int v = 0; v < 4; ++v) u[v] = f[v] - (z[v] - u[v]);
            row_st4<VEC>(regs.dual[k], j, col, ok, r, u);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// ONE row pass for the prox + dual steps of a whole penalty stack made of row-separable kinds, L2 balls and PARAFAC2,
// once the per-slab statistics are known (column norms -> colsq; polar factor T_i and the new Delta): F is read once,
// P = (F + U) T_i stays in registers for its dual update.  Replaces k_pf2_apply + k_rows_pf2_dual + k_rows_l2ball
// (+ k_rows_prox_rowsep), i.e. three to four passes over the B-sized arrays per inner iteration.
// ---------------------------------------------------------------------------------------------------------
template <int NBR, bool VEC, bool R64 = false>
__global__ __launch_bounds__(256) void k_rows_finish_fused(ModeView mv, RegSet regs, int r, const float *__restrict__ T,
                                                           const double *__restrict__ colsq,
                                                           double *__restrict__ diag_tile, int want_diag,
                                                           const double *__restrict__ T64 = nullptr) {
    typedef RowArith<R64> RA;  // R64: the r x r products on the fp64 MFMA, exact Y = F + U (see k_rows_solve_stats)
    TILE_PROLOGUE();
    double nf = 0.0, na = 0.0, gap[MCL_MAX_REGS];  // per-tile diagnostics (same sums as k_rows_diag)
#pragma unroll
    for (int k = 0; k < MCL_MAX_REGS; ++k) gap[k] = 0.0;
    const float rho = mv.rho[slab];
    int kpf2 = -1;
    for (int k = 0; k < regs.n; ++k)
        if (regs.kind[k] == MCL_PEN_PARAFAC2) kpf2 = k;
    typename RA::template Mat<NBR> Ts, D;
    if (kpf2 >= 0) {
        if constexpr (R64) Ts.load(T64 + (long)slab * r * r, r, lane);
        else Ts.load(T + (long)slab * r * r, r, lane);
        D.load(regs.aux2[kpf2], r, lane);
    }
    FOR_ROW_BLOCKS() {
        const bool ok = 16 * rb + row16 < nrows;
        const long j = row0 + 16 * rb + (ok ? row16 : 0);
        f32x4 f[NBR];
#pragma unroll
        for (int h = 0; h < NBR; ++h) {
            f[h] = (row_ld4<VEC>(mv.F, j, 16 * h + 4 * g, ok, r));  // zeros for padding rows / columns
            if (want_diag) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    nf += (double)f[h][v] * (double)f[h][v];
                    na += fabs((double)f[h][v]);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < MCL_MAX_REGS; ++k) {
            if (k >= regs.n) continue;
            const int kind = regs.kind[k];
            f32x4 u[NBR], z[NBR], zg[NBR];  // zg: what the feasibility gap is measured against (P Delta for PARAFAC2)
#pragma unroll
            for (int h = 0; h < NBR; ++h) u[h] = (row_ld4<VEC>(regs.dual[k], j, 16 * h + 4 * g, ok, r));
            if (kind == MCL_PEN_PARAFAC2) {
                typename RA::Y y[NBR], pw[NBR];
                f32x4 pd[NBR];
#pragma unroll
                for (int h = 0; h < NBR; ++h) y[h] = RA::ysum(f[h], u[h]);
                Ts.apply(y, pw);   // P = Y T_i      (the aux variable)
                D.apply(pw, pd);   // P Delta        (what the dual is measured against)
#pragma unroll
                for (int h = 0; h < NBR; ++h) {
                    z[h] = RA::narrow(pw[h]);
                    zg[h] = pd[h];
#pragma unroll
                    for (int v = 0; v < 4; ++v) u[h][v] = f[h][v] - (pd[h][v] - u[h][v]);
                }
            } else if (kind == MCL_PEN_UNIMODAL) {  // aux rows already written by the column regressions
#pragma unroll
                for (int h = 0; h < NBR; ++h) {
                    z[h] = (row_ld4<VEC>(regs.aux[k], j, 16 * h + 4 * g, ok, r));
#pragma unroll
                    for (int v = 0; v < 4; ++v) u[h][v] = f[h][v] - (z[h][v] - u[h][v]);
                }
            } else if (kind == MCL_PEN_L2BALL) {
                const float bound = regs.p0[k];
#pragma unroll
                for (int h = 0; h < NBR; ++h)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int col = 16 * h + 4 * g + v;
                        const float nrm = (col < r) ? (float)sqrt(colsq[((long)k * mv.n_slabs + slab) * r + col]) : 1.f;
                        float y = f[h][v] + u[h][v];
                        if (regs.nonneg[k]) y = fmaxf(y, 0.f);
                        z[h][v] = y * (bound / fmaxf(nrm, bound));
                        u[h][v] = f[h][v] - (z[h][v] - u[h][v]);
                    }
            } else {
                const float thr = regs.p0[k] / rho;
#pragma unroll
                for (int h = 0; h < NBR; ++h)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        z[h][v] = prox_elem_g(kind, regs.nonneg[k], regs.p0[k], regs.p1[k], thr, f[h][v] + u[h][v]);
                        u[h][v] = f[h][v] - (z[h][v] - u[h][v]);
                    }
            }
#pragma unroll
            for (int h = 0; h < NBR; ++h) {
                if (kind != MCL_PEN_UNIMODAL) row_st4<VEC>(regs.aux[k], j, 16 * h + 4 * g, ok, r, z[h]);
                row_st4<VEC>(regs.dual[k], j, 16 * h + 4 * g, ok, r, u[h]);
                if (kind != MCL_PEN_PARAFAC2) zg[h] = z[h];
                if (want_diag)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const bool valid = ok && (16 * h + 4 * g + v < r);
                    const double dlt = valid ? (double)zg[h][v] - (double)f[h][v] : 0.0;
                    gap[k] += dlt * dlt;
                }
            }
        }
    }
    if (!want_diag) return;
    nf = wave_sum_d(nf);
    na = wave_sum_d(na);
#pragma unroll
    for (int k = 0; k < MCL_MAX_REGS; ++k) gap[k] = wave_sum_d(gap[k]);
    if (lane == 0) {
        double *o = diag_tile + (long)tile * DIAG_COLS;
        o[0] = nf;
        o[1] = na;
#pragma unroll
        for (int k = 0; k < MCL_MAX_REGS; ++k) o[2 + k] = gap[k];
    }
}

// ---------------------------------------------------------------------------------------------------------
// Inner iteration t -> t+1 of a fused stack in ONE row pass (mode 1, single process): the prox + dual steps of
// iteration t (k_rows_finish_fused) followed, on the same registers, by the solve of iteration t+1 and the statistics
// of its new rows (k_rows_solve_stats).  Z_k - U_k of every penalty stays in registers, so between two inner iterations
// the pass reads F, the duals, rhs (+ the aux rows the column regressions wrote) and writes F and the duals:
// (2 + 2n + 1) S_B instead of (2 + 2n) + (1 + 4n) S_B for the two separate passes.  The aux rows are not written here -
// no later step of the inner loop reads them (PARAFAC2's Delta and T_i carry its state); the LAST inner iteration ends
// with the plain k_rows_finish_fused, which writes them and the diagnostics.
// ---------------------------------------------------------------------------------------------------------
template <int NBR, bool VEC, bool R64 = false>
__global__ __launch_bounds__(256) void k_rows_finish_solve_stats(ModeView mv, const float *__restrict__ rhs_src,
                                                                 const float *__restrict__ Arows,
                                                                 const float *__restrict__ Linv, RegSet regs, int r,
                                                                 const float *__restrict__ T,
                                                                 const double *__restrict__ colsq,
                                                                 double *__restrict__ stat_gram,
                                                                 double *__restrict__ stat_colsq,
                                                                 const double *__restrict__ Linv64 = nullptr,
                                                                 const double *__restrict__ T64 = nullptr) {
    typedef double f64x4s __attribute__((ext_vector_type(4)));
    typedef RowArith<R64> RA;  // R64: the r x r products on the fp64 MFMA, exact Y = F + U (see k_rows_solve_stats)
    __shared__ double ytile[R64 ? 4 * 16 * 17 : 1];  // R64: one padded 16 x 16 fp64 tile per wave (row -> column layout of Y)
    TILE_PROLOGUE();
    const float rho = mv.rho[slab];
    int kpf2 = -1;
    for (int k = 0; k < regs.n; ++k)
        if (regs.kind[k] == MCL_PEN_PARAFAC2) kpf2 = k;
    typename RA::template Mat<NBR> L, Ts, D;
    if constexpr (R64) L.load(Linv64 + (long)slab * r * r, r, lane);
    else L.load(Linv + (long)slab * r * r, r, lane);
    if (kpf2 >= 0) {
        if constexpr (R64) Ts.load(T64 + (long)slab * r * r, r, lane);
        else Ts.load(T + (long)slab * r * r, r, lane);
        D.load(regs.aux2[kpf2], r, lane);
    }
    float av[NBR][4];
#pragma unroll
    for (int h = 0; h < NBR; ++h)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int col = 16 * h + 4 * g + v;
            av[h][v] = (Arows != nullptr && col < r) ? Arows[(long)slab * r + col] : 1.f;
        }
    // L2-ball scale factors of iteration t (from the per-slab column norms); the host chains stacks with at most ONE
    // L2 ball (register budget: two waves per SIMD at rank 32)
    int kl2 = -1;
    for (int k = 0; k < regs.n; ++k)
        if (regs.kind[k] == MCL_PEN_L2BALL) kl2 = k;
    float l2s[NBR][4];
#pragma unroll
    for (int h = 0; h < NBR; ++h)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            l2s[h][v] = 1.f;
            if (kl2 >= 0) {
                const int col = 16 * h + 4 * g + v;
                const float bound = regs.p0[kl2];
                const float nrm = (col < r) ? (float)sqrt(colsq[((long)kl2 * mv.n_slabs + slab) * r + col]) : 1.f;
                l2s[h][v] = bound / fmaxf(nrm, bound);
            }
        }
    typename std::conditional<R64, double, float>::type bsel[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) bsel[v] = (row16 == 4 * g + v) ? 1.f : 0.f;
    f64x4s accS[NBR][NBR];
#pragma unroll
    for (int a = 0; a < NBR; ++a)
#pragma unroll
        for (int b = 0; b < NBR; ++b) accS[a][b] = f64x4s{0.0, 0.0, 0.0, 0.0};
    double csq[NBR][4];
#pragma unroll
    for (int h = 0; h < NBR; ++h)
#pragma unroll
        for (int v = 0; v < 4; ++v) csq[h][v] = 0.0;
    FOR_ROW_BLOCKS() {
        const bool ok = 16 * rb + row16 < nrows;
        const long j = row0 + 16 * rb + (ok ? row16 : 0);
        f32x4 f[NBR], t[NBR], upf[NBR], ul2[NBR];  // new duals of the PARAFAC2 / L2-ball penalty (statistics below)
#pragma unroll
        for (int h = 0; h < NBR; ++h) {
            f[h] = (row_ld4<VEC>(mv.F, j, 16 * h + 4 * g, ok, r));  // zeros for padding rows / columns
            t[h] = (row_ld4<VEC>(rhs_src, j, 16 * h + 4 * g, ok, r));
#pragma unroll
            for (int v = 0; v < 4; ++v) t[h][v] *= av[h][v];
        }
        // ---- iteration t: prox + dual of every penalty (same arithmetic as k_rows_finish_fused)
#pragma unroll
        for (int k = 0; k < MCL_MAX_REGS; ++k) {
            if (k >= regs.n) continue;
            const int kind = regs.kind[k];
            f32x4 u[NBR], zg[NBR];
#pragma unroll
            for (int h = 0; h < NBR; ++h) u[h] = (row_ld4<VEC>(regs.dual[k], j, 16 * h + 4 * g, ok, r));
            if (kind == MCL_PEN_PARAFAC2) {
                typename RA::Y y[NBR], pz[NBR];
#pragma unroll
                for (int h = 0; h < NBR; ++h) y[h] = RA::ysum(f[h], u[h]);
                Ts.apply(y, pz);   // P = Y T_i
                D.apply(pz, zg);   // P Delta
            } else if (kind == MCL_PEN_UNIMODAL) {  // aux rows written by the column regressions
#pragma unroll
                for (int h = 0; h < NBR; ++h) zg[h] = (row_ld4<VEC>(regs.aux[k], j, 16 * h + 4 * g, ok, r));
            } else if (kind == MCL_PEN_L2BALL) {
#pragma unroll
                for (int h = 0; h < NBR; ++h)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        float y = f[h][v] + u[h][v];
                        if (regs.nonneg[k]) y = fmaxf(y, 0.f);
                        zg[h][v] = y * l2s[h][v];
                    }
            } else {
                const float thr = regs.p0[k] / rho;
#pragma unroll
                for (int h = 0; h < NBR; ++h)
#pragma unroll
                    for (int v = 0; v < 4; ++v)
                        zg[h][v] = prox_elem_g(kind, regs.nonneg[k], regs.p0[k], regs.p1[k], thr, f[h][v] + u[h][v]);
            }
#pragma unroll
            for (int h = 0; h < NBR; ++h) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    u[h][v] = f[h][v] - (zg[h][v] - u[h][v]);
                    t[h][v] = fmaf(rho, zg[h][v] - u[h][v], t[h][v]);
                }
                row_st4<VEC>(regs.dual[k], j, 16 * h + 4 * g, ok, r, u[h]);
                if (k == kpf2) upf[h] = u[h];
                if (k == kl2) ul2[h] = u[h];
            }
        }
        // ---- iteration t + 1: solve, store, statistics of the new rows (same arithmetic as k_rows_solve_stats)
        f32x4 fn[NBR];
        L.apply(t, fn);
#pragma unroll
        for (int h = 0; h < NBR; ++h) row_st4<VEC>(mv.F, j, 16 * h + 4 * g, ok, r, fn[h]);
        if (kl2 >= 0) {
#pragma unroll
            for (int h = 0; h < NBR; ++h)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    float y = fn[h][v] + ul2[h][v];
                    if (regs.nonneg[kl2]) y = fmaxf(y, 0.f);
                    if (ok) csq[h][v] += (double)y * (double)y;
                }
        }
        if (kpf2 >= 0) {
            double yt[NBR][4];
#pragma unroll
            for (int nb = 0; nb < NBR; ++nb) {
                if constexpr (R64) {  // exact fp64 sum, transposed through the wave's LDS tile: lane (q, i16) reg w = Y[q + 4w][16nb + i16]
                    double *yl = ytile + (threadIdx.x >> 6) * (16 * 17);
#pragma unroll
                    for (int v = 0; v < 4; ++v) yl[row16 * 17 + 4 * g + v] = ok ? (double)fn[nb][v] + (double)upf[nb][v] : 0.0;
#pragma unroll
                    for (int w = 0; w < 4; ++w) yt[nb][w] = yl[(g + 4 * w) * 17 + row16];
                } else {
                    f32x4 tr = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const float y = ok ? fn[nb][v] + upf[nb][v] : 0.f;
                        tr = MFMA16(y, bsel[v], tr);  // COL layout: lane (q, i16) reg w = Y[4q + w][16nb + i16]
                    }
#pragma unroll
                    for (int w = 0; w < 4; ++w) yt[nb][w] = (double)tr[w];
                }
            }
#pragma unroll
            for (int w = 0; w < 4; ++w)
#pragma unroll
                for (int a = 0; a < NBR; ++a)
#pragma unroll
                    for (int b = 0; b < NBR; ++b)
                        accS[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(yt[a][w], yt[b][w], accS[a][b], 0, 0, 0);
        }
    }
    constexpr int W = 16 * NBR;
    if (kpf2 >= 0) {  // D layout of the f64 MFMA: col = l & 15, row = (l >> 4) + 4 reg
        double *out = stat_gram + (long)tile * W * W;
#pragma unroll
        for (int a = 0; a < NBR; ++a)
#pragma unroll
            for (int b = 0; b < NBR; ++b)
#pragma unroll
                for (int v = 0; v < 4; ++v) out[(16 * a + g + 4 * v) * W + 16 * b + row16] = accS[a][b][v];
    }
    if (kl2 >= 0) {
#pragma unroll
        for (int h = 0; h < NBR; ++h)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                double sq = csq[h][v];
                sq += __shfl_xor(sq, 1);
                sq += __shfl_xor(sq, 2);
                sq += __shfl_xor(sq, 4);
                sq += __shfl_xor(sq, 8);
                const int col = 16 * h + 4 * g + v;
                if (row16 == 0 && col < r) stat_colsq[((long)tile * MCL_MAX_REGS + kl2) * r + col] = sq;
            }
    }
}

// =========================================================================================================
// host launchers
// =========================================================================================================
// fp64 row algebra for the B-mode passes of a fused stack with a PARAFAC2 member, rank <= 16 (the kernels above with
// R64 = true): decided by the plan (it owns the fp64 copies of L_i^-1 and T_i); see k_rows_solve_stats
bool mcl_rows64(const mcl_context *c) { return c->rows64 && c->LinvB64 != nullptr && c->pf2_T64 != nullptr; }

static bool rows_vec_ok(const mcl_context *c, const ModeView &mv, const RegSet &rs, const float *extra) {
    bool vec = (c->r % 4 == 0) && ((reinterpret_cast<uintptr_t>(mv.F) & 15) == 0) &&
               ((reinterpret_cast<uintptr_t>(extra) & 15) == 0);
    for (int k = 0; k < rs.n; ++k)
        vec = vec && ((reinterpret_cast<uintptr_t>(rs.aux[k]) & 15) == 0) && ((reinterpret_cast<uintptr_t>(rs.dual[k]) & 15) == 0);
    return vec;
}

#define DISPATCH_ROWS(c, vec, KERNEL, grid, block, ...)                                                       \
    do {                                                                                                      \
        const int nbr_ = (c)->r <= 16 ? 1 : ((c)->r <= 32 ? 2 : 4);                                           \
        if (vec) {                                                                                            \
            if (nbr_ == 1) hipLaunchKernelGGL((KERNEL<1, true>), grid, block, 0, (c)->stream, __VA_ARGS__);   \
            else if (nbr_ == 2) hipLaunchKernelGGL((KERNEL<2, true>), grid, block, 0, (c)->stream, __VA_ARGS__); \
            else hipLaunchKernelGGL((KERNEL<4, true>), grid, block, 0, (c)->stream, __VA_ARGS__);             \
        } else {                                                                                              \
            if (nbr_ == 1) hipLaunchKernelGGL((KERNEL<1, false>), grid, block, 0, (c)->stream, __VA_ARGS__);  \
            else if (nbr_ == 2) hipLaunchKernelGGL((KERNEL<2, false>), grid, block, 0, (c)->stream, __VA_ARGS__); \
            else hipLaunchKernelGGL((KERNEL<4, false>), grid, block, 0, (c)->stream, __VA_ARGS__);            \
        }                                                                                                     \
    } while (0)

int mcl_launch_inner_check_raw(mcl_context *c, bool begin, const double *change_part, int n_change, const double *tab, int n_rows,
                               int n_regs) {
    ProfScope prof(c, MCL_PROF_OTHER);
    hipLaunchKernelGGL(k_inner_check, dim3(1), dim3(256), 0, c->stream, begin ? 1 : 0, begin ? c->gate_active : nullptr, c->inner_gate,
                       change_part, n_change, tab, n_rows, n_regs, c->opt.inner_tol);
    MCL_CHECK_HIP(c, hipGetLastError());
    return 0;
}

int mcl_launch_inner_check(mcl_context *c, int mode, bool begin) {
    const TileMap &tm = (mode == 1) ? c->tilesB : (mode == 2 ? c->tilesC : c->tilesA);
    const double *tab = (mode == 1) ? c->diagB_tile : (mode == 2 ? c->diagC_tile : c->diagA_tile);
    const int n_change = (mode == 0) ? (int)c->I : tm.n_tiles;
    ProfScope prof(c, MCL_PROF_OTHER);
    hipLaunchKernelGGL(k_inner_check, dim3(1), dim3(256), 0, c->stream, begin ? 1 : 0, begin ? c->gate_active : nullptr, c->inner_gate,
                       (const double *)c->inner_part, n_change, tab, c->diag_rows[mode], c->regs[mode].n, c->opt.inner_tol);
    MCL_CHECK_HIP(c, hipGetLastError());
    return 0;
}

int mcl_launch_rows_solve(mcl_context *c, int mode, double *change_part) {
    ModeView mv = view_of(c, mode);
    if (mv.n_tiles == 0) return 0;
    const float *rhs = (mode == 1) ? c->XC : c->GRf + (long)c->r * c->r;  // fp32 image of R (k_C_prepare)
    const float *Arows = (mode == 1) ? c->A : nullptr;
    const float *Linv = (mode == 1) ? c->LinvB : c->LinvC;
    dim3 grid((unsigned)((mv.n_tiles + 3) / 4)), block(256);
    const bool vec = rows_vec_ok(c, mv, c->regs[mode], rhs);
    ProfScope prof(c, mode == 1 ? MCL_PROF_ROWS_CHAIN : MCL_PROF_OTHER);
    if (mode == 1) c->variant[MCL_PROF_ROWS_CHAIN] = "k_rows_solve";
    DISPATCH_ROWS(c, vec, k_rows_solve, grid, block, mv, rhs, Arows, Linv, c->regs[mode], c->r, change_part);
    MCL_CHECK_HIP(c, hipGetLastError());
    return 0;
}

int mcl_launch_A_rows_solve(mcl_context *c, double *change_part) {
    if (c->I == 0) return 0;
    hipLaunchKernelGGL(k_A_rows_solve, dim3((unsigned)c->I), dim3(64), 0, c->stream, c->rhsA, c->rhoA, c->LinvA, c->A,
                       c->regs[0], c->r, change_part);
    MCL_CHECK_HIP(c, hipGetLastError());
    return 0;
}

// The per-slab sums of the solve pass's tile statistics are taken by the PARAFAC2 Newton-Schulz kernel itself when
// the B stack has a PARAFAC2 member handled by that kernel (otherwise k_stats_reduce runs after the solve pass).
bool mcl_stats_reduce_in_algebra(const mcl_context *c) {
    if (c->sw.stats_reduce || c->sw.pf2_jacobi || c->NB > 2) return false;
    const RegSet &rs = c->regs[1];
    for (int k = 0; k < rs.n; ++k)
        if (rs.kind[k] == MCL_PEN_PARAFAC2) return true;
    return false;
}

int mcl_launch_generic_prox_local(mcl_context *c, int mode, int k) {
    ModeView mv = view_of(c, mode);
    if (mv.n_tiles == 0) return 0;
    const RegSet &rs = c->regs[mode];
    dim3 grid((unsigned)((mv.n_tiles + 3) / 4)), block(256);
    const bool vec = rows_vec_ok(c, mv, rs, nullptr);
    const int kind_k = rs.kind[k];
    ProfScope prof(c, kind_k == MCL_PEN_PARAFAC2 ? MCL_PROF_PF2 : (kind_k == MCL_PEN_UNIMODAL ? MCL_PROF_UNIMODAL : MCL_PROF_OTHER));
    switch (rs.kind[k]) {
        case MCL_PEN_NN:
        case MCL_PEN_BOX:
        case MCL_PEN_L1:
            if (c->stack_fused) break;  // no statistics; the fused finish pass does the prox
            if (mode == 0 && !c->opt.constant_A) {  // per-row feasibility penalties (the un-fused A loop of the inner stopping test)
                hipLaunchKernelGGL(k_A_rows_prox, dim3((unsigned)((c->I * c->r + 255) / 256)), dim3(256), 0, c->stream,
                                   (const float *)c->rhoA, c->A, rs, k, (int)c->I, c->r);
                break;
            }
            DISPATCH_ROWS(c, vec, k_rows_prox_rowsep, grid, block, mv, rs, k, c->r);
            break;
        case MCL_PEN_L2BALL:
            if (c->stack_fused) {  // statistics only; slot k of the colsq table
                if (c->stats_in_solve) break;  // already produced by k_rows_solve_stats + k_stats_reduce
                hipLaunchKernelGGL(k_slab_colsq, dim3((unsigned)mv.n_slabs), dim3(256), 0, c->stream, mv.ext, mv.F,
                                   rs.dual[k], rs.nonneg[k], c->r, c->RP, c->colsq + (long)k * mv.n_slabs * c->r);
                break;
            }
            hipLaunchKernelGGL(k_slab_colsq, dim3((unsigned)mv.n_slabs), dim3(256), 0, c->stream, mv.ext, mv.F,
                               rs.dual[k], rs.nonneg[k], c->r, c->RP, c->colsq);
            DISPATCH_ROWS(c, vec, k_rows_l2ball, grid, block, mv, rs, k, c->r, (const double *)c->colsq);
            break;
        case MCL_PEN_UNIMODAL: {
            if (int rc = mcl_launch_unimodal(c, mv.ext, mv.n_slabs, mv.F, rs, mode, k)) return rc;  // unimodal.hip
            if (!c->stack_fused)  // the fused finish pass updates the dual
                DISPATCH_ROWS(c, vec, k_rows_dual, grid, block, mv, rs, k, c->r);
            break;
        }
        case MCL_PEN_TV: {
            const long nthreads = (long)mv.n_slabs * c->r;
            hipLaunchKernelGGL(k_slab_tv, dim3((unsigned)((nthreads + 63) / 64)), dim3(64), 0, c->stream, mv.ext, mv.n_slabs,
                               (const float *)mv.F, mv.rho, rs, k, c->r);
            DISPATCH_ROWS(c, vec, k_rows_dual, grid, block, mv, rs, k, c->r);
            break;
        }
        case MCL_PEN_PARAFAC2: {
            if (mode != 1) {
                c->err = "PARAFAC2 constraint can only be imposed with mode=1";
                return 1;
            }
            const int r = c->r, n2 = r * r;
            if (c->stats_in_solve) {
                // S_i already produced by k_rows_solve_stats + k_stats_reduce
            } else if (c->NB == 1)
                hipLaunchKernelGGL(k_pf2_gram<1>, dim3((unsigned)c->I), dim3(256), 0, c->stream, mv.ext, mv.F, rs.dual[k], r, c->pf2_S);
            else if (c->NB == 2)
                hipLaunchKernelGGL(k_pf2_gram<2>, dim3((unsigned)c->I), dim3(256), 0, c->stream, mv.ext, mv.F, rs.dual[k], r, c->pf2_S);
            else
                hipLaunchKernelGGL(k_pf2_gram<4>, dim3((unsigned)c->I), dim3(256), 0, c->stream, mv.ext, mv.F, rs.dual[k], r, c->pf2_S);
            const size_t sm = sizeof(double) * (size_t)(4 * n2 + r);
            if (sm > 65536) {
                MCL_CHECK_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_pf2_algebra),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
            }
            int *status = nullptr;
            if (c->NB <= 2 && !c->sw.pf2_jacobi) {
                status = c->pf2_status;
                TileStats ts{c->slab_tile_ptr, c->stat_gram, c->stat_colsq, c->colsq, 16 * c->NB, (int)c->I};
                const bool tiles = c->stats_in_solve && mcl_stats_reduce_in_algebra(c);
#define MCL_NS(NB_, TILES_)                                                                                          \
    hipLaunchKernelGGL((k_pf2_algebra_ns<NB_, TILES_>), dim3((unsigned)((c->I + NsShape<NB_>::SPW - 1) / NsShape<NB_>::SPW)),                   \
                       dim3(64 * NsShape<NB_>::SPW + (((TILES_) && (NB_) == 1) ? 64 : 0)), 0, c->stream, c->pf2_S,                  \
                       rs.aux2[k], c->rhoB, mv.ext, r, c->pf2_T, c->pf2_acc, c->pf2_status, ts, rs, c->pf2_xmin,     \
                       c->sw.ns_plain ? 0 : 1, c->pf2_T64)
                c->variant[MCL_PROF_PF2] = std::string("k_pf2_algebra_ns<NB=") + std::to_string(c->NB) + (tiles ? ",TILES> (+ k_pf2_sum_delta)" : "> (+ k_pf2_sum_delta)");
                if (c->NB == 1) {
                    if (tiles) MCL_NS(1, true);
                    else MCL_NS(1, false);
                } else {
                    if (tiles) MCL_NS(2, true);
                    else MCL_NS(2, false);
                }
#undef MCL_NS
                if (c->cond_monitor)  // (mcl_condition_monitor: a trial / an occasional monitored iteration only)
                    if (int rc = mcl_launch_pf2_cond_track(c)) return rc;
            }
            if (status == nullptr) c->variant[MCL_PROF_PF2] = "k_pf2_algebra (Jacobi) + k_pf2_polar_qr";
            // rank <= 16: the Newton-Schulz kernel runs the Jacobi route itself and flags the slabs whose Gram matrix is too
            // ill-conditioned (2); the QR kernel that redoes them is launched where its launch does not count - small problems
            // (up to 64 slabs, the exact-products mode): on a large rank <= 16 problem (config 4: 1024 slabs, 5 launches of
            // ~3 us per iteration for slabs that do not occur there) such a slab keeps the Gram-route factor
            const bool ink_qr = status != nullptr && c->NB == 1 && (c->I <= 64 || c->exact);
            if (status == nullptr || c->NB != 1)
                hipLaunchKernelGGL(k_pf2_algebra, dim3((unsigned)c->I), dim3(64), sm, c->stream, c->pf2_S, rs.aux2[k],
                                   c->rhoB, r, c->pf2_T, c->pf2_acc, c->pf2_status, status != nullptr ? 1 : 0, c->pf2_T64, mv.ext);
            if (status == nullptr || c->NB != 1 || ink_qr) {
                // ... and the slabs it finds too ill-conditioned for the Gram route (flag 2) are redone from Y Delta^T itself
                const size_t smq = sizeof(double) * (size_t)(3 * n2 + 4 * 64 + 64);
                if (smq > 65536) {
                    MCL_CHECK_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_pf2_polar_qr),
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)smq));
                }
                hipLaunchKernelGGL(k_pf2_polar_qr, dim3((unsigned)c->I), dim3(256), smq, c->stream, mv.ext, mv.F, rs.dual[k],
                                   rs.aux2[k], c->rhoB, r, c->pf2_S, c->pf2_status, c->pf2_qr, c->pf2_T, c->pf2_acc,
                                   c->pf2_T64, c->gate_active);
            }
            if (!c->stack_fused)  // the fused finish pass applies T_i itself
                DISPATCH_ROWS(c, vec, k_pf2_apply, grid, block, mv, (const float *)rs.dual[k], (const float *)c->pf2_T, rs.aux[k], r);
            if (c->pf2_delta_fused)  // single-process inner loop: Delta follows at once, no all-reduce in between
                hipLaunchKernelGGL(k_pf2_sum_delta, dim3((unsigned)n2), dim3(256), 0, c->stream, c->pf2_acc, (int)c->I, n2,
                                   c->pf2_red, rs.aux2[k], c->gate_active);
            else
                hipLaunchKernelGGL(k_pf2_sum, dim3((unsigned)(n2 + 1)), dim3(256), 0, c->stream, c->pf2_acc, (int)c->I,
                                   n2 + 1, c->pf2_red);
            break;
        }
        case MCL_PEN_GL2: {
            const int n = rs.mat_rows[k];
            const double *U = rs.mat[k], *eig = U + (long)n * n, *UT = eig + n;
            const dim3 g((unsigned)((n + 63) / 64) * (unsigned)mv.n_slabs);
            hipLaunchKernelGGL((k_gl2_pass<float, false>), g, dim3(256), 0, c->stream, mv.ext, U, eig, n, c->r, mv.rho, (const float *)mv.F,
                               (const float *)rs.dual[k], c->gl2_T, (float *)nullptr, (float *)nullptr, (double *)nullptr, (double *)nullptr,
                               mv.gate);
            hipLaunchKernelGGL((k_gl2_pass<float, true>), g, dim3(256), 0, c->stream, mv.ext, UT, eig, n, c->r, mv.rho, (const float *)mv.F,
                               (const float *)rs.dual[k], c->gl2_T, rs.aux[k], rs.dual[k], (double *)nullptr, (double *)nullptr, mv.gate);
            break;
        }
        case MCL_PEN_SIMPLEX: {
            const long nw = (long)mv.n_slabs * c->r;
            hipLaunchKernelGGL((k_slab_simplex<float>), dim3((unsigned)((nw + 3) / 4)), dim3(256), 0, c->stream, mv.ext, mv.n_slabs, c->r,
                               (const float *)mv.F, (const float *)rs.dual[k], rs.aux[k], (double *)nullptr, mv.gate);
            DISPATCH_ROWS(c, vec, k_rows_dual, grid, block, mv, rs, k, c->r);
            break;
        }
        default:
            c->err = "penalty kind has no native prox (EXTERNAL penalties are evaluated by the host)";
            return 1;
    }
    MCL_CHECK_HIP(c, hipGetLastError());
    return 0;
}

// trace(F^T M F) summed over the matrices of the mode (penalties.py:737-745) -> out[0]
int mcl_launch_gl2_value(mcl_context *c, int mode, int k, double *out) {
    ModeView mv = view_of(c, mode);
    const RegSet &rs = c->regs[mode];
    const int n = rs.mat_rows[k];
    const double *U = rs.mat[k], *eig = U + (long)n * n;
    const long rows = (long)n * mv.n_slabs;
    ProfScope prof(c, MCL_PROF_DIAG);
    if (rows == 0) {
        MCL_CHECK_HIP(c, hipMemsetAsync(out, 0, sizeof(double), c->stream));
        return 0;
    }
    hipLaunchKernelGGL((k_gl2_pass<float, false>), dim3((unsigned)((n + 63) / 64) * (unsigned)mv.n_slabs), dim3(256), 0, c->stream, mv.ext,
                       U, eig, n, c->r, mv.rho, (const float *)mv.F, (const float *)nullptr, c->gl2_T, (float *)nullptr, (float *)nullptr,
                       (double *)nullptr, (double *)nullptr, (const int *)nullptr);
    hipLaunchKernelGGL(k_gl2_value, dim3(1), dim3(256), 0, c->stream, (const double *)c->gl2_T, eig, rows, n, c->r, out);
    MCL_CHECK_HIP(c, hipGetLastError());
    return 0;
}

// The wide path's (wide.hip) use of the two kernels above on its fp64 state
int mcl_launch_gl2_wide(mcl_context *c, int mode, int k, const double *F64, double *Z64, double *D64) {
    ModeView mv = view_of(c, mode);
    const RegSet &rs = c->regs[mode];
    const int n = rs.mat_rows[k];
    const double *U = rs.mat[k], *eig = U + (long)n * n, *UT = eig + n;
    const dim3 g((unsigned)((n + 63) / 64) * (unsigned)mv.n_slabs);
    hipLaunchKernelGGL((k_gl2_pass<double, false>), g, dim3(256), 0, c->stream, mv.ext, U, eig, n, c->r, mv.rho, F64, (const double *)D64,
                       c->gl2_T, (float *)nullptr, (float *)nullptr, (double *)nullptr, (double *)nullptr, mv.gate);
    hipLaunchKernelGGL((k_gl2_pass<double, true>), g, dim3(256), 0, c->stream, mv.ext, UT, eig, n, c->r, mv.rho, F64, (const double *)D64,
                       c->gl2_T, rs.aux[k], rs.dual[k], Z64, D64, mv.gate);
    MCL_CHECK_HIP(c, hipGetLastError());
    return 0;
}

int mcl_launch_simplex_wide(mcl_context *c, int mode, int k, const double *F64, const double *D64, double *Z64) {
    ModeView mv = view_of(c, mode);
    const long nw = (long)mv.n_slabs * c->r;
    hipLaunchKernelGGL((k_slab_simplex<double>), dim3((unsigned)((nw + 3) / 4)), dim3(256), 0, c->stream, mv.ext, mv.n_slabs, c->r, F64, D64,
                       c->regs[mode].aux[k], Z64, mv.gate);
    MCL_CHECK_HIP(c, hipGetLastError());
    return 0;
}

// The polar factors of the PARAFAC2 member from the fp64 state of wide.hip (S_i = Y_i^T Y_i already in pf2_S): the Jacobi route
// for every slab, then the QR route for the slabs it flags (Gram matrix too ill-conditioned) - T_i, T_i in fp64, rho_i T_i^T S_i
int mcl_launch_pf2_jacobi_wide(mcl_context *c, int k, const double *F64, const double *U64, const double *D64) {
    const RegSet &rs = c->regs[1];
    const int r = c->r, n2 = r * r;
    const size_t sm = sizeof(double) * (size_t)(4 * n2 + r);
    if (sm > 65536)
        MCL_CHECK_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_pf2_algebra), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
    ProfScope prof(c, MCL_PROF_PF2);
    c->variant[MCL_PROF_PF2] = "k_pf2_algebra (Jacobi) + k_pf2_polar_qr on the fp64 state";
    hipLaunchKernelGGL(k_pf2_algebra, dim3((unsigned)c->I), dim3(64), sm, c->stream, c->pf2_S, rs.aux2[k], c->rhoB, r, c->pf2_T,
                       c->pf2_acc, c->pf2_status, 0, c->pf2_T64, c->row_ptr_dev, D64);
    const size_t smq = sizeof(double) * (size_t)(3 * n2 + 4 * 64 + 64);
    if (smq > 65536)
        MCL_CHECK_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_pf2_polar_qr), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smq));
    hipLaunchKernelGGL(k_pf2_polar_qr, dim3((unsigned)c->I), dim3(256), smq, c->stream, c->row_ptr_dev, c->B, rs.dual[k], rs.aux2[k],
                       c->rhoB, r, c->pf2_S, c->pf2_status, c->pf2_qr, c->pf2_T, c->pf2_acc, c->pf2_T64, c->gate_active, F64, U64, D64);
    MCL_CHECK_HIP(c, hipGetLastError());
    return 0;
}

int mcl_launch_generic_prox_finish(mcl_context *c, int mode, int k) {
    const RegSet &rs = c->regs[mode];
    if (rs.kind[k] != MCL_PEN_PARAFAC2) return 0;  // dual update already fused into the local step
    ModeView mv = view_of(c, mode);
    if (mv.n_tiles == 0) return 0;
    const int n2 = c->r * c->r;
    if (!c->pf2_delta_fused)
        hipLaunchKernelGGL(k_pf2_delta, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, c->stream, c->pf2_red, c->r,
                           rs.aux2[k], c->gate_active);
    if (c->stack_fused) {  // the dual update rides in the fused finish pass
        MCL_CHECK_HIP(c, hipGetLastError());
        return 0;
    }
    dim3 grid((unsigned)((mv.n_tiles + 3) / 4)), block(256);
    const bool vec = rows_vec_ok(c, mv, rs, nullptr);
    DISPATCH_ROWS(c, vec, k_rows_pf2_dual, grid, block, mv, rs, k, c->r);
    MCL_CHECK_HIP(c, hipGetLastError());
    return 0;
}

// A penalty stack of row-separable kinds, L2 balls, unimodality and PARAFAC2 (at least one of the latter three, no
// host-evaluated member) can take ONE row pass for all prox + dual steps after the per-slab work (statistics; for
// unimodality the column regressions themselves, which write the aux rows - the pass then only updates their dual).
bool mcl_stack_can_fuse(const mcl_context *c, int mode) {
    if (c->sw.no_stack_fusion) return false;
    const RegSet &rs = c->regs[mode];
    bool slabwise = false;
    for (int k = 0; k < rs.n; ++k) {
        const int kind = rs.kind[k];
        if (kind == MCL_PEN_L2BALL || kind == MCL_PEN_PARAFAC2 || kind == MCL_PEN_UNIMODAL) slabwise = true;
        else if (kind != MCL_PEN_NN && kind != MCL_PEN_BOX && kind != MCL_PEN_L1) return false;
    }
    return slabwise && mode != 0;
}

int mcl_launch_rows_finish_fused(mcl_context *c, int mode, bool want_diag) {
    ModeView mv = view_of(c, mode);
    if (mv.n_tiles == 0) return 0;
    const RegSet &rs = c->regs[mode];
    dim3 grid((unsigned)((mv.n_tiles + 3) / 4)), block(256);
    const bool vec = rows_vec_ok(c, mv, rs, nullptr);
    double *diag = (mode == 1) ? c->diagB_tile : c->diagC_tile;
    ProfScope prof(c, mode == 1 ? MCL_PROF_ROWS_CHAIN : MCL_PROF_OTHER);
    if (mode == 1) c->variant[MCL_PROF_ROWS_CHAIN] = mcl_rows64(c) ? "k_rows_finish_solve_stats<R64> chain (solve_stats -> finish_solve_stats x (n-1) -> finish_fused)" : "k_rows_finish_solve_stats chain (solve_stats -> finish_solve_stats x (n-1) -> finish_fused)";
    if (mode == 1 && mcl_try_rows_chain_last(c, mv, vec, mcl_rows64(c), diag, want_diag ? 1 : 0)) {
        // (rowchain.hip: software-pipelined form)
        c->variant[MCL_PROF_ROWS_CHAIN] = mcl_rows64(c) ? "k_rows_chain_first<R64> -> k_rows_chain_mid x (n-1) -> k_rows_chain_last (software-pipelined chain)"
                                                         : "k_rows_chain_first -> k_rows_chain_mid x (n-1) -> k_rows_chain_last (software-pipelined chain)";
    } else if (mode == 1 && mcl_rows64(c)) {  // rank <= 16 with a PARAFAC2 member: fp64 row algebra
        if (vec)
            hipLaunchKernelGGL((k_rows_finish_fused<1, true, true>), grid, block, 0, c->stream, mv, rs, c->r, (const float *)c->pf2_T,
                               (const double *)c->colsq, diag, want_diag ? 1 : 0, (const double *)c->pf2_T64);
        else
            hipLaunchKernelGGL((k_rows_finish_fused<1, false, true>), grid, block, 0, c->stream, mv, rs, c->r, (const float *)c->pf2_T,
                               (const double *)c->colsq, diag, want_diag ? 1 : 0, (const double *)c->pf2_T64);
    } else {
        DISPATCH_ROWS(c, vec, k_rows_finish_fused, grid, block, mv, rs, c->r, (const float *)c->pf2_T, (const double *)c->colsq,
                      diag, want_diag ? 1 : 0, (const double *)nullptr);
    }
    MCL_CHECK_HIP(c, hipGetLastError());
    if (want_diag) c->diag_rows[mode] = mv.n_tiles;  // one row per tile, as k_rows_diag writes them
    return 0;
}

// mode 1 only (that is where the passes are big); needs the per-tile tables of the plan
bool mcl_stats_can_ride_in_solve(const mcl_context *c, int mode) {
    if (mode != 1 || c->sw.no_solve_stats) return false;
    const RegSet &rs = c->regs[1];
    for (int k = 0; k < rs.n; ++k) {
        if (rs.kind[k] == MCL_PEN_PARAFAC2 && !c->stat_gram) return false;
        if (rs.kind[k] == MCL_PEN_L2BALL && !c->stat_colsq) return false;
    }
    return c->NB <= 2;  // fp64 accumulators: (16 NB)^2 / 64 doubles per lane
}

// finish of inner iteration t + solve / statistics of iteration t + 1 in one pass (see k_rows_finish_solve_stats)
int mcl_launch_rows_finish_solve_stats(mcl_context *c) {
    ModeView mv = view_of(c, 1);
    if (mv.n_tiles == 0) return 0;
    const float *rhs = c->XC;
    dim3 grid((unsigned)((mv.n_tiles + 3) / 4)), block(256);
    const bool vec = rows_vec_ok(c, mv, c->regs[1], rhs);
    {
    ProfScope prof(c, MCL_PROF_ROWS_CHAIN);
#define MCL_FSS(NBR_, VEC_, R64_)                                                                                    \
    hipLaunchKernelGGL((k_rows_finish_solve_stats<NBR_, VEC_, R64_>), grid, block, 0, c->stream, mv, rhs,              \
                       (const float *)c->A, (const float *)c->LinvB, c->regs[1], c->r, (const float *)c->pf2_T,        \
                       (const double *)c->colsq, c->stat_gram, c->stat_colsq, (const double *)c->LinvB64,              \
                       (const double *)c->pf2_T64)
    if (mcl_try_rows_chain_mid(c, mv, rhs, vec, mcl_rows64(c))) {
        // (rowchain.hip: the same pass with its loads software-pipelined, for the stacks it is instantiated for)
    } else if (mcl_rows64(c)) {
        if (vec) MCL_FSS(1, true, true);
        else MCL_FSS(1, false, true);
    } else if (c->NB == 1) {
        if (vec) MCL_FSS(1, true, false);
        else MCL_FSS(1, false, false);
    } else {
        if (vec) MCL_FSS(2, true, false);
        else MCL_FSS(2, false, false);
    }
#undef MCL_FSS
    }
    ProfScope prof2(c, MCL_PROF_OTHER);
    if (!mcl_stats_reduce_in_algebra(c))
        hipLaunchKernelGGL(k_stats_reduce, dim3((unsigned)c->I), dim3(256), 0, c->stream, (const int *)c->slab_tile_ptr,
                           (const double *)c->stat_gram, (const double *)c->stat_colsq, c->regs[1], c->r, 16 * c->NB,
                           (int)c->I, c->pf2_S, c->colsq);
    MCL_CHECK_HIP(c, hipGetLastError());
    return 0;
}

int mcl_launch_rows_solve_stats(mcl_context *c) {
    ModeView mv = view_of(c, 1);
    if (mv.n_tiles == 0) return 0;
    const float *rhs = c->XC;
    dim3 grid((unsigned)((mv.n_tiles + 3) / 4)), block(256);
    const bool vec = rows_vec_ok(c, mv, c->regs[1], rhs);
    {
    ProfScope prof(c, MCL_PROF_ROWS_CHAIN);
#define MCL_SS(NBR_, VEC_, R64_)                                                                                     \
    hipLaunchKernelGGL((k_rows_solve_stats<NBR_, VEC_, R64_>), grid, block, 0, c->stream, mv, rhs, (const float *)c->A, \
                       (const float *)c->LinvB, c->regs[1], c->r, c->stat_gram, c->stat_colsq, (const double *)c->LinvB64)
    if (mcl_try_rows_chain_first(c, mv, rhs, vec, mcl_rows64(c))) {
        // (rowchain.hip: software-pipelined form)
    } else if (mcl_rows64(c)) {
        if (vec) MCL_SS(1, true, true);
        else MCL_SS(1, false, true);
    } else if (c->NB == 1) {
        if (vec) MCL_SS(1, true, false);
        else MCL_SS(1, false, false);
    } else {
        if (vec) MCL_SS(2, true, false);
        else MCL_SS(2, false, false);
    }
#undef MCL_SS
    }
    ProfScope prof2(c, MCL_PROF_OTHER);
    if (!mcl_stats_reduce_in_algebra(c))
        hipLaunchKernelGGL(k_stats_reduce, dim3((unsigned)c->I), dim3(256), 0, c->stream, (const int *)c->slab_tile_ptr,
                           (const double *)c->stat_gram, (const double *)c->stat_colsq, c->regs[1], c->r, 16 * c->NB,
                           (int)c->I, c->pf2_S, c->colsq);
    MCL_CHECK_HIP(c, hipGetLastError());
    return 0;
}
