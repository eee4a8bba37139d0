// ADMM-side kernels of the AO-ADMM engine (gfx950): rank x rank normal-equation algebra (fp64, in LDS),
// the fused inner ADMM loop for row-separable penalties, the per-slab A-phase finish, and diagnostics.
//
// Reference sites: B1/B3-B5 (decomposition.py:240-256), B6-B9 (:259-285), C2-C3 (:319-338),
// A1/A3-A6 (:138,155-213), E1-E3 (:404-417, 445-452, 916-921), prox: penalties.py:503-586.
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "mcl_internal.h"
#include "rows_mfma.h"

#define FULL_TILE 64

// ---------------------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------------------
// Gauss-Jordan inverse of an SPD r x r matrix held ONE COLUMN PER LANE in registers (fp64): lane c (< r) owns
// col[i] = M[i][c]; lanes >= r and rows >= r hold identity padding.  Pivot column entries are fetched with
// v_readlane (compile-time lane index), so there is no LDS traffic and no barrier.  SPD => no pivoting.
// Replaces the reference's SVD-based solve `(v (U/s)) Uh` (decomposition.py:172,194) - identical for SPD systems.
static __device__ __forceinline__ double readlane_f64(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

static __device__ __forceinline__ float readlane_f32(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

template <int RP>
static __device__ __forceinline__ void gj_inverse_reg(double (&col)[RP], int r, int lane) {
#pragma unroll
    for (int p = 0; p < RP; ++p) {
        if (p < r) {
            double cp[RP];
#pragma unroll
            for (int i = 0; i < RP; ++i) cp[i] = readlane_f64(col[i], p);  // M[i][p]
            // 1 / pivot by v_rcp_f64 + two Newton steps (full double precision for the positive pivots of an SPD
            // system; the IEEE division sequence is a third of the step's latency)
            double piv = __builtin_amdgcn_rcp(cp[p]);
            piv = fma(fma(-cp[p], piv, 1.0), piv, piv);
            piv = fma(fma(-cp[p], piv, 1.0), piv, piv);
            const double myp = (lane == p) ? piv : col[p] * piv;  // scaled pivot-row entry of this column
#pragma unroll
            for (int i = 0; i < RP; ++i) {
                if (i != p) col[i] = (lane == p) ? -cp[i] * piv : col[i] - cp[i] * myp;
            }
            col[p] = myp;
        }
    }
}

// The same Gauss-Jordan with the ROWS of every column split over G = 64 / RP lane groups (RP = 16: 4 groups x 4 rows),
// so that all 64 lanes work instead of RP: lane l = g * RP + c holds col[j] = M[g * RL + j][c], j < RL = RP / G.
// Per pivot p the step needs M[i][p] for the lane's own rows (from lane g * RP + p: ds_bpermute, no LDS memory), the
// pivot-row entry M[p][c] (from lane pg * RP + c) and the pivot itself (v_readlane, uniform): 2 (RL + 1) permutes and
// RL updates instead of 2 RP readlanes and RP updates - about 4x shorter for RP = 16.
template <int RP>
struct GJRows {
    static constexpr int G = (64 / RP < RP) ? 64 / RP : RP;  // lane groups
    static constexpr int RL = RP / G;                        // rows per lane
    static constexpr int LANES = RP * G;
};

static __device__ __forceinline__ double bperm_f64(int src_lane, double v) {
    const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2hiint(v));
    return __hiloint2double(hi, lo);
}

static __device__ __forceinline__ float bperm_f32(int src_lane, float v) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane << 2, __float_as_int(v)));
}

template <class F, int... Ps>
static __device__ __forceinline__ void static_for_seq(F &&f, std::integer_sequence<int, Ps...>) {
    (f(std::integral_constant<int, Ps>{}), ...);
}

template <int RP>
static __device__ __forceinline__ void gj_inverse_rows(double (&col)[GJRows<RP>::RL], int r, int lane) {
    constexpr int RL = GJRows<RP>::RL;
    const int c = lane % RP, g = lane / RP;
    if constexpr (RP == 16) {
        // a lane group is exactly one DPP row of 16 lanes: M[i][p] of the lane's own rows is lane p of its row, which
        // row_newbcast delivers as an operand modifier of a v_mov (a few cycles) - the ds_bpermute round trips through the
        // LDS crossbar (8 of the 10 per pivot) were on the dependent chain of every elimination step
        static_for_seq(
            [&](auto P) {
                constexpr int p = decltype(P)::value;
                if (p < r) {  // wave-uniform
                    constexpr int pg = p / RL, pj = p % RL;
                    const double pivot = readlane_f64(col[pj], pg * RP + p);
                    double cp[RL];
#pragma unroll
                    for (int j = 0; j < RL; ++j) cp[j] = dpp_mov_f64<0x150 + p>(col[j]);  // M[g RL + j][p]
                    const double prow = bperm_f64(pg * RP + c, col[pj]);                  // M[p][c]
                    double piv = __builtin_amdgcn_rcp(pivot);
                    piv = fma(fma(-pivot, piv, 1.0), piv, piv);
                    piv = fma(fma(-pivot, piv, 1.0), piv, piv);
                    const double myp = (c == p) ? piv : prow * piv;
                    // column p itself becomes -M[i][p] piv = fma(-M[i][p], piv, 0): one select on the addend instead of
                    // two products and a select on the result (the elimination is bound by instruction issue)
#pragma unroll
                    for (int j = 0; j < RL; ++j) {
                        const double upd = fma(-cp[j], myp, (c == p) ? 0.0 : col[j]);
                        col[j] = (j == pj && g == pg) ? myp : upd;
                    }
                }
            },
            std::make_integer_sequence<int, 16>{});
        return;
    }
#pragma unroll
    for (int p = 0; p < RP; ++p) {
        if (p < r) {  // wave-uniform
            const int pg = p / RL, pj = p % RL;
            const double pivot = readlane_f64(col[pj], pg * RP + p);
            double cp[RL];
#pragma unroll
            for (int j = 0; j < RL; ++j) cp[j] = bperm_f64(g * RP + p, col[j]);  // M[g RL + j][p]
            const double prow = bperm_f64(pg * RP + c, col[pj]);                  // M[p][c]
            double piv = __builtin_amdgcn_rcp(pivot);
            piv = fma(fma(-pivot, piv, 1.0), piv, piv);
            piv = fma(fma(-pivot, piv, 1.0), piv, piv);
            const double myp = (c == p) ? piv : prow * piv;
#pragma unroll
            for (int j = 0; j < RL; ++j) {
                const bool is_p = (g == pg) && (j == pj);
                const double upd = (c == p) ? -cp[j] * piv : col[j] - cp[j] * myp;
                col[j] = is_p ? myp : upd;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// CtC = C^T C  (fp64 accumulation, one workgroup; C staged through LDS in row chunks)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ctc(const float *__restrict__ C, int K, int r, int RP,
                                               float *__restrict__ CtC, double *__restrict__ CtC64) {
    // block a computes row a of CtC; thread (b = t % RP, ks = t / RP) strides over k
    __shared__ double sm[256];
    const int a = blockIdx.x;
    const int b = threadIdx.x % RP, ks = threadIdx.x / RP, nks = 256 / RP;
    double acc = 0.0;
    if (b < r)
        for (int k = ks; k < K; k += nks) acc += (double)C[(long)k * r + a] * (double)C[(long)k * r + b];
    sm[threadIdx.x] = acc;
    __syncthreads();
    if ((int)threadIdx.x < RP && b < r) {
        double t = 0.0;
        for (int q = 0; q < nks; ++q) t += sm[q * RP + b];
        CtC[a * r + b] = (float)t;
        CtC64[a * r + b] = t;
    }
}

// ---------------------------------------------------------------------------------------------------------
// B-phase systems: rho_i = 1/2 tr(CtC o a_i a_i^T) * scale ; L_i = CtC o a_i a_i^T + (rho_i n + l2) I ; L_i^-1
// ---------------------------------------------------------------------------------------------------------
__global__ void k_B_rho(const double *__restrict__ CtC, const float *__restrict__ A, int I, int r, float scale,
                        float *__restrict__ rhoB, float *__restrict__ rho_max) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= I) return;
    double s = 0.0;
    for (int c = 0; c < r; ++c) {
        const double a = A[(long)i * r + c];
        s += CtC[c * r + c] * a * a;
    }
    const float rho = (float)(0.5 * s * scale);
    rhoB[i] = rho;
    atomicMax(reinterpret_cast<int *>(rho_max), __float_as_int(rho));  // rho >= 0: int order == float order
}

template <int RP>
__global__ __launch_bounds__(256) void k_B_systems(const double *__restrict__ CtC, const float *__restrict__ A, int I,
                                                   int r, float scale, float l2, int n_regs, int constant,
                                                   const float *__restrict__ rho_max, float *__restrict__ rhoB,
                                                   float *__restrict__ Linv, double *__restrict__ Linv64) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);  // one wave per slab
    if (i >= I) return;
    const float *a = A + (long)i * r;
    double tr = 0.0;
    for (int c = 0; c < r; ++c) tr += CtC[c * r + c] * (double)a[c] * (double)a[c];
    float rho = (float)(0.5 * tr * scale);
    if (constant) rho = rho_max[0];
    const double shift = (double)rho * n_regs + (double)l2;
    const bool act = lane < r;
    const double ac = act ? (double)a[lane] : 0.0;
    double col[RP];
#pragma unroll
    for (int d = 0; d < RP; ++d) {
        double v = (d == lane) ? 1.0 : 0.0;
        if (act && d < r) v = CtC[d * r + lane] * (double)a[d] * ac + (d == lane ? shift : 0.0);
        col[d] = v;
    }
    gj_inverse_reg<RP>(col, r, lane);
#pragma unroll
    for (int d = 0; d < RP; ++d)
        if (act && d < r) {
            Linv[((long)i * r + d) * r + lane] = (float)col[d];
            if (Linv64 != nullptr) Linv64[((long)i * r + d) * r + lane] = col[d];
        }
    if (lane == 0) rhoB[i] = rho;
}

// Penalty-free B (decomposition.py:266-273 with an empty penalty list): B_i = ((X_i C) o a_i) L_i^-1 with un-shifted (or
// only l2-shifted) systems - the product runs in fp64 with the fp64 inverse, one wave per tile of <= 64 rows, lane c
// owning column c of L_i^-1; the row's right-hand side entries are fetched with v_readlane.
// X C in fp64 for that solve: the un-shifted systems would amplify the rounding of an fp32 contraction (1e-7 over a
// K-long chain) by their condition number - the products of fp32 values are exact in fp64, so XC64 carries only the
// final fp64 roundings.  A wave owns 16 packed rows; per 16 columns of K: lane (i = l & 15, kk = l >> 4) holds
// X[row0 + i][k0 + 4 kk .. + 3], step s of the fp64 MFMA pairs it with C[k0 + 4 kk + s][16 nb + (l & 15)]
// (D: row = (l >> 4) + 4 reg, col = l & 15).  Only used when mode 1 has no penalty: a rare, accuracy-first path.
template <int NB>
__global__ __launch_bounds__(256) void k_contract_xc_f64(const float *__restrict__ X, const float *__restrict__ C, long N,
                                                         int K, int r, double *__restrict__ XC64,
                                                         float *__restrict__ XC32 = nullptr) {
    typedef double f64x4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63;
    const long row0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16;
    if (row0 >= N) return;
    const int i = lane & 15, kk = lane >> 4;
    const long jx = min(row0 + i, N - 1);
    f64x4 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[nb] = f64x4{0.0, 0.0, 0.0, 0.0};
    for (int k0 = 0; k0 < K; k0 += 32) {  // two 16-column steps per trip: their loads are independent and issued together
        float xv[2][4], cv[2][NB][4];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int k = k0 + 16 * h + 4 * kk + s;
                const int kc = min(k, K - 1);  // unconditional loads at clamped indices, masked by a multiplication (no branch per load)
                xv[h][s] = X[jx * K + kc] * ((k < K) ? 1.f : 0.f);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const int col = 16 * nb + i;
                    cv[h][nb][s] = C[(long)kc * r + min(col, r - 1)] * ((k < K && col < r) ? 1.f : 0.f);
                }
            }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    acc[nb] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)xv[h][s], (double)cv[h][nb][s], acc[nb], 0, 0, 0);
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const long j = row0 + kk + 4 * v;
            const int col = 16 * nb + i;
            if (j < N && col < r) {
                XC64[j * r + col] = acc[nb][v];
                if (XC32 != nullptr) XC32[j * r + col] = (float)acc[nb][v];  // exact-products mode: the image the row kernels read
            }
        }
}

template <int RP>
__global__ __launch_bounds__(256) void k_B_solve_f64(const int *__restrict__ tile_slab, const int *__restrict__ tile_row0,
                                                     const int *__restrict__ tile_nrows, int n_tiles,
                                                     const double *__restrict__ XC, const float *__restrict__ A,
                                                     const double *__restrict__ Linv64, int r, float *__restrict__ B,
                                                     const int *__restrict__ gate) {
    MCL_GATE(gate);
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= n_tiles) return;
    const int slab = __builtin_amdgcn_readfirstlane(tile_slab[tile]);
    const long row0 = __builtin_amdgcn_readfirstlane(tile_row0[tile]);
    const int nrows = __builtin_amdgcn_readfirstlane(tile_nrows[tile]);
    const bool act = lane < r;
    const int c = act ? lane : 0;
    double col[RP];
#pragma unroll
    for (int d = 0; d < RP; ++d) col[d] = (act && d < r) ? Linv64[((long)slab * r + d) * r + c] : 0.0;
    const double a = act ? (double)A[(long)slab * r + c] : 0.0;
    for (int j = 0; j < nrows; ++j) {
        const double t = act ? XC[(row0 + j) * r + c] * a : 0.0;
        double acc = 0.0;
#pragma unroll
        for (int d = 0; d < RP; ++d)
            if (d < r) acc = fma(readlane_f64(t, d), col[d], acc);
        if (act) B[(row0 + j) * r + c] = (float)acc;
    }
}

// C-phase system from the (all-reduced) fp64 normal equations [G | R]: block 0 builds and inverts G + (rho n + l2) I
// (fp32 and fp64 copies of the inverse), the other blocks round R to the fp32 image the fp32 row kernels read.
template <int RP>
__global__ __launch_bounds__(64) void k_C_prepare(const double *__restrict__ GR, int r, int K, float scale, float l2,
                                                  int n_regs, float *__restrict__ rhoC, float *__restrict__ LinvC,
                                                  double *__restrict__ LinvC64, float *__restrict__ Rf) {
    const int lane = threadIdx.x;
    if (blockIdx.x > 0) {
        const long e = (long)(blockIdx.x - 1) * 64 + lane;
        if (e < (long)K * r) Rf[e] = (float)GR[(long)r * r + e];
        return;
    }
    double tr = 0.0;
    for (int c = 0; c < r; ++c) tr += GR[c * r + c];
    const float rho = (float)(0.5 * tr * scale);
    const double shift = (double)rho * n_regs + (double)l2;
    const bool act = lane < r;
    double col[RP];
#pragma unroll
    for (int d = 0; d < RP; ++d) {
        double v = (d == lane) ? 1.0 : 0.0;
        if (act && d < r) v = GR[d * r + lane] + (d == lane ? shift : 0.0);
        col[d] = v;
    }
    gj_inverse_reg<RP>(col, r, lane);
#pragma unroll
    for (int d = 0; d < RP; ++d)
        if (act && d < r) {
            LinvC[d * r + lane] = (float)col[d];
            LinvC64[d * r + lane] = col[d];
        }
    if (lane == 0) rhoC[0] = rho;
}

// Penalty-free C (decomposition.py:328-331 with an empty penalty list): C = R G^-1 is a plain least-squares solve whose
// normal equations are not shifted, so the product is taken in fp64 from the fp64 [G | R] and rounded once.
__global__ __launch_bounds__(256) void k_C_solve_f64(const double *__restrict__ R, const double *__restrict__ Linv, int K,
                                                     int r, float *__restrict__ C, const int *__restrict__ gate) {
    MCL_GATE(gate);
    extern __shared__ double Ls[];
    for (int e = threadIdx.x; e < r * r; e += 256) Ls[e] = Linv[e];
    __syncthreads();
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)K * r) return;
    const int k = (int)(e / r), c = (int)(e - (long)k * r);
    double acc = 0.0;
    for (int d = 0; d < r; ++d) acc = fma(R[(long)k * r + d], Ls[d * r + c], acc);
    C[e] = (float)acc;
}

// ---------------------------------------------------------------------------------------------------------
// Fused inner ADMM loop for row-separable penalties (NN / Box / L1), `inner` iterations in registers:
//   t   = rhs + rho * sum_k (z_k - u_k)            decomposition.py:269-273 / 328-331
//   f   = t L^-1                                   as v_mfma_f32_16x16x4_f32 on the TRANSPOSED problem
//   z_k = prox_k(f + u_k, rho) ; u_k = f - (z_k - u_k)     decomposition.py:278-285 / 337-338
// A wave owns a tile of <= 64 rows of one slab = 4 blocks of 16 rows.  Lane l = (row = l&15, g = l>>4) holds,
// for every 16-column block h, the 4 consecutive columns 16h + 4g .. +3 of its row (one 16-B load; the 64
// lanes of a wave cover 16 full rows = one contiguous 1 KB region when r = 16).
// MFMA (h', h, kq):  D[c'][row] += A[c'][kk] * B[kk][row]  with  kk = g,  k = 16h + 4g + kq:
//     A (lane (c' = l&15, g)) = L^-1[k][16h' + c']        B (lane (row = l&15, g)) = t[row][k]
//     D (lane l, reg v)       = f[row = l&15][16h' + 4g + v]
// i.e. the k-permutation is chosen so that the INPUT fragment layout equals the OUTPUT layout: the inner
// iterations need no cross-lane data movement at all.
// Per-tile diagnostics (fp64): ||f||^2, sum|f|, ||z_k - f||^2.
// ---------------------------------------------------------------------------------------------------------

// One block of 16 rows of the fused inner loop in a wave's registers: rhs (already rounded to fp32), aux and dual.
// All loads are unconditional (padding lanes re-read a valid address of the same row: their values only ever meet the
// zero entries of the L^-1 fragments and are never stored), so they leave back to back and a caller can issue them
// long before the values are needed (k_C_finish_fused: under the Gauss-Jordan of wave 0).
template <int NBR, int NREG, bool VEC>
struct RowBlock {
    static constexpr int NR = NREG > 0 ? NREG : 1;
    f32x4 rhs[NBR], z[NR][NBR], u[NR][NBR];

    static __device__ __forceinline__ f32x4 ld4(const float *__restrict__ base, long j, int col, int r) {
        if (VEC) return *reinterpret_cast<const f32x4 *>(base + j * r + (col < r ? col : 0));
        f32x4 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = base[j * r + min(col + q, r - 1)];
        return v;
    }
    // j: a VALID row for every lane (lanes past the tile's end pass the tile's first row)
    __device__ __forceinline__ void load(int lane, long j, int r, const float *__restrict__ rhs_src,
                                         const double *__restrict__ rhs64, const RegSet &regs) {
        const int g = lane >> 4;
#pragma unroll
        for (int h = 0; h < NBR; ++h) {
            const int col = 16 * h + 4 * g;
            if (rhs64 != nullptr) {  // fp64 [G | R] of the C-phase, rounded on load
                if (VEC) {
                    const double *src = rhs64 + j * r + (col < r ? col : 0);
#pragma unroll
                    for (int v = 0; v < 4; ++v) rhs[h][v] = (float)src[v];
                } else {
#pragma unroll
                    for (int v = 0; v < 4; ++v) rhs[h][v] = (float)rhs64[j * r + min(col + v, r - 1)];
                }
            } else {
                rhs[h] = ld4(rhs_src, j, col, r);
            }
#pragma unroll
            for (int k = 0; k < NREG; ++k) {
                z[k][h] = ld4(regs.aux[k], j, col, r);
                u[k][h] = ld4(regs.dual[k], j, col, r);
            }
        }
    }
};

// One wave's share of the fused inner loop: a tile of <= 64 rows [row0, row0 + nrows) of one slab.
// Li: the slab's L^-1 (global or LDS), Arow: the slab's a_i or nullptr, Fcopy: optional second destination (LDS).
// pre: the tile's first row block, loaded by the caller (or nullptr).
// Returns the tile's diagnostic sums (already reduced over the wave) in dg[0..1+NREG].
template <int NBR, int NREG, bool VEC>
static __device__ __forceinline__ void rows_fused_tile(int lane, long row0, int nrows, float rho, const float *Li,
                                                       const float *Arow, const float *__restrict__ rhs_src,
                                                       float *__restrict__ F, float *Fcopy, const RegSet &regs, int r,
                                                       int inner, double *dg, const double *__restrict__ rhs64 = nullptr,
                                                       const RowBlock<NBR, NREG, VEC> *pre = nullptr) {
    const int row16 = lane & 15, g = lane >> 4;
    constexpr int NR = NREG > 0 ? NREG : 1;

    // A-operand fragments of (L^-1)^T (clamped addresses, masked values: the reads leave together)
    float LT[NBR][NBR][4];
#pragma unroll
    for (int hp = 0; hp < NBR; ++hp)
#pragma unroll
        for (int h = 0; h < NBR; ++h)
#pragma unroll
            for (int kq = 0; kq < 4; ++kq) {
                const int k = 16 * h + 4 * g + kq, c = 16 * hp + row16;
                const float v = Li[min(k, r - 1) * r + min(c, r - 1)];
                LT[hp][h][kq] = (k < r && c < r) ? v : 0.f;
            }
    float av[NBR][4];
#pragma unroll
    for (int h = 0; h < NBR; ++h)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int col = 16 * h + 4 * g + v;
            av[h][v] = (Arow != nullptr && col < r) ? Arow[col] : 1.f;
        }
    ProxClamp prox[NR];  // branch-free prox of the row-separable penalties (penalties.py:503-586)
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        if (k < NREG) prox[k].set(regs.kind[k], regs.nonneg[k], regs.p0[k], regs.p1[k], regs.p0[k] / rho);
        else prox[k].set(0, 0, 0.f, 0.f, 0.f);
    }

    double nf = 0.0, na = 0.0, gap[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) gap[k] = 0.0;

    auto st4 = [&](float *__restrict__ base, long j, int col, bool ok, f32x4 v) {
        if (VEC) {
            if (ok && col < r) *reinterpret_cast<f32x4 *>(base + j * r + col) = v;
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (ok && col + q < r) base[j * r + col + q] = v[q];
        }
    };

    const int n_it = (NREG == 0 && inner > 1) ? 1 : inner;  // without penalties every inner solve is identical
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
        if (16 * rb >= nrows) break;  // wave-uniform
        const bool ok = 16 * rb + row16 < nrows;
        const long j = (long)row0 + 16 * rb + (ok ? row16 : 0);
        RowBlock<NBR, NREG, VEC> blk;
        if (rb == 0 && pre != nullptr)
            blk = *pre;
        else
            blk.load(lane, j, r, rhs_src, rhs64, regs);
        f32x4(&rhs)[NBR] = blk.rhs;
        f32x4(&z)[NR][NBR] = blk.z;
        f32x4(&u)[NR][NBR] = blk.u;
        f32x4 f[NBR];
#pragma unroll
        for (int h = 0; h < NBR; ++h) {
#pragma unroll
            for (int v = 0; v < 4; ++v) rhs[h][v] *= av[h][v];
            f[h] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        for (int it = 0; it < n_it; ++it) {
            f32x4 t[NBR];
#pragma unroll
            for (int h = 0; h < NBR; ++h) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    float s = 0.f;
#pragma unroll
                    for (int k = 0; k < NREG; ++k) s += z[k][h][v] - u[k][h][v];
                    t[h][v] = (NREG > 0) ? fmaf(rho, s, rhs[h][v]) : rhs[h][v];
                }
            }
#pragma unroll
            for (int hp = 0; hp < NBR; ++hp) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int h = 0; h < NBR; ++h)
#pragma unroll
                    for (int kq = 0; kq < 4; ++kq) acc = MFMA16(LT[hp][h][kq], t[h][kq], acc);
                f[hp] = acc;
            }
#pragma unroll
            for (int k = 0; k < NREG; ++k) {
#pragma unroll
                for (int h = 0; h < NBR; ++h)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const float y = f[h][v] + u[k][h][v];
                        const float zn = prox[k](y);
                        u[k][h][v] = f[h][v] - (zn - u[k][h][v]);
                        z[k][h][v] = zn;
                    }
            }
        }
#pragma unroll
        for (int h = 0; h < NBR; ++h) {
            const int col = 16 * h + 4 * g;
            st4(F, j, col, ok, f[h]);
            if (Fcopy != nullptr) st4(Fcopy, j, col, ok, f[h]);
#pragma unroll
            for (int k = 0; k < NREG; ++k) {
                st4(regs.aux[k], j, col, ok, z[k][h]);
                st4(regs.dual[k], j, col, ok, u[k][h]);
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                if (ok && col + v < r) {
                    const double fv = (double)f[h][v];
                    nf += fv * fv;
                    na += fabs(fv);
#pragma unroll
                    for (int k = 0; k < NREG; ++k) {
                        const double dlt = (double)z[k][h][v] - fv;
                        gap[k] += dlt * dlt;
                    }
                }
            }
        }
    }
    dg[0] = wave_sum(nf);
    dg[1] = wave_sum(na);
#pragma unroll
    for (int k = 0; k < NREG; ++k) dg[2 + k] = wave_sum(gap[k]);
}

template <int NBR, int NREG, bool VEC>
__global__ __launch_bounds__(256) void k_rows_fused(const int *__restrict__ tile_slab, const int *__restrict__ tile_row0,
                                                    const int *__restrict__ tile_nrows, int n_tiles,
                                                    const float *__restrict__ rhs_src, const float *__restrict__ Arows,
                                                    const float *__restrict__ rho_arr, const float *__restrict__ Linv,
                                                    float *__restrict__ F, RegSet regs, int r, int inner,
                                                    double *__restrict__ diag_tile) {
    MCL_GATE(regs.gate);
    __shared__ double dsm[4][DIAG_COLS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tile_raw = blockIdx.x * 4 + wave;
    const bool live = tile_raw < n_tiles;       // wave-uniform; dead waves do no memory traffic (nrows = 0)
    const int tile = live ? tile_raw : n_tiles - 1;
    const int slab = __builtin_amdgcn_readfirstlane(tile_slab[tile]);
    const int row0 = __builtin_amdgcn_readfirstlane(tile_row0[tile]);
    const int nrows = live ? __builtin_amdgcn_readfirstlane(tile_nrows[tile]) : 0;
    double dg[DIAG_COLS];
    rows_fused_tile<NBR, NREG, VEC>(lane, row0, nrows, rho_arr[slab], Linv + (long)slab * r * r,
                                    Arows ? Arows + (long)slab * r : nullptr, rhs_src, F, nullptr, regs, r, inner, dg);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 2 + NREG; ++k) dsm[wave][k] = dg[k];
    }
    __syncthreads();
    if (threadIdx.x < 2 + NREG)  // one diagnostics row per BLOCK (4 tiles), fixed summation order
        diag_tile[(long)blockIdx.x * DIAG_COLS + threadIdx.x] =
            (dsm[0][threadIdx.x] + dsm[1][threadIdx.x]) + (dsm[2][threadIdx.x] + dsm[3][threadIdx.x]);
}

// ---------------------------------------------------------------------------------------------------------
// Whole C-side finish in ONE workgroup (K <= 1024 rows): system from the all-reduced [G | R] (decomposition.py:319-321)
// -> fused inner loop on the rows of C (:325-338) -> CtC = C^T C (fp64) and the MFMA-fragment image of C for the
// next X C pass.  Replaces four dependent launches (k_C_prepare, k_rows_fused, k_ctc, k_build_cfrag).
// LDS: L^-1 [r*r] | C [K*r] (fp32).
// ---------------------------------------------------------------------------------------------------------
template <int NBR, int NREG, bool VEC>
__global__ __launch_bounds__(1024) void k_C_finish_fused(const double *__restrict__ GR, int K, int r, float scale, float l2,
                                                         float *__restrict__ rhoC, float *__restrict__ LinvC,
                                                         float *__restrict__ C, RegSet regs, int inner,
                                                         float *__restrict__ CtC, double *__restrict__ CtC64,
                                                         float *__restrict__ Cfrag, int KC, int NBc,
                                                         double *__restrict__ diag_row, int rows_per_wave) {
    MCL_GATE(regs.gate);
    extern __shared__ float smc[];
    __shared__ double dsm[16][DIAG_COLS];
    __shared__ float rho_s;
    __shared__ double Ls64[NREG == 0 ? 256 * NBR * NBR : 1];  // penalty-free C: the solve itself runs in fp64
    float *Ls = smc, *Cs = smc + r * r;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    constexpr int RP = 16 * NBR;
    // up to 16 waves, 16 / 32 / 64 rows each: short tiles keep the serial 5-iteration chain per wave short
    const long row0 = (long)rows_per_wave * wave;
    const int nrows = max(0, min(rows_per_wave, K - rows_per_wave * wave));
    // every wave's first row block (R rows from the fp64 [G | R], aux, dual) is requested BEFORE the system is built:
    // the round trips run under wave 0's Gauss-Jordan instead of after the barrier
    RowBlock<NBR, NREG, VEC> pre;
    if (NREG > 0 && nrows > 0)
        pre.load(lane, row0 + ((lane & 15) < nrows ? (lane & 15) : 0), r, nullptr, GR + (long)r * r, regs);
    if (wave == 0) {
        // system in the row-split layout of GJRows<RP> (lane = g * RP + c holds rows g * RL + j of column c): all loads in
        // flight at once (clamped indices, masked at use), all 64 lanes busy in the Gauss-Jordan
        constexpr int RL = GJRows<RP>::RL;
        const int cc = lane % RP, g = lane / RP;
        const bool act = cc < r;
        const int cl = act ? cc : 0;
        double gv[RL];
        double tr = 0.0;
#pragma unroll
        for (int j = 0; j < RL; ++j) {
            const int d = g * RL + j;
            gv[j] = GR[(d < r ? d : r - 1) * r + cl];
        }
#pragma unroll
        for (int j = 0; j < RL; ++j)
            if (act && g * RL + j == cc) tr = gv[j];
        tr = wave_sum(tr);
        const float rho = (float)(0.5 * tr * scale);
        const double shift = (double)rho * NREG + (double)l2;
        double col[RL];
#pragma unroll
        for (int j = 0; j < RL; ++j) {
            const int d = g * RL + j;
            double v = (d == cc) ? 1.0 : 0.0;
            if (act && d < r) v = gv[j] + (d == cc ? shift : 0.0);
            col[j] = v;
        }
        gj_inverse_rows<RP>(col, r, lane);
#pragma unroll
        for (int j = 0; j < RL; ++j) {
            const int d = g * RL + j;
            if (act && d < r) {
                Ls[d * r + cc] = (float)col[j];
                LinvC[d * r + cc] = (float)col[j];
                if (NREG == 0) Ls64[d * r + cc] = col[j];
            }
        }
        if (lane == 0) {
            rho_s = rho;
            rhoC[0] = rho;
        }
    }
    __syncthreads();
    double dg[DIAG_COLS];
    if (NREG == 0) {
        // C = R G^-1 in fp64 (un-shifted normal equations: every rounding of the product is amplified by cond(G))
        const double *R = GR + (long)r * r;
        double nf = 0.0, na = 0.0;
        if (inner > 0)
            for (int e = lane; e < nrows * r; e += 64) {
                const int rl = e / r, cidx = e - rl * r;
                const long j = row0 + rl;
                double acc = 0.0;
                for (int d = 0; d < r; ++d) acc = fma(R[j * r + d], Ls64[d * r + cidx], acc);
                const float f = (float)acc;
                C[j * r + cidx] = f;
                Cs[j * r + cidx] = f;
                nf += (double)f * (double)f;
                na += fabs((double)f);
            }
        dg[0] = wave_sum(nf);
        dg[1] = wave_sum(na);
    } else {
        rows_fused_tile<NBR, NREG, VEC>(lane, row0, nrows, rho_s, Ls, nullptr, nullptr, C, Cs, regs, r, inner, dg,
                                        GR + (long)r * r, &pre);
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 2 + NREG; ++k) dsm[wave][k] = dg[k];
    }
    __syncthreads();
    if (threadIdx.x < 2 + NREG) {
        double t = 0.0;
        for (int wv = 0; wv < n_waves; ++wv) t += dsm[wv][threadIdx.x];
        diag_row[threadIdx.x] = t;
    }
    // CtC = C^T C from the LDS copy of the new C on the fp64 MFMA (fp32 x fp32 products are exact in fp64): lane
    // (rsub = l>>4, c16 = l&15) feeds C[4g + rsub][16nb + c16]; every wave takes every n_waves-th group of 4 rows;
    // the per-wave accumulators (D layout: row = (l>>4) + 4 reg, col = l&15) are summed through LDS in fixed order.
    {
        typedef double f64x4 __attribute__((ext_vector_type(4)));
        __shared__ double csm[16 * NBR][16 * NBR];
        const int rsub = lane >> 4, c16 = lane & 15;
        f64x4 acc[NBR][NBR];
#pragma unroll
        for (int a = 0; a < NBR; ++a)
#pragma unroll
            for (int b = 0; b < NBR; ++b) acc[a][b] = f64x4{0.0, 0.0, 0.0, 0.0};
        const int n_groups = (K + 3) >> 2;
        for (int gq = wave; gq < n_groups; gq += n_waves) {
            const int row = 4 * gq + rsub;
            double y[NBR];
#pragma unroll
            for (int nb = 0; nb < NBR; ++nb) {
                const int col = 16 * nb + c16;
                y[nb] = (row < K && col < r) ? (double)Cs[row * r + col] : 0.0;
            }
#pragma unroll
            for (int a = 0; a < NBR; ++a)
#pragma unroll
                for (int b = 0; b < NBR; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(y[a], y[b], acc[a][b], 0, 0, 0);
        }
        if (NBR == 1) {
            // every wave parks its accumulators in its own LDS slot (behind the C image), then one pass sums the slots
            // in wave order: 2 barriers instead of n_waves
            double *slots = reinterpret_cast<double *>(smc + ((r * r + K * r + 1) & ~1));
#pragma unroll
            for (int v = 0; v < 4; ++v) slots[wave * 256 + (rsub + 4 * v) * 16 + c16] = acc[0][0][v];
            __syncthreads();
            for (int e = threadIdx.x; e < 256; e += blockDim.x) {
                double t = 0.0;
                for (int wv = 0; wv < n_waves; ++wv) t += slots[wv * 256 + e];  // fixed order
                csm[e >> 4][e & 15] = t;
            }
            __syncthreads();
        } else {
            for (int wv = 0; wv < n_waves; ++wv) {  // fixed summation order
                if (wave == wv) {
#pragma unroll
                    for (int a = 0; a < NBR; ++a)
#pragma unroll
                        for (int b = 0; b < NBR; ++b)
#pragma unroll
                            for (int v = 0; v < 4; ++v) {
                                double &dst = csm[16 * a + rsub + 4 * v][16 * b + c16];
                                dst = (wv == 0) ? acc[a][b][v] : dst + acc[a][b][v];
                            }
                }
                __syncthreads();
            }
        }
        for (int pr = threadIdx.x; pr < r * r; pr += blockDim.x) {
            const int a = pr / r, b = pr - a * r;
            CtC[pr] = (float)csm[a][b];
            CtC64[pr] = csm[a][b];
        }
    }
    // fragment image of C for the X C kernels (see k_build_cfrag)
    const long total = (long)KC * 4 * NBc * 256;
    for (long idx = threadIdx.x; idx < total; idx += blockDim.x) {
        const int m = idx & 3, ln = (idx >> 2) & 63;
        long t = idx >> 8;
        const int nb = t % NBc;
        t /= NBc;
        const int kq = t & 3, kc = t >> 2;
        const int k = 64 * kc + 16 * kq + 4 * (ln >> 4) + m, col = 16 * nb + (ln & 15);
        Cfrag[idx] = (k < K && col < r) ? Cs[(long)k * r + col] : 0.f;
    }
}

// ---------------------------------------------------------------------------------------------------------
// A-phase finish, one wave per slab i (decomposition.py:155-219), everything in registers:
//   lane c owns column c of Q_i = BtB_i o CtC (stored back as cross_products) and of L_i = Q_i + (rho n + l2) I;
//   L_i^-1 by the register Gauss-Jordan; fused inner loop on the row a_i with row-separable penalties
//   (a_c = sum_d t_d L^-1[d][c], t_d fetched with v_readlane); per-slab terms of the fast reconstruction-error
//   formula (:445-449) and the mode-0 diagnostics.
// The per-slab Gram and right-hand side arrive as fp64 per-segment partials (seg_btb / seg_rhs; slab_seg_ptr == nullptr:
// one entry per slab) and stay fp64 through Q_i, the system and the solve: only the stored by-products are rounded.
// fused_inner == 0: only Q_i, rho_i and L_i^-1 are produced (the generic inner loop follows).
// ---------------------------------------------------------------------------------------------------------
template <int RP>
__global__ __launch_bounds__(256) void k_A_finish(float *__restrict__ BtB, const double *__restrict__ CtC, int I, int r,
                                                  float scale, float l2, int constant,
                                                  const float *__restrict__ rho_max, float *__restrict__ rhoA,
                                                  float *__restrict__ LinvA, float *__restrict__ A, RegSet regs, int inner,
                                                  int fused_inner, double *__restrict__ e1, double *__restrict__ diag_row,
                                                  int next_B, float l2_B, int n_regs_B, float *__restrict__ rhoB,
                                                  float *__restrict__ LinvB, const int *__restrict__ slab_seg_ptr,
                                                  const double *__restrict__ seg_rhs, const double *__restrict__ seg_btb,
                                                  float *__restrict__ rhsA_out, int wide_inner, double *__restrict__ LinvA64,
                                                  double *__restrict__ rhsA64, double *__restrict__ Q64) {
    MCL_GATE(regs.gate);
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= I) return;
    const bool act = lane < r;
    const int c = act ? lane : 0;
    double qcol[RP];
    double col[RP];
    double tr = 0.0;
    int sg0 = i, sg1 = i + 1;
    if (slab_seg_ptr != nullptr) {
        sg0 = slab_seg_ptr[i];
        sg1 = slab_seg_ptr[i + 1];
    }
    // Every global load of the kernel is issued up front, unconditionally (clamped column, masked at use): the loads
    // of one stage are all in flight together instead of one dependent L2 round trip per matrix row.
    double ctc[RP], btbv[RP];
#pragma unroll
    for (int d = 0; d < RP; ++d) {
        const int dd = d < r ? d : r - 1;
        ctc[d] = CtC[dd * r + c];
        btbv[d] = 0.0;
    }
    double rhs_pre = 0.0;
    float z_pre[MCL_MAX_REGS], u_pre[MCL_MAX_REGS];
#pragma unroll
    for (int k = 0; k < MCL_MAX_REGS; ++k) {
        z_pre[k] = u_pre[k] = 0.f;
        if (k < regs.n) {
            z_pre[k] = regs.aux[k][(long)i * r + c];
            u_pre[k] = regs.dual[k][(long)i * r + c];
        }
    }
    const float a_pre = A[(long)i * r + c];
    for (int sg = sg0; sg < sg1; ++sg) {  // per-segment partial Grams / right-hand sides of this slab (fixed order)
        double v[RP];
#pragma unroll
        for (int d = 0; d < RP; ++d) v[d] = seg_btb[((long)sg * r + (d < r ? d : r - 1)) * r + c];
        const double rv = seg_rhs[(long)sg * r + c];
#pragma unroll
        for (int d = 0; d < RP; ++d) btbv[d] += v[d];
        rhs_pre += rv;
    }
#pragma unroll
    for (int d = 0; d < RP; ++d) {
        double q = 0.0;
        if (act && d < r) {
            q = btbv[d] * ctc[d];
            BtB[((long)i * r + d) * r + c] = (float)q;
        }
        qcol[d] = q;
        if (d == lane) tr = q;
    }
    tr = wave_sum(act ? tr : 0.0);
    float rho = (float)(0.5 * tr * scale);
    if (constant) rho = rho_max[1];
    const int n = regs.n;
    const double shift = (double)rho * n + (double)l2;
#pragma unroll
    for (int d = 0; d < RP; ++d) {
        double v = (d == lane) ? 1.0 : 0.0;
        if (act && d < r) v = qcol[d] + (d == lane ? shift : 0.0);
        col[d] = v;
    }
    gj_inverse_reg<RP>(col, r, lane);
    if (lane == 0) rhoA[i] = rho;
    const double rhs = act ? rhs_pre : 0.0;
    if (act) rhsA_out[(long)i * r + c] = (float)rhs;  // the `rhses` by-product
    if (!fused_inner) {
#pragma unroll
        for (int d = 0; d < RP; ++d)
            if (act && d < r) {
                LinvA[((long)i * r + d) * r + c] = (float)col[d];
                if (LinvA64 != nullptr) LinvA64[((long)i * r + d) * r + c] = col[d];
            }
        if (act && rhsA64 != nullptr) rhsA64[(long)i * r + c] = rhs;
#pragma unroll
        for (int d = 0; d < RP; ++d)
            if (act && d < r && Q64 != nullptr) Q64[((long)i * r + d) * r + c] = qcol[d];
        return;
    }
    float z[MCL_MAX_REGS], u[MCL_MAX_REGS], on[MCL_MAX_REGS];
    ProxClamp prox[MCL_MAX_REGS];
#pragma unroll
    for (int k = 0; k < MCL_MAX_REGS; ++k) {
        z[k] = u[k] = 0.f;
        if (k < n && act) {
            z[k] = z_pre[k];
            u[k] = u_pre[k];
        }
        on[k] = (k < n) ? 1.f : 0.f;  // the inner loop is branch-free: absent penalties are masked out of the sum
        if (k < n) prox[k].set(regs.kind[k], regs.nonneg[k], regs.p0[k], regs.p1[k], regs.p0[k] / rho);
        else prox[k].set(0, 0, 0.f, 0.f, 0.f);
    }
    float a = act ? a_pre : 0.f;
    const int n_it = (n == 0 && inner > 1) ? 1 : inner;
    if (wide_inner) {  // small problems (exact-products mode): the row, its auxiliary and dual variables in fp64, rounded once
        double zd[MCL_MAX_REGS], ud[MCL_MAX_REGS], ad = (double)a;
        ProxClampD pd[MCL_MAX_REGS];
#pragma unroll
        for (int k = 0; k < MCL_MAX_REGS; ++k) {
            zd[k] = (double)z[k], ud[k] = (double)u[k];
            if (k < n) pd[k].set(regs.kind[k], regs.nonneg[k], regs.p0d[k], regs.p1d[k], regs.p0d[k] / (double)rho);
            else pd[k].set(0, 0, 0.0, 0.0, 0.0);
        }
        for (int it = 0; it < n_it; ++it) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < MCL_MAX_REGS; ++k)
                if (k < n) s += zd[k] - ud[k];
            const double t = (n > 0) ? fma((double)rho, s, rhs) : rhs;
            double acc = 0.0;
#pragma unroll
            for (int d = 0; d < RP; ++d) {
                if (d < r) acc = fma(readlane_f64(t, d), col[d], acc);
            }
            ad = act ? acc : 0.0;
#pragma unroll
            for (int k = 0; k < MCL_MAX_REGS; ++k) {
                const double zn = pd[k](ad + ud[k]);
                ud[k] = ad - (zn - ud[k]);
                zd[k] = zn;
            }
        }
        a = (float)ad;
#pragma unroll
        for (int k = 0; k < MCL_MAX_REGS; ++k) z[k] = (float)zd[k], u[k] = (float)ud[k];
    } else
    for (int it = 0; it < n_it; ++it) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < MCL_MAX_REGS; ++k) s = fmaf(on[k], z[k] - u[k], s);  // = s + (z - u) exactly when on
        const double t = (n > 0) ? fma((double)rho, (double)s, rhs) : rhs;
        double acc = 0.0;
#pragma unroll
        for (int d = 0; d < RP; ++d) {
            if (d < r) acc = fma(readlane_f64(t, d), col[d], acc);
        }
        a = (float)acc;
#pragma unroll
        for (int k = 0; k < MCL_MAX_REGS; ++k) {
            const float y = a + u[k];
            const float zn = prox[k](y);
            u[k] = a - (zn - u[k]);
            z[k] = zn;
        }
    }
    if (act) {
        A[(long)i * r + c] = a;
#pragma unroll
        for (int k = 0; k < MCL_MAX_REGS; ++k) {
            if (k < n) {
                regs.aux[k][(long)i * r + c] = z[k];
                regs.dual[k][(long)i * r + c] = u[k];
            }
        }
    }
    // <X_i, M_i> = rhs_i . a_i ;  ||M_i||^2 = a_i^T Q_i a_i   (Q symmetric: column c of Q = row c)
    double qa = 0.0;
#pragma unroll
    for (int d = 0; d < RP; ++d) {
        if (d < r) {
            const float ad = readlane_f32(a, d);
            qa += qcol[d] * (double)ad;
        }
    }
    const double inner_i = wave_sum(act ? rhs * (double)a : 0.0);
    const double model_i = wave_sum(act ? (double)a * qa : 0.0);
    const double nf = wave_sum(act ? (double)a * (double)a : 0.0);
    const double na = wave_sum(act ? fabs((double)a) : 0.0);
    double gap[MCL_MAX_REGS];
#pragma unroll
    for (int k = 0; k < MCL_MAX_REGS; ++k) {
        const double dlt = (act && k < n) ? (double)z[k] - (double)a : 0.0;
        gap[k] = wave_sum(dlt * dlt);
    }
    if (lane == 0) {
        e1[2 * i] = inner_i;
        e1[2 * i + 1] = model_i;
        double *o = diag_row + (long)i * DIAG_COLS;
        o[0] = nf;
        o[1] = na;
#pragma unroll
        for (int k = 0; k < MCL_MAX_REGS; ++k) o[2 + k] = gap[k];
    }
    if (next_B) {
        // The next B-phase's systems depend only on the a_i just computed and on CtC (unchanged until the next
        // C-phase): L_i = CtC o a_i a_i^T + (rho_i n_B + l2_B) I (decomposition.py:243-256) - build and invert here.
        double trb = 0.0;
#pragma unroll
        for (int d = 0; d < RP; ++d) {
            const float ad = (d < r) ? readlane_f32(a, d) : 0.f;
            double v = (d == lane) ? 1.0 : 0.0;
            if (act && d < r) v = ctc[d] * (double)ad * (double)a;
            if (d == lane && act) trb = v;
            col[d] = v;
        }
        trb = wave_sum(act ? trb : 0.0);
        const float rb = (float)(0.5 * trb * scale);
        const double shiftb = (double)rb * n_regs_B + (double)l2_B;
        if (act) {
#pragma unroll
            for (int d = 0; d < RP; ++d)
                if (d == lane) col[d] += shiftb;
        }
        gj_inverse_reg<RP>(col, r, lane);
#pragma unroll
        for (int d = 0; d < RP; ++d)
            if (act && d < r) LinvB[((long)i * r + d) * r + c] = (float)col[d];
        if (lane == 0) rhoB[i] = rb;
    }
}

// ---------------------------------------------------------------------------------------------------------
// k_A_finish with the rows of every system split over the lane groups of GJRows<RP> (all 64 lanes busy; the Gauss-Jordan
// sweeps, the matrix-vector products of the inner loop and of the error terms are ~4x shorter for rank 9..16).
// Lane l = g * RP + c: column c, rows d = g * RL + j.  Per-column quantities (rhs, a, aux, dual) are replicated in
// every group; group 0 stores them.  Same arithmetic as k_A_finish up to the association of the fp64 sums over d.
// ---------------------------------------------------------------------------------------------------------
// sum of a value that is non-zero only in lanes [0, RP), RP <= 32: butterfly inside the 16-lane DPP row (quad_perm xor 1,
// xor 2, row_half_mirror, row_mirror), one cross-row step for RP = 32; every lane of the row(s) ends with the total
template <int CTRL>
static __device__ __forceinline__ double dpp_mov_d(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int RP>
static __device__ __forceinline__ double lead_sum(double v) {
    static_assert(RP <= 32, "lead lanes must fit two DPP rows");
    v += dpp_mov_d<0xB1>(v);
    v += dpp_mov_d<0x4E>(v);
    v += dpp_mov_d<0x141>(v);
    v += dpp_mov_d<0x140>(v);
    if (RP > 16) v += __shfl_xor(v, 16);
    return v;
}

// What the A-phase finish absorbs from its predecessor on the sweep path: Mpart != nullptr - the sweep left
// M_bseg = X_bseg^T B_bseg per bseg (C-fragment order) and the wave forms rhs_i = sum_bsegs coldot(M_bseg, C) itself
// (decomposition.py:147-152): one launch (k_A_rhs_from_M) and its boundary less per outer iteration.
// (Measured and NOT kept: letting the last workgroup to finish also reduce the diagnostics tables - the 256-way ticket
// fan-in on one counter plus the epilogue cost 9 us, the k_diag_final launch it replaced 4.8 us.)
struct AFuse {
    const float *Mpart, *Cfrag;
    int MS, NBm;
    // the C-phase finish over several workgroups (k_C_finish_multi) leaves C^T C as ctc_parts partial 16 x 16 fp64 blocks:
    // the finish sums them (fixed order) and slab 0 writes the totals out for everybody else
    int ctc_parts;
    const double *CtCpart;
    double *CtC64_out;
    float *CtC_out;
    double *LinvB64;  // fp64 copy of the next B-phase's inverses (fp64 row passes of PARAFAC2 stacks: mcl_rows64), or NULL
    double *LinvA64, *rhsA64, *Q64;  // fp64 systems / right-hand sides / cross products of a host- or wide-driven inner loop (fused_inner = 0), or NULL
    int wide_inner;            // small problems (exact-products mode): the inner loop keeps the row and its ADMM variables in fp64
};

// sum_k M[k][c] C[k][c] over the bsegs sgA, sgA + step, ... < sgB, both operands in C-fragment order (element
// ((chunk * NB + nb) * 64 + l) * 4 + m <-> k = 16 chunk + 4 (l >> 4) + m, column 16 nb + (l & 15)): the wave streams
// M_bseg with 1 KB loads, fp64 accumulation (products of fp32 values are exact), the four lane quarters of a column are
// summed by two butterfly steps: every lane ends with the sums of columns (l & 15) and 16 + (l & 15) in acc0 / acc1.
template <int NQ>  // chunks of 256 floats per trip: 16 (partial images of K >= 256: MS is a multiple of 4096) or 8 (K <= 128: MS = 2048)
static __device__ __forceinline__ void m_coldot_q(const AFuse &F, int sgA, int sgB, int step, int lane, double (&pa)[2][2]) {
    for (int sg = sgA; sg < sgB; sg += step) {
        const float *mp = F.Mpart + (long)sg * F.MS;
        for (int e0 = 0; e0 < F.MS; e0 += 256 * NQ) {  // 2 NQ loads in flight
            f32x4 mv[NQ], cv[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                mv[q] = *reinterpret_cast<const f32x4 *>(mp + e0 + 256 * q + 4 * lane);
                cv[q] = *reinterpret_cast<const f32x4 *>(F.Cfrag + e0 + 256 * q + 4 * lane);
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                double d = (double)mv[q][0] * (double)cv[q][0];
                d = fma((double)mv[q][1], (double)cv[q][1], d);
                d = fma((double)mv[q][2], (double)cv[q][2], d);
                d = fma((double)mv[q][3], (double)cv[q][3], d);
                // chunk index = e >> 8 = NQ * trip + q: nb = chunk % NB
                if (F.NBm == 2 && (q & 1)) pa[1][(q >> 1) & 1] += d;
                else pa[0][(q >> (F.NBm == 2 ? 1 : 0)) & 1] += d;
            }
        }
    }
}

static __device__ __forceinline__ void m_coldot(const AFuse &F, int sgA, int sgB, int step, int lane, double &acc0,
                                                double &acc1) {
    double pa[2][2] = {{0.0, 0.0}, {0.0, 0.0}};  // [nb][chain]: two chains per column block shorten the FMA dependency
    if ((F.MS & 4095) == 0) m_coldot_q<16>(F, sgA, sgB, step, lane, pa);
    else m_coldot_q<8>(F, sgA, sgB, step, lane, pa);
    acc0 = pa[0][0] + pa[0][1], acc1 = pa[1][0] + pa[1][1];
    acc0 += __shfl_xor(acc0, 16), acc1 += __shfl_xor(acc1, 16);
    acc0 += __shfl_xor(acc0, 32), acc1 += __shfl_xor(acc1, 32);
}

// one wave = one slab
template <int RP>
static __device__ __forceinline__ void a_finish_rows_slab(const int i, const int lane, float *__restrict__ BtB,
                                                          const double *__restrict__ CtC, int r, float scale, float l2,
                                                          int constant, const float *__restrict__ rho_max,
                                                          float *__restrict__ rhoA, float *__restrict__ LinvA,
                                                          float *__restrict__ A, const RegSet &regs, int inner,
                                                          int fused_inner, double *__restrict__ e1,
                                                          double *__restrict__ diag_row, int next_B, float l2_B,
                                                          int n_regs_B, float *__restrict__ rhoB,
                                                          float *__restrict__ LinvB, const int *__restrict__ slab_seg_ptr,
                                                          const double *__restrict__ seg_rhs,
                                                          const double *__restrict__ seg_btb,
                                                          float *__restrict__ rhsA_out, const AFuse &F,
                                                          const double *rhs_lds = nullptr) {
    constexpr int RL = GJRows<RP>::RL, G = GJRows<RP>::G;
    const int cc = lane % RP, g = lane / RP;
    const bool in_range = lane < GJRows<RP>::LANES;  // RP = 4 uses 16 lanes only
    const bool act = in_range && cc < r;
    const bool lead = act && g == 0;                 // the lanes that own the per-column results
    const int c = act ? cc : 0;
    int drow[RL];
    bool dok[RL];
#pragma unroll
    for (int j = 0; j < RL; ++j) {
        const int d = (in_range ? g : 0) * RL + j;
        dok[j] = act && d < r;
        drow[j] = d < r ? d : r - 1;  // clamped: every load is unconditional
    }
    int sg0 = i, sg1 = i + 1;
    if (slab_seg_ptr != nullptr) {
        sg0 = slab_seg_ptr[i];
        sg1 = slab_seg_ptr[i + 1];
    }
    // all global loads up front
    double ctc[RL], btbv[RL];
    if (F.ctc_parts > 0) {
        for (int pb = 0; pb < F.ctc_parts; pb += 4) {  // four partial blocks per trip: independent clamped loads, added in order
            double v[4][RL];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < RL; ++j) v[u][j] = F.CtCpart[(long)min(pb + u, F.ctc_parts - 1) * 256 + drow[j] * 16 + c];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (pb + u < F.ctc_parts) {
#pragma unroll
                    for (int j = 0; j < RL; ++j) ctc[j] = (pb + u == 0) ? v[u][j] : ctc[j] + v[u][j];
                }
        }
        if (i == 0) {
#pragma unroll
            for (int j = 0; j < RL; ++j)
                if (dok[j]) {
                    F.CtC64_out[drow[j] * r + c] = ctc[j];
                    F.CtC_out[drow[j] * r + c] = (float)ctc[j];
                }
        }
#pragma unroll
        for (int j = 0; j < RL; ++j) btbv[j] = 0.0;
    } else {
#pragma unroll
        for (int j = 0; j < RL; ++j) {
            ctc[j] = CtC[drow[j] * r + c];
            btbv[j] = 0.0;
        }
    }
    double rhs_pre = 0.0;
    float z[MCL_MAX_REGS], u[MCL_MAX_REGS], on[MCL_MAX_REGS];
    ProxClamp prox[MCL_MAX_REGS];
    const int n = regs.n;
#pragma unroll
    for (int k = 0; k < MCL_MAX_REGS; ++k) {
        z[k] = u[k] = 0.f;
        if (k < n) {
            z[k] = regs.aux[k][(long)i * r + c];
            u[k] = regs.dual[k][(long)i * r + c];
        }
    }
    float a = A[(long)i * r + c];
    {  // fp64 per-segment partial Grams / right-hand sides of this slab, fixed order
        constexpr int SGB = 4;  // segments fetched per batch: independent clamped loads, summed in segment order
        for (int sb = sg0; sb < sg1; sb += SGB) {
            double v[SGB][RL], rv[SGB];
#pragma unroll
            for (int q = 0; q < SGB; ++q) {
                const long sg = min(sb + q, sg1 - 1);
#pragma unroll
                for (int j = 0; j < RL; ++j) v[q][j] = seg_btb[(sg * r + drow[j]) * r + c];
                rv[q] = (F.Mpart == nullptr) ? seg_rhs[sg * r + c] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < SGB; ++q)
                if (sb + q < sg1) {
#pragma unroll
                    for (int j = 0; j < RL; ++j) btbv[j] += v[q][j];
                    rhs_pre += rv[q];
                }
        }
    }
    if (rhs_lds != nullptr) {
        // k_A_finish_rows_wide: the slab's other wave(s) stream its partials of M while this wave builds and inverts the
        // system; their column sums are picked up behind the workgroup barrier after the Gauss-Jordan
    } else if (F.Mpart != nullptr) {
        // rhs_i[c] = sum_k M_i[k][c] C[k][c] over the slab's bsegs; the column's sum is fetched into the lanes that own it
        double acc0, acc1;
        m_coldot(F, sg0, sg1, 1, lane, acc0, acc1);
        const double v0 = bperm_f64(cc & 15, acc0), v1 = bperm_f64(cc & 15, acc1);
        rhs_pre = (cc < 16) ? v0 : v1;
    }
    double qf[RL];
    double tr = 0.0;
#pragma unroll
    for (int j = 0; j < RL; ++j) {
        double q = 0.0;
        if (dok[j]) {
            q = btbv[j] * ctc[j];
            BtB[((long)i * r + drow[j]) * r + c] = (float)q;
            if (drow[j] == c) tr = q;
        }
        qf[j] = q;
    }
    tr = wave_sum(tr);
    float rho = (float)(0.5 * tr * scale);
    if (constant) rho = rho_max[1];
    const double shift = (double)rho * n + (double)l2;
    double col[RL];
#pragma unroll
    for (int j = 0; j < RL; ++j) {
        const int d = g * RL + j;
        double v = (d == cc) ? 1.0 : 0.0;  // identity padding (also for the idle lanes of RP = 4)
        if (dok[j]) v = qf[j] + (d == cc ? shift : 0.0);
        col[j] = v;
    }
    gj_inverse_rows<RP>(col, r, in_range ? lane : cc);
    if (lane == 0) rhoA[i] = rho;
    if (rhs_lds != nullptr) {
        __syncthreads();
        rhs_pre = (rhs_lds[cc] + rhs_lds[32 + cc]) + rhs_lds[64 + cc];
    }
    const double rhs = act ? rhs_pre : 0.0;
    if (lead) rhsA_out[(long)i * r + c] = (float)rhs;  // the `rhses` by-product
    if (!fused_inner) {
#pragma unroll
        for (int j = 0; j < RL; ++j)
            if (dok[j]) {
                LinvA[((long)i * r + drow[j]) * r + c] = (float)col[j];
                if (F.LinvA64 != nullptr) F.LinvA64[((long)i * r + drow[j]) * r + c] = col[j];
            }
        if (lead && F.rhsA64 != nullptr) F.rhsA64[(long)i * r + c] = rhs;
#pragma unroll
        for (int j = 0; j < RL; ++j)
            if (dok[j] && F.Q64 != nullptr) F.Q64[((long)i * r + drow[j]) * r + c] = qf[j];
        return;
    }
#pragma unroll
    for (int k = 0; k < MCL_MAX_REGS; ++k) {
        if (!(k < n && act)) z[k] = u[k] = 0.f;
        on[k] = (k < n) ? 1.f : 0.f;
        if (k < n) prox[k].set(regs.kind[k], regs.nonneg[k], regs.p0[k], regs.p1[k], regs.p0[k] / rho);
        else prox[k].set(0, 0, 0.f, 0.f, 0.f);
    }
    if (!act) a = 0.f;
    // sum over the G row groups of one column: lanes c, RP + c, ... (fixed association)
    auto group_sum = [&](double v) -> double {
#pragma unroll
        for (int o = RP; o < 64 && o < RP * G; o <<= 1) v += __shfl_xor(v, o);
        return v;
    };
    const int n_it = (n == 0 && inner > 1) ? 1 : inner;
    if (F.wide_inner) {  // (see k_A_finish)
        double zd[MCL_MAX_REGS], ud[MCL_MAX_REGS], ad = (double)a;
        ProxClampD pd[MCL_MAX_REGS];
#pragma unroll
        for (int k = 0; k < MCL_MAX_REGS; ++k) {
            zd[k] = (double)z[k], ud[k] = (double)u[k];
            if (k < n) pd[k].set(regs.kind[k], regs.nonneg[k], regs.p0d[k], regs.p1d[k], regs.p0d[k] / (double)rho);
            else pd[k].set(0, 0, 0.0, 0.0, 0.0);
        }
        for (int it = 0; it < n_it; ++it) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < MCL_MAX_REGS; ++k)
                if (k < n) s += zd[k] - ud[k];
            const double t = (n > 0) ? fma((double)rho, s, rhs) : rhs;
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < RL; ++j) {
                const double td = bperm_f64(g * RP + g * RL + j, t);
                if (g * RL + j < r) acc = fma(td, col[j], acc);
            }
            ad = group_sum(acc);  // (every lane takes part in the shuffles)
            if (!act) ad = 0.0;
#pragma unroll
            for (int k = 0; k < MCL_MAX_REGS; ++k) {
                const double zn = pd[k](ad + ud[k]);
                ud[k] = ad - (zn - ud[k]);
                zd[k] = zn;
            }
        }
        a = (float)ad;
#pragma unroll
        for (int k = 0; k < MCL_MAX_REGS; ++k) z[k] = (float)zd[k], u[k] = (float)ud[k];
    } else
    for (int it = 0; it < n_it; ++it) {
        float sacc = 0.f;
#pragma unroll
        for (int k = 0; k < MCL_MAX_REGS; ++k) sacc = fmaf(on[k], z[k] - u[k], sacc);  // = sacc + (z - u) exactly when on
        const double t = (n > 0) ? fma((double)rho, (double)sacc, rhs) : rhs;
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < RL; ++j) {
            const double td = bperm_f64(g * RP + g * RL + j, t);  // t_d, d = g RL + j (column d of my own group)
            if (g * RL + j < r) acc = fma(td, col[j], acc);
        }
        a = (float)group_sum(acc);
#pragma unroll
        for (int k = 0; k < MCL_MAX_REGS; ++k) {
            const float y = a + u[k];
            const float zn = prox[k](y);
            u[k] = a - (zn - u[k]);
            z[k] = zn;
        }
    }
    if (!act) a = 0.f;
    if (lead) {
        A[(long)i * r + c] = a;
#pragma unroll
        for (int k = 0; k < MCL_MAX_REGS; ++k) {
            if (k < n) {
                regs.aux[k][(long)i * r + c] = z[k];
                regs.dual[k][(long)i * r + c] = u[k];
            }
        }
    }
    // <X_i, M_i> = rhs_i . a_i ;  ||M_i||^2 = a_i^T Q_i a_i
    float ad[RL];
    double qa = 0.0;
#pragma unroll
    for (int j = 0; j < RL; ++j) {
        ad[j] = bperm_f32(g * RP + g * RL + j, a);  // a_d
        if (g * RL + j < r) qa += qf[j] * (double)ad[j];
    }
    qa = group_sum(qa);
    // the contributions live in the lead lanes [0, RP): DPP row reductions (no LDS round trips), result in lane 0
    const double inner_i = lead_sum<RP>(lead ? rhs * (double)a : 0.0);
    const double model_i = lead_sum<RP>(lead ? (double)a * qa : 0.0);
    const double nf = lead_sum<RP>(lead ? (double)a * (double)a : 0.0);
    const double na = lead_sum<RP>(lead ? fabs((double)a) : 0.0);
    double gap[MCL_MAX_REGS];
#pragma unroll
    for (int k = 0; k < MCL_MAX_REGS; ++k) {
        const double dlt = (lead && k < n) ? (double)z[k] - (double)a : 0.0;
        gap[k] = lead_sum<RP>(dlt * dlt);
    }
    if (lane == 0) {
        e1[2 * i] = inner_i;
        e1[2 * i + 1] = model_i;
        double *o = diag_row + (long)i * DIAG_COLS;
        o[0] = nf;
        o[1] = na;
#pragma unroll
        for (int k = 0; k < MCL_MAX_REGS; ++k) o[2 + k] = gap[k];
    }
    if (next_B) {
        // systems of the next B-phase: L_i = CtC o a_i a_i^T + (rho_i n_B + l2_B) I (decomposition.py:243-256)
        double trb = 0.0;
#pragma unroll
        for (int j = 0; j < RL; ++j) {
            const int d = g * RL + j;
            double v = (d == cc) ? 1.0 : 0.0;
            if (dok[j]) v = ctc[j] * (double)ad[j] * (double)a;
            if (dok[j] && d == cc) trb = v;
            col[j] = v;
        }
        trb = wave_sum(trb);
        const float rb = (float)(0.5 * trb * scale);
        const double shiftb = (double)rb * n_regs_B + (double)l2_B;
#pragma unroll
        for (int j = 0; j < RL; ++j)
            if (dok[j] && g * RL + j == cc) col[j] += shiftb;
        gj_inverse_rows<RP>(col, r, in_range ? lane : cc);
#pragma unroll
        for (int j = 0; j < RL; ++j)
            if (dok[j]) {
                LinvB[((long)i * r + drow[j]) * r + c] = (float)col[j];
                if (F.LinvB64 != nullptr) F.LinvB64[((long)i * r + drow[j]) * r + c] = col[j];
            }
        if (lane == 0) rhoB[i] = rb;
    }
}

template <int RP>
__global__ __launch_bounds__(256) void k_A_finish_rows(float *__restrict__ BtB, const double *__restrict__ CtC, int I,
                                                       int r, float scale, float l2, int constant,
                                                       const float *__restrict__ rho_max, float *__restrict__ rhoA,
                                                       float *__restrict__ LinvA, float *__restrict__ A, RegSet regs,
                                                       int inner, int fused_inner, double *__restrict__ e1,
                                                       double *__restrict__ diag_row, int next_B, float l2_B,
                                                       int n_regs_B, float *__restrict__ rhoB, float *__restrict__ LinvB,
                                                       const int *__restrict__ slab_seg_ptr,
                                                       const double *__restrict__ seg_rhs,
                                                       const double *__restrict__ seg_btb, float *__restrict__ rhsA_out,
                                                       AFuse F) {
    MCL_GATE(regs.gate);
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= I) return;
    a_finish_rows_slab<RP>(i, lane, BtB, CtC, r, scale, l2, constant, rho_max, rhoA, LinvA, A, regs, inner, fused_inner, e1,
                           diag_row, next_B, l2_B, n_regs_B, rhoB, LinvB, slab_seg_ptr, seg_rhs, seg_btb, rhsA_out, F);
}

// One workgroup per slab: waves 1..3 stream the slab's partials of M (rhs_i = sum over partials of coldot(M_part, C)) WHILE
// wave 0 loads, builds and inverts the slab's system; one workgroup barrier behind the Gauss-Jordan hands the column sums
// over, then wave 0 finishes the slab as in k_A_finish_rows.  Used whenever the finish forms rhs_i itself (1..8 partials
// per slab): the M stream (16 KB per partial at K = 256, two 32-load trips) leaves the critical path of the iteration.
// SPB = 2 (more than 512 slabs, one partial per slab - config 3): two slabs per workgroup, one system wave and one streaming
// wave each, so that 1024 slabs still fit the device in one round (two workgroups per CU).
template <int RP, int SPB>
__global__ __launch_bounds__(256) void k_A_finish_rows_wide(float *__restrict__ BtB, const double *__restrict__ CtC, int I,
                                                            int r, float scale, float l2, int constant,
                                                            const float *__restrict__ rho_max, float *__restrict__ rhoA,
                                                            float *__restrict__ LinvA, float *__restrict__ A, RegSet regs,
                                                            int inner, int fused_inner, double *__restrict__ e1,
                                                            double *__restrict__ diag_row, int next_B, float l2_B,
                                                            int n_regs_B, float *__restrict__ rhoB,
                                                            float *__restrict__ LinvB,
                                                            const int *__restrict__ slab_seg_ptr,
                                                            const double *__restrict__ seg_rhs,
                                                            const double *__restrict__ seg_btb,
                                                            float *__restrict__ rhsA_out, AFuse F) {
    MCL_GATE(regs.gate);
    __shared__ double part[SPB][3][32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int WPS = 4 / SPB;                  // waves per slab: one system wave + WPS - 1 streaming waves
    const int sl = wave / WPS, role = wave % WPS;  // role 0: the system wave
    const int i = blockIdx.x * SPB + sl;
    if (role != 0) {
        double a0 = 0.0, a1 = 0.0;
        if (i < I) m_coldot(F, slab_seg_ptr[i] + role - 1, slab_seg_ptr[i + 1], WPS - 1, lane, a0, a1);
        if (lane < 16) {
            part[sl][role - 1][lane] = a0, part[sl][role - 1][16 + lane] = a1;
            if (SPB == 2) part[sl][1][lane] = part[sl][1][16 + lane] = part[sl][2][lane] = part[sl][2][16 + lane] = 0.0;
        }
        __syncthreads();  // pairs with the barrier of the system wave behind its Gauss-Jordan (a_finish_rows_slab)
        return;
    }
    if (i >= I) {  // the odd slab out of a two-slab workgroup
        __syncthreads();
        return;
    }
    a_finish_rows_slab<RP>(i, lane, BtB, CtC, r, scale, l2, constant, rho_max, rhoA, LinvA, A, regs, inner, fused_inner, e1,
                           diag_row, next_B, l2_B, n_regs_B, rhoB, LinvB, slab_seg_ptr, seg_rhs, seg_btb, rhsA_out, F,
                           &part[sl][0][0]);
}

// rho_i of the A-phase alone (needed before the systems when the feasibility penalty is constant)
__global__ void k_A_rho(const double *__restrict__ BtB, const double *__restrict__ CtC, int I, int r, float scale,
                        float *__restrict__ rhoA, float *__restrict__ rho_max) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= I) return;
    double s = 0.0;
    for (int c = 0; c < r; ++c) s += BtB[((long)i * r + c) * r + c] * CtC[c * r + c];
    const float rho = (float)(0.5 * s * scale);
    rhoA[i] = rho;
    atomicMax(reinterpret_cast<int *>(rho_max + 1), __float_as_int(rho));
}

// e1 / row diagnostics of mode 0 from (rhsA, BtB or Q, A) - used when the A-phase inner loop was not fused
// or when the by-products are recomputed for an error evaluation without an A update (decomposition.py:430-444)
__global__ __launch_bounds__(64) void k_A_e1(const float *__restrict__ rhsA, const float *__restrict__ BtB,
                                             const float *__restrict__ CtC, int btb_is_q, const float *__restrict__ A,
                                             RegSet regs, int r, double *__restrict__ e1,
                                             double *__restrict__ diag_row, const double *__restrict__ rhs64,
                                             const double *__restrict__ Q64) {
    // rhs64 / Q64: the fp64 right-hand sides / cross products the systems were built from (exact-products mode): the fast
    // error formula ||X||^2 - 2 <X, M> + ||M||^2 cancels to rec^2, so the fp32 rounding of rhs_i and Q_i (6e-8) would reach
    // the reconstruction error multiplied by ||X||^2 / rec^2
    MCL_GATE(regs.gate);
    const int i = blockIdx.x, lane = threadIdx.x;
    const bool act = lane < r;
    const int c = act ? lane : 0;
    const float a = A[(long)i * r + c];
    double qa = 0.0;
    for (int d = 0; d < r; ++d) {
        double q = Q64 != nullptr ? Q64[((long)i * r + c) * r + d] : (double)BtB[((long)i * r + c) * r + d];
        if (!btb_is_q) q *= (double)CtC[c * r + d];
        qa += q * (double)A[(long)i * r + d];
    }
    const double rhs_c = rhs64 != nullptr ? rhs64[(long)i * r + c] : (double)rhsA[(long)i * r + c];
    const double inner_i = wave_sum(act ? rhs_c * (double)a : 0.0);
    const double model_i = wave_sum(act ? (double)a * qa : 0.0);
    const double nf = wave_sum(act ? (double)a * (double)a : 0.0);
    const double na = wave_sum(act ? fabs((double)a) : 0.0);
    double gap[MCL_MAX_REGS];
#pragma unroll
    for (int k = 0; k < MCL_MAX_REGS; ++k) {
        double dlt = 0.0;
        if (act && k < regs.n) dlt = (double)regs.aux[k][(long)i * r + c] - (double)a;
        gap[k] = wave_sum(dlt * dlt);
    }
    if (lane == 0) {
        e1[2 * i] = inner_i;
        e1[2 * i + 1] = model_i;
        double *o = diag_row + (long)i * DIAG_COLS;
        o[0] = nf;
        o[1] = na;
#pragma unroll
        for (int k = 0; k < MCL_MAX_REGS; ++k) o[2 + k] = gap[k];
    }
}

// ---------------------------------------------------------------------------------------------------------
// diagnostics: ||X||^2 (once) and ONE launch that reduces every per-tile / per-slab table and assembles the
// MCL_DIAG_LEN vector.  Block b < 3*DIAG_COLS reduces column (b % DIAG_COLS) of table (b / DIAG_COLS) in
// {A rows, B tiles, C tiles}; the next two blocks reduce the e1 columns; the last block writes the constants.
// Every output entry is written by exactly one block (fixed summation order: deterministic).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sumsq_partial(const float *__restrict__ x, long n, double *__restrict__ part) {
    __shared__ double sm[4];
    double s = 0.0;
    const long stride = (long)gridDim.x * 256 * 4;
    for (long e = ((long)blockIdx.x * 256 + threadIdx.x) * 4; e < n; e += stride) {
        if (e + 3 < n) {
            const float4 v = *reinterpret_cast<const float4 *>(x + e);
            s += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
        } else {
            for (long t = e; t < n; ++t) s += (double)x[t] * (double)x[t];
        }
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

__global__ __launch_bounds__(256) void k_sum_doubles(const double *__restrict__ part, int n, double *__restrict__ out) {
    __shared__ double sm[4];
    double s = 0.0;
    for (int e = threadIdx.x; e < n; e += 256) s += part[e];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}


// column sum of a [n_rows, ncols] fp64 table by one workgroup of 256 or 1024 threads (fixed order for a given block size: the
// launchers pick the size from the table lengths alone, so it is the same on every rank and in every iteration)
static __device__ double block_colsum(const double *__restrict__ tab, int n_rows, int ncols, int col, double *sm) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    const int nt = blockDim.x;
    int e = threadIdx.x;
    for (; e + 3 * nt < n_rows; e += 4 * nt) {
        s0 += tab[(long)e * ncols + col];
        s1 += tab[(long)(e + nt) * ncols + col];
        s2 += tab[(long)(e + 2 * nt) * ncols + col];
        s3 += tab[(long)(e + 3 * nt) * ncols + col];
    }
    for (; e < n_rows; e += nt) s0 += tab[(long)e * ncols + col];
    double s = wave_sum((s0 + s1) + (s2 + s3));
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < (nt >> 6); w += 4) t += (sm[w] + sm[w + 1]) + (sm[w + 2] + sm[w + 3]);
    return t;
}

__global__ __launch_bounds__(1024) void k_diag_final(DiagTables T, int include_replicated, double *__restrict__ out) {
    __shared__ double sm[16];
    const int b = blockIdx.x;
    if (b < 3 * DIAG_COLS) {
        const int t = b / DIAG_COLS, col = b - t * DIAG_COLS;
        const bool live = (t < 2) || include_replicated;
        const double s = (live && T.rows[t] > 0) ? block_colsum(T.tab[t], T.rows[t], DIAG_COLS, col, sm) : 0.0;
        if (threadIdx.x == 0) {
            if (col == 0) {
                out[MCL_DIAG_NORM_SQ + t] = s;
            } else if (col == 1) {
                for (int k = 0; k < MCL_MAX_REGS; ++k)
                    out[MCL_DIAG_REG + (t * MCL_MAX_REGS + k) * 2 + 1] = (k < T.nreg[t]) ? s : 0.0;
            } else {
                const int k = col - 2;
                out[MCL_DIAG_REG + (t * MCL_MAX_REGS + k) * 2] = (k < T.nreg[t]) ? s : 0.0;
            }
        }
    } else if (b < 3 * DIAG_COLS + 2) {
        const int col = b - 3 * DIAG_COLS;
        const double s = (T.I > 0) ? block_colsum(T.e1, T.I, 2, col, sm) : 0.0;
        if (threadIdx.x == 0) out[col == 0 ? MCL_DIAG_INNER : MCL_DIAG_MODEL_SQ] = s;
    } else if (threadIdx.x == 0) {
        out[MCL_DIAG_X_SQ] = T.xsq[0];
        out[6] = 0.0;
        out[7] = 0.0;
    }
}

// ---- device-side stopping rule (mcl_run) ---------------------------------------------------------------------------
// The reduction of the diagnostics tables of k_diag_final, and in the LAST block to finish (ticket) the stopping test of
// the reference's outer loop (decomposition.py:990-1053) on the reduced vector: feasibility gaps (:404-417, :996),
// relative reconstruction error (:1013-1014), regularised loss (:1016-1023), relative / absolute loss criterion
// (:1037-1053) with the quirks kept (Q8: the absolute criterion only counts when `tol` is set; Q9: it tests the newest
// loss; Q10: no loss on an infeasible iterate unless the caller records errors, so "previous loss" means the previous
// COMPUTED one).  A hit sets the gate flag every state-writing kernel of the iterations enqueued behind this one tests
// (MCL_GATE), and is reported - like every iteration's progress - through four int32 in pinned host memory.
struct StopRuleDev {
    double tol, abs_tol, feas_tol;
    double l2[3];
    double w[3][MCL_MAX_REGS];
    int has_tol, has_feas, loss_always, it;
};

// one thread: the stopping test on a complete diagnostics vector (read with device-scope loads: other workgroups wrote it)
static __device__ void verdict_eval(const double *out, const int *nreg, const StopRuleDev &R, int *gate, double *state,
                                    double *verdict_row, int *status_host) {
    auto ld = [&](int i) { return __hip_atomic_load(out + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    // feasibility gaps ||aux - factor|| / ||factor||, worst over all penalties of all modes (vacuously feasible without any)
    double worst = -__builtin_inf();
    for (int m = 0; m < 3; ++m) {
        const double fn = sqrt(ld(MCL_DIAG_NORM_SQ + m));
        for (int k = 0; k < nreg[m]; ++k) {
            const double gap = sqrt(ld(MCL_DIAG_REG + (m * MCL_MAX_REGS + k) * 2)) / fn;
            worst = (gap > worst || gap != gap) ? gap : worst;  // a NaN gap is never feasible
        }
    }
    const bool feasible = R.has_feas && (worst < R.feas_tol);
    int code = 0, computed = 0;
    double rec = 0.0, loss = 0.0;
    if (feasible || R.loss_always) {
        const double xsq = ld(MCL_DIAG_X_SQ), inner = ld(MCL_DIAG_INNER), model = ld(MCL_DIAG_MODEL_SQ);
        rec = sqrt(fmax(0.0, xsq - 2.0 * inner + model)) / sqrt(xsq);
        double reg = 0.0;
        for (int m = 0; m < 3; ++m) {
            for (int k = 0; k < nreg[m]; ++k)
                if (R.w[m][k] != 0.0) reg += R.w[m][k] * ld(MCL_DIAG_REG + (m * MCL_MAX_REGS + k) * 2 + 1);
            if (R.l2[m] != 0.0) reg += 0.5 * R.l2[m] * ld(MCL_DIAG_NORM_SQ + m);
        }
        loss = 0.5 * (rec * rec) + reg;
        computed = 1;
        if (R.has_tol) {
            const double prev = state[0];
            const bool rel = fabs(prev - loss) < R.tol * prev;
            const bool absc = loss < R.abs_tol;
            if (feasible && rel) code = 1;
            else if (feasible && absc) code = 2;
        }
        state[0] = loss;
    }
    verdict_row[0] = rec, verdict_row[1] = loss, verdict_row[2] = worst;
    verdict_row[3] = (double)((feasible ? 1 : 0) | (computed << 1) | (code << 2));
    if (code) {
        gate[1] = R.it, gate[2] = code;
        __threadfence();
        gate[0] = 1;
        status_host[1] = R.it, status_host[2] = code;
        __threadfence_system();
        status_host[0] = 1;
    }
    status_host[3] = R.it + 1;  // progress: the host keeps its run-ahead bounded by this
    __threadfence_system();
}

__global__ __launch_bounds__(1024) void k_diag_verdict(DiagTables T, double *__restrict__ out, StopRuleDev R,
                                                       int *__restrict__ gate, double *__restrict__ state,
                                                       double *__restrict__ verdict_row, int *__restrict__ status_host) {
    __shared__ double sm[16];
    __shared__ int last;
    if (*gate != 0) return;  // an earlier iteration has stopped the run: nothing is evaluated any more
    const int b = blockIdx.x;
    if (b < 3 * DIAG_COLS) {
        const int t = b / DIAG_COLS, col = b - t * DIAG_COLS;
        const double s = (T.rows[t] > 0) ? block_colsum(T.tab[t], T.rows[t], DIAG_COLS, col, sm) : 0.0;
        if (threadIdx.x == 0) {
            if (col == 0) {
                out[MCL_DIAG_NORM_SQ + t] = s;
            } else if (col == 1) {
                for (int k = 0; k < MCL_MAX_REGS; ++k)
                    out[MCL_DIAG_REG + (t * MCL_MAX_REGS + k) * 2 + 1] = (k < T.nreg[t]) ? s : 0.0;
            } else {
                const int k = col - 2;
                out[MCL_DIAG_REG + (t * MCL_MAX_REGS + k) * 2] = (k < T.nreg[t]) ? s : 0.0;
            }
        }
    } else if (b < 3 * DIAG_COLS + 2) {
        const int col = b - 3 * DIAG_COLS;
        const double s = (T.I > 0) ? block_colsum(T.e1, T.I, 2, col, sm) : 0.0;
        if (threadIdx.x == 0) out[col == 0 ? MCL_DIAG_INNER : MCL_DIAG_MODEL_SQ] = s;
    } else if (threadIdx.x == 0) {
        out[MCL_DIAG_X_SQ] = T.xsq[0];
        out[6] = 0.0;
        out[7] = 0.0;
    }
    if (threadIdx.x == 0) {
        __threadfence();  // this block's entries of `out` are visible device-wide before its ticket
        const int ticket = atomicAdd(gate + 3, 1);
        last = ticket == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (!last || threadIdx.x != 0) return;
    __threadfence();
    gate[3] = 0;  // ticket counter ready for the next launch (stream order)
    verdict_eval(out, T.nreg, R, gate, state, verdict_row, status_host);
}

// The same test on a vector that is already complete - the sharded loop: every rank reduces its tables (mcl_diagnostics),
// the host all-reduces the vectors, then every rank evaluates the rule on the SAME bits and reaches the same verdict.
__global__ __launch_bounds__(64) void k_verdict(const double *__restrict__ vec, StopRuleDev R, int n0, int n1, int n2,
                                                int *__restrict__ gate, double *__restrict__ state,
                                                double *__restrict__ verdict_row, int *__restrict__ status_host) {
    if (*gate != 0 || threadIdx.x != 0) return;
    const int nreg[3] = {n0, n1, n2};
    verdict_eval(vec, nreg, R, gate, state, verdict_row, status_host);
}

// Per-tile diagnostics of a packed factor from memory (generic path / initial state):
// ||F||^2, sum|F|, ||Z_k - F||^2 with Z_k = aux_k or P Delta (PARAFAC2; product on the MFMA).  Tile layout of rows_mfma.h.
template <int NBR, bool VEC>
__global__ __launch_bounds__(256) void k_rows_diag(const int *__restrict__ tile_row0, const int *__restrict__ tile_nrows,
                                                   int n_tiles, const float *__restrict__ F, RegSet regs, int r,
                                                   double *__restrict__ diag_tile) {
    MCL_GATE(regs.gate);
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= n_tiles) return;
    const long row0 = __builtin_amdgcn_readfirstlane(tile_row0[tile]);
    const int nrows = __builtin_amdgcn_readfirstlane(tile_nrows[tile]);
    const int row16 = lane & 15, g = lane >> 4;
    int kpf2 = -1;
    for (int k = 0; k < regs.n; ++k)
        if (regs.kind[k] == MCL_PEN_PARAFAC2) kpf2 = k;
    RowMat<NBR> D;
    if (kpf2 >= 0) D.load(regs.aux2[kpf2], r, lane);
    double nf = 0.0, na = 0.0, gap[MCL_MAX_REGS];
#pragma unroll
    for (int k = 0; k < MCL_MAX_REGS; ++k) gap[k] = 0.0;
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
        if (16 * rb >= nrows) break;
        const bool ok = 16 * rb + row16 < nrows;
        const long j = row0 + 16 * rb + (ok ? row16 : 0);
        f32x4 f[NBR];
#pragma unroll
        for (int h = 0; h < NBR; ++h) {
            f[h] = row_ld4<VEC>(F, j, 16 * h + 4 * g, ok, r);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                nf += (double)f[h][v] * (double)f[h][v];
                na += fabs((double)f[h][v]);
            }
        }
#pragma unroll
        for (int k = 0; k < MCL_MAX_REGS; ++k) {
            if (k < regs.n) {
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
                for (int h = 0; h < NBR; ++h)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const double dlt = (double)z[h][v] - (double)f[h][v];  // padded / invalid entries are 0 - 0
                        gap[k] += dlt * dlt;
                    }
            }
        }
    }
    nf = wave_sum(nf);
    na = wave_sum(na);
#pragma unroll
    for (int k = 0; k < MCL_MAX_REGS; ++k) gap[k] = wave_sum(gap[k]);
    if (lane == 0) {
        double *o = diag_tile + (long)tile * DIAG_COLS;
        o[0] = nf;
        o[1] = na;
#pragma unroll
        for (int k = 0; k < MCL_MAX_REGS; ++k) o[2 + k] = gap[k];
    }
}

// =========================================================================================================
// host launchers
// =========================================================================================================
bool mcl_mode_is_row_separable(const mcl_context *c, int mode) {
    const RegSet &rs = c->regs[mode];
    for (int k = 0; k < rs.n; ++k)
        if (rs.kind[k] != MCL_PEN_NN && rs.kind[k] != MCL_PEN_BOX && rs.kind[k] != MCL_PEN_L1) return false;
    return true;
}

#define DISPATCH_RP_T(c, KERNEL, grid, block, ...)                                                       \
    switch ((c)->RP) {                                                                                   \
        case 4: hipLaunchKernelGGL((KERNEL<4>), grid, block, 0, (c)->stream, __VA_ARGS__); break;         \
        case 8: hipLaunchKernelGGL((KERNEL<8>), grid, block, 0, (c)->stream, __VA_ARGS__); break;         \
        case 16: hipLaunchKernelGGL((KERNEL<16>), grid, block, 0, (c)->stream, __VA_ARGS__); break;       \
        case 32: hipLaunchKernelGGL((KERNEL<32>), grid, block, 0, (c)->stream, __VA_ARGS__); break;       \
        default: hipLaunchKernelGGL((KERNEL<64>), grid, block, 0, (c)->stream, __VA_ARGS__); break;       \
    }

int mcl_launch_ctc(mcl_context *c) {
    ProfScope prof_(c, MCL_PROF_OTHER);
    const int RPc = c->RP < 4 ? 4 : c->RP;
    hipLaunchKernelGGL(k_ctc, dim3((unsigned)c->r), dim3(256), 0, c->stream, c->C, (int)c->K, c->r, RPc, c->CtC, c->CtC64);
    MCL_CHECK_HIP(c, hipGetLastError());
    return 0;
}

int mcl_launch_B_rho(mcl_context *c) {
    ProfScope prof_(c, MCL_PROF_OTHER);
    MCL_CHECK_HIP(c, hipMemsetAsync(c->rho_max, 0, sizeof(float), c->stream));
    if (c->I == 0) return 0;
    hipLaunchKernelGGL(k_B_rho, dim3((unsigned)((c->I + 255) / 256)), dim3(256), 0, c->stream, c->CtC64, c->A, (int)c->I,
                       c->r, (float)c->opt.feasibility_penalty_scale, c->rhoB, c->rho_max);
    MCL_CHECK_HIP(c, hipGetLastError());
    return 0;
}

int mcl_launch_B_systems(mcl_context *c) {
    ProfScope prof_(c, MCL_PROF_OTHER);
    if (c->I == 0) return 0;
    dim3 grid((unsigned)((c->I + 3) / 4)), block(256);
    DISPATCH_RP_T(c, k_B_systems, grid, block, c->CtC64, c->A, (int)c->I, c->r, (float)c->opt.feasibility_penalty_scale,
                  (float)c->opt.l2_penalty[1], c->regs[1].n, c->opt.constant_B, c->rho_max, c->rhoB, c->LinvB, c->LinvB64);
    MCL_CHECK_HIP(c, hipGetLastError());
    return 0;
}

// exact-products mode (mcl_exact_mode): X C as fp64 sums of exact products, with its once-rounded fp32 image
int mcl_launch_exact_xc(mcl_context *c) {
    ProfScope prof_(c, MCL_PROF_XC);
    c->variant[MCL_PROF_XC] = "k_contract_xc_f64";
    if (c->N == 0) return 0;
    const dim3 g((unsigned)(((c->N + 15) / 16 + 3) / 4));
    if (c->NB == 1)
        hipLaunchKernelGGL(k_contract_xc_f64<1>, g, dim3(256), 0, c->stream, c->X, c->C, (long)c->N, (int)c->K, c->r, c->XC64, c->XC);
    else if (c->NB == 2)
        hipLaunchKernelGGL(k_contract_xc_f64<2>, g, dim3(256), 0, c->stream, c->X, c->C, (long)c->N, (int)c->K, c->r, c->XC64, c->XC);
    else
        hipLaunchKernelGGL(k_contract_xc_f64<4>, g, dim3(256), 0, c->stream, c->X, c->C, (long)c->N, (int)c->K, c->r, c->XC64, c->XC);
    MCL_CHECK_HIP(c, hipGetLastError());
    return 0;
}

int mcl_launch_B_solve_f64(mcl_context *c) {
    ProfScope prof_(c, MCL_PROF_ROWS_FUSED);
    c->variant[MCL_PROF_ROWS_FUSED] = "k_contract_xc_f64 + k_B_solve_f64";
    if (c->tilesB.n_tiles == 0) return 0;
    {  // X C in fp64 (one inner iteration per phase for a penalty-free mode: computed right here)
        const dim3 g((unsigned)(((c->N + 15) / 16 + 3) / 4));
        if (c->NB == 1)
            hipLaunchKernelGGL(k_contract_xc_f64<1>, g, dim3(256), 0, c->stream, c->X, c->C, (long)c->N, (int)c->K, c->r, c->XC64);
        else if (c->NB == 2)
            hipLaunchKernelGGL(k_contract_xc_f64<2>, g, dim3(256), 0, c->stream, c->X, c->C, (long)c->N, (int)c->K, c->r, c->XC64);
        else
            hipLaunchKernelGGL(k_contract_xc_f64<4>, g, dim3(256), 0, c->stream, c->X, c->C, (long)c->N, (int)c->K, c->r, c->XC64);
    }
    dim3 grid((unsigned)((c->tilesB.n_tiles + 3) / 4)), block(256);
    DISPATCH_RP_T(c, k_B_solve_f64, grid, block, c->tilesB.slab, c->tilesB.row0, c->tilesB.nrows, c->tilesB.n_tiles, c->XC64,
                  c->A, c->LinvB64, c->r, c->B, c->gate_active);
    MCL_CHECK_HIP(c, hipGetLastError());
    return 0;
}

int mcl_launch_C_prepare(mcl_context *c) {
    dim3 grid((unsigned)(1 + (c->K * c->r + 63) / 64)), block(64);
    DISPATCH_RP_T(c, k_C_prepare, grid, block, c->GR, c->r, (int)c->K, (float)c->opt.feasibility_penalty_scale,
                  (float)c->opt.l2_penalty[2], c->regs[2].n, c->rhoC, c->LinvC, c->LinvC64, c->GRf + (long)c->r * c->r);
    MCL_CHECK_HIP(c, hipGetLastError());
    return 0;
}

int mcl_launch_C_solve_f64(mcl_context *c) {
    const long n = (long)c->K * c->r;
    hipLaunchKernelGGL(k_C_solve_f64, dim3((unsigned)((n + 255) / 256)), dim3(256), sizeof(double) * c->r * c->r, c->stream,
                       c->GR + (long)c->r * c->r, c->LinvC64, (int)c->K, c->r, c->C, c->gate_active);
    MCL_CHECK_HIP(c, hipGetLastError());
    return 0;
}

template <int NBR, int NREG>
static void launch_rows_fused_t(mcl_context *c, const TileMap &tm, const float *rhs, const float *Arows,
                                const float *rho, const float *Linv, float *F, const RegSet &rs, double *diag) {
    bool vec = (c->r % 4 == 0) && ((reinterpret_cast<uintptr_t>(rhs) & 15) == 0) &&
               ((reinterpret_cast<uintptr_t>(F) & 15) == 0);
    for (int k = 0; k < rs.n; ++k)
        vec = vec && ((reinterpret_cast<uintptr_t>(rs.aux[k]) & 15) == 0) && ((reinterpret_cast<uintptr_t>(rs.dual[k]) & 15) == 0);
    dim3 grid((unsigned)((tm.n_tiles + 3) / 4)), block(256);
    if (vec)
        hipLaunchKernelGGL((k_rows_fused<NBR, NREG, true>), grid, block, 0, c->stream, tm.slab, tm.row0, tm.nrows,
                           tm.n_tiles, rhs, Arows, rho, Linv, F, rs, c->r, c->opt.inner_n_iter_max, diag);
    else
        hipLaunchKernelGGL((k_rows_fused<NBR, NREG, false>), grid, block, 0, c->stream, tm.slab, tm.row0, tm.nrows,
                           tm.n_tiles, rhs, Arows, rho, Linv, F, rs, c->r, c->opt.inner_n_iter_max, diag);
}

template <int NBR>
static int launch_rows_fused_n(mcl_context *c, const TileMap &tm, const float *rhs, const float *Arows,
                               const float *rho, const float *Linv, float *F, const RegSet &rs, double *diag) {
    switch (rs.n) {
        case 0: launch_rows_fused_t<NBR, 0>(c, tm, rhs, Arows, rho, Linv, F, rs, diag); return 0;
        case 1: launch_rows_fused_t<NBR, 1>(c, tm, rhs, Arows, rho, Linv, F, rs, diag); return 0;
        case 2: launch_rows_fused_t<NBR, 2>(c, tm, rhs, Arows, rho, Linv, F, rs, diag); return 0;
        case 3:
            if (NBR == 1) {
                launch_rows_fused_t<1, 3>(c, tm, rhs, Arows, rho, Linv, F, rs, diag);
                return 0;
            }
            return -1;
        default: return -1;
    }
}

// returns -1 when the (rank, #penalties) combination has no fused instantiation (caller uses the generic path)
int mcl_rows_fused_dispatch(mcl_context *c, int mode, double *diag) {
    const RegSet &rs = c->regs[mode];
    const TileMap &tm = (mode == 1) ? c->tilesB : c->tilesC;
    if (tm.n_tiles == 0) return 0;
    const float *rhs = (mode == 1) ? c->XC : c->GRf + (long)c->r * c->r;  // fp32 image of R (k_C_prepare)
    const float *Arows = (mode == 1) ? c->A : nullptr;
    const float *rho = (mode == 1) ? c->rhoB : c->rhoC;
    const float *Linv = (mode == 1) ? c->LinvB : c->LinvC;
    float *F = (mode == 1) ? c->B : c->C;
    if (c->r <= 16) return launch_rows_fused_n<1>(c, tm, rhs, Arows, rho, Linv, F, rs, diag);
    if (c->r <= 32) return launch_rows_fused_n<2>(c, tm, rhs, Arows, rho, Linv, F, rs, diag);
    if (rs.n <= 1) return launch_rows_fused_n<4>(c, tm, rhs, Arows, rho, Linv, F, rs, diag);
    return -1;
}

int mcl_launch_rows_fused(mcl_context *c, int mode) {
    double *diag = (mode == 1) ? c->diagB_tile : c->diagC_tile;
    int rc;
    if (mode == 1) {
        ProfScope prof(c, MCL_PROF_ROWS_FUSED);
        rc = mcl_rows_fused_dispatch(c, mode, diag);
        char buf[64];
        snprintf(buf, sizeof buf, "k_rows_fused<NBR=%d,NREG=%d>", c->r <= 16 ? 1 : (c->r <= 32 ? 2 : 4), c->regs[1].n);
        c->variant[MCL_PROF_ROWS_FUSED] = buf;
    } else {
        rc = mcl_rows_fused_dispatch(c, mode, diag);
    }
    if (rc < 0) return rc;
    c->diag_rows[mode] = (((mode == 1) ? c->tilesB.n_tiles : c->tilesC.n_tiles) + 3) / 4;  // one row per block
    MCL_CHECK_HIP(c, hipGetLastError());
    return 0;
}

template <int NBR, int NREG>
static int launch_C_fused_t(mcl_context *c) {
    const RegSet &rs = c->regs[2];
    bool vec = (c->r % 4 == 0) && ((reinterpret_cast<uintptr_t>(c->C) & 15) == 0) && ((c->r * c->r) % 4 == 0);
    for (int k = 0; k < rs.n; ++k)
        vec = vec && ((reinterpret_cast<uintptr_t>(rs.aux[k]) & 15) == 0) && ((reinterpret_cast<uintptr_t>(rs.dual[k]) & 15) == 0);
    const int rpw = c->K <= 256 ? 16 : (c->K <= 512 ? 32 : 64);
    const int n_waves = (int)((c->K + rpw - 1) / rpw);
    size_t sm = sizeof(float) * (size_t)(c->r * c->r + c->K * c->r);
    if (NBR == 1) sm = ((sm + 7) & ~size_t(7)) + sizeof(double) * 256 * (size_t)n_waves;  // per-wave CtC slots
    int kct = 0;
    const int KC = mcl_xc_chunks(c, &kct);
#define MCL_CF(VEC_)                                                                                                  \
    do {                                                                                                              \
        if (sm > 65536)                                                                                               \
            MCL_CHECK_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_C_finish_fused<NBR, NREG, VEC_>),   \
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));               \
        hipLaunchKernelGGL((k_C_finish_fused<NBR, NREG, VEC_>), dim3(1), dim3(64 * n_waves), sm, c->stream, c->GR,    \
                           (int)c->K, c->r, (float)c->opt.feasibility_penalty_scale, (float)c->opt.l2_penalty[2],     \
                           c->rhoC, c->LinvC, c->C, rs, c->opt.inner_n_iter_max, c->CtC, c->CtC64, c->Cfrag, mcl_cfrag_chunks(c), c->NB, \
                           c->diagC_tile, rpw);                                                                       \
    } while (0)
    if (vec) MCL_CF(true);
    else MCL_CF(false);
#undef MCL_CF
    MCL_CHECK_HIP(c, hipGetLastError());
    c->variant[MCL_PROF_C_FINISH] = "k_C_finish_fused<NBR=" + std::to_string(NBR) + ",NREG=" + std::to_string(NREG) + ">";
    c->diag_rows[2] = 1;
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// The C-side finish over ceil(K / 64) workgroups (rank <= 16, K >= 128).  In the single workgroup of
// k_C_finish_fused the 16-row tiles of C queue four deep on every SIMD (9 k of its 23.5 k cycles at K = 256) and C^T C, the
// fragment image and the barriers see all K rows; here a workgroup owns 64 rows = one 16-row tile per wave = one 64-column
// chunk of the fragment image.  Every workgroup builds and inverts the (identical) r x r system itself - same inputs,
// same instructions, same bits - so nothing crosses workgroups inside the launch: C^T C leaves as one partial 16 x 16
// fp64 block per workgroup, which the A-phase finish sums in fixed order (AFuse::ctc_parts; ensure_ctc folds them for
// any other consumer).
// ---------------------------------------------------------------------------------------------------------
template <int NREG, bool VEC>
__global__ __launch_bounds__(256) void k_C_finish_multi(const double *__restrict__ GR, int K, int r, float scale, float l2,
                                                        float *__restrict__ rhoC, float *__restrict__ LinvC,
                                                        float *__restrict__ C, RegSet regs, int inner,
                                                        double *__restrict__ CtCpart, float *__restrict__ Cfrag, int KC,
                                                        double *__restrict__ diag_row) {
    MCL_GATE(regs.gate);
    __shared__ float Ls[256];
    __shared__ float Cs[64 * 16];
    __shared__ double dsm[4][DIAG_COLS];
    __shared__ double slots[4][256];
    __shared__ float rho_s;
    __shared__ double Ls64[NREG == 0 ? 256 : 1];  // penalty-free C: the solve itself runs in fp64 (as in k_C_finish_fused)
    constexpr int RP = 16, RL = GJRows<RP>::RL;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b0 = 64 * blockIdx.x;
    const long row0 = b0 + 16 * wave;
    const int nrows = max(0, min(16, K - (int)row0));
    RowBlock<1, NREG, VEC> pre;
    if (NREG > 0 && nrows > 0)
        pre.load(lane, row0 + ((lane & 15) < nrows ? (lane & 15) : 0), r, nullptr, GR + (long)r * r, regs);
    if (wave == 0) {  // the system, as in k_C_finish_fused
        const int cc = lane % RP, g = lane / RP;
        const bool act = cc < r;
        const int cl = act ? cc : 0;
        double gv[RL];
        double tr = 0.0;
#pragma unroll
        for (int j = 0; j < RL; ++j) {
            const int d = g * RL + j;
            gv[j] = GR[(d < r ? d : r - 1) * r + cl];
        }
#pragma unroll
        for (int j = 0; j < RL; ++j)
            if (act && g * RL + j == cc) tr = gv[j];
        tr = wave_sum(tr);
        const float rho = (float)(0.5 * tr * scale);
        const double shift = (double)rho * NREG + (double)l2;
        double col[RL];
#pragma unroll
        for (int j = 0; j < RL; ++j) {
            const int d = g * RL + j;
            double v = (d == cc) ? 1.0 : 0.0;
            if (act && d < r) v = gv[j] + (d == cc ? shift : 0.0);
            col[j] = v;
        }
        gj_inverse_rows<RP>(col, r, lane);
#pragma unroll
        for (int j = 0; j < RL; ++j) {
            const int d = g * RL + j;
            if (act && d < r) {
                Ls[d * r + cc] = (float)col[j];
                if (NREG == 0) Ls64[d * r + cc] = col[j];
                if (blockIdx.x == 0) LinvC[d * r + cc] = (float)col[j];
            }
        }
        if (lane == 0) {
            rho_s = rho;
            if (blockIdx.x == 0) rhoC[0] = rho;
        }
    }
    __syncthreads();
    double dg[DIAG_COLS];
    // (the LDS copy is indexed by the GLOBAL row like C itself: its base is shifted back by the workgroup's first row)
    float *Csg = Cs - (long)b0 * r;
    if (NREG == 0) {
        // C = R G^-1 in fp64 (un-shifted normal equations: every rounding of the product is amplified by cond(G))
        const double *R = GR + (long)r * r;
        double nf = 0.0, na = 0.0;
        if (inner > 0)
            for (int e = lane; e < nrows * r; e += 64) {
                const int rl = e / r, cidx = e - rl * r;
                const long j = row0 + rl;
                double acc = 0.0;
                for (int d = 0; d < r; ++d) acc = fma(R[j * r + d], Ls64[d * r + cidx], acc);
                const float f = (float)acc;
                C[j * r + cidx] = f;
                Csg[j * r + cidx] = f;
                nf += (double)f * (double)f;
                na += fabs((double)f);
            }
        dg[0] = wave_sum(nf);
        dg[1] = wave_sum(na);
    } else {
        rows_fused_tile<1, NREG, VEC>(lane, row0, nrows, rho_s, Ls, nullptr, nullptr, C, Csg, regs, r, inner, dg,
                                      GR + (long)r * r, &pre);
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 2 + NREG; ++k) dsm[wave][k] = dg[k];
    }
    __syncthreads();
    if (threadIdx.x < 2 + NREG)  // one diagnostics row per workgroup
        diag_row[(long)blockIdx.x * DIAG_COLS + threadIdx.x] =
            (dsm[0][threadIdx.x] + dsm[1][threadIdx.x]) + (dsm[2][threadIdx.x] + dsm[3][threadIdx.x]);
    // partial C^T C of the workgroup's rows on the fp64 MFMA (fp32 x fp32 products are exact in fp64)
    {
        typedef double f64x4 __attribute__((ext_vector_type(4)));
        const int rsub = lane >> 4, c16 = lane & 15;
        f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int rl = 4 * (4 * gq + wave) + rsub;  // groups of 4 rows, interleaved over the waves
            const double y = (b0 + rl < K && c16 < r) ? (double)Cs[rl * r + c16] : 0.0;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, acc, 0, 0, 0);
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) slots[wave][(rsub + 4 * v) * 16 + c16] = acc[v];
        __syncthreads();
        const int e = threadIdx.x;
        CtCpart[(long)blockIdx.x * 256 + e] = (slots[0][e] + slots[1][e]) + (slots[2][e] + slots[3][e]);
    }
    // the workgroup's chunk of the fragment image of C (see k_build_cfrag; NB = 1): 1024 floats; the last workgroup also
    // clears the chunks past K
    for (int idx = threadIdx.x; idx < 1024; idx += 256) {
        const int m = idx & 3, ln = (idx >> 2) & 63, kq = idx >> 8;
        const int kl = 16 * kq + 4 * (ln >> 4) + m, col = ln & 15;
        Cfrag[(long)blockIdx.x * 1024 + idx] = (b0 + kl < K && col < r) ? Cs[kl * r + col] : 0.f;
    }
    if (blockIdx.x == gridDim.x - 1)
        for (long idx = (long)gridDim.x * 1024 + threadIdx.x; idx < (long)KC * 1024; idx += 256) Cfrag[idx] = 0.f;
}

// C^T C from the partial blocks of k_C_finish_multi, for the consumers other than the A-phase finish
__global__ __launch_bounds__(256) void k_ctc_fold(const double *__restrict__ CtCpart, int parts, int r, float *__restrict__ CtC,
                                                  double *__restrict__ CtC64) {
    const int e = threadIdx.x, a = e >> 4, b = e & 15;
    double t = 0.0;
    for (int p = 0; p < parts; ++p) t = (p == 0) ? CtCpart[e] : t + CtCpart[(long)p * 256 + e];
    if (a < r && b < r) {
        CtC64[a * r + b] = t;
        CtC[a * r + b] = (float)t;
    }
}

int mcl_launch_ctc_fold(mcl_context *c) {
    ProfScope prof_(c, MCL_PROF_OTHER);
    hipLaunchKernelGGL(k_ctc_fold, dim3(1), dim3(256), 0, c->stream, c->CtCpart, c->ctc_parts, c->r, c->CtC, c->CtC64);
    MCL_CHECK_HIP(c, hipGetLastError());
    c->ctc_parts = 0;
    return 0;
}

template <int NREG>
static int launch_C_multi_t(mcl_context *c) {
    const RegSet &rs = c->regs[2];
    bool vec = (c->r % 4 == 0) && ((reinterpret_cast<uintptr_t>(c->C) & 15) == 0) && ((c->r * c->r) % 4 == 0);
    for (int k = 0; k < rs.n; ++k)
        vec = vec && ((reinterpret_cast<uintptr_t>(rs.aux[k]) & 15) == 0) && ((reinterpret_cast<uintptr_t>(rs.dual[k]) & 15) == 0);
    const int nblk = (int)((c->K + 63) / 64);
#define MCL_CM(VEC_)                                                                                                   \
    hipLaunchKernelGGL((k_C_finish_multi<NREG, VEC_>), dim3(nblk), dim3(256), 0, c->stream, c->GR, (int)c->K, c->r,       \
                       (float)c->opt.feasibility_penalty_scale, (float)c->opt.l2_penalty[2], c->rhoC, c->LinvC, c->C, rs, \
                       c->opt.inner_n_iter_max, c->CtCpart, c->Cfrag, mcl_cfrag_chunks(c), c->diagC_tile)
    if (vec) MCL_CM(true);
    else MCL_CM(false);
#undef MCL_CM
    MCL_CHECK_HIP(c, hipGetLastError());
    c->variant[MCL_PROF_C_FINISH] = "k_C_finish_multi<NREG=" + std::to_string(NREG) + ">";
    c->diag_rows[2] = nblk;
    c->ctc_parts = nblk;
    return 0;
}

// Single-workgroup C-side finish; returns -1 if the shape has no instantiation (caller uses the separate kernels)
int mcl_launch_C_finish_fused(mcl_context *c) {
    if (c->K > 1024 || c->sw.no_fused_c) return -1;
    if ((size_t)(c->r * c->r + c->K * c->r) * sizeof(float) > 150 * 1024) return -1;
    const int n = c->regs[2].n;
    c->ctc_parts = 0;
    // rank <= 16 and at least two 64-row chunks: one workgroup per chunk (k_C_finish_multi); the
    // fragment image is then the one of the sweep / X C kernels with NB = 1
    if (c->r <= 16 && n <= 2 && c->K >= 128 && c->NB == 1 && !c->sw.no_multi_c &&
        mcl_cfrag_chunks(c) >= (int)((c->K + 63) / 64))
        return n == 0 ? launch_C_multi_t<0>(c) : (n == 1 ? launch_C_multi_t<1>(c) : launch_C_multi_t<2>(c));
    if (c->r <= 16) {
        switch (n) {
            case 0: return launch_C_fused_t<1, 0>(c);
            case 1: return launch_C_fused_t<1, 1>(c);
            case 2: return launch_C_fused_t<1, 2>(c);
            default: return -1;
        }
    }
    if (c->r <= 32) {
        switch (n) {
            case 0: return launch_C_fused_t<2, 0>(c);
            case 1: return launch_C_fused_t<2, 1>(c);
            default: return -1;
        }
    }
    return -1;
}

int mcl_launch_A_rho(mcl_context *c) {
    ProfScope prof_(c, MCL_PROF_OTHER);
    MCL_CHECK_HIP(c, hipMemsetAsync(c->rho_max + 1, 0, sizeof(float), c->stream));
    if (c->I == 0) return 0;
    hipLaunchKernelGGL(k_A_rho, dim3((unsigned)((c->I + 255) / 256)), dim3(256), 0, c->stream, c->seg_btb, c->CtC64,
                       (int)c->I, c->r, (float)c->opt.feasibility_penalty_scale, c->rhoA, c->rho_max);
    MCL_CHECK_HIP(c, hipGetLastError());
    return 0;
}

int mcl_launch_A_finish(mcl_context *c, bool fused_inner) {
    if (c->I == 0) return 0;
    // also prepare the next B-phase's systems when that is exact: fused inner loop, per-slab rho for B
    const bool rows_kernel = !(c->RP == 64 || c->RP == 4 || c->sw.a_finish_cols);
    // (a B-phase with fp64 row passes also needs the fp64 inverses: only the row-split kernels write them)
    const bool b_needs_64 = mcl_rows64(c) || c->exact;  // the B-phase reads the fp64 inverses (fp64 row passes / wide.hip)
    const int next_B = (fused_inner && !c->opt.constant_B && c->regs[1].n > 0 && !c->sw.no_next_b &&
                        (rows_kernel || !b_needs_64)) ? 1 : 0;
    const bool seg = c->use_seg_gram;
    dim3 grid((unsigned)((c->I + 3) / 4)), block(256);
#define MCL_AF_ARGS                                                                                                   \
    c->BtB, c->CtC64, (int)c->I, c->r, (float)c->opt.feasibility_penalty_scale, (float)c->opt.l2_penalty[0],          \
        c->opt.constant_A, c->rho_max, c->rhoA, c->LinvA, c->A, c->regs[0], c->opt.inner_n_iter_max,                  \
        fused_inner ? 1 : 0, c->e1, c->diagA_row, next_B, (float)c->opt.l2_penalty[1], c->regs[1].n, c->rhoB, c->LinvB, \
        (const int *)(seg ? (c->seg_from_sweep ? c->slab_part_ptr : c->slab_seg_ptr) : nullptr),                      \
        (const double *)c->seg_rhs, (const double *)((seg && c->seg_from_sweep) ? c->part_btb : c->seg_btb), c->rhsA
    // what the row-split kernel absorbs: rhs_i from the sweep's M_bseg
    AFuse F{};
    F.ctc_parts = c->ctc_parts, F.CtCpart = c->CtCpart, F.CtC64_out = c->CtC64, F.CtC_out = c->CtC;
    if (c->a_rhs_from_M) {
        F.Mpart = c->Mpart, F.Cfrag = c->CfragS, F.NBm = c->NB;
        F.MS = mcl_sweep_KC(c) * 64 * 16 * c->NB;
    }
    F.LinvB64 = b_needs_64 ? c->LinvB64 : nullptr;
    F.LinvA64 = c->LinvA64, F.rhsA64 = c->rhsA64, F.Q64 = c->Q64;  // (allocated in the exact-products mode only)
    F.wide_inner = (c->exact && !c->sw.no_wide) ? 1 : 0;
    if (!rows_kernel && c->ctc_parts > 0)
        if (int rc = mcl_launch_ctc_fold(c)) return rc;
    // ranks 5..32: rows of every system split over the lane groups (all 64 lanes busy); 64 columns fill the wave anyway
    if (!rows_kernel) {
        DISPATCH_RP_T(c, k_A_finish, grid, block, MCL_AF_ARGS, F.wide_inner, c->LinvA64, c->rhsA64, c->Q64);
    } else if (c->a_rhs_wide) {
        if (c->a_rhs_pairs) {  // two slabs per workgroup (one partial per slab, many slabs)
            const dim3 gp((unsigned)((c->I + 1) / 2));
            if (c->RP == 8) hipLaunchKernelGGL((k_A_finish_rows_wide<8, 2>), gp, block, 0, c->stream, MCL_AF_ARGS, F);
            else if (c->RP == 16) hipLaunchKernelGGL((k_A_finish_rows_wide<16, 2>), gp, block, 0, c->stream, MCL_AF_ARGS, F);
            else hipLaunchKernelGGL((k_A_finish_rows_wide<32, 2>), gp, block, 0, c->stream, MCL_AF_ARGS, F);
        } else {
            const dim3 gw((unsigned)c->I);
            if (c->RP == 8) hipLaunchKernelGGL((k_A_finish_rows_wide<8, 1>), gw, block, 0, c->stream, MCL_AF_ARGS, F);
            else if (c->RP == 16) hipLaunchKernelGGL((k_A_finish_rows_wide<16, 1>), gw, block, 0, c->stream, MCL_AF_ARGS, F);
            else hipLaunchKernelGGL((k_A_finish_rows_wide<32, 1>), gw, block, 0, c->stream, MCL_AF_ARGS, F);
        }
    } else if (c->RP == 8) {
        hipLaunchKernelGGL((k_A_finish_rows<8>), grid, block, 0, c->stream, MCL_AF_ARGS, F);
    } else if (c->RP == 16) {
        hipLaunchKernelGGL((k_A_finish_rows<16>), grid, block, 0, c->stream, MCL_AF_ARGS, F);
    } else {
        hipLaunchKernelGGL((k_A_finish_rows<32>), grid, block, 0, c->stream, MCL_AF_ARGS, F);
    }
#undef MCL_AF_ARGS
    MCL_CHECK_HIP(c, hipGetLastError());
    c->a64_valid = !fused_inner && c->Q64 != nullptr;  // rhsA64 / Q64 / LinvA64 hold this phase's systems
    c->variant[MCL_PROF_A_FINISH] = !rows_kernel ? "k_A_finish" : (c->a_rhs_wide ? (c->a_rhs_pairs ? "k_A_finish_rows_wide<SPB=2>" : "k_A_finish_rows_wide<SPB=1>") : "k_A_finish_rows");
    c->ctc_parts = 0;  // the rows kernels wrote the totals
    c->b_systems_valid = (next_B != 0);
    return 0;
}

int mcl_launch_A_e1(mcl_context *c, bool btb_is_q) {
    ProfScope prof_(c, MCL_PROF_OTHER);
    if (c->I == 0) return 0;
    // (the fp64 copies exist in the exact-products mode and are current when the systems were just built from them)
    const bool wide = btb_is_q && c->exact && c->a64_valid;
    hipLaunchKernelGGL(k_A_e1, dim3((unsigned)c->I), dim3(64), 0, c->stream, c->rhsA, c->BtB, c->CtC, btb_is_q ? 1 : 0,
                       c->A, c->regs[0], c->r, c->e1, c->diagA_row, wide ? (const double *)c->rhsA64 : nullptr,
                       wide ? (const double *)c->Q64 : nullptr);
    MCL_CHECK_HIP(c, hipGetLastError());
    return 0;
}

int mcl_launch_rows_diag(mcl_context *c, int mode) {
    ProfScope prof_(c, MCL_PROF_DIAG);
    c->variant[MCL_PROF_DIAG] = "k_diag_final";
    const TileMap &tm = (mode == 1) ? c->tilesB : (mode == 2 ? c->tilesC : c->tilesA);
    if (tm.n_tiles == 0) return 0;
    const float *F = (mode == 1) ? c->B : (mode == 2 ? c->C : c->A);
    double *diag = (mode == 1) ? c->diagB_tile : (mode == 2 ? c->diagC_tile : c->diagA_tile);
    bool vec = (c->r % 4 == 0) && ((reinterpret_cast<uintptr_t>(F) & 15) == 0);
    for (int k = 0; k < c->regs[mode].n; ++k) vec = vec && ((reinterpret_cast<uintptr_t>(c->regs[mode].aux[k]) & 15) == 0);
    dim3 grid((unsigned)((tm.n_tiles + 3) / 4)), block(256);
#define MCL_RD(NBR_, VEC_)                                                                                       \
    hipLaunchKernelGGL((k_rows_diag<NBR_, VEC_>), grid, block, 0, c->stream, tm.row0, tm.nrows, tm.n_tiles, F,   \
                       c->regs[mode], c->r, diag)
    const int nbr = c->r <= 16 ? 1 : (c->r <= 32 ? 2 : 4);
    if (vec) {
        if (nbr == 1) MCL_RD(1, true);
        else if (nbr == 2) MCL_RD(2, true);
        else MCL_RD(4, true);
    } else {
        if (nbr == 1) MCL_RD(1, false);
        else if (nbr == 2) MCL_RD(2, false);
        else MCL_RD(4, false);
    }
#undef MCL_RD
    c->diag_rows[mode] = tm.n_tiles;  // one row per tile
    MCL_CHECK_HIP(c, hipGetLastError());
    return 0;
}

int mcl_launch_x_sq(mcl_context *c) {
    ProfScope prof_(c, MCL_PROF_OTHER);
    const long n = (long)c->N * c->K;
    const int nb = 1024;
    hipLaunchKernelGGL(k_sumsq_partial, dim3(nb), dim3(256), 0, c->stream, c->X, n, c->xsq_part);
    hipLaunchKernelGGL(k_sum_doubles, dim3(1), dim3(256), 0, c->stream, c->xsq_part, nb, c->x_sq);
    MCL_CHECK_HIP(c, hipGetLastError());
    return 0;
}

DiagTables mcl_diag_tables(const mcl_context *c, bool a_from_rows) {
    DiagTables T;
    T.tab[0] = a_from_rows ? c->diagA_row : c->diagA_tile;
    T.rows[0] = a_from_rows ? (int)c->I : c->tilesA.n_tiles;
    T.tab[1] = c->diagB_tile, T.rows[1] = c->diag_rows[1];
    T.tab[2] = c->diagC_tile, T.rows[2] = c->diag_rows[2];
    for (int m = 0; m < 3; ++m) T.nreg[m] = c->regs[m].n;
    T.e1 = c->e1, T.I = (int)c->I, T.xsq = c->x_sq;
    return T;
}

int mcl_launch_diag_tables(mcl_context *c, const DiagTables &T, double *out, int include_replicated) {
    ProfScope prof_(c, MCL_PROF_DIAG);
    c->variant[MCL_PROF_DIAG] = "k_diag_final";
    // long tables (thousands of B tiles: config 4 / 5) are summed by 1024 threads per column: a quarter of the dependent loads
    const int nt = std::max(T.rows[0], std::max(T.rows[1], T.rows[2])) > 2048 ? 1024 : 256;
    hipLaunchKernelGGL(k_diag_final, dim3(3 * DIAG_COLS + 3), dim3(nt), 0, c->stream, T, include_replicated, out);
    MCL_CHECK_HIP(c, hipGetLastError());
    return 0;
}

static StopRuleDev rule_dev(const mcl_context *c, const mcl_stop_rule *rule, int it) {
    StopRuleDev R{};
    R.tol = rule->tol, R.abs_tol = rule->absolute_tol, R.feas_tol = rule->feasibility_tol;
    R.has_tol = rule->tol != 0.0, R.has_feas = rule->feasibility_tol != 0.0;  // Python truthiness of the keyword values
    R.loss_always = rule->evaluate_loss_always != 0, R.it = it;
    for (int m = 0; m < 3; ++m) {
        R.l2[m] = c->opt.l2_penalty[m];
        for (int k = 0; k < MCL_MAX_REGS; ++k) R.w[m][k] = rule->penalty_weight[m][k];
    }
    return R;
}

int mcl_launch_verdict(mcl_context *c, const double *vec, const mcl_stop_rule *rule, int it, double *verdict_row,
                       int *status_dev) {
    ProfScope prof_(c, MCL_PROF_DIAG);
    hipLaunchKernelGGL(k_verdict, dim3(1), dim3(64), 0, c->stream, vec, rule_dev(c, rule, it), c->regs[0].n, c->regs[1].n,
                       c->regs[2].n, c->gate, c->stop_state, verdict_row, status_dev);
    MCL_CHECK_HIP(c, hipGetLastError());
    return 0;
}

int mcl_launch_diag_verdict(mcl_context *c, double *out, const mcl_stop_rule *rule, int it, double *verdict_row,
                            int *status_dev) {
    const StopRuleDev R = rule_dev(c, rule, it);
    const DiagTables T = mcl_diag_tables(c, true);
    const int nt = std::max(T.rows[0], std::max(T.rows[1], T.rows[2])) > 2048 ? 1024 : 256;
    ProfScope prof_(c, MCL_PROF_DIAG);
    c->variant[MCL_PROF_DIAG] = "k_diag_verdict";
    hipLaunchKernelGGL(k_diag_verdict, dim3(3 * DIAG_COLS + 3), dim3(nt), 0, c->stream, T, out, R,
                       c->gate, c->stop_state, verdict_row, status_dev);
    MCL_CHECK_HIP(c, hipGetLastError());
    return 0;
}

int mcl_launch_diag_final(mcl_context *c, double *out, int include_replicated, bool a_from_rows) {
    return mcl_launch_diag_tables(c, mcl_diag_tables(c, a_from_rows), out, include_replicated);
}
