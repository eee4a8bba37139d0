// ADMM-side kernels of the AO-ADMM engine (gfx950): rank x rank normal-equation algebra (fp64, in LDS),
// the fused inner ADMM loop for row-separable penalties, the per-slab A-phase finish, and diagnostics.
//
// Reference sites: B1/B3-B5 (decomposition.py:240-256), B6-B9 (:259-285), C2-C3 (:319-338),
// A1/A3-A6 (:138,155-213), E1-E3 (:404-417, 445-452, 916-921), prox: penalties.py:503-586.
#include "mcl_internal.h"

#define FULL_TILE 64

// ---------------------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------------------
static __device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// In-place Gauss-Jordan inverse of the SPD r x r matrix M (fp64, LDS, row-major) by ONE wave (64 threads).
// SPD => no pivoting needed.  colp: r doubles of LDS scratch.  Replaces the reference's SVD-based solve
// `(v (U/s)) Uh` (decomposition.py:172,194) - identical for symmetric positive definite systems.
static __device__ void gj_inverse(double *M, double *colp, int r, int lane) {
    for (int p = 0; p < r; ++p) {
        __syncthreads();
        const double piv = 1.0 / M[p * r + p];
        for (int i = lane; i < r; i += 64) colp[i] = M[i * r + p];
        __syncthreads();
        for (int c = lane; c < r; c += 64) M[p * r + c] = (c == p) ? piv : M[p * r + c] * piv;
        __syncthreads();
        for (int e = lane; e < r * r; e += 64) {
            const int i = e / r, c = e - i * r;
            if (i != p) {
                const double f = colp[i];
                M[e] = (c == p) ? -f * piv : M[e] - f * M[p * r + c];
            }
        }
    }
    __syncthreads();
}

static __device__ __forceinline__ float prox_elem(int kind, int nonneg, float p0, float p1, float thr, float y) {
    switch (kind) {
        case MCL_PEN_NN:
            return fmaxf(y, 0.f);
        case MCL_PEN_BOX:
            return fminf(fmaxf(y, p0), p1);
        case MCL_PEN_L1:
            if (nonneg) return fmaxf(y - thr, 0.f);
            return copysignf(fmaxf(fabsf(y) - thr, 0.f), y);
        default:
            return y;
    }
}

// ---------------------------------------------------------------------------------------------------------
// CtC = C^T C  (fp64 accumulation, one workgroup; C staged through LDS in row chunks)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ctc(const float *__restrict__ C, int K, int r, float *__restrict__ CtC) {
    extern __shared__ float smf[];
    const int chunk_rows = 8192 / r;  // <= 32 KB of LDS
    const int npairs = r * r;
    double acc[16];                   // r*r <= 4096 pairs / 256 threads
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] = 0.0;
    for (int k0 = 0; k0 < K; k0 += chunk_rows) {
        const int rows = min(chunk_rows, K - k0);
        __syncthreads();
        for (int e = threadIdx.x; e < rows * r; e += 256) smf[e] = C[(long)k0 * r + e];
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int pr = threadIdx.x + 256 * t;
            if (pr < npairs) {
                const int a = pr / r, b = pr - a * r;
                double s = 0.0;
                for (int k = 0; k < rows; ++k) s += (double)smf[k * r + a] * (double)smf[k * r + b];
                acc[t] += s;
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int pr = threadIdx.x + 256 * t;
        if (pr < npairs) CtC[pr] = (float)acc[t];
    }
}

// ---------------------------------------------------------------------------------------------------------
// B-phase systems: rho_i = 1/2 tr(CtC o a_i a_i^T) * scale ; L_i = CtC o a_i a_i^T + (rho_i n + l2) I ; L_i^-1
// ---------------------------------------------------------------------------------------------------------
__global__ void k_B_rho(const float *__restrict__ CtC, const float *__restrict__ A, int I, int r, float scale,
                        float *__restrict__ rhoB, float *__restrict__ rho_max) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= I) return;
    double s = 0.0;
    for (int c = 0; c < r; ++c) {
        const double a = A[(long)i * r + c];
        s += (double)CtC[c * r + c] * a * a;
    }
    const float rho = (float)(0.5 * s * scale);
    rhoB[i] = rho;
    atomicMax(reinterpret_cast<int *>(rho_max), __float_as_int(rho));  // rho >= 0: int order == float order
}

__global__ __launch_bounds__(64) void k_B_systems(const float *__restrict__ CtC, const float *__restrict__ A, int r,
                                                  float scale, float l2, int n_regs, int constant,
                                                  const float *__restrict__ rho_max, float *__restrict__ rhoB,
                                                  float *__restrict__ Linv) {
    extern __shared__ double smd[];
    double *M = smd, *colp = smd + r * r;
    const int i = blockIdx.x, lane = threadIdx.x;
    const float *a = A + (long)i * r;
    double tr = 0.0;
    for (int c = 0; c < r; ++c) tr += (double)CtC[c * r + c] * (double)a[c] * (double)a[c];
    float rho = (float)(0.5 * tr * scale);
    if (constant) rho = rho_max[0];
    const double shift = (double)rho * n_regs + (double)l2;
    for (int e = lane; e < r * r; e += 64) {
        const int c = e / r, d = e - c * r;
        M[e] = (double)CtC[e] * (double)a[c] * (double)a[d] + (c == d ? shift : 0.0);
    }
    gj_inverse(M, colp, r, lane);
    for (int e = lane; e < r * r; e += 64) Linv[(long)i * r * r + e] = (float)M[e];
    if (lane == 0) rhoB[i] = rho;
}

// C-phase system from the (all-reduced) normal equations [G | R]
__global__ __launch_bounds__(64) void k_C_prepare(const float *__restrict__ GR, int r, float scale, float l2,
                                                  int n_regs, float *__restrict__ rhoC, float *__restrict__ LinvC) {
    extern __shared__ double smd[];
    double *M = smd, *colp = smd + r * r;
    const int lane = threadIdx.x;
    double tr = 0.0;
    for (int c = 0; c < r; ++c) tr += (double)GR[c * r + c];
    const float rho = (float)(0.5 * tr * scale);
    const double shift = (double)rho * n_regs + (double)l2;
    for (int e = lane; e < r * r; e += 64) {
        const int c = e / r, d = e - c * r;
        M[e] = (double)GR[e] + (c == d ? shift : 0.0);
    }
    gj_inverse(M, colp, r, lane);
    for (int e = lane; e < r * r; e += 64) LinvC[e] = (float)M[e];
    if (lane == 0) rhoC[0] = rho;
}

// ---------------------------------------------------------------------------------------------------------
// Fused inner ADMM loop for row-separable penalties (NN / Box / L1): one lane per packed row, the whole
// `inner` iterations in registers; reads rhs/aux/dual once, writes factor/aux/dual once.
//   t   = rhs + rho * sum_k (z_k - u_k)            decomposition.py:269-273 / 328-331
//   f   = t L^-1                                   (L^-1 is wave-uniform: scalar loads)
//   z_k = prox_k(f + u_k, rho) ; u_k = f - (z_k - u_k)     decomposition.py:278-285 / 337-338
// Per-tile diagnostics (fp64): ||f||^2, sum|f|, ||z_k - f||^2.
// ---------------------------------------------------------------------------------------------------------
template <int RP, int NREG>
__global__ __launch_bounds__(256) void k_rows_fused(const int *__restrict__ tile_slab, const int *__restrict__ tile_row0,
                                                    const int *__restrict__ tile_nrows, int n_tiles,
                                                    const float *__restrict__ rhs_src, const float *__restrict__ Arows,
                                                    const float *__restrict__ rho_arr, const float *__restrict__ Linv,
                                                    float *__restrict__ F, RegSet regs, int r, int inner,
                                                    double *__restrict__ diag_tile) {
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= n_tiles) return;
    const int slab = __builtin_amdgcn_readfirstlane(tile_slab[tile]);
    const int row0 = __builtin_amdgcn_readfirstlane(tile_row0[tile]);
    const int nrows = __builtin_amdgcn_readfirstlane(tile_nrows[tile]);
    const bool valid = lane < nrows;
    const long j = (long)row0 + (valid ? lane : 0);
    const float rho = rho_arr[slab];
    const float *__restrict__ Li = Linv + (long)slab * r * r;
    const bool vec = (RP >= 4) && (r == RP);

    float rhs[RP];
    if (vec) {
#pragma unroll
        for (int c = 0; c < RP; c += 4) {
            const float4 v = *reinterpret_cast<const float4 *>(rhs_src + j * r + c);
            rhs[c] = v.x, rhs[c + 1] = v.y, rhs[c + 2] = v.z, rhs[c + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int c = 0; c < RP; ++c) rhs[c] = (c < r) ? rhs_src[j * r + c] : 0.f;
    }
    if (Arows != nullptr) {
        const float *__restrict__ a = Arows + (long)slab * r;
#pragma unroll
        for (int c = 0; c < RP; ++c)
            if (c < r) rhs[c] *= a[c];
    }

    constexpr int NR = NREG > 0 ? NREG : 1;
    float z[NR][RP], u[NR][RP], thr[NR];
#pragma unroll
    for (int k = 0; k < NREG; ++k) {
        thr[k] = regs.p0[k] / rho;
        if (vec) {
#pragma unroll
            for (int c = 0; c < RP; c += 4) {
                const float4 a = *reinterpret_cast<const float4 *>(regs.aux[k] + j * r + c);
                const float4 d = *reinterpret_cast<const float4 *>(regs.dual[k] + j * r + c);
                z[k][c] = a.x, z[k][c + 1] = a.y, z[k][c + 2] = a.z, z[k][c + 3] = a.w;
                u[k][c] = d.x, u[k][c + 1] = d.y, u[k][c + 2] = d.z, u[k][c + 3] = d.w;
            }
        } else {
#pragma unroll
            for (int c = 0; c < RP; ++c) {
                z[k][c] = (c < r) ? regs.aux[k][j * r + c] : 0.f;
                u[k][c] = (c < r) ? regs.dual[k][j * r + c] : 0.f;
            }
        }
    }

    float f[RP];
    const int n_it = (NREG == 0 && inner > 1) ? 1 : inner;  // without penalties every inner solve is identical
    for (int it = 0; it < n_it; ++it) {
        float t[RP];
#pragma unroll
        for (int c = 0; c < RP; ++c) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < NREG; ++k) s += z[k][c] - u[k][c];
            t[c] = (NREG > 0) ? fmaf(rho, s, rhs[c]) : rhs[c];
            f[c] = 0.f;
        }
#pragma unroll
        for (int c = 0; c < RP; ++c) {
            if (c < r) {
#pragma unroll
                for (int d = 0; d < RP; ++d)
                    if (d < r) f[d] = fmaf(t[c], Li[c * r + d], f[d]);
            }
        }
#pragma unroll
        for (int k = 0; k < NREG; ++k) {
            const int kind = regs.kind[k], nn = regs.nonneg[k];
            const float p0 = regs.p0[k], p1 = regs.p1[k];
#pragma unroll
            for (int c = 0; c < RP; ++c) {
                const float y = f[c] + u[k][c];
                const float zn = prox_elem(kind, nn, p0, p1, thr[k], y);
                u[k][c] = f[c] - (zn - u[k][c]);
                z[k][c] = zn;
            }
        }
    }

    if (valid) {
        if (vec) {
#pragma unroll
            for (int c = 0; c < RP; c += 4) {
                *reinterpret_cast<float4 *>(F + j * r + c) = make_float4(f[c], f[c + 1], f[c + 2], f[c + 3]);
#pragma unroll
                for (int k = 0; k < NREG; ++k) {
                    *reinterpret_cast<float4 *>(regs.aux[k] + j * r + c) =
                        make_float4(z[k][c], z[k][c + 1], z[k][c + 2], z[k][c + 3]);
                    *reinterpret_cast<float4 *>(regs.dual[k] + j * r + c) =
                        make_float4(u[k][c], u[k][c + 1], u[k][c + 2], u[k][c + 3]);
                }
            }
        } else {
#pragma unroll
            for (int c = 0; c < RP; ++c) {
                if (c < r) {
                    F[j * r + c] = f[c];
#pragma unroll
                    for (int k = 0; k < NREG; ++k) {
                        regs.aux[k][j * r + c] = z[k][c];
                        regs.dual[k][j * r + c] = u[k][c];
                    }
                }
            }
        }
    }
    // per-tile diagnostics
    double nf = 0.0, na = 0.0, gap[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) gap[k] = 0.0;
    if (valid) {
#pragma unroll
        for (int c = 0; c < RP; ++c) {
            if (c < r) {
                nf += (double)f[c] * (double)f[c];
                na += fabs((double)f[c]);
#pragma unroll
                for (int k = 0; k < NREG; ++k) {
                    const double dlt = (double)z[k][c] - (double)f[c];
                    gap[k] += dlt * dlt;
                }
            }
        }
    }
    nf = wave_sum(nf);
    na = wave_sum(na);
#pragma unroll
    for (int k = 0; k < NREG; ++k) gap[k] = wave_sum(gap[k]);
    if (lane == 0) {
        double *o = diag_tile + (long)tile * DIAG_COLS;
        o[0] = nf;
        o[1] = na;
#pragma unroll
        for (int k = 0; k < NREG; ++k) o[2 + k] = gap[k];
    }
}

// ---------------------------------------------------------------------------------------------------------
// A-phase finish, one wave per slab i (decomposition.py:155-219):
//   Q_i = BtB_i o CtC (stored as cross_products), rho_i, L_i^-1 (fp64), fused inner loop on row a_i with
//   row-separable penalties, and the per-slab terms of the fast reconstruction-error formula (:445-449).
// fused_inner == 0: only Q_i, rho_i and L_i^-1 are produced (generic inner loop follows).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_A_finish(const float *__restrict__ rhsA, float *__restrict__ BtB,
                                                 const float *__restrict__ CtC, int r, float scale, float l2,
                                                 int constant, const float *__restrict__ rho_max,
                                                 float *__restrict__ rhoA, float *__restrict__ LinvA,
                                                 float *__restrict__ A, RegSet regs, int inner, int fused_inner,
                                                 double *__restrict__ e1, double *__restrict__ diag_row) {
    extern __shared__ double smd[];
    double *M = smd, *colp = smd + r * r, *Q = colp + r, *tS = Q + r * r;
    const int i = blockIdx.x, lane = threadIdx.x;
    for (int e = lane; e < r * r; e += 64) {
        const double qv = (double)BtB[(long)i * r * r + e] * (double)CtC[e];
        Q[e] = qv;
        BtB[(long)i * r * r + e] = (float)qv;
    }
    __syncthreads();
    double tr = 0.0;
    for (int c = 0; c < r; ++c) tr += Q[c * r + c];
    float rho = (float)(0.5 * tr * scale);
    if (constant) rho = rho_max[1];
    const int n = regs.n;
    const double shift = (double)rho * n + (double)l2;
    for (int e = lane; e < r * r; e += 64) {
        const int c = e / r, d = e - c * r;
        M[e] = Q[e] + (c == d ? shift : 0.0);
    }
    gj_inverse(M, colp, r, lane);
    if (lane == 0) rhoA[i] = rho;
    if (!fused_inner) {
        for (int e = lane; e < r * r; e += 64) LinvA[(long)i * r * r + e] = (float)M[e];
        return;
    }
    const bool act = lane < r;
    const int c = act ? lane : 0;
    const float rhs = rhsA[(long)i * r + c];
    float z[MCL_MAX_REGS], u[MCL_MAX_REGS], thr[MCL_MAX_REGS];
#pragma unroll
    for (int k = 0; k < MCL_MAX_REGS; ++k) {
        z[k] = u[k] = thr[k] = 0.f;
        if (k < n) {
            z[k] = regs.aux[k][(long)i * r + c];
            u[k] = regs.dual[k][(long)i * r + c];
            thr[k] = regs.p0[k] / rho;
        }
    }
    float a = A[(long)i * r + c];
    const int n_it = (n == 0 && inner > 1) ? 1 : inner;
    for (int it = 0; it < n_it; ++it) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < MCL_MAX_REGS; ++k)
            if (k < n) s += z[k] - u[k];
        __syncthreads();
        if (act) tS[c] = (double)((n > 0) ? fmaf(rho, s, rhs) : rhs);
        __syncthreads();
        double acc = 0.0;
        for (int d = 0; d < r; ++d) acc += tS[d] * M[d * r + c];
        a = (float)acc;
#pragma unroll
        for (int k = 0; k < MCL_MAX_REGS; ++k) {
            if (k < n) {
                const float y = a + u[k];
                const float zn = prox_elem(regs.kind[k], regs.nonneg[k], regs.p0[k], regs.p1[k], thr[k], y);
                u[k] = a - (zn - u[k]);
                z[k] = zn;
            }
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
    // <X_i, M_i> = rhs_i . a_i ;  ||M_i||^2 = a_i^T Q_i a_i
    __syncthreads();
    if (act) tS[c] = (double)a;
    __syncthreads();
    double qa = 0.0;
    for (int d = 0; d < r; ++d) qa += Q[c * r + d] * tS[d];
    double inner_i = act ? (double)rhs * (double)a : 0.0;
    double model_i = act ? (double)a * qa : 0.0;
    double nf = act ? (double)a * (double)a : 0.0, na = act ? fabs((double)a) : 0.0;
    inner_i = wave_sum(inner_i);
    model_i = wave_sum(model_i);
    nf = wave_sum(nf);
    na = wave_sum(na);
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
}

// rho_i of the A-phase alone (needed before the systems when the feasibility penalty is constant)
__global__ void k_A_rho(const float *__restrict__ BtB, const float *__restrict__ CtC, int I, int r, float scale,
                        float *__restrict__ rhoA, float *__restrict__ rho_max) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= I) return;
    double s = 0.0;
    for (int c = 0; c < r; ++c) s += (double)BtB[((long)i * r + c) * r + c] * (double)CtC[c * r + c];
    const float rho = (float)(0.5 * s * scale);
    rhoA[i] = rho;
    atomicMax(reinterpret_cast<int *>(rho_max + 1), __float_as_int(rho));
}

// e1 / row diagnostics of mode 0 from (rhsA, BtB or Q, A) - used when the A-phase inner loop was not fused
// or when the by-products are recomputed for an error evaluation without an A update (decomposition.py:430-444)
__global__ __launch_bounds__(64) void k_A_e1(const float *__restrict__ rhsA, const float *__restrict__ BtB,
                                             const float *__restrict__ CtC, int btb_is_q, const float *__restrict__ A,
                                             RegSet regs, int r, double *__restrict__ e1,
                                             double *__restrict__ diag_row) {
    const int i = blockIdx.x, lane = threadIdx.x;
    const bool act = lane < r;
    const int c = act ? lane : 0;
    const float a = A[(long)i * r + c];
    double qa = 0.0;
    for (int d = 0; d < r; ++d) {
        double q = (double)BtB[((long)i * r + c) * r + d];
        if (!btb_is_q) q *= (double)CtC[c * r + d];
        qa += q * (double)A[(long)i * r + d];
    }
    const double inner_i = wave_sum(act ? (double)rhsA[(long)i * r + c] * (double)a : 0.0);
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
// diagnostics: ||X||^2 (once) and the final reduction of the per-tile / per-slab partial sums
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

// Column sums of a [n_rows, ncols] fp64 table (ncols <= 8): block b reduces column b.
__global__ __launch_bounds__(256) void k_colsum_table(const double *__restrict__ tab, int n_rows, int ncols,
                                                      double *__restrict__ out, int out_stride_is_one) {
    __shared__ double sm[4];
    const int col = blockIdx.x;
    double s = 0.0;
    for (int e = threadIdx.x; e < n_rows; e += 256) s += tab[(long)e * ncols + col];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[col] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

// Assemble the MCL_DIAG_LEN vector from the reduced tables (one thread; trivial)
__global__ void k_diag_assemble(const double *__restrict__ sA, const double *__restrict__ sB,
                                const double *__restrict__ sC, const double *__restrict__ sE,
                                const double *__restrict__ xsq, int nA, int nB, int nC, int include_replicated,
                                double *__restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    for (int e = 0; e < MCL_DIAG_LEN; ++e) out[e] = 0.0;
    out[MCL_DIAG_NORM_SQ + 0] = sA[0];
    out[MCL_DIAG_NORM_SQ + 1] = sB[0];
    out[MCL_DIAG_INNER] = sE[0];
    out[MCL_DIAG_MODEL_SQ] = sE[1];
    out[MCL_DIAG_X_SQ] = xsq[0];
    for (int k = 0; k < nA; ++k) {
        out[MCL_DIAG_REG + (0 * MCL_MAX_REGS + k) * 2] = sA[2 + k];
        out[MCL_DIAG_REG + (0 * MCL_MAX_REGS + k) * 2 + 1] = sA[1];
    }
    for (int k = 0; k < nB; ++k) {
        out[MCL_DIAG_REG + (1 * MCL_MAX_REGS + k) * 2] = sB[2 + k];
        out[MCL_DIAG_REG + (1 * MCL_MAX_REGS + k) * 2 + 1] = sB[1];
    }
    if (include_replicated) {
        out[MCL_DIAG_NORM_SQ + 2] = sC[0];
        for (int k = 0; k < nC; ++k) {
            out[MCL_DIAG_REG + (2 * MCL_MAX_REGS + k) * 2] = sC[2 + k];
            out[MCL_DIAG_REG + (2 * MCL_MAX_REGS + k) * 2 + 1] = sC[1];
        }
    }
}

// Per-tile diagnostics of a packed factor from memory (generic path / initial state):
// ||F||^2, sum|F|, ||Z_k - F||^2 with Z_k = aux_k or P Delta (PARAFAC2).
__global__ __launch_bounds__(256) void k_rows_diag(const int *__restrict__ tile_row0, const int *__restrict__ tile_nrows,
                                                   int n_tiles, const float *__restrict__ F, RegSet regs, int r,
                                                   double *__restrict__ diag_tile) {
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= n_tiles) return;
    const int row0 = tile_row0[tile], nrows = tile_nrows[tile];
    const bool valid = lane < nrows;
    const long j = (long)row0 + (valid ? lane : 0);
    double nf = 0.0, na = 0.0, gap[MCL_MAX_REGS];
#pragma unroll
    for (int k = 0; k < MCL_MAX_REGS; ++k) gap[k] = 0.0;
    if (valid) {
        for (int c = 0; c < r; ++c) {
            const double f = F[j * r + c];
            nf += f * f;
            na += fabs(f);
#pragma unroll
            for (int k = 0; k < MCL_MAX_REGS; ++k) {
                if (k < regs.n) {
                    double zv;
                    if (regs.kind[k] == MCL_PEN_PARAFAC2) {
                        float acc = 0.f;
                        for (int d = 0; d < r; ++d) acc = fmaf(regs.aux[k][j * r + d], regs.aux2[k][d * r + c], acc);
                        zv = acc;
                    } else {
                        zv = regs.aux[k][j * r + c];
                    }
                    gap[k] += (zv - f) * (zv - f);
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

int mcl_launch_ctc(mcl_context *c) {
    hipLaunchKernelGGL(k_ctc, dim3(1), dim3(256), 32768, c->stream, c->C, (int)c->K, c->r, c->CtC);
    MCL_CHECK_HIP(c, hipGetLastError());
    return 0;
}

int mcl_launch_B_rho(mcl_context *c) {
    MCL_CHECK_HIP(c, hipMemsetAsync(c->rho_max, 0, sizeof(float), c->stream));
    if (c->I == 0) return 0;
    hipLaunchKernelGGL(k_B_rho, dim3((unsigned)((c->I + 255) / 256)), dim3(256), 0, c->stream, c->CtC, c->A, (int)c->I,
                       c->r, (float)c->opt.feasibility_penalty_scale, c->rhoB, c->rho_max);
    MCL_CHECK_HIP(c, hipGetLastError());
    return 0;
}

int mcl_launch_B_systems(mcl_context *c) {
    if (c->I == 0) return 0;
    const size_t sm = sizeof(double) * (size_t)(c->r * c->r + c->r);
    hipLaunchKernelGGL(k_B_systems, dim3((unsigned)c->I), dim3(64), sm, c->stream, c->CtC, c->A, c->r,
                       (float)c->opt.feasibility_penalty_scale, (float)c->opt.l2_penalty[1], c->regs[1].n,
                       c->opt.constant_B, c->rho_max, c->rhoB, c->LinvB);
    MCL_CHECK_HIP(c, hipGetLastError());
    return 0;
}

int mcl_launch_C_prepare(mcl_context *c) {
    const size_t sm = sizeof(double) * (size_t)(c->r * c->r + c->r);
    hipLaunchKernelGGL(k_C_prepare, dim3(1), dim3(64), sm, c->stream, c->GR, c->r,
                       (float)c->opt.feasibility_penalty_scale, (float)c->opt.l2_penalty[2], c->regs[2].n, c->rhoC,
                       c->LinvC);
    MCL_CHECK_HIP(c, hipGetLastError());
    return 0;
}

template <int RP, int NREG>
static void launch_rows_fused_t(mcl_context *c, const TileMap &tm, const float *rhs, const float *Arows,
                                const float *rho, const float *Linv, float *F, const RegSet &rs, double *diag) {
    hipLaunchKernelGGL((k_rows_fused<RP, NREG>), dim3((unsigned)((tm.n_tiles + 3) / 4)), dim3(256), 0, c->stream,
                       tm.slab, tm.row0, tm.nrows, tm.n_tiles, rhs, Arows, rho, Linv, F, rs, c->r,
                       c->opt.inner_n_iter_max, diag);
}

template <int RP>
static int launch_rows_fused_n(mcl_context *c, const TileMap &tm, const float *rhs, const float *Arows,
                               const float *rho, const float *Linv, float *F, const RegSet &rs, double *diag) {
    switch (rs.n) {
        case 0: launch_rows_fused_t<RP, 0>(c, tm, rhs, Arows, rho, Linv, F, rs, diag); return 0;
        case 1: launch_rows_fused_t<RP, 1>(c, tm, rhs, Arows, rho, Linv, F, rs, diag); return 0;
        case 2:
            if (RP <= 32) {
                launch_rows_fused_t<(RP <= 32 ? RP : 32), 2>(c, tm, rhs, Arows, rho, Linv, F, rs, diag);
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
    const float *rhs = (mode == 1) ? c->XC : c->GR + (long)c->r * c->r;
    const float *Arows = (mode == 1) ? c->A : nullptr;
    const float *rho = (mode == 1) ? c->rhoB : c->rhoC;
    const float *Linv = (mode == 1) ? c->LinvB : c->LinvC;
    float *F = (mode == 1) ? c->B : c->C;
    switch (c->RP) {
        case 4: return launch_rows_fused_n<4>(c, tm, rhs, Arows, rho, Linv, F, rs, diag);
        case 8: return launch_rows_fused_n<8>(c, tm, rhs, Arows, rho, Linv, F, rs, diag);
        case 16: return launch_rows_fused_n<16>(c, tm, rhs, Arows, rho, Linv, F, rs, diag);
        case 32: return launch_rows_fused_n<32>(c, tm, rhs, Arows, rho, Linv, F, rs, diag);
        default: return launch_rows_fused_n<64>(c, tm, rhs, Arows, rho, Linv, F, rs, diag);
    }
}

int mcl_launch_rows_fused(mcl_context *c, int mode) {
    double *diag = (mode == 1) ? c->diagB_tile : c->diagC_tile;
    const int rc = mcl_rows_fused_dispatch(c, mode, diag);
    if (rc < 0) return rc;
    MCL_CHECK_HIP(c, hipGetLastError());
    return 0;
}

int mcl_launch_A_rho(mcl_context *c) {
    MCL_CHECK_HIP(c, hipMemsetAsync(c->rho_max + 1, 0, sizeof(float), c->stream));
    if (c->I == 0) return 0;
    hipLaunchKernelGGL(k_A_rho, dim3((unsigned)((c->I + 255) / 256)), dim3(256), 0, c->stream, c->BtB, c->CtC,
                       (int)c->I, c->r, (float)c->opt.feasibility_penalty_scale, c->rhoA, c->rho_max);
    MCL_CHECK_HIP(c, hipGetLastError());
    return 0;
}

int mcl_launch_A_finish(mcl_context *c, bool fused_inner) {
    if (c->I == 0) return 0;
    const size_t sm = sizeof(double) * (size_t)(2 * c->r * c->r + 2 * c->r);
    hipLaunchKernelGGL(k_A_finish, dim3((unsigned)c->I), dim3(64), sm, c->stream, c->rhsA, c->BtB, c->CtC, c->r,
                       (float)c->opt.feasibility_penalty_scale, (float)c->opt.l2_penalty[0], c->opt.constant_A,
                       c->rho_max, c->rhoA, c->LinvA, c->A, c->regs[0], c->opt.inner_n_iter_max, fused_inner ? 1 : 0,
                       c->e1, c->diagA_row);
    MCL_CHECK_HIP(c, hipGetLastError());
    return 0;
}

int mcl_launch_A_e1(mcl_context *c, bool btb_is_q) {
    if (c->I == 0) return 0;
    hipLaunchKernelGGL(k_A_e1, dim3((unsigned)c->I), dim3(64), 0, c->stream, c->rhsA, c->BtB, c->CtC, btb_is_q ? 1 : 0,
                       c->A, c->regs[0], c->r, c->e1, c->diagA_row);
    MCL_CHECK_HIP(c, hipGetLastError());
    return 0;
}

int mcl_launch_rows_diag(mcl_context *c, int mode) {
    const TileMap &tm = (mode == 1) ? c->tilesB : (mode == 2 ? c->tilesC : c->tilesA);
    if (tm.n_tiles == 0) return 0;
    const float *F = (mode == 1) ? c->B : (mode == 2 ? c->C : c->A);
    double *diag = (mode == 1) ? c->diagB_tile : (mode == 2 ? c->diagC_tile : c->diagA_tile);
    hipLaunchKernelGGL(k_rows_diag, dim3((unsigned)((tm.n_tiles + 3) / 4)), dim3(256), 0, c->stream, tm.row0, tm.nrows,
                       tm.n_tiles, F, c->regs[mode], c->r, diag);
    MCL_CHECK_HIP(c, hipGetLastError());
    return 0;
}

int mcl_launch_x_sq(mcl_context *c) {
    const long n = (long)c->N * c->K;
    const int nb = 1024;
    hipLaunchKernelGGL(k_sumsq_partial, dim3(nb), dim3(256), 0, c->stream, c->X, n, c->xsq_part);
    hipLaunchKernelGGL(k_sum_doubles, dim3(1), dim3(256), 0, c->stream, c->xsq_part, nb, c->x_sq);
    MCL_CHECK_HIP(c, hipGetLastError());
    return 0;
}

// Final reduction: column sums of the per-tile/per-row tables, then assembly of the MCL_DIAG_LEN vector.
// a_rows_table: mode-0 sums come from diagA_row [I] (A-phase kernels) or diagA_tile (k_rows_diag).
int mcl_launch_diag_final(mcl_context *c, double *out, int include_replicated, bool a_from_rows) {
    double *s = c->diag_sums;  // [3*DIAG_COLS + 2]
    MCL_CHECK_HIP(c, hipMemsetAsync(s, 0, sizeof(double) * (3 * DIAG_COLS + 2), c->stream));
    const int nA = a_from_rows ? (int)c->I : c->tilesA.n_tiles;
    if (nA > 0)
        hipLaunchKernelGGL(k_colsum_table, dim3(DIAG_COLS), dim3(256), 0, c->stream,
                           a_from_rows ? c->diagA_row : c->diagA_tile, nA, DIAG_COLS, s, 1);
    if (c->tilesB.n_tiles > 0)
        hipLaunchKernelGGL(k_colsum_table, dim3(DIAG_COLS), dim3(256), 0, c->stream, c->diagB_tile, c->tilesB.n_tiles,
                           DIAG_COLS, s + DIAG_COLS, 1);
    if (c->tilesC.n_tiles > 0)
        hipLaunchKernelGGL(k_colsum_table, dim3(DIAG_COLS), dim3(256), 0, c->stream, c->diagC_tile, c->tilesC.n_tiles,
                           DIAG_COLS, s + 2 * DIAG_COLS, 1);
    if (c->I > 0)
        hipLaunchKernelGGL(k_colsum_table, dim3(2), dim3(256), 0, c->stream, c->e1, (int)c->I, 2, s + 3 * DIAG_COLS, 1);
    hipLaunchKernelGGL(k_diag_assemble, dim3(1), dim3(64), 0, c->stream, s, s + DIAG_COLS, s + 2 * DIAG_COLS,
                       s + 3 * DIAG_COLS, c->x_sq, c->regs[0].n, c->regs[1].n, c->regs[2].n, include_replicated, out);
    MCL_CHECK_HIP(c, hipGetLastError());
    return 0;
}
