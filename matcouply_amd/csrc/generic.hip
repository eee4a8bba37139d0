// Un-fused inner ADMM loop (one kernel per step) for penalties that couple the rows of a slab or the slabs:
//   L2Ball       penalties.py:920-925    column norms over the rows of one slab
//   Unimodality  penalties.py:1014-1015, _unimodal_regression.py:27-104   per-column prefix isotonic regression (fp64)
//   Parafac2     penalties.py:1224-1250, 1280-1281   polar factors per slab + cross-slab coordinate matrix
// plus the row-separable ones when they share a mode with the above.  Used for all three modes: A and C are
// treated as a single slab (tile maps tilesA / tilesC, extents ext_A / ext_C).
#include <algorithm>

#include "mcl_internal.h"

static __device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

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

struct ModeView {
    const int *tile_slab, *tile_row0, *tile_nrows;
    int n_tiles;
    const int *ext;        // slab extents (row_ptr)
    int n_slabs;
    const float *rho;      // [n_slabs]
    float *F;              // factor [rows, r]
};

static ModeView view_of(mcl_context *c, int mode) {
    ModeView v{};
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
// solve:  F = (rhs (o a) + rho sum_k (Z_k - U_k)) L^-1   for every row of a slab-tiled matrix (modes 1, 2)
// ---------------------------------------------------------------------------------------------------------
template <int RP>
__global__ __launch_bounds__(256) void k_rows_solve(ModeView mv, const float *__restrict__ rhs_src,
                                                    const float *__restrict__ Arows, const float *__restrict__ Linv,
                                                    RegSet regs, int r) {
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= mv.n_tiles) return;
    const int slab = __builtin_amdgcn_readfirstlane(mv.tile_slab[tile]);
    const int row0 = __builtin_amdgcn_readfirstlane(mv.tile_row0[tile]);
    const int nrows = __builtin_amdgcn_readfirstlane(mv.tile_nrows[tile]);
    const bool valid = lane < nrows;
    const long j = (long)row0 + (valid ? lane : 0);
    const float rho = mv.rho[slab];
    const float *__restrict__ Li = Linv + (long)slab * r * r;
    float t[RP], f[RP];
#pragma unroll
    for (int c = 0; c < RP; ++c) {
        float v = 0.f;
        if (c < r) {
            v = rhs_src[j * r + c];
            if (Arows) v *= Arows[(long)slab * r + c];
        }
        t[c] = v;
        f[c] = 0.f;
    }
    for (int k = 0; k < regs.n; ++k) {
        if (regs.kind[k] == MCL_PEN_PARAFAC2) {
            float p[RP];
#pragma unroll
            for (int c = 0; c < RP; ++c) p[c] = (c < r) ? regs.aux[k][j * r + c] : 0.f;
#pragma unroll
            for (int c = 0; c < RP; ++c) {
                if (c < r) {
                    float z = 0.f;
#pragma unroll
                    for (int d = 0; d < RP; ++d)
                        if (d < r) z = fmaf(p[d], regs.aux2[k][d * r + c], z);
                    t[c] = fmaf(rho, z - regs.dual[k][j * r + c], t[c]);
                }
            }
        } else {
#pragma unroll
            for (int c = 0; c < RP; ++c)
                if (c < r) t[c] = fmaf(rho, regs.aux[k][j * r + c] - regs.dual[k][j * r + c], t[c]);
        }
    }
#pragma unroll
    for (int c = 0; c < RP; ++c) {
        if (c < r) {
#pragma unroll
            for (int d = 0; d < RP; ++d)
                if (d < r) f[d] = fmaf(t[c], Li[c * r + d], f[d]);
        }
    }
    if (valid) {
#pragma unroll
        for (int c = 0; c < RP; ++c)
            if (c < r) mv.F[j * r + c] = f[c];
    }
}

// mode 0: every row has its own system (decomposition.py:184-195); one wave per row
__global__ __launch_bounds__(64) void k_A_rows_solve(const float *__restrict__ rhsA, const float *__restrict__ rhoA,
                                                     const float *__restrict__ LinvA, float *__restrict__ A,
                                                     RegSet regs, int r) {
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
    if (act) A[(long)i * r + c] = a;
}

// ---------------------------------------------------------------------------------------------------------
// row-separable prox + dual update of penalty k (generic path)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rows_prox_rowsep(ModeView mv, RegSet regs, int k, int r) {
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= mv.n_tiles) return;
    const int slab = mv.tile_slab[tile];
    const int nrows = mv.tile_nrows[tile];
    if (lane >= nrows) return;
    const long j = (long)mv.tile_row0[tile] + lane;
    const float rho = mv.rho[slab];
    const float thr = regs.p0[k] / rho;
    for (int c = 0; c < r; ++c) {
        const float f = mv.F[j * r + c], u = regs.dual[k][j * r + c];
        const float z = prox_elem_g(regs.kind[k], regs.nonneg[k], regs.p0[k], regs.p1[k], thr, f + u);
        regs.aux[k][j * r + c] = z;
        regs.dual[k][j * r + c] = f - (z - u);
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

__global__ __launch_bounds__(256) void k_rows_l2ball(ModeView mv, RegSet regs, int k, int r,
                                                     const double *__restrict__ colsq) {
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= mv.n_tiles) return;
    const int slab = mv.tile_slab[tile];
    if (lane >= mv.tile_nrows[tile]) return;
    const long j = (long)mv.tile_row0[tile] + lane;
    const float bound = regs.p0[k];
    for (int c = 0; c < r; ++c) {
        const float f = mv.F[j * r + c], u = regs.dual[k][j * r + c];
        float y = f + u;
        if (regs.nonneg[k]) y = fmaxf(y, 0.f);
        const float nrm = (float)sqrt(colsq[(long)slab * r + c]);
        const float z = y * bound / fmaxf(nrm, bound);
        regs.aux[k][j * r + c] = z;
        regs.dual[k][j * r + c] = f - (z - u);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Unimodality: one thread per (slab, column); prefix isotonic regression in both directions in fp64
// (the projection is discontinuous in its split index, so the arithmetic is kept in double).
// Scratch arrays are column-interleaved (index * r + col) so the r threads of a slab access them coalesced.
// ---------------------------------------------------------------------------------------------------------
struct UniScratch {
    double *lvL, *lvR, *eL, *eR, *sy, *sy2, *sw, *cum2;
    int *stL, *stR;
};

static __device__ void prefix_isotonic_dev(const float *__restrict__ F, const float *__restrict__ U, long base, long step,
                                           int n, int r, int col, int nonneg, double *level, int *start, double *err,
                                           double *sy, double *sy2, double *sw, double *cum2, long sbase, long ebase) {
    // element idx of this pass lives at packed row (base + idx*step); scratch row sbase+idx; err row ebase+idx
    err[ebase * r + col] = 0.0;
    double run2 = 0.0;
    for (int i = 0; i < n; ++i) {
        const long src = (base + (long)i * step) * r + col;
        const double v = (double)(F[src] + U[src]);
        run2 += v * v;
        const long si = (sbase + i) * r + col;
        cum2[si] = run2;
        double csy = v, csy2 = v * v, csw = 1.0, lev = v;
        int st = i;
        while (st != 0 && lev <= level[(sbase + st - 1) * r + col]) {
            const long p = (sbase + st - 1) * r + col;
            csy += sy[p];
            csy2 += sy2[p];
            csw += sw[p];
            lev = csy / csw;
            st = start[p];
        }
        sy[si] = csy, sy2[si] = csy2, sw[si] = csw;
        level[si] = lev;
        start[si] = st;
        if (nonneg && lev < 0.0)
            err[(ebase + i + 1) * r + col] = run2;
        else
            err[(ebase + i + 1) * r + col] = (csy2 - csy * csy / csw) + err[(ebase + st) * r + col];
    }
}

__global__ __launch_bounds__(64) void k_slab_unimodal(const int *__restrict__ ext, int n_slabs, float *__restrict__ F,
                                                      RegSet regs, int k, int r, UniScratch sc) {
    const long t = (long)blockIdx.x * 64 + threadIdx.x;
    if (t >= (long)n_slabs * r) return;
    const int slab = (int)(t / r), col = (int)(t - (long)slab * r);
    const int s = ext[slab], e = ext[slab + 1], n = e - s;
    if (n <= 0) return;
    const int nonneg = regs.nonneg[k];
    float *__restrict__ Z = regs.aux[k];
    float *__restrict__ U = regs.dual[k];
    const long eb = (long)s + slab;  // n+1 error entries per slab
    // left pass: elements s .. e-1 ; right pass: elements e-1 .. s
    prefix_isotonic_dev(F, U, s, 1, n, r, col, nonneg, sc.lvL, sc.stL, sc.eL, sc.sy, sc.sy2, sc.sw, sc.cum2, s, eb);
    prefix_isotonic_dev(F, U, (long)e - 1, -1, n, r, col, nonneg, sc.lvR, sc.stR, sc.eR, sc.sy, sc.sy2, sc.sw, sc.cum2, s, eb);
    double best = sc.eR[(eb + n) * r + col];
    int split = 0;
    for (int i = 0; i <= n; ++i) {
        const double err = sc.eL[(eb + i) * r + col] + sc.eR[(eb + n - i) * r + col];
        if (err < best) best = err, split = i;
    }
    double *out = sc.sy;  // reuse as the projected column
    for (int idx = split - 1; idx >= 0;) {
        const long si = ((long)s + idx) * r + col;
        double lev = sc.lvL[si];
        if (nonneg && lev < 0.0) lev = 0.0;
        const int st = sc.stL[si];
        for (int q = st; q <= idx; ++q) out[((long)s + q) * r + col] = lev;
        idx = st - 1;
    }
    for (int idx = n - split - 1; idx >= 0;) {
        const long si = ((long)s + idx) * r + col;
        double lev = sc.lvR[si];
        if (nonneg && lev < 0.0) lev = 0.0;
        const int st = sc.stR[si];
        for (int q = st; q <= idx; ++q) out[((long)s + (n - 1 - q)) * r + col] = lev;
        idx = st - 1;
    }
    for (int q = 0; q < n; ++q) {
        const long g = ((long)s + q) * r + col;
        const float z = (float)out[g];
        const float f = F[g], u = U[g];
        Z[g] = z;
        U[g] = f - (z - u);
    }
}

// ---------------------------------------------------------------------------------------------------------
// PARAFAC2 prox (mode 1):  Y_i = B_i + U_i,  P_i = polar(Y_i Delta^T),  Delta <- sum rho_i P_i^T Y_i / sum rho_i
// Gram route in fp64:  S_i = Y_i^T Y_i,  G_i = Delta S_i Delta^T = V L V^T,  W_i = V L^-1/2 V^T,
//                      T_i = Delta^T W_i,  P_i = Y_i T_i,  P_i^T Y_i = T_i^T S_i.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pf2_gram(const int *__restrict__ ext, const float *__restrict__ F,
                                                  const float *__restrict__ U, int r, double *__restrict__ S) {
    extern __shared__ float ytile[];  // [64, r]
    const int slab = blockIdx.x;
    const int s = ext[slab], e = ext[slab + 1];
    const int npairs = r * r;
    double acc[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] = 0.0;
    for (int j0 = s; j0 < e; j0 += 64) {
        const int rows = min(64, e - j0);
        __syncthreads();
        for (int q = threadIdx.x; q < rows * r; q += 256) ytile[q] = F[(long)j0 * r + q] + U[(long)j0 * r + q];
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int pr = threadIdx.x + 256 * t;
            if (pr < npairs) {
                const int a = pr / r, b = pr - a * r;
                double sum = 0.0;
                for (int q = 0; q < rows; ++q) sum += (double)ytile[q * r + a] * (double)ytile[q * r + b];
                acc[t] += sum;
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int pr = threadIdx.x + 256 * t;
        if (pr < npairs) S[(long)slab * npairs + pr] = acc[t];
    }
}

__global__ __launch_bounds__(64) void k_pf2_algebra(const double *__restrict__ S, const float *__restrict__ Delta,
                                                    const float *__restrict__ rho, int r, float *__restrict__ T,
                                                    double *__restrict__ acc_out) {
    extern __shared__ double smd[];
    double *Sm = smd, *G = smd + r * r, *V = G + r * r, *lam = V + r * r, *D = lam + r;  // D: Delta in fp64 [r*r]
    const int slab = blockIdx.x, lane = threadIdx.x;
    const int n2 = r * r;
    for (int e = lane; e < n2; e += 64) {
        Sm[e] = S[(long)slab * n2 + e];
        D[e] = (double)Delta[e];
        V[e] = ((e / r) == (e % r)) ? 1.0 : 0.0;
    }
    __syncthreads();
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
    __syncthreads();
    // cyclic Jacobi eigen-decomposition of the symmetric G (fp64); V accumulates the eigenvectors
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
        for (int p = 0; p < r - 1; ++p) {
            for (int q = p + 1; q < r; ++q) {
                __syncthreads();
                const double apq = G[p * r + q], app = G[p * r + p], aqq = G[q * r + q];
                if (fabs(apq) > 1e-300) {
                    const double tau = (aqq - app) / (2.0 * apq);
                    const double tt = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                    const double cs = 1.0 / sqrt(1.0 + tt * tt), sn = tt * cs;
                    double gkp = 0.0, gkq = 0.0, vkp = 0.0, vkq = 0.0;
                    const int k = lane;
                    if (k < r) {
                        gkp = G[k * r + p], gkq = G[k * r + q];
                        vkp = V[k * r + p], vkq = V[k * r + q];
                    }
                    __syncthreads();
                    if (k < r) {
                        V[k * r + p] = cs * vkp - sn * vkq;
                        V[k * r + q] = sn * vkp + cs * vkq;
                        if (k != p && k != q) {
                            const double np = cs * gkp - sn * gkq, nq = sn * gkp + cs * gkq;
                            G[k * r + p] = np, G[p * r + k] = np;
                            G[k * r + q] = nq, G[q * r + k] = nq;
                        }
                    }
                    if (lane == 0) {
                        G[p * r + p] = app - tt * apq;
                        G[q * r + q] = aqq + tt * apq;
                        G[p * r + q] = 0.0;
                        G[q * r + p] = 0.0;
                    }
                }
            }
        }
        __syncthreads();
    }
    __syncthreads();
    double lmax = 0.0;
    for (int k = 0; k < r; ++k) lmax = fmax(lmax, G[k * r + k]);
    if (lane < r) {
        const double l = G[lane * r + lane];
        lam[lane] = (l > 1e-14 * lmax && l > 0.0) ? 1.0 / sqrt(l) : 0.0;  // pseudo-inverse square root
    }
    __syncthreads();
    // W = V diag(lam) V^T -> G
    for (int e = lane; e < n2; e += 64) {
        const int a = e / r, b = e - a * r;
        double sum = 0.0;
        for (int k = 0; k < r; ++k) sum += V[a * r + k] * lam[k] * V[b * r + k];
        G[e] = sum;
    }
    __syncthreads();
    // T = D^T W -> V
    for (int e = lane; e < n2; e += 64) {
        const int a = e / r, b = e - a * r;
        double sum = 0.0;
        for (int k = 0; k < r; ++k) sum += D[k * r + a] * G[k * r + b];
        V[e] = sum;
    }
    __syncthreads();
    const double rh = (double)rho[slab];
    for (int e = lane; e < n2; e += 64) {
        const int a = e / r, b = e - a * r;
        double sum = 0.0;
        for (int k = 0; k < r; ++k) sum += V[k * r + a] * Sm[k * r + b];
        acc_out[(long)slab * (n2 + 1) + e] = rh * sum;
        T[(long)slab * n2 + e] = (float)V[e];
    }
    if (lane == 0) acc_out[(long)slab * (n2 + 1) + n2] = rh;
}

// P = Y T_slab  (one lane per row)
template <int RP>
__global__ __launch_bounds__(256) void k_pf2_apply(ModeView mv, const float *__restrict__ U, const float *__restrict__ T,
                                                   float *__restrict__ P, int r) {
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= mv.n_tiles) return;
    const int slab = __builtin_amdgcn_readfirstlane(mv.tile_slab[tile]);
    const int nrows = __builtin_amdgcn_readfirstlane(mv.tile_nrows[tile]);
    const bool valid = lane < nrows;
    const long j = (long)mv.tile_row0[tile] + (valid ? lane : 0);
    const float *__restrict__ Ts = T + (long)slab * r * r;
    float y[RP], p[RP];
#pragma unroll
    for (int c = 0; c < RP; ++c) {
        y[c] = (c < r) ? mv.F[j * r + c] + U[j * r + c] : 0.f;
        p[c] = 0.f;
    }
#pragma unroll
    for (int c = 0; c < RP; ++c) {
        if (c < r) {
#pragma unroll
            for (int d = 0; d < RP; ++d)
                if (d < r) p[d] = fmaf(y[c], Ts[c * r + d], p[d]);
        }
    }
    if (valid) {
#pragma unroll
        for (int c = 0; c < RP; ++c)
            if (c < r) P[j * r + c] = p[c];
    }
}

__global__ void k_pf2_sum(const double *__restrict__ acc, int n_slabs, int n_el, float *__restrict__ red) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_el) return;
    double s = 0.0;
    for (int i = 0; i < n_slabs; ++i) s += acc[(long)i * n_el + e];
    red[e] = (float)s;
}

__global__ void k_pf2_delta(const float *__restrict__ red, int r, float *__restrict__ Delta) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < r * r) Delta[e] = red[e] / red[r * r];
}

// dual update of the PARAFAC2 penalty: U = F - (P Delta - U)
template <int RP>
__global__ __launch_bounds__(256) void k_rows_pf2_dual(ModeView mv, RegSet regs, int k, int r) {
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= mv.n_tiles) return;
    if (lane >= mv.tile_nrows[tile]) return;
    const long j = (long)mv.tile_row0[tile] + lane;
    const float *__restrict__ D = regs.aux2[k];
    float p[RP];
#pragma unroll
    for (int c = 0; c < RP; ++c) p[c] = (c < r) ? regs.aux[k][j * r + c] : 0.f;
#pragma unroll
    for (int c = 0; c < RP; ++c) {
        if (c < r) {
            float z = 0.f;
#pragma unroll
            for (int d = 0; d < RP; ++d)
                if (d < r) z = fmaf(p[d], D[d * r + c], z);
            const float f = mv.F[j * r + c], u = regs.dual[k][j * r + c];
            regs.dual[k][j * r + c] = f - (z - u);
        }
    }
}

// =========================================================================================================
// host launchers
// =========================================================================================================
#define DISPATCH_RP(c, KERNEL, grid, block, ...)                                                         \
    switch ((c)->RP) {                                                                                   \
        case 4: hipLaunchKernelGGL((KERNEL<4>), grid, block, 0, (c)->stream, __VA_ARGS__); break;         \
        case 8: hipLaunchKernelGGL((KERNEL<8>), grid, block, 0, (c)->stream, __VA_ARGS__); break;         \
        case 16: hipLaunchKernelGGL((KERNEL<16>), grid, block, 0, (c)->stream, __VA_ARGS__); break;       \
        case 32: hipLaunchKernelGGL((KERNEL<32>), grid, block, 0, (c)->stream, __VA_ARGS__); break;       \
        default: hipLaunchKernelGGL((KERNEL<64>), grid, block, 0, (c)->stream, __VA_ARGS__); break;       \
    }

int mcl_launch_rows_solve(mcl_context *c, int mode) {
    ModeView mv = view_of(c, mode);
    if (mv.n_tiles == 0) return 0;
    const float *rhs = (mode == 1) ? c->XC : c->GR + (long)c->r * c->r;
    const float *Arows = (mode == 1) ? c->A : nullptr;
    const float *Linv = (mode == 1) ? c->LinvB : c->LinvC;
    dim3 grid((unsigned)((mv.n_tiles + 3) / 4)), block(256);
    DISPATCH_RP(c, k_rows_solve, grid, block, mv, rhs, Arows, Linv, c->regs[mode], c->r);
    MCL_CHECK_HIP(c, hipGetLastError());
    return 0;
}

int mcl_launch_A_rows_solve(mcl_context *c) {
    if (c->I == 0) return 0;
    hipLaunchKernelGGL(k_A_rows_solve, dim3((unsigned)c->I), dim3(64), 0, c->stream, c->rhsA, c->rhoA, c->LinvA, c->A,
                       c->regs[0], c->r);
    MCL_CHECK_HIP(c, hipGetLastError());
    return 0;
}

static UniScratch uni_scratch(mcl_context *c) {
    const int64_t maxrows = std::max<int64_t>(c->N, std::max<int64_t>(c->I, c->K));
    const int64_t n1 = (maxrows + std::max<int64_t>(c->I, 1)) * c->r;
    UniScratch s;
    double *d = c->uni_f64;
    s.lvL = d, s.lvR = d + n1, s.eL = d + 2 * n1, s.eR = d + 3 * n1;
    s.sy = d + 4 * n1, s.sy2 = d + 5 * n1, s.sw = d + 6 * n1, s.cum2 = d + 7 * n1;
    s.stL = c->uni_i32, s.stR = c->uni_i32 + maxrows * c->r;
    return s;
}

int mcl_launch_generic_prox_local(mcl_context *c, int mode, int k) {
    ModeView mv = view_of(c, mode);
    if (mv.n_tiles == 0) return 0;
    const RegSet &rs = c->regs[mode];
    dim3 grid((unsigned)((mv.n_tiles + 3) / 4)), block(256);
    switch (rs.kind[k]) {
        case MCL_PEN_NN:
        case MCL_PEN_BOX:
        case MCL_PEN_L1:
            hipLaunchKernelGGL(k_rows_prox_rowsep, grid, block, 0, c->stream, mv, rs, k, c->r);
            break;
        case MCL_PEN_L2BALL:
            hipLaunchKernelGGL(k_slab_colsq, dim3((unsigned)mv.n_slabs), dim3(256), 0, c->stream, mv.ext, mv.F,
                               rs.dual[k], rs.nonneg[k], c->r, c->RP, c->colsq);
            hipLaunchKernelGGL(k_rows_l2ball, grid, block, 0, c->stream, mv, rs, k, c->r, c->colsq);
            break;
        case MCL_PEN_UNIMODAL: {
            const long nthreads = (long)mv.n_slabs * c->r;
            hipLaunchKernelGGL(k_slab_unimodal, dim3((unsigned)((nthreads + 63) / 64)), dim3(64), 0, c->stream, mv.ext,
                               mv.n_slabs, mv.F, rs, k, c->r, uni_scratch(c));
            break;
        }
        case MCL_PEN_PARAFAC2: {
            if (mode != 1) {
                c->err = "PARAFAC2 constraint can only be imposed with mode=1";
                return 1;
            }
            const int r = c->r, n2 = r * r;
            hipLaunchKernelGGL(k_pf2_gram, dim3((unsigned)c->I), dim3(256), sizeof(float) * 64 * r, c->stream, mv.ext,
                               mv.F, rs.dual[k], r, c->pf2_S);
            const size_t sm = sizeof(double) * (size_t)(4 * n2 + r);
            if (sm > 65536) {
                MCL_CHECK_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_pf2_algebra),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
            }
            hipLaunchKernelGGL(k_pf2_algebra, dim3((unsigned)c->I), dim3(64), sm, c->stream, c->pf2_S, rs.aux2[k],
                               c->rhoB, r, c->pf2_T, c->pf2_acc);
            DISPATCH_RP(c, k_pf2_apply, grid, block, mv, (const float *)rs.dual[k], (const float *)c->pf2_T, rs.aux[k], r);
            hipLaunchKernelGGL(k_pf2_sum, dim3((unsigned)((n2 + 1 + 255) / 256)), dim3(256), 0, c->stream, c->pf2_acc,
                               (int)c->I, n2 + 1, c->pf2_red);
            break;
        }
        default:
            c->err = "penalty kind has no native prox (EXTERNAL penalties are evaluated by the host)";
            return 1;
    }
    MCL_CHECK_HIP(c, hipGetLastError());
    return 0;
}

int mcl_launch_generic_prox_finish(mcl_context *c, int mode, int k) {
    const RegSet &rs = c->regs[mode];
    if (rs.kind[k] != MCL_PEN_PARAFAC2) return 0;  // dual update already fused into the local step
    ModeView mv = view_of(c, mode);
    if (mv.n_tiles == 0) return 0;
    const int n2 = c->r * c->r;
    hipLaunchKernelGGL(k_pf2_delta, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, c->stream, c->pf2_red, c->r,
                       rs.aux2[k]);
    dim3 grid((unsigned)((mv.n_tiles + 3) / 4)), block(256);
    DISPATCH_RP(c, k_rows_pf2_dual, grid, block, mv, rs, k, c->r);
    MCL_CHECK_HIP(c, hipGetLastError());
    return 0;
}
