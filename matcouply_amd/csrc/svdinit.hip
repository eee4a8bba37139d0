// init="svd" / "threshold_svd" (decomposition.py:41-54) for data that is resident in HBM: B_i = the leading `rank` left
// singular vectors of X_i, C = the leading right singular vectors of the stacked matrices, without a copy of X to the host.
//
// Per matrix M (a slab X_i, or the whole stack) the leading eigenpairs of the K x K Gram matrix G = M^T M come from subspace
// iteration with a Rayleigh-Ritz step per iteration, all in fp64:
//     Y = G Q;  S = Y^T Y = Q^T G^2 Q;  S = W diag(lam) W^T (cyclic Jacobi in LDS, m = min(K, rank + 8) <= 72 vectors);
//     Q <- Y W diag(lam)^-1/2  (orthonormal, columns ordered by eigenvalue: the Ritz vectors of G^2 in span(Y))
// until the leading `rank` Ritz values sqrt(lam_k) (the eigenvalues of G = the squared singular values) stop moving
// (relative 1e-13, twice in a row; the error of the vectors is the square root of the error of the values).  The stack's Gram
// matrix is the sum of the slabs' (fixed order).  Left vectors: U = X_i V, columns normalised (||X_i v_k|| = sigma_k).
// SIGNS: a singular vector is defined up to its sign and LAPACK's choice is not reproducible; here the entry of largest
// magnitude of every column of B_i, and of every column of C, is positive.  The reference's trajectory from this initialiser
// is therefore NOT what the device form reproduces: `matcouply_amd` takes it only for data that already
// lives on the device.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "mcl_internal.h"

namespace {

constexpr int SVD_OVERSAMPLE = 8;
constexpr int SVD_MAX_IT = 400;
static std::string g_svd_error;

inline int svd_m(int64_t K, int rank) { return (int)std::min<int64_t>(K, rank + SVD_OVERSAMPLE); }

// G_b = X_b^T X_b (fp64 sums of exact products of the stored fp32 values), 32 x 32 output tile per workgroup
__global__ __launch_bounds__(256) void k_svd_gram(const float *__restrict__ X, const int *__restrict__ ext, int b0, int K,
                                                  double *__restrict__ G) {
    __shared__ float As[32][33], Bs[32][33];
    const int b = blockIdx.z, slab = b0 + b;
    const long s0 = ext[slab];
    const int n = ext[slab + 1] - ext[slab];
    const int a0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;  // 2 x 2 outputs per thread: (2 ty + u, 2 tx + v)
    double acc[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
    for (int j0 = 0; j0 < n; j0 += 32) {
        __syncthreads();
        for (int e = threadIdx.x; e < 32 * 32; e += 256) {
            const int jj = e >> 5, cc = e & 31;
            const bool okj = j0 + jj < n;
            As[jj][cc] = (okj && a0 + cc < K) ? X[(s0 + j0 + jj) * K + a0 + cc] : 0.f;
            Bs[jj][cc] = (okj && c0 + cc < K) ? X[(s0 + j0 + jj) * K + c0 + cc] : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int jj = 0; jj < 32; ++jj) {
            const double a_0 = (double)As[jj][2 * ty], a_1 = (double)As[jj][2 * ty + 1];
            const double b_0 = (double)Bs[jj][2 * tx], b_1 = (double)Bs[jj][2 * tx + 1];
            acc[0][0] = fma(a_0, b_0, acc[0][0]), acc[0][1] = fma(a_0, b_1, acc[0][1]);
            acc[1][0] = fma(a_1, b_0, acc[1][0]), acc[1][1] = fma(a_1, b_1, acc[1][1]);
        }
    }
    double *Gb = G + (long)b * K * K;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int a = a0 + 2 * ty + u, c = c0 + 2 * tx + v;
            if (a < K && c < K) Gb[(long)a * K + c] = acc[u][v];
        }
}

// Gstack += sum of the batch's Gram matrices, slabs in ascending order
__global__ __launch_bounds__(256) void k_svd_add(const double *__restrict__ G, int nb, long KK, double *__restrict__ Gstack, int first) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= KK) return;
    double s = first ? 0.0 : Gstack[e];
    for (int b = 0; b < nb; ++b) s += G[(long)b * KK + e];
    Gstack[e] = s;
}

__device__ __forceinline__ double hash_unit(unsigned a, unsigned b) {  // deterministic start vectors in (-1, 1)
    unsigned h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA77u;
    h ^= h >> 15, h *= 0x2C1B3C6Du, h ^= h >> 12, h *= 0x297A2D39u, h ^= h >> 15;
    return (double)(h >> 8) * (2.0 / 16777216.0) - 1.0;
}

// eigen-decomposition of the symmetric m x m matrix S (LDS) by cyclic Jacobi with round-robin pairs: W <- eigenvectors
// (columns), the diagonal of S <- eigenvalues.  me = m rounded up to even (a dummy player idles).  All 256 threads call it.
__device__ void jacobi_lds(double *S, double *W, double *cs, int m) {
    const int tid = threadIdx.x;
    const int me = (m + 1) & ~1, half = me / 2;
    for (int e = tid; e < m * m; e += 256) W[e] = ((e / m) == (e % m)) ? 1.0 : 0.0;
    __shared__ double off_sh, diag_sh;
    __syncthreads();
    for (int sweep = 0; sweep < 40; ++sweep) {
        if (tid == 0) {
            double off = 0.0, dg = 0.0;
            for (int a = 0; a < m; ++a)
                for (int b = 0; b < m; ++b) (a == b ? dg : off) += S[a * m + b] * S[a * m + b];
            off_sh = off, diag_sh = dg;
        }
        __syncthreads();
        if (!(off_sh > 1e-30 * diag_sh)) break;
        for (int step = 0; step < me - 1; ++step) {
            // pair k of this step: (p, q)
            auto pair_of = [&](int k, int &p, int &q) {
                if (k == 0) p = me - 1, q = step;
                else p = (step + k) % (me - 1), q = (step - k + (me - 1)) % (me - 1);
                if (p > q) { const int t = p; p = q; q = t; }
            };
            if (tid < half) {
                int p, q;
                pair_of(tid, p, q);
                double c = 1.0, s = 0.0;
                if (q < m) {
                    const double apq = S[p * m + q], app = S[p * m + p], aqq = S[q * m + q];
                    if (fabs(apq) > 1e-300 && fabs(apq) > 1e-18 * sqrt(fabs(app * aqq))) {
                        const double tau = (aqq - app) / (2.0 * apq);
                        const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                        c = 1.0 / sqrt(1.0 + t * t), s = t * c;
                    }
                }
                cs[2 * tid] = c, cs[2 * tid + 1] = s;
            }
            __syncthreads();
            for (int e = tid; e < half * m; e += 256) {  // columns p, q of S and of W, every row i
                const int k = e / m, i = e - k * m;
                int p, q;
                pair_of(k, p, q);
                if (q >= m) continue;
                const double c = cs[2 * k], s = cs[2 * k + 1];
                const double sp = S[i * m + p], sq = S[i * m + q];
                S[i * m + p] = c * sp - s * sq, S[i * m + q] = s * sp + c * sq;
                const double wp = W[i * m + p], wq = W[i * m + q];
                W[i * m + p] = c * wp - s * wq, W[i * m + q] = s * wp + c * wq;
            }
            __syncthreads();
            for (int e = tid; e < half * m; e += 256) {  // rows p, q of S, every column j
                const int k = e / m, j = e - k * m;
                int p, q;
                pair_of(k, p, q);
                if (q >= m) continue;
                const double c = cs[2 * k], s = cs[2 * k + 1];
                const double sp = S[p * m + j], sq = S[q * m + j];
                S[p * m + j] = c * sp - s * sq, S[q * m + j] = s * sp + c * sq;
            }
            __syncthreads();
        }
    }
    __syncthreads();
}

// Subspace iteration of one matrix per workgroup (see the file header).  G: [K, K]; Q, Y: [K, m] scratch; out: Q holds the
// Ritz vectors (columns ordered by eigenvalue), theta [m] the Ritz values of G, info = iterations used (negative: not converged)
__global__ __launch_bounds__(256) void k_svd_subspace(const double *__restrict__ Gall, int K, int m, int rank, double *__restrict__ Qall,
                                                      double *__restrict__ Yall, double *__restrict__ theta_all, int *__restrict__ info,
                                                      int info0) {
    extern __shared__ double sm[];
    double *S = sm, *W = S + m * m, *lam = W + m * m, *cs = lam + m, *prev = cs + 2 * ((m + 1) / 2 + 1);
    int *order = reinterpret_cast<int *>(prev + m);
    __shared__ int done_sh;
    const int tid = threadIdx.x, b = blockIdx.x;
    const double *G = Gall + (long)b * K * K;
    double *Q = Qall + (long)b * K * m, *Y = Yall + (long)b * K * m, *theta = theta_all + (long)b * m;
    for (int e = tid; e < K * m; e += 256) Y[e] = hash_unit((unsigned)(e / m), (unsigned)(e % m) + 977u * (unsigned)(b + info0));
    for (int k = tid; k < m; k += 256) prev[k] = 0.0;
    int stable = 0, it_used = -SVD_MAX_IT;
    __syncthreads();
    for (int it = 0; it < SVD_MAX_IT; ++it) {
        if (it > 0) {  // Y = G Q
            for (int e = tid; e < K * m; e += 256) {
                const int i = e / m, c = e - i * m;
                double s0 = 0.0, s1 = 0.0;
                int j = 0;
                for (; j + 1 < K; j += 2) {
                    s0 = fma(G[(long)i * K + j], Q[(long)j * m + c], s0);
                    s1 = fma(G[(long)i * K + j + 1], Q[(long)(j + 1) * m + c], s1);
                }
                if (j < K) s0 = fma(G[(long)i * K + j], Q[(long)j * m + c], s0);
                Y[e] = s0 + s1;
            }
            __threadfence_block();
            __syncthreads();
        }
        for (int e = tid; e < m * m; e += 256) {  // S = Y^T Y
            const int a = e / m, c = e - a * m;
            double s = 0.0;
            if (c >= a)
                for (int i = 0; i < K; ++i) s = fma(Y[(long)i * m + a], Y[(long)i * m + c], s);
            S[e] = s;
        }
        __syncthreads();
        for (int e = tid; e < m * m; e += 256) {
            const int a = e / m, c = e - a * m;
            if (c < a) S[e] = S[c * m + a];
        }
        __syncthreads();
        jacobi_lds(S, W, cs, m);
        if (tid < m) lam[tid] = S[tid * m + tid];
        __syncthreads();
        if (tid < m) {  // rank of eigenvalue tid in descending order (ties: by index)
            int rk = 0;
            for (int k = 0; k < m; ++k) rk += (lam[k] > lam[tid]) || (lam[k] == lam[tid] && k < tid);
            order[rk] = tid;
        }
        __syncthreads();
        const double lam_max = lam[order[0]];
        for (int e = tid; e < K * m; e += 256) {  // Q = Y W diag(lam)^-1/2, columns in descending order
            const int i = e / m, c = e - i * m;
            const int src = order[c];
            const double l = lam[src];
            double s = 0.0;
            if (l > 1e-28 * lam_max && l > 0.0) {
                for (int k = 0; k < m; ++k) s = fma(Y[(long)i * m + k], W[k * m + src], s);
                s /= sqrt(l);
            }
            Q[e] = s;
        }
        if (tid == 0) {
            int ok = it > 0;
            const double t0 = sqrt(sqrt(fmax(lam_max, 0.0)));  // the largest Ritz value's scale (it > 0: sqrt(lam) ~ eig(G))
            for (int k = 0; k < m; ++k) {
                const double th = sqrt(fmax(lam[order[k]], 0.0));
                if (k < rank && !(fabs(th - prev[k]) <= 1e-13 * fmax(sqrt(fmax(lam_max, 0.0)), 1e-300))) ok = 0;
                prev[k] = th;
            }
            (void)t0;
            done_sh = ok;
        }
        __threadfence_block();
        __syncthreads();
        stable = done_sh ? stable + 1 : 0;
        if (stable >= 2) {
            it_used = it + 1;
            break;
        }
    }
    for (int k = tid; k < m; k += 256) theta[k] = prev[k];
    if (tid == 0) {
        // numerical rank below `rank` (an all-zero matrix, fewer independent rows than components): the Ritz vectors of the
        // null eigenvalues were set to zero above, where LAPACK returns an orthonormal completion - a zero column in the initial
        // B_i / C makes that component's normal equations singular from the first iteration on.  Reported as -(1000 + number of
        // such vectors): the host falls back to its LAPACK path
        int deficient = 0;
        const double lam_max = lam[order[0]];
        for (int k = 0; k < rank && k < m; ++k) {
            const double l = lam[order[k]];
            deficient += !(l > 1e-28 * lam_max && l > 0.0);
        }
        info[info0 + b] = deficient > 0 ? -(1000 + deficient) : it_used;
    }
}

// B_b = X_b V_b with unit columns, the entry of largest magnitude of every column positive (threshold: clipped at 0)
__global__ __launch_bounds__(256) void k_svd_left(const float *__restrict__ X, const int *__restrict__ ext, int b0, int K, int m, int r,
                                                  const double *__restrict__ Qall, double *__restrict__ Uws, long uws0, int threshold,
                                                  float *__restrict__ B) {
    __shared__ double red[256];
    __shared__ double red2[256];
    __shared__ double scale_sh[MCL_MAX_RANK];
    const int b = blockIdx.x, slab = b0 + b, tid = threadIdx.x;
    const long s0 = ext[slab];
    const int n = ext[slab + 1] - ext[slab];
    const double *V = Qall + (long)b * K * m;
    double *U = Uws + (s0 - uws0) * r;
    for (long e = tid; e < (long)n * r; e += 256) {
        const long j = e / r;
        const int c = (int)(e - j * r);
        double s = 0.0;
        for (int k = 0; k < K; ++k) s = fma((double)X[(s0 + j) * K + k], V[(long)k * m + c], s);
        U[e] = s;
    }
    __syncthreads();
    for (int c = 0; c < r; ++c) {  // column norm and the sign of the entry of largest magnitude (first one on ties)
        double sq = 0.0, best = -1.0, bsign = 1.0;
        long bidx = -1;
        for (long j = tid; j < n; j += 256) {
            const double u = U[j * r + c];
            sq = fma(u, u, sq);
            if (fabs(u) > best) best = fabs(u), bsign = u < 0.0 ? -1.0 : 1.0, bidx = j;
        }
        red[tid] = sq;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o) red[tid] += red[tid + o];
            __syncthreads();
        }
        const double nrm = sqrt(red[0]);
        __syncthreads();
        red[tid] = best, red2[tid] = bsign * (double)(bidx + 1);  // sign and (index + 1) packed
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o) {
                const double a1 = red[tid], a2 = red[tid + o];
                const double i1 = fabs(red2[tid]), i2 = fabs(red2[tid + o]);
                if (a2 > a1 || (a2 == a1 && i2 > 0.0 && (i1 == 0.0 || i2 < i1))) red[tid] = a2, red2[tid] = red2[tid + o];
            }
            __syncthreads();
        }
        if (tid == 0) scale_sh[c] = nrm > 0.0 ? (red2[0] < 0.0 ? -1.0 : 1.0) / nrm : 0.0;
        __syncthreads();
    }
    for (long e = tid; e < (long)n * r; e += 256) {
        const long j = e / r;
        const int c = (int)(e - j * r);
        double v = U[e] * scale_sh[c];
        if (threshold) v = fmax(v, 0.0);
        B[(s0 + j) * r + c] = (float)v;
    }
}

// C = the stack's Ritz vectors, the entry of largest magnitude of every column positive (threshold: clipped at 0)
__global__ __launch_bounds__(256) void k_svd_right(const double *__restrict__ Q, int K, int m, int r, int threshold, float *__restrict__ C) {
    __shared__ double sgn[MCL_MAX_RANK];
    const int tid = threadIdx.x;
    if (tid < r) {
        double best = -1.0, s = 1.0;
        for (int k = 0; k < K; ++k) {
            const double v = Q[(long)k * m + tid];
            if (fabs(v) > best) best = fabs(v), s = v < 0.0 ? -1.0 : 1.0;
        }
        sgn[tid] = s;
    }
    __syncthreads();
    for (int e = tid; e < K * r; e += 256) {
        const int k = e / r, c = e - k * r;
        double v = Q[(long)k * m + c] * sgn[c];
        if (threshold) v = fmax(v, 0.0);
        C[e] = (float)v;
    }
}

struct SvdPlan {
    int m, BS;
    int64_t off_ext, off_G, off_Gstack, off_Q, off_Y, off_theta, off_U, total;
};

SvdPlan svd_plan(int64_t I, int64_t K, int rank, int64_t max_batch_rows) {
    SvdPlan p{};
    p.m = svd_m(K, rank);
    p.BS = (int)std::min<int64_t>(std::max<int64_t>(I, 1), 256);
    while (p.BS > 1 && (int64_t)p.BS * K * K * 8 > (int64_t(2) << 30)) p.BS /= 2;  // <= 2 GiB of Gram matrices at a time
    int64_t off = 0;
    auto take = [&](int64_t bytes) {
        const int64_t o = off;
        off = (off + bytes + 255) & ~int64_t(255);
        return o;
    };
    p.off_ext = take((I + 1) * 4);
    p.off_G = take((int64_t)p.BS * K * K * 8);
    p.off_Gstack = take(K * K * 8);
    p.off_Q = take((int64_t)p.BS * K * p.m * 8);
    p.off_Y = take((int64_t)p.BS * K * p.m * 8);
    p.off_theta = take((int64_t)p.BS * p.m * 8);
    p.off_U = take(std::max<int64_t>(max_batch_rows, 1) * rank * 8);
    p.total = off;
    return p;
}

int64_t max_batch_rows_of(const int64_t *row_ptr, int64_t I, int BS) {
    int64_t mx = 0;
    for (int64_t b0 = 0; b0 < I; b0 += BS) mx = std::max(mx, row_ptr[std::min<int64_t>(I, b0 + BS)] - row_ptr[b0]);
    return mx;
}

}  // namespace

extern "C" {

const char *mcl_svd_init_last_error(void) { return g_svd_error.c_str(); }

int64_t mcl_svd_init_workspace_bytes(const int64_t *row_ptr, int64_t I, int64_t K, int32_t rank) {
    if (!row_ptr || I < 1 || K < 1 || rank < 1) return -1;
    const SvdPlan p0 = svd_plan(I, K, rank, 1);
    return svd_plan(I, K, rank, max_batch_rows_of(row_ptr, I, p0.BS)).total;
}

int mcl_svd_init(const float *X, const int64_t *row_ptr, int64_t I, int64_t K, int32_t rank, int32_t threshold, float *B, float *C,
                 void *workspace, int64_t workspace_bytes, int32_t *info, void *hip_stream) {
    auto fail = [](const std::string &msg) {
        g_svd_error = msg;
        return 1;
    };
    if (!X || !row_ptr || !B || !C || !workspace || !info) return fail("mcl_svd_init: NULL argument");
    if (I < 1 || K < 1 || rank < 1 || rank > MCL_MAX_RANK) return fail("mcl_svd_init: need I >= 1, K >= 1, 1 <= rank <= 64");
    if (rank > K) return fail("mcl_svd_init: rank exceeds the number of columns");
    for (int64_t i = 0; i < I; ++i)
        if (row_ptr[i + 1] - row_ptr[i] < rank) return fail("mcl_svd_init: a matrix has fewer rows than the rank");
    if (row_ptr[I] >= (int64_t(1) << 31)) return fail("mcl_svd_init: more than 2^31 packed rows are not supported");
    const SvdPlan p0 = svd_plan(I, K, rank, 1);
    const SvdPlan p = svd_plan(I, K, rank, max_batch_rows_of(row_ptr, I, p0.BS));
    if (workspace_bytes < p.total) return fail("mcl_svd_init: workspace too small (mcl_svd_init_workspace_bytes)");
    if (reinterpret_cast<uintptr_t>(workspace) & 255) return fail("mcl_svd_init: workspace must be 256-byte aligned");
    hipStream_t s = reinterpret_cast<hipStream_t>(hip_stream);
    char *ws = static_cast<char *>(workspace);
    int *ext = reinterpret_cast<int *>(ws + p.off_ext);
    double *G = reinterpret_cast<double *>(ws + p.off_G), *Gstack = reinterpret_cast<double *>(ws + p.off_Gstack);
    double *Q = reinterpret_cast<double *>(ws + p.off_Q), *Y = reinterpret_cast<double *>(ws + p.off_Y);
    double *theta = reinterpret_cast<double *>(ws + p.off_theta), *U = reinterpret_cast<double *>(ws + p.off_U);
    std::vector<int> h_ext((size_t)I + 1);
    for (int64_t i = 0; i <= I; ++i) h_ext[(size_t)i] = (int)row_ptr[i];
#define SVD_HIP(expr)                                                                  \
    do {                                                                               \
        const hipError_t e_ = (expr);                                                  \
        if (e_ != hipSuccess) return fail(std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
    SVD_HIP(hipMemcpyAsync(ext, h_ext.data(), sizeof(int) * ((size_t)I + 1), hipMemcpyHostToDevice, s));
    SVD_HIP(hipStreamSynchronize(s));  // (h_ext is a local)
    const int m = p.m;
    const size_t sm = sizeof(double) * (size_t)(2 * m * m + m + 2 * ((m + 1) / 2 + 1) + m) + sizeof(int) * (size_t)m + 64;
    SVD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_svd_subspace), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
    const int tiles = (int)((K + 31) / 32);
    const long KK = (long)K * K;
    for (int64_t b0 = 0; b0 < I; b0 += p.BS) {
        const int nb = (int)std::min<int64_t>(p.BS, I - b0);
        hipLaunchKernelGGL(k_svd_gram, dim3((unsigned)tiles, (unsigned)tiles, (unsigned)nb), dim3(256), 0, s, X, (const int *)ext, (int)b0,
                           (int)K, G);
        hipLaunchKernelGGL(k_svd_add, dim3((unsigned)((KK + 255) / 256)), dim3(256), 0, s, (const double *)G, nb, KK, Gstack, b0 == 0 ? 1 : 0);
        hipLaunchKernelGGL(k_svd_subspace, dim3((unsigned)nb), dim3(256), sm, s, (const double *)G, (int)K, m, (int)rank, Q, Y, theta, info,
                           (int)b0);
        hipLaunchKernelGGL(k_svd_left, dim3((unsigned)nb), dim3(256), 0, s, X, (const int *)ext, (int)b0, (int)K, m, (int)rank,
                           (const double *)Q, U, (long)row_ptr[b0], (int)threshold, B);
    }
    hipLaunchKernelGGL(k_svd_subspace, dim3(1), dim3(256), sm, s, (const double *)Gstack, (int)K, m, (int)rank, Q, Y, theta, info, (int)I);
    hipLaunchKernelGGL(k_svd_right, dim3(1), dim3(256), 0, s, (const double *)Q, (int)K, m, (int)rank, (int)threshold, C);
    SVD_HIP(hipGetLastError());
#undef SVD_HIP
    return 0;
}

}  // extern "C"
