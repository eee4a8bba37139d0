// Condition estimates of the normal equations a PENALTY-FREE mode solves (mcl_condition_probe).
//
// A mode without any penalty solves its un-shifted (or only l2-shifted) r x r normal equations - the reference with an fp64 SVD
// (decomposition.py:172, 252-256, 319-321):
//     mode 0 (A):  Q_i = (B_i^T B_i) o (C^T C) + l2_A I            one system per matrix   (decomposition.py:155-172)
//     mode 1 (B):  L_i = (a_i a_i^T) o (C^T C) + l2_B I            one system per matrix   (decomposition.py:240-256)
//     mode 2 (C):  G   = sum_i (a_i a_i^T) o (B_i^T B_i) + l2_C I   one system              (decomposition.py:312-321)
// Whatever the fp32 storage and the fp32 matrix-core contractions of the fast kernels leave in the right-hand sides and in
// the state of the preceding phase (1e-8 .. 4e-7 relative) comes out of such a solve multiplied by the
// condition number of its system.  The probe computes that number FROM THE FACTORS ALONE (no pass over X): per matrix the
// fp64 Gram matrix B_i^T B_i, the systems above, their inverses by in-place Gauss-Jordan elimination in LDS, and
//     kappa = ||M||_F ||M^-1||_F      (cond_2(M) <= kappa <= r cond_2(M))
// - the largest over the matrices for modes 0 / 1.  A host uses it to move a mid-size problem with an ill-conditioned
// penalty-free mode to the exact arithmetic BEFORE it iterates (matcouply_amd/decomposition.py: `arithmetic="auto"`).
// Deterministic: the cross-matrix sum of G runs over <= 256 fixed groups of consecutive matrices, each summed in order.
#include <algorithm>

#include "mcl_internal.h"

namespace {

constexpr int COND_GROUPS = 256;  // workgroups of the slab pass = partial sums of G (workspace: COND_GROUPS x (r^2 + 2) doubles)
constexpr int COND_EPT = 16;      // matrix elements per thread (rank <= 64: 4096 / 256)

// sum over the workgroup, fixed order (wave sums by butterfly, then the four waves in order)
__device__ double wg_sum(double v, double *red) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// kappa_F of the symmetric positive (semi-)definite r x r matrix in W (LDS, overwritten by its inverse)
__device__ double kappa_inplace(double *W, int r, double *red) {
    const int t = threadIdx.x, r2 = r * r;
    double f2 = 0.0;
    for (int e = t; e < r2; e += 256) f2 = fma(W[e], W[e], f2);
    f2 = wg_sum(f2, red);
    // in-place Gauss-Jordan inversion without pivoting (the matrix is symmetric positive definite, or the result is not
    // finite and the caller reports "singular"): pivot p turns column p of the working matrix into column p of the inverse
    for (int p = 0; p < r; ++p) {
        __syncthreads();
        const double d = 1.0 / W[p * r + p];
        double nv[COND_EPT];
#pragma unroll
        for (int q = 0; q < COND_EPT; ++q) {
            const int e = t + 256 * q;
            if (e < r2) {
                const int i = e / r, j = e - i * r;
                const double wip = W[i * r + p], wpj = W[p * r + j];
                nv[q] = (i == p) ? (j == p ? d : wpj * d) : (j == p ? -wip * d : fma(-wip * d, wpj, W[e]));
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < COND_EPT; ++q) {
            const int e = t + 256 * q;
            if (e < r2) W[e] = nv[q];
        }
    }
    __syncthreads();
    double g2 = 0.0;
    for (int e = t; e < r2; e += 256) g2 = fma(W[e], W[e], g2);
    g2 = wg_sum(g2, red);
    const double k = sqrt(f2) * sqrt(g2);
    return (k < 1e300 && k == k) ? k : 1e300;  // singular / not finite: "as bad as it gets"
}

// One workgroup per group of consecutive matrices.  want bit m: mode m is penalty-free and will be updated.
__global__ __launch_bounds__(256) void k_cond_slabs(const float *__restrict__ A, const float *__restrict__ B,
                                                    const int *__restrict__ row_ptr, const double *__restrict__ CtC, int I, int r,
                                                    double l2A, double l2B, int want, double *__restrict__ part) {
    extern __shared__ double csm[];
    double *W = csm, *red = csm + r * r;
    const int t = threadIdx.x, r2 = r * r, g = blockIdx.x, n_g = gridDim.x;
    const int i0 = (int)((long)I * g / n_g), i1 = (int)((long)I * (g + 1) / n_g);
    double gacc[COND_EPT], ctc[COND_EPT];
#pragma unroll
    for (int q = 0; q < COND_EPT; ++q) {
        const int e = t + 256 * q;
        gacc[q] = 0.0;
        ctc[q] = e < r2 ? CtC[e] : 0.0;
    }
    double kA = 0.0, kB = 0.0;
    for (int i = i0; i < i1; ++i) {
        const int j0 = row_ptr[i], j1 = row_ptr[i + 1];
        if (j1 <= j0) continue;  // an empty matrix has no system
        double s[COND_EPT];
#pragma unroll
        for (int q = 0; q < COND_EPT; ++q) s[q] = 0.0;
        for (int j = j0; j < j1; ++j) {
            const float *b = B + (long)j * r;
#pragma unroll
            for (int q = 0; q < COND_EPT; ++q) {
                const int e = t + 256 * q;
                if (e < r2) {
                    const int p = e / r, c = e - p * r;
                    s[q] = fma((double)b[p], (double)b[c], s[q]);  // exact products of the stored fp32 values
                }
            }
        }
        double aa[COND_EPT];
#pragma unroll
        for (int q = 0; q < COND_EPT; ++q) {
            const int e = t + 256 * q;
            aa[q] = 0.0;
            if (e < r2) {
                const int p = e / r, c = e - p * r;
                aa[q] = (double)A[(long)i * r + p] * (double)A[(long)i * r + c];
                gacc[q] = fma(aa[q], s[q], gacc[q]);
            }
        }
        if (want & 1) {
            __syncthreads();
#pragma unroll
            for (int q = 0; q < COND_EPT; ++q) {
                const int e = t + 256 * q;
                if (e < r2) W[e] = fma(s[q], ctc[q], (e / r == e % r) ? l2A : 0.0);
            }
            __syncthreads();
            kA = fmax(kA, kappa_inplace(W, r, red));
        }
        if (want & 2) {
            __syncthreads();
#pragma unroll
            for (int q = 0; q < COND_EPT; ++q) {
                const int e = t + 256 * q;
                if (e < r2) W[e] = fma(aa[q], ctc[q], (e / r == e % r) ? l2B : 0.0);
            }
            __syncthreads();
            kB = fmax(kB, kappa_inplace(W, r, red));
        }
    }
    double *out = part + (long)g * (r2 + 2);
#pragma unroll
    for (int q = 0; q < COND_EPT; ++q) {
        const int e = t + 256 * q;
        if (e < r2) out[e] = gacc[q];
    }
    if (t == 0) out[r2] = kA, out[r2 + 1] = kB;
}

__global__ __launch_bounds__(256) void k_cond_final(const double *__restrict__ part, int n_g, int r, double l2C, int want,
                                                    int accumulate, double *__restrict__ out) {
    extern __shared__ double csm[];
    double *W = csm, *red = csm + r * r;
    const int t = threadIdx.x, r2 = r * r;
    for (int e = t; e < r2; e += 256) {
        double s = 0.0;
        for (int g = 0; g < n_g; ++g) s += part[(long)g * (r2 + 2) + e];  // fixed order
        W[e] = s + ((e / r == e % r) ? l2C : 0.0);
    }
    __syncthreads();
    const double kC = (want & 4) ? kappa_inplace(W, r, red) : 0.0;
    if (t == 0) {
        double kA = 0.0, kB = 0.0;
        for (int g = 0; g < n_g; ++g) kA = fmax(kA, part[(long)g * (r2 + 2) + r2]), kB = fmax(kB, part[(long)g * (r2 + 2) + r2 + 1]);
        if (accumulate) {  // monitoring (mcl_condition_monitor): the running maximum of the modes asked for, the others untouched
            if (want & 1) out[0] = fmax(out[0], kA);
            if (want & 2) out[1] = fmax(out[1], kB);
            if (want & 4) out[2] = fmax(out[2], kC);
        } else {
            out[0] = (want & 1) ? kA : 0.0;
            out[1] = (want & 2) ? kB : 0.0;
            out[2] = kC;
        }
    }
}

}  // namespace

// While a monitor is installed: the worst conditioning the PARAFAC2 polar factors of the inner iteration just enqueued have met.
// The Newton-Schulz kernel leaves x_i = 1 / ||(G_i / tr G_i)^-1/2||_F per matrix (G_i = (Y_i Delta^T)^T (Y_i Delta^T)), i.e.
// 1 / x_i ~ ||sigma|| / sigma_min of Y_i Delta^T, and flags the matrices it had to hand to the Jacobi / QR routes (status > 0:
// no convergence, rank-deficient, lambda_min <= 1e-10 lambda_max) - those count as 1e8.  out[3] keeps the running maximum.
namespace {
__global__ __launch_bounds__(256) void k_pf2_cond_track(const float *__restrict__ xmin, const int *__restrict__ status, int I,
                                                        double *__restrict__ out) {
    __shared__ double red[4];
    double worst = 0.0;
    for (int i = threadIdx.x; i < I; i += 256) {
        const float x = xmin[i];
        const double k = status[i] > 0 ? 1e8 : (x > 0.f ? 1.0 / (double)x : 0.0);
        worst = fmax(worst, k);
    }
    for (int off = 32; off > 0; off >>= 1) worst = fmax(worst, __shfl_xor(worst, off));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = worst;
    __syncthreads();
    if (threadIdx.x == 0) out[3] = fmax(out[3], fmax(fmax(red[0], red[1]), fmax(red[2], red[3])));
}
}  // namespace

int mcl_launch_pf2_cond_track(mcl_context *c) {
    if (!c->cond_monitor || !c->pf2_xmin || !c->pf2_status || c->I == 0) return 0;
    hipLaunchKernelGGL(k_pf2_cond_track, dim3(1), dim3(256), 0, c->stream, c->pf2_xmin, c->pf2_status, (int)c->I, c->cond_monitor);
    MCL_CHECK_HIP(c, hipGetLastError());
    return 0;
}

int64_t mcl_cond_part_doubles(const mcl_context *c) { return (int64_t)COND_GROUPS * ((int64_t)c->r * c->r + 2); }

// CtC64 must be current (api.hip: ensure_ctc).  `want`: bit m set = report mode m.
int mcl_launch_cond_probe(mcl_context *c, int want, double *out, bool accumulate) {
    const int r = c->r;
    const int n_g = (int)std::max<int64_t>(1, std::min<int64_t>(COND_GROUPS, c->I));
    const size_t lds = ((size_t)r * r + 8) * sizeof(double);
    hipLaunchKernelGGL(k_cond_slabs, dim3(n_g), dim3(256), lds, c->stream, c->A, c->B, c->row_ptr_dev, c->CtC64, (int)c->I, r,
                       c->opt.l2_penalty[0], c->opt.l2_penalty[1], want, c->cond_part);
    hipLaunchKernelGGL(k_cond_final, dim3(1), dim3(256), lds, c->stream, c->cond_part, n_g, r, c->opt.l2_penalty[2], want, accumulate ? 1 : 0, out);
    MCL_CHECK_HIP(c, hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// mcl_read_bandwidth: what this box's memory system delivers to a pure streaming read (bench.py prices its kernels against
// the 8 TB/s data-sheet peak AND against this: roofline.frac_achievable).  Two access geometries, the better one counts:
// grid-stride 16-byte reads (consecutive lanes / waves / workgroups read consecutive memory) and one contiguous chunk per
// wave, 16 KB in flight per wave in both.
// ---------------------------------------------------------------------------------------------------------
namespace {
typedef float bw_f4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k_read_bw(const bw_f4 *__restrict__ p, long n_f4, long chunk_f4, float *__restrict__ out) {
    constexpr int U = 16;
    bw_f4 acc = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 0) {
        const long stride = (long)gridDim.x * 256;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i + (U - 1) * stride < n_f4; i += U * stride) {
            bw_f4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(p + i + u * stride);
#pragma unroll
            for (int u = 0; u < U; ++u) acc += v[u];
        }
    } else {
        const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6), base = w * chunk_f4;
        const int lane = threadIdx.x & 63;
        for (long s = 0; s + U <= chunk_f4 / 64; s += U) {
            bw_f4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(p + base + (s + u) * 64 + lane);
#pragma unroll
            for (int u = 0; u < U; ++u) acc += v[u];
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = acc[0];  // (keeps the loads)
}
}  // namespace

extern "C" int mcl_read_bandwidth(const void *buf, int64_t bytes, int32_t repeats, float *scratch, void *hip_stream, double *gbps) {
    if (!buf || !scratch || !gbps || bytes < (int64_t(1) << 20) || repeats < 1) return 1;
    hipStream_t s = reinterpret_cast<hipStream_t>(hip_stream);
    const long n_f4 = bytes / 16;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return 1;
    double best = 0.0;
    for (int mode = 0; mode < 2; ++mode) {
        const int blocks = 2048;
        const long chunk_f4 = (n_f4 / ((long)blocks * 4)) / (64 * 16) * (64 * 16);
        const double moved = mode == 0 ? 16.0 * ((n_f4 / ((long)blocks * 256 * 16)) * ((long)blocks * 256 * 16)) : 16.0 * chunk_f4 * blocks * 4;
        if (moved <= 0) continue;
        for (int rep = 0; rep < repeats + 2; ++rep) {
            if (rep == 2) (void)hipEventRecord(e0, s);
            if (mode == 0) hipLaunchKernelGGL(k_read_bw<0>, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const bw_f4 *>(buf), n_f4, chunk_f4, scratch);
            else hipLaunchKernelGGL(k_read_bw<1>, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const bw_f4 *>(buf), n_f4, chunk_f4, scratch);
        }
        (void)hipEventRecord(e1, s);
        if (hipEventSynchronize(e1) != hipSuccess) return 1;
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms > 0.f) best = std::max(best, moved * repeats / (ms * 1e-3) / 1e9);
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *gbps = best;
    return hipGetLastError() == hipSuccess && best > 0.0 ? 0 : 1;
}
