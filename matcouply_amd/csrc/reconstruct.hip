// Dense reconstruction of a coupled matrix factorisation on the device (gfx950):
//     M_i = (B_i diag(w o a_i)) C^T   for all I matrices at once, packed along rows like X
// Replaces the reference's cmf_to_matrices / cmf_to_matrix (coupled_matrices.py:365-497), the step after the solver
// (SURVEY.md 8f.2).  Write-bound (4 N K bytes out, 4 N r in): a wave owns 16 packed rows, keeps their weighted B rows
// in registers in the row layout of rows_mfma.h and walks the K columns 64 at a time - 4 r/4 fp32 MFMAs per 16 x 16 tile
// against C rows fetched from L2, one 16-byte store per lane and tile (16 rows x 64 contiguous bytes per instruction).
#include "mcl_internal.h"
#include "rows_mfma.h"

template <int NBR, bool VEC>
__global__ __launch_bounds__(256) void k_reconstruct(const float *__restrict__ A, const float *__restrict__ B,
                                                     const float *__restrict__ C, const float *__restrict__ weights,
                                                     const int *__restrict__ slab_of_row, long N, int K, int r,
                                                     float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long blk = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long j0 = blk * 16;
    if (j0 >= N) return;
    const int row16 = lane & 15, g = lane >> 4;
    const bool ok = j0 + row16 < N;
    const long j = ok ? j0 + row16 : N - 1;
    const int slab = slab_of_row[j];
    // t[h] = columns 16h + 4g .. +3 of the row's B o (w o a_slab)   (zeros outside the rank)
    f32x4 t[NBR];
#pragma unroll
    for (int h = 0; h < NBR; ++h) {
        const int col = 16 * h + 4 * g;
        f32x4 b = row_ld4<VEC>(B, j, col, ok, r);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int cidx = col + v;
            float a = 0.f;
            if (cidx < r) a = A[(long)slab * r + cidx] * (weights ? weights[cidx] : 1.f);
            b[v] *= a;
        }
        t[h] = b;
    }
    const bool st_vec = VEC && (K % 4 == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
    for (int ct = 0; ct < K; ct += 16) {
        // A-operand: lane (c' = l & 15, g) feeds C[ct + c'][16h + 4g + kq]; D lane (row, g) reg v = M[row][ct + 4g + v]
        const int crow = min(ct + row16, K - 1);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int h = 0; h < NBR; ++h) {
            const f32x4 cf = row_ld4<VEC>(C, crow, 16 * h + 4 * g, ct + row16 < K, r);
#pragma unroll
            for (int kq = 0; kq < 4; ++kq) acc = MFMA16(cf[kq], t[h][kq], acc);
        }
        const int col = ct + 4 * g;
        if (!ok) continue;
        if (st_vec && col + 3 < K) {
            *reinterpret_cast<f32x4 *>(out + j * K + col) = acc;
        } else {
#pragma unroll
            for (int v = 0; v < 4; ++v)
                if (col + v < K) out[j * K + col + v] = acc[v];
        }
    }
}

extern "C" int mcl_cmf_to_packed(const float *A, const float *B, const float *C, const float *weights,
                                 const int32_t *slab_of_row, int64_t N, int64_t K, int32_t rank, float *out,
                                 void *hip_stream) {
    if (N == 0) return 0;
    if (!A || !B || !C || !slab_of_row || !out || K < 1 || rank < 1 || rank > MCL_MAX_RANK) return 1;
    hipStream_t s = reinterpret_cast<hipStream_t>(hip_stream);
    const bool vec = (rank % 4 == 0) && ((reinterpret_cast<uintptr_t>(B) & 15) == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0);
    const dim3 grid((unsigned)(((N + 15) / 16 + 3) / 4)), block(256);
    const int nbr = rank <= 16 ? 1 : (rank <= 32 ? 2 : 4);
#define MCL_RC(NBR_, VEC_)                                                                                           \
    hipLaunchKernelGGL((k_reconstruct<NBR_, VEC_>), grid, block, 0, s, A, B, C, weights, slab_of_row, (long)N, (int)K, \
                       (int)rank, out)
    if (vec) {
        if (nbr == 1) MCL_RC(1, true);
        else if (nbr == 2) MCL_RC(2, true);
        else MCL_RC(4, true);
    } else {
        if (nbr == 1) MCL_RC(1, false);
        else if (nbr == 2) MCL_RC(2, false);
        else MCL_RC(4, false);
    }
#undef MCL_RC
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
