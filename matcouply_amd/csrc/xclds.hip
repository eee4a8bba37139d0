// k_contract_xc_lds : XC = X C for K % 512 == 0 with the WHOLE fragment image of C resident in LDS (K x 16 NB floats <= 128 KB).
//
// k_contract_xc_row (contract.hip) re-reads the C fragments of every 256-column super-chunk from L2 into 64 NB registers, and
// because memory returns in order those loads have to be issued BEFORE the prefetch of the next X tile and waited for in front of
// it: one tile (16 KB per wave) is all a wave ever has in flight, at 492 VGPRs a SIMD holds one wave, and at config 5 (K = 1024,
// rank 32) the pass waits for memory half of its life (profiles/r6_c5_sq_counters.json: SQ_WAIT_INST_ANY 52 % of SQ_WAVE_CYCLES)
// and moves 70.9 GB in 16.7 ms = 4.4 TB/s where a streaming read reaches 6.8.  Here
//   * the workgroup copies the fragment image of C into LDS once (config 5: 128 KB of the CU's 160 KB; the MFMA B operands are
//     16-byte LDS reads, lane-linear: conflict-free),
//   * so the ONLY loads of the main loop are the X tiles (and, once per round, the rows of B for the fused A-phase reductions, in
//     the same batch): a DEPTH-slot register ring keeps four or eight 16-row x 128-column tiles (32 / 64 KB per wave) in
//     flight under exact s_waitcnt vmcnt(N) counts - every load unconditional at clamped addresses, a fixed number per step,
//   * tiles are staged through a wave-private 8 KB LDS image (16-byte slot index XORed with the row: conflict-free writes and
//     fragment reads), as in k_contract_xc_row.
// Arithmetic and its order are those of k_contract_xc_row: four fp32 chains per output (one per 64-column chunk modulo 4,
// ascending in K), summed pairwise at the end of a 16-row block; the fused reductions (GRAM) word for word.  Same segment /
// wave tables.  Reference: decomposition.py:147-158 (X_i C, diag(B_i^T X_i C), B_i^T B_i) and :242.
#include "mcl_internal.h"

typedef float xf32x4 __attribute__((ext_vector_type(4)));
typedef double xf64x4 __attribute__((ext_vector_type(4)));
#define XMFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

namespace {

// position in the wave's tile stream: segment, 16-row block, 128-column tile
struct TileCursor {
    int sg, s1;       // current segment, end of the wave's segments
    long row0;
    int nrows, nblk, blk, hs;
};

template <int NB, int GRAM, bool XNT, int DEPTH>
__global__ __launch_bounds__(256) void k_contract_xc_lds(const float *__restrict__ X, const float *__restrict__ Cfrag,
                                                         float *__restrict__ XC, const float *__restrict__ B,
                                                         const int *__restrict__ seg_row0, const int *__restrict__ seg_rows,
                                                         const int *__restrict__ wave_seg_ptr, int n_waves, int K, int r,
                                                         double *__restrict__ seg_rhs, double *__restrict__ seg_btb) {
    extern __shared__ float lds_dyn[];  // fragment image of C, then 4 waves x 16 rows x 128 floats
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, i16 = lane & 15;
    const int cf_f4 = (K >> 6) * 4 * NB * 64;  // float4 elements of the image
    {
        xf32x4 *dst = reinterpret_cast<xf32x4 *>(lds_dyn);
        const xf32x4 *src = reinterpret_cast<const xf32x4 *>(Cfrag);
        for (int e = threadIdx.x; e < cf_f4; e += 256) dst[e] = src[e];
    }
    __syncthreads();
    const xf32x4 *Cs = reinterpret_cast<const xf32x4 *>(lds_dyn);
    float *L = lds_dyn + 4 * cf_f4 + wave * (16 * 128);
    const int w = blockIdx.x * 4 + wave;
    if (w >= n_waves) return;
    const int s0 = wave_seg_ptr[w], s1 = wave_seg_ptr[w + 1];
    if (s0 >= s1) return;
    const int TPB = K >> 7;  // tiles per 16-row block (a multiple of DEPTH)

    // ---- the prefetch side: four tiles in registers
    xf32x4 xr[DEPTH][8];
    float bnx[NB][4];  // rows 4q + v, column 16 nb + i16 of B for the block the prefetch cursor is in
    int bcolc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) bcolc[nb] = min(16 * nb + i16, r - 1);
    auto seg_of = [&](TileCursor &c, int sg) {
        c.sg = sg;
        c.row0 = __builtin_amdgcn_readfirstlane(seg_row0[sg]);
        c.nrows = __builtin_amdgcn_readfirstlane(seg_rows[sg]);
        c.nblk = (c.nrows + 15) >> 4;
        c.blk = 0, c.hs = 0;
    };
    auto advance = [&](TileCursor &c) {  // next tile; at the end of the wave's work the cursor stays on its last tile
        if (c.hs + 1 < TPB) {
            c.hs += 1;
        } else if (c.blk + 1 < c.nblk) {
            c.blk += 1, c.hs = 0;
        } else if (c.sg + 1 < c.s1) {
            seg_of(c, c.sg + 1);
        }
    };
    const int half = lane >> 5, slot = lane & 31;
    auto issue = [&](const TileCursor &c, xf32x4 (&dst)[8]) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const long j = c.row0 + min(16 * c.blk + 2 * t + half, c.nrows - 1);
            const xf32x4 *p = reinterpret_cast<const xf32x4 *>(X + j * K + 128 * c.hs + 4 * slot);
            dst[t] = XNT ? __builtin_nontemporal_load(p) : *p;
        }
    };
    auto issue_b = [&](const TileCursor &c) {
        if (GRAM) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const long j = c.row0 + min(16 * c.blk + 4 * q + v, c.nrows - 1);
                    bnx[nb][v] = B[j * r + bcolc[nb]];
                }
        }
    };

    TileCursor pc;  // prefetch cursor
    pc.s1 = s1;
    seg_of(pc, s0);
    TileCursor cc = pc;  // compute cursor
    issue_b(pc);
#pragma unroll
    for (int s = 0; s < DEPTH; ++s) {
        issue(pc, xr[s]);
        advance(pc);
    }

    long total_rounds = 0;
    for (int sg = s0; sg < s1; ++sg) total_rounds += (long)((__builtin_amdgcn_readfirstlane(seg_rows[sg]) + 15) >> 4) * (TPB / DEPTH);

    constexpr int NCH = (NB == 4) ? 1 : 4;
    xf32x4 acc4[NCH][NB];
#pragma unroll
    for (int kc = 0; kc < NCH; ++kc)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc4[kc][nb] = xf32x4{0.f, 0.f, 0.f, 0.f};
    // per-segment reductions (see k_contract_xc_row)
    double p[NB];
    float pf[NB];
    xf64x4 accG[NB][NB];
    xf32x4 accGf[NB][NB];
    auto seg_reset = [&]() {
#pragma unroll
        for (int a = 0; a < NB; ++a) {
            p[a] = 0.0, pf[a] = 0.f;
#pragma unroll
            for (int b = 0; b < NB; ++b) accG[a][b] = xf64x4{0.0, 0.0, 0.0, 0.0}, accGf[a][b] = xf32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    seg_reset();
    float bcur[NB][4];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int v = 0; v < 4; ++v) bcur[nb][v] = 0.f;

    for (long rnd = 0; rnd < total_rounds; ++rnd) {
        if (GRAM && cc.hs == 0) {  // (wave-uniform; register moves only) the rows of B of the block that starts now
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int v = 0; v < 4; ++v) bcur[nb][v] = bnx[nb][v];
        }
#pragma unroll
        for (int s = 0; s < DEPTH; ++s) {
            // tile `cc` sits in xr[s]: registers -> LDS (row R, logical 16-B slot l -> physical slot l ^ (R & 15))
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int R = 2 * t + half;
                *reinterpret_cast<xf32x4 *>(L + R * 128 + ((slot ^ (R & 15)) << 2)) = xr[s][t];
            }
            // the slot is free: the tile DEPTH ahead (once per round with the rows of B of its block: a fixed number of loads)
            if (s == 0) issue_b(pc);
            issue(pc, xr[s]);
            advance(pc);
            __builtin_amdgcn_sched_barrier(0);
            const int chunk0 = 2 * cc.hs;  // 64-column chunks 2 hs, 2 hs + 1 of C
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) {
                xf32x4 fr[4];
#pragma unroll
                for (int kq = 0; kq < 4; ++kq)
                    fr[kq] = *reinterpret_cast<const xf32x4 *>(L + i16 * 128 + (((16 * kc + 4 * kq + q) ^ i16) << 2));
                xf32x4 cf[4][NB];
#pragma unroll
                for (int kq = 0; kq < 4; ++kq)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) cf[kq][nb] = Cs[(((long)(chunk0 + kc) * 4 + kq) * NB + nb) * 64 + lane];
                // the chain of chunk (2 hs + kc) mod 4: hs = s (mod 4) inside a round (DEPTH is a multiple of 4)
                const int ch = (NCH == 4) ? ((2 * s + kc) & 3) : 0;
#pragma unroll
                for (int kq = 0; kq < 4; ++kq)
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) acc4[ch][nb] = XMFMA16(fr[kq][m], cf[kq][nb][m], acc4[ch][nb]);
            }
            cc.hs += 1;
        }
        if (cc.hs < TPB) continue;
        // ---- epilogue of the 16-row block
        xf32x4 acc[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            if (NCH == 4) acc[nb] = (acc4[0][nb] + acc4[1][nb]) + (acc4[2 % NCH][nb] + acc4[3 % NCH][nb]);
            else acc[nb] = acc4[0][nb];
#pragma unroll
            for (int kc = 0; kc < NCH; ++kc) acc4[kc][nb] = xf32x4{0.f, 0.f, 0.f, 0.f};
        }
        float bv[NB][4];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int col = 16 * nb + i16;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int rl = 16 * cc.blk + 4 * q + v;
                const long j = cc.row0 + rl;
                const bool ok = (rl < cc.nrows) && (col < r);
                if (ok) XC[j * r + col] = acc[nb][v];
                if (GRAM) {
                    const float b = ok ? bcur[nb][v] : 0.f;
                    bv[nb][v] = b;
                    if (GRAM == 2) p[nb] = fma((double)b, (double)acc[nb][v], p[nb]);
                    else pf[nb] = fmaf(b, acc[nb][v], pf[nb]);
                }
            }
        }
        if (GRAM) {
#pragma unroll
            for (int v = 0; v < 4; ++v)
#pragma unroll
                for (int a = 0; a < NB; ++a)
#pragma unroll
                    for (int b = 0; b < NB; ++b) {
                        if (GRAM == 2)
                            accG[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)bv[a][v], (double)bv[b][v], accG[a][b], 0, 0, 0);
                        else
                            accGf[a][b] = XMFMA16(bv[a][v], bv[b][v], accGf[a][b]);
                    }
        }
        cc.hs = 0;
        cc.blk += 1;
        if (cc.blk < cc.nblk) continue;
        // ---- end of the segment
        if (GRAM) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                double t = (GRAM == 2) ? p[nb] : (double)pf[nb];
                t += __shfl_xor(t, 16);
                t += __shfl_xor(t, 32);
                const int col = 16 * nb + i16;
                if (q == 0 && col < r) seg_rhs[(long)cc.sg * r + col] = t;
            }
#pragma unroll
            for (int a = 0; a < NB; ++a)
#pragma unroll
                for (int b = 0; b < NB; ++b)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int ra = 16 * a + ((GRAM == 2) ? q + 4 * v : 4 * q + v), cb = 16 * b + i16;
                        const double val = (GRAM == 2) ? accG[a][b][v] : (double)accGf[a][b][v];
                        if (ra < r && cb < r) seg_btb[((long)cc.sg * r + ra) * r + cb] = val;
                    }
            seg_reset();
        }
        if (cc.sg + 1 < s1) seg_of(cc, cc.sg + 1);
    }
}

}  // namespace

// Launches the LDS-resident-C form of the X C pass when the shape allows it; returns 1 when launched, 0 otherwise.
// gram: 0 X C only, 1 fused per-segment reductions in fp32 chains, 2 in fp64.  MCL_XC_LDS_DEPTH=4 / 8 overrides the ring depth.
int mcl_try_contract_xc_lds(mcl_context *c, int gram) {
    if (c->sw.no_xc_lds || c->NB > 2 || (c->K % 512) != 0 || (c->K % 4) != 0) return 0;
    if ((reinterpret_cast<uintptr_t>(c->X) & 15) != 0) return 0;
    const size_t cf_bytes = (size_t)(c->K >> 6) * 4 * c->NB * 256 * sizeof(float);
    const size_t sm = cf_bytes + sizeof(float) * 4 * 16 * 128;
    if (sm > 160 * 1024) return 0;
    const int n_segs = c->segs.n_tiles;
    if (n_segs == 0) return 0;
    if (mcl_cfrag_chunks(c) != (int)(c->K >> 6)) return 0;  // (the image the kernel copies is exactly K / 64 chunks)
    const unsigned g = (unsigned)((c->n_seg_waves + 3) / 4);
    // tiles in flight per wave: four (32 KB).  Eight (MCL_XC_LDS_DEPTH=8, K % 1024 == 0) measured 13.68 against 13.39 ms at
    // config 5: with four the pass no longer waits for bytes in flight
    int depth = 4;
    if (c->sw.xc_lds_depth == 4 || c->sw.xc_lds_depth == 8) depth = (c->K % (128 * c->sw.xc_lds_depth) == 0) ? c->sw.xc_lds_depth : depth;
#define MCL_XCL__(NB_, GRAM_, NT_, D_)                                                                                    \
    do {                                                                                                                  \
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_contract_xc_lds<NB_, GRAM_, NT_, D_>),                    \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm) != hipSuccess) {                      \
            (void)hipGetLastError();                                                                                      \
            return 0;                                                                                                     \
        }                                                                                                                 \
        hipLaunchKernelGGL((k_contract_xc_lds<NB_, GRAM_, NT_, D_>), dim3(g), dim3(256), sm, c->stream, c->X, c->Cfrag,     \
                           c->XC, c->B, c->segs.row0, c->segs.nrows, c->wave_seg_ptr, c->n_seg_waves, (int)c->K, c->r,      \
                           c->seg_rhs, c->seg_btb);                                                                       \
    } while (0)
#define MCL_XCL_(NB_, GRAM_, NT_)                                \
    do {                                                         \
        if (depth == 8) MCL_XCL__(NB_, GRAM_, NT_, 8);           \
        else MCL_XCL__(NB_, GRAM_, NT_, 4);                      \
    } while (0)
#define MCL_XCL(NB_, GRAM_)                                      \
    do {                                                         \
        if (c->x_streams) MCL_XCL_(NB_, GRAM_, true);            \
        else MCL_XCL_(NB_, GRAM_, false);                        \
    } while (0)
    if (c->NB == 1) {
        if (gram == 2) MCL_XCL(1, 2);
        else if (gram == 1) MCL_XCL(1, 1);
        else MCL_XCL(1, 0);
    } else {
        if (gram == 2) MCL_XCL(2, 2);
        else if (gram == 1) MCL_XCL(2, 1);
        else MCL_XCL(2, 0);
    }
#undef MCL_XCL
#undef MCL_XCL_
#undef MCL_XCL__
    return hipGetLastError() == hipSuccess ? 1 : 0;
}
