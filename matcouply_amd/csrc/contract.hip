// The two passes over X per outer AO-ADMM iteration, as fp32-input MFMA kernels for gfx950 (CDNA4).
//
//   k_contract_xt : [G | R] partials,  R = sum_i X_i^T (B_i o a_i),  G = sum_i (B_i o a_i)^T (B_i o a_i)
//                   (reference: decomposition.py:312-315, sites C1 of SURVEY.md 2.3)
//   k_contract_xc : XC = X C  (rows of all slabs at once; reference: decomposition.py:147-152 and,
//                   through rhs_i = (X_i C) o a_i, decomposition.py:242)
//   k_slab_gram   : rhs_i = diag(B_i^T X_i C), B_i^T B_i per slab (decomposition.py:155-158)
//
// Both X passes are HBM-bound (r/2 flop per byte); v_mfma_f32_16x16x4_f32 does the contraction so that the
// VALU stays free for address/guard work, and every global load of X is a 16-byte-per-lane load of
// 256-byte row segments (4 rows per wave instruction).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <type_traits>

#include "mcl_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

static __device__ __forceinline__ f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }

// ---------------------------------------------------------------------------------------------------------
// k_contract_xt
//   Work unit = SEGMENT: <= 256 consecutive packed rows of ONE slab (host-built table), so a_i is loaded once per
//   segment and the pipelined inner loop over 4-row groups is branch-free.
//   lane l = (rsub = l>>4, c16 = l&15).  One "group" = 4 consecutive rows; lane loads
//   X[row][kbase + 64kb + 4c16 .. +3] (a 256-B row segment per 16 lanes) for kb < KB.
//   MFMA (kb, m): A-operand = component m of that float4  -> output row index i <-> k = kbase+64kb+4i+m,
//                 B-operand = (B o a)[row][16nb + c16]; reduction index (l>>4) <-> the 4 rows of the group.
//   Accumulator (kb, m, nb), lane l, reg v  <->  R[kbase + 64kb + 4(4(l>>4)+v) + m][16nb + (l&15)].
// Every load in the inner loop is UNCONDITIONAL (row/column indices clamped into the segment / matrix) so that
// the compiler keeps DEPTH groups in flight with counted s_waitcnt vmcnt(N): rows past the segment end get a
// zero B-operand, columns past K land in accumulator rows that are never written out.
// ---------------------------------------------------------------------------------------------------------
// MODE 0: R slice of blockIdx.y, and G in the blocks with blockIdx.y == 0.  Rank > 32 (NB = 4) has no registers for both
// accumulator sets with their fp64 shadows: MODE 1 (R only) and MODE 2 (G only, one launch with gridDim.y = 1, X untouched).
template <int KB, int NB, int VEC, int DEPTH, int MODE, bool XNT = false>  // XNT: non-temporal loads of X (X >> last-level cache)
__global__ __launch_bounds__(256) void k_contract_xt(const float *__restrict__ X, const float *__restrict__ B,
                                                     const float *__restrict__ A, const int *__restrict__ seg_slab,
                                                     const int *__restrict__ seg_row0, const int *__restrict__ seg_rows,
                                                     const int *__restrict__ wave_seg_ptr, int n_waves, int K, int r,
                                                     double *__restrict__ part, int part_stride, int dbg, int n_slices,
                                                     int n_rowblocks) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rsub = lane >> 4, c16 = lane & 15;
    // XCD-aware mapping (round 3).  The K-slices of one row range all need the same rows of B (and a_i); with the row
    // range as the fastest grid index (round 2) every slice was a pass of its own over all the rows and config 5 fetched
    // B eight times (86.2 GB for 68.7 GB of X, profiles/r2_c5_pmc_traffic.json).  Workgroup ids go round-robin over the
    // 8 XCDs, each with its own L2: id = 8 m + x runs on XCD x.  Here x = row range mod 8 and the slices are
    // consecutive m, so the `n_slices` workgroups of a row range sit on ONE XCD, are dispatched back to back and share
    // their rows of B through that XCD's L2; the 4 KB rows of X at K = 1024 are read as a whole at about the same time.
    const int xcd = blockIdx.x & 7, m = blockIdx.x >> 3;
    const int bslice = m % n_slices, brow = (m / n_slices) * 8 + xcd;
    if (brow >= n_rowblocks) return;
    const int kbase = bslice * (64 * KB);
    const int w = brow * 4 + wave;
    // the wave's segments: a contiguous range with (nearly) the same number of 16-row blocks in every wave
    // (mcl_set_problem balances ragged slabs); waves past the end have an empty range
    const int s0 = wave_seg_ptr[min(w, n_waves)];
    const int s1 = wave_seg_ptr[min(w + 1, n_waves)];
    constexpr bool DO_R = MODE != 2;
    constexpr int NG_ = (MODE == 1) ? 1 : NB;  // extent of the G accumulator arrays
    const bool doG = (MODE == 2) || (MODE == 0 && bslice == 0);

    int kcol[KB];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        int kc = kbase + 64 * kb + 4 * c16;
        if (VEC == 4) kc = min(kc, K - 4);  // clamped: results for k >= K are discarded
        kcol[kb] = kc;
    }
    int bcol[NB];
    bool bok[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        bok[nb] = (16 * nb + c16) < r;
        bcol[nb] = min(16 * nb + c16, r - 1);
    }

    // fp32 MFMA chains end with their SEGMENT (<= 256 rows, shorter on small problems: mcl_set_problem): the fp32
    // accumulators are then added into fp64 shadows and cleared, so the rounding of a chain is relative to one segment's
    // worth of products, not to the running total over all the wave's segments - the normal equations of a penalty-free
    // mode amplify every relative error of [G | R] by their condition number (decomposition.py:307-331).  With s rows
    // per segment and N rows in all the fp64 total keeps ~ 3e-8 s / sqrt(3 N) relative rounding.
    f32x4 acc[KB][4][NB];
    f32x4 accG[NG_][NG_];
    f64x4 dacc[KB][4][NB];
    f64x4 daccG[NG_][NG_];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[kb][m][nb] = zero4(), dacc[kb][m][nb] = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int a = 0; a < NG_; ++a)
#pragma unroll
        for (int b = 0; b < NG_; ++b) accG[a][b] = zero4(), daccG[a][b] = f64x4{0.0, 0.0, 0.0, 0.0};
    auto flush = [&]() {
        if (DO_R) {
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
                        for (int v = 0; v < 4; ++v) dacc[kb][m][nb][v] += (double)acc[kb][m][nb][v];
                        acc[kb][m][nb] = zero4();
                    }
        }
        if (MODE != 1 && doG) {
#pragma unroll
            for (int a = 0; a < NG_; ++a)
#pragma unroll
                for (int b = 0; b < NG_; ++b) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) daccG[a][b][v] += (double)accG[a][b][v];
                    accG[a][b] = zero4();
                }
        }
    };

    for (int sg = s0; sg < s1; ++sg) {
        const int slab = __builtin_amdgcn_readfirstlane(seg_slab[sg]);
        const long row0 = __builtin_amdgcn_readfirstlane(seg_row0[sg]);
        const int nrows = __builtin_amdgcn_readfirstlane(seg_rows[sg]);
        const int ng = (nrows + 3) >> 2;
        float a_val[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) a_val[nb] = bok[nb] ? A[(long)slab * r + bcol[nb]] : 0.f;

        f32x4 fx[DEPTH][KB];
        float fb[DEPTH][NB], fa[DEPTH][NB];
        auto load = [&](int d, int g) {
            const int rl = 4 * g + rsub;
            const long j = row0 + min(rl, nrows - 1);
#pragma unroll
            for (int kb = 0; kb < (DO_R ? KB : 0); ++kb) {
                if (VEC == 4) {
                    fx[d][kb] = XNT ? __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(X + j * K + kcol[kb]))
                                    : *reinterpret_cast<const f32x4 *>(X + j * K + kcol[kb]);
                } else {
                    f32x4 v;
#pragma unroll
                    for (int m = 0; m < 4; ++m) v[m] = X[j * K + min(kcol[kb] + m, K - 1)];
                    fx[d][kb] = v;
                }
            }
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                fb[d][nb] = B[j * r + bcol[nb]];
                fa[d][nb] = (rl < nrows) ? a_val[nb] : 0.f;
            }
        };
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) load(d, d);
        for (int g = 0; g < ng; g += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                float ba[NB];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) ba[nb] = fb[d][nb] * fa[d][nb];
                // (kept on purpose: besides the loads-only timing experiment, this never-taken branch pins the order
                // "MFMAs of the slot, then its next loads" - with it removed the scheduler interleaves the loads with the
                // MFMAs and the counted vmcnt waits drop from 12..15 to 8..12: 6 % slower at K = 1024, measured)
                if (DO_R && (dbg & 1)) {  // timing experiment: loads only
#pragma unroll
                    for (int kb = 0; kb < KB; ++kb) acc[kb][0][0] += fx[d][kb] * ba[0];
                    load(d, g + DEPTH + d);
                    continue;
                }
                if (DO_R) {
#pragma unroll
                    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                        for (int m = 0; m < 4; ++m)
#pragma unroll
                            for (int nb = 0; nb < NB; ++nb) acc[kb][m][nb] = MFMA16(fx[d][kb][m], ba[nb], acc[kb][m][nb]);
                }
                if (MODE != 1 && doG) {
#pragma unroll
                    for (int a = 0; a < NG_; ++a)
#pragma unroll
                        for (int b = 0; b < NG_; ++b) accG[a][b] = MFMA16(ba[a], ba[b], accG[a][b]);
                }
                load(d, g + DEPTH + d);
            }
        }
        flush();  // once per segment, outside the pipelined loop (a branch inside it collapses the counted vmcnt waits)
    }

    // deterministic cross-wave reduction through LDS (fp64), then one fp64 partial slab per block
    constexpr int W = 16 * NB;
    constexpr int LR = DO_R ? 64 * KB * W : 0, LG = (MODE != 1) ? W * W : 0;
    __shared__ double lds[LR + LG];
    double *ldsG = lds + LR;
    for (int wv = 0; wv < 4; ++wv) {
        if (wave == wv) {
#pragma unroll
            for (int kb = 0; kb < (DO_R ? KB : 0); ++kb)
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            const int kl = 64 * kb + 4 * (4 * rsub + v) + m;
                            const int idx = kl * W + 16 * nb + c16;
                            lds[idx] = (wv == 0) ? dacc[kb][m][nb][v] : lds[idx] + dacc[kb][m][nb][v];
                        }
            if (MODE != 1 && doG) {
#pragma unroll
                for (int a = 0; a < NG_; ++a)
#pragma unroll
                    for (int b = 0; b < NG_; ++b)
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            const int idx = (16 * a + 4 * rsub + v) * W + 16 * b + c16;
                            ldsG[idx] = (wv == 0) ? daccG[a][b][v] : ldsG[idx] + daccG[a][b][v];
                        }
            }
        }
        __syncthreads();
    }
    double *out = part + (long)brow * part_stride;
    for (int e = threadIdx.x; e < LR; e += 256) {
        const int kl = e / W, n = e - kl * W;
        const int k = kbase + kl;
        if (k < K && n < r) out[r * r + k * r + n] = lds[e];
    }
    if (MODE != 1 && doG) {
        for (int e = threadIdx.x; e < W * W; e += 256) {
            const int a = e / W, b = e - a * W;
            if (a < r && b < r) out[a * r + b] = ldsG[e];
        }
    }
}

// GR[e] = sum_p part[p][e]; fixed summation order (deterministic, identical on every rank for identical input)
__global__ __launch_bounds__(256) void k_reduce_partials(const double *__restrict__ part, int n_part, int E,
                                                         double *__restrict__ out) {
    __shared__ double sm[4][64];
    const int el = threadIdx.x & 63, pc = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + el;
    double s = 0.0;
    if (e < E) {
        int p = pc;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        for (; p + 12 < n_part; p += 16) {
            s0 += part[(long)p * E + e];
            s1 += part[(long)(p + 4) * E + e];
            s2 += part[(long)(p + 8) * E + e];
            s3 += part[(long)(p + 12) * E + e];
        }
        for (; p < n_part; p += 4) s0 += part[(long)p * E + e];
        s = (s0 + s1) + (s2 + s3);
    }
    sm[pc][el] = s;
    __syncthreads();
    if (pc == 0 && e < E) out[e] = (sm[0][el] + sm[1][el]) + (sm[2][el] + sm[3][el]);
}

// ---------------------------------------------------------------------------------------------------------
// C in MFMA-fragment order for k_contract_xc:
//   Cfrag[(((kc*4 + kq)*NB + nb)*64 + lane)*4 + m] = C[64kc + 16kq + 4(lane>>4) + m][16nb + (lane&15)]  (0 outside)
// ---------------------------------------------------------------------------------------------------------
__global__ void k_build_cfrag(const float *__restrict__ C, int K, int r, int KC, int NB, float *__restrict__ Cfrag) {
    const long total = (long)KC * 4 * NB * 256;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int m = idx & 3, lane = (idx >> 2) & 63;
    long t = idx >> 8;
    const int nb = t % NB;
    t /= NB;
    const int kq = t & 3, kc = t >> 2;
    const int k = 64 * kc + 16 * kq + 4 * (lane >> 4) + m, col = 16 * nb + (lane & 15);
    Cfrag[idx] = (k < K && col < r) ? C[(long)k * r + col] : 0.f;
}

// ---------------------------------------------------------------------------------------------------------
// k_contract_xc : XC = X C.  A wave owns blocks of 16 packed rows and walks K in chunks of 64 columns.
//   global -> registers: lane (q = l>>4, i16 = l&15), t<4: X[j0 + 4t + q][64kc + 4 i16 .. +3]  (256-B row segments)
//   registers -> LDS   : wave-private 16 x 64 fp32 tile, 16-B slot index XORed with the row (conflict-free)
//   LDS -> fragments   : lane (q, i16), kq<4: X[j0 + i16][64kc + 16kq + 4q .. +3]
//   MFMA (kq, m)       : A = component m (row i16, k = 64kc+16kq+4q+m), B = Cfrag(kc,kq,nb)[m]
//   accumulator nb, lane l, reg v <-> XC[j0 + 4(l>>4) + v][16nb + (l&15)]
// The (block, chunk) steps of a wave are flattened and software-pipelined through a 4-slot register ring:
// loads are UNCONDITIONAL (addresses clamped; columns >= K meet zero C fragments, rows >= N are never stored),
// so 4 chunks (16 KB per wave) stay in flight under counted s_waitcnt vmcnt(N).
// KCT in {2, 4}: K <= 64*KCT, C fragments live in registers.  KCT == 0: runtime chunk count (multiple of 4, the
// host pads Cfrag with zeros), fragments re-read from the L1/L2-resident Cfrag buffer each chunk.
// ---------------------------------------------------------------------------------------------------------
template <int NB, int VEC, int KCT>
__global__ __launch_bounds__(256) void k_contract_xc(const float *__restrict__ X, const float *__restrict__ Cfrag,
                                                     float *__restrict__ XC, long N, int K, int r, int KCrt,
                                                     long blocks_per_wave, long n_blocks16, int dbg) {
    __shared__ float lds_all[4][16 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, i16 = lane & 15;
    float *L = lds_all[wave];
    const int KC = (KCT > 0) ? KCT : KCrt;
    const long w = (long)blockIdx.x * 4 + wave;
    const long b0 = w * blocks_per_wave;
    long b1 = b0 + blocks_per_wave;
    if (b1 > n_blocks16) b1 = n_blocks16;
    if (b0 >= b1) return;

    constexpr int CR = (KCT > 0) ? KCT : 1;
    f32x4 creg[CR][4][NB];
    if (KCT > 0) {
#pragma unroll
        for (int kc = 0; kc < CR; ++kc)
#pragma unroll
            for (int kq = 0; kq < 4; ++kq)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    creg[kc][kq][nb] =
                        *reinterpret_cast<const f32x4 *>(Cfrag + ((((long)kc * 4 + kq) * NB + nb) * 64 + lane) * 4);
    }

    f32x4 xr[4][4];
    // stage the 16 x 64 chunk (blk, kc) into ring slot `slot`
    auto issue = [&](int slot, long blk, int kc) {
        const long bc = min(blk, b1 - 1);
        int col = 64 * kc + 4 * i16;
        if (VEC == 4) col = min(col, K - 4);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const long j = min(bc * 16 + 4 * t + q, N - 1);
            if (VEC == 4) {
                xr[slot][t] = *reinterpret_cast<const f32x4 *>(X + j * K + col);
            } else {
                f32x4 v;
#pragma unroll
                for (int m = 0; m < 4; ++m) v[m] = X[j * K + min(col + m, K - 1)];
                xr[slot][t] = v;
            }
        }
    };

    f32x4 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[nb] = zero4();
    auto step = [&](int slot, long blk, int kc, long nblk, int nkc, const f32x4 (&cf)[4][NB]) {
        // registers -> LDS (row = 4t + q, slot = i16 ^ row)
        f32x4 fr[4];
        if (dbg & 2) {  // timing experiment: no LDS staging (wrong results)
#pragma unroll
            for (int kq = 0; kq < 4; ++kq) fr[kq] = xr[slot][kq];
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int row = 4 * t + q;
                *reinterpret_cast<f32x4 *>(L + row * 64 + ((i16 ^ row) << 2)) = xr[slot][t];
            }
#pragma unroll
            for (int kq = 0; kq < 4; ++kq)
                fr[kq] = *reinterpret_cast<const f32x4 *>(L + i16 * 64 + (((4 * kq + q) ^ i16) << 2));
        }
        issue(slot, nblk, nkc);
        if (dbg & 1) {  // timing experiment: no MFMA
#pragma unroll
            for (int kq = 0; kq < 4; ++kq) acc[0] += fr[kq] * cf[kq][0];
        } else {
#pragma unroll
            for (int kq = 0; kq < 4; ++kq)
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) acc[nb] = MFMA16(fr[kq][m], cf[kq][nb][m], acc[nb]);
        }
        if (kc == KC - 1) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const long j = blk * 16 + 4 * q + v;
                    const int col = 16 * nb + i16;
                    if (j < N && col < r) XC[j * r + col] = acc[nb][v];
                }
                acc[nb] = zero4();
            }
        }
    };

    if (KCT > 0) {
        // 4 flattened steps per trip: (blk + d / KCT, d % KCT)
        constexpr int BPT = 4 / CR;  // blocks per trip
#pragma unroll
        for (int d = 0; d < 4; ++d) issue(d, b0 + d / CR, d % CR);
        for (long blk = b0; blk < b1; blk += BPT) {
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const long cb = blk + d / CR;
                if (cb < b1) step(d, cb, d % CR, cb + BPT, d % CR, creg[d % CR]);
            }
        }
    } else {
        // KCrt is a multiple of 4
#pragma unroll
        for (int d = 0; d < 4; ++d) issue(d, b0, d);
        for (long blk = b0; blk < b1; ++blk) {
            for (int kc = 0; kc < KC; kc += 4) {
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    f32x4 cf[4][NB];
#pragma unroll
                    for (int kq = 0; kq < 4; ++kq)
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
                            cf[kq][nb] = *reinterpret_cast<const f32x4 *>(
                                Cfrag + ((((long)(kc + d) * 4 + kq) * NB + nb) * 64 + lane) * 4);
                    int nk = kc + d + 4;
                    long nblk = blk;
                    if (nk >= KC) {
                        nk -= KC;
                        nblk += 1;
                    }
                    step(d, blk, kc + d, nblk, nk, cf);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// k_contract_xc_row : XC = X C for K % 256 == 0, optionally FUSED with the per-slab A-phase reductions.
// Same MFMA mapping as k_contract_xc, but the global access is ROW-CONTIGUOUS: a wave stages a 16-row x 256-column
// super-chunk with 16 wave-loads of ONE 1 KB row segment each (lane l reads columns 4l..4l+3), i.e. the access
// geometry of a plain streaming copy; for K = 256 the 16 loads cover one contiguous 16 KB region.  The tile lives
// in a wave-private 16 KB LDS image whose 16-B slot index is XORed with the row (conflict-free ds_write_b128 /
// ds_read_b128).  The next super-chunk's 16 loads are in flight while the current one is multiplied.
// Work unit = SEGMENT (<= 256 rows of one slab).  GRAM: while the XC block is still in the accumulators
// (lane (q, i16) holds rows 4q+v, column i16), the same-layout block of B is loaded and
//     rhs_seg[c]  += sum_rows B[row][c] * XC[row][c]           (decomposition.py:147-158: diag(B_i^T X_i C))
//     BtB_seg     += B_blk^T B_blk   (4 MFMAs: A = B-operand = b[v], reduction index <-> the 4 lane quarters)
// are accumulated per segment; k_A_finish sums the segments of its slab.  This removes the separate
// k_slab_gram pass (a 2 S_B re-read and a launch).
// CREG: K == 256 and NB == 1: the 64 C-fragment registers stay resident.
// ---------------------------------------------------------------------------------------------------------
template <int NB, bool CREG, int GRAM, bool XNT = false>
__global__ __launch_bounds__(256) void k_contract_xc_row(const float *__restrict__ X, const float *__restrict__ Cfrag,
                                                         float *__restrict__ XC, const float *__restrict__ B,
                                                         const int *__restrict__ seg_row0,
                                                         const int *__restrict__ seg_rows,
                                                         const int *__restrict__ wave_seg_ptr, int n_waves, int K, int r,
                                                         double *__restrict__ seg_rhs, double *__restrict__ seg_btb) {
    extern __shared__ float lds_dyn[];  // 4 waves x 16 rows x 256 floats
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, i16 = lane & 15;
    float *L = lds_dyn + wave * (16 * 256);
    const int SC = K >> 8;  // super-chunks per row block
    const int w = blockIdx.x * 4 + wave;
    if (w >= n_waves) return;
    const int s0 = wave_seg_ptr[w], s1 = wave_seg_ptr[w + 1];
    if (s0 >= s1) return;

    constexpr int CR = CREG ? 4 : 1;
    constexpr bool CPRE = !CREG && NB <= 2;  // 64 NB registers of C fragments per super-chunk
    f32x4 creg[CR][4][NB];
    if (CREG) {
#pragma unroll
        for (int kc = 0; kc < 4; ++kc)
#pragma unroll
            for (int kq = 0; kq < 4; ++kq)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    creg[kc][kq][nb] =
                        *reinterpret_cast<const f32x4 *>(Cfrag + ((((long)kc * 4 + kq) * NB + nb) * 64 + lane) * 4);
    }

    f32x4 xr[16];
    float bnx[NB][4];  // B block (rows 4q+v, column 16nb+i16) of the block whose X loads are in flight
    int bcolc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) bcolc[nb] = min(16 * nb + i16, r - 1);
    // stage rows [row0 + 16 blk, +16) x columns [256 sc, +256) of segment (row0, nrows); rows clamped into the segment
    auto issue = [&](long row0, int nrows, int blk, int sc) {
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const long j = row0 + min(16 * blk + t, nrows - 1);
            xr[t] = XNT ? __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(X + j * K + 256 * sc + 4 * lane))
                        : *reinterpret_cast<const f32x4 *>(X + j * K + 256 * sc + 4 * lane);
        }
        if (GRAM && sc == 0) {  // unconditional clamped loads; masked at use
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const long j = row0 + min(16 * blk + 4 * q + v, nrows - 1);
                    bnx[nb][v] = B[j * r + bcolc[nb]];
                }
        }
    };

    // four independent fp32 chains per output (one per 64-column chunk of a super-chunk), summed pairwise at the end of
    // the block: the rounding of a K-long dot product grows with the chain length, and the B right-hand sides feed
    // every later phase of the iteration
    constexpr int NCH = (NB == 4) ? 1 : 4;  // rank > 32 has no registers for more than one chain
    f32x4 acc4[NCH][NB];
#pragma unroll
    for (int kc = 0; kc < NCH; ++kc)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc4[kc][nb] = zero4();

    long row0 = __builtin_amdgcn_readfirstlane(seg_row0[s0]);
    int nrows = __builtin_amdgcn_readfirstlane(seg_rows[s0]);
    issue(row0, nrows, 0, 0);
    for (int sg = s0; sg < s1; ++sg) {
        const int nblk = (nrows + 15) >> 4;
        // next segment (for the prefetch across the segment boundary)
        long nrow0 = row0;
        int nnrows = nrows;
        if (sg + 1 < s1) {
            nrow0 = __builtin_amdgcn_readfirstlane(seg_row0[sg + 1]);
            nnrows = __builtin_amdgcn_readfirstlane(seg_rows[sg + 1]);
        }
        // per-segment reductions.  GRAM == 2 (penalty-free A: its systems are not shifted and amplify every relative
        // error of these sums): fp64 throughout - the products b * xc and b * b' of fp32 values are exact in fp64, so the
        // only rounding left in rhs_i and B_i^T B_i is the fp32 rounding of X C itself.  GRAM == 1 (penalised A): fp32
        // chains over the segment's <= 256 rows (4 fp32 MFMAs per block instead of 4 NB^2 fp64 ones at twice the cycles:
        // 12 % of the kernel at rank 32), widened to fp64 when the segment is stored.
        double p[NB];
        float pf[NB];
        f64x4 accG[NB][NB];
        f32x4 accGf[NB][NB];
#pragma unroll
        for (int a = 0; a < NB; ++a) {
            p[a] = 0.0, pf[a] = 0.f;
#pragma unroll
            for (int b = 0; b < NB; ++b) accG[a][b] = f64x4{0.0, 0.0, 0.0, 0.0}, accGf[a][b] = zero4();
        }
        for (int blk = 0; blk < nblk; ++blk) {
            float bcur[NB][4];
            if (GRAM) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int v = 0; v < 4; ++v) bcur[nb][v] = bnx[nb][v];
            }
            for (int sc = 0; sc < SC; ++sc) {
                // K > 256: the C fragments of this super-chunk are loaded BEFORE the X prefetch is issued, so the wait
                // on them leaves the prefetch in flight (memory returns in order: fragment loads issued after it
                // would drain it in every kc step - the K = 1024 pass ran at 2.4 TB/s that way)
                f32x4 cpre[CPRE ? 4 : 1][4][NB];
                if (CPRE) {
#pragma unroll
                    for (int kc = 0; kc < 4; ++kc)
#pragma unroll
                        for (int kq = 0; kq < 4; ++kq)
#pragma unroll
                            for (int nb = 0; nb < NB; ++nb)
                                cpre[kc][kq][nb] = *reinterpret_cast<const f32x4 *>(
                                    Cfrag + ((((long)(4 * sc + kc) * 4 + kq) * NB + nb) * 64 + lane) * 4);
                    __builtin_amdgcn_sched_barrier(0);
                }
                // registers -> LDS: row t, logical 16-B slot = lane, physical slot = lane ^ t
#pragma unroll
                for (int t = 0; t < 16; ++t) *reinterpret_cast<f32x4 *>(L + t * 256 + ((lane ^ t) << 2)) = xr[t];
                // prefetch the next super-chunk (possibly the first one of the next segment)
                if (sc + 1 < SC) issue(row0, nrows, blk, sc + 1);
                else if (blk + 1 < nblk) issue(row0, nrows, blk + 1, 0);
                else issue(nrow0, nnrows, 0, 0);
                if (CPRE) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kc = 0; kc < 4; ++kc) {
                    f32x4 fr[4];
#pragma unroll
                    for (int kq = 0; kq < 4; ++kq)
                        fr[kq] = *reinterpret_cast<const f32x4 *>(L + i16 * 256 + (((16 * kc + 4 * kq + q) ^ i16) << 2));
                    f32x4 cf[4][NB];
#pragma unroll
                    for (int kq = 0; kq < 4; ++kq)
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) {
                            if (CREG) cf[kq][nb] = creg[kc][kq][nb];
                            else if (CPRE) cf[kq][nb] = cpre[kc][kq][nb];
                            else
                                cf[kq][nb] = *reinterpret_cast<const f32x4 *>(
                                    Cfrag + ((((long)(4 * sc + kc) * 4 + kq) * NB + nb) * 64 + lane) * 4);
                        }
#pragma unroll
                    for (int kq = 0; kq < 4; ++kq)
#pragma unroll
                        for (int m = 0; m < 4; ++m)
#pragma unroll
                            for (int nb = 0; nb < NB; ++nb) acc4[kc % NCH][nb] = MFMA16(fr[kq][m], cf[kq][nb][m], acc4[kc % NCH][nb]);
                }
            }
            // epilogue of the 16-row block
            f32x4 acc[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                if (NCH == 4) acc[nb] = (acc4[0][nb] + acc4[1][nb]) + (acc4[2 % NCH][nb] + acc4[3 % NCH][nb]);
                else acc[nb] = acc4[0][nb];
#pragma unroll
                for (int kc = 0; kc < NCH; ++kc) acc4[kc][nb] = zero4();
            }
            float bv[NB][4];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int col = 16 * nb + i16;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int rl = 16 * blk + 4 * q + v;
                    const long j = row0 + rl;
                    const bool ok = (rl < nrows) && (col < r);
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
                                accGf[a][b] = MFMA16(bv[a][v], bv[b][v], accGf[a][b]);
                        }
            }
        }
        if (GRAM) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                double t = (GRAM == 2) ? p[nb] : (double)pf[nb];
                t += __shfl_xor(t, 16);
                t += __shfl_xor(t, 32);
                const int col = 16 * nb + i16;
                if (q == 0 && col < r) seg_rhs[(long)sg * r + col] = t;
            }
#pragma unroll
            for (int a = 0; a < NB; ++a)
#pragma unroll
                for (int b = 0; b < NB; ++b)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        // D layouts: f64 MFMA row = (l >> 4) + 4 reg, f32 MFMA row = 4 (l >> 4) + reg; col = l & 15
                        const int ra = 16 * a + ((GRAM == 2) ? q + 4 * v : 4 * q + v), cb = 16 * b + i16;
                        const double val = (GRAM == 2) ? accG[a][b][v] : (double)accGf[a][b][v];
                        if (ra < r && cb < r) seg_btb[((long)sg * r + ra) * r + cb] = val;
                    }
        }
        row0 = nrow0;
        nrows = nnrows;
    }
}

// ---------------------------------------------------------------------------------------------------------
// k_contract_xc_256 : the K = 256, rank <= 16 case of k_contract_xc_row (C fragments resident in 64 registers) with TWO
// 16-row blocks of X in flight per wave instead of one (32 KB per wave, 128 KB per CU): a wave walks the 16-row blocks of
// all its segments as one flat sequence through a 2-slot register ring; block b + 2 is requested as soon as block b sits
// in the LDS tile.  Same arithmetic, tile layout, epilogue and per-segment outputs as k_contract_xc_row.  Config 4:
// 180 -> 164 us (a third slot: 164 us - two blocks cover the latency; MCL_XC_DEPTH1=1 selects the one-slot kernel).
// The flat walk: `cur` is the block being multiplied, `pre` the one being requested (two blocks ahead); past the wave's last
// block `pre` keeps pointing at the last valid rows (unconditional clamped loads, nothing stored).
// ---------------------------------------------------------------------------------------------------------
template <int GRAM, int D, bool XNT = false>
__global__ __launch_bounds__(256) void k_contract_xc_256(const float *__restrict__ X, const float *__restrict__ Cfrag,
                                                         float *__restrict__ XC, const float *__restrict__ B,
                                                         const int *__restrict__ seg_row0,
                                                         const int *__restrict__ seg_rows,
                                                         const int *__restrict__ wave_seg_ptr, int n_waves, int r,
                                                         double *__restrict__ seg_rhs,
                                                         double *__restrict__ seg_btb) {
    extern __shared__ float lds_dyn[];  // 4 waves x 16 rows x 256 floats
    constexpr int K = 256;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, i16 = lane & 15;
    float *L = lds_dyn + wave * (16 * 256);
    const int w = blockIdx.x * 4 + wave;
    if (w >= n_waves) return;
    const int s0 = wave_seg_ptr[w], s1 = wave_seg_ptr[w + 1];
    if (s0 >= s1) return;

    f32x4 creg[4][4];
#pragma unroll
    for (int kc = 0; kc < 4; ++kc)
#pragma unroll
        for (int kq = 0; kq < 4; ++kq)
            creg[kc][kq] = *reinterpret_cast<const f32x4 *>(Cfrag + (((long)kc * 4 + kq) * 64 + lane) * 4);
    const int bcol = min(i16, r - 1);

    struct Cursor {
        int sg, blk, nblk, nrows;
        long row0;
    };
    auto seg_at = [&](Cursor &c, int sg) {
        c.sg = sg, c.blk = 0;
        c.row0 = __builtin_amdgcn_readfirstlane(seg_row0[sg]);
        c.nrows = __builtin_amdgcn_readfirstlane(seg_rows[sg]);
        c.nblk = (c.nrows + 15) >> 4;
    };
    auto advance = [&](Cursor &c) {  // wave-uniform
        if (c.blk + 1 < c.nblk) {
            c.blk += 1;
        } else if (c.sg + 1 < s1) {
            seg_at(c, c.sg + 1);
        } else {
            c.blk = c.nblk;  // past the end: rows clamp to the last valid one
        }
    };
    int total = 0;
    for (int sg = s0; sg < s1; ++sg) total += (__builtin_amdgcn_readfirstlane(seg_rows[sg]) + 15) >> 4;

    f32x4 xr[D][16];
    float bnx[D][4];
    auto issue = [&](auto dc, const Cursor &c) {
        constexpr int d = decltype(dc)::value;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const long j = c.row0 + min(16 * c.blk + t, c.nrows - 1);
            xr[d][t] = XNT ? __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(X + j * K + 4 * lane))
                           : *reinterpret_cast<const f32x4 *>(X + j * K + 4 * lane);
        }
        if (GRAM) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const long j = c.row0 + min(16 * c.blk + 4 * q + v, c.nrows - 1);
                bnx[d][v] = B[j * r + bcol];
            }
        }
    };

    Cursor cur, pre;
    seg_at(cur, s0);
    seg_at(pre, s0);
    issue(std::integral_constant<int, 0>{}, pre);
    advance(pre);
    __builtin_amdgcn_sched_barrier(0);
    if (D > 1) {
        issue(std::integral_constant<int, (D > 1 ? 1 : 0)>{}, pre);
        advance(pre);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (D > 2) {
        issue(std::integral_constant<int, (D > 2 ? 2 : 0)>{}, pre);
        advance(pre);
        __builtin_amdgcn_sched_barrier(0);
    }

    double p = 0.0;
    float pf = 0.f;
    f64x4 accG = {0.0, 0.0, 0.0, 0.0};
    f32x4 accGf = zero4();
    auto body = [&](auto dc, bool live) {
        constexpr int d = decltype(dc)::value;
        float bcur[4];
        if (GRAM) {
#pragma unroll
            for (int v = 0; v < 4; ++v) bcur[v] = bnx[d][v];
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) *reinterpret_cast<f32x4 *>(L + t * 256 + ((lane ^ t) << 2)) = xr[d][t];
        issue(dc, pre);  // the slot is free again: its next block (two ahead) goes out now
        advance(pre);
        // EIGHT fp32 accumulation chains per output (32 columns = 8 MFMAs each; round 2: four of 16), summed as a tree in
        // fp64 and rounded once: the 256-term dot products of X C are the second largest rounding of config 4's B-phase
        // (tools/pf2_rounding_study.py), and that configuration's penalty-free A / C systems amplify it
        f32x4 acc8[4][2];
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            acc8[kc][0] = zero4(), acc8[kc][1] = zero4();
            f32x4 fr[4];
#pragma unroll
            for (int kq = 0; kq < 4; ++kq)
                fr[kq] = *reinterpret_cast<const f32x4 *>(L + i16 * 256 + (((16 * kc + 4 * kq + q) ^ i16) << 2));
#pragma unroll
            for (int kq = 0; kq < 4; ++kq)
#pragma unroll
                for (int m = 0; m < 4; ++m) acc8[kc][kq >> 1] = MFMA16(fr[kq][m], creg[kc][kq][m], acc8[kc][kq >> 1]);
        }
        f32x4 acc;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const double s01 = ((double)acc8[0][0][v] + (double)acc8[0][1][v]) + ((double)acc8[1][0][v] + (double)acc8[1][1][v]);
            const double s23 = ((double)acc8[2][0][v] + (double)acc8[2][1][v]) + ((double)acc8[3][0][v] + (double)acc8[3][1][v]);
            acc[v] = (float)(s01 + s23);
        }
        if (!live) return;  // the dummy half of an odd trip: nothing stored (wave-uniform)
        float bv[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int rl = 16 * cur.blk + 4 * q + v;
            const bool ok = (rl < cur.nrows) && (i16 < r);
            if (ok) XC[(cur.row0 + rl) * r + i16] = acc[v];
            if (GRAM) {
                const float b = ok ? bcur[v] : 0.f;
                bv[v] = b;
                if (GRAM == 2) p = fma((double)b, (double)acc[v], p);
                else pf = fmaf(b, acc[v], pf);
            }
        }
        if (GRAM) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                if (GRAM == 2) accG = __builtin_amdgcn_mfma_f64_16x16x4f64((double)bv[v], (double)bv[v], accG, 0, 0, 0);
                else accGf = MFMA16(bv[v], bv[v], accGf);
            }
            if (cur.blk == cur.nblk - 1) {  // the segment ends with this block: its reductions go out (wave-uniform)
                double t = (GRAM == 2) ? p : (double)pf;
                t += __shfl_xor(t, 16);
                t += __shfl_xor(t, 32);
                if (q == 0 && i16 < r) seg_rhs[(long)cur.sg * r + i16] = t;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int ra = (GRAM == 2) ? q + 4 * v : 4 * q + v;  // D layouts of the f64 / f32 MFMA
                    const double val = (GRAM == 2) ? accG[v] : (double)accGf[v];
                    if (ra < r && i16 < r) seg_btb[((long)cur.sg * r + ra) * r + i16] = val;
                }
                p = 0.0, pf = 0.f;
                accG = f64x4{0.0, 0.0, 0.0, 0.0}, accGf = zero4();
            }
        }
        advance(cur);
    };
    for (int b = 0; b < total; b += D) {
        body(std::integral_constant<int, 0>{}, true);
        if (D > 1) body(std::integral_constant<int, (D > 1 ? 1 : 0)>{}, b + 1 < total);
        if (D > 2) body(std::integral_constant<int, (D > 2 ? 2 : 0)>{}, b + 2 < total);
    }
}

// ---------------------------------------------------------------------------------------------------------
// k_slab_gram : one workgroup per slab.  rhs_i[c] = sum_j B[j][c] XC[j][c];  BtB_i = B_i^T B_i (fp64 MFMA: exact products).
// Writes the fp64 per-slab tables k_A_finish reads (one "segment" per slab) and their fp32 images (rhses by-product,
// k_A_e1).
// ---------------------------------------------------------------------------------------------------------
template <int NB>
__global__ __launch_bounds__(256) void k_slab_gram(const float *__restrict__ B, const float *__restrict__ XC,
                                                   const int *__restrict__ row_ptr, int r, float *__restrict__ rhsA,
                                                   float *__restrict__ BtB, double *__restrict__ rhs64,
                                                   double *__restrict__ btb64, const double *__restrict__ XC64) {
    constexpr int W = 16 * NB;
    __shared__ double ldsG[W * W];
    __shared__ double ldsP[4][W];
    const int slab = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rsub = lane >> 4, c16 = lane & 15;
    const int s = row_ptr[slab], e = row_ptr[slab + 1];
    const int n_groups = (e - s + 3) >> 2;
    double p[NB];
    f64x4 accG[NB][NB];
#pragma unroll
    for (int a = 0; a < NB; ++a) {
        p[a] = 0.0;
#pragma unroll
        for (int b = 0; b < NB; ++b) accG[a][b] = f64x4{0.0, 0.0, 0.0, 0.0};
    }
    for (int g = wave; g < n_groups; g += 4) {
        const long j = (long)s + 4 * g + rsub;
        double bv[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int col = 16 * nb + c16;
            float b = 0.f;
            double x = 0.0;
            if (j < e && col < r) {
                b = B[j * r + col];
                x = XC64 != nullptr ? XC64[j * r + col] : (double)XC[j * r + col];  // exact-products mode: the unrounded X C
            }
            bv[nb] = (double)b;
            p[nb] = fma((double)b, x, p[nb]);
        }
#pragma unroll
        for (int a = 0; a < NB; ++a)
#pragma unroll
            for (int b = 0; b < NB; ++b) accG[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(bv[a], bv[b], accG[a][b], 0, 0, 0);
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        p[nb] += __shfl_xor(p[nb], 16);
        p[nb] += __shfl_xor(p[nb], 32);
        if (rsub == 0) ldsP[wave][16 * nb + c16] = p[nb];
    }
    for (int wv = 0; wv < 4; ++wv) {
        if (wave == wv) {
#pragma unroll
            for (int a = 0; a < NB; ++a)
#pragma unroll
                for (int b = 0; b < NB; ++b)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int idx = (16 * a + rsub + 4 * v) * W + 16 * b + c16;  // f64 MFMA D layout
                        ldsG[idx] = (wv == 0) ? accG[a][b][v] : ldsG[idx] + accG[a][b][v];
                    }
        }
        __syncthreads();
    }
    for (int t = threadIdx.x; t < W * W; t += 256) {
        const int a = t / W, b = t - a * W;
        if (a < r && b < r) {
            BtB[((long)slab * r + a) * r + b] = (float)ldsG[t];
            btb64[((long)slab * r + a) * r + b] = ldsG[t];
        }
    }
    if (threadIdx.x < W && (int)threadIdx.x < r) {
        const double t = (ldsP[0][threadIdx.x] + ldsP[1][threadIdx.x]) + (ldsP[2][threadIdx.x] + ldsP[3][threadIdx.x]);
        rhsA[(long)slab * r + threadIdx.x] = (float)t;
        rhs64[(long)slab * r + threadIdx.x] = t;
    }
}

// =========================================================================================================
// host launchers
// =========================================================================================================
static inline int xt_KB(const mcl_context *c) {
    int kb = 4 / c->NB;
    if (kb < 1) kb = 1;
    const int need = (int)((c->K + 63) / 64);
    return need < kb ? (need <= 1 ? 1 : (need <= 2 ? 2 : 4)) : kb;
}

// both X passes run one wave per entry of the balanced wave -> segment table of mcl_set_problem (<= 1024 waves = 256 CUs x 4:
// measured best, one 256-thread block per CU)
int mcl_contract_n_partials(const mcl_context *c) { return std::max(1, (c->n_seg_waves + 3) / 4); }

template <int KB, int NB>
static int launch_xt(mcl_context *c) {
    const int nb = mcl_contract_n_partials(c);
    const int E = (int)(c->K * c->r + c->r * c->r);
    if (c->segs.n_tiles == 0) {  // no rows: the partial slab is all zeros
        MCL_CHECK_HIP(c, hipMemsetAsync(c->partials, 0, sizeof(double) * (size_t)E, c->stream));
        c->n_part = 1;
        return 0;
    }
    const int n_slices = (int)((c->K + 64 * KB - 1) / (64 * KB));
    dim3 grid((unsigned)(((nb + 7) / 8) * 8 * n_slices));  // 1-D: (row range mod 8 = XCD, slice, row range div 8), see the kernel
    const bool vec = (c->K % 4 == 0) && ((reinterpret_cast<uintptr_t>(c->X) & 15) == 0);
    ProfScope prof(c, MCL_PROF_XT);
    int dbg = 0, depth = 4;
    dbg = c->sw.xt_dbg;
    if (c->sw.xt_depth > 0) depth = c->sw.xt_depth;
    constexpr int RMODE = (NB == 4) ? 1 : 0;
#define MCL_XT_(VEC_, DEPTH_, MODE_, GRID_, NT_)                                                                      \
    hipLaunchKernelGGL((k_contract_xt<KB, NB, VEC_, DEPTH_, MODE_, NT_>), GRID_, dim3(256), 0, c->stream, c->X, c->B, c->A, \
                       c->segs.slab, c->segs.row0, c->segs.nrows, c->wave_seg_ptr, c->n_seg_waves, (int)c->K, c->r, c->partials, \
                       E, dbg, ((GRID_).x / (((nb + 7) / 8) * 8)), nb)
#define MCL_XT(VEC_, DEPTH_, MODE_, GRID_)                                                                            \
    do {                                                                                                              \
        if (c->x_streams && (VEC_) == 4 && (MODE_) != 2) MCL_XT_(VEC_, DEPTH_, MODE_, GRID_, true);                   \
        else MCL_XT_(VEC_, DEPTH_, MODE_, GRID_, false);                                                              \
    } while (0)
    if (vec) {
        if (depth == 2) MCL_XT(4, 2, RMODE, grid);
        else MCL_XT(4, 4, RMODE, grid);  // (round 6: eight groups in flight measured 13.57 against 13.91 ms at config 5 - not taken)
    } else {
        MCL_XT(1, 2, RMODE, grid);
    }
    if constexpr (NB == 4) MCL_XT(1, 2, 2, dim3((unsigned)(((nb + 7) / 8) * 8)));  // G of the same row ranges into the same partial slabs
#undef MCL_XT
#undef MCL_XT_
    c->n_part = nb;
    char buf[96];
    snprintf(buf, sizeof buf, "k_contract_xt<KB=%d,NB=%d,VEC=%d>", KB, NB, vec ? 4 : 1);
    c->variant[MCL_PROF_XT] = buf;
    MCL_CHECK_HIP(c, hipGetLastError());
    return 0;
}

int mcl_launch_contract_xt(mcl_context *c) {
    const int KB = xt_KB(c);
    switch (c->NB) {
        case 1:
            if (KB == 4) return launch_xt<4, 1>(c);
            if (KB == 2) return launch_xt<2, 1>(c);
            return launch_xt<1, 1>(c);
        case 2:
            if (KB >= 2) return launch_xt<2, 2>(c);
            return launch_xt<1, 2>(c);
        default:
            return launch_xt<1, 4>(c);
    }
}

int mcl_launch_reduce_partials(mcl_context *c) {
    const int E = (int)(c->K * c->r + c->r * c->r);
    ProfScope prof(c, MCL_PROF_REDUCE);
    c->variant[MCL_PROF_REDUCE] = "k_reduce_partials";
    hipLaunchKernelGGL(k_reduce_partials, dim3((E + 63) / 64), dim3(256), 0, c->stream, c->partials, c->n_part, E,
                       c->GR);
    MCL_CHECK_HIP(c, hipGetLastError());
    return 0;
}

static inline int xc_KC(const mcl_context *c) { return mcl_xc_chunks(c, nullptr); }

int mcl_launch_build_cfrag(mcl_context *c) {
    ProfScope prof_(c, MCL_PROF_OTHER);
    const int KC = mcl_cfrag_chunks(c);  // the image is shared with the sweep
    const long total = (long)KC * 4 * c->NB * 256;
    hipLaunchKernelGGL(k_build_cfrag, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, c->C, (int)c->K,
                       c->r, KC, c->NB, c->Cfrag);
    MCL_CHECK_HIP(c, hipGetLastError());
    return 0;
}

template <int NB>
static int launch_xc(mcl_context *c) {
    const int KC = xc_KC(c);
    const long nblk = (c->N + 15) / 16;
    long target_waves = 1024;  // measured best: one 256-thread block per CU
    if (c->sw.xc_waves > 0) target_waves = c->sw.xc_waves;
    int dbg = 0;
    dbg = c->sw.xc_dbg;
    long bpw = (nblk + target_waves - 1) / target_waves;
    if (bpw < 1) bpw = 1;
    const long waves = (nblk + bpw - 1) / bpw;
    const unsigned grid = (unsigned)((waves + 3) / 4);
    if (grid == 0) return 0;
    const bool vec = (c->K % 4 == 0) && ((reinterpret_cast<uintptr_t>(c->X) & 15) == 0);
    ProfScope prof(c, MCL_PROF_XC);
    int kct = 0;
    mcl_xc_chunks(c, &kct);
    if (vec && (c->K % 256 == 0) && !c->sw.xc_norow) {
        const size_t sm = sizeof(float) * 4 * 16 * 256;
        const bool creg = (c->K == 256) && (NB == 1);
        const int n_segs = c->segs.n_tiles;
        const unsigned g = (unsigned)((c->n_seg_waves + 3) / 4);
        // 0: X C only; 1: + per-segment rhs / Gram in fp32 chains (penalised A); 2: in fp64 (penalty-free A)
        int gram = !c->xc_with_gram ? 0 : (c->regs[0].n == 0 ? 2 : 1);
        if (NB == 4 && gram == 2) gram = 0;  // rank > 32 has no registers for the fp64 Gram tiles: k_slab_gram follows
#define MCL_XCR_(CREG_, GRAM_, NT_)                                                                                  \
    hipLaunchKernelGGL((k_contract_xc_row<NB, CREG_, GRAM_, NT_>), dim3(g), dim3(256), sm, c->stream, c->X, c->Cfrag, \
                       c->XC, c->B, c->segs.row0, c->segs.nrows, c->wave_seg_ptr, c->n_seg_waves, (int)c->K, c->r, c->seg_rhs, \
                       c->seg_btb)
#define MCL_XCR(CREG_, GRAM_)                                                                                        \
    do {                                                                                                              \
        if (c->x_streams) MCL_XCR_(CREG_, GRAM_, true);                                                               \
        else MCL_XCR_(CREG_, GRAM_, false);                                                                           \
    } while (0)
        if (n_segs > 0 && !creg && mcl_try_contract_xc_lds(c, gram)) {
            // K % 512 == 0 with the fragment image of C resident in LDS and four X tiles in flight per wave (xclds.hip)
            c->xc_did_gram = gram != 0;
            char bufl[96];
            snprintf(bufl, sizeof bufl, "k_contract_xc_lds<NB=%d,GRAM=%d>", NB, gram);
            c->variant[MCL_PROF_XC] = bufl;
            MCL_CHECK_HIP(c, hipGetLastError());
            return 0;
        }
        if (n_segs > 0) {
            if constexpr (NB == 1) {  // resident C fragments: K = 256, rank <= 16 only
                if (creg && !c->sw.xc_depth1) {  // two blocks of X in flight per wave
#define MCL_XC256_(GRAM_, NT_)                                                                                        \
    hipLaunchKernelGGL((k_contract_xc_256<GRAM_, 2, NT_>), dim3(g), dim3(256), sm, c->stream, c->X, c->Cfrag, c->XC, c->B, \
                       c->segs.row0, c->segs.nrows, c->wave_seg_ptr, c->n_seg_waves, c->r, c->seg_rhs, c->seg_btb)
#define MCL_XC256(GRAM_)                                                                                              \
    do {                                                                                                              \
        if (c->x_streams) MCL_XC256_(GRAM_, true);                                                                    \
        else MCL_XC256_(GRAM_, false);                                                                                \
    } while (0)
                    if (gram == 2) MCL_XC256(2);
                    else if (gram == 1) MCL_XC256(1);
                    else MCL_XC256(0);
#undef MCL_XC256
#undef MCL_XC256_
                } else if (creg) {
                    if (gram == 2) MCL_XCR(true, 2);
                    else if (gram == 1) MCL_XCR(true, 1);
                    else MCL_XCR(true, 0);
                }
            }
            if (!creg) {
                if (gram == 2) {
                    if constexpr (NB < 4) MCL_XCR(false, 2);
                } else if (gram == 1) {
                    MCL_XCR(false, 1);
                } else {
                    MCL_XCR(false, 0);
                }
            }
        }
#undef MCL_XCR
#undef MCL_XCR_
        c->xc_did_gram = gram != 0;
        char buf[96];
        if (creg && !c->sw.xc_depth1) snprintf(buf, sizeof buf, "k_contract_xc_256<DEPTH=2,GRAM=%d>", gram);
        else snprintf(buf, sizeof buf, "k_contract_xc_row<NB=%d,CREG=%d,GRAM=%d>", NB, creg ? 1 : 0, gram);
        c->variant[MCL_PROF_XC] = buf;
        MCL_CHECK_HIP(c, hipGetLastError());
        return 0;
    }
    c->xc_did_gram = false;
#define MCL_XC(NB_, VEC_, KCT_)                                                                                     \
    hipLaunchKernelGGL((k_contract_xc<NB_, VEC_, KCT_>), dim3(grid), dim3(256), 0, c->stream, c->X, c->Cfrag, c->XC, \
                       (long)c->N, (int)c->K, c->r, KC, bpw, nblk, dbg)
    if (vec) {
        if (kct == 4) MCL_XC(NB, 4, 4);
        else if (kct == 2) MCL_XC(NB, 4, 2);
        else MCL_XC(NB, 4, 0);
    } else {
        if (kct == 4) MCL_XC(NB, 1, 4);
        else if (kct == 2) MCL_XC(NB, 1, 2);
        else MCL_XC(NB, 1, 0);
    }
#undef MCL_XC
    char buf[96];
    snprintf(buf, sizeof buf, "k_contract_xc<NB=%d,VEC=%d,KCT=%d>", NB, vec ? 4 : 1, kct);
    c->variant[MCL_PROF_XC] = buf;
    MCL_CHECK_HIP(c, hipGetLastError());
    return 0;
}

int mcl_launch_contract_xc(mcl_context *c) {
    if (c->NB == 1) return launch_xc<1>(c);
    if (c->NB == 2) return launch_xc<2>(c);
    return launch_xc<4>(c);
}

int mcl_launch_slab_gram(mcl_context *c) {
    ProfScope prof_(c, MCL_PROF_OTHER);
    if (c->I == 0) return 0;
    dim3 grid((unsigned)c->I);
    if (c->NB == 1)
        hipLaunchKernelGGL(k_slab_gram<1>, grid, dim3(256), 0, c->stream, c->B, c->XC, c->row_ptr_dev, c->r, c->rhsA,
                           c->BtB, c->seg_rhs, c->seg_btb, c->exact ? (const double *)c->XC64 : nullptr);
    else if (c->NB == 2)
        hipLaunchKernelGGL(k_slab_gram<2>, grid, dim3(256), 0, c->stream, c->B, c->XC, c->row_ptr_dev, c->r, c->rhsA,
                           c->BtB, c->seg_rhs, c->seg_btb, c->exact ? (const double *)c->XC64 : nullptr);
    else
        hipLaunchKernelGGL(k_slab_gram<4>, grid, dim3(256), 0, c->stream, c->B, c->XC, c->row_ptr_dev, c->r, c->rhsA,
                           c->BtB, c->seg_rhs, c->seg_btb, c->exact ? (const double *)c->XC64 : nullptr);
    MCL_CHECK_HIP(c, hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// Exact-products mode.  The MFMA contractions above multiply exactly and accumulate in fp32 chains (summed in fp64 per
// segment): [G | R], X C and the A-phase tables carry relative errors of 1e-8 .. 4e-8, and a penalty-free mode - solved with
// the un-shifted normal equations, as the reference does with its fp64 SVD (decomposition.py:252-256, 319-321) - multiplies
// them by the condition number of its system (tools/parity_probe.py fuzz:142: cond 2e4 -> C off by 2.3e-5; fuzz:260: cond
// 1.5e5 -> A off by 7e-4).  On LARGE problems the roundings of 1e5..1e7 rows average out (config 4: [G | R] to 6e-9, C to
// 8e-8 at cond 600) and the passes over X are what the iteration costs; on SMALL problems neither holds: every kernel is
// launch-bound and a few hundred rows do not average anything.  So problems of at most 2^20 elements of X (4 MB: an eighth of
// config 2, the smallest BASELINE configuration) take every contraction as fp64 sums of EXACT products of the stored fp32
// values - what the reference's arithmetic gives on the same inputs up to 1e-16 - and keep fp64 through the solves.  Cost
// (tools/exact_mode_cost.py; no sweep, two more launches per iteration): 43 against 39 us per iteration at the size of
// BASELINE config 1, 61 against 39 us at the 2^20 limit (rank 16), 154 against 112 us there at rank 32.
//   X C            k_contract_xc_f64 (fp64 MFMA; admm.hip) -> XC64 and its once-rounded fp32 image for the row kernels
//   [G | R]        k_exact_gr below (fp64 MFMA over 256-row chunks, summed in a fixed order) -> GR, no sweep
//   rhs_i, B_i^T B_i   k_slab_gram from XC64 -> the fp64 tables k_A_finish reads (and the reconstruction error)
// MCL_EXACT=1 / 0 forces the mode on / off (the parity tests of the fast kernels run small problems with MCL_EXACT=0).
// ---------------------------------------------------------------------------------------------------------
bool mcl_exact_mode(const mcl_context *c) {
    // decided ONCE per installed workspace (mcl_set_workspace): the carve-up depends on it, so neither a size query nor
    // a later mcl_reload_switches may flip it under an installed workspace
    if (c->has_workspace) return c->exact;
    if (c->sw.exact >= 0) return c->sw.exact != 0;
    // a host that shards one problem over several contexts says which arithmetic ALL of them use (mcl_options.exact_products:
    // the decision belongs to the WHOLE problem, not to a rank's share of it)
    if (c->opt.exact_products == 1) return true;
    if (c->opt.exact_products == 2) return false;
    return c->N * c->K <= (int64_t(1) << 20);
}

// [G | R] of the exact-products mode on the fp64 MFMA.  Block (x, y): row chunk y (256 rows: the unit of the fixed-order sum),
// x < ceil(K / 16): the 16 x r tile R[16 x .. + 15][:], x == ceil(K / 16): G.  One wave; per group of 4 rows lane
// (i = l & 15, kk = l >> 4) feeds A[i][kk] = X[row + kk][16 x + i] (resp. (b a)[row + kk][16 na + i]) and
// B[kk][j = i] = (b a)[row + kk][16 nb + i]; (b a) is an exact product of two fp32 values, the MFMA rounds each fp64 multiply-add
// once.  D: lane l, register v = D[(l >> 4) + 4 v][l & 15].  Every output is a sum over the rows in ascending order
// (k_exact_gr_reduce adds the chunks in ascending order): deterministic, and the same on every rank layout of the same rows.
template <int NB>
__global__ __launch_bounds__(64) void k_exact_gr(const float *__restrict__ X, const float *__restrict__ B,
                                                 const float *__restrict__ A, const int *__restrict__ slab_of_row, long N, int K,
                                                 int r, double *__restrict__ part) {
    const int lane = threadIdx.x, i = lane & 15, kk = lane >> 4;
    const int kblocks = (K + 15) / 16;
    const long c0 = (long)blockIdx.y * 256, c1 = min(c0 + 256, N);
    const long E = (long)r * r + (long)K * r;
    double *out = part + (long)blockIdx.y * E;
    // unconditional loads at clamped indices, masked by a MULTIPLICATION (a select lets the compiler sink the loads into a
    // branch again, and a branch per load serialises the memory round trips of a trip); rows / columns past the end
    // contribute an exact zero (the clamped element is finite data of the same arrays)
    auto ba = [&](long row, int col) -> double {
        const long rc = min(row, c1 - 1);
        const int cc = min(col, r - 1);
        const double mask = (row < c1 && col < r) ? 1.0 : 0.0;
        return ((double)B[rc * r + cc] * mask) * (double)A[(long)slab_of_row[rc] * r + cc];
    };
    if ((int)blockIdx.x < kblocks) {
        const int k = 16 * blockIdx.x + i;
        f64x4 acc[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = f64x4{0.0, 0.0, 0.0, 0.0};
        for (long g = c0; g < c1; g += 16) {  // four groups of 4 rows per trip: their loads are independent and issued together
            double x[4], w[4][NB];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long row = g + 4 * u + kk;
                x[u] = (double)X[min(row, c1 - 1) * K + min(k, K - 1)] * ((row < c1 && k < K) ? 1.0 : 0.0);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) w[u][nb] = ba(row, 16 * nb + i);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[u], w[u][nb], acc[nb], 0, 0, 0);
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int ko = 16 * blockIdx.x + kk + 4 * v, col = 16 * nb + i;
                if (ko < K && col < r) out[(long)r * r + (long)ko * r + col] = acc[nb][v];
            }
    } else {
        f64x4 acc[NB][NB];
#pragma unroll
        for (int na = 0; na < NB; ++na)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[na][nb] = f64x4{0.0, 0.0, 0.0, 0.0};
        for (long g = c0; g < c1; g += 16) {
            double v[4][NB];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) v[u][nb] = ba(g + 4 * u + kk, 16 * nb + i);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int na = 0; na < NB; ++na)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        acc[na][nb] = __builtin_amdgcn_mfma_f64_16x16x4f64(v[u][na], v[u][nb], acc[na][nb], 0, 0, 0);
        }
#pragma unroll
        for (int na = 0; na < NB; ++na)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int a = 16 * na + kk + 4 * v, b = 16 * nb + i;
                    if (a < r && b < r) out[(long)a * r + b] = acc[na][nb][v];
                }
    }
}

__global__ __launch_bounds__(256) void k_exact_gr_reduce(const double *__restrict__ part, int n_chunks, long E,
                                                         double *__restrict__ GR) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= E) return;
    double s = 0.0;
    // ascending: fixed order.  Eight chunks' loads in flight per trip (round 6: one dependent load per chunk made this kernel 29 us
    // on a 32 K-row problem - the exact arithmetic serves up to 2^24 elements since the condition monitor); the sum's order is the same
    for (int c = 0; c < n_chunks; c += 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = part[(long)min(c + u, n_chunks - 1) * E + e];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (c + u < n_chunks) s += v[u];
    }
    GR[e] = s;
}

int mcl_launch_exact_gr(mcl_context *c) {
    const int kblocks = (int)((c->K + 15) / 16);
    const int n_chunks = (int)std::max<int64_t>(1, (c->N + 255) / 256);
    const long E = (long)c->K * c->r + (long)c->r * c->r;
    const dim3 grid((unsigned)(kblocks + 1), (unsigned)n_chunks);
    double *part = n_chunks == 1 ? c->GR : c->exact_part;  // one chunk (<= 256 rows): its sums ARE [G | R], no second launch
    ProfScope prof(c, MCL_PROF_XT);  // the exact-products form of the X^T pass (and its reduction)
    c->variant[MCL_PROF_XT] = "k_exact_gr (+ k_exact_gr_reduce)";
    if (c->NB == 1)
        hipLaunchKernelGGL(k_exact_gr<1>, grid, dim3(64), 0, c->stream, c->X, c->B, c->A, c->slab_of_row, (long)c->N, (int)c->K, c->r, part);
    else if (c->NB == 2)
        hipLaunchKernelGGL(k_exact_gr<2>, grid, dim3(64), 0, c->stream, c->X, c->B, c->A, c->slab_of_row, (long)c->N, (int)c->K, c->r, part);
    else
        hipLaunchKernelGGL(k_exact_gr<4>, grid, dim3(64), 0, c->stream, c->X, c->B, c->A, c->slab_of_row, (long)c->N, (int)c->K, c->r, part);
    if (n_chunks > 1)
        hipLaunchKernelGGL(k_exact_gr_reduce, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, c->stream, c->exact_part, n_chunks, E, c->GR);
    MCL_CHECK_HIP(c, hipGetLastError());
    return 0;
}
