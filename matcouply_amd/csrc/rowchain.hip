// The chained B-mode row pass of a fused penalty stack (k_rows_finish_solve_stats, generic.hip) with its loads SOFTWARE-PIPELINED.
//
// The chained pass of rounds 2-5 reads the rows of a 16-row block (factor, right-hand side, the dual of every penalty, the
// auxiliary rows the column regressions wrote: 12 x 16 B per lane at rank 32 with three penalties), waits for them, runs the
// block's ~50 matrix-core instructions and stores - one block after the other, two waves per SIMD: while a wave computes
// nothing of its own is in flight, and config 5's pass moved 19.3 GB in 4.68 ms (0.52 of the HBM peak, 4.1 TB/s) where a
// streaming read reaches 6.3.  Here the loads of block rb + 1 are issued BEFORE block rb is computed.  For the hardware's
// in-order return counter to let the block's wait cover exactly its own loads (s_waitcnt vmcnt(N) with N = everything younger),
// the number of memory operations between two waits has to be a compile-time constant:
//   * the composition of the stack (how many penalties, which of them PARAFAC2 / unimodality / L2 ball) is a TEMPLATE
//     argument - the stacks of the BASELINE configurations are instantiated, every other stack keeps the kernel of generic.hip;
//   * loads are unconditional at clamped addresses and masked at use, stores are unconditional with the lanes outside the
//     matrix writing to a per-lane sink;
//   * a block is entered only from its predecessor (unrolled loop with an early exit).
// Arithmetic, its order and every rounding are those of k_rows_finish_solve_stats: results are bit-identical
// (tests/test_gpu_end_to_end.py::test_full_size_config4_properties compares the chained with the un-chained form, MCL_NO_ROW_PREFETCH=1
// selects the old kernel for A/B runs).  Reference: the inner loop of admm_update_B, decomposition.py:259-285.
#include <type_traits>

#include "mcl_internal.h"
#include "rows_mfma.h"

namespace {

enum { CLS_ROWSEP = 0, CLS_PF2 = 1, CLS_UNI = 2, CLS_L2 = 3 };
// SIG = n | cls_0 << 3 | cls_1 << 5 | cls_2 << 7 | cls_3 << 9
constexpr int sig_n(int sig) { return sig & 7; }
constexpr int sig_cls(int sig, int k) { return (sig >> (3 + 2 * k)) & 3; }
constexpr int sig_last(int sig, int cls) {
    int found = -1;
    for (int k = 0; k < sig_n(sig); ++k)
        if (sig_cls(sig, k) == cls) found = k;
    return found;
}
constexpr int make_sig(int n, int c0, int c1 = 0, int c2 = 0, int c3 = 0) { return n | c0 << 3 | c1 << 5 | c2 << 7 | c3 << 9; }

static __device__ __forceinline__ float prox_rowsep(int kind, int nonneg, float p0, float p1, float thr, float y) {
    switch (kind) {  // (generic.hip: prox_elem_g, float form)
        case MCL_PEN_NN: return fmaxf(y, 0.f);
        case MCL_PEN_BOX: return fminf(fmaxf(y, p0), p1);
        case MCL_PEN_L1:
            if (nonneg) return fmaxf(y - thr, 0.f);
            return copysignf(fmaxf(fabsf(y) - thr, 0.f), y);
        default: return y;
    }
}

template <int NBR, bool R64, int SIG>
__global__ __launch_bounds__(256) void k_rows_chain_mid(ModeView mv, const float *__restrict__ rhs_src, const float *__restrict__ Arows,
                                                        const float *__restrict__ Linv, RegSet regs, int r,
                                                        const float *__restrict__ T, const double *__restrict__ colsq,
                                                        double *__restrict__ stat_gram, double *__restrict__ stat_colsq,
                                                        const double *__restrict__ Linv64, const double *__restrict__ T64,
                                                        float *__restrict__ sink_base) {
    typedef double f64x4s __attribute__((ext_vector_type(4)));
    typedef RowArith<R64> RA;
    constexpr int N = sig_n(SIG);
    constexpr int kpf2 = sig_last(SIG, CLS_PF2), kl2 = sig_last(SIG, CLS_L2);
    __shared__ double ytile[R64 ? 4 * 16 * 17 : 1];
    MCL_GATE(mv.gate);
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= mv.n_tiles) return;
    const int slab = __builtin_amdgcn_readfirstlane(mv.tile_slab[tile]);
    const long row0 = __builtin_amdgcn_readfirstlane(mv.tile_row0[tile]);
    const int nrows = __builtin_amdgcn_readfirstlane(mv.tile_nrows[tile]);
    const int row16 = lane & 15, g = lane >> 4;
    const float rho = mv.rho[slab];
    typename RA::template Mat<NBR> L, Ts, D;
    if constexpr (R64) L.load(Linv64 + (long)slab * r * r, r, lane);
    else L.load(Linv + (long)slab * r * r, r, lane);
    if constexpr (kpf2 >= 0) {
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
    float l2s[NBR][4];
#pragma unroll
    for (int h = 0; h < NBR; ++h)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            l2s[h][v] = 1.f;
            if constexpr (kl2 >= 0) {
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

    // column of the lane's 16-B access per column block, clamped into the matrix for the loads (r % 4 == 0, r >= 4)
    int colc[NBR];
    bool colok[NBR];
#pragma unroll
    for (int h = 0; h < NBR; ++h) colok[h] = 16 * h + 4 * g < r, colc[h] = min(16 * h + 4 * g, r - 4);
    float *sink = sink_base + (((tile & 63) * 64 + lane) << 2);

    struct Blk {
        f32x4 f[NBR], t[NBR], u[N > 0 ? N : 1][NBR], zu[N > 0 ? N : 1][NBR];
    };
    auto load_blk = [&](int rb, Blk &b) {
        const long j = row0 + min(16 * rb + row16, nrows - 1);
#pragma unroll
        for (int h = 0; h < NBR; ++h) {
            b.f[h] = *reinterpret_cast<const f32x4 *>(mv.F + j * r + colc[h]);
            b.t[h] = *reinterpret_cast<const f32x4 *>(rhs_src + j * r + colc[h]);
        }
#pragma unroll
        for (int k = 0; k < N; ++k)
#pragma unroll
            for (int h = 0; h < NBR; ++h) {
                b.u[k][h] = *reinterpret_cast<const f32x4 *>(regs.dual[k] + j * r + colc[h]);
                if (sig_cls(SIG, k) == CLS_UNI) b.zu[k][h] = *reinterpret_cast<const f32x4 *>(regs.aux[k] + j * r + colc[h]);
            }
    };
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

    Blk cur;
    load_blk(0, cur);
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
        if (16 * rb >= nrows) break;  // (wave-uniform; a block is only entered from its predecessor)
        Blk nxt;
        load_blk((16 * (rb + 1) < nrows) ? rb + 1 : rb, nxt);  // the last block re-requests its own rows (cache hits): a fixed count
        __builtin_amdgcn_sched_barrier(0);
        const bool ok = 16 * rb + row16 < nrows;
        const long j = row0 + 16 * rb + (ok ? row16 : 0);
        f32x4 f[NBR], t[NBR], upf[NBR], ul2[NBR];
#pragma unroll
        for (int h = 0; h < NBR; ++h) {
            const bool m = ok && colok[h];
            f[h] = m ? cur.f[h] : zero;  // zeros for padding rows / columns, as row_ld4 returns them
            t[h] = m ? cur.t[h] : zero;
#pragma unroll
            for (int v = 0; v < 4; ++v) t[h][v] *= av[h][v];
        }
        // ---- iteration t: prox + dual of every penalty
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const int kind = regs.kind[k];
            f32x4 u[NBR], zg[NBR];
#pragma unroll
            for (int h = 0; h < NBR; ++h) u[h] = (ok && colok[h]) ? cur.u[k][h] : zero;
            if (sig_cls(SIG, k) == CLS_PF2) {
                typename RA::Y y[NBR], pz[NBR];
#pragma unroll
                for (int h = 0; h < NBR; ++h) y[h] = RA::ysum(f[h], u[h]);
                Ts.apply(y, pz);   // P = Y T_i
                D.apply(pz, zg);   // P Delta
            } else if (sig_cls(SIG, k) == CLS_UNI) {  // aux rows written by the column regressions
#pragma unroll
                for (int h = 0; h < NBR; ++h) zg[h] = (ok && colok[h]) ? cur.zu[k][h] : zero;
            } else if (sig_cls(SIG, k) == CLS_L2) {
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
                        zg[h][v] = prox_rowsep(kind, regs.nonneg[k], regs.p0[k], regs.p1[k], thr, f[h][v] + u[h][v]);
            }
#pragma unroll
            for (int h = 0; h < NBR; ++h) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    u[h][v] = f[h][v] - (zg[h][v] - u[h][v]);
                    t[h][v] = fmaf(rho, zg[h][v] - u[h][v], t[h][v]);
                }
                float *dst = (ok && colok[h]) ? regs.dual[k] + j * r + 16 * h + 4 * g : sink;
                *reinterpret_cast<f32x4 *>(dst) = u[h];
                if (k == kpf2) upf[h] = u[h];
                if (k == kl2) ul2[h] = u[h];
            }
        }
        // ---- iteration t + 1: solve, store, statistics of the new rows
        f32x4 fn[NBR];
        L.apply(t, fn);
#pragma unroll
        for (int h = 0; h < NBR; ++h) {
            float *dst = (ok && colok[h]) ? mv.F + j * r + 16 * h + 4 * g : sink;
            *reinterpret_cast<f32x4 *>(dst) = fn[h];
        }
        if constexpr (kl2 >= 0) {
#pragma unroll
            for (int h = 0; h < NBR; ++h)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    float y = fn[h][v] + ul2[h][v];
                    if (regs.nonneg[kl2]) y = fmaxf(y, 0.f);
                    if (ok) csq[h][v] += (double)y * (double)y;
                }
        }
        if constexpr (kpf2 >= 0) {
            double yt[NBR][4];
#pragma unroll
            for (int nb = 0; nb < NBR; ++nb) {
                if constexpr (R64) {  // exact fp64 sum, transposed through the wave's LDS tile
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
                        tr = MFMA16(y, bsel[v], tr);
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
        cur = nxt;
    }
    constexpr int W = 16 * NBR;
    if constexpr (kpf2 >= 0) {
        double *out = stat_gram + (long)tile * W * W;
#pragma unroll
        for (int a = 0; a < NBR; ++a)
#pragma unroll
            for (int b = 0; b < NBR; ++b)
#pragma unroll
                for (int v = 0; v < 4; ++v) out[(16 * a + g + 4 * v) * W + 16 * b + row16] = accS[a][b][v];
    }
    if constexpr (kl2 >= 0) {
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

// ---------------------------------------------------------------------------------------------------------
// The FIRST pass of the chain (k_rows_solve_stats, generic.hip): solve of inner iteration 0 from the aux / dual rows the phase
// starts with + the statistics of the new rows.  Same pipeline as above; every L2 ball of the stack has its column sums here.
// ---------------------------------------------------------------------------------------------------------
template <int NBR, bool R64, int SIG>
__global__ __launch_bounds__(256) void k_rows_chain_first(ModeView mv, const float *__restrict__ rhs_src, const float *__restrict__ Arows,
                                                          const float *__restrict__ Linv, RegSet regs, int r,
                                                          double *__restrict__ stat_gram, double *__restrict__ stat_colsq,
                                                          const double *__restrict__ Linv64, float *__restrict__ sink_base) {
    typedef double f64x4s __attribute__((ext_vector_type(4)));
    typedef RowArith<R64> RA;
    constexpr int N = sig_n(SIG);
    constexpr int kpf2 = sig_last(SIG, CLS_PF2);
    __shared__ double ytile[R64 ? 4 * 16 * 17 : 1];
    MCL_GATE(mv.gate);
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= mv.n_tiles) return;
    const int slab = __builtin_amdgcn_readfirstlane(mv.tile_slab[tile]);
    const long row0 = __builtin_amdgcn_readfirstlane(mv.tile_row0[tile]);
    const int nrows = __builtin_amdgcn_readfirstlane(mv.tile_nrows[tile]);
    const int row16 = lane & 15, g = lane >> 4;
    const float rho = mv.rho[slab];
    typename RA::template Mat<NBR> L, D;
    if constexpr (R64) L.load(Linv64 + (long)slab * r * r, r, lane);
    else L.load(Linv + (long)slab * r * r, r, lane);
    if constexpr (kpf2 >= 0) D.load(regs.aux2[kpf2], r, lane);
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
    double csq[N > 0 ? N : 1][NBR][4];
#pragma unroll
    for (int k = 0; k < N; ++k)
#pragma unroll
        for (int h = 0; h < NBR; ++h)
#pragma unroll
            for (int v = 0; v < 4; ++v) csq[k][h][v] = 0.0;
    int colc[NBR];
    bool colok[NBR];
#pragma unroll
    for (int h = 0; h < NBR; ++h) colok[h] = 16 * h + 4 * g < r, colc[h] = min(16 * h + 4 * g, r - 4);
    float *sink = sink_base + (((tile & 63) * 64 + lane) << 2);

    struct Blk {
        f32x4 t[NBR], z[N > 0 ? N : 1][NBR], u[N > 0 ? N : 1][NBR];
    };
    auto load_blk = [&](int rb, Blk &b) {
        const long j = row0 + min(16 * rb + row16, nrows - 1);
#pragma unroll
        for (int h = 0; h < NBR; ++h) b.t[h] = *reinterpret_cast<const f32x4 *>(rhs_src + j * r + colc[h]);
#pragma unroll
        for (int k = 0; k < N; ++k)
#pragma unroll
            for (int h = 0; h < NBR; ++h) {
                b.z[k][h] = *reinterpret_cast<const f32x4 *>(regs.aux[k] + j * r + colc[h]);
                b.u[k][h] = *reinterpret_cast<const f32x4 *>(regs.dual[k] + j * r + colc[h]);
            }
    };
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    Blk cur;
    load_blk(0, cur);
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
        if (16 * rb >= nrows) break;
        Blk nxt;
        load_blk((16 * (rb + 1) < nrows) ? rb + 1 : rb, nxt);
        __builtin_amdgcn_sched_barrier(0);
        const bool ok = 16 * rb + row16 < nrows;
        const long j = row0 + 16 * rb + (ok ? row16 : 0);
        f32x4 t[NBR], f[NBR], ukeep[N > 0 ? N : 1][NBR];
#pragma unroll
        for (int h = 0; h < NBR; ++h) {
            t[h] = (ok && colok[h]) ? cur.t[h] : zero;
#pragma unroll
            for (int v = 0; v < 4; ++v) t[h][v] *= av[h][v];
        }
#pragma unroll
        for (int k = 0; k < N; ++k) {
            f32x4 z[NBR];
#pragma unroll
            for (int h = 0; h < NBR; ++h) z[h] = (ok && colok[h]) ? cur.z[k][h] : zero;
            if (k == kpf2) {
                f32x4 pz[NBR];
                D.apply(z, pz);
#pragma unroll
                for (int h = 0; h < NBR; ++h) z[h] = pz[h];
            }
#pragma unroll
            for (int h = 0; h < NBR; ++h) {
                ukeep[k][h] = (ok && colok[h]) ? cur.u[k][h] : zero;
#pragma unroll
                for (int v = 0; v < 4; ++v) t[h][v] = fmaf(rho, z[h][v] - ukeep[k][h][v], t[h][v]);
            }
        }
        L.apply(t, f);
#pragma unroll
        for (int h = 0; h < NBR; ++h) {
            float *dst = (ok && colok[h]) ? mv.F + j * r + 16 * h + 4 * g : sink;
            *reinterpret_cast<f32x4 *>(dst) = f[h];
        }
#pragma unroll
        for (int k = 0; k < N; ++k) {
            if (sig_cls(SIG, k) == CLS_L2) {
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
                    if constexpr (R64) {
                        double *yl = ytile + (threadIdx.x >> 6) * (16 * 17);
#pragma unroll
                        for (int v = 0; v < 4; ++v) yl[row16 * 17 + 4 * g + v] = ok ? (double)f[nb][v] + (double)ukeep[k][nb][v] : 0.0;
#pragma unroll
                        for (int w = 0; w < 4; ++w) yt[nb][w] = yl[(g + 4 * w) * 17 + row16];
                    } else {
                        f32x4 tr = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            const float y = ok ? f[nb][v] + ukeep[k][nb][v] : 0.f;
                            tr = MFMA16(y, bsel[v], tr);
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
        cur = nxt;
    }
    constexpr int W = 16 * NBR;
    if constexpr (kpf2 >= 0) {
        double *out = stat_gram + (long)tile * W * W;
#pragma unroll
        for (int a = 0; a < NBR; ++a)
#pragma unroll
            for (int b = 0; b < NBR; ++b)
#pragma unroll
                for (int v = 0; v < 4; ++v) out[(16 * a + g + 4 * v) * W + 16 * b + row16] = accS[a][b][v];
    }
#pragma unroll
    for (int k = 0; k < N; ++k) {
        if (sig_cls(SIG, k) == CLS_L2) {
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

// ---------------------------------------------------------------------------------------------------------
// The LAST pass of the chain (k_rows_finish_fused, generic.hip): prox + dual of the last inner iteration, the auxiliary rows
// written out, the mode's per-tile diagnostics.
// ---------------------------------------------------------------------------------------------------------
template <int NBR, bool R64, int SIG>
__global__ __launch_bounds__(256) void k_rows_chain_last(ModeView mv, RegSet regs, int r, const float *__restrict__ T,
                                                         const double *__restrict__ colsq, double *__restrict__ diag_tile,
                                                         int want_diag, const double *__restrict__ T64, float *__restrict__ sink_base) {
    typedef RowArith<R64> RA;
    constexpr int N = sig_n(SIG);
    constexpr int kpf2 = sig_last(SIG, CLS_PF2);
    MCL_GATE(mv.gate);
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= mv.n_tiles) return;
    const int slab = __builtin_amdgcn_readfirstlane(mv.tile_slab[tile]);
    const long row0 = __builtin_amdgcn_readfirstlane(mv.tile_row0[tile]);
    const int nrows = __builtin_amdgcn_readfirstlane(mv.tile_nrows[tile]);
    const int row16 = lane & 15, g = lane >> 4;
    double nf = 0.0, na = 0.0, gap[MCL_MAX_REGS];
#pragma unroll
    for (int k = 0; k < MCL_MAX_REGS; ++k) gap[k] = 0.0;
    const float rho = mv.rho[slab];
    typename RA::template Mat<NBR> Ts, D;
    if constexpr (kpf2 >= 0) {
        if constexpr (R64) Ts.load(T64 + (long)slab * r * r, r, lane);
        else Ts.load(T + (long)slab * r * r, r, lane);
        D.load(regs.aux2[kpf2], r, lane);
    }
    // the L2-ball scale factors (k_rows_finish_fused recomputes them per element from the same column norms: the same values)
    float l2s[N > 0 ? N : 1][NBR][4];
#pragma unroll
    for (int k = 0; k < N; ++k)
#pragma unroll
        for (int h = 0; h < NBR; ++h)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                l2s[k][h][v] = 1.f;
                if (sig_cls(SIG, k) == CLS_L2) {
                    const int col = 16 * h + 4 * g + v;
                    const float bound = regs.p0[k];
                    const float nrm = (col < r) ? (float)sqrt(colsq[((long)k * mv.n_slabs + slab) * r + col]) : 1.f;
                    l2s[k][h][v] = bound / fmaxf(nrm, bound);
                }
            }
    int colc[NBR];
    bool colok[NBR];
#pragma unroll
    for (int h = 0; h < NBR; ++h) colok[h] = 16 * h + 4 * g < r, colc[h] = min(16 * h + 4 * g, r - 4);
    float *sink = sink_base + (((tile & 63) * 64 + lane) << 2);

    struct Blk {
        f32x4 f[NBR], u[N > 0 ? N : 1][NBR], zu[N > 0 ? N : 1][NBR];
    };
    auto load_blk = [&](int rb, Blk &b) {
        const long j = row0 + min(16 * rb + row16, nrows - 1);
#pragma unroll
        for (int h = 0; h < NBR; ++h) b.f[h] = *reinterpret_cast<const f32x4 *>(mv.F + j * r + colc[h]);
#pragma unroll
        for (int k = 0; k < N; ++k)
#pragma unroll
            for (int h = 0; h < NBR; ++h) {
                b.u[k][h] = *reinterpret_cast<const f32x4 *>(regs.dual[k] + j * r + colc[h]);
                if (sig_cls(SIG, k) == CLS_UNI) b.zu[k][h] = *reinterpret_cast<const f32x4 *>(regs.aux[k] + j * r + colc[h]);
            }
    };
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    Blk cur;
    load_blk(0, cur);
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
        if (16 * rb >= nrows) break;
        Blk nxt;
        load_blk((16 * (rb + 1) < nrows) ? rb + 1 : rb, nxt);
        __builtin_amdgcn_sched_barrier(0);
        const bool ok = 16 * rb + row16 < nrows;
        const long j = row0 + 16 * rb + (ok ? row16 : 0);
        f32x4 f[NBR];
#pragma unroll
        for (int h = 0; h < NBR; ++h) {
            f[h] = (ok && colok[h]) ? cur.f[h] : zero;
            if (want_diag) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    nf += (double)f[h][v] * (double)f[h][v];
                    na += fabs((double)f[h][v]);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const int kind = regs.kind[k];
            f32x4 u[NBR], z[NBR], zg[NBR];
#pragma unroll
            for (int h = 0; h < NBR; ++h) u[h] = (ok && colok[h]) ? cur.u[k][h] : zero;
            if (sig_cls(SIG, k) == CLS_PF2) {
                typename RA::Y y[NBR], pw[NBR];
                f32x4 pd[NBR];
#pragma unroll
                for (int h = 0; h < NBR; ++h) y[h] = RA::ysum(f[h], u[h]);
                Ts.apply(y, pw);
                D.apply(pw, pd);
#pragma unroll
                for (int h = 0; h < NBR; ++h) {
                    z[h] = RA::narrow(pw[h]);
                    zg[h] = pd[h];
#pragma unroll
                    for (int v = 0; v < 4; ++v) u[h][v] = f[h][v] - (pd[h][v] - u[h][v]);
                }
            } else if (sig_cls(SIG, k) == CLS_UNI) {
#pragma unroll
                for (int h = 0; h < NBR; ++h) {
                    z[h] = (ok && colok[h]) ? cur.zu[k][h] : zero;
#pragma unroll
                    for (int v = 0; v < 4; ++v) u[h][v] = f[h][v] - (z[h][v] - u[h][v]);
                }
            } else if (sig_cls(SIG, k) == CLS_L2) {
#pragma unroll
                for (int h = 0; h < NBR; ++h)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        float y = f[h][v] + u[h][v];
                        if (regs.nonneg[k]) y = fmaxf(y, 0.f);
                        z[h][v] = y * l2s[k][h][v];
                        u[h][v] = f[h][v] - (z[h][v] - u[h][v]);
                    }
            } else {
                const float thr = regs.p0[k] / rho;
#pragma unroll
                for (int h = 0; h < NBR; ++h)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        z[h][v] = prox_rowsep(kind, regs.nonneg[k], regs.p0[k], regs.p1[k], thr, f[h][v] + u[h][v]);
                        u[h][v] = f[h][v] - (z[h][v] - u[h][v]);
                    }
            }
#pragma unroll
            for (int h = 0; h < NBR; ++h) {
                const bool m = ok && colok[h];
                if (sig_cls(SIG, k) != CLS_UNI) *reinterpret_cast<f32x4 *>(m ? regs.aux[k] + j * r + 16 * h + 4 * g : sink) = z[h];
                *reinterpret_cast<f32x4 *>(m ? regs.dual[k] + j * r + 16 * h + 4 * g : sink) = u[h];
                if (sig_cls(SIG, k) != CLS_PF2) zg[h] = z[h];
                if (want_diag)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const bool valid = ok && (16 * h + 4 * g + v < r);
                        const double dlt = valid ? (double)zg[h][v] - (double)f[h][v] : 0.0;
                        gap[k] += dlt * dlt;
                    }
            }
        }
        cur = nxt;
    }
    if (!want_diag) return;
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

int class_of(int kind) {
    return kind == MCL_PEN_PARAFAC2 ? CLS_PF2 : (kind == MCL_PEN_UNIMODAL ? CLS_UNI : (kind == MCL_PEN_L2BALL ? CLS_L2 : CLS_ROWSEP));
}

}  // namespace

// 0 when the stack / shape of mode 1 is not covered by the software-pipelined kernels, else the signature of its stack
static int chain_signature(const mcl_context *c, bool vec) {
    if (!vec || c->sw.no_row_prefetch || c->row_sink == nullptr || c->r < 4 || c->NB > 2) return 0;
    const RegSet &rs = c->regs[1];
    if (rs.n < 1 || rs.n > 3) return 0;
    int sig = rs.n;
    for (int k = 0; k < rs.n; ++k) {
        if (rs.kind[k] == MCL_PEN_EXTERNAL || rs.kind[k] == MCL_PEN_TV || rs.kind[k] == MCL_PEN_GL2 || rs.kind[k] == MCL_PEN_SIMPLEX) return 0;
        sig |= class_of(rs.kind[k]) << (3 + 2 * k);
    }
    // BASELINE config 4: PARAFAC2 + L2 ball; config 5: PARAFAC2 + unimodality + L2 ball; config 1 / the README's model family:
    // PARAFAC2 + a row-separable kind
    if (sig == make_sig(2, CLS_PF2, CLS_L2) || sig == make_sig(3, CLS_PF2, CLS_UNI, CLS_L2) || sig == make_sig(2, CLS_PF2, CLS_ROWSEP)) return sig;
    return 0;
}

#define MCL_RC_DISPATCH(LAUNCH)                                             \
    do {                                                                    \
        if (sig == make_sig(2, CLS_PF2, CLS_L2)) {                          \
            if (rows64) LAUNCH(1, true, make_sig(2, CLS_PF2, CLS_L2));      \
            else if (c->NB == 1) LAUNCH(1, false, make_sig(2, CLS_PF2, CLS_L2)); \
            else LAUNCH(2, false, make_sig(2, CLS_PF2, CLS_L2));            \
        } else if (sig == make_sig(3, CLS_PF2, CLS_UNI, CLS_L2)) {          \
            if (rows64) LAUNCH(1, true, make_sig(3, CLS_PF2, CLS_UNI, CLS_L2)); \
            else if (c->NB == 1) LAUNCH(1, false, make_sig(3, CLS_PF2, CLS_UNI, CLS_L2)); \
            else LAUNCH(2, false, make_sig(3, CLS_PF2, CLS_UNI, CLS_L2));   \
        } else {                                                            \
            if (rows64) LAUNCH(1, true, make_sig(2, CLS_PF2, CLS_ROWSEP));  \
            else if (c->NB == 1) LAUNCH(1, false, make_sig(2, CLS_PF2, CLS_ROWSEP)); \
            else LAUNCH(2, false, make_sig(2, CLS_PF2, CLS_ROWSEP));        \
        }                                                                   \
    } while (0)

// The software-pipelined forms of the three kernels of the chained B row pass.  Each returns 1 when it has launched, 0 when the
// stack / shape is not covered (the caller launches the kernel of generic.hip).
int mcl_try_rows_chain_mid(mcl_context *c, const ModeView &mv, const float *rhs, bool vec, bool rows64) {
    const int sig = chain_signature(c, vec);
    if (!sig) return 0;
    const RegSet &rs = c->regs[1];
    const dim3 grid((unsigned)((mv.n_tiles + 3) / 4)), block(256);
#define MCL_RC_MID(NBR_, R64_, SIG_)                                                                                      \
    hipLaunchKernelGGL((k_rows_chain_mid<NBR_, R64_, SIG_>), grid, block, 0, c->stream, mv, rhs, (const float *)c->A,      \
                       (const float *)c->LinvB, rs, c->r, (const float *)c->pf2_T, (const double *)c->colsq, c->stat_gram, \
                       c->stat_colsq, (const double *)c->LinvB64, (const double *)c->pf2_T64, c->row_sink)
    MCL_RC_DISPATCH(MCL_RC_MID);
#undef MCL_RC_MID
    return 1;
}

int mcl_try_rows_chain_first(mcl_context *c, const ModeView &mv, const float *rhs, bool vec, bool rows64) {
    const int sig = chain_signature(c, vec);
    if (!sig) return 0;
    const RegSet &rs = c->regs[1];
    const dim3 grid((unsigned)((mv.n_tiles + 3) / 4)), block(256);
#define MCL_RC_FIRST(NBR_, R64_, SIG_)                                                                                    \
    hipLaunchKernelGGL((k_rows_chain_first<NBR_, R64_, SIG_>), grid, block, 0, c->stream, mv, rhs, (const float *)c->A,    \
                       (const float *)c->LinvB, rs, c->r, c->stat_gram, c->stat_colsq, (const double *)c->LinvB64, c->row_sink)
    MCL_RC_DISPATCH(MCL_RC_FIRST);
#undef MCL_RC_FIRST
    return 1;
}

int mcl_try_rows_chain_last(mcl_context *c, const ModeView &mv, bool vec, bool rows64, double *diag, int want_diag) {
    const int sig = chain_signature(c, vec);
    if (!sig) return 0;
    const RegSet &rs = c->regs[1];
    const dim3 grid((unsigned)((mv.n_tiles + 3) / 4)), block(256);
#define MCL_RC_LAST(NBR_, R64_, SIG_)                                                                                     \
    hipLaunchKernelGGL((k_rows_chain_last<NBR_, R64_, SIG_>), grid, block, 0, c->stream, mv, rs, c->r, (const float *)c->pf2_T, \
                       (const double *)c->colsq, diag, want_diag, (const double *)(rows64 ? c->pf2_T64 : nullptr), c->row_sink)
    MCL_RC_DISPATCH(MCL_RC_LAST);
#undef MCL_RC_LAST
    return 1;
}
