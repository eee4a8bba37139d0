// ONE pass over X per outer AO-ADMM iteration (gfx950), for row-separable penalties on the B_i.
//
// The reference touches X three times per iteration: X_i C for the B right-hand sides (decomposition.py:242),
// X_i^T (B_i o a_i) for the C right-hand side (:312-315) and diag(B_i^T X_i C) for the A right-hand sides (:147-152).
// With M_i = X_i^T B_i (K x r, per slab) two of them are algebra on a small matrix:
//     C-phase:  sum_i X_i^T (B_i o a_i)   = sum_i M_i diag(a_i)                       (k_reduce_weighted)
//     A-phase:  diag(B_i^T X_i C)[c]      = sum_k M_i[k][c] C[k][c]                   (k_A_rhs_from_M)
// and M_i needs the NEW B_i rows, which (row-separable prox) depend only on the same rows of X C.  So k_sweep,
// for every 16-row block of X staged ONCE in LDS:
//     X C rows (MFMA, C fragments)  ->  B-phase inner ADMM loop in registers (decomposition.py:259-285)
//     ->  B / aux / dual rows stored  ->  M_i += X_blk^T B_blk,  B_i^T B_i += B_blk^T B_blk  (MFMA)
// X C is never written to memory.  HBM traffic per iteration: S_X + S_B (1 + 4 n_reg) instead of
// 2 S_X + S_B (5 + 4 n_reg) for the two-pass path (k_contract_xc_row + k_rows_fused + k_contract_xt).
//
// Work unit = BSEG: <= bseg_rows consecutive rows of ONE slab, owned by ONE wave: the waves of a block only share the
// LDS image of C and never synchronise inside the loop.  Per bseg the wave writes, straight from its accumulators and in
// the SAME fragment order as the C image used by the X C product (k_build_cfrag, so both consumers stream them with
// 16-byte coalesced loads): M_bseg (for the A-phase), its a-weighted copy with the weighted Gram (summed over bsegs
// by k_reduce_frag for the C-phase) and B^T B.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "mcl_internal.h"
#include "rows_mfma.h"

// ---------------------------------------------------------------------------------------------------------
// lane naming: (q = l>>4, i16 = l&15).  Two register layouts of a 16 x 16 block of rows x columns occur:
//   ROW layout (inner loop, memory): lane (row = i16, g = q) holds columns 4g..4g+3 of its row       (f32x4)
//   COL layout (MFMA reduction over rows): lane (col = i16, q) reg v holds row 4q + v of its column
// MFMA16(a, b, acc): D[i][j] += sum_kk A[i][kk] B[kk][j]; lane l feeds A[l&15][l>>4] and B[l>>4][l&15]; D lane l,
// reg w = D[4(l>>4) + w][l&15].
//
// (1) X C, transposed:  A = C fragment (i <-> column c), B = X fragment (j <-> row)   -> D in ROW layout.
//     both fragments index k = 64kc + 16kq + 4q + m (kq: which float4 of the lane, m: its component)
// (2) inner loop in ROW layout (rows_mfma.h k-permutation: no cross-lane movement)
// (3) B_new ROW -> COL layout through a 16 x 20-float wave-private LDS patch (conflict-free both ways)
// (4) M += X_blk^T B_blk:  A = X[row 4q+v][4 s(i16) + m] with slot s(i) = 16kb + 4(i&3) + (i>>2),
//     B = B_new COL reg v  ->  D lane (q,i16) reg w = M[64kb + 16w + 4q + m][i16]: exactly element
//     (kc = kb, kq = w, lane, m) of the C-fragment order.
// ---------------------------------------------------------------------------------------------------------
// wave-uniform 64-bit element offset (SGPR pair): `base + uniform_off(..) + lane offset` compiles to the scalar-base +
// 32-bit lane offset form of global_load, so per-row addresses cost no vector registers.  (The offset, not the pointer,
// goes through readfirstlane: an integer -> pointer cast would lose the global address space and emit FLAT loads, which
// also count on lgkmcnt and would serialise against every LDS wait.)
static __device__ __forceinline__ long uniform_off(long v) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)v >> 32));
    return (long)(((unsigned long long)hi << 32) | lo);
}

// DBG: the MCL_SWEEP_DBG experiments (phase elimination, per-section cycle counters: tools/run_dbg.sh, sweep_cycles.py)
// are compiled into a second instantiation of the config-2/3 variants only; the production kernels carry none of it
// (the counters alone cost 12 registers in kernels that sit at the 512-register limit).
template <int KS, int NB, int NREG, int DEPTH, int NW, bool VEC, bool DBG = false, bool GRP = false, bool XNT = false>
__global__ __launch_bounds__(64 * NW) void k_sweep(const float *__restrict__ X, const float *__restrict__ Cfrag,
                                               const float *__restrict__ A, const float *__restrict__ rhoB,
                                               const float *__restrict__ LinvB, float *__restrict__ Bout, RegSet regs,
                                               const int *__restrict__ bs_slab, const int *__restrict__ bs_row0,
                                               const int *__restrict__ bs_nrows, const int *__restrict__ wave_bseg_ptr, int n_waves,
                                               int K, int r, int inner, float *__restrict__ Mpart,
                                               double *__restrict__ part_btb, float *__restrict__ GRpart,
                                               double *__restrict__ diag_block, int dbg_rt,
                                               long long *__restrict__ cyc_out, const int *__restrict__ bs_part) {
    MCL_GATE(regs.gate);
    const int dbg = DBG ? dbg_rt : 0;
    // KS = 0: the half-width form for K <= 128 (config 2) - tile rows of 128 floats, a wave load / LDS store covers TWO
    // rows (lanes 0..31 row 2t, lanes 32..63 row 2t + 1): half the MFMAs, LDS traffic and partial bytes of the 256-wide form
    constexpr bool HALF = KS == 0;
    constexpr int KSA = HALF ? 1 : KS;           // 256-column super-chunks (array extents)
    constexpr int XL = HALF ? 8 : 16;            // wave loads per super-chunk of a 16-row block
    constexpr int KW = HALF ? 128 : 256 * KS;    // floats per tile row: K rounded up (K % 4 == 0, K <= KW)
    constexpr int KC = KW / 64;                  // 64-column chunks
    constexpr int W = 16 * NB;
    constexpr int MS = KW * W;     // floats per M partial
    constexpr int NR = NREG > 0 ? NREG : 1;
    extern __shared__ float lds_dyn[];  // [4 wave tiles: 16 x K each][C fragments: K x 16 NB]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // uniform: row addresses stay in SGPRs
    const int q = lane >> 4, i16 = lane & 15;
    float *L = lds_dyn + wave * (16 * KW);
    constexpr int NT = 64 * NW;    // threads per block
    float *Cs = lds_dyn + NW * 16 * KW;  // C in fragment order: LDS reads count on lgkmcnt, so they never force the
                                        // in-flight X prefetch (vmcnt, in-order) to drain the way global loads would

    // LDS addressing: the XOR swizzle only touches the low 4 bits of the 16-byte slot index, so every access is one of
    // a few per-lane bases plus a COMPILE-TIME offset (ds_read/ds_write immediate) - no per-access address registers.
    int rd1[4], rd4[4];
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) rd1[kq] = i16 * KW + (((4 * kq + q) ^ i16) << 2);  // (1): row i16, slot 16kc + 4kq + q
    {
        const int sl = 4 * (i16 & 3) + (i16 >> 2);
#pragma unroll
        for (int v = 0; v < 4; ++v) rd4[v] = (4 * q + v) * KW + ((sl ^ (4 * q + v)) << 2);  // (4): row 4q+v, slot 16kb + sl
    }
    float bsel[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) bsel[v] = (i16 == 4 * q + v) ? 1.f : 0.f;
    // tile write: row t, slot 64 sc + lane -> physical slot (lane ^ t) in the low 4 bits.  HALF: row 2t + hi (hi = lane >> 5),
    // slot lane & 31: ((slot & 15) ^ (2t + hi)) = ((slot & 15) ^ hi) ^ 2t, so the per-lane part still precedes an immediate
    const int hi = lane >> 5;
    const int wr_lo = HALF ? (((lane & 15) ^ hi) << 2) : ((lane & 15) << 2);
    const int wr_hi = HALF ? (hi * KW + (((lane & 31) >> 4) << 6)) : ((lane >> 4) << 6);

    const long long wall_entry = (dbg & 64) ? (long long)wall_clock64() : 0;
    for (int e = threadIdx.x * 4; e < MS; e += 4 * NT)
        *reinterpret_cast<f32x4 *>(Cs + e) = *reinterpret_cast<const f32x4 *>(Cfrag + e);
    __syncthreads();

    // (dbg & 32): cycles per section, per wave; [5] = 100 MHz wall ticks.  (dbg & 64), tools/sweep_stamps.py: [0] entry, [1] start of
    // the block loop, [2] end (absolute 100 MHz ticks), [3] HW_ID, [4] XCC_ID
    long long cyc[6] = {0, 0, 0, 0, 0, 0};
    const long long wall0 = (dbg & (32 | 64)) ? (long long)wall_clock64() : 0;
    const long long core0 = (dbg & 64) ? (long long)__builtin_readcyclecounter() : 0;  // (dbg & 64): [5] = core clock cycles of the block loop
    auto tick = [&](int sec, long long &t0) {
        if (dbg & 32) {
            const long long t1 = __builtin_readcyclecounter();
            cyc[sec] += t1 - t0;
            t0 = t1;
        }
    };

    const int wg = blockIdx.x * NW + wave;  // global wave index
    const int bs0 = wave_bseg_ptr[min(wg, n_waves)];  // the wave's bsegs (balanced by blocks: mcl_set_problem)
    const int bs1 = wave_bseg_ptr[min(wg + 1, n_waves)];

    double nf = 0.0, na = 0.0, gap[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) gap[k] = 0.0;

    for (int bs = bs0; bs < bs1; ++bs) {
        const int slab = __builtin_amdgcn_readfirstlane(bs_slab[bs]);
        const long row0 = __builtin_amdgcn_readfirstlane(bs_row0[bs]);
        const int nrows = __builtin_amdgcn_readfirstlane(bs_nrows[bs]);
        const int wr0 = 0, wn = nrows;  // the wave owns the whole bseg
        const int nblk = (wn + 15) >> 4;

        f32x4 accM[KC][4][NB];
        f32x4 accG[NB][NB];
#pragma unroll
        for (int kb = 0; kb < KC; ++kb)
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) accM[kb][m][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < NB; ++a)
#pragma unroll
            for (int b = 0; b < NB; ++b) accG[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

        if (nblk > 0) {
            // per-slab operands of the inner loop (ROW layout: row = i16, g = q)
            const float *Li = LinvB + (long)slab * r * r;
            float LT[NB][NB][4];
#pragma unroll
            for (int hp = 0; hp < NB; ++hp)
#pragma unroll
                for (int h = 0; h < NB; ++h)
#pragma unroll
                    for (int kq = 0; kq < 4; ++kq) {
                        const int k = 16 * h + 4 * q + kq, c = 16 * hp + i16;
                        LT[hp][h][kq] = (k < r && c < r) ? Li[k * r + c] : 0.f;
                    }
            float av[NB][4];
#pragma unroll
            for (int h = 0; h < NB; ++h)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int col = 16 * h + 4 * q + v;
                    av[h][v] = (col < r) ? A[(long)slab * r + col] : 0.f;
                }
            const float rho = rhoB[slab];
            // branch-free prox of the row-separable penalties (penalties.py:503-586), identical results to prox_elem:
            //     prox(y) = clamp(y - clamp(y, pa, pb), plo, phi)          (clamp = v_med3_f32)
            //   NN: (-inf, 0 | -inf, inf)   Box: (0, 0 | lo, hi)   L1: (-thr, thr | -inf, inf)   L1 + NN: (-inf, thr | ..)
            float pa[NR], pb[NR], plo[NR], phi[NR];
            bool all_nn = NREG > 0;
#pragma unroll
            for (int k = 0; k < NREG; ++k) all_nn = all_nn && regs.kind[k] == MCL_PEN_NN;
#pragma unroll
            for (int k = 0; k < NR; ++k) {
                const float thr = (k < NREG) ? regs.p0[k] / rho : 0.f;
                const int kind = (k < NREG) ? regs.kind[k] : 0;
                pa[k] = pb[k] = 0.f;
                plo[k] = -INFINITY, phi[k] = INFINITY;
                if (kind == MCL_PEN_NN) pa[k] = -INFINITY;
                if (kind == MCL_PEN_BOX) plo[k] = regs.p0[k], phi[k] = regs.p1[k];
                if (kind == MCL_PEN_L1) pa[k] = regs.nonneg[k] ? -INFINITY : -thr, pb[k] = thr;
            }
            int zcol[NB];
            bool zok[NB];
#pragma unroll
            for (int h = 0; h < NB; ++h) {
                // VEC (r % 4 == 0): a lane's 4 columns are all valid or all padding, one 16-byte access; otherwise 4 scalar
                // accesses with per-element clamps / guards
                zok[h] = (16 * h + 4 * q) < r;
                zcol[h] = VEC ? min(16 * h + 4 * q, r - 4) : 16 * h + 4 * q;  // clamped: every load is unconditional
            }

            const long base = row0 + wr0;
            const unsigned lane4 = 4u * (unsigned)lane;
            // column offset of this lane in super-chunk sc, clamped into the row when K is not a multiple of 256: the
            // padding columns then hold copies of real (finite) data that only ever meet zero C fragments, and the rows
            // of M they produce (k >= K) are never read
            unsigned xcol[KSA];
#pragma unroll
            for (int sc = 0; sc < KSA; ++sc) xcol[sc] = (unsigned)min(256 * sc + 4 * (HALF ? (lane & 31) : lane), K - 4);
            f32x4 xr[DEPTH][KSA][XL];
            f32x4 zs[DEPTH][NR][NB], us[DEPTH][NR][NB];  // aux / dual rows of the slot's block, updated IN PLACE
            // Stage block `blk` of this wave into ring slot d.  Every load is unconditional (rows clamped into the
            // wave's range, scalar arithmetic): a branch around loads makes the compiler's counted s_waitcnt vmcnt(N)
            // collapse to the pessimistic merge of both paths and drains the prefetch at every block.
            // XNT: X is far larger than the last-level cache (256 MB) and read once per iteration - its loads carry the
            // non-temporal hint, so the stream does not evict B / aux / dual (100 MB at config 3), the partials and the C image
            // from the cache between the kernels of an iteration (config 3: k_sweep 161 -> 137 us on the same box).  Problems
            // that fit the cache (per-rank shards, config 2) keep ordinary loads: with the hint they ran 3-5 % slower
            auto ldx = [](const f32x4 *p) -> f32x4 { return XNT ? __builtin_nontemporal_load(p) : *p; };
            auto issue_x = [&](auto dc, int blk) {  // 16 rows x K columns, one 1 KB row segment per wave load
                constexpr int d = decltype(dc)::value;
                // scalar addressing: ONE 64-bit product per block, then a 32-bit row offset per row (row clamped into
                // the wave's range: a negative tmax - a block past the end - lands every row on the last valid one).
                // The per-row 64-bit products cost ~10 scalar instructions per row, and with one wave per SIMD every
                // instruction of the wave, scalar ones included, takes an issue slot of its own.
                const long blk_off = (base + 16 * (long)blk) * K;
                const int tmax = wn - 1 - 16 * blk;
                if (HALF) {
                    // two rows per load: the upper half-wave adds one row (K floats) unless that row lies past the end
#pragma unroll
                    for (int t = 0; t < XL; ++t) {
                        const unsigned up = (2 * t + 1 <= tmax) ? (unsigned)K : 0u;  // wave-uniform
                        xr[d][0][t] = ldx(reinterpret_cast<const f32x4 *>(X + uniform_off(blk_off + (long)(min(2 * t, tmax) * K)) +
                                                                          (xcol[0] + (hi ? up : 0u))));
                    }
                } else {
#pragma unroll
                    for (int sc = 0; sc < KSA; ++sc)
#pragma unroll
                        for (int t = 0; t < 16; ++t)
                            xr[d][sc][t] = ldx(reinterpret_cast<const f32x4 *>(X + uniform_off(blk_off + (long)(min(t, tmax) * K)) + xcol[sc]));
                }
            };
            // aux / dual rows go straight into the registers the inner loop works on.  They are issued AFTER the slot's
            // previous block has stored its rows, so the registers are never live across the load and the compiler
            // needs no loop-carried copies (a copy of a loaded register is an early wait on the in-order queue: with
            // ~36 KB per wave outstanding the queueing latency is several microseconds).  Padding columns (>= r) hold
            // clamped real data: finite, multiplied by zero rows of L^-1, never stored.
            auto issue_zu = [&](auto dc, int blk) {
                constexpr int d = decltype(dc)::value;
                const long jr = base + min(16 * blk + i16, wn - 1);
#pragma unroll
                for (int k = 0; k < NREG; ++k)
#pragma unroll
                    for (int h = 0; h < NB; ++h) {
                        if (VEC) {
                            zs[d][k][h] = *reinterpret_cast<const f32x4 *>(regs.aux[k] + jr * r + zcol[h]);
                            us[d][k][h] = *reinterpret_cast<const f32x4 *>(regs.dual[k] + jr * r + zcol[h]);
                        } else {
#pragma unroll
                            for (int v = 0; v < 4; ++v) {
                                const int cv = min(zcol[h] + v, r - 1);
                                zs[d][k][h][v] = regs.aux[k][jr * r + cv];
                                us[d][k][h][v] = regs.dual[k][jr * r + cv];
                            }
                        }
                    }
            };
            // the scheduling barriers pin the queue order slot 0 | slot 1: the loop-head s_waitcnt vmcnt(N) is ONE
            // instruction shared by the entry and the back edge, and a reordered prologue would shrink its N to ~0
            issue_x(std::integral_constant<int, 0>{}, 0);
            issue_zu(std::integral_constant<int, 0>{}, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (DEPTH == 2) {
                issue_x(std::integral_constant<int, DEPTH - 1>{}, 1);
                issue_zu(std::integral_constant<int, DEPTH - 1>{}, 1);
            }
            __builtin_amdgcn_sched_barrier(0);

            auto body = [&](auto dc, int blk) {
                constexpr int d = decltype(dc)::value;
                long long t0 = (dbg & 32) ? (long long)__builtin_readcyclecounter() : 0;
                // opaque copies of the per-lane LDS bases: the compiler must form every LDS address as base + immediate
                // inside the block instead of hoisting one precomputed address register per access out of the loop
                // (that costs ~50 registers, i.e. spills to scratch, whose reloads also sit in the vmcnt queue)
                int wl = wr_lo;
                unsigned l4 = lane4;
                asm volatile("" : "+v"(wl), "+v"(l4));
                const float *csl = Cs + l4;
                // ---- registers -> LDS tile (row t, 16-B slot 64 sc + lane, physical slot XORed with the row)
                if (HALF) {
#pragma unroll
                    for (int t = 0; t < XL; ++t)
                        *reinterpret_cast<f32x4 *>(L + 2 * t * KW + wr_hi + (wl ^ ((2 * t) << 2))) = xr[d][0][t];
                } else {
#pragma unroll
                    for (int sc = 0; sc < KSA; ++sc)
#pragma unroll
                        for (int t = 0; t < 16; ++t)
                            *reinterpret_cast<f32x4 *>(L + t * KW + 256 * sc + wr_hi + (wl ^ (t << 2))) = xr[d][sc][t];
                }
                f32x4(&z)[NR][NB] = zs[d];
                f32x4(&u)[NR][NB] = us[d];
                // the X slot is free again: its next block goes out now and stays in flight for DEPTH block times
                issue_x(dc, blk + DEPTH);
                tick(0, t0);

                // ---- (1) rhs^T = (X C)^T for the 16 rows, ROW layout
                f32x4 acc[NB], acc2[NB];  // two chains: consecutive MFMAs never wait on each other
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[nb] = acc2[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
                // C and X fragments of the next 64-column chunk are fetched from LDS while the current chunk multiplies
                // (double buffer by hand); the scheduling barriers keep the compiler from hoisting every LDS read of the
                // block to the top (register pressure: the M accumulators and the staging registers take half the file)
                {
                    constexpr int DB = (KS <= 1 && NB == 1 && NW == 4) ? 2 : 1;  // no registers left for a second buffer otherwise
                    f32x4 cf[DB][4][NB], fr[DB][4];
                    auto ld1 = [&](int sl, int kc) {
#pragma unroll
                        for (int kq = 0; kq < 4; ++kq) {
#pragma unroll
                            for (int nb = 0; nb < NB; ++nb)
                                cf[sl][kq][nb] = *reinterpret_cast<const f32x4 *>(csl + ((kc * 4 + kq) * NB + nb) * 256);
                            fr[sl][kq] = *reinterpret_cast<const f32x4 *>(L + rd1[kq] + 64 * kc);
                        }
                    };
                    if (DB == 2) ld1(0, 0);
                    if (!(dbg & 1))
#pragma unroll
                    for (int kc = 0; kc < KC; ++kc) {
                        if (DB == 1) ld1(0, kc);
                        else if (kc + 1 < KC) ld1((kc + 1) & 1, kc + 1);
#pragma unroll
                        for (int kq = 0; kq < 4; ++kq)
#pragma unroll
                            for (int m = 0; m < 4; ++m)
#pragma unroll
                                for (int nb = 0; nb < NB; ++nb) {
                                    constexpr int sb = 0;
                                    const int sl = (DB == 2) ? (kc & 1) : sb;
                                    if (m & 1) acc2[nb] = MFMA16(cf[sl][kq][nb][m], fr[sl][kq][m], acc2[nb]);
                                    else acc[nb] = MFMA16(cf[sl][kq][nb][m], fr[sl][kq][m], acc[nb]);
                                }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[nb] += acc2[nb];

                tick(1, t0);
                // ---- (2) inner ADMM loop (decomposition.py:259-285), same arithmetic as rows_fused_tile
                f32x4 rhs[NB], f[NB];
#pragma unroll
                for (int h = 0; h < NB; ++h) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) rhs[h][v] = acc[h][v] * av[h][v];
                    f[h] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                const int n_it = (dbg & 2) ? 0 : ((NREG == 0 && inner > 1) ? 1 : inner);
                // one inner iteration; NNONLY: every penalty is a plain non-negativity constraint, whose prox is one v_max
                auto inner_iter = [&](auto nn_only) {
                    constexpr bool NNONLY = decltype(nn_only)::value;
                    f32x4 t[NB];
#pragma unroll
                    for (int h = 0; h < NB; ++h)
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            float s = (NREG > 0) ? z[0][h][v] - u[0][h][v] : 0.f;
#pragma unroll
                            for (int k = 1; k < NREG; ++k) s += z[k][h][v] - u[k][h][v];
                            t[h][v] = (NREG > 0) ? fmaf(rho, s, rhs[h][v]) : rhs[h][v];
                        }
#pragma unroll
                    for (int hp = 0; hp < NB; ++hp) {
                        f32x4 a4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int h = 0; h < NB; ++h)
#pragma unroll
                            for (int kq = 0; kq < 4; ++kq) a4 = MFMA16(LT[hp][h][kq], t[h][kq], a4);
                        f[hp] = a4;
                    }
#pragma unroll
                    for (int k = 0; k < NREG; ++k)
#pragma unroll
                        for (int h = 0; h < NB; ++h)
#pragma unroll
                            for (int v = 0; v < 4; ++v) {
                                const float y = f[h][v] + u[k][h][v];
                                const float znew =
                                    NNONLY ? fmaxf(y, 0.f)
                                           : __builtin_amdgcn_fmed3f(y - __builtin_amdgcn_fmed3f(y, pa[k], pb[k]), plo[k], phi[k]);
                                u[k][h][v] = f[h][v] - (znew - u[k][h][v]);
                                z[k][h][v] = znew;
                            }
                };
                if (all_nn && NREG == 1 && n_it > 0 && !(dbg & 64)) {  // MCL_SWEEP_DBG=64: the general form below
                    // One non-negativity constraint: with y = B + U the prox and dual steps are Z = max(y, 0), U = min(y, 0)
                    // (= B - (Z - U) without its roundings), so Z - U = |y| exactly and the next right-hand side is
                    // rhs + rho |y|: two instructions per element and iteration (v_min, v_fma with the |.| modifier) besides
                    // the add, Z itself only after the last iteration.
                    f32x4 t[NB], y[NB];
#pragma unroll
                    for (int h = 0; h < NB; ++h)
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            t[h][v] = fmaf(rho, z[0][h][v] - u[0][h][v], rhs[h][v]);
                            y[h][v] = 0.f;
                        }
                    for (int it = 0; it < n_it; ++it) {
#pragma unroll
                        for (int hp = 0; hp < NB; ++hp) {
                            f32x4 a4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                            for (int h = 0; h < NB; ++h)
#pragma unroll
                                for (int kq = 0; kq < 4; ++kq) a4 = MFMA16(LT[hp][h][kq], t[h][kq], a4);
                            f[hp] = a4;
                        }
#pragma unroll
                        for (int h = 0; h < NB; ++h)
#pragma unroll
                            for (int v = 0; v < 4; ++v) {
                                y[h][v] = f[h][v] + u[0][h][v];
                                u[0][h][v] = fminf(y[h][v], 0.f);
                                t[h][v] = fmaf(rho, fabsf(y[h][v]), rhs[h][v]);
                            }
                    }
#pragma unroll
                    for (int h = 0; h < NB; ++h)
#pragma unroll
                        for (int v = 0; v < 4; ++v) z[0][h][v] = fmaxf(y[h][v], 0.f);
                } else if (all_nn) {
                    for (int it = 0; it < n_it; ++it) inner_iter(std::true_type{});
                } else {
                    for (int it = 0; it < n_it; ++it) inner_iter(std::false_type{});
                }

                tick(2, t0);
                // ---- store the rows, diagnostics, ROW -> COL transposition of B_new
                const bool ok = 16 * blk + i16 < wn;
                const long j = base + 16 * blk + i16;
#pragma unroll
                for (int h = 0; h < NB; ++h) {
                    const bool live = ok && zok[h];
                    const int col = 16 * h + 4 * q;
                    if (live && !(dbg & 8)) {
                        if (VEC) {
                            *reinterpret_cast<f32x4 *>(Bout + j * r + col) = f[h];
#pragma unroll
                            for (int k = 0; k < NREG; ++k) {
                                *reinterpret_cast<f32x4 *>(regs.aux[k] + j * r + col) = z[k][h];
                                *reinterpret_cast<f32x4 *>(regs.dual[k] + j * r + col) = u[k][h];
                            }
                        } else {
#pragma unroll
                            for (int v = 0; v < 4; ++v)
                                if (col + v < r) {
                                    Bout[j * r + col + v] = f[h][v];
#pragma unroll
                                    for (int k = 0; k < NREG; ++k) {
                                        regs.aux[k][j * r + col + v] = z[k][h][v];
                                        regs.dual[k][j * r + col + v] = u[k][h][v];
                                    }
                                }
                        }
                        // 4-term partial sums in fp32, accumulated across blocks in fp64 (f is exactly 0 in padding columns)
                        float s_nf = 0.f, s_na = 0.f, s_gap[NR];
#pragma unroll
                        for (int k = 0; k < NR; ++k) s_gap[k] = 0.f;
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            s_nf = fmaf(f[h][v], f[h][v], s_nf);
                            s_na += fabsf(f[h][v]);
#pragma unroll
                            for (int k = 0; k < NREG; ++k) {
                                const float dlt = (VEC || col + v < r) ? z[k][h][v] - f[h][v] : 0.f;
                                s_gap[k] = fmaf(dlt, dlt, s_gap[k]);
                            }
                        }
                        nf += (double)s_nf;
                        na += (double)s_na;
#pragma unroll
                        for (int k = 0; k < NREG; ++k) gap[k] += (double)s_gap[k];
                    }
                    if (!live) f[h] = f32x4{0.f, 0.f, 0.f, 0.f};  // padding rows / columns must not reach M
                }
                issue_zu(dc, blk + DEPTH);  // rows of this slot's next block, into the registers just stored
                // ROW -> COL layout on the matrix core: with A = f[nb][v] (A[row][kk] = B_new[row][4kk + v]) and the constant
                // selector B_v[kk][j] = (j == 4kk + v), sum_v A_v B_v = B_new, and the MFMA result layout IS the COL
                // layout (lane (q, i16) reg w = B_new[4q + w][i16]).  Exact: products with 1 and sums with 0.
                float bt[NB][4];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    f32x4 tr = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int v = 0; v < 4; ++v) tr = MFMA16(f[nb][v], bsel[v], tr);
#pragma unroll
                    for (int w = 0; w < 4; ++w) bt[nb][w] = tr[w];
                }

                tick(3, t0);
                // ---- (4) M += X_blk^T B_blk ; B^T B += B_blk^T B_blk
                {
                    constexpr int DB = (KS <= 1 && NB == 1 && NW == 4) ? 2 : 1;
                    f32x4 xa[DB][4];
                    auto ld4x = [&](int sl, int kb) {
#pragma unroll
                        for (int v = 0; v < 4; ++v) xa[sl][v] = *reinterpret_cast<const f32x4 *>(L + rd4[v] + 64 * kb);
                    };
                    if (DB == 2) ld4x(0, 0);
                    if (!(dbg & 4))
#pragma unroll
                    for (int kb = 0; kb < KC; ++kb) {
                        if (DB == 1) ld4x(0, kb);
                        else if (kb + 1 < KC) ld4x((kb + 1) & 1, kb + 1);
#pragma unroll
                        for (int v = 0; v < 4; ++v)
#pragma unroll
                            for (int m = 0; m < 4; ++m)
#pragma unroll
                                for (int nb = 0; nb < NB; ++nb)
                                    accM[kb][m][nb] = MFMA16(xa[(DB == 2) ? (kb & 1) : 0][v][m], bt[nb][v], accM[kb][m][nb]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
#pragma unroll
                for (int v = 0; v < 4; ++v)
#pragma unroll
                    for (int a = 0; a < NB; ++a)
#pragma unroll
                        for (int b = 0; b < NB; ++b) accG[a][b] = MFMA16(bt[a][v], bt[b][v], accG[a][b]);
                tick(4, t0);
            };
            // branch-free ring: with DEPTH == 2 an odd tail runs one dummy block (all rows invalid: nothing stored,
            // zero contribution to M)
            for (int blk = 0; blk < nblk; blk += DEPTH) {
                body(std::integral_constant<int, 0>{}, blk);
                if (DEPTH == 2) body(std::integral_constant<int, DEPTH - 1>{}, blk + 1);
            }
        }

        // ---- flush this bseg straight from the accumulators (fragment order: 1 KB per store instruction).
        // The partial a bseg goes to is the planner's (mcl_set_problem): normally its own; GRP - short bsegs, every wave of
        // the workgroup holding ONE bseg, and all four (or the two of an aligned pair) of the SAME slab - the waves of a
        // group first add their accumulators through the LDS tiles (no longer read by anybody) in the fixed order
        // ((w0 + w1) + w2) + w3 and flush ONE partial: a quarter (half) of the partial traffic of the sweep, of k_reduce_frag
        // and of the A-phase on the per-rank shards of a multi-GPU run and on config 2, where a bseg is 64 rows
        {
            const int pinfo = __builtin_amdgcn_readfirstlane(bs_part[bs]);
            const int part = pinfo & 0x0fffffff;
            // bit 28: the workgroup runs the cooperative flush (workgroup-uniform); bits 29-30: this wave's group is a pair /
            // all four waves (waves g0 .. g0 + gs - 1 hold bsegs of one slab; a wave outside any group flushes on its own
            // but keeps the workgroup's barriers company)
            const bool coop = GRP && ((pinfo >> 28) & 1) != 0;
            const int gs = GRP ? (1 << ((pinfo >> 29) & 3)) : 1;
            const int g0 = wave & ~(gs - 1);
            float *mp = Mpart + (long)part * MS;
            float *gp = GRpart + (long)part * (W * W + W);  // [weighted Gram | a_i]: k_reduce_frag weights M_part itself
            const float *arow = A + (long)slab * r;
            float a_c[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) a_c[nb] = (16 * nb + i16 < r) ? arow[16 * nb + i16] : 0.f;
            constexpr int NI = KC * 4 * NB;  // 1 KB fragments of a partial
            f32x4 gsum[NB][NB];
#pragma unroll
            for (int a = 0; a < NB; ++a)
#pragma unroll
                for (int b = 0; b < NB; ++b) gsum[a][b] = accG[a][b];
            bool writer = true;  // this wave writes the Gram / weights of the partial
            auto flush_own = [&]() {
#pragma unroll
                for (int kb = 0; kb < KC; ++kb)
#pragma unroll
                    for (int w = 0; w < 4; ++w)
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) {
                            const f32x4 val = {accM[kb][0][nb][w], accM[kb][1][nb][w], accM[kb][2][nb][w], accM[kb][3][nb][w]};
                            const int e = ((((kb * 4 + w) * NB + nb) * 64 + lane) << 2);
                            *reinterpret_cast<f32x4 *>(mp + e) = val;
                        }
            };
            if (coop) {
                static_assert(!GRP || NI % NW == 0, "fragments split evenly over the waves");
                if (gs > 1) {
#pragma unroll
                    for (int kb = 0; kb < KC; ++kb)
#pragma unroll
                        for (int w = 0; w < 4; ++w)
#pragma unroll
                            for (int nb = 0; nb < NB; ++nb) {
                                const f32x4 val = {accM[kb][0][nb][w], accM[kb][1][nb][w], accM[kb][2][nb][w], accM[kb][3][nb][w]};
                                *reinterpret_cast<f32x4 *>(L + ((((kb * 4 + w) * NB + nb) * 64 + lane) << 2)) = val;
                            }
                } else {
                    flush_own();
                }
                __syncthreads();
                if (gs > 1) {
                    const int per = NI / gs;  // fragments this wave sums: the group's tiles in the fixed order g0, g0 + 1, ...
                    for (int ii = 0; ii < per; ++ii) {
                        const int e = (((wave - g0) * per + ii) * 64 + lane) << 2;
                        f32x4 t = *reinterpret_cast<const f32x4 *>(lds_dyn + g0 * (16 * KW) + e);
                        for (int wv = 1; wv < gs; ++wv) t += *reinterpret_cast<const f32x4 *>(lds_dyn + (g0 + wv) * (16 * KW) + e);
                        *reinterpret_cast<f32x4 *>(mp + e) = t;
                    }
                }
                __syncthreads();
                if (gs > 1) {
#pragma unroll
                    for (int a = 0; a < NB; ++a)
#pragma unroll
                        for (int b = 0; b < NB; ++b) *reinterpret_cast<f32x4 *>(L + (((a * NB + b) * 64 + lane) << 2)) = accG[a][b];
                }
                __syncthreads();
                writer = wave == g0;
                if (writer && gs > 1) {
#pragma unroll
                    for (int a = 0; a < NB; ++a)
#pragma unroll
                        for (int b = 0; b < NB; ++b) {
                            const int eo = ((a * NB + b) * 64 + lane) << 2;
                            f32x4 t = *reinterpret_cast<const f32x4 *>(lds_dyn + g0 * (16 * KW) + eo);
                            for (int wv = 1; wv < gs; ++wv) t += *reinterpret_cast<const f32x4 *>(lds_dyn + (g0 + wv) * (16 * KW) + eo);
                            gsum[a][b] = t;
                        }
                }
            } else {
                flush_own();
            }
            if (writer) {
                if (q == 0) {
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) gp[W * W + 16 * nb + i16] = a_c[nb];  // the a_i of the moment (0 in padding)
                }
#pragma unroll
                for (int a = 0; a < NB; ++a)
#pragma unroll
                    for (int b = 0; b < NB; ++b)
#pragma unroll
                        for (int w = 0; w < 4; ++w) {
                            const int ra = 16 * a + 4 * q + w, cb = 16 * b + i16;
                            if (ra < r && cb < r) {
                                part_btb[((long)part * r + ra) * r + cb] = (double)gsum[a][b][w];
                                gp[ra * W + cb] = arow[ra] * arow[cb] * gsum[a][b][w];
                            } else {
                                gp[ra * W + cb] = 0.f;
                            }
                        }
            }
        }
    }

    if (dbg & 32) cyc[5] = (long long)wall_clock64() - wall0;
    if (dbg & 64) {
        cyc[0] = wall_entry, cyc[1] = wall0, cyc[2] = (long long)wall_clock64();
        cyc[5] = (long long)__builtin_readcyclecounter() - core0;
        cyc[3] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 4), cyc[4] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 20);
    }
    if ((dbg & (32 | 64)) && lane == 0) {
#pragma unroll
        for (int i = 0; i < 6; ++i) cyc_out[((long)blockIdx.x * NW + wave) * 6 + i] = cyc[i];
    }
    // one diagnostics row per block
    nf = wave_sum(nf);
    na = wave_sum(na);
#pragma unroll
    for (int k = 0; k < NR; ++k) gap[k] = wave_sum(gap[k]);
    __syncthreads();  // every wave is done with its tile: the area is free
    double(*dsm)[DIAG_COLS] = reinterpret_cast<double(*)[DIAG_COLS]>(lds_dyn);
    if (lane == 0) {
        dsm[wave][0] = nf;
        dsm[wave][1] = na;
#pragma unroll
        for (int k = 0; k < NR; ++k) dsm[wave][2 + k] = gap[k];
    }
    __syncthreads();
    if (threadIdx.x < 2 + NREG)
    {
        double t = 0.0;
#pragma unroll
        for (int wv = 0; wv < NW; ++wv) t += dsm[wv][threadIdx.x];  // fixed order
        diag_block[(long)blockIdx.x * DIAG_COLS + threadIdx.x] = t;
    }
}

// ---------------------------------------------------------------------------------------------------------
// [G | R] = sum over the sweep's per-bseg a-weighted partials (decomposition.py:312-318).  Thread e walks the partials in fragment
// order (coalesced) and scatters its sum to the row-major [G | R] image.  Fixed summation order (16 interleaved groups
// of partials, 4 chains each, then the groups in order): deterministic, identical on every rank for identical input.
// ---------------------------------------------------------------------------------------------------------
// A deferred diagnostics reduction (mcl_diagnostics_deferred) rides on TWO spare workgroups of this kernel: each of their
// 32 waves takes one of the 3 DIAG_COLS + 2 column sums of k_diag_final (lanes stride the rows, butterfly sum: fixed
// order) while the other workgroups reduce [G | R] - no launch of its own on the critical path, and no wave with more
// than one dependent chain of loads (with one spare workgroup the kernel grew from 5.8 to 8.3 us).
#define MCL_PIGGY_BLOCKS 2
struct DiagPiggy {
    DiagTables T;
    double *out;  // nullptr: no spare workgroup was launched
    int include_replicated;
};

static __device__ void diag_piggy_block(const DiagPiggy &P, int which) {
    const int lane = threadIdx.x & 63, n_waves = MCL_PIGGY_BLOCKS * (blockDim.x >> 6);
    const int wave = which * (blockDim.x >> 6) + (threadIdx.x >> 6);
    constexpr int NS = 3 * DIAG_COLS + 2;
    for (int b = wave; b < NS; b += n_waves) {
        const double *tab;
        int rows, ncols, col, t = -1;
        if (b < 3 * DIAG_COLS) {
            t = b / DIAG_COLS, col = b - t * DIAG_COLS;
            tab = P.T.tab[t], rows = (t < 2 || P.include_replicated) ? P.T.rows[t] : 0, ncols = DIAG_COLS;
        } else {
            col = b - 3 * DIAG_COLS, tab = P.T.e1, rows = P.T.I, ncols = 2;
        }
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int e = lane;
        for (; e + 192 < rows; e += 256) {
            s0 += tab[(long)e * ncols + col];
            s1 += tab[(long)(e + 64) * ncols + col];
            s2 += tab[(long)(e + 128) * ncols + col];
            s3 += tab[(long)(e + 192) * ncols + col];
        }
        for (; e < rows; e += 64) s0 += tab[(long)e * ncols + col];
        const double s = wave_sum((s0 + s1) + (s2 + s3));
        if (lane != 0) continue;
        double *out = P.out;
        if (t < 0) {
            out[col == 0 ? MCL_DIAG_INNER : MCL_DIAG_MODEL_SQ] = s;
        } else if (col == 0) {
            out[MCL_DIAG_NORM_SQ + t] = s;
        } else if (col == 1) {
            for (int k = 0; k < MCL_MAX_REGS; ++k) out[MCL_DIAG_REG + (t * MCL_MAX_REGS + k) * 2 + 1] = (k < P.T.nreg[t]) ? s : 0.0;
        } else {
            const int k = col - 2;
            out[MCL_DIAG_REG + (t * MCL_MAX_REGS + k) * 2] = (k < P.T.nreg[t]) ? s : 0.0;
        }
    }
    if (threadIdx.x == 0 && which == 0) {
        P.out[MCL_DIAG_X_SQ] = P.T.xsq[0];
        P.out[6] = 0.0;
        P.out[7] = 0.0;
    }
}

template <int EL>  // elements per block (64: 256-byte wave loads, PS / 64 blocks; 32: twice the blocks for small PS)
__global__ __launch_bounds__(1024) void k_reduce_frag(const float *__restrict__ Mpart, const float *__restrict__ GRpart,
                                                      int n_part, int K, int r, int NB, int MS, double *__restrict__ GR,
                                                      DiagPiggy piggy) {
    if (piggy.out != nullptr && blockIdx.x >= gridDim.x - MCL_PIGGY_BLOCKS) {  // the spare workgroups
        diag_piggy_block(piggy, (int)(gridDim.x - 1 - blockIdx.x));
        return;
    }
    constexpr int NG = 1024 / EL;  // interleaved groups of partials
    __shared__ double sm[NG][EL];  // fp32 per-bseg partials, summed in fp64 (the C-phase system sees >= 1e-8 inputs)
    const int el = threadIdx.x % EL, pc = threadIdx.x / EL;
    const int e = blockIdx.x * EL + el;
    const int W = 16 * NB, PS = MS + W * W;
    const int GS = W * W + W;  // per-bseg row of GRpart: the a-weighted Gram, then the a_i the sweep ran with
    int out = -1, wcol = 0;    // R = sum_bsegs M_bseg diag(a_i): the fp32 product M a is the one the sweep used to store
    if (e < MS) {
        wcol = W * W + 16 * ((e >> 8) % NB) + ((e >> 2) & 15);
        const int m = e & 3, ln = (e >> 2) & 63;
        int t = e >> 8;
        const int nb = t % NB;
        t /= NB;
        const int kq = t & 3, kc = t >> 2;
        const int k = 64 * kc + 16 * kq + 4 * (ln >> 4) + m, c = 16 * nb + (ln & 15);
        if (k < K && c < r) out = r * r + k * r + c;
    } else if (e < PS) {
        const int g = e - MS, a = g / W, b = g - a * W;
        if (a < r && b < r) out = a * r + b;
    }
    double s = 0.0;
    if (out >= 0) {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int p = pc;
        if (e < MS) {
            for (; p + 3 * NG < n_part; p += 4 * NG) {
                const float m0 = Mpart[(long)p * MS + e], m1 = Mpart[(long)(p + NG) * MS + e];
                const float m2 = Mpart[(long)(p + 2 * NG) * MS + e], m3 = Mpart[(long)(p + 3 * NG) * MS + e];
                const float a0 = GRpart[(long)p * GS + wcol], a1 = GRpart[(long)(p + NG) * GS + wcol];
                const float a2 = GRpart[(long)(p + 2 * NG) * GS + wcol], a3 = GRpart[(long)(p + 3 * NG) * GS + wcol];
                s0 += (double)(m0 * a0);
                s1 += (double)(m1 * a1);
                s2 += (double)(m2 * a2);
                s3 += (double)(m3 * a3);
            }
            for (; p < n_part; p += NG) s0 += (double)(Mpart[(long)p * MS + e] * GRpart[(long)p * GS + wcol]);
        } else {
            const int g = e - MS;
            for (; p + 3 * NG < n_part; p += 4 * NG) {
                s0 += (double)GRpart[(long)p * GS + g];
                s1 += (double)GRpart[(long)(p + NG) * GS + g];
                s2 += (double)GRpart[(long)(p + 2 * NG) * GS + g];
                s3 += (double)GRpart[(long)(p + 3 * NG) * GS + g];
            }
            for (; p < n_part; p += NG) s0 += (double)GRpart[(long)p * GS + g];
        }
        s = (s0 + s1) + (s2 + s3);
    }
    sm[pc][el] = s;
    __syncthreads();
    if (pc == 0 && out >= 0) {
        double t = 0.0;
#pragma unroll
        for (int g = 0; g < NG; ++g) t += sm[g][el];  // fixed order
        GR[out] = t;
    }
}

// ---------------------------------------------------------------------------------------------------------
// rhs of the A-phase per bseg: seg_rhs[bseg][c] = sum_k M_bseg[k][c] C[k][c] with both operands in fragment order
// (decomposition.py:147-152: diag(B_i^T X_i C)).  One block per bseg; thread t owns column 16 nb + (t & 15) of
// wave-row (t >> 6) = nb (mod NB); fixed-order LDS tree.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_A_rhs_from_M(const float *__restrict__ Mpart, const float *__restrict__ Cfrag,
                                                      int MS, int NB, int r, double *__restrict__ seg_rhs) {
    __shared__ double sm[256];
    const int bs = blockIdx.x;
    const float *mp = Mpart + (long)bs * MS;
    double s0 = 0.0, s1 = 0.0;  // products of fp32 values are exact in fp64
    auto dot4 = [](const f32x4 a, const f32x4 b, double s) {
        s = fma((double)a[0], (double)b[0], s);
        s = fma((double)a[1], (double)b[1], s);
        s = fma((double)a[2], (double)b[2], s);
        return fma((double)a[3], (double)b[3], s);
    };
    int e = threadIdx.x * 4;
    for (; e + 1024 < MS; e += 2048) {
        const f32x4 a = *reinterpret_cast<const f32x4 *>(mp + e), b = *reinterpret_cast<const f32x4 *>(Cfrag + e);
        const f32x4 c = *reinterpret_cast<const f32x4 *>(mp + e + 1024), d = *reinterpret_cast<const f32x4 *>(Cfrag + e + 1024);
        s0 = dot4(a, b, s0);
        s1 = dot4(c, d, s1);
    }
    for (; e < MS; e += 1024) {
        const f32x4 a = *reinterpret_cast<const f32x4 *>(mp + e), b = *reinterpret_cast<const f32x4 *>(Cfrag + e);
        s0 = dot4(a, b, s0);
    }
    sm[threadIdx.x] = s0 + s1;
    __syncthreads();
    // threads with the same column: lane quarters q = 0..3 of every wave w with w % NB == nb
    if ((int)threadIdx.x < 16 * NB) {
        const int nb = threadIdx.x >> 4, c16 = threadIdx.x & 15;
        double t = 0.0;
        for (int w = nb; w < 4; w += NB)
            for (int qq = 0; qq < 4; ++qq) t += sm[w * 64 + qq * 16 + c16];
        const int col = 16 * nb + c16;
        if (col < r) seg_rhs[(long)bs * r + col] = t;
    }
}

// =========================================================================================================
// host side
// =========================================================================================================
// Shapes the sweep kernel is instantiated for: K <= 512 and K % 4 == 0 (tile rows of 256 or 512 floats in LDS, 16-byte
// row accesses of X; other K are zero-padded through the C fragments), Kpad * NB <= 512 (accumulator registers).
int mcl_sweep_KS(const mcl_context *c) { return (int)((c->K + 255) / 256); }
// 64-column chunks of a tile row / of the M partials: 2 for the half-width kernels (K <= 128, rank <= 16), else 4 per 256
// (decided once per problem, mcl_set_problem: the partial buffers are sized with it)
int mcl_sweep_KC(const mcl_context *c) { return c->sweep_kc; }

bool mcl_sweep_shape_ok(const mcl_context *c) {
    if (c->sw.no_sweep) return false;
    if (c->K < 4 || c->K > 512 || c->K % 4 != 0) return false;
    if (c->NB > 2 || mcl_sweep_KS(c) * c->NB > 2) return false;
    if (c->N == 0 || c->I == 0) return false;
    if (c->N / c->I < 64) return false;  // tiny slabs: the per-bseg flush would dominate
    return true;
}

bool mcl_sweep_eligible(const mcl_context *c) {
    if (!c->sweep_planned || !mcl_sweep_shape_ok(c)) return false;
    if (c->regs[1].n == 0 || c->regs[1].n > 2 || !mcl_mode_is_row_separable(c, 1)) return false;  // n = 0: fp64 solve
    if (c->opt.inner_n_iter_max <= 0) return false;
    if (c->opt.inner_tol > 0.0) return false;  // the inner stopping test needs a launch per inner iteration
    if (reinterpret_cast<uintptr_t>(c->X) & 15) return false;
    return true;
}

static inline int sweep_MS(const mcl_context *c) { return mcl_sweep_KC(c) * 64 * 16 * c->NB; }

template <int KS, int NB, int NREG, int NW, int DEPTH, bool VEC>
static int launch_sweep_v(mcl_context *c) {
    const int n = c->bsegs.n_tiles;
    const int n_waves = c->n_bseg_waves;  // <= 1024 = one per SIMD (the register file and the LDS tiles allow one or two)
    const int grid = (n_waves + NW - 1) / NW;
    constexpr int KWH = KS == 0 ? 128 : 256 * KS;
    const size_t sm = sizeof(float) * (size_t)(NW * 16 * KWH + KWH * 16 * NB);  // up to the full 160 KB
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_sweep<KS, NB, NREG, DEPTH, NW, VEC>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm) != hipSuccess) {
        (void)hipGetLastError();
        return -1;  // caller falls back to the two-pass path
    }
    // the sweep alternates between two diagnostics tables of mode 1: a deferred reduction of the previous iteration's
    // tables (mcl_diagnostics_deferred) may still be waiting for the coming C-phase reduction kernel
    c->diagB_parity ^= 1;
    c->diagB_tile = c->diagB_bufs[c->diagB_parity];
#define MCL_SWEEP_LAUNCH(DBG_, GRP_, NT_)                                                                              \
    hipLaunchKernelGGL((k_sweep<KS, NB, NREG, DEPTH, NW, VEC, DBG_, GRP_, NT_>), dim3(grid), dim3(64 * NW), sm, c->stream, c->X,    \
                       c->CfragS, c->A, c->rhoB, c->LinvB, c->B, c->regs[1], c->bsegs.slab, c->bsegs.row0, c->bsegs.nrows, \
                       c->wave_bseg_ptr, n_waves, (int)c->K, c->r, c->opt.inner_n_iter_max, c->Mpart, c->part_btb, c->GRpart, c->diagB_tile, \
                       c->sw.sweep_dbg, c->sweep_cycles, c->bseg_part)
    bool launched = false;
    if constexpr (KS == 1 && NB == 1 && VEC) {  // the instrumented twin exists for the config-2/3 variants only
        if (c->sw.sweep_dbg != 0 && c->n_parts == n) {  // (the twin has no grouped flush: a plan with grouped partials keeps GRP)
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_sweep<KS, NB, NREG, DEPTH, NW, VEC, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm) == hipSuccess) {
                MCL_SWEEP_LAUNCH(true, false, false);
                launched = true;
            }
        }
    }
    if constexpr (KS <= 1 && NB == 1) {  // the instantiations with the grouped flush (mcl_set_problem only groups for them)
        if (!launched && c->n_parts < n) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_sweep<KS, NB, NREG, DEPTH, NW, VEC, false, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm) != hipSuccess) {
                (void)hipGetLastError();
                return -1;
            }
            MCL_SWEEP_LAUNCH(false, true, false);
            launched = true;
        }
    }
    if (!launched) {
        if (c->n_parts < n) {
            c->err = "k_sweep: grouped partials planned for a kernel without the grouped flush";
            return 1;
        }
        if (c->x_streams) {  // X does not fit the last-level cache: non-temporal loads of X
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_sweep<KS, NB, NREG, DEPTH, NW, VEC, false, false, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm) != hipSuccess) {
                (void)hipGetLastError();
                return -1;
            }
            MCL_SWEEP_LAUNCH(false, false, true);
        } else {
            MCL_SWEEP_LAUNCH(false, false, false);
        }
    }
#undef MCL_SWEEP_LAUNCH
    MCL_CHECK_HIP(c, hipGetLastError());
    c->diag_rows[1] = grid;
    c->n_grpart = c->n_parts;  // one partial per bseg, or per group of four bsegs of a slab
    char buf[96];
    snprintf(buf, sizeof buf, "k_sweep<KS=%d,NB=%d,NREG=%d,DEPTH=%d,NW=%d,VEC=%d>", KS, NB, NREG, DEPTH, NW, VEC ? 4 : 1);
    c->variant[MCL_PROF_SWEEP] = buf;
    return 0;
}

template <int KS, int NB, int NREG, int NW, int DEPTH>
static int launch_sweep_w(mcl_context *c) {
    // 16-byte row accesses of B / aux / dual need r % 4 == 0 and 16-byte aligned bases
    auto al = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    bool vec = (c->r % 4 == 0) && al(c->B);
    for (int k = 0; k < c->regs[1].n; ++k) vec = vec && al(c->regs[1].aux[k]) && al(c->regs[1].dual[k]);
    if (vec) return launch_sweep_v<KS, NB, NREG, NW, DEPTH, true>(c);
    return launch_sweep_v<KS, NB, NREG, NW, DEPTH, false>(c);
}

template <int KS, int NB, int NREG>
static int launch_sweep_t(mcl_context *c) {
    // 4 waves per block = one per SIMD (the kernel needs > 256 registers); K = 256, r <= 16 has room for a second
    // staging slot.  (Measured: 8 leaner waves per block = two per SIMD run no faster - fp32 MFMA and VALU work of two
    // waves do not co-execute (SQ_VALU_MFMA_COEXEC_CYCLES = 0) - and double the per-bseg partial traffic.)
    return launch_sweep_w<KS, NB, NREG, 4, (KS <= 1 && NB == 1) ? 2 : 1>(c);
}

int mcl_launch_sweep(mcl_context *c) {
    ProfScope prof(c, MCL_PROF_SWEEP);
    const int ks = mcl_sweep_KS(c), n = c->regs[1].n;
#define MCL_SW(KS_, NB_)                                  \
    switch (n) {                                          \
        case 1: return launch_sweep_t<KS_, NB_, 1>(c);    \
        default: return launch_sweep_t<KS_, NB_, 2>(c);   \
    }
    if (c->NB == 1) {
        if (mcl_sweep_KC(c) == 2) { MCL_SW(0, 1) }
        if (ks == 1) { MCL_SW(1, 1) }
        MCL_SW(2, 1)
    }
    MCL_SW(1, 2)
#undef MCL_SW
}

int mcl_launch_reduce_weighted(mcl_context *c) {
    const int MS = sweep_MS(c), W = 16 * c->NB;
    // 64 elements per block need PS / 64 blocks (68 at K = 256, rank 16: a quarter of the CUs); with 32 there are twice as many
    int el = (MS + W * W) / 64 >= 192 ? 64 : 32;
    if (c->sw.reduce_el > 0) el = c->sw.reduce_el == 64 ? 64 : (c->sw.reduce_el == 16 ? 16 : 32);
    DiagPiggy piggy{};
    if (c->diag_pending) {  // a deferred diagnostics reduction takes a spare workgroup of this launch
        piggy.T = c->diag_pending_T, piggy.out = c->diag_pending_out, piggy.include_replicated = c->diag_pending_incl;
        c->diag_pending = false;
        c->diag_crossed_sweep = false;
    }
    const int blocks = (MS + W * W + el - 1) / el + (piggy.out ? MCL_PIGGY_BLOCKS : 0);
    ProfScope prof(c, MCL_PROF_REDUCE);
    c->variant[MCL_PROF_REDUCE] = "k_reduce_frag<" + std::to_string(el) + ">";
    if (el == 64)
        hipLaunchKernelGGL(k_reduce_frag<64>, dim3(blocks), dim3(1024), 0, c->stream, c->Mpart, c->GRpart, c->n_grpart,
                           (int)c->K, c->r, c->NB, MS, c->GR, piggy);
    else if (el == 32)
        hipLaunchKernelGGL(k_reduce_frag<32>, dim3(blocks), dim3(1024), 0, c->stream, c->Mpart, c->GRpart, c->n_grpart,
                           (int)c->K, c->r, c->NB, MS, c->GR, piggy);
    else
        hipLaunchKernelGGL(k_reduce_frag<16>, dim3(blocks), dim3(1024), 0, c->stream, c->Mpart, c->GRpart, c->n_grpart,
                           (int)c->K, c->r, c->NB, MS, c->GR, piggy);
    MCL_CHECK_HIP(c, hipGetLastError());
    return 0;
}

int mcl_launch_A_rhs_from_M(mcl_context *c) {
    ProfScope prof(c, MCL_PROF_OTHER);
    hipLaunchKernelGGL(k_A_rhs_from_M, dim3(c->n_parts), dim3(256), 0, c->stream, c->Mpart, c->CfragS, sweep_MS(c),
                       c->NB, c->r, c->seg_rhs);
    MCL_CHECK_HIP(c, hipGetLastError());
    return 0;
}
