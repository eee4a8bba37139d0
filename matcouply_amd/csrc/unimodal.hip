// Unimodality prox (penalties.py:983-1015 -> _unimodal_regression.py:27-141): best unimodal (optionally non-negative) L2 fit
// of every column of every slab, one lane per (slab, column), decisions in fp64.  Used for all three modes (A and C are a
// single slab).  Kernel forms (k_slab_unimodal_v4<MODE>):
//   3  throughput form (default when the columns fill the device): the two prefix-isotonic sweeps are PRUNED by a bound -
//      see "pruned sweeps" below;
//   0  the same without pruning (both sweeps over the whole column; MCL_UNI_NOPRUNE=1: the A/B baseline);
//   1 + 2  latency form for few columns: the two sweeps run concurrently in different waves, a second launch searches the
//      split and emits.
#include <algorithm>
#include <cstdlib>

#include "mcl_internal.h"
#include "rows_mfma.h"

// ---------------------------------------------------------------------------------------------------------
// Unimodality: one thread per (slab, column); prefix isotonic regression in both directions in fp64
// (the projection is discontinuous in its split index, so the arithmetic is kept in double).
// Scratch arrays are column-interleaved (index * r + col) so the r threads of a slab access them coalesced.
// ---------------------------------------------------------------------------------------------------------
struct UniScratch {
    double *lvL, *lvR, *eL, *eR;
    float *sink;  // two floats per lane: target of the emit loops' predicated-off stores
#ifdef MCL_UNI_STAMPS  // instrumented builds (tools/uni_stamps.py): per column group [start, end (100 MHz), HW_ID, XCC_ID, sweeps done]
    long long *stamps;
#endif
    int coop;     // cooperative ring refill (ur4_refill_coop); 0: MCL_NO_UNI_COOP (A/B switch)
    double *spL, *spR;  // packed spill areas (3 doubles per entry) of the left-to-right / right-to-left sweep
// ring entries per lane / entries per refill / elements per load batch (build-time; the defaults are the measured best)
#ifndef MCL_UNI_RC
#define MCL_UNI_RC 16
#endif
#ifndef MCL_UNI_NRF
#define MCL_UNI_NRF 8
#endif
#ifndef MCL_UNI_UB
#define MCL_UNI_UB 8
#endif
#ifdef MCL_UNI_DBG  // timing experiments only (wrong results): 1 no record stores, 2 no error stores, 4 no spill stores, 8 no emit,
                    // 32 no pooling loop, 64 no push (ring / spill), 128 no cooperative refill
    int dbg;
    unsigned long long *ctr;  // event counters summed over waves (mcl_uni_dbg_counters): see UNI_CNT
#endif
};
#ifdef MCL_UNI_DBG
#define UNI_DBG(bit) (sc.dbg & (bit))
// wave-level event counts: 0 element steps, 1 pooling trips, 2 steps with a spill, 3 cooperative refills served from prefetched
// registers, 4 cooperative refills with a blocking load, 5 dry refills from prefetched registers, 6 dry refills with a blocking load,
// 7 emit positions
#define UNI_CNT(i) (st.c[i] += 1)
#define UNI_CNT_IF(i, cond) (st.c[i] += __builtin_amdgcn_ballot_w64(cond) != 0 ? 1 : 0)
#else
#define UNI_DBG(bit) false
#define UNI_CNT(i) ((void)0)
#define UNI_CNT_IF(i, cond) ((void)0)
#endif

// ---------------------------------------------------------------------------------------------------------
// Unimodality, fourth version (default).  Same projection and the same decision rules as v3 / the reference
// (_unimodal_regression.py:27-104), organised so that a column costs TWO pooling sweeps instead of three and a
// pooling (merge) iteration - the divergent inner loop that bounds the kernel - carries no division:
//   * a stack entry is (sum, count, Q) with Q = sum of q over the stack up to and including the entry, so the prefix
//     error after a merge is read off the newly exposed top (v3 recomputed q = sum^2 / count of every popped block);
//     one reciprocal of the (integer) count per element gives both the level and q of the block being built;
//   * each sweep records, per position, (level, length) of the block that ENDS there at that time.  The stack at time
//     t is the chain  block ending at t-1 -> block ending at (its start - 1) -> ...: a block below the top is never
//     touched again, so its record from the time it was the top is still its state.  The two fits are therefore
//     emitted from the records by walking the chain from the split outwards (no third pooling sweep);
//   * the emit loops walk POSITIONS in lockstep over the wave (loads and stores of the r lanes of a slab coalesce),
//     records are fetched in unconditional 8-element batches;
//   * ring entries are 20 bytes and the ring holds MCL_UNI_RC = 16 of them in EVERY form: 20 KB of LDS per wave, two waves
//     per SIMD (measured against 8 entries at three waves per SIMD and against deeper / shallower load batches: slower).
// The records of the right-to-left sweep are stored at the position they belong to, so both emit loops index rows.
// ---------------------------------------------------------------------------------------------------------
template <class T>
static __device__ __forceinline__ unsigned lds_addr(T *p) {  // byte offset of a __shared__ object in LDS
    return (unsigned)(unsigned long)(__attribute__((address_space(3))) T *)p;
}
struct UniRing4 {
    double *sy, *q;  // LDS [RC][64]
    int *cw;         // LDS [RC][64]
    int h, cnt;      // ring index of its top entry, number of entries in the ring
    int mem_n;       // entries spilled to global memory
#ifdef MCL_UNI_DBG
    int dbg;
    unsigned c[8];
    long long cyc[6], tprev;  // (dbg & 256): cycles per section of a step: coop refill, push, pooling, post; between steps; total
#endif
};
struct UniRec {
    float lev;
    int len;
};
// the per-position arrays of the column regressions (prefix errors, block records) are written once and read once, a column
// batch apart - at config-5 scale gigabytes each: streamed past the caches (non-temporal)
static __device__ __forceinline__ void st_rec_nt(UniRec *p, UniRec rc) {
    long long bits;
    __builtin_memcpy(&bits, &rc, 8);
    __builtin_nontemporal_store(bits, reinterpret_cast<long long *>(p));
}
static __device__ __forceinline__ UniRec ld_rec_nt(const UniRec *p) {
    const long long bits = __builtin_nontemporal_load(reinterpret_cast<const long long *>(p));
    UniRec rc;
    __builtin_memcpy(&rc, &bits, 8);
    return rc;
}

// Spill area of a lane: packed 24-byte entries (sum, Q: fp64; count: int32 + pad), contiguous by depth - `sp` points at the
// lane's entry 0.  Every address below is the lane's base plus a small offset (one 64-bit add per access, no multiplies).
static __device__ __forceinline__ void sp_store(double *sp, int k, double sy, double q, int cw) {
    double *e = sp + 3 * (long)k;
    e[0] = sy, e[1] = q;
    *reinterpret_cast<int *>(e + 2) = cw;
}

// The top NRF entries of the spill area, prefetched into registers (see ur4_refill_coop): n of them are valid copies of the
// spill indices [mem_n - n, mem_n).
template <int NRF>
struct UniPrefetch {
    double sy[NRF], q[NRF];
    int cw[NRF];
    int n;
};

// Push; a full ring spills its bottom entry to global memory first.  (Tried: cooperative spills of 2 / 4 / 8 entries at a time, so
// that the lanes of a rising flank fall into step - faster on synthetic columns, slower on the iterates of a converged run,
// whose stacks are hundreds of entries deep: emptier rings mean more refills, and there the kernel is bound by those bytes.
// Round 4: spilling the bottom HALF of a full ring at once with per-lane refills at <= 3 entries (hysteresis instead of the
// cooperative top-up) moves the same bytes - config 5 at outer iteration 30: 40.7 GB fetched + 26.6 GB written per call against
// 42.1 + 26.6 - and is 12 % slower: the ~1 spilled entry per element is not churn of the policy, the stack of a noisy flank
// really buries most entries more than a ring deep before a collapse exposes them again.)
template <int RC, int NRF>
static __device__ __forceinline__ void ur4_push(UniRing4 &st, UniPrefetch<NRF> &pf, int lane, bool act, double sy, int cw, double q,
                                                double *__restrict__ sp) {
    UNI_CNT_IF(2, act && st.cnt == RC);
    if (act) {
        if (st.cnt == RC) {
            const int b = ((st.h - RC + 1) & (RC - 1)) * 64 + lane;
#ifdef MCL_UNI_DBG
            if (!(st.dbg & 4))
#endif
                sp_store(sp, st.mem_n, st.sy[b], st.q[b], st.cw[b]);
            st.mem_n += 1;
            st.cnt = RC - 1;
            pf.n = 0;  // the prefetched entries are no longer the top of the spill area
        }
        st.h = (st.h + 1) & (RC - 1);
        const int t = st.h * 64 + lane;
        st.sy[t] = sy;
        st.q[t] = q;
        st.cw[t] = cw;
        st.cnt += 1;
    }
}

// prefetched entries -> ring, below its bottom entry (needs room for pf.n entries)
template <int RC, int NRF>
static __device__ __forceinline__ void ur4_take_prefetched(UniRing4 &st, UniPrefetch<NRF> &pf, int lane) {
#pragma unroll
    for (int i = 0; i < NRF; ++i) {
        if (i < pf.n) {
            const int t = ((st.h - st.cnt - i) & (RC - 1)) * 64 + lane;
            st.sy[t] = pf.sy[i], st.q[t] = pf.q[i], st.cw[t] = pf.cw[i];
        }
    }
    st.mem_n -= pf.n;
    st.cnt += pf.n;
    pf.n = 0;
}

// request the next entries of the spill area (top first); nothing waits for them here.  Unconditional loads at clamped depths
// (entries past the bottom read slot 0 and are never used: pf.n counts the valid ones)
template <int NRF>
static __device__ __forceinline__ void ur4_prefetch(const UniRing4 &st, UniPrefetch<NRF> &pf, const double *__restrict__ sp) {
#pragma unroll
    for (int i = 0; i < NRF; ++i) {
        const double *e = sp + 3 * (long)max(st.mem_n - 1 - i, 0);
        pf.sy[i] = e[0], pf.q[i] = e[1], pf.cw[i] = *reinterpret_cast<const int *>(e + 2);
    }
    pf.n = st.mem_n < NRF ? st.mem_n : NRF;
}

// a lane that has to pop finds its ring empty but has spilled entries (rare: the cooperative refill below keeps the rings
// topped up): prefetched entries if it has them, else a blocking refill
template <int RC, int NRF>
static __device__ __forceinline__ void ur4_refill_dry(UniRing4 &st, UniPrefetch<NRF> &pf, int lane, const double *__restrict__ sp) {
    if (pf.n > 0) {
        ur4_take_prefetched<RC, NRF>(st, pf, lane);
    } else {
        const int nref = st.mem_n >= RC / 2 ? RC / 2 : st.mem_n;  // independent loads, one latency
        for (int i = 0; i < nref; ++i) {
            const double *e = sp + 3 * (long)(st.mem_n - 1 - i);
            const int t = ((st.h - i) & (RC - 1)) * 64 + lane;
            st.sy[t] = e[0];
            st.q[t] = e[1];
            st.cw[t] = *reinterpret_cast<const int *>(e + 2);
        }
        st.mem_n -= nref;
        st.cnt = nref;
    }
}

// Cooperative, prefetched refill (round 3).  On smooth, nearly unimodal columns - what the iterates of a converging run
// look like - the stack is as deep as the rising flank is long (one block per element; config 5 at outer iteration 25:
// median depth 157, maximum 870, tools/uni_depth.py), the ring spills hundreds of entries, and the falling flank pops
// them back one per step: every lane then ran dry every RC / 2 steps at its own phase, so nearly EVERY step of the wave
// waited a memory round trip for one lane or another (the regressions of config 5 took 8-11 ms per call on the noise-like
// iterates of the first outer iterations and 21-26 ms from iteration ~20 on, same kernel; 1.7 -> 4.5 ms on the 1/8 shard).
// Now, once per element step: if ANY lane is about to run dry, every lane with spilled entries and room tops its ring
// up - from the NRF entries it PREFETCHED into registers at its previous refill (no wait: they were requested several
// steps ago), or with a blocking load when a spill has invalidated them - and requests the next NRF.  The lanes fall
// into step, and on a falling flank no refill waits for memory at all.
template <int RC, int NRF>
static __device__ __forceinline__ void ur4_refill_coop(UniRing4 &st, UniPrefetch<NRF> &pf, int lane, const double *__restrict__ sp) {
    if (__builtin_amdgcn_ballot_w64(st.cnt <= 1 && st.mem_n > 0) == 0) return;  // wave-uniform
    // a lane joins when its own ring is at most half full: it will run dry soon (joining whenever there was room made every
    // lane prefetch four times as often as it refilled, and most of those prefetches were invalidated by the next spill)
    const int room = st.cnt <= RC / 2 ? RC - 2 - st.cnt : 0;
    UNI_CNT_IF(3, pf.n > 0 && room >= pf.n);
    UNI_CNT_IF(4, pf.n <= 0 && st.mem_n > 0 && room > 0);
    if (pf.n > 0) {
        if (room >= pf.n) {
            ur4_take_prefetched<RC, NRF>(st, pf, lane);
            ur4_prefetch<NRF>(st, pf, sp);
        }
    } else if (st.mem_n > 0 && room > 0) {
        const int want = room < NRF ? room : NRF;
        const int nref = st.mem_n < want ? st.mem_n : want;
        double vsy[NRF], vq[NRF];
        int vcw[NRF];
#pragma unroll
        for (int i = 0; i < NRF; ++i) {  // independent clamped loads: one latency
            const double *e = sp + 3 * (long)max(st.mem_n - 1 - i, 0);
            vsy[i] = e[0], vq[i] = e[1], vcw[i] = *reinterpret_cast<const int *>(e + 2);
        }
#pragma unroll
        for (int i = 0; i < NRF; ++i) {
            if (i < nref) {
                const int t = ((st.h - st.cnt - i) & (RC - 1)) * 64 + lane;  // below the ring's bottom entry
                st.sy[t] = vsy[i], st.q[t] = vq[i], st.cw[t] = vcw[i];
            }
        }
        st.mem_n -= nref;
        st.cnt += nref;
        ur4_prefetch<NRF>(st, pf, sp);
    }
}

// 1 / w for a positive integer-valued w: hardware estimate + two Newton steps (<= 1 ulp)
static __device__ __forceinline__ double rcp_count(double w) {
    double x = __builtin_amdgcn_rcp(w);
    x = __builtin_fma(__builtin_fma(-w, x, 1.0), x, x);
    x = __builtin_fma(__builtin_fma(-w, x, 1.0), x, x);
    return x;
}

// Offset (in elements) of row `idx` of a lane's column inside its slab: idx * r with both factors below 2^24 (the launcher checks
// the rows) - ONE full-rate instruction, and as an unsigned 32-bit index the address is one v_lshl_add_u64 away; the 64-bit
// products `(long)idx * rs` of rounds 3-5 cost 6-8 instructions per load (24 loads per batch of 8 steps in the sweeps).
static __device__ __forceinline__ unsigned row_off(int idx, int r) { return __umul24((unsigned)idx, (unsigned)r); }

static __device__ __forceinline__ int wave_max_i(int v) {
    for (int o = 32; o; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}
static __device__ __forceinline__ int wave_min_i(int v) {
    for (int o = 32; o; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}

// MODE 0: everything in one lane per column (throughput: fewest bytes, used when the columns fill the device).
// MODE 1 + MODE 2: few columns (less than about one wave per SIMD) - the two sweeps run concurrently in different waves
// (direction = block parity; the right-to-left one stores its errors instead of searching the split), then a second
// launch searches the split with the same comparisons in the same order and emits.  Halves the serial chain.
// WPB (MODE 3): independent waves per workgroup.  With one-wave workgroups the SIMD of every wave is the dispatcher's choice, and
// eight of them per CU (the LDS rings allow two per SIMD) do not land two per SIMD (round 6: the PARAFAC2 Newton-Schulz kernel
// had 14 % of its waves doubled up on a SIMD with another SIMD idle, HW_ID stamps); the waves of ONE workgroup are placed on the
// four SIMDs in turn.  WPB = 4: four column groups per workgroup, rings in dynamic LDS (80 KB), two workgroups per CU.
// (Round 6, measured and dropped: a PERSISTENT form - as many workgroups as the device holds, every wave drawing column groups from a
// counter, so that four waves do not wait for the slowest of them before the next four start - is no faster: 95.2 / 97.7 ms per
// config-5 iteration against 94.2 / 96.9, same box; two waves per workgroup, which the dispatcher puts on SIMD pairs, 104-105.)
template <int MODE, int WPB = 1>
__global__ __launch_bounds__(MODE == 2 ? 256 : 64 * WPB) __attribute__((amdgpu_waves_per_eu(2))) void k_slab_unimodal_v4(const int *__restrict__ ext, int n_slabs, float *__restrict__ F,
                                                         RegSet regs, int k, int r, UniScratch sc) {
    MCL_GATE(regs.gate);
    static_assert(WPB == 1 || MODE == 3, "several independent waves per workgroup: the throughput form only");
    // ring entries per lane and entries per refill (see the defaults above; round 4 re-measured RC = 8 / NRF = 4 / UB = 4 at
    // three waves per SIMD: 14.5 ms against 10.2 ms per call at config 5 - the depth of the load batches matters more)
    constexpr int RC = MCL_UNI_RC, NRF = MCL_UNI_NRF;
    __shared__ double ring_d_st[2][(MODE == 2 || WPB > 1) ? 1 : RC * 64];
    __shared__ int ring_i_st[(MODE == 2 || WPB > 1) ? 1 : RC * 64];
    extern __shared__ double ring_dyn[];  // WPB > 1: per wave [2][RC * 64] doubles + [RC * 64] ints
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;  // MODE 2: four waves work on the same 64 columns
    double (*ring_d)[(MODE == 2 || WPB > 1) ? 1 : RC * 64] = ring_d_st;
    int *ring_i = ring_i_st;
    double *ring_sy = WPB > 1 ? ring_dyn + (long)wv * (2 * RC * 64 + RC * 32) : &ring_d[0][0];
    double *ring_q = WPB > 1 ? ring_sy + RC * 64 : &ring_d[1][0];
    int *ring_cw = WPB > 1 ? reinterpret_cast<int *>(ring_sy + 2 * RC * 64) : ring_i;
    const unsigned wblk = WPB > 1 ? blockIdx.x * WPB + wv : blockIdx.x;  // this wave's column group
#ifdef MCL_UNI_STAMPS
    const long long stamp_t0 = (long long)__builtin_amdgcn_s_memrealtime();
    long long stamp_t1 = 0;
#endif
    const bool do_L = MODE == 0 || (MODE == 1 && (blockIdx.x & 1) == 0);  // (MODE 3 has its own schedule below)
    const bool do_R = MODE == 0 || (MODE == 1 && (blockIdx.x & 1) == 1);
    const long t = (long)(MODE == 1 ? blockIdx.x >> 1 : wblk) * 64 + lane;
    const bool live = t < (long)n_slabs * r;  // no early exit: the emit loops use wave-wide reductions
    const int slab = live ? (int)(t / r) : 0, col = live ? (int)(t - (long)slab * r) : 0;
    const int s = ext[slab], e = ext[slab + 1];
    const int n = live ? max(e - s, 0) : 0;
    const int nonneg = regs.nonneg[k];
    float *__restrict__ Z = regs.aux[k];
    const float *__restrict__ U = regs.dual[k];
    double *__restrict__ errL = sc.eL, *__restrict__ errR = sc.eR;
    // spill areas of the block stack: the concurrent right-to-left sweep of MODE 1 has its own
    // (MODE 3 switches between the two areas: its left-to-right sweep pauses while the right-to-left one runs)
    double *sp;  // this lane's spill area of the sweep that is running (see "Spill areas" below)
    auto spill_area = [&](bool right) { sp = (right ? sc.spR : sc.spL) + 3 * ((long)s * r + (long)col * n); };
    spill_area(MODE == 1 && do_R);
    UniRec *__restrict__ recL = reinterpret_cast<UniRec *>(sc.lvL), *__restrict__ recR = reinterpret_cast<UniRec *>(sc.lvR);
    const long rs = r;
    const long eb = (long)s + slab;  // n + 1 error entries per slab
    // Spill areas.  The per-position arrays (errors, records) are column-interleaved because the lanes of a slab walk the
    // POSITIONS in lockstep; the depth of the spilled stack is the lane's own, so its spill area is lane-private and contiguous
    // (round 4): packed 24-byte entries (sum, Q, count), entry k of (slab, column) at 3 (s r + col n + k) doubles of the sweep's
    // array.  Consecutive spills of a lane fill the same 64-byte sectors - the write-back L2 sends them out as whole lines -
    // and a refill of 8 entries reads 192 contiguous bytes.  (Same box, config 5 at outer iteration 30, pruned sweeps: 13.2 ms
    // with round 3's column-interleaved areas, 11.7 ms lane-private with one array per field, 10.2 ms packed.)

    UniRing4 st;
    UniPrefetch<NRF> pf;
    st.sy = ring_sy, st.q = ring_q, st.cw = ring_cw;
#ifdef MCL_UNI_DBG
    st.dbg = sc.dbg;
    for (int i = 0; i < 8; ++i) st.c[i] = 0;
    for (int i = 0; i < 6; ++i) st.cyc[i] = 0;
    st.tprev = 0;
    const long long t_kernel0 = (long long)__builtin_readcyclecounter();
    struct CtrOut {  // lane 0 adds the wave's counts when the kernel returns (every exit path)
        const UniRing4 &st; unsigned long long *ctr; int lane; long long t0;
        __device__ ~CtrOut() {
            if (lane == 0 && ctr) {
                for (int i = 0; i < 8; ++i) atomicAdd(ctr + i, (unsigned long long)st.c[i]);
                for (int i = 0; i < 5; ++i) atomicAdd(ctr + 8 + i, (unsigned long long)st.cyc[i]);
                atomicAdd(ctr + 14, (unsigned long long)st.cyc[5]);
                atomicAdd(ctr + 13, (unsigned long long)((long long)__builtin_readcyclecounter() - t0));
            }
        }
    } ctr_out{st, sc.ctr, lane, t_kernel0};
#endif
    // byte addresses of this lane's slot 0 in the rings (the pooling loop addresses LDS itself)
    const int lds_d = (int)lds_addr(ring_sy) + lane * 8, lds_i = (int)lds_addr(ring_cw) + lane * 4;
    static_assert((RC & (RC - 1)) == 0, "ring indices wrap by masking");
    double csy, ccw, curQ, cum2;  // block being built (count kept as a double), Q including it, sum of y^2
    double tsy, tcw, tQ;          // cached top of the stack below it
    int ht;  // 1: there is a cached top
    float levf;
    // (Round 5, measured and dropped: a sentinel entry (sum -inf, count 1, Q 0) at the bottom of every stack takes `ht` out of
    // the pooling loop - 23 instead of 27 instructions per trip, ~8 fewer per element step, bit-identical fits - and is 1.5 %
    // SLOWER on the steady-state iterates of config 5 (9.85 against 9.70 ms, same box): the kernel is not bound by the count of
    // its cheap vector instructions, see profiles/r5_uni_anatomy.txt.)
    auto reset = [&]() {
        st.h = 0, st.cnt = 0, st.mem_n = 0, pf.n = 0;
        cum2 = 0.0, csy = 0.0, ccw = 1.0, curQ = 0.0;
        tsy = 0.0, tcw = 1.0, tQ = 0.0;
        ht = 0, levf = 0.f;
    };
    // one element: returns the prefix error; leaves (levf, ccw) = record of the block ending at this element
    auto step = [&](double v, bool first) -> double {
        UNI_CNT(0);
#ifdef MCL_UNI_DBG
        long long tk0 = 0;
        if (sc.dbg & 256) {
            tk0 = (long long)__builtin_readcyclecounter();
            if (st.tprev) st.cyc[4] += tk0 - st.tprev;
        }
        auto tick = [&](int sec) {
            if (sc.dbg & 256) {
                const long long t1 = (long long)__builtin_readcyclecounter();
                st.cyc[sec] += t1 - tk0;
                tk0 = t1;
            }
        };
#else
        auto tick = [](int) {};
#endif
        if (sc.coop && !UNI_DBG(128)) ur4_refill_coop<RC, NRF>(st, pf, lane, sp);
        tick(0);
        cum2 += v * v;
        // the finished block becomes the cached top; the previous top moves into the ring
        if (!UNI_DBG(64)) ur4_push<RC, NRF>(st, pf, lane, !first && ht != 0, tsy, (int)tcw, tQ, sp);
        tick(1);
        if (!first) {
            tsy = csy, tcw = ccw, tQ = curQ;
            ht = 1;
        }
        csy = v;
        ccw = 1.0;
        // pooling: while mean(cur) <= mean(top).  The loop is the divergent core of the kernel and the kernel is bound by
        // instruction issue, so the loop is written out (the compiler's version of the same source carries 70-85 instructions
        // per trip, most of them copies and mask algebra; this one 27): its exits - no lane pools / a pooling lane has run dry -
        // are decided wave-wide, what remains under the lane predicate is the straight-line pop.  With an empty ring the pop
        // reads a slot whose contents are not used (ht = 0) and moves h, which is as good a position as any other.
#ifdef MCL_UNI_DBG
        unsigned trips = 0;
#endif
        while (!UNI_DBG(32)) {
            int dry_exit;
            double t0, t1;
            unsigned long long m_need, m_b, m_save;
            int a1, a2, icw;
            asm volatile(
                "s_mov_b32 %[flag], 0\n"
                "L_pool_%=:\n"
                "v_mul_f64 %[t0], %[csy], %[tcw]\n"
                "v_mul_f64 %[t1], %[tsy], %[ccw]\n"
                "v_cmp_ne_u32_e32 vcc, 0, %[ht]\n"
                "v_cmp_le_f64_e64 %[mn], %[t0], %[t1]\n"
                "s_and_b64 %[mn], %[mn], vcc\n"
                "s_cbranch_scc0 L_done_%=\n"
                "v_cmp_eq_u32_e32 vcc, 0, %[cnt]\n"
                "v_cmp_lt_i32_e64 %[mb], 0, %[memn]\n"
                "s_and_b64 vcc, vcc, %[mb]\n"
                "s_and_b64 vcc, vcc, %[mn]\n"
                "s_cbranch_scc1 L_dry_%=\n"
                "s_and_saveexec_b64 %[ms], %[mn]\n"
#ifdef MCL_UNI_DBG
                "s_add_u32 %[trips], %[trips], 1\n"
#endif
                "v_add_f64 %[csy], %[csy], %[tsy]\n"
                "v_add_f64 %[ccw], %[ccw], %[tcw]\n"
                "v_lshl_add_u32 %[a1], %[h], 9, %[ldsd]\n"
                "v_lshl_add_u32 %[a2], %[h], 8, %[ldsi]\n"
                "ds_read_b64 %[tsy], %[a1]\n"
                "ds_read_b64 %[tq], %[a1] offset:%[qoff]\n"
                "ds_read_b32 %[icw], %[a2]\n"
                "v_cmp_lt_i32_e32 vcc, 0, %[cnt]\n"
                "v_add_u32_e32 %[h], -1, %[h]\n"      // two VALU instructions between the compare and the select that reads
                "v_and_b32_e32 %[h], %[msk], %[h]\n"  // its VCC: gfx940+ needs two wait states there (no interlock)
                "v_cndmask_b32_e64 %[ht], 0, 1, vcc\n"
                "v_max_i32_e32 %[cnt], 1, %[cnt]\n"
                "v_add_u32_e32 %[cnt], -1, %[cnt]\n"
                "s_waitcnt lgkmcnt(0)\n"
                "v_cvt_f64_i32_e32 %[tcw], %[icw]\n"
                "s_mov_b64 exec, %[ms]\n"
                "s_branch L_pool_%=\n"
                "L_dry_%=:\n"
                "s_mov_b32 %[flag], 1\n"
                "L_done_%=:\n"
                : [flag] "=&s"(dry_exit), [t0] "=&v"(t0), [t1] "=&v"(t1), [mn] "=&s"(m_need), [mb] "=&s"(m_b), [ms] "=&s"(m_save),
                  [a1] "=&v"(a1), [a2] "=&v"(a2), [icw] "=&v"(icw), [csy] "+v"(csy), [ccw] "+v"(ccw), [tsy] "+v"(tsy),
                  [tcw] "+v"(tcw), [tq] "+v"(tQ), [ht] "+v"(ht), [h] "+v"(st.h), [cnt] "+v"(st.cnt)
#ifdef MCL_UNI_DBG
                  , [trips] "+s"(trips)
#endif
                : [memn] "v"(st.mem_n), [ldsd] "v"(lds_d), [ldsi] "v"(lds_i), [qoff] "n"(RC * 64 * 8), [msk] "n"(RC - 1)
                : "vcc", "scc", "memory");
            if (dry_exit == 0) break;  // wave-uniform
            const bool dry = ht != 0 && csy * tcw <= tsy * ccw && st.cnt == 0 && st.mem_n > 0;
            UNI_CNT_IF(5, dry && pf.n > 0);
            UNI_CNT_IF(6, dry && pf.n <= 0);
            if (dry) ur4_refill_dry<RC, NRF>(st, pf, lane, sp);
        }
#ifdef MCL_UNI_DBG
        st.c[1] += trips;
#endif
        tick(2);
        // a block with a negative mean is clamped to level 0 and contributes q = 0; every block below it has a smaller
        // mean, so their Q is exactly 0 too and the prefix error comes out as cum2 without a special case
        const double lev = csy * rcp_count(ccw);
        const double levc = nonneg ? fmax(lev, 0.0) : lev;
        curQ = (ht != 0 ? tQ : 0.0) + csy * levc;
        levf = (float)levc;
        const double err_out = cum2 - curQ;
        tick(3);
#ifdef MCL_UNI_DBG
        if (sc.dbg & 256) st.tprev = tk0;
#endif
        return err_out;
    };

    constexpr int UB = MCL_UNI_UB;  // elements per load batch; the NEXT batch is in flight while the current one is pooled
    int n_split_out = 0;   // MODE 3: the split its schedule found
    // ---- pruned sweeps (MODE 3) ------------------------------------------------------------------------------------
    // total(t) = eL[t] + eR[t] (error of the best increasing fit of [0, t) + of the best decreasing fit of [t, n)) is
    // wanted at its minimum only, eL is non-decreasing in t and eR non-increasing, both are >= 0: once a total `best` is
    // known, every t with eL[t] > best lies to the right of all minima and every t with eR[t] > best to their left.  So:
    //   A  the left-to-right sweep runs up to m = n / 2, then pauses (its ring goes to its spill area, the few scalars stay
    //      in registers);
    //   B  the right-to-left sweep runs from the end; from t = m downwards it knows both errors, keeps the best total
    //      and STOPS as soon as eR[t] > best + delta;
    //   C  the left-to-right sweep resumes at m against the errors B stored and stops as soon as eL[t] > best + delta.
    // delta = 1e-9 x (sum of squares of the column) is far above the rounding noise of the errors (1e-16 relative to that
    // sum: they are differences of such sums) and far below anything that moves a stopping point: the totals skipped are
    // strictly above the minimum, the split is the one the exhaustive search of MODE 0 finds (smallest t among the
    // minima: B scans downwards with <=, C upwards with <).  On the smooth, nearly unimodal columns of a converged run a
    // lane does ~1.25 n element steps instead of 2 n and errors are read only where the two sweeps overlap.  The meeting
    // point is the SAME for all columns of a slab on purpose: the lanes of a wave walk the rows in lockstep and the 32 columns
    // of a slab share every cache line of a row, so time and bytes follow the number of rows the WAVE visits - n + (largest
    // stop of C - smallest stop of B over its lanes), whatever m is.  (Measured, config 5 at outer iteration 30: letting
    // every column meet at the split its previous call found cuts the lane steps to 1.1 n, but the phases of the lanes then
    // end at different rows: 17.0 ms and 64.6 GB per call against 13.0 ms and 52.0 GB with m = n / 2; 14.7 ms, 69.2 GB
    // without pruning.)
    if constexpr (MODE == 3) {
        int split3 = n;
        if (n > 0) {
            const int m = n >> 1;  // the same for all columns of a slab: see above
            double eLm = 0.0;  // eL[m]
            // A: elements [0, m)
            reset();
            errL[eb * rs + col] = 0.0;
            if (m > 0) {
                const float *fp = F + (long)s * rs + col, *up = U + (long)s * rs + col;
                double *ep = errL + eb * rs + col;
                UniRec *rp = recL + (long)s * rs + col;
                float fb[UB], ub[UB];
#pragma unroll
                for (int j = 0; j < UB; ++j) {
                    const unsigned o = row_off(min(j, m - 1), r);
                    fb[j] = fp[o], ub[j] = up[o];
                }
                for (int i0 = 0; i0 < m; i0 += UB) {
                    double vb[UB];
#pragma unroll
                    for (int j = 0; j < UB; ++j) vb[j] = (double)(fb[j] + ub[j]);
#pragma unroll
                    for (int j = 0; j < UB; ++j) {
                        const unsigned o = row_off(min(i0 + UB + j, m - 1), r);
                        fb[j] = fp[o], ub[j] = up[o];
                    }
#pragma unroll
                    for (int j = 0; j < UB; ++j) {
                        if (i0 + j < m) {
                            eLm = step(vb[j], i0 + j == 0);
                            ep += rs;
                            if (!UNI_DBG(2)) __builtin_nontemporal_store(eLm, ep);
                            UniRec rc;
                            rc.lev = levf, rc.len = (int)ccw;
                            if (!UNI_DBG(1)) st_rec_nt(rp, rc);
                            rp += rs;
                        }
                    }
                }
            }
            // pause: ring -> spill area (bottom entry first), scalars aside
#pragma unroll
            for (int kk = 0; kk < RC; ++kk) {
                if (kk < st.cnt) {
                    const int b = ((st.h - st.cnt + 1 + kk) & (RC - 1)) * 64 + lane;
                    sp_store(sp, st.mem_n + kk, st.sy[b], st.q[b], st.cw[b]);
                }
            }
            const int L_mem = st.mem_n + st.cnt, L_ht = ht;
            const double L_csy = csy, L_ccw = ccw, L_curQ = curQ, L_cum2 = cum2, L_tsy = tsy, L_tcw = tcw, L_tQ = tQ;
            // B: right to left
            spill_area(true);
            reset();
            double best = (m == n) ? eLm : __builtin_inf();
            split3 = n;
            {
                const float *fp = F + (long)s * rs + col, *up = U + (long)s * rs + col;  // indexed by POSITION n - 1 - i
                const double *eLp = errL + eb * rs + col;  // eL[t] at eLp[t * rs]
                double *eRp = errR + eb * rs + col;        // eR[t] at eRp[t * rs], stored for t > m only
                UniRec *rp = recR + ((long)e - 1) * rs + col;
                float fb[UB], ub[UB];
                double eb_n[UB];
#pragma unroll
                for (int j = 0; j < UB; ++j) {
                    const int pos = max(n - 1 - j, 0);
                    const unsigned o = row_off(pos, r);
                    fb[j] = fp[o], ub[j] = up[o];
                    eb_n[j] = __builtin_nontemporal_load(eLp + row_off(min(pos, m), r));
                }
                bool done = false;
                for (int i0 = 0; i0 < n && !done; i0 += UB) {
                    double vb[UB], eb_l[UB];
#ifdef MCL_UNI_DBG
                    long long tb0 = (sc.dbg & 512) ? (long long)__builtin_readcyclecounter() : 0;
#endif
#pragma unroll
                    for (int j = 0; j < UB; ++j) vb[j] = (double)(fb[j] + ub[j]), eb_l[j] = eb_n[j];
#ifdef MCL_UNI_DBG
                    if (sc.dbg & 512) {  // force the batch's loaded values here: the wait for the loads is then this section's
#pragma unroll
                        for (int j = 0; j < UB; ++j) asm volatile("" : "+v"(vb[j]), "+v"(eb_l[j]));
                        const long long t1 = (long long)__builtin_readcyclecounter();
                        st.cyc[5] += t1 - tb0;
                        tb0 = t1;
                    }
#endif
#pragma unroll
                    for (int j = 0; j < UB; ++j) {
                        const int pos = max(n - 1 - (i0 + UB + j), 0);
                        const unsigned o = row_off(pos, r);
                        fb[j] = fp[o], ub[j] = up[o];
                        eb_n[j] = __builtin_nontemporal_load(eLp + row_off(min(pos, m), r));  // (not read when t > m)
                    }
#ifdef MCL_UNI_DBG
                    if (sc.dbg & 512) {
                        asm volatile("" ::: "memory");
                        st.cyc[3] += (long long)__builtin_readcyclecounter() - tb0;  // (section 3 re-used: issue of the batch's loads)
                        st.c[7] += 1;  // batches of phase B
                    }
#endif
#pragma unroll
                    for (int j = 0; j < UB; ++j) {
                        const int i = i0 + j;
                        if (i < n && !done) {
                            const double er = step(vb[j], i == 0);
                            UniRec rc;
                            rc.lev = levf, rc.len = (int)ccw;
                            if (!UNI_DBG(1)) st_rec_nt(rp, rc);
                            rp -= rs;
                            const int tt = n - 1 - i;
                            if (tt > m) {
                                if (!UNI_DBG(2)) __builtin_nontemporal_store(er, eRp + row_off(tt, r));
                            } else {
                                const double tot = eb_l[j] + er;
                                if (tot <= best) {
                                    best = tot;
                                    split3 = tt;
                                }
                                done = er > best + 1e-9 * (L_cum2 + cum2);
                            }
                        }
                    }
                }
            }
            // C: resume the left-to-right sweep at element m
            if (m < n) {
                const double R_cum2 = cum2;
                spill_area(false);
                st.h = 0, st.cnt = 0, st.mem_n = L_mem, pf.n = 0;
                csy = L_csy, ccw = L_ccw, curQ = L_curQ, cum2 = L_cum2, tsy = L_tsy, tcw = L_tcw, tQ = L_tQ, ht = L_ht;
                const float *fp = F + ((long)s + m) * rs + col, *up = U + ((long)s + m) * rs + col;
                const double *eRp = errR + (eb + m + 1) * rs + col;  // eR[m + 1 + k] at eRp[k * rs]; eR[n] = 0 is not stored
                UniRec *rp = recL + ((long)s + m) * rs + col;
                const int nc = n - m;  // elements left
                float fb[UB], ub[UB];
                double er_n[UB];
#pragma unroll
                for (int j = 0; j < UB; ++j) {
                    const unsigned o = row_off(min(j, nc - 1), r);
                    fb[j] = fp[o], ub[j] = up[o];
                    er_n[j] = __builtin_nontemporal_load(eRp + row_off(min(j, max(nc - 2, 0)), r));
                }
                bool done = false;
                for (int i0 = 0; i0 < nc && !done; i0 += UB) {
                    double vb[UB], er_l[UB];
#pragma unroll
                    for (int j = 0; j < UB; ++j) vb[j] = (double)(fb[j] + ub[j]), er_l[j] = er_n[j];
#pragma unroll
                    for (int j = 0; j < UB; ++j) {
                        const unsigned o = row_off(min(i0 + UB + j, nc - 1), r);
                        fb[j] = fp[o], ub[j] = up[o];
                        er_n[j] = __builtin_nontemporal_load(eRp + row_off(min(i0 + UB + j, max(nc - 2, 0)), r));
                    }
#pragma unroll
                    for (int j = 0; j < UB; ++j) {
                        const int i = i0 + j;
                        if (i < nc && !done) {
                            const double el = step(vb[j], m + i == 0);
                            UniRec rc;
                            rc.lev = levf, rc.len = (int)ccw;
                            if (!UNI_DBG(1)) st_rec_nt(rp, rc);
                            rp += rs;
                            const double tot = el + (i == nc - 1 ? 0.0 : er_l[j]);  // t = m + i + 1
                            if (tot < best) {
                                best = tot;
                                split3 = m + i + 1;
                            }
                            done = el > best + 1e-9 * (cum2 + R_cum2);
                        }
                    }
                }
            }
        }
        n_split_out = split3;
    }
    // sweep 1: prefix errors and block records, left to right
    reset();
    if (do_L && n > 0) {
        const float *fp = F + (long)s * rs + col, *up = U + (long)s * rs + col;
        double *ep = errL + eb * rs + col;
        UniRec *rp = recL + (long)s * rs + col;
        *ep = 0.0;
        float fb[UB], ub[UB];
#pragma unroll
        for (int j = 0; j < UB; ++j) {
            const long o = (long)min(j, n - 1) * rs;
            fb[j] = fp[o], ub[j] = up[o];
        }
        for (int i0 = 0; i0 < n; i0 += UB) {
            double vb[UB];
#pragma unroll
            for (int j = 0; j < UB; ++j) vb[j] = (double)(fb[j] + ub[j]);
#pragma unroll
            for (int j = 0; j < UB; ++j) {  // unconditional clamped loads of the next batch
                const long o = (long)min(i0 + UB + j, n - 1) * rs;
                fb[j] = fp[o], ub[j] = up[o];
            }
#pragma unroll
            for (int j = 0; j < UB; ++j) {
                if (i0 + j < n) {
                    const double er = step(vb[j], i0 + j == 0);
                    ep += rs;
                    __builtin_nontemporal_store(er, ep);
                    UniRec rc;
                    rc.lev = levf, rc.len = (int)ccw;
                    if (!UNI_DBG(1)) st_rec_nt(rp, rc);
                    rp += rs;
                }
            }
        }
    }
    // sweep 2: suffix errors right to left + best split (smallest t among the minima); records stored by position
    reset();
    int split = MODE == 3 ? n_split_out : n;
    if (do_R && n > 0) {
        double best = (MODE == 0) ? errL[(eb + n) * rs + col] : 0.0;
        const float *fp = F + ((long)e - 1) * rs + col, *up = U + ((long)e - 1) * rs + col;
        const double *ep = errL + (eb + n - 1) * rs + col;
        double *erp = errR + (eb + 1) * rs + col;  // MODE 1: errR[i + 1] = error of the suffix of length i + 1
        UniRec *rp = recR + ((long)e - 1) * rs + col;
        float fb[UB], ub[UB];
        double eb_n[UB];
#pragma unroll
        for (int j = 0; j < UB; ++j) {
            const long o = (long)min(j, n - 1) * rs;
            fb[j] = fp[-o], ub[j] = up[-o];
            eb_n[j] = (MODE == 0) ? __builtin_nontemporal_load(ep - o) : 0.0;
        }
        for (int i0 = 0; i0 < n; i0 += UB) {
            double vb[UB], eb_l[UB];
#pragma unroll
            for (int j = 0; j < UB; ++j) vb[j] = (double)(fb[j] + ub[j]), eb_l[j] = eb_n[j];
#pragma unroll
            for (int j = 0; j < UB; ++j) {
                const long o = (long)min(i0 + UB + j, n - 1) * rs;
                fb[j] = fp[-o], ub[j] = up[-o];
                if (MODE == 0) eb_n[j] = __builtin_nontemporal_load(ep - o);
            }
#pragma unroll
            for (int j = 0; j < UB; ++j) {
                const int i = i0 + j;
                if (i < n) {
                    const double er = step(vb[j], i == 0);
                    UniRec rc;
                    rc.lev = levf, rc.len = (int)ccw;
                    if (!UNI_DBG(1)) st_rec_nt(rp, rc);
                    rp -= rs;
                    if (MODE == 0) {
                        const double tot = eb_l[j] + er;
                        if (tot <= best) {
                            best = tot;
                            split = n - 1 - i;
                        }
                    } else {
                        *erp = er;
                        erp += rs;
                    }
                }
            }
        }
    }
    if (MODE == 1) return;
    if (MODE == 2) {
        // the split search of sweep 2 from the stored errors: the same sums compared in the same order (minimum total,
        // ties -> largest i = smallest split).  Each of the four waves scans a quarter of the positions, the partial
        // results are combined in ascending order of i; then wave 0 emits the left fit and wave 1 the right one.
        __shared__ double sbest[4][64];
        __shared__ int sidx[4][64];
        double bw = 0.0;
        int iw = -1;
        if (n > 0) {
            const int q4 = (n + 3) >> 2;
            const int ia = wv * q4, ib = min(ia + q4, n);
            const double *epl = errL + (eb + n - 1) * rs + col;
            const double *epr = errR + (eb + 1) * rs + col;
            constexpr int SB = 16;
            double ln[SB], rn_[SB];
#pragma unroll
            for (int j = 0; j < SB; ++j) {
                const long o = (long)min(ia + j, n - 1) * rs;
                ln[j] = epl[-o], rn_[j] = epr[o];
            }
            for (int i0 = ia; i0 < ib; i0 += SB) {
                double lb[SB], rb[SB];
#pragma unroll
                for (int j = 0; j < SB; ++j) lb[j] = ln[j], rb[j] = rn_[j];
#pragma unroll
                for (int j = 0; j < SB; ++j) {
                    const long o = (long)min(i0 + SB + j, n - 1) * rs;
                    ln[j] = epl[-o], rn_[j] = epr[o];
                }
#pragma unroll
                for (int j = 0; j < SB; ++j) {
                    const int i = i0 + j;
                    if (i < ib) {
                        const double tot = lb[j] + rb[j];
                        if (iw < 0 || tot <= bw) {
                            bw = tot;
                            iw = i;
                        }
                    }
                }
            }
        }
        sbest[wv][lane] = bw;
        sidx[wv][lane] = iw;
        __syncthreads();
        if (wv >= 2) return;
        if (n > 0) {
            double best = errL[(eb + n) * rs + col];
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const int i = sidx[w][lane];
                const double b4 = sbest[w][lane];
                if (i >= 0 && b4 <= best) {
                    best = b4;
                    split = n - 1 - i;
                }
            }
        }
    }
    if (UNI_DBG(8)) return;
#ifdef MCL_UNI_STAMPS
    stamp_t1 = (long long)__builtin_amdgcn_s_memrealtime();
#endif
    // emit, positions in lockstep over the wave (EB records per batch, the next batch in flight).
    // Left fit: chain from position split-1 downwards.
    constexpr int EB = 16;
    const int nm1 = max(n - 1, 0);
    // stores of lanes that have nothing to write at a position go to a per-lane sink (an unused scratch array)
    float *sink = sc.sink + ((long)wblk * 64 + lane) * 2 + (MODE == 2 ? wv : 0);
    if (MODE != 2 || wv == 0) {
        const UniRec *rp = recL + (long)s * rs + col;
        float *zp = Z + (long)s * rs + col;
        int rem = 0;
        float z = 0.f;
        const int jtop = wave_max_i(split) - 1;
        UniRec rn[EB];
#pragma unroll
        for (int u = 0; u < EB; ++u) rn[u] = ld_rec_nt(rp + row_off(min(max(jtop - u, 0), nm1), r));
        for (int j0 = jtop; j0 >= 0; j0 -= EB) {
            UniRec rb[EB];
#pragma unroll
            for (int u = 0; u < EB; ++u) rb[u] = rn[u];
#pragma unroll
            for (int u = 0; u < EB; ++u) rn[u] = ld_rec_nt(rp + row_off(min(max(j0 - EB - u, 0), nm1), r));
#pragma unroll
            for (int u = 0; u < EB; ++u) {  // branch-free (a branch around the stores would turn the counted waits on
                const int j = j0 - u;       // the next batch's loads into waits for every store of this one)
                const bool on = j >= 0 && j < split;
                const bool take = on && rem == 0;
                z = take ? rb[u].lev : z;
                rem = take ? rb[u].len : rem;
                *(on ? zp + row_off(min(max(j, 0), nm1), r) : sink) = z;
                rem -= on ? 1 : 0;
            }
        }
    }
    // Right fit: chain from position split upwards (records of sweep 2 extend to the right of their position).
    if (MODE != 2 || wv == 1) {
        const UniRec *rp = recR + (long)s * rs + col;
        float *zp = Z + (long)s * rs + col;
        int rem = 0;
        float z = 0.f;
        const int jend = wave_max_i(n);
        const int jbot = wave_min_i(live && n > 0 ? split : 0x7fffffff);
        UniRec rn[EB];
        if (jbot < jend) {
#pragma unroll
            for (int u = 0; u < EB; ++u) rn[u] = ld_rec_nt(rp + row_off(min(jbot + u, nm1), r));
        }
        for (int j0 = jbot; j0 < jend; j0 += EB) {
            UniRec rb[EB];
#pragma unroll
            for (int u = 0; u < EB; ++u) rb[u] = rn[u];
#pragma unroll
            for (int u = 0; u < EB; ++u) rn[u] = ld_rec_nt(rp + row_off(min(j0 + EB + u, nm1), r));
#pragma unroll
            for (int u = 0; u < EB; ++u) {
                const int j = j0 + u;
                const bool on = j >= split && j < n;
                const bool take = on && rem == 0;
                z = take ? rb[u].lev : z;
                rem = take ? rb[u].len : rem;
                *(on ? zp + row_off(min(max(j, 0), nm1), r) : sink) = z;
                rem -= on ? 1 : 0;
            }
        }
    }
#ifdef MCL_UNI_STAMPS
    if (MODE == 3 && lane == 0 && sc.stamps != nullptr) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        long long *o = sc.stamps + (long)wblk * 8;
        o[0] = stamp_t0, o[1] = (long long)__builtin_amdgcn_s_memrealtime(), o[4] = stamp_t1;
        o[2] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 4), o[3] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 20);
    }
#endif
}

// =========================================================================================================
// host side
// =========================================================================================================
#ifdef MCL_UNI_DBG
static unsigned long long *uni_dbg_ctr() {
    static unsigned long long *p = nullptr;
    if (!p && hipMalloc(&p, 16 * sizeof(unsigned long long)) == hipSuccess) (void)hipMemset(p, 0, 16 * sizeof(unsigned long long));
    return p;
}
// debug builds only: read and clear the event counters of the unimodal kernels (UNI_CNT)
extern "C" int mcl_uni_dbg_counters(unsigned long long *out16) {  // [0..8) events, [8..13) section cycles, [13] kernel cycles
    if (hipDeviceSynchronize() != hipSuccess || !uni_dbg_ctr()) return 1;
    if (hipMemcpy(out16, uni_dbg_ctr(), 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    return hipMemset(uni_dbg_ctr(), 0, 16 * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
#endif
// Round 6, per-wave stamps of a config-5 launch (tools/uni_stamps.py, profiles/r6_uni_stamps.txt): 4096 waves on 2048 slots, every
// SIMD runs exactly four of them two at a time; a wave takes 2.2 ... 4.2 ms (mean 3.1-3.2) and the launch 7.5-7.7 ms where twice
// the mean would be 6.3: 16 % of the slot time is idle - the second workgroup of a CU slot starts when the slowest of the first
// four waves is done, and the launch ends with its slowest CU.  Measured and dropped: launching the column groups longest-first
// (durations of the previous launch, ranked on the device: k_uni_order) - the CUs end closer together (p10-p90 of their last ends
// 423 -> 228 us) but a wave's duration follows its neighbours as much as its data (the same group one iteration apart: correlation
// 0.94 in index order, 0.47 reordered), the waves take 3 % longer out of their memory order, and the launch is 2 % SLOWER
// (98.1-99.2 against 96.3-96.4 ms per config-5 iteration, same box).
static UniScratch uni_scratch(mcl_context *c) {
    const int64_t maxrows = std::max<int64_t>(c->N, std::max<int64_t>(c->I, c->K));
    const int64_t n1 = (maxrows + std::max<int64_t>(c->I, 1)) * c->r;
    UniScratch s;
    double *d = c->uni_f64;
    s.lvL = d, s.lvR = d + n1, s.eL = d + 2 * n1, s.eR = d + 3 * n1;
    s.spL = d + 4 * n1, s.spR = d + 7 * n1;  // 3 n1 doubles each
    s.sink = c->uni_sink;
#ifdef MCL_UNI_STAMPS
    s.stamps = c->pf2_xmin ? reinterpret_cast<long long *>(c->pf2_xmin + c->I) : nullptr;
#endif
    s.coop = c->sw.no_uni_coop ? 0 : 1;
#ifdef MCL_UNI_DBG
    s.dbg = getenv("MCL_UNI_DBG") ? atoi(getenv("MCL_UNI_DBG")) : 0;
    s.ctr = uni_dbg_ctr();
#endif
    return s;
}

// Unimodal prox of penalty k on the factor F (slab extents ext[0 .. n_slabs]); the dual step is the caller's.
int mcl_launch_unimodal(mcl_context *c, const int *ext, int n_slabs, float *F, const RegSet &rs, int mode, int k) {
    const long nthreads = (long)n_slabs * c->r;
    if (nthreads == 0) return 0;
    // the kernels index a column by 24-bit row numbers (row_off): a single matrix of 2^24 rows under this constraint is refused
    const int64_t longest = mode == 1 ? c->max_slab_rows : (mode == 0 ? c->I : c->K);
    if (longest >= (int64_t(1) << 24)) {
        c->err = "unimodality: a factor matrix of 2^24 or more rows is not supported";
        return 1;
    }
    UniScratch sc = uni_scratch(c);
    const unsigned nwav = (unsigned)((nthreads + 63) / 64);
    // fewer columns than about one wave per SIMD: the two sweeps run concurrently in different waves
    int wave_split = nwav <= 1024;
    if (c->sw.uni_split >= 0) wave_split = c->sw.uni_split;
    c->variant[MCL_PROF_UNIMODAL] = wave_split ? "k_slab_unimodal_v4<1> + k_slab_unimodal_v4<2>"
                                               : (c->sw.uni_noprune ? "k_slab_unimodal_v4<0>" : "k_slab_unimodal_v4<3>");
    if (wave_split) {
        hipLaunchKernelGGL(k_slab_unimodal_v4<1>, dim3(2 * nwav), dim3(64), 0, c->stream, ext, n_slabs, F, rs, k, c->r, sc);
        hipLaunchKernelGGL(k_slab_unimodal_v4<2>, dim3(nwav), dim3(256), 0, c->stream, ext, n_slabs, F, rs, k, c->r, sc);
    } else if (c->sw.uni_noprune) {
        hipLaunchKernelGGL(k_slab_unimodal_v4<0>, dim3(nwav), dim3(64), 0, c->stream, ext, n_slabs, F, rs, k, c->r, sc);
    } else {
        size_t pad = 0;
#ifdef MCL_UNI_DBG  // occupancy experiments: unused dynamic LDS limits the workgroups a CU holds (tools/uni_occ.py)
        pad = getenv("MCL_UNI_PAD_LDS") ? (size_t)atoi(getenv("MCL_UNI_PAD_LDS")) : 0;
#endif
        constexpr size_t wave_lds = (size_t)(2 * MCL_UNI_RC * 64) * sizeof(double) + (size_t)(MCL_UNI_RC * 64) * sizeof(int);
        // waves per workgroup: 4 (see the kernel), MCL_UNI_WPB = 1 / 2 / 4 for the A/B; the padded debug launches keep the one-wave form
        int wpb = c->sw.uni_wpb > 0 ? c->sw.uni_wpb : 4;
        if (pad != 0 || (wpb != 2 && wpb != 4)) wpb = 1;
        int (&attr_set)[5] = c->uni_attr_set;  // per context (= per device): 0 not tried, 1 set, -1 refused
        if (wpb > 1 && attr_set[wpb] == 0) {
            const void *fn = wpb == 4 ? reinterpret_cast<const void *>(k_slab_unimodal_v4<3, 4>) : reinterpret_cast<const void *>(k_slab_unimodal_v4<3, 2>);
            attr_set[wpb] = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(wpb * wave_lds)) == hipSuccess ? 1 : -1;
            if (attr_set[wpb] < 0) (void)hipGetLastError();
        }
        if (wpb > 1 && attr_set[wpb] < 0) wpb = 1;
#ifdef MCL_UNI_DBG
        wpb = 1;  // the instrumented kernel has early exits: one-wave form only
#endif
        if (wpb == 4) {
            c->variant[MCL_PROF_UNIMODAL] = "k_slab_unimodal_v4<3, WPB=4>";
            hipLaunchKernelGGL((k_slab_unimodal_v4<3, 4>), dim3((nwav + 3) / 4), dim3(256), 4 * wave_lds, c->stream, ext, n_slabs, F, rs, k, c->r, sc);
        } else if (wpb == 2) {
            c->variant[MCL_PROF_UNIMODAL] = "k_slab_unimodal_v4<3, WPB=2>";
            hipLaunchKernelGGL((k_slab_unimodal_v4<3, 2>), dim3((nwav + 1) / 2), dim3(128), 2 * wave_lds, c->stream, ext, n_slabs, F, rs, k, c->r, sc);
        } else
            hipLaunchKernelGGL(k_slab_unimodal_v4<3>, dim3(nwav), dim3(64), pad, c->stream, ext, n_slabs, F, rs, k, c->r, sc);
    }
    MCL_CHECK_HIP(c, hipGetLastError());
    return 0;
}
