// Row-tile helpers shared by the fused and the generic ADMM kernels (gfx950).
//
// A wave owns a tile of <= 64 rows of one slab = 4 blocks of 16 rows.  Lane l = (row16 = l&15, g = l>>4) holds, for
// every 16-column block h, the 4 consecutive columns 16h + 4g .. +3 of its row: one 16-B access per (row block, h);
// the 64 lanes of a wave cover 16 full rows = one contiguous 1 KB region when r = 16.
//
// out = t M (M: r x r, wave-uniform) as v_mfma_f32_16x16x4_f32 on the TRANSPOSED problem, with the k-permutation
// k = 16h + 4g + kq that makes the input fragment layout equal to the output layout:
//     A (lane (c' = l&15, g)) = M[k][16h' + c']      B (lane (row = l&15, g)) = t[row][k]
//     D (lane l, reg v)       = out[row = l&15][16h' + 4g + v]
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/matcouply_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
#ifndef MFMA16
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#endif

// Sum of a double over the 64 lanes, the total in every lane.  Four DPP steps inside each 16-lane row (quad_perm xor 1,
// xor 2, row_half_mirror, row_mirror: a few cycles each), then the four row totals through v_readlane - a butterfly of
// six ds_bpermute pairs costs ~1200 cycles of dependent latency, which bounded the Newton-Schulz iteration of the
// PARAFAC2 kernel (one norm per step) and the tail of every row pass.  Fixed association: ((r0 + r1) + (r2 + r3)).
template <int CTRL>
static __device__ __forceinline__ double dpp_mov_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
static __device__ __forceinline__ double readlane_f64_u(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
static __device__ __forceinline__ double wave_sum(double v) {
    v += dpp_mov_f64<0xB1>(v);   // quad_perm [1, 0, 3, 2]
    v += dpp_mov_f64<0x4E>(v);   // quad_perm [2, 3, 0, 1]
    v += dpp_mov_f64<0x141>(v);  // row_half_mirror
    v += dpp_mov_f64<0x140>(v);  // row_mirror: every lane of a row holds the row's total
    return (readlane_f64_u(v, 0) + readlane_f64_u(v, 16)) + (readlane_f64_u(v, 32) + readlane_f64_u(v, 48));
}

// elementwise prox of the row-separable penalties (penalties.py:503-586)
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

// branch-free form of prox_elem with identical results:  prox(y) = clamp(y - clamp(y, pa, pb), plo, phi), clamp = v_med3_f32
//   NN: (-inf, 0 | -inf, inf)   Box: (0, 0 | lo, hi)   L1: (-thr, thr | -inf, inf)   L1 + NN: (-inf, thr | -inf, inf)
// The finish kernels run one wave per SIMD with a serial dependence chain: every branch in their inner loop is exposed.
struct ProxClamp {
    float pa, pb, plo, phi;
    __device__ __forceinline__ void set(int kind, int nonneg, float p0, float p1, float thr) {
        pa = pb = 0.f;
        plo = -INFINITY, phi = INFINITY;
        if (kind == MCL_PEN_NN) pa = -INFINITY;
        if (kind == MCL_PEN_BOX) plo = p0, phi = p1;
        if (kind == MCL_PEN_L1) pa = nonneg ? -INFINITY : -thr, pb = thr;
    }
    __device__ __forceinline__ float operator()(float y) const {
        return __builtin_amdgcn_fmed3f(y - __builtin_amdgcn_fmed3f(y, pa, pb), plo, phi);
    }
};

// the same operator in fp64 (the A-phase finish of small problems keeps its inner loop in double: admm.hip, wide_inner)
struct ProxClampD {
    double pa, pb, plo, phi;
    __device__ __forceinline__ void set(int kind, int nonneg, double p0, double p1, double thr) {
        pa = pb = 0.0;
        plo = -INFINITY, phi = INFINITY;
        if (kind == MCL_PEN_NN) pa = -INFINITY;
        if (kind == MCL_PEN_BOX) plo = p0, phi = p1;
        if (kind == MCL_PEN_L1) pa = nonneg ? -INFINITY : -thr, pb = thr;
    }
    __device__ __forceinline__ double operator()(double y) const { return fmin(fmax(y - fmin(fmax(y, pa), pb), plo), phi); }
};

template <int NBR>
struct RowMat {
    float m[NBR][NBR][4];  // m[h'][h][kq] = M[16h + 4g + kq][16h' + row16]
    __device__ __forceinline__ void load(const float *M, int r, int lane) {
        const int row16 = lane & 15, g = lane >> 4;
#pragma unroll
        for (int hp = 0; hp < NBR; ++hp)
#pragma unroll
            for (int h = 0; h < NBR; ++h)
#pragma unroll
                for (int kq = 0; kq < 4; ++kq) {
                    const int k = 16 * h + 4 * g + kq, c = 16 * hp + row16;
                    m[hp][h][kq] = (k < r && c < r) ? M[k * r + c] : 0.f;
                }
    }
    __device__ __forceinline__ void apply(const f32x4 (&t)[NBR], f32x4 (&out)[NBR]) const {
#pragma unroll
        for (int hp = 0; hp < NBR; ++hp) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int h = 0; h < NBR; ++h)
#pragma unroll
                for (int kq = 0; kq < 4; ++kq) acc = MFMA16(m[hp][h][kq], t[h][kq], acc);
            out[hp] = acc;
        }
    }
};

template <bool VEC>
static __device__ __forceinline__ f32x4 row_ld4(const float *__restrict__ base, long j, int col, bool ok, int r) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (VEC) {
        if (ok && col < r) v = *reinterpret_cast<const f32x4 *>(base + j * r + col);
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (ok && col + q < r) v[q] = base[j * r + col + q];
    }
    return v;
}

template <bool VEC>
static __device__ __forceinline__ void row_st4(float *__restrict__ base, long j, int col, bool ok, int r, f32x4 v) {
    if (VEC) {
        if (ok && col < r) *reinterpret_cast<f32x4 *>(base + j * r + col) = v;
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (ok && col + q < r) base[j * r + col + q] = v[q];
    }
}

// ---- the same row algebra in fp64 (PARAFAC2 stacks of rank <= 16, generic.hip) ----------------------------------------
// v_mfma_f64_16x16x4_f64 has the operand layouts of the fp32 instruction (A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15])
// but ANOTHER result layout: lane l, register v holds D[i = (l >> 4) + 4 v][j = l & 15].  For the transposed problem
// out^T = M^T t^T the row index i of D is an output COLUMN, so loading the A operand with the columns of M permuted by
// colmap(i) = 4 (i & 3) + (i >> 2) makes register v of lane (row, g) the output column 4 g + v again: input and output
// fragments coincide exactly as in the fp32 form, and products chain without any cross-lane traffic.
typedef double rd64x4 __attribute__((ext_vector_type(4)));

template <int NBR>
struct RowMat64 {
    double m[NBR][NBR][4];  // m[h'][h][kq] = M[16h + 4g + kq][16h' + colmap(row16)]
    template <typename SRC>
    __device__ __forceinline__ void load(const SRC *M, int r, int lane) {
        const int row16 = lane & 15, g = lane >> 4;
        const int cm = 4 * (row16 & 3) + (row16 >> 2);
#pragma unroll
        for (int hp = 0; hp < NBR; ++hp)
#pragma unroll
            for (int h = 0; h < NBR; ++h)
#pragma unroll
                for (int kq = 0; kq < 4; ++kq) {
                    const int k = 16 * h + 4 * g + kq, c = 16 * hp + cm;
                    m[hp][h][kq] = (k < r && c < r) ? (double)M[k * r + c] : 0.0;
                }
    }
    __device__ __forceinline__ void apply(const rd64x4 (&t)[NBR], rd64x4 (&out)[NBR]) const {
#pragma unroll
        for (int hp = 0; hp < NBR; ++hp) {
            rd64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int h = 0; h < NBR; ++h)
#pragma unroll
                for (int kq = 0; kq < 4; ++kq) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(m[hp][h][kq], t[h][kq], acc, 0, 0, 0);
            out[hp] = acc;
        }
    }
    // fp32 rows in and / or out: exact widening, ONE rounding of the result
    __device__ __forceinline__ void apply(const rd64x4 (&t)[NBR], f32x4 (&out)[NBR]) const {
        rd64x4 o[NBR];
        apply(t, o);
#pragma unroll
        for (int h = 0; h < NBR; ++h) out[h] = f32x4{(float)o[h][0], (float)o[h][1], (float)o[h][2], (float)o[h][3]};
    }
    __device__ __forceinline__ void apply(const f32x4 (&t)[NBR], f32x4 (&out)[NBR]) const {
        rd64x4 w[NBR];
#pragma unroll
        for (int h = 0; h < NBR; ++h) w[h] = rd64x4{(double)t[h][0], (double)t[h][1], (double)t[h][2], (double)t[h][3]};
        apply(w, out);
    }
};

// Row arithmetic of a kernel.  R64 = false: everything in fp32.  R64 = true (PARAFAC2 stacks of rank <= 16): the vectors
// stay fp32 - they are what is stored - but every r x r product runs on the fp64 MFMA with fp64 matrices, the polar input
// Y = F + U is the EXACT fp64 sum of the two stored values, and P = Y T_i stays in fp64 until P Delta has been formed.  Each
// product's result is rounded once, where it is stored anyway.
template <bool R64>
struct RowArith;
template <>
struct RowArith<false> {
    typedef f32x4 Y;
    template <int NBR>
    using Mat = RowMat<NBR>;
    static __device__ __forceinline__ Y ysum(f32x4 f, f32x4 u) { return f + u; }
    static __device__ __forceinline__ f32x4 narrow(Y v) { return v; }
};
template <>
struct RowArith<true> {
    typedef rd64x4 Y;
    template <int NBR>
    using Mat = RowMat64<NBR>;
    static __device__ __forceinline__ Y ysum(f32x4 f, f32x4 u) {
        return Y{(double)f[0] + (double)u[0], (double)f[1] + (double)u[1], (double)f[2] + (double)u[2], (double)f[3] + (double)u[3]};
    }
    static __device__ __forceinline__ f32x4 narrow(Y v) { return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]}; }
};
