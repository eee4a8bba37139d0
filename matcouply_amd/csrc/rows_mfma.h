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
