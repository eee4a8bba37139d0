// Do VALU instructions of the SAME wave issue while its MFMAs execute?  (1 wave per SIMD, like k_sweep.)
//   mode 0: 64 independent MFMAs, then 192 VALU FMAs          mode 1: the same work, 1 MFMA : 3 VALU interleaved
//   mode 2: MFMAs only                                         mode 3: VALU only
// The same four modes with v_mfma_f32_32x32x2_f32 (16 passes, 32 of them = the same 2048 busy cycles): k32<MODE>.
// build: hipcc --offload-arch=gfx950 -O3 tools/coexec.hip -o gpurun_out/coexec
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

template <int MODE>
__global__ __launch_bounds__(512) void k(float *out, int iters, float seed) {
    f32x4 acc[16];
    float v[12];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{seed, seed, seed, seed};
    for (int i = 0; i < 12; ++i) v[i] = seed + i + threadIdx.x;
    const float a = seed * 1.0001f, b = seed * 0.9999f;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = MF(a, b, acc[i]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE == 0 || MODE == 3) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int i = 0; i < 12; ++i) v[i] = fmaf(v[i], a, b);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE == 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    acc[i] = MF(a, b, acc[i]);
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const int q = ((r * 16 + i) * 3 + j) % 12;
                        v[q] = fmaf(v[q], a, b);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 12; ++i) s += v[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MF32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

template <int MODE>
__global__ __launch_bounds__(256) void k32(float *out, int iters, float seed) {
    f32x16 acc[8];
    float v[12];
    for (int i = 0; i < 8; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = seed;
    for (int i = 0; i < 12; ++i) v[i] = seed + i + threadIdx.x;
    const float a = seed * 1.0001f, b = seed * 0.9999f;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = MF32(a, b, acc[i]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE == 0 || MODE == 3) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int i = 0; i < 12; ++i) v[i] = fmaf(v[i], a, b);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE == 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    acc[i] = MF32(a, b, acc[i]);
#pragma unroll
                    for (int j = 0; j < 6; ++j) {
                        const int q = ((r * 8 + i) * 6 + j) % 12;
                        v[q] = fmaf(v[q], a, b);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i)
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    for (int i = 0; i < 12; ++i) s += v[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

// The same question for the bf16 matrix core (v_mfma_f32_16x16x32_bf16, 16 cycles each): 64 of them = 1024 busy cycles, with the
// same 192 (mode 0 / 1) VALU FMAs - does a three-way bf16 split of fp32 operands (VALU work) hide behind its own MFMAs?
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MFB(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

template <int MODE>
__global__ __launch_bounds__(512) void kb(float *out, int iters, float seed) {
    f32x4 acc[16];
    float v[12];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{seed, seed, seed, seed};
    for (int i = 0; i < 12; ++i) v[i] = seed + i + threadIdx.x;
    const float a = seed * 1.0001f, b = seed * 0.9999f;
    bf16x8 pa, pb;
    for (int i = 0; i < 8; ++i) pa[i] = (__bf16)(seed * 0.01f * (i + 1)), pb[i] = (__bf16)(seed * 0.02f * (i + 1));
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = MFB(pa, pb, acc[i]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE == 0 || MODE == 3) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int i = 0; i < 12; ++i) v[i] = fmaf(v[i], a, b);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE == 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    acc[i] = MFB(pa, pb, acc[i]);
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const int q = ((r * 16 + i) * 3 + j) % 12;
                        v[q] = fmaf(v[q], a, b);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 12; ++i) s += v[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE>
void runb(float *out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    kb<MODE><<<256, 256>>>(out, 10, 1.f);
    hipEventRecord(e0);
    kb<MODE><<<256, 256>>>(out, 2000, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("bf16 16x16x32 mode %d: %.1f us per 2000 iterations -> %.0f cycles/iter at 2.4 GHz (64 MFMA = 1024 busy, 192 VALU = 768 issue)\n", MODE,
           ms * 1e3, ms * 1e-3 / 2000 * 2.4e9);
}

// two waves per SIMD (512 threads per workgroup, one workgroup per CU): does one wave's vector work run beside the OTHER wave's MFMAs?
template <int MODE, bool BF>
void run2(float *out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    if (BF) kb<MODE><<<256, 512>>>(out, 10, 1.f); else k<MODE><<<256, 512>>>(out, 10, 1.f);
    hipEventRecord(e0);
    if (BF) kb<MODE><<<256, 512>>>(out, 2000, 1.f); else k<MODE><<<256, 512>>>(out, 2000, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("TWO waves per SIMD, %s mode %d: %.0f cycles per iteration of BOTH waves (each: 64 MFMA + 192 VALU in modes 0 / 1)\n",
           BF ? "bf16 16x16x32" : "fp32 16x16x4", MODE, ms * 1e-3 / 2000 * 2.4e9);
}

template <int MODE>
void run32(float *out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k32<MODE><<<256, 256>>>(out, 10, 1.f);
    hipEventRecord(e0);
    k32<MODE><<<256, 256>>>(out, 2000, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("32x32x2 mode %d: %.1f us per 2000 iterations -> %.0f cycles/iter at 2.4 GHz (32 MFMA = 2048 busy, 192 VALU = 768 issue)\n", MODE,
           ms * 1e3, ms * 1e-3 / 2000 * 2.4e9);
}

template <int MODE>
void run(float *out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<256, 256>>>(out, 10, 1.f);
    hipEventRecord(e0);
    k<MODE><<<256, 256>>>(out, 2000, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("mode %d: %.1f us per 2000 iterations -> %.0f cycles/iter at 2.4 GHz (64 MFMA = 2048 busy, 192 VALU = 768 issue)\n", MODE,
           ms * 1e3, ms * 1e-3 / 2000 * 2.4e9);
}

int main() {
    float *out; hipMalloc(&out, 256 * 512 * 4);
    run<2>(out); run<3>(out); run<0>(out); run<1>(out);
    run32<2>(out); run32<3>(out); run32<0>(out); run32<1>(out);
    runb<2>(out); runb<3>(out); runb<0>(out); runb<1>(out);
    run2<2, false>(out); run2<3, false>(out); run2<0, false>(out); run2<1, false>(out);
    run2<2, true>(out); run2<3, true>(out); run2<0, true>(out); run2<1, true>(out);
    return 0;
}
