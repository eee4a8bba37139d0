// f64 MFMA issue rate in s_memtime ticks and in wall time (one wave per SIMD, as k_pf2_algebra_ns runs): 4 independent
// accumulation chains of v_mfma_f64_16x16x4_f64, N rounds.  hipcc --offload-arch=gfx950 -O3 tools/mfma64_rate.hip -o /tmp/mfma64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64) void k(double *out, long long *ticks, int rounds) {
    f64x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    double x = 1.0 + threadIdx.x * 1e-9, y = 1.0 - threadIdx.x * 1e-9;
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < rounds; ++i) {
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, a3, 0, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
int main() {
    double *out; long long *ticks;
    const int rounds = 2000;
    hipMalloc(&out, 8192 * 64 * 8); hipMalloc(&ticks, 8192 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int sizes[] = {1024, 1024, 64, 256, 512, 1024, 2048, 4096};
    for (int blocks : sizes) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, out, ticks, rounds);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long h[4]; hipMemcpy(h, ticks, 32, hipMemcpyDeviceToHost);
        printf("blocks %d: %d MFMAs per wave in %.1f us = %.1f ns per MFMA; ticks per MFMA %.1f (=> tick rate %.2f GHz)\n", blocks, 4 * rounds,
               ms * 1e3, ms * 1e6 / (4 * rounds), (double)h[0] / (4 * rounds), (double)h[0] / (ms * 1e6));
    }
    return 0;
}
