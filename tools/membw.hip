// Read-bandwidth microbenchmark: which access geometry reaches the HBM rate on MI355X?
//   mode 0: classic grid-stride float4 reads (consecutive lanes/waves/blocks read consecutive memory)
//   mode 1: every wave streams its own contiguous chunk (chunk = bytes/waves), 4 x 1 KB rows per step, UNROLL steps in flight
//   mode 2: as 1, but each wave starts at a rotated offset inside its chunk
//   mode 3: waves of a block interleave 4 KB groups inside the block's contiguous region
// build: hipcc --offload-arch=gfx950 -O3 tools/membw.hip -o gpurun_out/membw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int UNROLL>
__global__ __launch_bounds__(256) void k_read(const float* __restrict__ x, long n_f4, int mode, long chunk_f4, float* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long w = (long)blockIdx.x * 4 + wave;
    f32x4 acc = {0, 0, 0, 0};
    const f32x4* p = reinterpret_cast<const f32x4*>(x);
    if (mode == 0) {
        const long stride = (long)gridDim.x * 256;
        long i = (long)blockIdx.x * 256 + threadIdx.x;
        for (; i + (UNROLL - 1) * stride < n_f4; i += UNROLL * stride) {
            f32x4 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) v[u] = p[i + u * stride];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) acc += v[u];
        }
    } else if (mode == 1 || mode == 2) {
        const long base = w * chunk_f4;
        const long steps = chunk_f4 / 64;  // one step = 64 lanes x 16 B = 1 KB
        long rot = (mode == 2) ? ((w * 37) % (steps / UNROLL)) * UNROLL : 0;
        for (long s = 0; s < steps; s += UNROLL) {
            long ss = s + rot;
            if (ss >= steps) ss -= steps;
            f32x4 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) v[u] = p[base + (ss + u) * 64 + lane];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) acc += v[u];
        }
    } else if (mode == 4) {
        // the X^T(B o a) kernel's pattern: per step a wave reads 4 consecutive 1 KB rows as 4 loads of
        // (4 rows x 256 B): lane (rsub = lane>>4, c16 = lane&15) reads row 4g+rsub, floats 64kb + 4c16
        const long base = w * chunk_f4;              // in float4 units; a row = 64 float4
        const long groups = chunk_f4 / 256;          // 4 rows per group
        const int rsub = lane >> 4, c16 = lane & 15;
        for (long g = 0; g < groups; g += UNROLL / 4) {
            f32x4 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const long gg = g + u / 4;
                const int kb = u & 3;
                v[u] = p[base + (gg * 4 + rsub) * 64 + kb * 16 + c16];
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) acc += v[u];
        }
    } else if (mode == 5) {
        // as mode 1 (one full 1 KB row per load) but issued in bursts of 16 loads (the X C kernel's 16-row blocks)
        const long base = w * chunk_f4;
        const long rows = chunk_f4 / 64;
        for (long r0 = 0; r0 < rows; r0 += 16) {
            f32x4 v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = p[base + (r0 + u) * 64 + lane];
#pragma unroll
            for (int u = 0; u < 16; ++u) acc += v[u];
        }
    } else {
        const long bchunk = chunk_f4 * 4;  // block region
        const long base = (long)blockIdx.x * bchunk;
        const long steps = bchunk / 64;
        for (long s = wave * UNROLL; s < steps; s += 4 * UNROLL) {
            f32x4 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) v[u] = p[base + (s + u) * 64 + lane];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) acc += v[u];
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = acc[0];
}

int main() {
    const long bytes = 512L << 20;
    float *x, *out;
    hipMalloc(&x, bytes); hipMalloc(&out, 64);
    hipMemset(x, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const long n_f4 = bytes / 16;
    for (int mode = 0; mode < 6; ++mode) {
        for (int blocks : {256, 512, 1024}) {
            const long waves = (long)blocks * 4;
            const long chunk_f4 = n_f4 / waves;
            for (int rep = 0; rep < 3; ++rep) k_read<16><<<blocks, 256>>>(x, n_f4, mode, chunk_f4, out);
            hipEventRecord(e0);
            const int iters = 10;
            for (int rep = 0; rep < iters; ++rep) k_read<16><<<blocks, 256>>>(x, n_f4, mode, chunk_f4, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("mode %d blocks %4d unroll16: %.1f us  %.2f TB/s\n", mode, blocks, 1e3 * ms / iters, bytes / (ms / iters * 1e-3) / 1e12);
            hipEventRecord(e0);
            for (int rep = 0; rep < iters; ++rep) k_read<4><<<blocks, 256>>>(x, n_f4, mode, chunk_f4, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            printf("mode %d blocks %4d unroll4 : %.1f us  %.2f TB/s\n", mode, blocks, 1e3 * ms / iters, bytes / (ms / iters * 1e-3) / 1e12);
        }
    }
    return 0;
}
