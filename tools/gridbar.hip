// What does a grid-wide barrier INSIDE a kernel cost on MI355X, against the boundary between two dependent launches?
// (one persistent kernel per outer iteration for L2-resident problems - "two grid barriers replace three
// launch boundaries"; the only in-kernel fan-in measured so far was a 256-way ticket into ONE workgroup, 9 us.)
//   launch   : N dependent launches of an (almost) empty kernel of G workgroups on one stream -> time per launch
//   flat     : one kernel, N barriers: every workgroup adds 1 to a counter (agent scope) and polls it
//   2-level  : arrival per XCD (HW_REG_XCC_ID), the last arriver of an XCD adds 1 to the global counter, everyone polls the global one
//   2-level* : as above with the per-XCD arrival at WORKGROUP scope (served by the XCD's own L2: outside the memory model, timing only)
// Every spin is bounded (no hang if the grid is not co-resident); `err` reports a timeout.
// build: hipcc --offload-arch=gfx950 -O3 tools/gridbar.hip -o gpurun_out/gridbar
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(256) void k_nop(unsigned *sink) {
    if (sink != nullptr && threadIdx.x == 0 && blockIdx.x == 0xffffffu) *sink = 1;
}

static __device__ __forceinline__ bool spin_until(unsigned *p, unsigned target, unsigned *err) {
    int spins = 0;
    while (__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if (++spins > (1 << 20)) {
            *err = 1;
            return false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    return true;
}

__global__ __launch_bounds__(256) void k_flat(unsigned *ctr, int nbar, unsigned *err, float *work) {
    unsigned target = 0;
    float acc = 0.f;
    for (int b = 0; b < nbar; ++b) {
        acc += work ? work[(blockIdx.x * 256 + threadIdx.x) & 1023] : 0.f;
        __syncthreads();
        if (threadIdx.x == 0) {
            target += gridDim.x;
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            if (!spin_until(ctr, target, err)) nbar = 0;
        }
        __syncthreads();
    }
    if (work && acc == 123.f) work[0] = acc;
}

// xcd_n[8]: workgroups per XCD (counted by the kernel's own prologue + one flat barrier); xcd_ctr[8 * 32]: padded per-XCD counters
template <bool LOCAL>
__global__ __launch_bounds__(256) void k_two_level(unsigned *ctr, unsigned *xcd_n, unsigned *xcd_ctr, unsigned *gen, int nbar, unsigned *err) {
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7;  // HW_REG_XCC_ID[3:0]
    __shared__ unsigned n_mine;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(xcd_n + xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        spin_until(ctr, gridDim.x, err);
        n_mine = __hip_atomic_load(xcd_n + xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const unsigned mine = n_mine;
    unsigned local_target = 0, g_target = 0;
    unsigned n_xcd = 0;
    for (int x = 0; x < 8; ++x) n_xcd += __hip_atomic_load(xcd_n + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    for (int b = 0; b < nbar; ++b) {
        __syncthreads();
        if (threadIdx.x == 0) {
            local_target += mine;
            g_target += n_xcd;
            unsigned old;
            if (LOCAL) old = __hip_atomic_fetch_add(xcd_ctr + 32 * xcc, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            else old = __hip_atomic_fetch_add(xcd_ctr + 32 * xcc, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            if (old + 1 == local_target) __hip_atomic_fetch_add(gen, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            if (!spin_until(gen, g_target, err)) nbar = 0;
        }
        __syncthreads();
    }
}

int main(int argc, char **argv) {
    const int nbar = argc > 1 ? atoi(argv[1]) : 2000;
    unsigned *d;
    float *work;
    hipMalloc(&d, 4096 * sizeof(unsigned));
    hipMalloc(&work, 1024 * sizeof(float));
    hipMemset(work, 0, 1024 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grids[] = {256, 256, 512, 1024, 2048};
    for (int gi = 0; gi < 5; ++gi) {
        const int G = grids[gi];
        for (int threads : {64, 256}) {
            float ms;
            unsigned h[16];
            // dependent launches
            hipMemset(d, 0, 4096 * sizeof(unsigned));
            for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k_nop, dim3(G), dim3(threads), 0, 0, d);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int i = 0; i < nbar; ++i) hipLaunchKernelGGL(k_nop, dim3(G), dim3(threads), 0, 0, d);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            const double us_launch = ms * 1e3 / nbar;
            // flat barrier
            hipMemset(d, 0, 4096 * sizeof(unsigned));
            hipLaunchKernelGGL(k_flat, dim3(G), dim3(threads), 0, 0, d, 10, d + 8, (float *)nullptr);
            hipDeviceSynchronize();
            hipMemset(d, 0, 4096 * sizeof(unsigned));
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_flat, dim3(G), dim3(threads), 0, 0, d, nbar, d + 8, (float *)nullptr);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
            const double us_flat = ms * 1e3 / nbar;
            const unsigned err_flat = h[8];
            // two-level
            double us2[2];
            unsigned err2[2], nx[8];
            for (int local = 0; local < 2; ++local) {
                hipMemset(d, 0, 4096 * sizeof(unsigned));
                hipEventRecord(e0);
                if (local) hipLaunchKernelGGL(k_two_level<true>, dim3(G), dim3(threads), 0, 0, d, d + 16, d + 64, d + 32, nbar, d + 8);
                else hipLaunchKernelGGL(k_two_level<false>, dim3(G), dim3(threads), 0, 0, d, d + 16, d + 64, d + 32, nbar, d + 8);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
                hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
                hipMemcpy(nx, d + 16, 32, hipMemcpyDeviceToHost);
                us2[local] = ms * 1e3 / nbar;
                err2[local] = h[8];
            }
            printf("G %4d x %3d threads: dependent launch %.2f us | flat barrier %.2f us (err %u) | two-level %.2f us (err %u) | two-level, XCD-local "
                   "arrival %.2f us (err %u) | workgroups per XCD %u %u %u %u %u %u %u %u\n",
                   G, threads, us_launch, us_flat, err_flat, us2[0], err2[0], us2[1], err2[1], nx[0], nx[1], nx[2], nx[3], nx[4], nx[5], nx[6], nx[7]);
        }
    }
    return 0;
}
