// Microbenchmark: throughput of device-scope integer atomics on MI355X for the HEM per-child sums
// (109 M pairs onto 5 M children).  Build: hipcc --offload-arch=gfx950 -O3 -o atomics atomics.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ unsigned h32(unsigned long long x) {
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 27; x *= 0x94d049bb133111ebull; x ^= x >> 31;
    return (unsigned)x;
}
template <int MODE>   // 0: random child, 1: local child (within a window after p * n / m)
__global__ void k_idx(long m, long n, unsigned* child) {
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < m; p += (long)gridDim.x * blockDim.x) {
        if (MODE == 0) child[p] = h32(p) % n;
        else { long base = (long)((double)p * n / m); child[p] = (unsigned)((base + h32(p) % 3000) % n); }
    }
}
__global__ void k_max32(long m, const unsigned* child, unsigned* mx) {
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < m; p += (long)gridDim.x * blockDim.x)
        atomicMax(&mx[child[p]], h32(p + 7) >> 1);
}
__global__ void k_add64(long m, const unsigned* child, unsigned long long* acc) {
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < m; p += (long)gridDim.x * blockDim.x)
        atomicAdd(&acc[child[p]], (unsigned long long)(h32(p + 9)));
}
__global__ void k_addf(long m, const unsigned* child, float* acc) {
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < m; p += (long)gridDim.x * blockDim.x)
        atomicAdd(&acc[child[p]], 1.0f);
}
__global__ void k_plain(long m, const unsigned* child, unsigned* out) {      // baseline: plain scattered store
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < m; p += (long)gridDim.x * blockDim.x)
        out[child[p]] = (unsigned)p;
}
int main() {
    const long m = 109000000, n = 5000000;
    unsigned *child, *mx; unsigned long long* acc; float* accf;
    hipMalloc(&child, m * 4); hipMalloc(&mx, n * 4); hipMalloc(&acc, n * 8); hipMalloc(&accf, n * 4);
    hipMemset(mx, 0, n * 4); hipMemset(acc, 0, n * 8); hipMemset(accf, 0, n * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) {
        if (mode == 0) hipLaunchKernelGGL(k_idx<0>, dim3(4096), dim3(256), 0, 0, m, n, child);
        else hipLaunchKernelGGL(k_idx<1>, dim3(4096), dim3(256), 0, 0, m, n, child);
        for (int which = 0; which < 4; ++which) {
            float best = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (which == 0) hipLaunchKernelGGL(k_max32, dim3(8192), dim3(256), 0, 0, m, child, mx);
                if (which == 1) hipLaunchKernelGGL(k_add64, dim3(8192), dim3(256), 0, 0, m, child, acc);
                if (which == 2) hipLaunchKernelGGL(k_addf, dim3(8192), dim3(256), 0, 0, m, child, accf);
                if (which == 3) hipLaunchKernelGGL(k_plain, dim3(8192), dim3(256), 0, 0, m, child, mx);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            const char* names[] = {"atomicMax u32", "atomicAdd u64", "atomicAdd f32", "plain store"};
            printf("%s children: %-14s %.3f ms  (%.1f G/s)\n", mode ? "local " : "random", names[which], best, m / best / 1e6);
        }
    }
    return 0;
}
