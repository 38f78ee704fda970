// Calibration of rocprofv3's FETCH_SIZE on MI355X for the access shapes the HEM kernels use.
// MI355X_MICROARCH.md: "on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read
// (16 B/lane) ... other access widths are uncalibrated: calibrate on a known byte count in your own access pattern".
// Every kernel below reads a KNOWN number of distinct bytes exactly once from a 2 GiB buffer (far beyond the 256 MiB
// Infinity Cache), in one of the shapes of csrc/hem.hip:
//   k_stream16   16 B per lane, consecutive lanes consecutive addresses          (k_prep / k_gather_sh style streams)
//   k_stream4    4 B per lane                                                    (pair_child / pair_wl streams)
//   k_rec64      random 64-B records, four lanes x 16 B per record               (geo gathers of k_mstep part 1)
//   k_rec64_1    random 64-B records, ONE lane reads the four float4 of a record (stage 2 of k_select)
//   k_row192     random 192-B rows, four lanes x three float4 (64-B stride)      (SH rows of k_mstep part 2)
//   k_pos16      random 16-B records, one lane each                              (scattered A[] reads)
// Build: hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip
// Run:   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- ./fetch_calib
//        and compare FETCH_SIZE (KiB) x 1024 of every kernel with the "expect" column printed here.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

__device__ __forceinline__ unsigned perm(unsigned i, unsigned n_pow2_mask) {      // a bijection on [0, 2^k): odd multiplier + xorshift
    unsigned x = (i * 0x9E3779B1u) & n_pow2_mask;
    x ^= x >> 7; x &= n_pow2_mask;
    x = (x * 0x85EBCA6Bu) & n_pow2_mask;
    x ^= x >> 11; x &= n_pow2_mask;
    return x;
}
__global__ void k_stream16(const float4* __restrict__ p, size_t n16, float* out) {
    float s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { float4 v = p[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 123.456f) *out = s;
}
__global__ void k_stream4(const float* __restrict__ p, size_t n4, float* out) {
    float s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) s += p[i];
    if (s == 123.456f) *out = s;
}
__global__ void k_rec64(const float4* __restrict__ p, unsigned nrec_mask, size_t nrec, float* out) {      // 4 lanes per record
    float s = 0;
    const size_t t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t t = t0; t < nrec * 4; t += stride) { float4 v = p[4 * (size_t)perm((unsigned)(t >> 2), nrec_mask) + (t & 3)]; s += v.x + v.w; }
    if (s == 123.456f) *out = s;
}
__global__ void k_rec64_1(const float4* __restrict__ p, unsigned nrec_mask, size_t nrec, float* out) {    // one lane per record
    float s = 0;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < nrec; t += (size_t)gridDim.x * blockDim.x) {
        const float4* r = p + 4 * (size_t)perm((unsigned)t, nrec_mask);
        float4 a = r[0], b = r[1], c = r[2], d = r[3];
        s += a.x + b.y + c.z + d.w;
    }
    if (s == 123.456f) *out = s;
}
__global__ void k_row192(const float4* __restrict__ p, unsigned nrow_mask, size_t nrow, float* out) {     // rows of 12 float4: 4 lanes x 3
    float s = 0;
    const size_t t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t t = t0; t < nrow * 4; t += stride) {
        const float4* r = p + 12 * (size_t)perm((unsigned)(t >> 2), nrow_mask);
        const int gl = (int)(t & 3);
        float4 a = r[gl], b = r[gl + 4], c = r[gl + 8];
        s += a.x + b.y + c.z;
    }
    if (s == 123.456f) *out = s;
}
__global__ void k_pos16(const float4* __restrict__ p, unsigned n_mask, size_t n, float* out) {
    float s = 0;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) { float4 v = p[perm((unsigned)t, n_mask)]; s += v.x; }
    if (s == 123.456f) *out = s;
}

int main() {
    const size_t bytes = (size_t)2 << 30;
    float4* buf; float* out;
    hipMalloc(&buf, bytes); hipMalloc(&out, 4);
    hipMemset(buf, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char* name, double expect, auto launch) {
        launch();                                       // warm (code object)
        hipDeviceSynchronize();
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-10s expect %.4f GB per launch   %.3f ms  %.2f TB/s\n", name, expect / 1e9, ms, expect / ms / 1e9);
    };
    const dim3 g(8192), b(256);
    timeit("k_stream16", (double)bytes, [&] { hipLaunchKernelGGL(k_stream16, g, b, 0, 0, buf, bytes / 16, out); });
    timeit("k_stream4", (double)bytes / 2, [&] { hipLaunchKernelGGL(k_stream4, g, b, 0, 0, (const float*)buf, bytes / 8, out); });
    {   // 2^24 records of 64 B = 1 GiB of distinct bytes out of the 2 GiB buffer... use all 2^25 records
        const size_t nrec = (size_t)1 << 25;
        timeit("k_rec64", (double)nrec * 64, [&] { hipLaunchKernelGGL(k_rec64, g, b, 0, 0, buf, (unsigned)(nrec - 1), nrec, out); });
        timeit("k_rec64_1", (double)nrec * 64, [&] { hipLaunchKernelGGL(k_rec64_1, g, b, 0, 0, buf, (unsigned)(nrec - 1), nrec, out); });
    }
    {   // 2^23 rows of 192 B = 1.5 GiB
        const size_t nrow = (size_t)1 << 23;
        timeit("k_row192", (double)nrow * 192, [&] { hipLaunchKernelGGL(k_row192, g, b, 0, 0, buf, (unsigned)(nrow - 1), nrow, out); });
    }
    {   // 2^26 records of 16 B = 1 GiB; a 16-B read fetches at least its 64-B (or 128-B) line: the counter says which
        const size_t n = (size_t)1 << 26;
        timeit("k_pos16", (double)n * 16, [&] { hipLaunchKernelGGL(k_pos16, g, b, 0, 0, buf, (unsigned)(n - 1), n, out); });
    }
    return 0;
}
