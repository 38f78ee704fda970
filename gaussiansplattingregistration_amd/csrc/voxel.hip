// voxel.hip -- PointCloud::VoxelDownSample on the device (gfx950), the first step of the reference's voxel multiscale
// registration (src/gui/workers/registration/qt_multiscale_registrator.py:127-128 -> Open3D 0.16.0
// cpp/open3d/geometry/PointCloud.cpp VoxelDownSample):
//     voxel_min_bound = min_bound - voxel_size / 2
//     index           = floor((p - voxel_min_bound) / voxel_size) per axis, in float64
//     every voxel averages its points, colours and covariances: sum in insertion order, divided by the count.
// Open3D emits the voxels in unordered_map iteration order (implementation defined); this build emits them in
// ascending (ix, iy, iz) order.
//
// Pipeline: min bound (per-block partials, no atomics) -> 63-bit keys ix:iy:iz (21 bits each) -> rocPRIM stable radix
// sort of (key, point index) -> run heads + exclusive scan -> one thread per voxel sums its run SEQUENTIALLY in
// float64.  The stable sort keeps the points of a voxel in ascending input index = Open3D's insertion order, so the
// float64 means are bit-equal to the oracle's (tests/test_voxel_gpu.py).
#include <float.h>
#include <math.h>
#include <string.h>

#include <rocprim/rocprim.hpp>

#include "gsr_common.h"

namespace gsr {
namespace {

__global__ __launch_bounds__(256) void k_vox_min(int64_t n, const float* __restrict__ xyz, float* __restrict__ part) {
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        for (int k = 0; k < 3; ++k) mn[k] = fminf(mn[k], xyz[3 * i + k]);       // fminf drops NaN
    __shared__ float s_mn[4][3];
    for (int k = 0; k < 3; ++k)
        for (int o = 32; o > 0; o >>= 1) mn[k] = fminf(mn[k], __shfl_xor(mn[k], o));
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 3; ++k) s_mn[threadIdx.x >> 6][k] = mn[k];
    __syncthreads();
    if (threadIdx.x < 3) {
        const int k = threadIdx.x;
        part[3 * blockIdx.x + k] = fminf(fminf(s_mn[0][k], s_mn[1][k]), fminf(s_mn[2][k], s_mn[3][k]));
    }
}
__global__ void k_vox_min_reduce(int nblocks, const float* __restrict__ part, float* __restrict__ out) {
    const int k = threadIdx.x;
    if (k >= 3) return;
    float m = FLT_MAX;
    for (int b = 0; b < nblocks; ++b) m = fminf(m, part[3 * b + k]);
    out[k] = m;
}

struct VoxGeom { double mn[3]; double voxel; };

__global__ __launch_bounds__(256) void k_vox_keys(int64_t n, const float* __restrict__ xyz, VoxGeom g, unsigned long long* __restrict__ keys,
                                                  unsigned* __restrict__ idx, int* __restrict__ bad) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        unsigned long long key = 0;
        for (int a = 0; a < 3; ++a) {
            const double ref = ((double)xyz[3 * i + a] - g.mn[a]) / g.voxel;       // PointCloud.cpp: ref_coord
            const double fl = floor(ref);
            if (!(fl >= 0.0 && fl < 2097152.0)) { *bad = 1; key = 0; break; }         // NaN / outside 21 bits
            key = (key << 21) | (unsigned long long)fl;
        }
        keys[i] = key;
        idx[i] = (unsigned)i;
    }
}
__global__ __launch_bounds__(256) void k_vox_heads(int64_t n, const unsigned long long* __restrict__ skeys, int* __restrict__ head) {
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x)
        head[j] = (j == 0 || skeys[j] != skeys[j - 1]) ? 1 : 0;
}
__global__ __launch_bounds__(256) void k_vox_starts(int64_t n, const int* __restrict__ head, const int* __restrict__ rank, int64_t V,
                                                    int64_t* __restrict__ start) {
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) {
        if (head[j]) start[rank[j]] = j;
        if (j == n - 1) start[V] = n;
    }
}
// one thread per voxel: sequential float64 sums in ascending input index (the stable sort's order)
__global__ __launch_bounds__(256) void k_vox_mean(int64_t V, const int64_t* __restrict__ start, const unsigned* __restrict__ order,
                                                  const float* __restrict__ xyz, const float* __restrict__ cov6, const float* __restrict__ color,
                                                  double* __restrict__ o_xyz, double* __restrict__ o_cov6, double* __restrict__ o_color) {
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < V; v += (int64_t)gridDim.x * blockDim.x) {
        double p[3] = {0, 0, 0}, c[3] = {0, 0, 0}, C[6] = {0, 0, 0, 0, 0, 0};
        const int64_t s = start[v], e = start[v + 1];
        for (int64_t k = s; k < e; ++k) {
            const int64_t i = order[k];
            for (int a = 0; a < 3; ++a) p[a] += (double)xyz[3 * i + a];
            if (cov6) for (int a = 0; a < 6; ++a) C[a] += (double)cov6[6 * i + a];
            if (color) for (int a = 0; a < 3; ++a) c[a] += (double)color[3 * i + a];
        }
        const double d = (double)(e - s);
        for (int a = 0; a < 3; ++a) o_xyz[3 * v + a] = p[a] / d;
        if (cov6) for (int a = 0; a < 6; ++a) o_cov6[6 * v + a] = C[a] / d;
        if (color) for (int a = 0; a < 3; ++a) o_color[3 * v + a] = c[a] / d;
    }
}

}  // namespace
}  // namespace gsr

using namespace gsr;

struct gsr_voxel_result {
    int device = 0;
    hipStream_t stream = nullptr;
    int64_t V = 0;
    bool has_cov = false, has_color = false;
    DevBuf xyz, cov6, color;
};

extern "C" {

int32_t gsr_voxel_down_sample(int32_t device, void* stream, const float* xyz, const float* cov6, const float* color, int64_t n,
                              double voxel_size, int32_t on_device, gsr_voxel_result** out, int64_t* n_voxels) {
    if (!out || !n_voxels) return fail(GSR_E_INVALID, "gsr_voxel_down_sample: NULL argument");
    *out = nullptr; *n_voxels = 0;
    if (!(voxel_size > 0.0)) return fail(GSR_E_PRECONDITION, "[VoxelDownSample] voxel_size <= 0.");
    if (n < 0 || (n > 0 && !xyz)) return fail(GSR_E_INVALID, "gsr_voxel_down_sample: bad input");
    if (n >= ((int64_t)1 << 31) - 1) return fail(GSR_E_INVALID, "gsr_voxel_down_sample: n too large");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(GSR_E_NO_DEVICE, "no HIP device: this library has no CPU fallback");
    GSR_HIP(hipSetDevice(device));
    hipStream_t st = (hipStream_t)stream;
    gsr_voxel_result* r = new gsr_voxel_result();
    r->device = device; r->stream = st; r->has_cov = cov6 != nullptr; r->has_color = color != nullptr;
    *out = r;
    if (n == 0) return GSR_OK;
    // every failure below returns through GSR_HIP / GSR_TRY: the caller frees *out with gsr_voxel_free
    DevBuf sx, sc, sk, part, keys, skeys, idx, order, head, rank, start, tmp, bad;
    struct Guard { DevBuf* b[13]; ~Guard() { for (DevBuf* p : b) p->release(); } } guard{{&sx, &sc, &sk, &part, &keys, &skeys, &idx, &order, &head, &rank, &start, &tmp, &bad}};
    const float *dx = xyz, *dc = cov6, *dk = color;
    if (!on_device) {
        GSR_TRY(sx.reserve((size_t)n * 12));
        GSR_HIP(hipMemcpyAsync(sx.p, xyz, (size_t)n * 12, hipMemcpyHostToDevice, st));
        dx = sx.as<float>();
        if (cov6) { GSR_TRY(sc.reserve((size_t)n * 24)); GSR_HIP(hipMemcpyAsync(sc.p, cov6, (size_t)n * 24, hipMemcpyHostToDevice, st)); dc = sc.as<float>(); }
        if (color) { GSR_TRY(sk.reserve((size_t)n * 12)); GSR_HIP(hipMemcpyAsync(sk.p, color, (size_t)n * 12, hipMemcpyHostToDevice, st)); dk = sk.as<float>(); }
    }
    const int nb = stride_grid(n);
    GSR_TRY(part.reserve((size_t)nb * 12 + 64));
    hipLaunchKernelGGL(k_vox_min, dim3(nb), dim3(256), 0, st, n, dx, part.as<float>() + 16);
    hipLaunchKernelGGL(k_vox_min_reduce, dim3(1), dim3(64), 0, st, nb, part.as<float>() + 16, part.as<float>());
    float hmn[3];
    GSR_HIP(hipMemcpyAsync(hmn, part.p, 12, hipMemcpyDeviceToHost, st));
    GSR_HIP(hipStreamSynchronize(st));
    VoxGeom g;
    for (int a = 0; a < 3; ++a) g.mn[a] = (double)hmn[a] - voxel_size * 0.5;      // voxel_min_bound
    g.voxel = voxel_size;
    GSR_TRY(keys.reserve((size_t)n * 8)); GSR_TRY(skeys.reserve((size_t)n * 8)); GSR_TRY(idx.reserve((size_t)n * 4)); GSR_TRY(order.reserve((size_t)n * 4));
    GSR_TRY(bad.reserve(64));
    GSR_HIP(hipMemsetAsync(bad.p, 0, 4, st));
    hipLaunchKernelGGL(k_vox_keys, dim3(nb), dim3(256), 0, st, n, dx, g, keys.as<unsigned long long>(), idx.as<unsigned>(), bad.as<int>());
    size_t bytes = 0;
    GSR_HIP(rocprim::radix_sort_pairs(nullptr, bytes, keys.as<unsigned long long>(), skeys.as<unsigned long long>(), idx.as<unsigned>(),
                                      order.as<unsigned>(), (size_t)n, 0u, 63u, st));
    GSR_TRY(tmp.reserve(bytes));
    GSR_HIP(rocprim::radix_sort_pairs(tmp.p, bytes, keys.as<unsigned long long>(), skeys.as<unsigned long long>(), idx.as<unsigned>(),
                                      order.as<unsigned>(), (size_t)n, 0u, 63u, st));
    GSR_TRY(head.reserve((size_t)n * 4)); GSR_TRY(rank.reserve((size_t)n * 4));
    hipLaunchKernelGGL(k_vox_heads, dim3(nb), dim3(256), 0, st, n, skeys.as<unsigned long long>(), head.as<int>());
    bytes = 0;
    GSR_HIP(rocprim::exclusive_scan(nullptr, bytes, head.as<int>(), rank.as<int>(), 0, (size_t)n, rocprim::plus<int>(), st));
    GSR_TRY(tmp.reserve(bytes));
    GSR_HIP(rocprim::exclusive_scan(tmp.p, bytes, head.as<int>(), rank.as<int>(), 0, (size_t)n, rocprim::plus<int>(), st));
    int last_rank = 0, last_head = 0, hbad = 0;
    GSR_HIP(hipMemcpyAsync(&last_rank, rank.as<int>() + (n - 1), 4, hipMemcpyDeviceToHost, st));
    GSR_HIP(hipMemcpyAsync(&last_head, head.as<int>() + (n - 1), 4, hipMemcpyDeviceToHost, st));
    GSR_HIP(hipMemcpyAsync(&hbad, bad.p, 4, hipMemcpyDeviceToHost, st));
    GSR_HIP(hipStreamSynchronize(st));
    if (hbad) return fail(GSR_E_PRECONDITION, "[VoxelDownSample] voxel_size is too small (or a coordinate is not finite).");
    const int64_t V = (int64_t)last_rank + last_head;
    GSR_TRY(start.reserve(((size_t)V + 1) * 8));
    hipLaunchKernelGGL(k_vox_starts, dim3(nb), dim3(256), 0, st, n, head.as<int>(), rank.as<int>(), V, start.as<int64_t>());
    GSR_TRY(r->xyz.reserve((size_t)V * 24));
    if (cov6) GSR_TRY(r->cov6.reserve((size_t)V * 48));
    if (color) GSR_TRY(r->color.reserve((size_t)V * 24));
    hipLaunchKernelGGL(k_vox_mean, dim3(stride_grid(V)), dim3(256), 0, st, V, start.as<int64_t>(), order.as<unsigned>(), dx, dc, dk,
                       r->xyz.as<double>(), cov6 ? r->cov6.as<double>() : (double*)nullptr, color ? r->color.as<double>() : (double*)nullptr);
    GSR_HIP(hipStreamSynchronize(st));
    r->V = V;
    *n_voxels = V;
    return GSR_OK;
}

int32_t gsr_voxel_fetch(gsr_voxel_result* r, double* xyz, double* cov6, double* color, int32_t on_device) {
    if (!r) return fail(GSR_E_INVALID, "gsr_voxel_fetch: NULL result");
    GSR_HIP(hipSetDevice(r->device));
    const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    if (r->V > 0) {
        if (xyz) GSR_HIP(hipMemcpyAsync(xyz, r->xyz.p, (size_t)r->V * 24, kind, r->stream));
        if (cov6) {
            if (!r->has_cov) return fail(GSR_E_PRECONDITION, "gsr_voxel_fetch: the input had no covariances");
            GSR_HIP(hipMemcpyAsync(cov6, r->cov6.p, (size_t)r->V * 48, kind, r->stream));
        }
        if (color) {
            if (!r->has_color) return fail(GSR_E_PRECONDITION, "gsr_voxel_fetch: the input had no colours");
            GSR_HIP(hipMemcpyAsync(color, r->color.p, (size_t)r->V * 24, kind, r->stream));
        }
        GSR_HIP(hipStreamSynchronize(r->stream));
    }
    return GSR_OK;
}

int32_t gsr_voxel_free(gsr_voxel_result* r) {
    if (!r) return GSR_OK;
    (void)hipSetDevice(r->device);
    r->xyz.release(); r->cov6.release(); r->color.release();
    delete r;
    return GSR_OK;
}

}  // extern "C"
