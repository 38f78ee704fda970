// icp.hip -- the per-iteration ICP step on MI355X (gfx950), behind include/gsr_hip.h.
//
// Replaces what the reference reaches through Open3D 0.16.0's registration_icp
// (src/utils/local_registration_util.py:76-100): for every source point, transform, nearest target
// neighbour within max_corr, residual / Jacobian, and the reduction that the estimator solves.
// The reference's Open3D walks a KD-tree per point under OpenMP; here:
//
//   target (once per call)  cell-sorted float4 {x, y, z, bits(input index)} + optional double3
//                           normals, dense uniform grid (about 2 points per cell, never finer than
//                           max_corr/8), prefix table cellStart[cells+1]
//   source (once per call)  sorted by the same grid's cell so that neighbouring lanes walk the same cells
//   k_icp_accumulate<KIND>  ONE fused pass per ICP iteration: p = T*p0 in float64, expanding ring
//                           search over contiguous row spans of the sorted target, exact 1-NN
//                           (ties -> lowest input index), strict d^2 < max_corr^2, residual and
//                           Jacobian, float64 accumulators reduced wave -> block -> per-block
//                           partials; k_icp_finalize sums the partials in a fixed order.
//                           No per-point writes; algorithmic bytes 24 (point-to-point) or
//                           48 (point-to-plane, float64 normals) per source point.
//   k_icp_nn                the search alone (thread per point, 56 VGPRs): used with a streaming accumulate kernel
//                           for sources of >= 10^6 points, where occupancy decides (the search is latency bound)
//   estimators              point-to-point (Umeyama), point-to-plane, generalized ICP (W = (Ct + R Cs R^T)^-1/2 per
//                           pair, float64 Jacobi), colored ICP (k_icp_color_gradient prepares the target: 30 nearest
//                           neighbours within 2 max_corr, tangent-plane intensity gradient); robust kernel weights
//   k_icp_step              device-resident loop: reduces the partials, tests convergence, solves (3x3 Jacobi SVD /
//                           6x6 LDL^T) and updates T; the host enqueues chunks of iterations
//   host (this file)        the same solve for the multi-GPU source split (all-reduce callback of 32 doubles).
//
// Accumulator layout (GSR_ICP_ACC_LEN = 32 doubles), see include/gsr_hip.h.
#include "gsr_common.h"
#include "gsr_normals.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <vector>

#include <rocprim/rocprim.hpp>

namespace gsr {
// rocPRIM's Onesweep with its gfx950 kernel shapes but 11 bits per pass: the 22-bit cell keys of a 5 M-point grid take two passes
// instead of three (up to 2^20 keys rocPRIM's merge sort runs as before)
// (the merge sorts of up to 2^20 keys with block sorts of 1 024 x 4 keys: 20-25 us less per target / source sort at 556 k points than rocPRIM's
// default shape, equal at 185 k: profiles/r05aj_icp_merge_sort_configs.txt)
#ifndef GSR_ICP_MERGE_BS
#define GSR_ICP_MERGE_BS 1024
#define GSR_ICP_MERGE_IPT 4
#endif
using icp_merge_cfg = rocprim::merge_sort_config<512, GSR_ICP_MERGE_BS, GSR_ICP_MERGE_IPT, 128, 128, 4>;
using icp_sort_cfg = rocprim::radix_sort_config<rocprim::default_config, icp_merge_cfg,
                                                rocprim::radix_sort_onesweep_config<rocprim::kernel_config<1024, 16>, rocprim::kernel_config<1024, 16>, 11,
                                                                                    rocprim::block_radix_rank_algorithm::match>>;

#ifndef ICP_CELL_FLOOR_DIV
#define ICP_CELL_FLOOR_DIV 64.0     // a grid cell is never finer than max_corr / this (bounds the rings of a query that finds nothing)
#endif
struct IcpGrid {
    double ox, oy, oz, inv_c, c;
    double cx, cy, cz;     // centre used to condition the point-to-point sums
    int gx, gy, gz, ncells;
    int rings;             // ceil(max_corr / c): cells beyond this Chebyshev ring cannot hold an accepted neighbour
    double bx0, by0, bz0, bx1, by1, bz1;      // box of ALL finite target points (the grid may lie over a trimmed one): a query farther from it than max_corr has no neighbour
};

__device__ __forceinline__ int icp_cell(double v, double o, double inv_c, int g) {
    double t = (v - o) * inv_c;
    t = fmin(fmax(t, 0.0), (double)(g - 1));      // NaN -> 0
    return (int)t;
}

__global__ __launch_bounds__(256) void k_icp_bbox(int64_t n, const float* __restrict__ xyz, float* __restrict__ bbox_part) {
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
        if (fabsf(x) <= FLT_MAX && fabsf(y) <= FLT_MAX && fabsf(z) <= FLT_MAX) {
            mn[0] = fminf(mn[0], x); mx[0] = fmaxf(mx[0], x);
            mn[1] = fminf(mn[1], y); mx[1] = fmaxf(mx[1], y);
            mn[2] = fminf(mn[2], z); mx[2] = fmaxf(mx[2], z);
        }
    }
    // per-block partial box, no atomics (10^4 same-address atomics cost ~0.5 ms: they serialise across the XCDs)
    __shared__ float s_mn[4][3], s_mx[4][3];
    for (int k = 0; k < 3; ++k)
        for (int o = 32; o > 0; o >>= 1) { mn[k] = fminf(mn[k], __shfl_xor(mn[k], o)); mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], o)); }
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 3; ++k) { s_mn[threadIdx.x >> 6][k] = mn[k]; s_mx[threadIdx.x >> 6][k] = mx[k]; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int k = threadIdx.x;
        float a = s_mn[0][k], b = s_mx[0][k];
        for (int w = 1; w < 4; ++w) { a = fminf(a, s_mn[w][k]); b = fmaxf(b, s_mx[w][k]); }
        bbox_part[6 * blockIdx.x + k] = a;
        bbox_part[6 * blockIdx.x + 3 + k] = b;
    }
}
// bbox[0..2] = min, bbox[3..5] = max over the per-block partial boxes
__global__ __launch_bounds__(256) void k_icp_bbox_reduce(int nblocks, const float* __restrict__ part, float* __restrict__ bbox) {
    __shared__ float s_v[4][6];
    float v[6] = {FLT_MAX, FLT_MAX, FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int b = threadIdx.x; b < nblocks; b += blockDim.x)
        for (int k = 0; k < 3; ++k) { v[k] = fminf(v[k], part[6 * b + k]); v[3 + k] = fmaxf(v[3 + k], part[6 * b + 3 + k]); }
    for (int k = 0; k < 3; ++k)
        for (int o = 32; o > 0; o >>= 1) { v[k] = fminf(v[k], __shfl_xor(v[k], o)); v[3 + k] = fmaxf(v[3 + k], __shfl_xor(v[3 + k], o)); }
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 6; ++k) s_v[threadIdx.x >> 6][k] = v[k];
    __syncthreads();
    if (threadIdx.x < 6) {
        const int k = threadIdx.x;
        float r = s_v[0][k];
        for (int w = 1; w < 4; ++w) r = k < 3 ? fminf(r, s_v[w][k]) : fmaxf(r, s_v[w][k]);
        bbox[k] = r;
    }
}

// A robust box for the grid.  The reference's scenes come with floaters: a dozen points at 50 scene radii make the box of ALL points
// 10^5 times the scene's volume, the cells -- sized for two points each over that box -- swallow the scene whole, and every query
// scans millions of points (found with bench.py --workload clustered: 116 s per registration).  So: per-axis histograms over the raw
// box (ICP_HIST_BINS bins), the range that holds all but the outermost 0.1 % of the points per side; when that range is less than
// half the raw extent on some axis the grid is laid over it (after ONE refinement pass at the finer resolution), and the points
// beyond are clamped into the boundary cells, which every search treats as half-infinite (icp_slab_dist; the ring logic skips the
// boundary faces anyway).  A cloud without outliers keeps the raw box: nothing changes for it.
#define ICP_HIST_BINS 1024
__global__ __launch_bounds__(256) void k_icp_hist(int64_t n, const float* __restrict__ xyz, const float* __restrict__ box, unsigned* __restrict__ hist) {
    __shared__ unsigned s_h[3 * ICP_HIST_BINS];
    for (int k = threadIdx.x; k < 3 * ICP_HIST_BINS; k += blockDim.x) s_h[k] = 0u;
    __syncthreads();
    float mn[3], sc[3];
    for (int k = 0; k < 3; ++k) {
        mn[k] = box[k];
        const float ext = box[3 + k] - mn[k];
        sc[k] = ext > 0.0f ? (float)ICP_HIST_BINS / ext : 0.0f;
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v[3] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
        if (fabsf(v[0]) <= FLT_MAX && fabsf(v[1]) <= FLT_MAX && fabsf(v[2]) <= FLT_MAX) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                float t = (v[k] - mn[k]) * sc[k];
                t = fminf(fmaxf(t, 0.0f), (float)(ICP_HIST_BINS - 1));
                atomicAdd(&s_h[k * ICP_HIST_BINS + (int)t], 1u);
            }
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 3 * ICP_HIST_BINS; k += blockDim.x)
        if (s_h[k]) atomicAdd(&hist[k], s_h[k]);
}
// box_out = per axis the bins [lo, hi] of box_in that hold all but the outermost 0.1 % of the points per side, widened by one bin on
// either side; *flag = 1 when on some axis that is less than half of box_in's extent.  One thread per axis.
__global__ void k_icp_trim(const unsigned* __restrict__ hist, const float* __restrict__ box_in, float* __restrict__ box_out, float* __restrict__ flag) {
    __shared__ int s_narrow[3];
    if (threadIdx.x < 3) {
        const int k = threadIdx.x;
        const unsigned* h = hist + k * ICP_HIST_BINS;
        unsigned long long total = 0;
        for (int b = 0; b < ICP_HIST_BINS; ++b) total += h[b];
        const unsigned long long trim = total / 1000ull;
        int lo = 0, hi = ICP_HIST_BINS - 1;
        unsigned long long acc = 0;
        while (lo < ICP_HIST_BINS - 1 && acc + h[lo] <= trim) { acc += h[lo]; ++lo; }
        acc = 0;
        while (hi > lo && acc + h[hi] <= trim) { acc += h[hi]; --hi; }
        const float mn = box_in[k], ext = box_in[3 + k] - mn, w = ext / (float)ICP_HIST_BINS;
        lo = lo > 0 ? lo - 1 : 0;
        hi = hi < ICP_HIST_BINS - 1 ? hi + 1 : ICP_HIST_BINS - 1;
        box_out[k] = mn + (float)lo * w;
        box_out[3 + k] = mn + (float)(hi + 1) * w;
        s_narrow[k] = (float)(hi + 1 - lo) * w < 0.5f * ext ? 1 : 0;
    }
    __syncthreads();
    if (threadIdx.x == 0) *flag = (s_narrow[0] | s_narrow[1] | s_narrow[2]) ? 1.0f : 0.0f;
}

__global__ __launch_bounds__(256) void k_icp_keys(int64_t n, const float* __restrict__ xyz, IcpGrid g,
                                                  unsigned* __restrict__ keys, unsigned* __restrict__ idx) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int cx = icp_cell((double)xyz[3 * i], g.ox, g.inv_c, g.gx);
        int cy = icp_cell((double)xyz[3 * i + 1], g.oy, g.inv_c, g.gy);
        int cz = icp_cell((double)xyz[3 * i + 2], g.oz, g.inv_c, g.gz);
        keys[i] = (unsigned)((cz * g.gy + cy) * g.gx + cx);
        idx[i] = (unsigned)i;
    }
}

__global__ __launch_bounds__(256) void k_icp_cell_starts(int64_t m, const unsigned* __restrict__ skeys, int64_t nkeys,
                                                         int* __restrict__ start) {
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += (int64_t)gridDim.x * blockDim.x) {
        unsigned k = skeys[j];
        int64_t prev = j == 0 ? -1 : (int64_t)skeys[j - 1];
        if ((int64_t)k != prev)
            for (int64_t c = prev + 1; c <= (int64_t)k; ++c) start[c] = (int)j;
        if (j == m - 1)
            for (int64_t c = (int64_t)k + 1; c <= nkeys; ++c) start[c] = (int)m;
    }
}

// sum over the cells of (points in the cell)^2 = n * (the occupancy of the cell an average POINT sits in): 3 for a uniform cloud at two
// points per cell, hundreds when most points sit in clumps the box-volume rule does not see
__global__ __launch_bounds__(256) void k_icp_occupancy(int64_t ncells, const int* __restrict__ cellStart, unsigned long long* __restrict__ out) {
    unsigned long long acc = 0;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < ncells; k += (int64_t)gridDim.x * blockDim.x) {
        const unsigned long long len = (unsigned long long)(cellStart[k + 1] - cellStart[k]);
        acc += len * len;
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0 && acc) atomicAdd(out, acc);
}

__global__ __launch_bounds__(256) void k_icp_gather_target(int64_t n, const unsigned* __restrict__ order,
                                                           const float* __restrict__ xyz, const double* __restrict__ nrm,
                                                           float4* __restrict__ Tq, double* __restrict__ Tn) {
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) {
        const unsigned i = order[j];
        Tq[j] = make_float4(xyz[3 * (int64_t)i], xyz[3 * (int64_t)i + 1], xyz[3 * (int64_t)i + 2], __uint_as_float(i));
        if (nrm) {
            Tn[3 * j] = nrm[3 * (int64_t)i]; Tn[3 * j + 1] = nrm[3 * (int64_t)i + 1]; Tn[3 * j + 2] = nrm[3 * (int64_t)i + 2];
        }
    }
}

__global__ __launch_bounds__(256) void k_icp_gather_source(int64_t n, const unsigned* __restrict__ order, const float* __restrict__ xyz,
                                                           float* __restrict__ out) {
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = order[j];
        out[3 * j] = xyz[3 * i]; out[3 * j + 1] = xyz[3 * i + 1]; out[3 * j + 2] = xyz[3 * i + 2];
    }
}

__global__ __launch_bounds__(256) void k_icp_gather_cov(int64_t n, const unsigned* __restrict__ order, const double* __restrict__ in,
                                                        double* __restrict__ out) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < 6 * n; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t j = t / 6;
        const int k = (int)(t - 6 * j);
        out[t] = in[6 * (order ? (int64_t)order[j] : j) + k];
    }
}

struct Xform { double m[12]; };   // rows 0..2 of the 4x4

// Colored ICP operands (KIND 3): target intensity / colour gradient (cell-sorted), source intensity (source order)
struct ColorArgs {
    const double* t_int;
    const double* t_grad;
    const double* s_int;
    double sqrt_lg, sqrt_lp;       // sqrt(lambda_geometric), sqrt(1 - lambda_geometric)
};

// Exact nearest target neighbour of p (float64) by an expanding ring search over the uniform grid.
// Ring r = the cells at Chebyshev distance r from p's (clamped) cell.  Every point in a cell beyond
// ring r is at Euclidean distance >= r*c from p, so the search stops as soon as the best squared
// distance is below (r*c)^2, and never needs to go past ring `rings` = ceil(max_corr / c) because
// farther points cannot be accepted (d^2 < max_corr^2) anyway.  Returns the SORTED position (or -1)
// and d^2; ties go to the lowest input index, as the CPU oracle does.
__device__ __forceinline__ void icp_scan_span(const int* __restrict__ cellStart, const float4* __restrict__ Tq, int first, int last,
                                              double px, double py, double pz, double& bd, int& best, unsigned& best_i) {
    const int s = cellStart[first], e = cellStart[last + 1];
    for (int j = s; j < e; ++j) {
        const float4 q = Tq[j];
        const double dx = px - (double)q.x, dy = py - (double)q.y, dz = pz - (double)q.z;
        const double d2 = dx * dx + dy * dy + dz * dz;
        const unsigned qi = __float_as_uint(q.w);
        if (d2 < bd || (d2 == bd && qi < best_i)) { bd = d2; best = j; best_i = qi; }
    }
}

// Distance from coordinate v to the slab of cell k along one axis (0 inside), shrunk by a relative 1e-9 so that the
// float64 rounding of the cell classification can never make a bound too large.
// The first and the last cell of an axis also hold every point CLAMPED in from beyond the grid's box (the box is a robust one when
// the cloud has far outliers, gsr_icp_set_target): they are half-infinite slabs.
__device__ __forceinline__ double icp_slab_dist(double v, double o, double c, int k, double eps, int g) {
    const double lo = o + (double)k * c, hi = o + (double)(k + 1) * c;
    const double d = ((v < lo && k > 0) ? lo - v : ((v > hi && k < g - 1) ? v - hi : 0.0)) * 0.999999999 - eps;
    return d > 0.0 ? d : 0.0;
}

__device__ __forceinline__ void icp_consider(const float4 q, int j, double px, double py, double pz, double& bd, int& best, unsigned& best_i) {
    const double dx = px - (double)q.x, dy = py - (double)q.y, dz = pz - (double)q.z;
    const double d2 = dx * dx + dy * dy + dz * dz;
    const unsigned qi = __float_as_uint(q.w);
    if (d2 < bd || (d2 == bd && qi < best_i)) { bd = d2; best = j; best_i = qi; }
}

// BLOCK: rings 0 and 1 as ONE block of 3 x 3 rows of three cells, for the levels whose correspondence distance spans two or
// more cells (the coarse levels of a multiscale run): there the neighbour is rarely in the query's own cell, and the ring
// loop is a chain of dependent loads -- a row's cell bounds, then its points one by one: ~65 round trips to L2 for ~55
// candidates -- with nothing else for the thread to do.  The block loads its 18 cell bounds together and the points four
// at a time (index clamped to the row's last point: a point seen twice changes nothing); any scan order finds the same
// minimum of (d^2, input index), and the ring loop continues at ring 2 when the bound of ring 1 does not close.
// Coarse-to-fine pair of the bench: -11 % at 185 k points, -13 % at 1.67 M.  On a converged fine level (max_corr ~ one
// cell) the neighbour sits in the own cell and the block costs 2.3x: BLOCK = false keeps the ring loop from ring 0.
// Measured and rejected on the coarse levels (185 k - 556 k source points, where the chip is not full): four lanes per query
// (each a z-plane of the block; 1024-thread workgroups), nine rows' loads in flight per lane, and the wave staging the joined
// candidate runs of its 64 neighbouring queries in LDS -- all slower (what they add in registers / LDS costs a
// round of workgroups on the chip, and the kernel's time is its slowest wave, not its average one).
#define ICP_BLOCK_CROWDED 24     // points in a row of the 27-cell block from which on the row is tested against the bound before it is scanned
#ifndef ICP_ROW_BATCH
#define ICP_ROW_BATCH 4
#endif
template <int BLOCK>
__device__ __forceinline__ int icp_nearest(const IcpGrid& g, const int* __restrict__ cellStart,
                                           const float4* __restrict__ Tq, double px, double py, double pz, double& best_d2,
                                           double bound2 = 1.0 / 0.0) {
    // bound2: only a neighbour with d^2 < bound2 is of use to the caller (max_corr^2): the search starts from that bound instead of
    // +inf, so a query with nothing in reach prunes its rings like one that has found something (the outliers of a coarse level
    // walked all (2 rings + 1)^3 cells, one lane holding up its wave)
    int best = -1;
    unsigned best_i = 0xffffffffu;
    double bd = bound2;
    if (!(px == px) || !(py == py) || !(pz == pz)) { best_d2 = 1.0 / 0.0; return -1; }
    {   // a query farther from the box of all target points than the caller's bound has nothing to find: no ring is walked (with a cell of two
        // points and a correspondence distance of many cells -- the reference's default max_correspondence is 5 scene units,
        // registration_parameters.py:9 -- a floater of the source would walk (2 rings + 1)^3 cells and hold up its wave)
        const double ex = fmax(fmax(g.bx0 - px, px - g.bx1), 0.0), ey = fmax(fmax(g.by0 - py, py - g.by1), 0.0), ez = fmax(fmax(g.bz0 - pz, pz - g.bz1), 0.0);
        if ((ex * ex + ey * ey + ez * ez) * 0.999999999 >= bound2) { best_d2 = bound2; return -1; }
    }
    const int cx = icp_cell(px, g.ox, g.inv_c, g.gx), cy = icp_cell(py, g.oy, g.inv_c, g.gy), cz = icp_cell(pz, g.oz, g.inv_c, g.gz);
    // absolute slack of every geometric bound: ~500 ulp of the largest coordinate involved
    const double eps = 1e-13 * (fabs(px) + fabs(py) + fabs(pz) + fabs(g.ox) + fabs(g.oy) + fabs(g.oz) + g.c * (double)(g.gx + g.gy + g.gz));
    int r0 = 0;
    if (BLOCK == 1 && g.rings >= 1) {
        int rs[9], re[9];
        const int xa = cx - 1 > 0 ? cx - 1 : 0, xb = cx + 1 < g.gx - 1 ? cx + 1 : g.gx - 1;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int z = cz + t / 3 - 1, y = cy + t % 3 - 1;
            const bool ok = z >= 0 && z < g.gz && y >= 0 && y < g.gy;
            const int rowbase = ok ? (z * g.gy + y) * g.gx : 0;
            rs[t] = ok ? cellStart[rowbase + xa] : 0;
            re[t] = ok ? cellStart[rowbase + xb + 1] : 0;
        }
        {
        // The query's own row first: its points give the bound a CROWDED neighbour row is tested against before it is scanned.  On
        // a cloud at two points per cell no row is crowded and nothing is tested (the unconditional batched loads are what made the
        // block fast); in a clump -- 45 points per cell on the coarse level of the clustered scene -- a query scanned 1 200 points
        // of which its own row's 135 already held the neighbour.  A row is skipped only when it is STRICTLY farther than the best
        // so far (a point at exactly the best distance could still win the tie on its index).
        constexpr int order9[9] = {4, 1, 3, 5, 7, 0, 2, 6, 8};
#pragma unroll
        for (int tt = 0; tt < 9; ++tt) {
            const int t = order9[tt];
            if (t != 4 && re[t] - rs[t] > ICP_BLOCK_CROWDED) {
                const double dzb = icp_slab_dist(pz, g.oz, g.c, cz + t / 3 - 1, eps, g.gz), dyb = icp_slab_dist(py, g.oy, g.c, cy + t % 3 - 1, eps, g.gy);
                if (dzb * dzb + dyb * dyb > bd) continue;
            }
            // ICP_ROW_BATCH points per round trip; the slots behind the row's end re-read its last point.  (A row of three cells holds
            // ~6 points at two per cell, so four per batch are two dependent round trips per row -- and yet 2, 3 and 4 per batch measure
            // the same, 6 per batch +3 % and 8 per batch +15-35 % on the coarse levels: the registers cost more waves than the shorter
            // chain buys, profiles/archive/r04q_icp_row_batch_ab.txt.)
            for (int j = rs[t]; j < re[t]; j += ICP_ROW_BATCH) {
                const int l = re[t] - 1;
                int jj[ICP_ROW_BATCH];
                float4 qq[ICP_ROW_BATCH];
#pragma unroll
                for (int b = 0; b < ICP_ROW_BATCH; ++b) { jj[b] = j + b < l ? j + b : l; qq[b] = Tq[jj[b]]; }
#pragma unroll
                for (int b = 0; b < ICP_ROW_BATCH; ++b) icp_consider(qq[b], jj[b], px, py, pz, bd, best, best_i);
            }
        }
        }
    }
    if (BLOCK && g.rings >= 1) {
        double reach = 1.0 / 0.0;
        if (cx - 1 > 0) reach = fmin(reach, px - (g.ox + (double)(cx - 1) * g.c));
        if (cx + 1 < g.gx - 1) reach = fmin(reach, (g.ox + (double)(cx + 2) * g.c) - px);
        if (cy - 1 > 0) reach = fmin(reach, py - (g.oy + (double)(cy - 1) * g.c));
        if (cy + 1 < g.gy - 1) reach = fmin(reach, (g.oy + (double)(cy + 2) * g.c) - py);
        if (cz - 1 > 0) reach = fmin(reach, pz - (g.oz + (double)(cz - 1) * g.c));
        if (cz + 1 < g.gz - 1) reach = fmin(reach, (g.oz + (double)(cz + 2) * g.c) - pz);
        reach = reach * 0.999999999 - eps;
        reach = reach > 0.0 ? reach : 0.0;
        if (bd < reach * reach) { best_d2 = bd; return best; }
        r0 = 2;
    }
    for (int r = r0; r <= g.rings; ++r) {
        for (int dz = -r; dz <= r; ++dz) {
            const int z = cz + dz;
            if (z < 0 || z >= g.gz) continue;
            const int adz = dz < 0 ? -dz : dz;
            const double dzb = icp_slab_dist(pz, g.oz, g.c, z, eps, g.gz);
            for (int dy = -r; dy <= r; ++dy) {
                const int y = cy + dy;
                if (y < 0 || y >= g.gy) continue;
                const int ady = dy < 0 ? -dy : dy;
                // prune by the best so far: every point of this row is at least (dyb, dzb) away in y / z, and a cell
                // at x-distance dxb adds that.  STRICT comparisons: a candidate at exactly the best distance can still
                // win the tie on its index, so only cells that are strictly farther are skipped.
                const double dyb = icp_slab_dist(py, g.oy, g.c, y, eps, g.gy);
                const double rem = bd - dyb * dyb - dzb * dzb;
                if (rem < 0.0) continue;
                int xlo = 0, xhi = g.gx - 1;
                if (rem < 1e300) {                                // a finite best: clip the x span to |dx| <= sqrt(rem)
                    const double hx = (double)(sqrtf((float)rem) * 1.0001f) + eps + 1e-30;
                    xlo = icp_cell(px - hx, g.ox, g.inv_c, g.gx);
                    xhi = icp_cell(px + hx, g.ox, g.inv_c, g.gx);
                }
                const int rowbase = (z * g.gy + y) * g.gx;
                if (adz == r || ady == r) {                       // a face row of the ring: the whole x span
                    int xa = cx - r > 0 ? cx - r : 0, xb = cx + r < g.gx - 1 ? cx + r : g.gx - 1;
                    xa = xa > xlo ? xa : xlo;
                    xb = xb < xhi ? xb : xhi;
                    if (xa <= xb) icp_scan_span(cellStart, Tq, rowbase + xa, rowbase + xb, px, py, pz, bd, best, best_i);
                } else {                                          // interior row: only the two end cells
                    if (cx - r >= 0 && cx - r >= xlo) icp_scan_span(cellStart, Tq, rowbase + cx - r, rowbase + cx - r, px, py, pz, bd, best, best_i);
                    if (cx + r < g.gx && cx + r <= xhi) icp_scan_span(cellStart, Tq, rowbase + cx + r, rowbase + cx + r, px, py, pz, bd, best, best_i);
                }
            }
        }
        // every cell within Chebyshev distance r has been seen: an unseen point lies beyond one of the faces of that
        // block (faces on the grid boundary have nothing behind them), i.e. at least `reach` away
        double reach = 1.0 / 0.0;
        if (cx - r > 0) reach = fmin(reach, px - (g.ox + (double)(cx - r) * g.c));
        if (cx + r < g.gx - 1) reach = fmin(reach, (g.ox + (double)(cx + r + 1) * g.c) - px);
        if (cy - r > 0) reach = fmin(reach, py - (g.oy + (double)(cy - r) * g.c));
        if (cy + r < g.gy - 1) reach = fmin(reach, (g.oy + (double)(cy + r + 1) * g.c) - py);
        if (cz - r > 0) reach = fmin(reach, pz - (g.oz + (double)(cz - r) * g.c));
        if (cz + r < g.gz - 1) reach = fmin(reach, (g.oz + (double)(cz + r + 1) * g.c) - pz);
        reach = reach * 0.999999999 - eps;
        reach = reach > 0.0 ? reach : 0.0;
        if (bd < reach * reach) break;
    }
    best_d2 = bd;
    return best;
}

// Which source points a workgroup takes.  The source is sorted by the target's cells, so a CONTIGUOUS run of it queries a compact
// region of the target.  Workgroups go to the 8 XCDs round robin: logical block L = (bid % 8) * ceil(nb / 8) + bid / 8 gives every
// XCD one contiguous eighth of the source, and with it an eighth of the target (+ halo) for its own L2 -- block b taking the
// points b * 256 + k * stride made every XCD pull the WHOLE target through its 4 MB (L2 hit rate 46 % on the coarse levels).
// Launch 8 * ceil(nb / 8) workgroups; padding workgroups get nothing.  The partial sums are stored by logical block.
struct IcpRange { int row; int64_t lo, hi; };
__device__ __forceinline__ IcpRange icp_block_range(int64_t ns, int nb, bool remap) {
    int L = (int)blockIdx.x;
    if (remap) {
        const int chunk = (nb + 7) >> 3;
        L = ((int)blockIdx.x & 7) * chunk + ((int)blockIdx.x >> 3);
    }
    const int64_t ppb = ((ns + nb - 1) / nb + 255) / 256 * 256;       // points per logical block, whole waves of a 256-thread block
    IcpRange r;
    r.row = L < nb ? L : -1;
    r.lo = (int64_t)L * ppb;
    r.hi = r.lo + ppb < ns ? r.lo + ppb : ns;
    return r;
}

struct IcpState;
__device__ __forceinline__ bool icp_state_done(const IcpState* st);
__device__ __forceinline__ void icp_state_T(const IcpState* st, double T[12]);
template <int THREADS, bool SC1 = false> __device__ __forceinline__ void icp_fold_partials(int nblocks, const double* partials, const double* __restrict__ reduced, double (*s_x)[32]);
__device__ __forceinline__ void icp_step_solve(const double* acc32, const IcpState& s_st, IcpState* st);

// nn_j[i] = sorted target position of the accepted nearest neighbour of source point i, or -1.  One thread per source
// point: a search kernel with few registers (56 VGPRs, 8 waves per SIMD) in front of a streaming accumulate kernel.
// (Measured and rejected: eight lanes per point with shuffle-combined partial searches -- 1.5x slower.)
template <bool FROM_STATE, int BLOCK>
__global__ __launch_bounds__(256) void k_icp_nn(int64_t ns, const float* __restrict__ src, Xform X, const IcpState* __restrict__ st,
                                                IcpGrid g, const int* __restrict__ cellStart, const float4* __restrict__ Tq,
                                                double max_corr2, int* __restrict__ nn_j, int nb_logical) {
    double T[12];
    if (FROM_STATE) {
        if (icp_state_done(st)) return;
        icp_state_T(st, T);
    } else {
#pragma unroll
        for (int i = 0; i < 12; ++i) T[i] = X.m[i];
    }
    const IcpRange rg = icp_block_range(ns, nb_logical < 0 ? -nb_logical : nb_logical, nb_logical > 0);
    if (rg.row < 0) return;
    for (int64_t i = rg.lo + threadIdx.x; i < rg.hi; i += blockDim.x) {
        const double x = (double)src[3 * i], y = (double)src[3 * i + 1], z = (double)src[3 * i + 2];
        const double px = T[0] * x + T[1] * y + T[2] * z + T[3];
        const double py = T[4] * x + T[5] * y + T[6] * z + T[7];
        const double pz = T[8] * x + T[9] * y + T[10] * z + T[11];
        double d2;
        const int j = icp_nearest<BLOCK>(g, cellStart, Tq, px, py, pz, d2, max_corr2);
        nn_j[i] = (j >= 0 && d2 < max_corr2) ? j : -1;
    }
}

// The search of a FINE level (rings == 1: the correspondence distance is at most one cell, so the 27 cells around the query's hold every
// neighbour that can be accepted) over an LDS-staged tile.  k_icp_nn walks the grid per thread: 27 cell spans, two dependent table
// look-ups and a chain of point loads each, every lane with its own trip counts -- 24 % of the lanes at work, ~3 000 issue slots per
// query (profiles/archive/r04t_pmc_icp.json).  The source is sorted by the target's cells, so the 256 queries of a workgroup sit in a run of
// consecutive cells: ONE box of cells [xlo, xhi] x [ylo, yhi] x [zlo, zhi] around their cells (+ 1 ring) holds all their candidates.
// Its rows are contiguous runs of the cell-sorted target: the workgroup copies them into LDS with coalesced loads (and the rows' cell
// table with them), and every lane then scans its 3 x 3 spans of three cells from LDS.  Same candidates, same float64 distance, same
// tie rule (lowest input index): the same neighbour as icp_nearest<0>, whatever the order.  A workgroup whose box does not fit (more
// than ICP_TILE_ROWS rows or ICP_TILE_PTS points: queries far apart -- the ends of a level's source, a source that moved by cells)
// searches the old way.
#define ICP_TILE_ROWS 16
#define ICP_TILE_PTS 3840
#define ICP_TILE_XW 256
#define ICP_TILE_LDS (ICP_TILE_PTS * 16 + ICP_TILE_ROWS * ICP_TILE_XW * 4)
template <bool FROM_STATE>
__global__ __launch_bounds__(256) void k_icp_nn_tile(int64_t ns, const float* __restrict__ src, Xform X, const IcpState* __restrict__ st,
                                                     IcpGrid g, const int* __restrict__ cellStart, const float4* __restrict__ Tq,
                                                     double max_corr2, int* __restrict__ nn_j, int nb_logical) {
    extern __shared__ float4 s_dyn[];
    float4* s_pts = s_dyn;                                             // the staged target points of the box, row after row
    int* s_cs = reinterpret_cast<int*>(s_dyn + ICP_TILE_PTS);          // [rows][xw + 1]: cell starts as indices into s_pts
    __shared__ int s_box[6];                                           // min / max cell of the workgroup's queries
    __shared__ int s_gs[ICP_TILE_ROWS], s_off[ICP_TILE_ROWS + 1];      // a row's first point in Tq / in s_pts
    __shared__ int s_ok;
    double T[12];
    if (FROM_STATE) {
        if (icp_state_done(st)) return;
        icp_state_T(st, T);
    } else {
#pragma unroll
        for (int i = 0; i < 12; ++i) T[i] = X.m[i];
    }
    const IcpRange rg = icp_block_range(ns, nb_logical < 0 ? -nb_logical : nb_logical, nb_logical > 0);
    if (rg.row < 0) return;
    for (int64_t base = rg.lo; base < rg.hi; base += blockDim.x) {
        const int64_t i = base + threadIdx.x;
        const bool live = i < rg.hi;
        double px = 0.0, py = 0.0, pz = 0.0;
        int cx = 0, cy = 0, cz = 0;
        bool finite = false;
        if (live) {
            const double x = (double)src[3 * i], y = (double)src[3 * i + 1], z = (double)src[3 * i + 2];
            px = T[0] * x + T[1] * y + T[2] * z + T[3];
            py = T[4] * x + T[5] * y + T[6] * z + T[7];
            pz = T[8] * x + T[9] * y + T[10] * z + T[11];
            finite = px == px && py == py && pz == pz;
            cx = icp_cell(px, g.ox, g.inv_c, g.gx); cy = icp_cell(py, g.oy, g.inv_c, g.gy); cz = icp_cell(pz, g.oz, g.inv_c, g.gz);
        }
        if (threadIdx.x < 3) { s_box[threadIdx.x] = 0x7fffffff; s_box[3 + threadIdx.x] = -1; }
        __syncthreads();
        if (live && finite) {
            atomicMin(&s_box[0], cx); atomicMin(&s_box[1], cy); atomicMin(&s_box[2], cz);
            atomicMax(&s_box[3], cx); atomicMax(&s_box[4], cy); atomicMax(&s_box[5], cz);
        }
        __syncthreads();
        const int xlo = s_box[0] - 1 > 0 ? s_box[0] - 1 : 0, xhi = s_box[3] + 1 < g.gx - 1 ? s_box[3] + 1 : g.gx - 1;
        const int ylo = s_box[1] - 1 > 0 ? s_box[1] - 1 : 0, yhi = s_box[4] + 1 < g.gy - 1 ? s_box[4] + 1 : g.gy - 1;
        const int zlo = s_box[2] - 1 > 0 ? s_box[2] - 1 : 0, zhi = s_box[5] + 1 < g.gz - 1 ? s_box[5] + 1 : g.gz - 1;
        const int ny = yhi - ylo + 1, nz = zhi - zlo + 1, xw = xhi - xlo + 1;
        const int rows = s_box[3] < 0 ? 0 : ny * nz;                    // (no finite query at all: nothing to stage)
        bool ok = rows > 0 && rows <= ICP_TILE_ROWS && xw + 1 <= ICP_TILE_XW;
        if (ok && (int)threadIdx.x < rows) {
            const int y = ylo + (int)threadIdx.x % ny, z = zlo + (int)threadIdx.x / ny;
            const int rowbase = (z * g.gy + y) * g.gx;
            s_gs[threadIdx.x] = cellStart[rowbase + xlo];
            s_off[threadIdx.x + 1] = cellStart[rowbase + xhi + 1] - s_gs[threadIdx.x];       // (its length for now)
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int tot = 0;
            s_off[0] = 0;
            if (ok) for (int r = 0; r < rows; ++r) { const int len = s_off[r + 1]; s_off[r + 1] = tot + len; tot += len; }
            s_ok = ok && tot <= ICP_TILE_PTS ? 1 : 0;
        }
        __syncthreads();
        ok = s_ok != 0;
        int best = -1;
        double bd = max_corr2;
        if (ok) {
            // the box's points and its cell table into LDS: contiguous runs of Tq / cellStart, consecutive lanes on consecutive addresses
            const int total = s_off[rows];
            for (int t = threadIdx.x; t < total; t += blockDim.x) {
                int r = 0;
                while (t >= s_off[r + 1]) ++r;
                s_pts[t] = Tq[s_gs[r] + (t - s_off[r])];
            }
            for (int t = threadIdx.x; t < rows * (xw + 1); t += blockDim.x) {
                const int r = t / (xw + 1), x = t - r * (xw + 1);
                const int y = ylo + r % ny, z = zlo + r / ny;
                s_cs[t] = cellStart[(z * g.gy + y) * g.gx + xlo + x] - s_gs[r] + s_off[r];
            }
            __syncthreads();
            if (live && finite) {
                unsigned best_i = 0xffffffffu;
                int bk = -1, br = 0;
                const int xa = (cx - 1 > 0 ? cx - 1 : 0) - xlo, xb = (cx + 1 < g.gx - 1 ? cx + 1 : g.gx - 1) - xlo;
#pragma unroll 1
                for (int dz = -1; dz <= 1; ++dz) {
                    const int z = cz + dz;
                    if (z < 0 || z >= g.gz) continue;
#pragma unroll 1
                    for (int dy = -1; dy <= 1; ++dy) {
                        const int y = cy + dy;
                        if (y < 0 || y >= g.gy) continue;
                        const int r = (z - zlo) * ny + (y - ylo);
                        const int s = s_cs[r * (xw + 1) + xa], e = s_cs[r * (xw + 1) + xb + 1];
                        for (int k = s; k < e; ++k) {
                            const float4 q = s_pts[k];
                            const double dx = px - (double)q.x, dyy = py - (double)q.y, dzz = pz - (double)q.z;
                            const double d2 = dx * dx + dyy * dyy + dzz * dzz;
                            const unsigned qi = __float_as_uint(q.w);
                            if (d2 < bd || (d2 == bd && qi < best_i)) { bd = d2; bk = k; br = r; best_i = qi; }
                        }
                    }
                }
                if (bk >= 0) best = s_gs[br] + (bk - s_off[br]);
            }
        } else if (live) {
            double d2;
            best = icp_nearest<0>(g, cellStart, Tq, px, py, pz, d2, max_corr2);
            bd = d2;
        }
        if (live) nn_j[i] = (best >= 0 && bd < max_corr2) ? best : -1;
        __syncthreads();
    }
}

__device__ __forceinline__ double icp_weight(int loss, double k, double r) {   // Open3D RobustKernel.cpp
    switch (loss) {
        case GSR_LOSS_TUKEY: { double t = fmin(1.0, fabs(r) / k); double u = 1.0 - t * t; return u * u; }
        case GSR_LOSS_CAUCHY: { double t = r / k; return 1.0 / (1.0 + t * t); }
        case GSR_LOSS_GM: { double t = k + r * r; return k / (t * t); }
        case GSR_LOSS_HUBER: { double e = fabs(r); return k / fmax(e, k); }
        default: return 1.0;
    }
}

// Symmetric 3x3 eigen-decomposition (cyclic Jacobi, float64): A = V diag(lam) V^T.
__host__ __device__ static void sym_eig3(double a00, double a01, double a02, double a11, double a12, double a22, double V[3][3], double lam[3]) {
    double A[3][3] = {{a00, a01, a02}, {a01, a11, a12}, {a02, a12, a22}};
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) V[i][j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; ++sweep) {
        const double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
        const double dg = A[0][0] * A[0][0] + A[1][1] * A[1][1] + A[2][2] * A[2][2];
        if (!(off > 1e-40 * dg)) break;
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int q = p + 1; q < 3; ++q) {
                if (A[p][q] == 0.0) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
#pragma unroll
                for (int k = 0; k < 3; ++k) { const double x = A[k][p], y = A[k][q]; A[k][p] = c * x - sn * y; A[k][q] = sn * x + c * y; }
#pragma unroll
                for (int k = 0; k < 3; ++k) { const double x = A[p][k], y = A[q][k]; A[p][k] = c * x - sn * y; A[q][k] = sn * x + c * y; }
#pragma unroll
                for (int k = 0; k < 3; ++k) { const double x = V[k][p], y = V[k][q]; V[k][p] = c * x - sn * y; V[k][q] = sn * x + c * y; }
            }
    }
    lam[0] = A[0][0]; lam[1] = A[1][1]; lam[2] = A[2][2];
}

// Generalized ICP (Open3D GeneralizedICP.cpp, ComputeTransformation): three residual rows of one matched pair into
// acc[2..29].  cs6 = the source covariance in the ORIGINAL frame (rotated here by R = T[0:3,0:3]), ct6 = target's.
template <int NACC>
__device__ __forceinline__ void icp_gicp_rows(double (&acc)[NACC], const double* T, double px, double py, double pz, double qx, double qy,
                                              double qz, const double* __restrict__ cs6, const double* __restrict__ ct6, int loss, double kparam) {
    const double c0[3][3] = {{cs6[0], cs6[1], cs6[2]}, {cs6[1], cs6[3], cs6[4]}, {cs6[2], cs6[4], cs6[5]}};
    const double R[3][3] = {{T[0], T[1], T[2]}, {T[4], T[5], T[6]}, {T[8], T[9], T[10]}};
    double RC[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) RC[a][b] = R[a][0] * c0[0][b] + R[a][1] * c0[1][b] + R[a][2] * c0[2][b];
    double M[6];                                   // Ct + R Cs R^T, upper triangle
    {
        int t = 0;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = a; b < 3; ++b) { M[t] = ct6[t] + (RC[a][0] * R[b][0] + RC[a][1] * R[b][1] + RC[a][2] * R[b][2]); ++t; }
    }
    double V[3][3], lam[3], W[3][3];
    sym_eig3(M[0], M[1], M[2], M[3], M[4], M[5], V, lam);
    const double is[3] = {1.0 / sqrt(lam[0]), 1.0 / sqrt(lam[1]), 1.0 / sqrt(lam[2])};
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) W[a][b] = V[a][0] * is[0] * V[b][0] + V[a][1] * is[1] * V[b][1] + V[a][2] * is[2] * V[b][2];
    const double d[3] = {px - qx, py - qy, pz - qz};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        // J_i = [ (W * -skew(p)).row(i), W.row(i) ],  -skew(p) = [[0, pz, -py], [-pz, 0, px], [py, -px, 0]]
        const double J[6] = {W[i][1] * -pz + W[i][2] * py, W[i][0] * pz + W[i][2] * -px, W[i][0] * -py + W[i][1] * px,
                             W[i][0], W[i][1], W[i][2]};
        const double r = W[i][0] * d[0] + W[i][1] * d[1] + W[i][2] * d[2];
        const double w = icp_weight(loss, kparam, r);
        int t = 2;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b = a; b < 6; ++b) acc[t++] += J[a] * w * J[b];
#pragma unroll
        for (int a = 0; a < 6; ++a) acc[23 + a] += J[a] * w * r;
        acc[29] += r * r;
    }
}

// Colored ICP (Open3D ColoredICP.cpp, ComputeTransformation): the geometric and the photometric row of one pair.
template <int NACC>
__device__ __forceinline__ void icp_colored_rows(double (&acc)[NACC], const ColorArgs& ca, double px, double py, double pz, double qx, double qy,
                                                 double qz, double nx, double ny, double nz, int64_t i, int64_t j, int loss, double kparam) {
    const double dn = (px - qx) * nx + (py - qy) * ny + (pz - qz) * nz;
    double J[2][6], r[2];
    J[0][0] = ca.sqrt_lg * (py * nz - pz * ny); J[0][1] = ca.sqrt_lg * (pz * nx - px * nz); J[0][2] = ca.sqrt_lg * (px * ny - py * nx);
    J[0][3] = ca.sqrt_lg * nx; J[0][4] = ca.sqrt_lg * ny; J[0][5] = ca.sqrt_lg * nz;
    r[0] = ca.sqrt_lg * dn;
    const double pjx = px - dn * nx - qx, pjy = py - dn * ny - qy, pjz = pz - dn * nz - qz;      // vs_proj - vt
    const double d0 = ca.t_grad[3 * j], d1 = ca.t_grad[3 * j + 1], d2 = ca.t_grad[3 * j + 2];
    const double is0 = d0 * pjx + d1 * pjy + d2 * pjz + ca.t_int[j];
    const double dd = d0 * nx + d1 * ny + d2 * nz;
    const double m0 = -(d0 - dd * nx), m1 = -(d1 - dd * ny), m2 = -(d2 - dd * nz);               // -dit^T (I - n n^T)
    J[1][0] = ca.sqrt_lp * (py * m2 - pz * m1); J[1][1] = ca.sqrt_lp * (pz * m0 - px * m2); J[1][2] = ca.sqrt_lp * (px * m1 - py * m0);
    J[1][3] = ca.sqrt_lp * m0; J[1][4] = ca.sqrt_lp * m1; J[1][5] = ca.sqrt_lp * m2;
    r[1] = ca.sqrt_lp * (ca.s_int[i] - is0);
#pragma unroll
    for (int row = 0; row < 2; ++row) {
        const double w = icp_weight(loss, kparam, r[row]);
        int t = 2;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b = a; b < 6; ++b) acc[t++] += J[row][a] * w * J[row][b];
#pragma unroll
        for (int a = 0; a < 6; ++a) acc[23 + a] += J[row][a] * w * r[row];
        acc[29] += r[row] * r[row];
    }
}

// Colour gradient of every target point (Open3D InitializePointCloudForColoredICP): its ICP_KNN nearest neighbours
// (itself first) within `radius`, ordered by (distance, input index); with >= 4 of them a 3x3 least-squares fit of the
// intensity differences over the tangent-plane projections, plus the orthogonality row (nn-1) n.  One thread per
// point; the k-best list lives in per-thread scratch.  Same expanding-ring search geometry as icp_nearest.
#define ICP_KNN 30
__global__ __launch_bounds__(256) void k_icp_color_gradient(int64_t nt, IcpGrid g, const int* __restrict__ cellStart, const float4* __restrict__ Tq,
                                                            const double* __restrict__ Tn, const double* __restrict__ t_int, double radius,
                                                            double* __restrict__ t_grad) {
    for (int64_t jq = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; jq < nt; jq += (int64_t)gridDim.x * blockDim.x) {
        const float4 qv = Tq[jq];
        const double px = (double)qv.x, py = (double)qv.y, pz = (double)qv.z;
        t_grad[3 * jq] = 0.0; t_grad[3 * jq + 1] = 0.0; t_grad[3 * jq + 2] = 0.0;
        if (!(px == px) || !(py == py) || !(pz == pz)) continue;
        double kd[ICP_KNN];
        int kj[ICP_KNN];
        unsigned ki[ICP_KNN];
        int cnt = 0;
        const double r2 = radius * radius;
        const int cx = icp_cell(px, g.ox, g.inv_c, g.gx), cy = icp_cell(py, g.oy, g.inv_c, g.gy), cz = icp_cell(pz, g.oz, g.inv_c, g.gz);
        const double eps = 1e-13 * (fabs(px) + fabs(py) + fabs(pz) + fabs(g.ox) + fabs(g.oy) + fabs(g.oz) + g.c * (double)(g.gx + g.gy + g.gz));
        const int rmax = (int)ceil(radius / g.c) + 1;
        for (int r = 0; r <= rmax; ++r) {
            for (int dz = -r; dz <= r; ++dz) {
                const int z = cz + dz;
                if (z < 0 || z >= g.gz) continue;
                const int adz = dz < 0 ? -dz : dz;
                for (int dy = -r; dy <= r; ++dy) {
                    const int y = cy + dy;
                    if (y < 0 || y >= g.gy) continue;
                    const int ady = dy < 0 ? -dy : dy;
                    const int rowbase = (z * g.gy + y) * g.gx;
                    int spans[2][2];
                    int nsp = 0;
                    if (adz == r || ady == r) {
                        const int xa = cx - r > 0 ? cx - r : 0, xb = cx + r < g.gx - 1 ? cx + r : g.gx - 1;
                        if (xa <= xb) { spans[0][0] = rowbase + xa; spans[0][1] = rowbase + xb; nsp = 1; }
                    } else {
                        if (cx - r >= 0) { spans[nsp][0] = spans[nsp][1] = rowbase + cx - r; ++nsp; }
                        if (cx + r < g.gx) { spans[nsp][0] = spans[nsp][1] = rowbase + cx + r; ++nsp; }
                    }
                    for (int sidx = 0; sidx < nsp; ++sidx) {
                        const int s0 = cellStart[spans[sidx][0]], e0 = cellStart[spans[sidx][1] + 1];
                        for (int j = s0; j < e0; ++j) {
                            const float4 q = Tq[j];
                            const double dx = px - (double)q.x, dy2 = py - (double)q.y, dz2 = pz - (double)q.z;
                            const double d2 = dx * dx + dy2 * dy2 + dz2 * dz2;
                            const unsigned qi = __float_as_uint(q.w);
                            if (!(d2 < r2)) continue;                        // the hybrid search keeps d^2 < radius^2
                            if (cnt == ICP_KNN && !(d2 < kd[cnt - 1] || (d2 == kd[cnt - 1] && qi < ki[cnt - 1]))) continue;
                            int pos = cnt < ICP_KNN ? cnt : ICP_KNN - 1;     // insertion into the list sorted by (d2, index)
                            while (pos > 0 && (d2 < kd[pos - 1] || (d2 == kd[pos - 1] && qi < ki[pos - 1]))) {
                                kd[pos] = kd[pos - 1]; kj[pos] = kj[pos - 1]; ki[pos] = ki[pos - 1];
                                --pos;
                            }
                            kd[pos] = d2; kj[pos] = j; ki[pos] = qi;
                            if (cnt < ICP_KNN) ++cnt;
                        }
                    }
                }
            }
            // an unseen point lies at least `reach` away (see icp_nearest_from); done when the list is full and its
            // worst entry is strictly closer, or when the radius is covered
            double reach = 1.0 / 0.0;
            if (cx - r > 0) reach = fmin(reach, px - (g.ox + (double)(cx - r) * g.c));
            if (cx + r < g.gx - 1) reach = fmin(reach, (g.ox + (double)(cx + r + 1) * g.c) - px);
            if (cy - r > 0) reach = fmin(reach, py - (g.oy + (double)(cy - r) * g.c));
            if (cy + r < g.gy - 1) reach = fmin(reach, (g.oy + (double)(cy + r + 1) * g.c) - py);
            if (cz - r > 0) reach = fmin(reach, pz - (g.oz + (double)(cz - r) * g.c));
            if (cz + r < g.gz - 1) reach = fmin(reach, (g.oz + (double)(cz + r + 1) * g.c) - pz);
            reach = reach * 0.999999999 - eps;
            if (reach > 0.0 && ((cnt == ICP_KNN && kd[cnt - 1] < reach * reach) || r2 < reach * reach)) break;
        }
        if (cnt < 4) continue;
        const double nx = Tn[3 * jq], ny = Tn[3 * jq + 1], nz = Tn[3 * jq + 2];
        const double it = t_int[jq];
        double A[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, b[3] = {0, 0, 0};
        for (int i = 1; i < cnt; ++i) {
            const float4 q = Tq[kj[i]];
            const double vx = (double)q.x - px, vy = (double)q.y - py, vz = (double)q.z - pz;
            const double dn = vx * nx + vy * ny + vz * nz;
            const double row[3] = {(double)q.x - dn * nx - px, (double)q.y - dn * ny - py, (double)q.z - dn * nz - pz};
            const double rhs = t_int[kj[i]] - it;
            for (int p = 0; p < 3; ++p) { for (int c2 = 0; c2 < 3; ++c2) A[p][c2] += row[p] * row[c2]; b[p] += row[p] * rhs; }
        }
        const double f = (double)(cnt - 1);
        const double row[3] = {f * nx, f * ny, f * nz};
        for (int p = 0; p < 3; ++p) for (int c2 = 0; c2 < 3; ++c2) A[p][c2] += row[p] * row[c2];
        // 3x3 solve: Gaussian elimination with partial pivoting
        double M[3][4] = {{A[0][0], A[0][1], A[0][2], b[0]}, {A[1][0], A[1][1], A[1][2], b[1]}, {A[2][0], A[2][1], A[2][2], b[2]}};
        for (int c2 = 0; c2 < 3; ++c2) {
            int piv = c2;
            for (int rr = c2 + 1; rr < 3; ++rr) if (fabs(M[rr][c2]) > fabs(M[piv][c2])) piv = rr;
            for (int k2 = 0; k2 < 4; ++k2) { const double tmp = M[c2][k2]; M[c2][k2] = M[piv][k2]; M[piv][k2] = tmp; }
            for (int rr = c2 + 1; rr < 3; ++rr) {
                const double fct = M[rr][c2] / M[c2][c2];
                for (int k2 = c2; k2 < 4; ++k2) M[rr][k2] -= fct * M[c2][k2];
            }
        }
        double x[3];
        for (int i = 2; i >= 0; --i) {
            double v = M[i][3];
            for (int k2 = i + 1; k2 < 3; ++k2) v -= M[i][k2] * x[k2];
            x[i] = v / M[i][i];
        }
        t_grad[3 * jq] = x[0]; t_grad[3 * jq + 1] = x[1]; t_grad[3 * jq + 2] = x[2];
    }
}
// intensity = mean of the three colour channels (float64), gathered into `order` (NULL = identity)
__global__ __launch_bounds__(256) void k_icp_intensity(int64_t n, const unsigned* __restrict__ order, const double* __restrict__ rgb,
                                                       double* __restrict__ out) {
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = order ? (int64_t)order[j] : j;
        out[j] = (rgb[3 * i] + rgb[3 * i + 1] + rgb[3 * i + 2]) / 3.0;
    }
}

// Sum of a double over the 64 lanes (every lane gets it) in a fixed order, without the LDS crossbar: four DPP row rotations
// and the two permlane swaps, on the two dwords of the value.  (30 accumulators x 6 steps of __shfl_xor were 360
// ds_bpermute per wave, 24 LDS-pipe cycles each: a third of the accumulate kernel on a 185 k-point level.)
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, false);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}
template <bool HALF>        // HALF: lanes l and l ^ 32, else l and l ^ 16
__device__ __forceinline__ double swap_sum_d(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const unsigned lo = (unsigned)b, hi = (unsigned)(b >> 32);
    const auto rl = HALF ? __builtin_amdgcn_permlane32_swap(lo, lo, false, false) : __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto rh = HALF ? __builtin_amdgcn_permlane32_swap(hi, hi, false, false) : __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    const double a = __builtin_bit_cast(double, ((long long)(unsigned)rh[0] << 32) | (unsigned)rl[0]);
    const double c = __builtin_bit_cast(double, ((long long)(unsigned)rh[1] << 32) | (unsigned)rl[1]);
    return a + c;
}
__device__ __forceinline__ double wave_sum_d(double v) {
    v += dpp_d<0x128>(v);       // row_ror:8
    v += dpp_d<0x124>(v);       // row_ror:4
    v += dpp_d<0x122>(v);       // row_ror:2
    v += dpp_d<0x121>(v);       // row_ror:1
    v = swap_sum_d<false>(v);
    v = swap_sum_d<true>(v);
    return v;
}

// WT: the partials leave with write-through (sc1) stores -- what a workgroup that hands them to another workgroup of the SAME
// launch must use (k_icp_accumulate_dev's fused step); plain stores otherwise (the reader is a later kernel)
template <int NACC, bool WT = false>
__device__ __forceinline__ void block_reduce_store(double (&acc)[NACC], double* __restrict__ partials, int row = -1) {
    if (row < 0) row = (int)blockIdx.x;
    __shared__ double s_red[4][NACC];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NACC; ++k) {
        const double v = wave_sum_d(acc[k]);
        if (lane == 0) s_red[wv][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < NACC) {
        const int k = threadIdx.x;
        const double v = ((s_red[0][k] + s_red[1][k]) + s_red[2][k]) + s_red[3][k];
        if (WT) __hip_atomic_store(partials + (int64_t)row * GSR_ICP_ACC_LEN + k, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else partials[(int64_t)row * GSR_ICP_ACC_LEN + k] = v;
    }
}

// KIND 0: point-to-point sums (17 values); KIND 1: point-to-plane normal equations (30 values)
template <int KIND>
__global__ __launch_bounds__(256) void k_icp_accumulate(int64_t ns, const float* __restrict__ src, Xform T, IcpGrid g,
                                                        const int* __restrict__ cellStart, const int* __restrict__ nn_j,
                                                        const float4* __restrict__ Tq, const double* __restrict__ Tn,
                                                        const double* __restrict__ Sc, ColorArgs ca, double max_corr2,
                                                        int loss, double kparam, double* __restrict__ partials) {
    constexpr int NACC = KIND == 0 ? 17 : 30;
    double acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k) acc[k] = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ns; i += (int64_t)gridDim.x * blockDim.x) {
        const double x = (double)src[3 * i], y = (double)src[3 * i + 1], z = (double)src[3 * i + 2];
        const double px = T.m[0] * x + T.m[1] * y + T.m[2] * z + T.m[3];
        const double py = T.m[4] * x + T.m[5] * y + T.m[6] * z + T.m[7];
        const double pz = T.m[8] * x + T.m[9] * y + T.m[10] * z + T.m[11];
        int j;
        if (nn_j) {                                 // optional two-kernel form: neighbours from k_icp_nn (accepted or -1)
            j = nn_j[i];
            if (j < 0) continue;
        } else {
            double dd;
            j = icp_nearest<0>(g, cellStart, Tq, px, py, pz, dd, max_corr2);
            if (j < 0 || !(dd < max_corr2)) continue;
        }
        const float4 q = Tq[j];
        const double qx = (double)q.x, qy = (double)q.y, qz = (double)q.z;
        const double ddx = px - qx, ddy = py - qy, ddz = pz - qz;
        const double d2 = ddx * ddx + ddy * ddy + ddz * ddz;
        acc[0] += 1.0;
        acc[1] += d2;
        if constexpr (KIND == 0) {
            const double ax = px - g.cx, ay = py - g.cy, az = pz - g.cz;
            const double bx = qx - g.cx, by = qy - g.cy, bz = qz - g.cz;
            acc[2] += ax; acc[3] += ay; acc[4] += az;
            acc[5] += bx; acc[6] += by; acc[7] += bz;
            acc[8] += ax * bx; acc[9] += ax * by; acc[10] += ax * bz;
            acc[11] += ay * bx; acc[12] += ay * by; acc[13] += ay * bz;
            acc[14] += az * bx; acc[15] += az * by; acc[16] += az * bz;
        } else if constexpr (KIND == 2) {
            icp_gicp_rows<NACC>(acc, T.m, px, py, pz, qx, qy, qz, Sc + 6 * i, Tn + 6 * (int64_t)j, loss, kparam);
        } else if constexpr (KIND == 3) {
            icp_colored_rows<NACC>(acc, ca, px, py, pz, qx, qy, qz, Tn[3 * (int64_t)j], Tn[3 * (int64_t)j + 1], Tn[3 * (int64_t)j + 2], i, j, loss, kparam);
        } else {
            const double nx = Tn[3 * (int64_t)j], ny = Tn[3 * (int64_t)j + 1], nz = Tn[3 * (int64_t)j + 2];
            const double r = (px - qx) * nx + (py - qy) * ny + (pz - qz) * nz;
            const double w = icp_weight(loss, kparam, r);
            const double J[6] = {py * nz - pz * ny, pz * nx - px * nz, px * ny - py * nx, nx, ny, nz};
            int t = 2;
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int b = a; b < 6; ++b) acc[t++] += J[a] * w * J[b];
#pragma unroll
            for (int a = 0; a < 6; ++a) acc[23 + a] += J[a] * w * r;
            acc[29] += r * r;
        }
    }
    block_reduce_store<NACC>(acc, partials);
}

// One wavefront per accumulator: lane l sums the partials of blocks l, l+64, ... in order, then a fixed
// shuffle tree combines the 64 lane sums -- a deterministic order whatever the launch.
__global__ __launch_bounds__(64) void k_icp_finalize(int nblocks, const double* __restrict__ partials, double* __restrict__ out) {
    const int k = blockIdx.x, lane = threadIdx.x;
    double s = 0.0;
    for (int b = lane; b < nblocks; b += 64) s += partials[(int64_t)b * GSR_ICP_ACC_LEN + k];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) out[k] = s;
}

__global__ __launch_bounds__(256) void k_icp_correspond(int64_t ns, const float* __restrict__ src, Xform T, const int* __restrict__ nn_j,
                                                        const float4* __restrict__ Tq, const unsigned* __restrict__ src_order,
                                                        int64_t* __restrict__ out_idx, double* __restrict__ out_d2) {
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < ns; k += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = src_order ? (int64_t)src_order[k] : k;          // results go back to the caller's order
        const int j = nn_j[k];
        if (j < 0) { out_idx[i] = -1; out_d2[i] = 0.0; continue; }
        const double x = (double)src[3 * k], y = (double)src[3 * k + 1], z = (double)src[3 * k + 2];
        const double px = T.m[0] * x + T.m[1] * y + T.m[2] * z + T.m[3];
        const double py = T.m[4] * x + T.m[5] * y + T.m[6] * z + T.m[7];
        const double pz = T.m[8] * x + T.m[9] * y + T.m[10] * z + T.m[11];
        const float4 q = Tq[j];
        const double dx = px - (double)q.x, dy = py - (double)q.y, dz = pz - (double)q.z;
        out_idx[i] = (int64_t)__float_as_uint(q.w);
        out_d2[i] = dx * dx + dy * dy + dz * dz;
    }
}

// ---- normals from splat covariances: gsr_normals.h (shared with hem.hip: the normals of a level leave with the level)
__global__ __launch_bounds__(256) void k_normals_from_cov(int64_t n, const float* __restrict__ cov6, double* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double v[3];
        normal_of_cov_d(cov6[6 * i], cov6[6 * i + 1], cov6[6 * i + 2], cov6[6 * i + 3], cov6[6 * i + 4], cov6[6 * i + 5], v);
        out[3 * i] = v[0]; out[3 * i + 1] = v[1]; out[3 * i + 2] = v[2];
    }
}

// Covariances of a cloud that has none, for generalized ICP (Open3D GeneralizedICP.cpp, InitializePointCloudForGeneralizedICP):
// C_i = Rx diag(epsilon, 1, 1) Rx^T with Rx = GetRotationFromE1ToX(n_i) = I + [v]x + [v]x^2 / (1 + c), v = e1 x n_i, c = e1 . n_i
// (the identity when c < -0.99, as Open3D has it).  The reference reaches this with its SPARSE input clouds, which carry KNN-30
// normals but no covariances (point_cloud_converter.py:9-28, qt_multiscale_registrator.py:82-85).  cov6 out: xx xy xz yy yz zz.
__global__ __launch_bounds__(256) void k_cov_from_normals(int64_t n, const double* __restrict__ nrm, double eps, double* __restrict__ cov6) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double x = nrm[3 * i], y = nrm[3 * i + 1], z = nrm[3 * i + 2];
        double R[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
        const double c = x;                                   // e1 . n
        if (!(c < -0.99)) {
            const double v[3] = {0.0, -z, y};                 // e1 x n
            const double sv[3][3] = {{0, -v[2], v[1]}, {v[2], 0, -v[0]}, {-v[1], v[0], 0}};
            const double f = 1.0 / (1.0 + c);
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) {
                    double sq = 0;
                    for (int k = 0; k < 3; ++k) sq += sv[a][k] * sv[k][b];
                    R[a][b] = (a == b ? 1.0 : 0.0) + sv[a][b] + sq * f;
                }
        }
        const double d[3] = {eps, 1.0, 1.0};
        int t = 0;
        for (int a = 0; a < 3; ++a)
            for (int b = a; b < 3; ++b) {
                // (Rx * C) * Rx^T in Eigen's evaluation order: the product with the diagonal first
                double sum = 0;
                for (int k = 0; k < 3; ++k) sum += (R[a][k] * d[k]) * R[b][k];
                cov6[6 * i + t++] = sum;
            }
    }
}

// Normals of a cloud WITHOUT covariances: Open3D EstimateNormals(KDTreeSearchParamKNN(knn)) -- what the reference does to a
// sparse input cloud (src/utils/point_cloud_converter.py:9-28, default knn = 30).  One thread per point: its knn nearest
// points (itself included) by (distance, input index) through the same expanding-ring search as the colour gradients;
// with >= 3 of them the covariance by cumulants (Open3D ComputeCovariance), else the identity; normal_of_cov_d.
// Output in the CALLER's point order (order[j] = input index of sorted point j).
__global__ __launch_bounds__(256) void k_knn_normals(int64_t nt, IcpGrid g, const int* __restrict__ cellStart, const float4* __restrict__ Tq,
                                                     const unsigned* __restrict__ order, int knn, double* __restrict__ out) {
    for (int64_t jq = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; jq < nt; jq += (int64_t)gridDim.x * blockDim.x) {
        const float4 qv = Tq[jq];
        const double px = (double)qv.x, py = (double)qv.y, pz = (double)qv.z;
        double* o = out + 3 * (int64_t)order[jq];
        o[0] = 0.0; o[1] = 0.0; o[2] = 1.0;
        if (!(px == px) || !(py == py) || !(pz == pz)) continue;
        double kd[ICP_KNN];
        int kj[ICP_KNN];
        unsigned ki[ICP_KNN];
        int cnt = 0;
        const int cx = icp_cell(px, g.ox, g.inv_c, g.gx), cy = icp_cell(py, g.oy, g.inv_c, g.gy), cz = icp_cell(pz, g.oz, g.inv_c, g.gz);
        const double eps = 1e-13 * (fabs(px) + fabs(py) + fabs(pz) + fabs(g.ox) + fabs(g.oy) + fabs(g.oz) + g.c * (double)(g.gx + g.gy + g.gz));
        const int rmax = (g.gx > g.gy ? g.gx : g.gy) > g.gz ? (g.gx > g.gy ? g.gx : g.gy) : g.gz;
        for (int r = 0; r <= rmax; ++r) {
            for (int dz = -r; dz <= r; ++dz) {
                const int z = cz + dz;
                if (z < 0 || z >= g.gz) continue;
                const int adz = dz < 0 ? -dz : dz;
                for (int dy = -r; dy <= r; ++dy) {
                    const int y = cy + dy;
                    if (y < 0 || y >= g.gy) continue;
                    const int ady = dy < 0 ? -dy : dy;
                    const int rowbase = (z * g.gy + y) * g.gx;
                    int spans[2][2];
                    int nsp = 0;
                    if (adz == r || ady == r) {
                        const int xa = cx - r > 0 ? cx - r : 0, xb = cx + r < g.gx - 1 ? cx + r : g.gx - 1;
                        if (xa <= xb) { spans[0][0] = rowbase + xa; spans[0][1] = rowbase + xb; nsp = 1; }
                    } else {
                        if (cx - r >= 0) { spans[nsp][0] = spans[nsp][1] = rowbase + cx - r; ++nsp; }
                        if (cx + r < g.gx) { spans[nsp][0] = spans[nsp][1] = rowbase + cx + r; ++nsp; }
                    }
                    for (int sidx = 0; sidx < nsp; ++sidx) {
                        const int s0 = cellStart[spans[sidx][0]], e0 = cellStart[spans[sidx][1] + 1];
                        for (int j = s0; j < e0; ++j) {
                            const float4 q = Tq[j];
                            const double dx = px - (double)q.x, dy2 = py - (double)q.y, dz2 = pz - (double)q.z;
                            const double d2 = dx * dx + dy2 * dy2 + dz2 * dz2;
                            const unsigned qi = __float_as_uint(q.w);
                            if (!(d2 == d2)) continue;
                            if (cnt == knn && !(d2 < kd[cnt - 1] || (d2 == kd[cnt - 1] && qi < ki[cnt - 1]))) continue;
                            int pos = cnt < knn ? cnt : knn - 1;             // insertion into the list sorted by (d2, index)
                            while (pos > 0 && (d2 < kd[pos - 1] || (d2 == kd[pos - 1] && qi < ki[pos - 1]))) {
                                kd[pos] = kd[pos - 1]; kj[pos] = kj[pos - 1]; ki[pos] = ki[pos - 1];
                                --pos;
                            }
                            kd[pos] = d2; kj[pos] = j; ki[pos] = qi;
                            if (cnt < knn) ++cnt;
                        }
                    }
                }
            }
            // an unseen point lies at least `reach` away: done when the list is full and its worst entry is strictly closer
            double reach = 1.0 / 0.0;
            if (cx - r > 0) reach = fmin(reach, px - (g.ox + (double)(cx - r) * g.c));
            if (cx + r < g.gx - 1) reach = fmin(reach, (g.ox + (double)(cx + r + 1) * g.c) - px);
            if (cy - r > 0) reach = fmin(reach, py - (g.oy + (double)(cy - r) * g.c));
            if (cy + r < g.gy - 1) reach = fmin(reach, (g.oy + (double)(cy + r + 1) * g.c) - py);
            if (cz - r > 0) reach = fmin(reach, pz - (g.oz + (double)(cz - r) * g.c));
            if (cz + r < g.gz - 1) reach = fmin(reach, (g.oz + (double)(cz + r + 1) * g.c) - pz);
            reach = reach * 0.999999999 - eps;
            if (reach > 0.0 && cnt == knn && kd[cnt - 1] < reach * reach) break;
        }
        double c00 = 1, c01 = 0, c02 = 0, c11 = 1, c12 = 0, c22 = 1;        // fewer than 3 neighbours: identity
        if (cnt >= 3) {
            double m[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            for (int i = 0; i < cnt; ++i) {
                const float4 q = Tq[kj[i]];
                const double x = (double)q.x, y = (double)q.y, z = (double)q.z;
                m[0] += x; m[1] += y; m[2] += z;
                m[3] += x * x; m[4] += x * y; m[5] += x * z; m[6] += y * y; m[7] += y * z; m[8] += z * z;
            }
            for (int k = 0; k < 9; ++k) m[k] /= (double)cnt;
            c00 = m[3] - m[0] * m[0]; c11 = m[6] - m[1] * m[1]; c22 = m[8] - m[2] * m[2];
            c01 = m[4] - m[0] * m[1]; c02 = m[5] - m[0] * m[2]; c12 = m[7] - m[1] * m[2];
        }
        double v[3];
        normal_of_cov_d(c00, c01, c02, c11, c12, c22, v);
        o[0] = v[0]; o[1] = v[1]; o[2] = v[2];
    }
}

// ---- float64 solves (host and device: the device-resident ICP loop runs them in one thread) ----------
__host__ __device__ static void svd3(const double Ain[3][3], double U[3][3], double s[3], double V[3][3]) {
    double B[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { B[i][j] = Ain[i][j]; V[i][j] = i == j; }
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int i = 0; i < 3; ++i) { alpha += B[i][p] * B[i][p]; beta += B[i][q] * B[i][q]; gamma += B[i][p] * B[i][q]; }
                if (gamma == 0) continue;
                off = fmax(off, fabs(gamma) / sqrt(alpha * beta + 1e-300));
                const double zeta = (beta - alpha) / (2 * gamma);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1 + zeta * zeta));
                const double c = 1 / sqrt(1 + t * t), sn = c * t;
                for (int i = 0; i < 3; ++i) {
                    const double bp = B[i][p], bq = B[i][q];
                    B[i][p] = c * bp - sn * bq; B[i][q] = sn * bp + c * bq;
                    const double vp = V[i][p], vq = V[i][q];
                    V[i][p] = c * vp - sn * vq; V[i][q] = sn * vp + c * vq;
                }
            }
        if (off < 1e-17) break;
    }
    int order[3] = {0, 1, 2};
    double nrm[3];
    for (int j = 0; j < 3; ++j) nrm[j] = sqrt(B[0][j] * B[0][j] + B[1][j] * B[1][j] + B[2][j] * B[2][j]);
    for (int a = 0; a < 2; ++a) for (int b = a + 1; b < 3; ++b) if (nrm[order[b]] > nrm[order[a]]) { int t = order[a]; order[a] = order[b]; order[b] = t; }
    double Vs[3][3], Bs[3][3];
    for (int j = 0; j < 3; ++j) { s[j] = nrm[order[j]]; for (int i = 0; i < 3; ++i) { Vs[i][j] = V[i][order[j]]; Bs[i][j] = B[i][order[j]]; } }
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) V[i][j] = Vs[i][j];
    if (!(s[0] > 0)) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) U[i][j] = i == j; return; }
    for (int j = 0; j < 3; ++j) for (int i = 0; i < 3; ++i) U[i][j] = s[j] > 0 ? Bs[i][j] / s[j] : 0.0;
    if (s[1] <= 1e-12 * s[0]) {
        double u0[3] = {U[0][0], U[1][0], U[2][0]};
        int k = fabs(u0[0]) < fabs(u0[1]) ? (fabs(u0[0]) < fabs(u0[2]) ? 0 : 2) : (fabs(u0[1]) < fabs(u0[2]) ? 1 : 2);
        double e[3] = {0, 0, 0};
        e[k] = 1;
        const double d = u0[k];
        double v[3] = {e[0] - d * u0[0], e[1] - d * u0[1], e[2] - d * u0[2]};
        const double n = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        for (int i = 0; i < 3; ++i) U[i][1] = v[i] / n;
    }
    if (s[2] <= 1e-12 * s[0]) {
        U[0][2] = U[1][0] * U[2][1] - U[2][0] * U[1][1];
        U[1][2] = U[2][0] * U[0][1] - U[0][0] * U[2][1];
        U[2][2] = U[0][0] * U[1][1] - U[1][0] * U[0][1];
    }
}
__host__ __device__ static double det3(const double m[3][3]) {
    return m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) +
           m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
}
// x = A^-1 b by LDL^T with diagonal pivoting (what Eigen's LDLT, which Open3D's solvers call, does).  Every index below
// is a compile-time constant once the loops are unrolled -- the pivot's row / column exchange is a chain of tests against
// the constant candidates -- so on the device the 36 + 15 + 18 doubles live in registers: with run-time indices the
// arrays went to scratch memory (720 bytes per lane) and the single-thread solve of k_icp_step took ~15 us.
__host__ __device__ static void solve6(const double A_[6][6], const double b_[6], double x[6]) {
    double A[6][6], L[6][6], D[6], bp[6];
    int perm[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        perm[i] = i; bp[i] = b_[i];
#pragma unroll
        for (int j = 0; j < 6; ++j) { A[i][j] = A_[i][j]; L[i][j] = 0.0; }
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        int piv = k;
        double best = fabs(A[k][k]);
#pragma unroll
        for (int i = k + 1; i < 6; ++i) { const double v = fabs(A[i][i]); if (v > best) { best = v; piv = i; } }
#pragma unroll
        for (int c = k + 1; c < 6; ++c) {
            if (piv == c) {
#pragma unroll
                for (int j = 0; j < 6; ++j) { const double t = A[k][j]; A[k][j] = A[c][j]; A[c][j] = t; }
#pragma unroll
                for (int i = 0; i < 6; ++i) { const double t = A[i][k]; A[i][k] = A[i][c]; A[i][c] = t; }
#pragma unroll
                for (int j = 0; j < k; ++j) { const double t = L[k][j]; L[k][j] = L[c][j]; L[c][j] = t; }
                const int t = perm[k]; perm[k] = perm[c]; perm[c] = t;
                const double tb = bp[k]; bp[k] = bp[c]; bp[c] = tb;        // bp[i] == b[perm[i]] throughout
            }
        }
        D[k] = A[k][k];
        L[k][k] = 1;
#pragma unroll
        for (int i = k + 1; i < 6; ++i) L[i][k] = A[i][k] / D[k];
#pragma unroll
        for (int i = k + 1; i < 6; ++i)
#pragma unroll
            for (int j = k + 1; j < 6; ++j) A[i][j] -= L[i][k] * D[k] * L[j][k];
    }
    double y[6], z[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double s = bp[i];
#pragma unroll
        for (int j = 0; j < i; ++j) s -= L[i][j] * y[j];
        y[i] = s;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) y[i] /= D[i];
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double s = y[i];
#pragma unroll
        for (int j = i + 1; j < 6; ++j) s -= L[j][i] * z[j];
        z[i] = s;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int c = 0; c < 6; ++c)
            if (perm[i] == c) x[c] = z[i];
}
__host__ __device__ static void mat4_identity(double T[16]) { for (int i = 0; i < 16; ++i) T[i] = (i % 5 == 0) ? 1.0 : 0.0; }
__host__ __device__ static void mat4_mul(const double A[16], const double B[16], double C[16]) {
    double R[16];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s += A[4 * i + k] * B[4 * k + j]; R[4 * i + j] = s; }
    for (int i = 0; i < 16; ++i) C[i] = R[i];
}

// estimator update from the reduced accumulators (Open3D TransformationEstimation*.ComputeTransformation)
__host__ __device__ static void estimate_update(const double ctr[3], int kind, const double* acc, double update[16]) {
    mat4_identity(update);
    const double n = acc[0];
    if (!(n > 0)) return;                                   // no correspondences -> identity
    if (kind == GSR_ICP_POINT_TO_POINT) {                   // Eigen::umeyama(src, dst, false)
        const double mp[3] = {acc[2] / n, acc[3] / n, acc[4] / n}, mq[3] = {acc[5] / n, acc[6] / n, acc[7] / n};
        double sigma[3][3], U[3][3], V[3][3], s[3];
        for (int r = 0; r < 3; ++r)
            for (int col = 0; col < 3; ++col) sigma[r][col] = acc[8 + 3 * col + r] / n - mq[r] * mp[col];   // dst x src^T
        svd3(sigma, U, s, V);
        double S[3] = {1, 1, 1};
        if (det3(U) * det3(V) < 0) S[2] = -1;
        double R[3][3];
        for (int r = 0; r < 3; ++r)
            for (int col = 0; col < 3; ++col) { double v = 0; for (int k = 0; k < 3; ++k) v += U[r][k] * S[k] * V[col][k]; R[r][col] = v; }
        for (int r = 0; r < 3; ++r) {
            for (int col = 0; col < 3; ++col) update[4 * r + col] = R[r][col];
            double Rp = 0;
            for (int col = 0; col < 3; ++col) Rp += R[r][col] * (mp[col] + ctr[col]);
            update[4 * r + 3] = mq[r] + ctr[r] - Rp;
        }
    } else {                                                 // x = solve(JTJ, -JTr); Rz(x2) Ry(x1) Rx(x0), t = x3..5
        double JTJ[6][6], nb[6], x[6];
        int t = 2;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b = a; b < 6; ++b) { JTJ[a][b] = acc[t]; JTJ[b][a] = acc[t]; ++t; }
#pragma unroll
        for (int a = 0; a < 6; ++a) nb[a] = -acc[23 + a];
        solve6(JTJ, nb, x);
        const double ca = cos(x[0]), sa = sin(x[0]), cb = cos(x[1]), sb = sin(x[1]), cg = cos(x[2]), sg = sin(x[2]);
        update[0] = cg * cb; update[1] = cg * sb * sa - sg * ca; update[2] = cg * sb * ca + sg * sa; update[3] = x[3];
        update[4] = sg * cb; update[5] = sg * sb * sa + cg * ca; update[6] = sg * sb * ca - cg * sa; update[7] = x[4];
        update[8] = -sb;     update[9] = cb * sa;                update[10] = cb * ca;               update[11] = x[5];
    }
}


// ---- device-resident ICP loop -------------------------------------------------------------------------
// registration_icp's loop needs, per iteration, one correspondence pass and a 3x3 / 6x6 solve on 32
// doubles.  Driving it from the host costs a device->host copy and a stream synchronisation per
// iteration (~60 us, comparable to the kernel itself on small levels); here the transform, the
// convergence test and the solve stay on the device: k_icp_accumulate_dev reads T from IcpState,
// k_icp_step reduces the block partials (same fixed order as k_icp_finalize), tests convergence,
// solves and updates T in ONE thread.  The host enqueues iterations in chunks and looks at `done` once
// per chunk; iterations enqueued after convergence return immediately.
struct IcpState {
    double T[16];
    double fit, rmse;          // of the latest evaluation
    double ctr[3], nsg, rel_fit, rel_rmse;
    int iters, done, max_iter, kind, evals;
};

__device__ __forceinline__ bool icp_state_done(const IcpState* st) { return st->done != 0; }
__device__ __forceinline__ void icp_state_T(const IcpState* st, double T[12]) {
#pragma unroll
    for (int i = 0; i < 12; ++i) T[i] = st->T[i];
}

// FUSE: what the LAST workgroup to finish does with the block partials (it learns that it is the last from a device-scope
// ticket; release / acquire at agent scope around it as MI355X_MICROARCH.md prescribes for a cross-CU hand-off):
//   0  nothing (k_icp_reduce / k_icp_step follow as their own launches)
//   1  the whole step -- fold the partials in k_icp_finalize's order, test convergence, solve, update T: ONE launch per ICP
//      iteration instead of two (a coarse level's iteration was 36 us + 8.5 us of k_icp_step and its launch gap)
//   2  fold the partials into acc_out (the rank-local vector of a multi-GPU source split; the all-reduce and k_icp_step follow)
// (FUSE is a template parameter: the step's code costs the kernel ~24 VGPRs, and with them a wave per SIMD on the levels whose
// blocks just fill the chip once; the bound of 3 waves per SIMD keeps the fused forms at <= 168 registers -- whatever does not
// fit spills in the step's code, which one workgroup runs once)
template <int KIND, int BLOCK, int FUSE>
__global__ __launch_bounds__(256, (FUSE != 0 && KIND != 2) ? 3 : 1) void k_icp_accumulate_dev(int64_t ns, const float* __restrict__ src, IcpState* st,
                                                            IcpGrid g, const int* __restrict__ cellStart, const int* __restrict__ nn_j,
                                                            const float4* __restrict__ Tq, const double* __restrict__ Tn,
                                                            const double* __restrict__ Sc, ColorArgs ca, double max_corr2,
                                                            int loss, double kparam, double* partials, unsigned* ticket,
                                                            double* acc_out, int nb_logical) {
    constexpr int fuse = FUSE;
    // FUSE == 3: the RESIDENT form (GSR_ICP_PERSISTENT=1, a cooperative launch: every workgroup is on the chip).  The workgroups stay
    // for the whole registration: evaluate, arrive at a device-scope ticket, the last one folds + solves + publishes a generation
    // word, the others spin on it (s_sleep, bounded), everybody reads the new transform and goes again.  ticket[0] = arrivals
    // (monotonic), ticket[1] = generation, ticket[2] = "a spin ran out" (the host reports it).
    constexpr bool PERSIST = FUSE == 3;
    constexpr int NACC = KIND == 0 ? 17 : 30;
    __shared__ int s_last;
    __shared__ int s_stop;
    __shared__ double s_x[8][GSR_ICP_ACC_LEN];
    __shared__ IcpState s_st;
    const IcpRange rg = icp_block_range(ns, nb_logical < 0 ? -nb_logical : nb_logical, FUSE == 0 && nb_logical > 0);
    for (unsigned ev = 0;; ++ev) {
    double T[12];
    if constexpr (PERSIST) {
        // the state another workgroup wrote: agent-scope loads (past the L1 and the scalar cache), behind the acquire of the spin
        if (threadIdx.x == 0) s_stop = __hip_atomic_load(&st->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) | (int)__hip_atomic_load(ticket + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (s_stop) return;
#pragma unroll
        for (int i = 0; i < 12; ++i) T[i] = __hip_atomic_load(&st->T[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
    if (st->done) {
        // converged: nothing to search.  The ranks of a multi-GPU run still meet in the collective: zeros
        if (fuse == 2 && blockIdx.x == 0 && threadIdx.x < GSR_ICP_ACC_LEN) acc_out[threadIdx.x] = 0.0;
        return;
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) T[i] = st->T[i];
    }
    double acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k) acc[k] = 0.0;
    if (rg.row < 0) return;                         // (a padding workgroup of the XCD mapping; FUSE != 0 launches none)
    for (int64_t i = rg.lo + threadIdx.x; i < rg.hi; i += blockDim.x) {
        const double x = (double)src[3 * i], y = (double)src[3 * i + 1], z = (double)src[3 * i + 2];
        const double px = T[0] * x + T[1] * y + T[2] * z + T[3];
        const double py = T[4] * x + T[5] * y + T[6] * z + T[7];
        const double pz = T[8] * x + T[9] * y + T[10] * z + T[11];
        int j;
        if (nn_j) {                                 // optional two-kernel form: neighbours from k_icp_nn (accepted or -1)
            j = nn_j[i];
            if (j < 0) continue;
        } else {
            double dd;
            j = icp_nearest<BLOCK>(g, cellStart, Tq, px, py, pz, dd, max_corr2);
            if (j < 0 || !(dd < max_corr2)) continue;
        }
        const float4 q = Tq[j];
        const double qx = (double)q.x, qy = (double)q.y, qz = (double)q.z;
        const double ddx = px - qx, ddy = py - qy, ddz = pz - qz;
        const double d2 = ddx * ddx + ddy * ddy + ddz * ddz;
        acc[0] += 1.0;
        acc[1] += d2;
        if constexpr (KIND == 0) {
            const double ax = px - g.cx, ay = py - g.cy, az = pz - g.cz;
            const double bx = qx - g.cx, by = qy - g.cy, bz = qz - g.cz;
            acc[2] += ax; acc[3] += ay; acc[4] += az;
            acc[5] += bx; acc[6] += by; acc[7] += bz;
            acc[8] += ax * bx; acc[9] += ax * by; acc[10] += ax * bz;
            acc[11] += ay * bx; acc[12] += ay * by; acc[13] += ay * bz;
            acc[14] += az * bx; acc[15] += az * by; acc[16] += az * bz;
        } else if constexpr (KIND == 2) {
            icp_gicp_rows<NACC>(acc, T, px, py, pz, qx, qy, qz, Sc + 6 * i, Tn + 6 * (int64_t)j, loss, kparam);
        } else if constexpr (KIND == 3) {
            icp_colored_rows<NACC>(acc, ca, px, py, pz, qx, qy, qz, Tn[3 * (int64_t)j], Tn[3 * (int64_t)j + 1], Tn[3 * (int64_t)j + 2], i, j, loss, kparam);
        } else {
            const double nx = Tn[3 * (int64_t)j], ny = Tn[3 * (int64_t)j + 1], nz = Tn[3 * (int64_t)j + 2];
            const double r = (px - qx) * nx + (py - qy) * ny + (pz - qz) * nz;
            const double w = icp_weight(loss, kparam, r);
            const double J[6] = {py * nz - pz * ny, pz * nx - px * nz, px * ny - py * nx, nx, ny, nz};
            int t = 2;
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int b = a; b < 6; ++b) acc[t++] += J[a] * w * J[b];
#pragma unroll
            for (int a = 0; a < 6; ++a) acc[23 + a] += J[a] * w * r;
            acc[29] += r * r;
        }
    }
    block_reduce_store<NACC, FUSE != 0>(acc, partials, rg.row);
    if constexpr (FUSE == 0) return;
    // Hand-off inside one launch (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility"): the
    // partials were stored by the lanes of wave 0; lane 0 RELEASES them at agent scope and takes a device-scope ticket
    // (acquire-release); the workgroup whose ticket is the last one ACQUIRES and reads every block's partials.  (Round 3 shipped
    // this with relaxed atomics and a bare vmcnt(0) -- no release / acquire pair ordered the partials before the ticket across
    // XCDs; the fences cost ~4 us per iteration, which is one more reason the knob stays off: two launches are faster.)
    if (threadIdx.x < 64) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            s_last = t == (PERSIST ? (ev + 1u) * gridDim.x - 1u : gridDim.x - 1u) ? 1 : 0;
            if (s_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
    }
    __syncthreads();
    if (s_last) {
        if constexpr (PERSIST) {
            if (threadIdx.x < sizeof(IcpState) / 4)
                reinterpret_cast<int*>(&s_st)[threadIdx.x] = __hip_atomic_load(reinterpret_cast<const int*>(st) + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (threadIdx.x < sizeof(IcpState) / 4) reinterpret_cast<int*>(&s_st)[threadIdx.x] = reinterpret_cast<const int*>(st)[threadIdx.x];
        }
        icp_fold_partials<256, true>((int)gridDim.x, partials, nullptr, s_x);
        if (fuse == 2) {
            if (threadIdx.x < GSR_ICP_ACC_LEN) acc_out[threadIdx.x] = s_x[0][threadIdx.x];
        } else if (threadIdx.x == 0) {
            icp_step_solve(&s_x[0][0], s_st, st);
        }
        if constexpr (!PERSIST) {
            if (threadIdx.x == 0) *ticket = 0u;               // for the next launch (a kernel boundary away)
        }
    }
    if constexpr (!PERSIST) return;
    if (s_last) {
        __syncthreads();                                      // (uniform within the workgroup: s_last is shared)
        if (threadIdx.x == 0) {                               // the new state is out: everybody may go on
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_store(ticket + 1, ev + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else if (threadIdx.x == 0) {
        unsigned spins = 0;
        // relaxed polls (a load past the L1, nothing invalidated), ONE acquire when the word has moved
        while (__hip_atomic_load(ticket + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < ev + 1u) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > (1u << 21)) {                       // ~0.5 s: something is badly wrong
                __hip_atomic_store(ticket + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    }
}

// `reduced` == NULL: fold the block partials here (single GPU).  Multi-GPU source split: k_icp_reduce folds them into a
// device vector, the all-reduce callback sums that vector over the ranks (RCCL, on this stream), and this kernel starts
// from the reduced vector -- every rank then takes the identical decision and the identical update.
// 512 threads: the single-thread solve behind the reduction needs ~150 registers (a 1024-thread launch bound allows 128: it
// spilled to scratch).  The reduction keeps k_icp_finalize's summation ORDER per accumulator k -- s_l = sum over the blocks
// b = l, l + 64, ... ascending, then the butterfly s_l + s_(l ^ 32), + (l ^ 16), ... down to lane 0 -- but reads the
// partials as rows: thread (g = t / 32, k = t % 32) forms s_g, s_(g+16), s_(g+32), s_(g+48) of accumulator k (32 threads
// read one 256-byte row per load; the old wave-per-accumulator loop read 8 bytes of 64 different rows), folds the two
// butterfly steps it holds both operands of, and an LDS tree over g does the remaining four.
#define ICP_STEP_THREADS 512
// The fold in k_icp_finalize's summation ORDER per accumulator k -- s_l = sum over the blocks b = l, l + 64, ... ascending, then the
// butterfly s_l + s_(l ^ 32), + (l ^ 16), ... down to lane 0 -- with THREADS = 32 G threads: thread (g = t / 32, k = t % 32) forms
// the lane sums s_g, s_(g + G), ... of accumulator k (32 threads read one 256-byte row per load, four rounds of loads in flight),
// folds the butterfly steps it holds both operands of, and an LDS tree over g does the remaining log2 G.  Result: s_x[0][k].
// SC1: the partials were stored write-through by other workgroups of the SAME launch: read them with sc1 loads (past the L1)
template <int THREADS, bool SC1>
__device__ __forceinline__ void icp_fold_partials(int nblocks, const double* partials, const double* __restrict__ reduced, double (*s_x)[32]) {
    constexpr int G = THREADS / 32, NL = 64 / G;
    const int k = threadIdx.x & 31, g = threadIdx.x >> 5;
    if (reduced) {
        if (g == 0) s_x[0][k] = reduced[k];
        __syncthreads();
        return;
    }
    double sl[NL];
#pragma unroll
    for (int j = 0; j < NL; ++j) sl[j] = 0.0;
    for (int b0 = g; b0 < nblocks; b0 += 256) {
        double v[4][NL];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                const int b = b0 + 64 * u + G * j;
                if (SC1) v[u][j] = b < nblocks ? __hip_atomic_load(partials + (int64_t)b * GSR_ICP_ACC_LEN + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
                else v[u][j] = b < nblocks ? partials[(int64_t)b * GSR_ICP_ACC_LEN + k] : 0.0;
            }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < NL; ++j)
                if (b0 + 64 * u + G * j < nblocks) sl[j] += v[u][j];
    }
#pragma unroll
    for (int h = NL / 2; h >= 1; h >>= 1)                   // butterfly steps 32, 16, ... between the lanes this thread holds
#pragma unroll
        for (int j = 0; j < h; ++j) sl[j] = sl[j] + sl[j + h];
    s_x[g][k] = sl[0];
    __syncthreads();
    for (int o = G / 2; o > 0; o >>= 1) {
        if (g < o) s_x[g][k] = s_x[g][k] + s_x[g + o][k];
        __syncthreads();
    }
}
// one thread: fitness / RMSE, the relative-change test, the estimator's solve and T <- update * T (registration_icp's loop body)
__device__ __forceinline__ void icp_step_solve(const double* acc32, const IcpState& s_st, IcpState* st) {
    double acc[GSR_ICP_ACC_LEN];
#pragma unroll
    for (int i = 0; i < GSR_ICP_ACC_LEN; ++i) acc[i] = acc32[i];
    const double fit = acc[0] > 0 ? acc[0] / s_st.nsg : 0.0, rmse = acc[0] > 0 ? sqrt(acc[1] / acc[0]) : 0.0;
    const bool first = s_st.evals == 0;
    const bool stop = !first && fabs(s_st.fit - fit) < s_st.rel_fit && fabs(s_st.rmse - rmse) < s_st.rel_rmse;
    st->fit = fit; st->rmse = rmse;
    st->evals = s_st.evals + 1;
    if (stop || s_st.iters >= s_st.max_iter) { st->done = 1; return; }
    double update[16], T[16];
    estimate_update(s_st.ctr, s_st.kind, acc, update);
#pragma unroll
    for (int i = 0; i < 16; ++i) T[i] = s_st.T[i];
    mat4_mul(update, T, T);
#pragma unroll
    for (int i = 0; i < 16; ++i) st->T[i] = T[i];
    st->iters = s_st.iters + 1;
}
// 512 threads: the single-thread solve behind the reduction needs ~120 registers (a 1024-thread launch bound allows 128: it
// spilled to scratch)
__global__ __launch_bounds__(ICP_STEP_THREADS) void k_icp_step(int nblocks, const double* __restrict__ partials, const double* __restrict__ reduced,
                                                   IcpState* __restrict__ st) {
    __shared__ double s_x[16][GSR_ICP_ACC_LEN];
    __shared__ IcpState s_st;
    static_assert(GSR_ICP_ACC_LEN == 32 && sizeof(IcpState) % 4 == 0, "k_icp_step layout");
    if (threadIdx.x < sizeof(IcpState) / 4) reinterpret_cast<int*>(&s_st)[threadIdx.x] = reinterpret_cast<const int*>(st)[threadIdx.x];
    icp_fold_partials<ICP_STEP_THREADS>(nblocks, partials, reduced, s_x);
    if (threadIdx.x != 0 || s_st.done) return;
    icp_step_solve(&s_x[0][0], s_st, st);
}

// rank-local accumulator vector of one iteration (same fixed summation order as k_icp_finalize); zeros once converged, so
// that the ranks keep calling the collective in step
__global__ __launch_bounds__(1024) void k_icp_reduce(int nblocks, const double* __restrict__ partials, const IcpState* __restrict__ st,
                                                     double* __restrict__ out) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const bool done = st->done != 0;
    for (int k = wv; k < GSR_ICP_ACC_LEN; k += 16) {
        double s = 0.0;
        if (!done)
            for (int b = lane; b < nblocks; b += 64) s += partials[(int64_t)b * GSR_ICP_ACC_LEN + k];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) out[k] = s;
    }
}

}  // namespace gsr

using namespace gsr;

struct gsr_icp_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    IcpGrid grid;
    bool have_target = false, have_normals = false, have_source = false;
    int64_t nt = 0, ns = 0, ns_global = 0;
    double max_corr = 0;
    DevBuf src_raw, src_order, state, nn_j, Tc, Sc, stage_cov, Ti, Tg, Si, ticket;
    bool have_tcov = false, have_scov = false, have_tcol = false, have_scol = false;
    double lambda_geometric = 0.968;            // Open3D TransformationEstimationForColoredICP default
    bool src_sorted = false;
    bool device_loop = true;        // GSR_ICP_DEVICE_LOOP=0 selects the host-driven loop
    bool block_search = true;       // GSR_ICP_BLOCK_SEARCH=0: always the ring loop from ring 0
    int tile_search = 0;            // GSR_ICP_TILE: the search of a fine level (rings == 1) over LDS-staged tiles (k_icp_nn_tile): 0 (default) = never, -1 = where
                                    // the search has its own kernel (nn_mode), 1 = also for gsr_icp_correspondences on any size (tests).  Built for VERDICT r04
                                    // item 5 and measured SLOWER on the bench's converged 5 M x 5 M level: 1.21 against 0.39 ms per iteration
                                    // (profiles/r05g_icp_tile_ab.txt) -- a workgroup stages 3 264 target points (12 rows of the grid) for its 256 queries, of
                                    // which a converged query needs the six in its own row: the per-thread search reads what it needs, the tile what it might
    bool xcd_ranges = true;         // GSR_ICP_XCD=0: logical block = physical block (every XCD walks the whole source)
    bool fused_step = false;        // GSR_ICP_FUSED_STEP=1: the accumulate kernel's last workgroup does k_icp_step's (or k_icp_reduce's) work instead of a launch of its own: measured equal (44.8 vs 44.0 us at 185 k), so off
                                    // accumulate kernel's last workgroup
    gsr_comm* comm = nullptr;       // multi-GPU source split through a communicator (gsr_icp_set_comm)
    unsigned* host_rb = nullptr;    // pinned host memory for small read-backs: 256 words + the sequence flag
    unsigned long long rb_seq = 0;
    // Search / accumulate split (GSR_ICP_NN_KERNEL): 0 = one fused kernel (120 VGPRs with the 30 float64 accumulators:
    // 4 waves per SIMD); 2 = a thread-per-point search kernel (56 VGPRs, 8 waves per SIMD) writes nn_j and a streaming
    // kernel accumulates -- the search is latency bound, so occupancy wins: 21 % faster at 5 M points, equal at 0.5 M,
    // 8 % slower at 0.2 M (one more launch).  -1 = choose by size (the default).  Same results either way
    // (tests/test_icp_gpu.py::test_icp_knobs_change_nothing).
    int nn_kernel = -1;
    // the search in its own kernel (k_icp_nn, few registers, full occupancy) + a streaming accumulate from 400 k source points on:
    // measured on the bench's levels with the XCD-contiguous ranges, fused / split: 185 k 51.7 / 59.2 us, 556 k 90.1 / 81.6, 1.67 M 233 / 190,
    // 5 M 563 / 397 per iteration
    int nn_mode() const { return nn_kernel >= 0 ? (nn_kernel ? 2 : 0) : (ns >= 400000 ? 2 : 0); }
    double cell_target = 2.0;       // target points per grid cell (GSR_ICP_CELL_TARGET).  Measured at 5M x 5M: 0.5 makes a
                                    // converged iteration 1.7x faster but a cold start (offsets ~ max_corr) 1.6x slower: keep 2
    DevBuf hist;
    bool robust_box = false;        // the current target's grid lies over the trimmed box (far outliers clamped into the boundary cells)
    bool persistent = false;        // GSR_ICP_PERSISTENT=1: one RESIDENT kernel per registration on the levels that take the fused search (experiment)
    bool adapt_cells = false;       // GSR_ICP_ADAPT=1: a finer grid when the points are clumped (experiment)
    double occupancy = 0.0;         // of the cell an average point sits in (measured when adapt_cells)
    bool robust_allowed = true;     // GSR_ICP_ROBUST_BOX=0: always the box of all points (test knob: results must not change)
    DevBuf bbox, keys, idx, skeys, order, cellStart, Tq, Tn, stage_xyz, stage_nrm, src, partials, acc_dev, rocprim_tmp, corr_idx, corr_d2;
    gsr_allreduce_fn allreduce = nullptr;
    void* allreduce_user = nullptr;
    gsr_allreduce_dev64_fn allreduce_dev = nullptr;      // device-resident loop with a stream-ordered collective per iteration
    void* allreduce_dev_user = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr, eb0 = nullptr, eb1 = nullptr;      // e0 / e1: the iteration loop; eb0 / eb1: the target index build
    bool defer_sync = false;        // gsr_icp_register_clouds: set_target / set_source do not wait for the stream (the registration behind them does)
    bool build_pending = false;     // ms_build has not been read from eb0 / eb1 yet
    float ms_build = 0, ms_iter = 0;
    int n_iter_kernels = 0;
    int max_cells = 1 << 25;
    // workgroups of the search + accumulate kernel: three fit a CU (141 VGPRs), 768 are ONE round on the chip.  With 1 024 the last
    // 256 ran as a second, quarter-full round: 108 -> 95 us per iteration at 556 k source points (GSR_ICP_BLOCKS)
    int nblocks = 768;
};

namespace {

// Logical blocks for ns points when at most `cap` workgroups are wanted: every block takes a whole number of 256-point trips
// (icp_block_range), and there are exactly as many blocks as have points -- the XCD mapping deals them out in eighths, so
// blocks without work at the end would leave whole XCDs idle.
inline int icp_blocks(int64_t ns, int cap) {
    if (ns <= 0) return 1;
    const int64_t ppb = ((ns + cap - 1) / cap + 255) / 256 * 256;
    return (int)((ns + ppb - 1) / ppb);
}
inline int nn_grid1(int64_t ns) { return icp_blocks(ns, 16384); }      // grid of the search kernel: one thread per point, capped
// the search kernel of one evaluation: over LDS tiles on a fine level (rings == 1), else per thread (27-cell block first where the
// correspondence distance spans two cells or more: blockf)
template <bool FROM_STATE>
void launch_icp_nn(gsr_icp_ctx* c, hipStream_t st, const Xform& X, const IcpState* state, double mc2, bool blockf) {
    const int g1 = nn_grid1(c->ns);
    const dim3 grid(8 * ((g1 + 7) / 8)), blk(256);
    const int nbl = c->xcd_ranges ? g1 : -g1;
    if (c->tile_search != 0 && c->grid.rings == 1)
        hipLaunchKernelGGL((k_icp_nn_tile<FROM_STATE>), grid, blk, ICP_TILE_LDS, st, c->ns, c->src.as<float>(), X, state, c->grid, c->cellStart.as<int>(), c->Tq.as<float4>(),
                           mc2, c->nn_j.as<int>(), nbl);
    else if (blockf)
        hipLaunchKernelGGL((k_icp_nn<FROM_STATE, 1>), grid, blk, 0, st, c->ns, c->src.as<float>(), X, state, c->grid, c->cellStart.as<int>(), c->Tq.as<float4>(), mc2,
                           c->nn_j.as<int>(), nbl);
    else
        hipLaunchKernelGGL((k_icp_nn<FROM_STATE, 0>), grid, blk, 0, st, c->ns, c->src.as<float>(), X, state, c->grid, c->cellStart.as<int>(), c->Tq.as<float4>(), mc2,
                           c->nn_j.as<int>(), nbl);
}

int32_t run_accumulate(gsr_icp_ctx* c, const double* T, int kind, int loss, double k, double* acc, bool timed) {
    if (!c->have_target || !c->have_source) return fail(GSR_E_INVALID, "icp: target and source must be set first");
    if (kind < GSR_ICP_POINT_TO_POINT || kind > GSR_ICP_COLORED) return fail(GSR_E_INVALID, "icp: unknown estimation kind %d", kind);
    if (kind == GSR_ICP_POINT_TO_PLANE && !c->have_normals)
        return fail(GSR_E_PRECONDITION, "TransformationEstimationPointToPlane requires target normals");
    if (kind == GSR_ICP_COLORED && (!c->have_normals || !c->have_tcol || !c->have_scol))
        return fail(GSR_E_PRECONDITION, "ColoredICP requires target normals and the colours of both clouds");
    if (kind == GSR_ICP_GENERALIZED && (!c->have_tcov || !c->have_scov))
        return fail(GSR_E_PRECONDITION, "TransformationEstimationForGeneralizedICP requires source and target covariances");
    hipStream_t st = c->stream;
    Xform X;
    for (int i = 0; i < 12; ++i) X.m[i] = T[i];
    const int nb = icp_blocks(c->ns, c->nblocks);
    GSR_TRY(c->partials.reserve((size_t)nb * GSR_ICP_ACC_LEN * 8));
    GSR_TRY(c->acc_dev.reserve(GSR_ICP_ACC_LEN * 8));
    GSR_HIP(hipMemsetAsync(c->partials.p, 0, (size_t)nb * GSR_ICP_ACC_LEN * 8, st));
    if (timed) GSR_HIP(hipEventRecord(c->e0, st));
    const double mc2 = c->max_corr * c->max_corr;
    const int* nnj = nullptr;
    if (c->nn_mode()) {
        GSR_TRY(c->nn_j.reserve((size_t)c->ns * 4));
        launch_icp_nn<false>(c, st, X, (const IcpState*)nullptr, mc2, false);
        nnj = c->nn_j.as<int>();
    }
    const ColorArgs cargs = {c->Ti.as<double>(), c->Tg.as<double>(), c->Si.as<double>(), sqrt(c->lambda_geometric), sqrt(1.0 - c->lambda_geometric)};
    if (kind == GSR_ICP_COLORED)
        hipLaunchKernelGGL(k_icp_accumulate<3>, dim3(nb), dim3(256), 0, st, c->ns, c->src.as<float>(), X, c->grid, c->cellStart.as<int>(), nnj,
                           c->Tq.as<float4>(), c->Tn.as<double>(), (const double*)nullptr, cargs, mc2, loss, k, c->partials.as<double>());
    else if (kind == GSR_ICP_POINT_TO_POINT)
        hipLaunchKernelGGL(k_icp_accumulate<0>, dim3(nb), dim3(256), 0, st, c->ns, c->src.as<float>(), X, c->grid, c->cellStart.as<int>(), nnj,
                           c->Tq.as<float4>(), (const double*)nullptr, (const double*)nullptr, cargs, mc2, loss, k, c->partials.as<double>());
    else if (kind == GSR_ICP_POINT_TO_PLANE)
        hipLaunchKernelGGL(k_icp_accumulate<1>, dim3(nb), dim3(256), 0, st, c->ns, c->src.as<float>(), X, c->grid, c->cellStart.as<int>(), nnj,
                           c->Tq.as<float4>(), c->Tn.as<double>(), (const double*)nullptr, cargs, mc2, loss, k, c->partials.as<double>());
    else
        hipLaunchKernelGGL(k_icp_accumulate<2>, dim3(nb), dim3(256), 0, st, c->ns, c->src.as<float>(), X, c->grid, c->cellStart.as<int>(), nnj,
                           c->Tq.as<float4>(), c->Tc.as<double>(), c->Sc.as<double>(), cargs, mc2, loss, k, c->partials.as<double>());
    hipLaunchKernelGGL(k_icp_finalize, dim3(GSR_ICP_ACC_LEN), dim3(64), 0, st, nb, c->partials.as<double>(), c->acc_dev.as<double>());
    if (timed) GSR_HIP(hipEventRecord(c->e1, st));
    GSR_HIP(hipMemcpyAsync(acc, c->acc_dev.p, GSR_ICP_ACC_LEN * 8, hipMemcpyDeviceToHost, st));
    GSR_HIP(hipStreamSynchronize(st));
    if (timed) {
        float ms = 0;
        (void)hipEventElapsedTime(&ms, c->e0, c->e1);
        c->ms_iter += ms;
        c->n_iter_kernels += 1;
    }
    if (c->allreduce) {
        int32_t r = c->allreduce(acc, GSR_ICP_ACC_LEN, c->allreduce_user);
        if (r != 0) return fail(GSR_E_INVALID, "icp: all-reduce callback returned %d", r);
    }
    return GSR_OK;
}

// The resident form of one registration (k_icp_accumulate_dev<KIND, BLOCK, 3>): a cooperative launch -- every workgroup must be on the
// chip, they wait for one another --, so it is taken only when the occupancy query says all nb workgroups fit.  Returns 1 when it
// was launched, 0 when the caller should take the launch-per-iteration loop, < 0 on error.
template <int KIND, int BLOCK>
int32_t launch_persistent(gsr_icp_ctx* c, int nb, const double* TN, const double* SC, ColorArgs cargs, double mc2, int loss, double k) {
    const void* fn = (const void*)k_icp_accumulate_dev<KIND, BLOCK, 3>;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, 0) != hipSuccess) { (void)hipGetLastError(); return 0; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, c->device) != hipSuccess) { (void)hipGetLastError(); return 0; }
    if (!prop.cooperativeLaunch || (int64_t)per_cu * prop.multiProcessorCount < nb) return 0;
    int64_t ns = c->ns;
    const float* src = c->src.as<float>();
    IcpState* st = c->state.as<IcpState>();
    IcpGrid g = c->grid;
    const int* cellStart = c->cellStart.as<int>();
    const int* nnj = nullptr;
    const float4* Tq = c->Tq.as<float4>();
    double* partials = c->partials.as<double>();
    unsigned* ticket = c->ticket.as<unsigned>();
    double* acc_out = c->acc_dev.as<double>();
    int nb_logical = -nb;                    // (logical block = physical block: the resident form launches exactly nb workgroups)
    void* args[] = {&ns, &src, &st, &g, &cellStart, &nnj, &Tq, &TN, &SC, &cargs, &mc2, &loss, &k, &partials, &ticket, &acc_out, &nb_logical};
    const hipError_t e = hipLaunchCooperativeKernel(fn, dim3(nb), dim3(256), args, 0, c->stream);
    if (e != hipSuccess) { (void)hipGetLastError(); return 0; }
    return 1;
}

}  // namespace

extern "C" {

// Small device results the host waits for (the bounding box, the loop state) are copied by one tiny kernel into pinned host
// memory, the sequence number of the round trip last, and the host polls that word: a hipMemcpyAsync into pageable memory
// plus hipStreamSynchronize costs 30-50 us per round trip (hem.hip does the same).
__global__ void k_icp_publish(const unsigned* __restrict__ src, int nwords, unsigned* __restrict__ host, unsigned long long* __restrict__ flag,
                              unsigned long long seq) {
    for (int t = threadIdx.x; t < nwords; t += blockDim.x) __hip_atomic_store(host + t, src[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
static int32_t icp_fetch(gsr_icp_ctx* c, const void* dev, void* out, size_t bytes) {
    hipStream_t st = c->stream;
    if (!c->host_rb || bytes > 1024 || (bytes & 3)) {
        GSR_HIP(hipMemcpyAsync(out, dev, bytes, hipMemcpyDeviceToHost, st));
        GSR_HIP(hipStreamSynchronize(st));
        return GSR_OK;
    }
    const unsigned long long seq = ++c->rb_seq;
    unsigned long long* flag = reinterpret_cast<unsigned long long*>(c->host_rb + 256);
    hipLaunchKernelGGL(k_icp_publish, dim3(1), dim3(64), 0, st, (const unsigned*)dev, (int)(bytes >> 2), c->host_rb, flag, seq);
    GSR_HIP(hipGetLastError());
    (void)hipStreamQuery(st);
    (void)hipGetLastError();
    bool seen = false;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 1; !(seen = __atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq); ++spins) {
        if ((spins & 0x3ffu) == 0u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) break;
        gsr::cpu_relax(spins);
    }
    if (!seen) GSR_HIP(hipStreamSynchronize(st));
    memcpy(out, (const void*)c->host_rb, bytes);
    return GSR_OK;
}

int32_t gsr_icp_create(gsr_icp_ctx** out, int32_t device, void* stream) {
    if (!out) return fail(GSR_E_INVALID, "gsr_icp_create: out is NULL");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(GSR_E_NO_DEVICE, "gsr_icp_create: no HIP device visible (this backend has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(GSR_E_INVALID, "gsr_icp_create: device %d out of range", device);
    GSR_HIP(hipSetDevice(device));
    gsr_icp_ctx* c = new gsr_icp_ctx();
    c->device = device;
    c->stream = (hipStream_t)stream;
    if (hipEventCreate(&c->e0) != hipSuccess || hipEventCreate(&c->e1) != hipSuccess || hipEventCreate(&c->eb0) != hipSuccess || hipEventCreate(&c->eb1) != hipSuccess) {
        delete c; return fail(GSR_E_HIP, "hipEventCreate failed");
    }
    // Environment knobs (all of them): none changes a result, tests/test_icp_gpu.py::test_icp_knobs_change_nothing
    if (const char* e = getenv("GSR_ICP_DEVICE_LOOP")) c->device_loop = atoi(e) != 0;
    if (const char* e = getenv("GSR_ICP_BLOCK_SEARCH")) c->block_search = atoi(e) != 0;
    if (const char* e = getenv("GSR_ICP_TILE")) c->tile_search = atoi(e);
    (void)hipFuncSetAttribute((const void*)k_icp_nn_tile<true>, hipFuncAttributeMaxDynamicSharedMemorySize, ICP_TILE_LDS);
    (void)hipFuncSetAttribute((const void*)k_icp_nn_tile<false>, hipFuncAttributeMaxDynamicSharedMemorySize, ICP_TILE_LDS);
    if (const char* e = getenv("GSR_ICP_XCD")) c->xcd_ranges = atoi(e) != 0;
    if (const char* e = getenv("GSR_ICP_ROBUST_BOX")) c->robust_allowed = atoi(e) != 0;
    if (const char* e = getenv("GSR_ICP_ADAPT")) c->adapt_cells = atoi(e) != 0;
    if (const char* e = getenv("GSR_ICP_PERSISTENT")) c->persistent = atoi(e) != 0;
    if (const char* e = getenv("GSR_ICP_FUSED_STEP")) c->fused_step = atoi(e) != 0;
    if (const char* e = getenv("GSR_ICP_BLOCKS")) { int v = atoi(e); if (v >= 1 && v <= 65536) c->nblocks = v; }
    // pinned, device-mapped, COHERENT host memory: the device's system-scope stores must reach the host while the stream is still
    // running (a non-coherent mapping would only show them at the end of the kernel).  GSR_ICP_RB_POLL=0: no polling at all
    bool poll = true;
    if (const char* e = getenv("GSR_ICP_RB_POLL")) poll = atoi(e) != 0;
    if (poll && hipHostMalloc((void**)&c->host_rb, 1024 + 64, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) memset(c->host_rb, 0, 1024 + 64);
    else { c->host_rb = nullptr; (void)hipGetLastError(); }
    if (const char* e = getenv("GSR_ICP_NN_KERNEL")) c->nn_kernel = atoi(e);
    if (const char* e = getenv("GSR_ICP_CELL_TARGET")) { double v = atof(e); if (v > 0.01 && v < 1000) c->cell_target = v; }
    if (const char* e = getenv("GSR_ICP_MAX_CELLS")) { int v = atoi(e); if (v >= 1024) c->max_cells = v; }
    *out = c;
    return GSR_OK;
}

int32_t gsr_icp_destroy(gsr_icp_ctx* c) {
    if (!c) return GSR_OK;
    (void)hipSetDevice(c->device);
    DevBuf* all[] = {&c->hist, &c->ticket, &c->src_raw, &c->src_order, &c->state, &c->nn_j, &c->Tc, &c->Sc, &c->stage_cov, &c->Ti, &c->Tg, &c->Si, &c->bbox, &c->keys, &c->idx, &c->skeys, &c->order, &c->cellStart, &c->Tq, &c->Tn, &c->stage_xyz, &c->stage_nrm,
                     &c->src, &c->partials, &c->acc_dev, &c->rocprim_tmp, &c->corr_idx, &c->corr_d2};
    for (DevBuf* b : all) b->release();
    if (c->e0) (void)hipEventDestroy(c->e0);
    if (c->e1) (void)hipEventDestroy(c->e1);
    if (c->eb0) (void)hipEventDestroy(c->eb0);
    if (c->eb1) (void)hipEventDestroy(c->eb1);
    if (c->host_rb) (void)hipHostFree(c->host_rb);
    delete c;
    return GSR_OK;
}

int32_t gsr_icp_set_target(gsr_icp_ctx* c, const float* xyz, const double* normals, int64_t n, double max_corr, int32_t on_device) {
    if (!c) return fail(GSR_E_INVALID, "gsr_icp_set_target: NULL context");
    if (!(max_corr > 0)) return fail(GSR_E_PRECONDITION, "max_correspondence_distance must be > 0 (got %g)", max_corr);
    if (n <= 0 || !xyz) return fail(GSR_E_PRECONDITION, "gsr_icp_set_target: empty target cloud");
    if (n >= ((int64_t)1 << 31) - 1) return fail(GSR_E_INVALID, "gsr_icp_set_target: n too large");
    GSR_HIP(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    GSR_HIP(hipEventRecord(c->eb0, st));
    const float* dxyz = xyz;
    const double* dnrm = normals;
    if (!on_device) {
        GSR_TRY(c->stage_xyz.reserve((size_t)n * 12));
        GSR_HIP(hipMemcpyAsync(c->stage_xyz.p, xyz, (size_t)n * 12, hipMemcpyHostToDevice, st));
        dxyz = c->stage_xyz.as<float>();
        if (normals) {
            GSR_TRY(c->stage_nrm.reserve((size_t)n * 24));
            GSR_HIP(hipMemcpyAsync(c->stage_nrm.p, normals, (size_t)n * 24, hipMemcpyHostToDevice, st));
            dnrm = c->stage_nrm.as<double>();
        }
    }
    // bounding box of the finite points
    const int nbb = stride_grid(n);
    GSR_TRY(c->bbox.reserve(64 + (size_t)nbb * 24));
    hipLaunchKernelGGL(k_icp_bbox, dim3(nbb), dim3(256), 0, st, n, dxyz, c->bbox.as<float>() + 16);
    hipLaunchKernelGGL(k_icp_bbox_reduce, dim3(1), dim3(256), 0, st, nbb, c->bbox.as<float>() + 16, c->bbox.as<float>());
    // the robust candidate of the raw box: header floats [0..5] raw, [6..11] trimmed, [12] "the raw box is far too large"
    GSR_TRY(c->hist.reserve(3 * ICP_HIST_BINS * 4));
    GSR_HIP(hipMemsetAsync(c->hist.p, 0, 3 * ICP_HIST_BINS * 4, st));
    hipLaunchKernelGGL(k_icp_hist, dim3(nbb > 512 ? 512 : nbb), dim3(256), 0, st, n, dxyz, c->bbox.as<float>(), c->hist.as<unsigned>());
    hipLaunchKernelGGL(k_icp_trim, dim3(1), dim3(64), 0, st, c->hist.as<unsigned>(), c->bbox.as<float>(), c->bbox.as<float>() + 6, c->bbox.as<float>() + 12);
    float hb[13];
    GSR_TRY(icp_fetch(c, c->bbox.p, hb, 13 * 4));
    const float raw_box[6] = {hb[0], hb[1], hb[2], hb[3], hb[4], hb[5]};
    c->robust_box = hb[12] != 0.0f && c->robust_allowed;
    if (c->robust_box) {        // far outliers: one refinement at the resolution of the trimmed range, then the grid goes over THAT
        GSR_HIP(hipMemsetAsync(c->hist.p, 0, 3 * ICP_HIST_BINS * 4, st));
        hipLaunchKernelGGL(k_icp_hist, dim3(nbb > 512 ? 512 : nbb), dim3(256), 0, st, n, dxyz, c->bbox.as<float>() + 6, c->hist.as<unsigned>());
        hipLaunchKernelGGL(k_icp_trim, dim3(1), dim3(64), 0, st, c->hist.as<unsigned>(), c->bbox.as<float>() + 6, c->bbox.as<float>(), c->bbox.as<float>() + 12);
        GSR_TRY(icp_fetch(c, c->bbox.p, hb, 6 * 4));
    }
    double mn[3], mx[3];
    for (int k = 0; k < 3; ++k) { mn[k] = hb[k]; mx[k] = hb[3 + k]; }
    IcpGrid g;
    if (!(mx[0] >= mn[0])) { mn[0] = mn[1] = mn[2] = 0; mx[0] = mx[1] = mx[2] = 0; }
    // cell: about two target points per cell, but no finer than max_corr/8 (bounds the ring count)
    double cell;
    {
        const double ex = mx[0] - mn[0], ey = mx[1] - mn[1], ez = mx[2] - mn[2];
        const double emax = fmax(ex, fmax(ey, ez)), eps = emax * 1e-6 + 1e-30;
        cell = cbrt((ex + eps) * (ey + eps) * (ez + eps) * c->cell_target / (double)n);
        if (!(cell > 0)) cell = max_corr;
        // (round 6: max_corr / 64, was / 8.  The floor bounds the ring count of a query that finds nothing; with / 8 the reference's DEFAULT
        // max_correspondence of 5 scene units put 450 points into every cell of a 1 M-splat cloud and an iteration took 4.9 ms instead of
        // 0.15, scripts/icp_default_params.py.  A query beyond the target's box + max_corr leaves before the first ring, icp_nearest)
        if (cell < max_corr / ICP_CELL_FLOOR_DIV) cell = max_corr / ICP_CELL_FLOOR_DIV;
    }
    GSR_TRY(c->keys.reserve(n * 4)); GSR_TRY(c->idx.reserve(n * 4)); GSR_TRY(c->skeys.reserve(n * 4)); GSR_TRY(c->order.reserve(n * 4));
    c->occupancy = 0.0;
    for (int attempt = 0;; ++attempt) {
        for (;;) {
            double fx = floor((mx[0] - mn[0]) / cell) + 1, fy = floor((mx[1] - mn[1]) / cell) + 1, fz = floor((mx[2] - mn[2]) / cell) + 1;
            if (fx * fy * fz <= (double)c->max_cells) { g.gx = (int)fx; g.gy = (int)fy; g.gz = (int)fz; break; }
            cell *= 1.2599210498948732;
        }
        g.ox = mn[0]; g.oy = mn[1]; g.oz = mn[2];
        g.c = cell; g.inv_c = 1.0 / cell;
        g.cx = 0.5 * (mn[0] + mx[0]); g.cy = 0.5 * (mn[1] + mx[1]); g.cz = 0.5 * (mn[2] + mx[2]);
        g.ncells = g.gx * g.gy * g.gz;
        g.rings = (int)ceil(max_corr / cell);
        if (g.rings < 1) g.rings = 1;
        g.bx0 = raw_box[0]; g.by0 = raw_box[1]; g.bz0 = raw_box[2]; g.bx1 = raw_box[3]; g.by1 = raw_box[4]; g.bz1 = raw_box[5];
        if (!(raw_box[3] >= raw_box[0])) { g.bx0 = g.by0 = g.bz0 = -1.0 / 0.0; g.bx1 = g.by1 = g.bz1 = 1.0 / 0.0; }      // (no finite point: no box, no early leave)
        hipLaunchKernelGGL(k_icp_keys, dim3(stride_grid(n)), dim3(256), 0, st, n, dxyz, g, c->keys.as<unsigned>(), c->idx.as<unsigned>());
        int bits = 1;
        while (bits < 32 && ((int64_t)1 << bits) < g.ncells) ++bits;
        size_t bytes = 0;
        GSR_HIP(rocprim::radix_sort_pairs<icp_sort_cfg>(nullptr, bytes, c->keys.as<unsigned>(), c->skeys.as<unsigned>(), c->idx.as<unsigned>(),
                                          c->order.as<unsigned>(), (size_t)n, 0u, (unsigned)bits, st));
        GSR_TRY(c->rocprim_tmp.reserve(bytes));
        GSR_HIP(rocprim::radix_sort_pairs<icp_sort_cfg>(c->rocprim_tmp.p, bytes, c->keys.as<unsigned>(), c->skeys.as<unsigned>(), c->idx.as<unsigned>(),
                                          c->order.as<unsigned>(), (size_t)n, 0u, (unsigned)bits, st));
        GSR_TRY(c->cellStart.reserve(((size_t)g.ncells + 1) * 4));
        hipLaunchKernelGGL(k_icp_cell_starts, dim3(stride_grid(n)), dim3(256), 0, st, n, c->skeys.as<unsigned>(), (int64_t)g.ncells, c->cellStart.as<int>());
        if (!c->adapt_cells || attempt >= 2) break;
        // Clustered clouds: the box-volume rule gives two points per cell ON AVERAGE OVER THE BOX, and a scene whose points sit in
        // clumps has hundreds in the cells that matter.  When the average point shares its cell with more than ICP_OCC_MAX others the
        // grid is rebuilt finer (GSR_ICP_ADAPT=1; measured).
        GSR_HIP(hipMemsetAsync(c->hist.p, 0, 8, st));
        hipLaunchKernelGGL(k_icp_occupancy, dim3(stride_grid(g.ncells)), dim3(256), 0, st, (int64_t)g.ncells, c->cellStart.as<int>(), c->hist.as<unsigned long long>());
        unsigned long long sumsq = 0;
        GSR_TRY(icp_fetch(c, c->hist.p, &sumsq, 8));
        c->occupancy = (double)sumsq / (double)n;
        const double floor_cell = max_corr / ICP_CELL_FLOOR_DIV;
        if (!(c->occupancy > 12.0) || cell <= floor_cell * 1.0001 || (double)g.ncells * 1.9 > (double)c->max_cells) break;
        double f = cbrt(4.0 / c->occupancy);
        f = f < 0.4 ? 0.4 : f;
        cell = fmax(cell * f, floor_cell);
    }
    c->grid = g;
    GSR_TRY(c->Tq.reserve((size_t)n * 16));
    if (normals) GSR_TRY(c->Tn.reserve((size_t)n * 24));
    hipLaunchKernelGGL(k_icp_gather_target, dim3(stride_grid(n)), dim3(256), 0, st, n, c->order.as<unsigned>(), dxyz, dnrm,
                       c->Tq.as<float4>(), normals ? c->Tn.as<double>() : (double*)nullptr);
    GSR_HIP(hipEventRecord(c->eb1, st));
    if (c->defer_sync) c->build_pending = true;      // (gsr_icp_register_clouds: the registration behind this waits for the stream)
    else {
        GSR_HIP(hipStreamSynchronize(st));
        (void)hipEventElapsedTime(&c->ms_build, c->eb0, c->eb1);
        c->build_pending = false;
    }
    GSR_HIP(hipGetLastError());                      // the launches of the index build
    c->nt = n; c->max_corr = max_corr; c->have_target = true; c->have_normals = normals != nullptr;
    c->have_tcov = false; c->have_scov = false; c->have_tcol = false; c->have_scol = false;
    c->have_source = false;              // the source is sorted by the target grid: set it again after a new target
    return GSR_OK;
}

int32_t gsr_icp_set_source(gsr_icp_ctx* c, const float* xyz, int64_t n, int32_t on_device) {
    if (!c) return fail(GSR_E_INVALID, "gsr_icp_set_source: NULL context");
    if (n < 0 || (n > 0 && !xyz)) return fail(GSR_E_PRECONDITION, "gsr_icp_set_source: empty source cloud");
    if (n >= ((int64_t)1 << 31) - 1) return fail(GSR_E_INVALID, "gsr_icp_set_source: n too large");
    GSR_HIP(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    GSR_TRY(c->src.reserve((size_t)(n > 0 ? n : 1) * 12));
    c->src_sorted = false;
    if (n == 0) {
        // an empty shard of a multi-GPU source split (fewer points than ranks): it contributes zero sums and still joins
        // every collective; a registration without an all-reduce refuses it (gsr_icp_register)
    } else if (c->have_target) {
        // sort the source by the TARGET grid's cell (rigid motions keep neighbours neighbours): lanes of a
        // wave then walk the same target cells, and the per-point ring searches share cache lines
        const float* raw = xyz;                       // device arrays are read in place (this call returns synchronised)
        if (!on_device) {
            GSR_TRY(c->src_raw.reserve((size_t)n * 12));
            GSR_HIP(hipMemcpyAsync(c->src_raw.p, xyz, (size_t)n * 12, hipMemcpyHostToDevice, st));
            raw = c->src_raw.as<float>();
        }
        GSR_TRY(c->keys.reserve(n * 4)); GSR_TRY(c->idx.reserve(n * 4)); GSR_TRY(c->skeys.reserve(n * 4)); GSR_TRY(c->src_order.reserve(n * 4));
        hipLaunchKernelGGL(k_icp_keys, dim3(stride_grid(n)), dim3(256), 0, st, n, raw, c->grid, c->keys.as<unsigned>(), c->idx.as<unsigned>());
        int bits = 1;
        while (bits < 32 && ((int64_t)1 << bits) < c->grid.ncells) ++bits;
        size_t bytes = 0;
        GSR_HIP(rocprim::radix_sort_pairs<icp_sort_cfg>(nullptr, bytes, c->keys.as<unsigned>(), c->skeys.as<unsigned>(), c->idx.as<unsigned>(),
                                          c->src_order.as<unsigned>(), (size_t)n, 0u, (unsigned)bits, st));
        GSR_TRY(c->rocprim_tmp.reserve(bytes));
        GSR_HIP(rocprim::radix_sort_pairs<icp_sort_cfg>(c->rocprim_tmp.p, bytes, c->keys.as<unsigned>(), c->skeys.as<unsigned>(), c->idx.as<unsigned>(),
                                          c->src_order.as<unsigned>(), (size_t)n, 0u, (unsigned)bits, st));
        hipLaunchKernelGGL(k_icp_gather_source, dim3(stride_grid(n)), dim3(256), 0, st, n, c->src_order.as<unsigned>(), raw, c->src.as<float>());
        c->src_sorted = true;
    } else {
        GSR_HIP(hipMemcpyAsync(c->src.p, xyz, (size_t)n * 12, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
    }
    if (!c->defer_sync) GSR_HIP(hipStreamSynchronize(st));
    c->ns = n; c->have_source = true; c->have_scov = false; c->have_scol = false;
    if (!c->allreduce && !c->allreduce_dev && !c->comm) c->ns_global = n;
    return GSR_OK;
}

static int32_t set_cov(gsr_icp_ctx* c, const double* cov6, int64_t n, const unsigned* order, DevBuf& dst, int32_t on_device) {
    if (n == 0) return dst.reserve(48);               // an empty shard of a multi-GPU source split: nothing to gather
    GSR_HIP(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const double* in = cov6;
    if (!on_device) {
        GSR_TRY(c->stage_cov.reserve((size_t)n * 48));
        GSR_HIP(hipMemcpyAsync(c->stage_cov.p, cov6, (size_t)n * 48, hipMemcpyHostToDevice, st));
        in = c->stage_cov.as<double>();
    }
    GSR_TRY(dst.reserve((size_t)n * 48));
    hipLaunchKernelGGL(k_icp_gather_cov, dim3(stride_grid(6 * n)), dim3(256), 0, st, n, order, in, dst.as<double>());
    GSR_HIP(hipStreamSynchronize(st));
    return GSR_OK;
}
int32_t gsr_icp_set_target_cov(gsr_icp_ctx* c, const double* cov6, int32_t on_device) {
    if (!c || !cov6) return fail(GSR_E_INVALID, "gsr_icp_set_target_cov: NULL argument");
    if (!c->have_target) return fail(GSR_E_PRECONDITION, "gsr_icp_set_target_cov: set the target first");
    GSR_TRY(set_cov(c, cov6, c->nt, c->order.as<unsigned>(), c->Tc, on_device));
    c->have_tcov = true;
    return GSR_OK;
}
int32_t gsr_icp_set_source_cov(gsr_icp_ctx* c, const double* cov6, int32_t on_device) {
    if (!c || (!cov6 && !(c->have_source && c->ns == 0))) return fail(GSR_E_INVALID, "gsr_icp_set_source_cov: NULL argument");
    if (!c->have_source) return fail(GSR_E_PRECONDITION, "gsr_icp_set_source_cov: set the source first");
    GSR_TRY(set_cov(c, cov6, c->ns, c->src_sorted ? c->src_order.as<unsigned>() : (const unsigned*)nullptr, c->Sc, on_device));
    c->have_scov = true;
    return GSR_OK;
}

int32_t gsr_icp_set_target_color(gsr_icp_ctx* c, const double* rgb, int32_t on_device) {
    if (!c || !rgb) return fail(GSR_E_INVALID, "gsr_icp_set_target_color: NULL argument");
    if (!c->have_target) return fail(GSR_E_PRECONDITION, "gsr_icp_set_target_color: set the target first");
    if (!c->have_normals) return fail(GSR_E_PRECONDITION, "ColoredICP requires pre-computed normal vectors for target PointCloud.");
    GSR_HIP(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const int64_t n = c->nt;
    const double* in = rgb;
    if (!on_device) {
        GSR_TRY(c->stage_cov.reserve((size_t)n * 24));
        GSR_HIP(hipMemcpyAsync(c->stage_cov.p, rgb, (size_t)n * 24, hipMemcpyHostToDevice, st));
        in = c->stage_cov.as<double>();
    }
    GSR_TRY(c->Ti.reserve((size_t)n * 8)); GSR_TRY(c->Tg.reserve((size_t)n * 24));
    hipLaunchKernelGGL(k_icp_intensity, dim3(stride_grid(n)), dim3(256), 0, st, n, c->order.as<unsigned>(), in, c->Ti.as<double>());
    // InitializePointCloudForColoredICP: KDTreeSearchParamHybrid(max_distance * 2, 30)
    hipLaunchKernelGGL(k_icp_color_gradient, dim3(stride_grid(n)), dim3(256), 0, st, n, c->grid, c->cellStart.as<int>(), c->Tq.as<float4>(),
                       c->Tn.as<double>(), c->Ti.as<double>(), c->max_corr * 2.0, c->Tg.as<double>());
    GSR_HIP(hipStreamSynchronize(st));
    c->have_tcol = true;
    return GSR_OK;
}
int32_t gsr_icp_set_source_color(gsr_icp_ctx* c, const double* rgb, int32_t on_device) {
    if (!c || (!rgb && !(c->have_source && c->ns == 0))) return fail(GSR_E_INVALID, "gsr_icp_set_source_color: NULL argument");
    if (!c->have_source) return fail(GSR_E_PRECONDITION, "gsr_icp_set_source_color: set the source first");
    GSR_HIP(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const int64_t n = c->ns;
    if (n == 0) { GSR_TRY(c->Si.reserve(8)); c->have_scol = true; return GSR_OK; }      // an empty shard of a multi-GPU source split
    const double* in = rgb;
    if (!on_device) {
        GSR_TRY(c->stage_cov.reserve((size_t)n * 24));
        GSR_HIP(hipMemcpyAsync(c->stage_cov.p, rgb, (size_t)n * 24, hipMemcpyHostToDevice, st));
        in = c->stage_cov.as<double>();
    }
    GSR_TRY(c->Si.reserve((size_t)n * 8));
    hipLaunchKernelGGL(k_icp_intensity, dim3(stride_grid(n)), dim3(256), 0, st, n, c->src_sorted ? c->src_order.as<unsigned>() : (const unsigned*)nullptr,
                       in, c->Si.as<double>());
    GSR_HIP(hipStreamSynchronize(st));
    c->have_scol = true;
    return GSR_OK;
}
int32_t gsr_icp_set_lambda_geometric(gsr_icp_ctx* c, double lambda_geometric) {
    if (!c) return fail(GSR_E_INVALID, "gsr_icp_set_lambda_geometric: NULL context");
    if (!(lambda_geometric >= 0.0 && lambda_geometric <= 1.0)) return fail(GSR_E_INVALID, "lambda_geometric must lie in [0, 1]");
    c->lambda_geometric = lambda_geometric;
    return GSR_OK;
}
int32_t gsr_icp_get_color_gradient(gsr_icp_ctx* c, double* out) {
    if (!c || !out) return fail(GSR_E_INVALID, "gsr_icp_get_color_gradient: NULL argument");
    if (!c->have_tcol) return fail(GSR_E_PRECONDITION, "gsr_icp_get_color_gradient: set the target colours first");
    GSR_HIP(hipSetDevice(c->device));
    // back to the caller's point order (host memory)
    std::vector<double> sorted((size_t)c->nt * 3);
    std::vector<unsigned> order((size_t)c->nt);
    GSR_HIP(hipMemcpy(sorted.data(), c->Tg.p, (size_t)c->nt * 24, hipMemcpyDeviceToHost));
    GSR_HIP(hipMemcpy(order.data(), c->order.p, (size_t)c->nt * 4, hipMemcpyDeviceToHost));
    for (int64_t j = 0; j < c->nt; ++j)
        for (int a = 0; a < 3; ++a) out[3 * (size_t)order[j] + a] = sorted[3 * j + a];
    return GSR_OK;
}

int32_t gsr_icp_set_allreduce_dev(gsr_icp_ctx* c, gsr_allreduce_dev64_fn fn, void* user, int64_t n_source_global) {
    if (!c) return fail(GSR_E_INVALID, "gsr_icp_set_allreduce_dev: NULL context");
    c->allreduce_dev = fn; c->allreduce_dev_user = user;
    if (fn) { c->allreduce = nullptr; c->allreduce_user = nullptr; c->comm = nullptr; }
    c->ns_global = fn ? n_source_global : c->ns;
    return GSR_OK;
}
int32_t gsr_icp_set_comm(gsr_icp_ctx* c, gsr_comm* comm, int64_t n_source_global) {
    if (!c) return fail(GSR_E_INVALID, "gsr_icp_set_comm: NULL context");
    c->comm = comm;
    if (comm) { c->allreduce = nullptr; c->allreduce_user = nullptr; c->allreduce_dev = nullptr; c->allreduce_dev_user = nullptr; }
    c->ns_global = comm ? n_source_global : c->ns;
    return GSR_OK;
}
int32_t gsr_icp_set_allreduce(gsr_icp_ctx* c, gsr_allreduce_fn fn, void* user, int64_t n_source_global) {
    if (!c) return fail(GSR_E_INVALID, "gsr_icp_set_allreduce: NULL context");
    c->allreduce = fn; c->allreduce_user = user;
    if (fn) { c->allreduce_dev = nullptr; c->allreduce_dev_user = nullptr; c->comm = nullptr; }
    c->ns_global = fn ? n_source_global : c->ns;
    return GSR_OK;
}

int32_t gsr_icp_accumulate(gsr_icp_ctx* c, const double* T, int32_t kind, int32_t loss, double k, double* acc) {
    if (!c || !T || !acc) return fail(GSR_E_INVALID, "gsr_icp_accumulate: NULL argument");
    GSR_HIP(hipSetDevice(c->device));
    return run_accumulate(c, T, kind, loss, k, acc, false);
}

int32_t gsr_icp_register(gsr_icp_ctx* c, const double* init_T, int32_t kind, int32_t loss, double k, double rel_fitness,
                         double rel_rmse, int32_t max_iter, double* out_T, double* fitness, double* inlier_rmse, int32_t* iterations) {
    if (!c || !init_T || !out_T) return fail(GSR_E_INVALID, "gsr_icp_register: NULL argument");
    GSR_HIP(hipSetDevice(c->device));
    c->ms_iter = 0; c->n_iter_kernels = 0;
    const bool multi = c->allreduce_dev != nullptr || c->comm != nullptr;      // a device-side all-reduce per iteration
    if (!c->allreduce && !multi && c->have_source && c->ns == 0)
        return fail(GSR_E_PRECONDITION, "gsr_icp_register: empty source cloud");
    if (!c->allreduce && (c->device_loop || multi)) {
        // device-resident loop: no per-iteration host round trip.  With a device all-reduce (multi-GPU source split) the
        // only addition per iteration is one stream-ordered collective on 32 doubles between the reduction and the solve.
        if (!c->have_target || !c->have_source) return fail(GSR_E_INVALID, "icp: target and source must be set first");
        if (kind < GSR_ICP_POINT_TO_POINT || kind > GSR_ICP_COLORED) return fail(GSR_E_INVALID, "icp: unknown estimation kind %d", kind);
        if (kind == GSR_ICP_POINT_TO_PLANE && !c->have_normals)
            return fail(GSR_E_PRECONDITION, "TransformationEstimationPointToPlane requires target normals");
        if (kind == GSR_ICP_COLORED && (!c->have_normals || !c->have_tcol || !c->have_scol))
            return fail(GSR_E_PRECONDITION, "ColoredICP requires target normals and the colours of both clouds");
        if (kind == GSR_ICP_GENERALIZED && (!c->have_tcov || !c->have_scov))
            return fail(GSR_E_PRECONDITION, "TransformationEstimationForGeneralizedICP requires source and target covariances");
        hipStream_t st = c->stream;
        IcpState hs;
        memset(&hs, 0, sizeof(hs));
        memcpy(hs.T, init_T, sizeof(hs.T));
        hs.ctr[0] = c->grid.cx; hs.ctr[1] = c->grid.cy; hs.ctr[2] = c->grid.cz;
        hs.nsg = (double)(multi ? c->ns_global : c->ns); hs.rel_fit = rel_fitness; hs.rel_rmse = rel_rmse;
        hs.max_iter = max_iter < 0 ? 0 : max_iter; hs.kind = kind;
        GSR_TRY(c->state.reserve(sizeof(IcpState)));
        GSR_HIP(hipMemcpyAsync(c->state.p, &hs, sizeof(hs), hipMemcpyHostToDevice, st));
        const int nb = icp_blocks(c->ns, c->nblocks);
        GSR_TRY(c->partials.reserve((size_t)nb * GSR_ICP_ACC_LEN * 8));
        GSR_TRY(c->nn_j.reserve((size_t)(c->ns > 0 ? c->ns : 1) * 4));
        GSR_TRY(c->acc_dev.reserve(GSR_ICP_ACC_LEN * 8));
        GSR_TRY(c->ticket.reserve(64));
        GSR_HIP(hipMemsetAsync(c->partials.p, 0, (size_t)nb * GSR_ICP_ACC_LEN * 8, st));
        GSR_HIP(hipMemsetAsync(c->ticket.p, 0, 64, st));
        // what the accumulate kernel's last workgroup does: the whole step (single GPU), the rank-local fold (multi-GPU), nothing
        const int fuse = !c->fused_step ? 0 : (multi ? 2 : 1);
        const double mc2 = c->max_corr * c->max_corr;
        const ColorArgs cargs = {c->Ti.as<double>(), c->Tg.as<double>(), c->Si.as<double>(), sqrt(c->lambda_geometric), sqrt(1.0 - c->lambda_geometric)};
        const int total_evals = hs.max_iter + 1;
        // the first 27 cells of the search as one block of batched loads when max_corr spans two or more cells (icp_nearest)
        const bool blockf = c->block_search && c->grid.rings >= 2;
        int issued = 0;
        GSR_HIP(hipEventRecord(c->e0, st));
        // GSR_ICP_PERSISTENT=1: single GPU, fused search (the levels below 4 * 10^5 source points): ONE resident kernel runs the whole loop
        if (c->persistent && !multi && c->nn_mode() == 0 && c->ns > 0) {
            int32_t pr = 0;
            const double* tn = c->Tn.as<double>();
            const double* nul = nullptr;
            if (kind == GSR_ICP_POINT_TO_POINT) pr = blockf ? launch_persistent<0, 1>(c, nb, nul, nul, cargs, mc2, loss, k) : launch_persistent<0, 0>(c, nb, nul, nul, cargs, mc2, loss, k);
            else if (kind == GSR_ICP_POINT_TO_PLANE) pr = blockf ? launch_persistent<1, 1>(c, nb, tn, nul, cargs, mc2, loss, k) : launch_persistent<1, 0>(c, nb, tn, nul, cargs, mc2, loss, k);
            if (pr < 0) return pr;
            if (pr == 1) {
                GSR_TRY(icp_fetch(c, c->state.p, &hs, sizeof(hs)));
                unsigned tk[4] = {0, 0, 0, 0};
                GSR_TRY(icp_fetch(c, c->ticket.p, tk, sizeof(tk)));
                if (tk[2]) return fail(GSR_E_HIP, "gsr_icp_register: the resident kernel's barrier ran out of patience (GSR_ICP_PERSISTENT=0 takes the launch-per-iteration loop)");
                issued = total_evals;
            }
        }
        while (issued < total_evals) {
            const int chunk = total_evals - issued < 8 ? total_evals - issued : 8;
            const int* nnj = c->nn_mode() ? c->nn_j.as<int>() : (const int*)nullptr;
            for (int i = 0; i < chunk; ++i) {
#define GSR_ICP_ACC1(KIND, BLK, FUSE, TN, SC)                                                                                        \
    hipLaunchKernelGGL((k_icp_accumulate_dev<KIND, BLK, FUSE>), dim3(FUSE == 0 ? 8 * ((nb + 7) / 8) : nb), dim3(256), 0, st, c->ns, c->src.as<float>(), \
                       c->state.as<IcpState>(), c->grid, c->cellStart.as<int>(), nnj, c->Tq.as<float4>(), TN, SC, cargs, mc2, loss, k,          \
                       c->partials.as<double>(), c->ticket.as<unsigned>(), c->acc_dev.as<double>(), c->xcd_ranges ? nb : -nb)
#define GSR_ICP_ACC(KIND, TN, SC)                                                                                                    \
    do {                                                                                                                             \
        if (blockf) { if (fuse == 0) GSR_ICP_ACC1(KIND, 1, 0, TN, SC); else if (fuse == 1) GSR_ICP_ACC1(KIND, 1, 1, TN, SC); else GSR_ICP_ACC1(KIND, 1, 2, TN, SC); } \
        else { if (fuse == 0) GSR_ICP_ACC1(KIND, 0, 0, TN, SC); else if (fuse == 1) GSR_ICP_ACC1(KIND, 0, 1, TN, SC); else GSR_ICP_ACC1(KIND, 0, 2, TN, SC); } \
    } while (0)
                if (c->nn_mode()) launch_icp_nn<true>(c, st, Xform(), c->state.as<IcpState>(), mc2, blockf);
                if (kind == GSR_ICP_COLORED) GSR_ICP_ACC(3, c->Tn.as<double>(), (const double*)nullptr);
                else if (kind == GSR_ICP_POINT_TO_POINT) GSR_ICP_ACC(0, (const double*)nullptr, (const double*)nullptr);
                else if (kind == GSR_ICP_POINT_TO_PLANE) GSR_ICP_ACC(1, c->Tn.as<double>(), (const double*)nullptr);
                else GSR_ICP_ACC(2, c->Tc.as<double>(), c->Sc.as<double>());
#undef GSR_ICP_ACC
#undef GSR_ICP_ACC1
                if (multi) {
                    if (fuse == 0)
                        hipLaunchKernelGGL(k_icp_reduce, dim3(1), dim3(1024), 0, st, nb, c->partials.as<double>(), c->state.as<IcpState>(), c->acc_dev.as<double>());
                    if (c->comm) {                            // RCCL enqueued on this stream: no host involvement
                        GSR_TRY(gsr_comm_allreduce(c->comm, c->acc_dev.p, GSR_ICP_ACC_LEN, GSR_DT_F64, GSR_OP_SUM, (void*)st));
                    } else {
                        const int32_t rc = c->allreduce_dev(c->acc_dev.p, GSR_ICP_ACC_LEN, c->allreduce_dev_user);
                        if (rc != 0) return fail(GSR_E_INVALID, "icp: device all-reduce callback returned %d", rc);
                    }
                    hipLaunchKernelGGL(k_icp_step, dim3(1), dim3(ICP_STEP_THREADS), 0, st, nb, c->partials.as<double>(), c->acc_dev.as<double>(), c->state.as<IcpState>());
                } else if (fuse == 0) {
                    hipLaunchKernelGGL(k_icp_step, dim3(1), dim3(ICP_STEP_THREADS), 0, st, nb, c->partials.as<double>(), (const double*)nullptr, c->state.as<IcpState>());
                }
            }
            issued += chunk;
            GSR_HIP(hipGetLastError());              // a failed launch of the chunk (configuration, LDS) is reported here, not as a hang
            GSR_TRY(icp_fetch(c, c->state.p, &hs, sizeof(hs)));
            if (hs.done) break;
        }
        GSR_HIP(hipEventRecord(c->e1, st));
        GSR_HIP(hipStreamSynchronize(st));
        (void)hipEventElapsedTime(&c->ms_iter, c->e0, c->e1);
        c->n_iter_kernels = hs.evals;
        memcpy(out_T, hs.T, sizeof(hs.T));
        if (fitness) *fitness = hs.fit;
        if (inlier_rmse) *inlier_rmse = hs.rmse;
        if (iterations) *iterations = hs.iters;
        return GSR_OK;
    }
    double T[16], acc[GSR_ICP_ACC_LEN], update[16];
    memcpy(T, init_T, sizeof(T));
    GSR_TRY(run_accumulate(c, T, kind, loss, k, acc, true));
    const double nsg = (double)(c->ns_global > 0 ? c->ns_global : c->ns);
    double fit = acc[0] > 0 ? acc[0] / nsg : 0.0, rmse = acc[0] > 0 ? sqrt(acc[1] / acc[0]) : 0.0;
    int it = 0;
    for (; it < max_iter; ++it) {
        const double ctr[3] = {c->grid.cx, c->grid.cy, c->grid.cz};
        estimate_update(ctr, kind, acc, update);
        mat4_mul(update, T, T);
        GSR_TRY(run_accumulate(c, T, kind, loss, k, acc, true));
        const double fit2 = acc[0] > 0 ? acc[0] / nsg : 0.0, rmse2 = acc[0] > 0 ? sqrt(acc[1] / acc[0]) : 0.0;
        const bool stop = fabs(fit - fit2) < rel_fitness && fabs(rmse - rmse2) < rel_rmse;
        fit = fit2; rmse = rmse2;
        if (stop) { ++it; break; }
    }
    memcpy(out_T, T, sizeof(T));
    if (fitness) *fitness = fit;
    if (inlier_rmse) *inlier_rmse = rmse;
    if (iterations) *iterations = it;
    return GSR_OK;
}

int32_t gsr_icp_correspondences(gsr_icp_ctx* c, const double* T, int64_t* idx, double* d2) {
    if (!c || !T || !idx || !d2) return fail(GSR_E_INVALID, "gsr_icp_correspondences: NULL argument");
    if (!c->have_target || !c->have_source) return fail(GSR_E_INVALID, "icp: target and source must be set first");
    GSR_HIP(hipSetDevice(c->device));
    GSR_TRY(c->corr_idx.reserve((size_t)c->ns * 8)); GSR_TRY(c->corr_d2.reserve((size_t)c->ns * 8));
    Xform X;
    for (int i = 0; i < 12; ++i) X.m[i] = T[i];
    GSR_TRY(c->nn_j.reserve((size_t)c->ns * 4));
    {   // (the ring loop per thread -- or, GSR_ICP_TILE=1 and a fine level, the tile search: the tests compare both with the oracle's KD-tree)
        const int keep = c->tile_search;
        if (keep != 1) c->tile_search = 0;
        launch_icp_nn<false>(c, c->stream, X, (const IcpState*)nullptr, c->max_corr * c->max_corr, false);
        c->tile_search = keep;
    }
    hipLaunchKernelGGL(k_icp_correspond, dim3(stride_grid(c->ns)), dim3(256), 0, c->stream, c->ns, c->src.as<float>(), X, c->nn_j.as<int>(),
                       c->Tq.as<float4>(), c->src_sorted ? c->src_order.as<unsigned>() : (const unsigned*)nullptr, c->corr_idx.as<int64_t>(),
                       c->corr_d2.as<double>());
    GSR_HIP(hipMemcpyAsync(idx, c->corr_idx.p, (size_t)c->ns * 8, hipMemcpyDeviceToHost, c->stream));
    GSR_HIP(hipMemcpyAsync(d2, c->corr_d2.p, (size_t)c->ns * 8, hipMemcpyDeviceToHost, c->stream));
    GSR_HIP(hipStreamSynchronize(c->stream));
    return GSR_OK;
}

int32_t gsr_icp_solve(const double* acc, int32_t kind, const double* centre, double* update) {
    if (!acc || !update) return fail(GSR_E_INVALID, "gsr_icp_solve: NULL argument");
    if (kind < GSR_ICP_POINT_TO_POINT || kind > GSR_ICP_COLORED) return fail(GSR_E_INVALID, "gsr_icp_solve: unknown kind %d", kind);
    const double zero[3] = {0, 0, 0};
    estimate_update(centre ? centre : zero, kind, acc, update);
    return GSR_OK;
}

int32_t gsr_icp_get_centre(gsr_icp_ctx* c, double* centre3) {
    if (!c || !centre3 || !c->have_target) return fail(GSR_E_INVALID, "gsr_icp_get_centre: no target set");
    centre3[0] = c->grid.cx; centre3[1] = c->grid.cy; centre3[2] = c->grid.cz;
    return GSR_OK;
}

static void icp_read_build_ms(gsr_icp_ctx* c) {
    if (!c->build_pending) return;
    (void)hipEventSynchronize(c->eb1);
    (void)hipEventElapsedTime(&c->ms_build, c->eb0, c->eb1);
    c->build_pending = false;
}

// registration_icp(source, target, max_correspondence_distance, init, estimation, criteria) -- Open3D's own entry takes the two CLOUDS
// (local_registration_util.py:88-90): target index, source order and the iteration loop in one call, no return to the host language and
// no stream synchronisation between them (a Python caller spent ~0.2 ms per registration in those gaps: profiles/r06d_icp_timeline.txt).
int32_t gsr_icp_register_clouds(gsr_icp_ctx* c, const float* src_xyz, int64_t ns, const float* tgt_xyz, const double* tgt_normals, int64_t nt,
                                int32_t on_device, double max_corr, const double* init_T, int32_t kind, int32_t loss, double k, double rel_fitness,
                                double rel_rmse, int32_t max_iter, double* out_T, double* fitness, double* inlier_rmse, int32_t* iterations) {
    if (!c) return fail(GSR_E_INVALID, "gsr_icp_register_clouds: NULL context");
    if (kind != GSR_ICP_POINT_TO_POINT && kind != GSR_ICP_POINT_TO_PLANE)
        return fail(GSR_E_INVALID, "gsr_icp_register_clouds: point-to-point and point-to-plane only (covariances / colours: the step-by-step entry points)");
    if (kind == GSR_ICP_POINT_TO_PLANE && !tgt_normals) return fail(GSR_E_PRECONDITION, "TransformationEstimationPointToPlane requires target normals");
    // one process, one GPU: a communicator or an all-reduce callback of an earlier sharded call must not stay behind
    c->comm = nullptr; c->allreduce = nullptr; c->allreduce_user = nullptr; c->allreduce_dev = nullptr; c->allreduce_dev_user = nullptr;
    c->defer_sync = c->device_loop;                 // (the host-driven loop, GSR_ICP_DEVICE_LOOP=0, keeps the waits)
    int32_t r = gsr_icp_set_target(c, tgt_xyz, kind == GSR_ICP_POINT_TO_PLANE ? tgt_normals : nullptr, nt, max_corr, on_device);
    if (r == GSR_OK) r = gsr_icp_set_source(c, src_xyz, ns, on_device);
    c->defer_sync = false;
    if (r == GSR_OK) { c->ns_global = ns; r = gsr_icp_register(c, init_T, kind, loss, k, rel_fitness, rel_rmse, max_iter, out_T, fitness, inlier_rmse, iterations); }
    if (r != GSR_OK) (void)hipStreamSynchronize(c->stream);         // (the caller's arrays are free again whatever happened)
    icp_read_build_ms(c);
    return r;
}

// The coarse-to-fine schedule in one call (MultiScaleRegistratorMixture._register_main_point_clouds, qt_multiscale_registrator.py:197-236: entry k registers
// level list[-(k + 1)] of the two clouds from the transform entry k - 1 ended with): gsr_icp_register_clouds per entry, no return to the host language between
// the entries.  entries[k] describes entry k (coarsest first); results[k] receives its outcome; out_T the last entry's transform.
int32_t gsr_icp_register_multiscale(gsr_icp_ctx* c, int32_t n_entries, const gsr_icp_entry* entries, int32_t on_device, const double* init_T, int32_t kind,
                                    int32_t loss, double k, double rel_fitness, double rel_rmse, gsr_icp_entry_result* results, double* out_T) {
    if (!c || !init_T || !out_T || n_entries < 0 || (n_entries > 0 && (!entries || !results)))
        return fail(GSR_E_INVALID, "gsr_icp_register_multiscale: bad argument");
    double T[16];
    memcpy(T, init_T, sizeof(T));
    for (int e = 0; e < n_entries; ++e) {
        const gsr_icp_entry& E = entries[e];
        gsr_icp_entry_result& R = results[e];
        memset(&R, 0, sizeof(R));
        memcpy(R.init_T, T, sizeof(T));
        const int32_t r = gsr_icp_register_clouds(c, E.src_xyz, E.ns, E.tgt_xyz, E.tgt_normals, E.nt, on_device, E.max_corr, T, kind, loss, k, rel_fitness, rel_rmse,
                                                  E.max_iter, R.T, &R.fitness, &R.inlier_rmse, &R.iterations);
        if (r != GSR_OK) return r;
        R.ms_build = c->ms_build; R.ms_iters = c->ms_iter; R.evaluations = c->n_iter_kernels;
        memcpy(T, R.T, sizeof(T));
    }
    memcpy(out_T, T, sizeof(T));
    return GSR_OK;
}

int32_t gsr_icp_get_timing(gsr_icp_ctx* c, float* out3) {
    if (!c || !out3) return fail(GSR_E_INVALID, "gsr_icp_get_timing: NULL argument");
    icp_read_build_ms(c);
    out3[0] = c->ms_build; out3[1] = c->ms_iter; out3[2] = (float)c->n_iter_kernels;
    return GSR_OK;
}

int32_t gsr_normals_from_cov(const float* cov6, int64_t n, double* normals, int32_t on_device, int32_t device, void* stream) {
    if (n < 0 || (n > 0 && (!cov6 || !normals))) return fail(GSR_E_INVALID, "gsr_normals_from_cov: bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(GSR_E_NO_DEVICE, "gsr_normals_from_cov: no HIP device visible (this backend has no CPU fallback)");
    if (n == 0) return GSR_OK;
    GSR_HIP(hipSetDevice(device));
    hipStream_t st = (hipStream_t)stream;
    if (on_device) {
        hipLaunchKernelGGL(k_normals_from_cov, dim3(stride_grid(n)), dim3(256), 0, st, n, cov6, normals);
        GSR_HIP(hipStreamSynchronize(st));
        return GSR_OK;
    }
    DevBuf in, out;
    int32_t r = in.reserve((size_t)n * 24);
    if (r == GSR_OK) r = out.reserve((size_t)n * 24);
    if (r != GSR_OK) { in.release(); out.release(); return r; }
    hipError_t e = hipMemcpyAsync(in.p, cov6, (size_t)n * 24, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_normals_from_cov, dim3(stride_grid(n)), dim3(256), 0, st, n, in.as<float>(), out.as<double>());
        e = hipMemcpyAsync(normals, out.p, (size_t)n * 24, hipMemcpyDeviceToHost, st);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    in.release(); out.release();
    if (e != hipSuccess) return fail(GSR_E_HIP, "gsr_normals_from_cov: %s", hipGetErrorString(e));
    return GSR_OK;
}

int32_t gsr_cov_from_normals(const double* normals, int64_t n, double epsilon, double* cov6, int32_t on_device, int32_t device, void* stream) {
    if (n < 0 || (n > 0 && (!normals || !cov6))) return fail(GSR_E_INVALID, "gsr_cov_from_normals: bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(GSR_E_NO_DEVICE, "gsr_cov_from_normals: no HIP device visible (this backend has no CPU fallback)");
    if (n == 0) return GSR_OK;
    GSR_HIP(hipSetDevice(device));
    hipStream_t st = (hipStream_t)stream;
    if (on_device) {
        hipLaunchKernelGGL(k_cov_from_normals, dim3(stride_grid(n)), dim3(256), 0, st, n, normals, epsilon, cov6);
        GSR_HIP(hipStreamSynchronize(st));
        return GSR_OK;
    }
    DevBuf in, out;
    int32_t r = in.reserve((size_t)n * 24);
    if (r == GSR_OK) r = out.reserve((size_t)n * 48);
    if (r != GSR_OK) { in.release(); out.release(); return r; }
    hipError_t e = hipMemcpyAsync(in.p, normals, (size_t)n * 24, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_cov_from_normals, dim3(stride_grid(n)), dim3(256), 0, st, n, in.as<double>(), epsilon, out.as<double>());
        e = hipMemcpyAsync(cov6, out.p, (size_t)n * 48, hipMemcpyDeviceToHost, st);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    in.release(); out.release();
    if (e != hipSuccess) return fail(GSR_E_HIP, "gsr_cov_from_normals: %s", hipGetErrorString(e));
    return GSR_OK;
}

int32_t gsr_normals_knn(const float* xyz, int64_t n, int32_t knn, double* normals, int32_t on_device, int32_t device, void* stream) {
    if (n < 0 || (n > 0 && (!xyz || !normals))) return fail(GSR_E_INVALID, "gsr_normals_knn: bad argument");
    if (knn < 1 || knn > ICP_KNN) return fail(GSR_E_INVALID, "gsr_normals_knn: knn must lie in [1, %d] (got %d)", ICP_KNN, knn);
    if (n == 0) return GSR_OK;
    gsr_icp_ctx* c = nullptr;
    GSR_TRY(gsr_icp_create(&c, device, stream));
    // the grid of the ICP target index (about two points per cell); the correspondence distance plays no role here
    int32_t r = gsr_icp_set_target(c, xyz, nullptr, n, 1e-300, on_device);
    if (r == GSR_OK) {
        DevBuf out;
        r = out.reserve((size_t)n * 24);
        if (r == GSR_OK) {
            hipLaunchKernelGGL(k_knn_normals, dim3(stride_grid(n)), dim3(256), 0, c->stream, n, c->grid, c->cellStart.as<int>(), c->Tq.as<float4>(),
                               c->order.as<unsigned>(), (int)knn, out.as<double>());
            hipError_t e = hipMemcpyAsync(normals, out.p, (size_t)n * 24, on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            if (e != hipSuccess) r = fail(GSR_E_HIP, "gsr_normals_knn: %s", hipGetErrorString(e));
        }
        out.release();
    }
    (void)gsr_icp_destroy(c);
    return r;
}

}  // extern "C"
