// hem.hip -- Hierarchical-EM Gaussian-mixture level on MI355X (gfx950), behind include/gsr_hip.h.
//
// One call of gsr_hem_run_level is one Mixture::createClusterLevel of the reference
// (src/cpp_ext/src/mixture.cpp:66-285).  The reference walks an AoS vector with a hash grid whose
// cell is the LARGEST parent radius; here the level is laid out for the GPU instead:
//
//   level arrays (HBM, row-major, as the C ABI hands them over)
//       xyz[n][3] color[n][3] cov6[n][6] opacity[n] weight[n] sh[n][F] is_parent[n]
//   per-level working set, all in CELL-SORTED order (j = sorted position, order[j] = input index)
//       A[j] = {x, y, z, flags}   compact float4 array of all components (parents' positions, the irregular list of pass B)
//       Ac[k] = {x, y, z, (j << 2) | flags}   the NON-PARENT components only, in cell order: what stage 1 of k_select
//                                 streams (16 bytes per candidate, candidates of a cell row contiguous, 1 KiB per wave
//                                 instruction) + cellStartC[cells+1], the grid's prefix table counted over them.  A parent
//                                 can be a child of no parent but itself: it queues itself
//       geo[j][4]                 one 64-byte record per component (half a cache line, one round trip):
//                                 {x, y, z, flags} {c00, c01, c02, c11} {c12, c22, col_r, col_g} {col_b, opacity, weight, det}
//                                 -- what stage 2 of k_select, the parents' set-up and the M-step moments gather; after the
//                                 selection the det slot is overwritten with the child's sumLw (one gather less per pair)
//       shs[j][RSH]               SH rest, rows padded to whole float4 (RSH = F rounded up to 4)
//       Rs[j] query radius, cellStart[cells+1] prefix table of a dense uniform grid
//   pair list (parent-major CSR): pair_child[M] (sorted position), pair_wl[M] (w_s * clamp(L_si))
//
// Kernels of one level (DESIGN.md section 4 has the measurements):
//   k_prep            det, regular flag, one packed 64-byte record per component (input order), bounding box partials
//   k_hist / k_grid_params   robust grid box (0.1 % trimmed per side), ~16 components per cell
//   k_keys / sort     cell key per component, radix sort (rocPRIM) -> order[]
//   k_gather / k_gather_sh   the cell-sorted working set: A, geo, shs (+ parent radius: closed-form eigenvalue, f64 trig)
//   k_child_stream    Ac and cellStartC from A, the parent flags and their scan
//   k_spans           candidates every parent will scan (capacity of its output segment, LPT work estimate)
//   k_select<SPARSE|COUNT|FILL, WPB, QUEUE>   a wavefront takes four consecutive parents one after the other (SEL_NP) and keeps its
//                     two rings across them: grid rows clipped to the pre-reject ellipsoid,
//                     flattened candidate stream -> stage 1 (radius test, Mahalanobis pre-reject) -> LDS ring -> stage 2
//                     (colour gate, KL gate, parent rule; bit-exact float32 maths, gsr_math.h) -> LDS queue -> stage 3
//                     (likelihood, pair records).  VALU issue + its chain of dependent loads.
//   k_heavy_items + k_select<.., QUEUE = true>   the heavy parents (capacity > 16x the mean) cut into work items of <= 8192
//                     candidates, served from a queue by a second launch beside the light parents' (second stream):
//                     one wave per parent made the heaviest parents the kernel's critical path
//   k_compact_pairs   sparse segments (the parts of a split parent in order) -> parent-major CSR
//   k_bucket_hist / k_bucket_scatter / k_bucket_sum   per-child sums of wL: counting sort into buckets of 64 ... 8192
//                     children (4096 at 5 M), then LDS accumulation on a per-child fixed-point scale -- deterministic
//                     without a sort (GSR_HEM_SUMLW=sort: rocPRIM stable sort by child + k_sumlw)
//   k_mstep           one wavefront per four parents: responsibilities and moment sums with a lane per pair, SH rows with a
//                     lane group per child (3 float4 per lane); reductions by DPP rotations and v_permlane swaps
//   k_orphans*, k_valid, k_compact   orphans, validity erase, output in the reference's order
//                     (parents by ascending input index, then orphans by ascending input index)
//   k_rng_block_state / k_flags_glibc   the next level's parent flags from the libc rand() stream
//
// Any conservative neighbour search is legal: the reference's candidate set is exactly
// { i : |mu_i - mu_s|^2 < R_s^2 } (its 27-cell scan with cell >= R_s loses nothing), and that test
// is re-evaluated here with the same float expression.
#include "gsr_common.h"
#include "gsr_math.h"
#include "gsr_normals.h"
#include "gsr_test_hooks.h"
#include "hem_device.h"
#include "hem_select.h"

#include <float.h>
#include <math.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <vector>

#include <rocprim/rocprim.hpp>

namespace gsr {

std::string& last_error() {
    static thread_local std::string s;
    return s;
}
int32_t fail(int32_t code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    last_error() = buf;
    return code;
}

void GlibcRng::seed(uint32_t s) {
    if (s == 0) s = 1;
    int32_t word = (int32_t)s;
    st[0] = (uint32_t)word;
    for (int i = 1; i < 31; ++i) {
        int32_t hi = word / 127773, lo = word % 127773;
        word = 16807 * lo - 2836 * hi;
        if (word < 0) word += 2147483647;
        st[i] = (uint32_t)word;
    }
    f = 3; r = 0;
    for (int i = 0; i < 310; ++i) next();
}

// ------------------------------------------------------------------------------------------------
// k_prep: det, parent query radius, bounding box of the finite centres
//   radius = delta * sqrtf(lambda_max)                       (mixture.cpp:88)
// ------------------------------------------------------------------------------------------------
// Is this component "regular", i.e. may the stage-1 filter of k_select judge it?  The filter's exactness argument
// ("stage-1 bound") needs, of a CHILD: its covariance (the float32 entries read as real numbers) is
// symmetric positive definite, and the float32 determinant the reference computes for it (det6, vec.hpp:863-866) is
// within DET_TOL of the true one; and of both children and parents a determinant inside [1e-18, 1e18], so that the
// quotient det_c / det_p of the KL gate is a normal float32 number (an overflowing quotient makes the reference ACCEPT
// the pair through log(inf); nothing in front of the exact gates may then drop it).  All of it is VERIFIED here in
// float64 on the component's own numbers -- no bound on the condition number: a flat disc with axis ratio 300 passes,
// a needle whose float32 determinant is cancellation noise does not, and takes the exact gates only (pass B).
//   products of two float32 values are exact in float64; the 2x2 minor and the cofactor expansion then carry a few
//   float64 roundings of their largest term, so "minor > 1e-12 a00 a11" and "det > 1e-9 (sum of |terms|)" certify positive
//   definiteness (Sylvester) and a determinant good to 1e-6 relative.
#define ERASE_MAX 32          // erased rows a level compacts in place on the device (k_erase_save / k_erase_shift); more: the host path
__device__ __forceinline__ bool is_regular(const s6& c, float det, float x, float y, float z) {
    const float big = 1e12f;
    if (!(fabsf(x) < big && fabsf(y) < big && fabsf(z) < big)) return false;
    if (!(fabsf(c.e00) < big && fabsf(c.e01) < big && fabsf(c.e02) < big && fabsf(c.e11) < big && fabsf(c.e12) < big && fabsf(c.e22) < big)) return false;
    if (!(det >= 1e-18f && det <= 1e18f)) return false;
    double det64;
    if (!spd_det64((double)c.e00, (double)c.e01, (double)c.e02, (double)c.e11, (double)c.e12, (double)c.e22, det64)) return false;
    return fabs((double)det - det64) <= GSR_DET_TOL * det64;
}

__global__ __launch_bounds__(256) void k_prep(int64_t n, const float* __restrict__ xyz, const float* __restrict__ color,
                                              const float* __restrict__ cov6, const float* __restrict__ opacity,
                                              const float* __restrict__ weight, const uint8_t* __restrict__ is_parent,
                                              float4* __restrict__ rec /* [n][4], input order */, unsigned* __restrict__ bbox_part,
                                              const long long* __restrict__ n_dev /* NULL, or the level's size where the host does not know it yet */) {
    if (n_dev) n = *n_dev;
    // The four float4 of a component (the A/B/C/D layout of the working set) are packed here, in INPUT order and
    // with coalesced reads, so that the gather into cell order fetches one 64-byte record per component instead
    // of touching nine arrays at a random index (9 x 128-byte lines -> 0.9 ms at 5 M; one line -> 0.3 ms).
    // A lane builds the record of its component; the wave then writes its 64 records as 4 KiB of contiguous memory (through LDS,
    // four lanes per record): four float4 stores per lane straight to rec[4 i + u] touched 64 cache lines per instruction.
    __shared__ float4 s_t[4][64 * 5];
    float4* st = s_t[threadIdx.x >> 6];
    const int lane = threadIdx.x & 63;
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    unsigned n_par = 0u, n_irr = 0u;                // parents / irregular components seen by this thread (the level's P and the size of
                                                    // pass B's list: read back with the grid, one host round trip less per level)
    for (int64_t base = (int64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63); base < n; base += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = base + lane;
        if (i < n) {
            s6 c = {cov6[6 * i], cov6[6 * i + 1], cov6[6 * i + 2], cov6[6 * i + 3], cov6[6 * i + 4], cov6[6 * i + 5]};
            const float dt = det6(c);
            float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
            const unsigned fl = (is_parent[i] ? 1u : 0u) | (is_regular(c, dt, x, y, z) ? 2u : 0u);   // bit 0 parent, bit 1 regular
            n_par += fl & 1u;
            n_irr += (fl & 2u) ? 0u : 1u;
            st[lane * 5] = make_float4(x, y, z, __uint_as_float(fl));
            st[lane * 5 + 1] = make_float4(c.e00, c.e01, c.e02, c.e11);
            st[lane * 5 + 2] = make_float4(c.e12, c.e22, color[3 * i], color[3 * i + 1]);
            st[lane * 5 + 3] = make_float4(color[3 * i + 2], opacity[i], weight[i], dt);
            if (fabsf(x) <= FLT_MAX && fabsf(y) <= FLT_MAX && fabsf(z) <= FLT_MAX) {   // finite only
                mn[0] = fminf(mn[0], x); mx[0] = fmaxf(mx[0], x);
                mn[1] = fminf(mn[1], y); mx[1] = fmaxf(mx[1], y);
                mn[2] = fminf(mn[2], z); mx[2] = fmaxf(mx[2], z);
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = (lane >> 2) + 16 * u;
            if (base + r < n) rec[4 * (base + r) + (lane & 3)] = st[r * 5 + (lane & 3)];
        }
        __builtin_amdgcn_wave_barrier();
    }
    // per-block partial box, no atomics: 10^4 same-address atomics cost ~0.5 ms on this part (they serialise
    // across the XCDs); k_bbox_reduce folds the partials
    __shared__ float s_mn[4][3], s_mx[4][3];
    __shared__ unsigned s_cnt[4][2];
    for (int k = 0; k < 3; ++k) { mn[k] = wave_min(mn[k]); mx[k] = wave_max(mx[k]); }
    for (int o = 32; o > 0; o >>= 1) { n_par += (unsigned)__shfl_xor((int)n_par, o); n_irr += (unsigned)__shfl_xor((int)n_irr, o); }
    if ((threadIdx.x & 63) == 0) {
        for (int k = 0; k < 3; ++k) { s_mn[threadIdx.x >> 6][k] = mn[k]; s_mx[threadIdx.x >> 6][k] = mx[k]; }
        s_cnt[threadIdx.x >> 6][0] = n_par; s_cnt[threadIdx.x >> 6][1] = n_irr;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int k = threadIdx.x;
        float a = s_mn[0][k], b = s_mx[0][k];
        for (int w = 1; w < 4; ++w) { a = fminf(a, s_mn[w][k]); b = fmaxf(b, s_mx[w][k]); }
        bbox_part[8 * blockIdx.x + k] = enc_f(a);
        bbox_part[8 * blockIdx.x + 3 + k] = enc_f(b);
    } else if (threadIdx.x < 5) {
        const int k = threadIdx.x - 3;
        bbox_part[8 * blockIdx.x + 6 + k] = s_cnt[0][k] + s_cnt[1][k] + s_cnt[2][k] + s_cnt[3][k];
    }
}
// bbox[0..2] = min, bbox[3..5] = max over the per-block partial boxes (order-preserving uint encoding)
// (also clears the level's counters and the axis histograms: two memset launches less per level)
__global__ __launch_bounds__(256) void k_bbox_reduce(int nblocks, const unsigned* __restrict__ part, unsigned* __restrict__ bbox,
                                                     unsigned* __restrict__ zero_a, int na, unsigned* __restrict__ zero_b, int nzb,
                                                     unsigned* __restrict__ zero_c, int nzc) {
    __shared__ unsigned s_v[4][8];
    for (int i = threadIdx.x; i < na; i += blockDim.x) zero_a[i] = 0u;
    for (int i = threadIdx.x; i < nzb; i += blockDim.x) zero_b[i] = 0u;
    for (int i = threadIdx.x; i < nzc; i += blockDim.x) zero_c[i] = 0u;      // (the bucket cursors of the pair partition: two memset launches less per level)
    unsigned v[8];
    for (int k = 0; k < 3; ++k) { v[k] = 0xffffffffu; v[3 + k] = 0u; }
    v[6] = v[7] = 0u;                                // bbox[6] = parents of the level, bbox[7] = irregular components (k_prep's counts)
    for (int b = threadIdx.x; b < nblocks; b += blockDim.x) {
        for (int k = 0; k < 3; ++k) {
            const unsigned lo = part[8 * b + k], hi = part[8 * b + 3 + k];
            v[k] = lo < v[k] ? lo : v[k];
            v[3 + k] = hi > v[3 + k] ? hi : v[3 + k];
        }
        v[6] += part[8 * b + 6]; v[7] += part[8 * b + 7];
    }
    for (int k = 0; k < 3; ++k)
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned lo = (unsigned)__shfl_xor((int)v[k], o), hi = (unsigned)__shfl_xor((int)v[3 + k], o);
            v[k] = lo < v[k] ? lo : v[k];
            v[3 + k] = hi > v[3 + k] ? hi : v[3 + k];
        }
    for (int k = 6; k < 8; ++k)
        for (int o = 32; o > 0; o >>= 1) v[k] += (unsigned)__shfl_xor((int)v[k], o);
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 8; ++k) s_v[threadIdx.x >> 6][k] = v[k];
    __syncthreads();
    if (threadIdx.x < 8) {
        const int k = threadIdx.x;
        unsigned r = s_v[0][k];
        for (int w = 1; w < 4; ++w) r = k < 3 ? (s_v[w][k] < r ? s_v[w][k] : r) : (k < 6 ? (s_v[w][k] > r ? s_v[w][k] : r) : r + s_v[w][k]);
        bbox[k] = r;
    }
}

// Per-axis histograms (HIST_BINS bins over the bounding box of the finite centres): the grid is laid over
// the box that holds all but the outermost 0.1 % of the centres per side, so that a few far-away
// background splats cannot blow the cells up.  Centres outside that box are clamped into the boundary
// cells (cell_of is a monotone, non-expanding map, so the neighbour search stays conservative; the
// boundary rows are treated as half-infinite slabs when rows are culled).
#define HIST_BINS 1024
__global__ __launch_bounds__(256) void k_hist(int64_t n, const float* __restrict__ xyz, const unsigned* __restrict__ bbox,
                                              unsigned* __restrict__ hist /* [3][HIST_BINS] */, const long long* __restrict__ n_dev) {
    if (n_dev) n = *n_dev;
    __shared__ unsigned s_h[3 * HIST_BINS];
    for (int k = threadIdx.x; k < 3 * HIST_BINS; k += blockDim.x) s_h[k] = 0u;
    __syncthreads();
    float mn[3], sc[3];
    for (int k = 0; k < 3; ++k) {
        mn[k] = dec_f(bbox[k]);
        const float ext = dec_f(bbox[3 + k]) - mn[k];
        sc[k] = ext > 0.0f ? (float)HIST_BINS / ext : 0.0f;
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
        if (fabsf(x) <= FLT_MAX && fabsf(y) <= FLT_MAX && fabsf(z) <= FLT_MAX) {
            const float v[3] = {x, y, z};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                int b = (int)((v[k] - mn[k]) * sc[k]);
                b = b < 0 ? 0 : (b > HIST_BINS - 1 ? HIST_BINS - 1 : b);
                atomicAdd(&s_h[k * HIST_BINS + b], 1u);
            }
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 3 * HIST_BINS; k += blockDim.x)
        if (s_h[k]) atomicAdd(&hist[k], s_h[k]);
}

// Grid geometry: about `target` components per cell over the robust box, at most `max_cells` cells.
__global__ void k_grid_params(const unsigned* __restrict__ bbox, const unsigned* __restrict__ hist, int64_t n, float target,
                              int max_cells, GridParams* __restrict__ gp, const long long* __restrict__ n_dev) {
    if (n_dev) n = *n_dev;
    const int lane = threadIdx.x;                    // launched with ONE wavefront
    float mn[3], mx[3];
    for (int k = 0; k < 3; ++k) { mn[k] = dec_f(bbox[k]); mx[k] = dec_f(bbox[3 + k]); }
    GridParams g;
    if (!(mx[0] >= mn[0]) || !(mx[1] >= mn[1]) || !(mx[2] >= mn[2])) {   // no finite point at all
        g.ox = g.oy = g.oz = 0.0f; g.c = 1.0f; g.inv_c = 1.0f; g.slack = 0.0f; g.gx = g.gy = g.gz = 1; g.ncells = 1;
        if (lane == 0) *gp = g;
        return;
    }
    // robust box: drop the outermost 0.1 % per side (whole bins), then one bin of margin
    double inside = (double)n;
    for (int k = 0; k < 3; ++k) {
        const float ext = mx[k] - mn[k];
        if (!(ext > 0.0f)) continue;
        // one wavefront: lane l owns bins [16 l, 16 l + 16); the cumulative counts are monotone, so the number of
        // bins whose running total stays <= cut IS the index the sequential scan would stop at
        unsigned hv[HIST_BINS / 64];
        unsigned long long mine = 0;
        for (int b = 0; b < HIST_BINS / 64; ++b) { hv[b] = hist[k * HIST_BINS + lane * (HIST_BINS / 64) + b]; mine += hv[b]; }
        unsigned long long incl = mine;                                   // inclusive scan over the lanes
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned long long t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        const unsigned long long tot = __shfl(incl, 63);
        const unsigned long long cut = tot / 1000ull;
        int cl = 0, ch = 0;
        {
            unsigned long long run = incl - mine;                         // total of the lanes below
            for (int b = 0; b < HIST_BINS / 64; ++b) { run += hv[b]; cl += run <= cut ? 1 : 0; }
            run = tot - incl;                                             // total of the lanes above
            for (int b = HIST_BINS / 64 - 1; b >= 0; --b) { run += hv[b]; ch += run <= cut ? 1 : 0; }
        }
        for (int o = 32; o > 0; o >>= 1) { cl += __shfl_xor(cl, o); ch += __shfl_xor(ch, o); }
        int lo = cl < HIST_BINS - 1 ? cl : HIST_BINS - 1;
        int hi = HIST_BINS - 1 - ch;
        hi = hi > lo ? hi : lo;
        lo = lo > 0 ? lo - 1 : 0;
        hi = hi < HIST_BINS - 1 ? hi + 1 : HIST_BINS - 1;
        const float w = ext / (float)HIST_BINS;
        const float nmn = mn[k] + w * (float)lo, nmx = mn[k] + w * (float)(hi + 1);
        mn[k] = nmn; mx[k] = nmx < mx[k] ? nmx : mx[k];
        inside *= 0.998;
    }
    float ex = mx[0] - mn[0], ey = mx[1] - mn[1], ez = mx[2] - mn[2];
    float emax = fmaxf(ex, fmaxf(ey, ez));
    float eps = emax * 1e-6f + 1e-30f;
    double vol = (double)(ex + eps) * (double)(ey + eps) * (double)(ez + eps);
    double c = cbrt(vol * (double)target / (inside > 1.0 ? inside : 1.0));
    if (!(c > 0.0)) c = 1.0;
    int gx, gy, gz;
    for (;;) {
        double fx = floor(ex / c) + 1.0, fy = floor(ey / c) + 1.0, fz = floor(ez / c) + 1.0;
        if (fx * fy * fz <= (double)max_cells && fx < 2e6 && fy < 2e6 && fz < 2e6) { gx = (int)fx; gy = (int)fy; gz = (int)fz; break; }
        c *= 1.2599210498948732;
    }
    g.ox = mn[0]; g.oy = mn[1]; g.oz = mn[2];
    g.c = (float)c; g.inv_c = 1.0f / g.c;
    g.slack = 1e-5f * (emax + g.c);
    g.gx = gx; g.gy = gy; g.gz = gz; g.ncells = gx * gy * gz;
    if (lane == 0) *gp = g;
}

__global__ __launch_bounds__(256) void k_keys(int64_t n, const float* __restrict__ xyz,
                                              const GridParams* __restrict__ gpp,
                                              unsigned* __restrict__ keys, unsigned* __restrict__ idx) {
    const GridParams g = *gpp;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int cx = cell_of(xyz[3 * i], g.ox, g.inv_c, g.gx);
        int cy = cell_of(xyz[3 * i + 1], g.oy, g.inv_c, g.gy);
        int cz = cell_of(xyz[3 * i + 2], g.oz, g.inv_c, g.gz);
        keys[i] = (unsigned)((cz * g.gy + cy) * g.gx + cx);
        idx[i] = (unsigned)i;
    }
}

// start[k] = number of sorted keys < k, for k in [0, nkeys]; sorted ascending.  Each run head fills
// the (possibly empty) range of key values that precede it.
template <typename OffT>
__global__ __launch_bounds__(256) void k_run_starts(int64_t m, const unsigned* __restrict__ skeys, int64_t nkeys,
                                                    OffT* __restrict__ start) {
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += (int64_t)gridDim.x * blockDim.x) {
        unsigned k = skeys[j];
        int64_t prev = j == 0 ? -1 : (int64_t)skeys[j - 1];
        if ((int64_t)k != prev)
            for (int64_t c = prev + 1; c <= (int64_t)k; ++c) start[c] = (OffT)j;
        if (j == m - 1)
            for (int64_t c = (int64_t)k + 1; c <= nkeys; ++c) start[c] = (OffT)m;
    }
}
template <typename OffT>
__global__ void k_fill_const(int64_t n, OffT* p, OffT v) {
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) p[j] = v;
}

// Sorted working set.
//   radius = delta * sqrtf(lambda_max) for the parents                       (mixture.cpp:88)
// (ONE trip per wave, no grid-stride loop: with MachineLICM -- this translation unit is built with it since round 5 -- the compiler hoisted
// the float64 constants of eig_max6 out of the loop into 116 VGPRs and halved the occupancy of a kernel that is nothing but gather
// latency.  Launch with ceil(n / 256) workgroups.)
__global__ __launch_bounds__(256) void k_gather(int64_t n, const unsigned* __restrict__ order, const float4* __restrict__ rec, float delta,
                                                float4* __restrict__ A, float4* __restrict__ geo, float* __restrict__ Rs,
                                                int* __restrict__ pflag, int* __restrict__ iflag,
                                                const float* __restrict__ sh, int F, float* __restrict__ sh_tail /* [F rounded up to 4 + 4], or NULL */) {
    // The 64-byte records move with FOUR lanes per record: a load instruction touches 16 records' lines and a store
    // instruction writes 1 KiB of contiguous memory (a lane per record: 64 lines per instruction, both ways).  The lane that
    // owns sorted position j then reads its record back from LDS for the radius and the flags.
    __shared__ float4 s_t[4][64 * 5];
    float4* st = s_t[threadIdx.x >> 6];
    const int lane = threadIdx.x & 63;
    if (blockIdx.x == 0 && threadIdx.x == 0) iflag[n] = 0;       // the scan runs over n + 1 entries: irank[n] = total
    // the last row of the level's SH array, padded with zeros: the M-step reads THAT row here (load4_unaligned)
    if (sh_tail && blockIdx.x == 0)
        for (int t = threadIdx.x; t < ((F + 3) & ~3) + 4; t += blockDim.x) sh_tail[t] = t < F ? sh[(n - 1) * F + t] : 0.0f;
    const int64_t base = (int64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63);
    if (base < n) {
        const int64_t j = base + lane;
        const unsigned oi = j < n ? order[j] : 0u;
        // (the four rounds' loads together, pinned in front of the stores: inside `if (base + r < n)` every round was a round trip of its
        // own; the lanes behind the end read record order[0])
        float4 v[4];
        unsigned iu[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            iu[u] = (unsigned)__shfl((int)oi, (lane >> 2) + 16 * u, 64);
            v[u] = rec[4 * (int64_t)iu[u] + (lane & 3)];
        }
        asm volatile("" : "+v"(v[0].x), "+v"(v[0].y), "+v"(v[0].z), "+v"(v[0].w), "+v"(v[1].x), "+v"(v[1].y), "+v"(v[1].z), "+v"(v[1].w),
                          "+v"(v[2].x), "+v"(v[2].y), "+v"(v[2].z), "+v"(v[2].w), "+v"(v[3].x), "+v"(v[3].y), "+v"(v[3].z), "+v"(v[3].w));
        // the flags word {bit 0 parent, bit 1 regular} also carries the component's INPUT index (<< 2; n < 2^30): the M-step finds a
        // child's SH row where the level lies -- no cell-sorted copy of the SH block (k_gather_sh: 1.8 GB of traffic at 5 M)
        if ((lane & 3) == 0) {
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u].w = __uint_as_float((__float_as_uint(v[u].w) & 3u) | (iu[u] << 2));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = (lane >> 2) + 16 * u;
            if (base + r < n) {
                geo[4 * (base + r) + (lane & 3)] = v[u];
                st[r * 5 + (lane & 3)] = v[u];
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (j < n) {
            const float4 a = st[lane * 5], b = st[lane * 5 + 1], cc = st[lane * 5 + 2];
            const unsigned fl = __float_as_uint(a.w);
            A[j] = a;
            float R = 0.0f;
            if (fl & 1u) {
                const s6 cov = {b.x, b.y, b.z, b.w, cc.x, cc.y};
                R = delta * sqrtf(eig_max6(cov));
            }
            Rs[j] = R;
            pflag[j] = (int)(fl & 1u);
            iflag[j] = (fl & 2u) ? 0 : 1;
        }
        __builtin_amdgcn_wave_barrier();
    }
}
struct __attribute__((packed, aligned(4))) f4u { float x, y, z, w; };      // a float4 at a 4-byte aligned address (gfx950 loads it with one dwordx4)
// shs rows: the F SH-rest coefficients of the component, zero padded to RSH = whole float4.  One thread per float4.
// Whether this copy is MADE is decided here, on the device, from the level's pair count (the kernel runs behind the selection, on
// the third stream beside the pair partition and the per-child sums): a cell-sorted copy costs one pass over the SH block (1.8 GB of
// traffic at 5 M) and buys the M-step locality -- rows of neighbouring children next to each other, 16-byte aligned -- for every READ
// of a row.  An isotropic level reads a row 22 times (k_mstep 2.84 ms from the copy, 3.25 ms from the level's own array: the copy
// pays); a level of thin discs reads it twice (1.09 against 1.14 ms: the copy, 0.47 ms, does not).  policy: 0 always copy, 1 never,
// 2 copy iff pairs >= thr * n.  *mode_out = 1: no copy, the M-step and the orphans read the level's own array (k_mstep: sh_mode_p).
__global__ __launch_bounds__(256) void k_gather_sh(int64_t n, int F, int RSH, const unsigned* __restrict__ order,
                                                   const float* __restrict__ sh, float* __restrict__ shs,
                                                   const int64_t* __restrict__ poff, const unsigned* __restrict__ pcnt, int P, int policy, float thr,
                                                   int* __restrict__ mode_out) {
    {
        int direct = policy == 1 ? 1 : 0;
        if (policy == 2) {
            const double pairs = P > 0 ? (double)poff[P - 1] + (double)pcnt[P - 1] : 0.0;
            direct = pairs < (double)thr * (double)n ? 1 : 0;
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) *mode_out = direct;
        if (direct) return;
    }
    const int Q = RSH >> 2;
    const int64_t total = n * Q;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    auto row_piece = [&](int64_t t, int64_t& j, int& q) -> float4 {
        j = total < ((int64_t)1 << 31) ? (int64_t)((unsigned)t / (unsigned)Q) : t / Q;      // 32-bit division when it fits
        q = (int)(t - j * Q);
        const float* src = sh + (int64_t)order[j] * F + 4 * q;
        const int left = F - 4 * q;
        float4 v;
        if (left >= 4) {                 // rows of F floats are only 4-byte aligned: ONE unaligned 16-byte load instead of four dword loads
            const f4u u = *reinterpret_cast<const f4u*>(src);
            v = make_float4(u.x, u.y, u.z, u.w);
        } else {
            v.x = left > 0 ? src[0] : 0.0f; v.y = left > 1 ? src[1] : 0.0f; v.z = left > 2 ? src[2] : 0.0f; v.w = 0.0f;
        }
        return v;
    };
    // two pieces per trip: twice the bytes in flight behind the order[] look-up (a dependent pair of loads per piece).
    // (Measured in round 4: both look-ups first, then both pieces by branch-free unaligned loads pinned together -- the ISA then shows
    // LL W LL W S S instead of L W L.. L W L.. -- 0.50 against 0.48 ms at 5 M: the pass is bound by its 180-byte random reads, not by
    // the order of its loads.)
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; t + stride < total; t += 2 * stride) {
        int64_t j0, j1;
        int q0, q1;
        const float4 v0 = row_piece(t, j0, q0), v1 = row_piece(t + stride, j1, q1);
        reinterpret_cast<float4*>(shs + j0 * RSH)[q0] = v0;
        reinterpret_cast<float4*>(shs + j1 * RSH)[q1] = v1;
    }
    if (t < total) {
        int64_t j0;
        int q0;
        const float4 v0 = row_piece(t, j0, q0);
        reinterpret_cast<float4*>(shs + j0 * RSH)[q0] = v0;
    }
}
__global__ __launch_bounds__(256) void k_scatter_list(int64_t n, const int* __restrict__ flag, const int* __restrict__ pos,
                                                      unsigned* __restrict__ list) {
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x)
        if (flag[j]) list[pos[j]] = (unsigned)j;
}

// The candidate stream of pass A: the non-parent components in cell order, the sorted position packed beside the two flag bits
// (n < 2^30), and the grid's prefix table counted over them (cellStartC[c] = children before cell c = cellStart[c] - parents
// before it).  ppos = exclusive scan of pflag; P = its total.
__global__ __launch_bounds__(256) void k_child_stream(int64_t n, int64_t cells, int P, const float4* __restrict__ A, const int* __restrict__ pflag,
                                                      const int* __restrict__ ppos, const int* __restrict__ cellStart,
                                                      float4* __restrict__ Ac, int* __restrict__ cellStartC, int pad,
                                                      const int* __restrict__ irank /* NULL: the level has no irregular component */, int* __restrict__ cellStartI) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t t = t0; t < pad; t += stride) Ac[(n - P) + t] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);     // the pad k_select reads past the last row
    for (int64_t j = t0; j < n; j += stride) {
        if (pflag[j]) continue;
        const float4 a = A[j];
        Ac[j - ppos[j]] = make_float4(a.x, a.y, a.z, __uint_as_float(((unsigned)j << 2) | (__float_as_uint(a.w) & 3u)));
    }
    for (int64_t cidx = t0; cidx <= cells; cidx += stride) {
        const int sidx = cellStart[cidx];
        cellStartC[cidx] = sidx - (sidx < n ? ppos[sidx] : P);
        // the grid's prefix table counted over the IRREGULAR components (pass B's row spans: two look-ups instead of a dependent pair of pairs)
        if (irank) cellStartI[cidx] = irank[sidx];
    }
}

// Morton bits per axis of the ordering key: the curve runs over blocks of (grid / 2^bits)^3 cells, the parents of a block stay in
// cell order (stable sort).  6 bits: 20-bit keys = two 10-bit Onesweep passes where 10 bits per axis took four (-0.1 ms at 5 M;
// k_select and k_mstep, which run in this order, measure the same with 6, 7 and 10 bits).
// The ordering key: [work class : 2 bits][Morton code of the parent's block of cells : kb bits].  Large levels: 6 bits per axis (kb = 18: two
// Onesweep passes of 10 bits); levels of fewer than ORDER_FINE_FROM parents: 3 bits per axis less the finest x bit (kb = 8) -- the whole key
// fits ONE Onesweep pass, and on a small level the coarser blocks cost the selection nothing (measured, profiles/r06f_ab_order_key_bits.txt:
// 556 k level 1.185 -> 1.135 ms, 1.67 M level 2.98 -> 2.94, 5 M level 8.31 -> 8.34: the 5 M level keeps the fine key).
#ifndef ORDER_AXIS_BITS
#define ORDER_AXIS_BITS 6
#endif
#ifndef ORDER_FINE_FROM
#define ORDER_FINE_FROM 1000000
#endif
__host__ __device__ inline int order_axis_bits(int P) { return P >= ORDER_FINE_FROM ? ORDER_AXIS_BITS : 3; }
__host__ __device__ inline int order_drop_bits(int P) { return P >= ORDER_FINE_FROM ? 0 : 1; }
__host__ __device__ inline int order_key_bits(int P) { return 3 * order_axis_bits(P) - order_drop_bits(P); }
// The work items of the heavy parents (the first *nheavy slots of the processing order): parts of SEL_PART candidates.
// ONE workgroup: the heavy parents are a few thousand.  hq[0] = number of items, hq[1] = the queue cursor, which starts
// behind the items the waves of the serving workgroups take without asking (see k_select).
// It also COUNTS the heavy parents (the sorted ordering keys below class 3: k_count_heavy's binary search, a launch less) and leaves
// the count in *nheavy_p for the selection kernels.  hfirst[] is written for every heavy parent and read for no other.
__global__ __launch_bounds__(1024) void k_heavy_items(int P, const unsigned* __restrict__ sorted_keys, int* __restrict__ nheavy_p,
                                                      const unsigned* __restrict__ porder,
                                                      const unsigned* __restrict__ pcap, const int64_t* __restrict__ coff, unsigned* __restrict__ part_out,
                                                      int own_lo, int own_hi, int first_pull, int max_items,
                                                      uint2* __restrict__ hitem, int* __restrict__ hfirst, unsigned* __restrict__ pcnt,
                                                      int* __restrict__ hq, int* __restrict__ error_flag) {
    __shared__ int s_wsum[16];
    __shared__ int s_base, s_nheavy;
    // part size: ~4 items per wave slot of the chip, between 2048 and SEL_PART candidates (on a small level one item of 8192
    // candidates outlasts the whole light launch) -- from the level's candidate total, here on the device
    unsigned part = SEL_PART;
    {
        const unsigned long long cand = (unsigned long long)coff[P - 1] + pcap[P - 1];
        while (part > 2048u && (unsigned long long)part * 4ull * 7168ull > cand) part >>= 1;
    }
    if (threadIdx.x == 0) {
        *part_out = part;
        int lo = 0, hi = P;                       // first key of class 3
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (sorted_keys[mid] >= (3u << order_key_bits(P))) hi = mid; else lo = mid + 1; }
        s_nheavy = lo;
        *nheavy_p = lo;
        s_base = 0;
    }
    __syncthreads();
    const int nheavy = s_nheavy;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int h0 = 0; h0 < nheavy; h0 += 1024) {
        const int h = h0 + (int)threadIdx.x;
        int p = -1, np = 0;
        if (h < nheavy) {
            p = (int)porder[h];
            if (p >= own_lo && p < own_hi) { const unsigned c = pcap[p]; np = (int)(c / part + (c % part ? 1u : 0u)); np = np < 1 ? 1 : np; }
        }
        int incl = np;
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
        if (lane == 63) s_wsum[wv] = incl;
        __syncthreads();
        int off = s_base;
        for (int w = 0; w < wv; ++w) off += s_wsum[w];
        off += incl - np;
        if (p >= 0) {
            const bool fits = off + np <= max_items;                 // max_items bounds sum(ceil(cap / part)) -- unless a capacity saturated
            if (np > 0 && !fits) *error_flag = 1;                    // (the light launch skips this slot: its pairs would vanish; the host checks)
            hfirst[p] = np > 0 && fits ? off : -1;
            pcnt[p] = 0u;
            if (fits) for (int k = 0; k < np; ++k) hitem[off + k] = make_uint2((unsigned)h, (unsigned)k);
        }
        __syncthreads();
        if (threadIdx.x == 1023) s_base = off + np;
        __syncthreads();
    }
    if (threadIdx.x == 0) { hq[0] = s_base < max_items ? s_base : max_items; hq[1] = first_pull; }
}

// Longest-processing-time-first: work per parent is heavy-tailed (a few parents scan 10^5 candidates), so the
// heavy ones are launched first and the many light ones fill in behind them; otherwise a heavy parent that
// happens to sit late in the spatial order is the kernel's tail.  key = 0 heavy / 1 light, stable sort.
__device__ __forceinline__ unsigned spread10(unsigned v) {      // 10 bits -> every third bit
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}
// key = [work class : 2 bits][Morton code of the parent's block of cells : 3 ORDER_AXIS_BITS bits].  Within a class the parents are
// processed along a Z-order curve: the parents in flight at any time then cover a compact 3-D block, so
// the children / candidates they share stay in L2 (the x-fastest linear order of the arrays makes the
// in-flight set a full-width slab of the scene, which does not fit).
__global__ __launch_bounds__(256) void k_heavy_keys(int P, const unsigned* __restrict__ work, const int64_t* __restrict__ coff,
                                                    const unsigned* __restrict__ plist, const float4* __restrict__ A,
                                                    const GridParams* __restrict__ gpp, unsigned* __restrict__ keys, unsigned* __restrict__ idx) {
    const GridParams g = *gpp;
    // "heavy" = 16x the mean capacity (8x: +0.8 % at 5 M, 32x: +10 % at 200 k); the total from the capacities' scan, on the device
    const unsigned long long cand = (unsigned long long)coff[P - 1] + work[P - 1];
    const unsigned thr = (unsigned)(16.0 * (double)cand / (double)P) + 1u;
    int gm = g.gx > g.gy ? g.gx : g.gy;
    gm = gm > g.gz ? gm : g.gz;
    int sh = 0;
    const int ab = order_axis_bits(P), db = order_drop_bits(P), kb = order_key_bits(P);
    while ((gm >> sh) > (1 << ab)) ++sh;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
        const unsigned w = work[p];
        // classes: 0 = >= 64 thr, 1 = >= 8 thr, 2 = >= thr, 3 = light
        const unsigned cls = w >= 64u * thr ? 0u : (w >= 8u * thr ? 1u : (w >= thr ? 2u : 3u));
        const float4 a = A[plist[p]];
        const unsigned cx = (unsigned)cell_of(a.x, g.ox, g.inv_c, g.gx) >> sh;
        const unsigned cy = (unsigned)cell_of(a.y, g.oy, g.inv_c, g.gy) >> sh;
        const unsigned cz = (unsigned)cell_of(a.z, g.oz, g.inv_c, g.gz) >> sh;
        keys[p] = (cls << kb) | ((spread10(cx) | (spread10(cy) << 1) | (spread10(cz) << 2)) >> db);
        idx[p] = (unsigned)p;
    }
}

__global__ void k_count_heavy(int P, const unsigned* __restrict__ sorted_keys, int* __restrict__ out) {
    int lo = 0, hi = P;                       // first key of class 3
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (sorted_keys[mid] >= (3u << order_key_bits(P))) hi = mid; else lo = mid + 1; }
    *out = lo;
}

// pack the sparse per-parent segments [coff[p], coff[p]+pcnt[p]) into the compact parent-major CSR [poff[p], ...); the parts
// of a split parent (hfirst[p] >= 0: segments SEL_PART apart, part_cnt pairs each) are concatenated in part order
__global__ __launch_bounds__(256) void k_compact_pairs(int P, const int64_t* __restrict__ coff, const unsigned* __restrict__ pcnt,
                                                       const int64_t* __restrict__ poff, const int* __restrict__ hfirst,
                                                       const unsigned* __restrict__ part_cnt, const unsigned* __restrict__ pcap, const unsigned* __restrict__ part_p,
                                                       const unsigned* __restrict__ sc, const float* __restrict__ sw,
                                                       unsigned* __restrict__ dc, float* __restrict__ dw) {
    const unsigned part = part_p ? *part_p : SEL_PART;
    // 32 lanes per parent (a parent has ~65 pairs): two parents per wavefront (8 lanes: +9 % of the pass, 16: +3 %)
    const int sub = threadIdx.x & 31;
    const int p = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 5);
    if (p >= P) return;
    int64_t dof = poff[p];
    const int64_t so = coff[p];
    const int h = hfirst ? hfirst[p] : -1;
    if (h < 0) {
        const unsigned cnt = pcnt[p];
        for (unsigned i = sub; i < cnt; i += 32) { dc[dof + i] = sc[so + i]; dw[dof + i] = sw[so + i]; }
        return;
    }
    const int np = (int)(pcap[p] / part + (pcap[p] % part ? 1u : 0u));
    for (int k = 0; k < (np < 1 ? 1 : np); ++k) {
        const unsigned cnt = part_cnt[h + k];
        const int64_t sk = so + (int64_t)k * part;
        for (unsigned i = sub; i < cnt; i += 32) { dc[dof + i] = sc[sk + i]; dw[dof + i] = sw[sk + i]; }
        dof += cnt;
    }
}

// The parts of a split (heavy) parent lie `part` slots apart in its sparse segment, part_cnt pairs each: slide them together
// IN PLACE, in part order, so that every parent's pairs are ONE contiguous run [coff[p], coff[p] + pcnt[p]) -- what the
// partition pass and the M-step walk.  One workgroup per heavy parent; a chunk is read whole (registers), then written behind a
// barrier: the destination never runs ahead of the source (dst <= src), so no pair is overwritten before it has been read.
__global__ __launch_bounds__(256) void k_join_parts(const int* __restrict__ nheavy_p, const unsigned* __restrict__ porder, const int64_t* __restrict__ coff,
                                                    const int* __restrict__ hfirst, const unsigned* __restrict__ part_cnt,
                                                    const unsigned* __restrict__ pcap, const unsigned* __restrict__ part_p, unsigned* __restrict__ sc, float* __restrict__ sw) {
    const int nheavy = *nheavy_p;
    const unsigned part = *part_p;
    for (int h = blockIdx.x; h < nheavy; h += gridDim.x) {
        const int p = (int)porder[h];
        const int h0 = hfirst[p];
        if (h0 < 0) continue;
        const int np = (int)(pcap[p] / part + (pcap[p] % part ? 1u : 0u));
        int64_t dst = coff[p] + part_cnt[h0];
        for (int k = 1; k < np; ++k) {
            const unsigned cnt = part_cnt[h0 + k];
            const int64_t src = coff[p] + (int64_t)k * part;
            if (dst != src) {
                // (eight pairs of a thread in flight between the two barriers: one pair per trip made a part of 3 000 pairs twelve
                // dependent round trips, and the largest parent's chain is this kernel's duration -- 0.10 ms at the 5 M level)
                constexpr int JU = 8;
                for (unsigned i0 = 0; i0 < cnt; i0 += 256 * JU) {
                    unsigned vc[JU]; float vw[JU];
#pragma unroll
                    for (int u = 0; u < JU; ++u) {
                        const unsigned i = i0 + 256u * u + threadIdx.x;
                        vc[u] = 0u; vw[u] = 0.0f;
                        if (i < cnt) { vc[u] = sc[src + i]; vw[u] = sw[src + i]; }
                    }
                    __syncthreads();
#pragma unroll
                    for (int u = 0; u < JU; ++u) {
                        const unsigned i = i0 + 256u * u + threadIdx.x;
                        if (i < cnt) { sc[dst + i] = vc[u]; sw[dst + i] = vw[u]; }
                    }
                    __syncthreads();
                }
            }
            dst += cnt;
        }
    }
}

// The pairs of the per-parent segments [seg[p], seg[p] + pcnt[p]) -- the sparse capacity layout of the one-pass selection, or
// the compact layout of the two-pass fallback -- straight into per-bucket regions of FIXED capacity `cap` (bucket b owns
// [b cap, (b + 1) cap) of o_child / o_wl): no histogram pass, no scan and no compacted copy of the pair list in between.
// A workgroup takes PART_PPW consecutive parents, lays their pair counts out as a prefix table in LDS and strides over the
// FLAT space of their pairs (a lane finds its pair's parent by binary search in that table: every lane is busy whether the
// parents hold 7 pairs each or one of them 10^5), counts the pairs per bucket in LDS, reserves a run in every bucket it
// touches with ONE global atomic each, and places the pairs on a second walk (L2 resident).  A bucket that overflows raises
// *overflow; the caller then takes the exact path (histogram + scan of a compacted copy).
#define PART_PPW 128
// The bucketed copy of a pair: its child's SLOT inside the bucket (sorted position mod the bucket size, 16 bits: buckets hold at most
// 8 192 children) and wL -- what the sums need and nothing else (the bucket itself is where the pair lies): 6 bytes where rounds 2-4
// wrote 8 (the child's whole sorted position), read twice by k_bucket_sum.
typedef unsigned short slot_t;
#define PART_STAGE_T 1024
// STAGE > 0: the workgroup (1 024 threads, STAGE * 112 / 8192 parents of ~60 pairs each) reads every pair ONCE into registers,
// ranks it inside its bucket with an LDS atomic, lays the pairs out by bucket in LDS (8 bytes each) and writes every bucket's run
// with consecutive lanes on consecutive addresses (runs of ~70 pairs instead of the 3-pair fragments of a 256-lane walk; the
// second read of the pair list is gone: 0.99 -> 0.65 ms at the 5 M level).  Beside the stage the kernel keeps 8 bytes
// of LDS per bucket; the launcher picks the largest stage that still lets TWO workgroups share a CU: 8 192 pairs up to 1 536
// buckets (n <= 6.3 M), 6 144 up to 3 584, 4 096 up to 5 632 (a 40 M-splat level), the two-walk form beyond.  A chunk with more
// pairs than the stage holds (heavy parents) takes the two-walk form inside the same kernel.
// The order of the pairs inside a bucket is arbitrary in both forms: k_bucket_sum adds integers.
template <int STAGE>
__global__ __launch_bounds__(STAGE ? PART_STAGE_T : 256, STAGE ? 8 : 1) void k_partition(int P, const int64_t* __restrict__ seg, const unsigned* __restrict__ pcnt,
                                                   const unsigned* __restrict__ sc, const float* __restrict__ sw, int nb, int shift, unsigned cap,
                                                   unsigned* __restrict__ cursor, slot_t* __restrict__ o_child, float* __restrict__ o_wl,
                                                   int* __restrict__ overflow) {
    constexpr bool STAGED = STAGE > 0;
    const unsigned smask = (1u << shift) - 1u;
    constexpr int PPW = STAGED ? STAGE * 112 / 8192 : PART_PPW;
    constexpr int PART_STAGE = STAGED ? STAGE : 1;
    extern __shared__ unsigned s_h[];
    __shared__ unsigned long long s_off[PPW + 1];            // exclusive prefix of the parents' pair counts (a heavy level: > 2^32 in one chunk is impossible, 64 bits anyway)
    __shared__ long long s_seg[PPW];
    __shared__ unsigned long long s_w0;
    __shared__ unsigned s_ws[PART_STAGE_T / 64];
    for (int b = threadIdx.x; b < nb; b += blockDim.x) s_h[b] = 0u;
    const int p0 = (int)blockIdx.x * PPW;
    const int np = P - p0 < PPW ? P - p0 : PPW;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x < 128) {
        const int t = (int)threadIdx.x;
        unsigned long long cnt = t < np ? (unsigned long long)pcnt[p0 + t] : 0ull;
        if (t < np) s_seg[t] = seg[p0 + t];
        unsigned long long incl = cnt;
        for (int o = 1; o < 64; o <<= 1) { const unsigned long long v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
        if (wv == 0 && lane == 63) s_w0 = incl;
        if (t < PPW) s_off[t + 1] = incl;                     // wave 1's entries still lack wave 0's total
    }
    if (threadIdx.x == 0) s_off[0] = 0ull;
    __syncthreads();
    if (threadIdx.x >= 64 && threadIdx.x < PPW) s_off[threadIdx.x + 1] += s_w0;
    __syncthreads();
    const unsigned long long total = s_off[PPW];
    auto pair_at = [&](unsigned long long f) -> long long {
        int a = 0, b = PPW;                                    // last parent with s_off[parent] <= f
        while (b - a > 1) { const int m = (a + b) >> 1; if (s_off[m] <= f) a = m; else b = m; }
        return s_seg[a] + (long long)(f - s_off[a]);
    };
    if (STAGED && total <= (unsigned long long)PART_STAGE) {
        constexpr int IT = STAGED ? PART_STAGE / PART_STAGE_T : 1;
        unsigned* s_g = s_h + nb;                              // global slot of the bucket's run minus its LDS position
        unsigned* st_c = s_g + nb;
        float* st_w = reinterpret_cast<float*>(st_c + PART_STAGE);
        const unsigned tot = (unsigned)total;
        // the parent of every flat pair index as a byte table (in the not yet used wl stage): one LDS read per pair instead of a
        // seven-step binary search (16 dependent searches per thread were 3/4 of this kernel's time)
        unsigned char* s_par = reinterpret_cast<unsigned char*>(st_w);
        for (int pp = wv; pp < np; pp += PART_STAGE_T / 64) {
            const unsigned o = (unsigned)s_off[pp], cnt = (unsigned)s_off[pp + 1] - o;
            for (unsigned k = lane; k < cnt; k += 64) s_par[o + k] = (unsigned char)pp;
        }
        __syncthreads();
        if (threadIdx.x < PPW) s_seg[threadIdx.x] -= (long long)s_off[threadIdx.x];       // pair address = s_seg[parent] + flat index
        __syncthreads();
        unsigned ch[IT], slot[IT];
        float wl[IT];
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const unsigned f = threadIdx.x + (unsigned)i * PART_STAGE_T;
            ch[i] = 0u; wl[i] = 0.0f;
            if (f < tot) { const long long at = s_seg[s_par[f]] + (long long)f; ch[i] = sc[at]; wl[i] = sw[at]; }
        }
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const unsigned f = threadIdx.x + (unsigned)i * PART_STAGE_T;
            // consecutive pairs of a parent mostly share their bucket: one LDS atomic per RUN of equal buckets in the wave (64 lanes on
            // one LDS address would take 64 turns), the run's lanes take consecutive slots behind its head
            const unsigned bk = f < tot ? ch[i] >> shift : 0xffffffffu;
            const unsigned prev = __shfl_up(bk, 1, 64);
            const bool head = lane == 0 || bk != prev;
            const unsigned long long hm = __ballot(head);
            const unsigned long long upto = (2ull << lane) - 1ull;            // bits 0..lane (lane 63: all)
            const int hl = 63 - __clzll((long long)(hm & upto));
            const unsigned long long above = hm & ~upto;
            const int nxt = above ? __ffsll((long long)above) - 1 : 64;
            unsigned base = 0u;
            if (head && f < tot) base = atomicAdd(&s_h[bk], (unsigned)(nxt - lane));
            base = __shfl(base, hl, 64);
            slot[i] = base + (unsigned)(lane - hl);
        }
        __syncthreads();
        // exclusive prefix of the bucket counts (thread t owns K consecutive buckets), one global atomic per touched bucket
        const int K = (nb + PART_STAGE_T - 1) / PART_STAGE_T;
        const int b0 = (int)threadIdx.x * K;
        unsigned mine = 0u;
        for (int k = 0; k < K; ++k) if (b0 + k < nb) mine += s_h[b0 + k];
        unsigned incl = mine;
        for (int o = 1; o < 64; o <<= 1) { const unsigned v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
        if (lane == 63) s_ws[wv] = incl;
        __syncthreads();
        unsigned before = incl - mine;
        for (int w = 0; w < wv; ++w) before += s_ws[w];
        for (int k = 0; k < K; ++k) {
            const int b = b0 + k;
            if (b >= nb) break;
            const unsigned cnt = s_h[b];
            s_h[b] = before;
            if (cnt) {
                const unsigned base = atomicAdd(&cursor[b], cnt);
                if (base + cnt > cap || base + cnt < base) *overflow = 1;
                s_g[b] = base - before;
            }
            before += cnt;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const unsigned f = threadIdx.x + (unsigned)i * PART_STAGE_T;
            if (f < tot) { const unsigned pos = s_h[ch[i] >> shift] + slot[i]; st_c[pos] = ch[i]; st_w[pos] = wl[i]; }
        }
        __syncthreads();
        for (unsigned pos = threadIdx.x; pos < tot; pos += PART_STAGE_T) {
            const unsigned c = st_c[pos];
            const unsigned bk = c >> shift;
            const unsigned sl = s_g[bk] + pos;
            if (sl < cap) { const size_t gp = (size_t)bk * cap + sl; o_child[gp] = (slot_t)(c & smask); o_wl[gp] = st_w[pos]; }
        }
        return;
    }
    for (int pass = 0; pass < 2; ++pass) {
        for (unsigned long long f = threadIdx.x; f < total; f += blockDim.x) {
            const long long at = pair_at(f);
            const unsigned ch = sc[at];
            const int bk = (int)(ch >> shift);
            const unsigned slot = atomicAdd(&s_h[bk], 1u);
            if (pass == 1 && slot < cap) { const size_t pos = (size_t)bk * cap + slot; o_child[pos] = (slot_t)(ch & smask); o_wl[pos] = sw[at]; }
        }
        __syncthreads();
        if (pass == 0) {
            for (int b = threadIdx.x; b < nb; b += blockDim.x) {
                const unsigned cnt = s_h[b];
                if (cnt) {
                    const unsigned base = atomicAdd(&cursor[b], cnt);
                    if (base + cnt > cap || base + cnt < base) *overflow = 1;
                    s_h[b] = base;
                }
            }
            __syncthreads();
        }
    }
}
// one launch of the pair partition: the largest stage whose LDS fits twice on a CU
template <int STAGE>
static inline void launch_partition_staged(hipStream_t st, int P, const int64_t* seg, const unsigned* pcnt, const unsigned* sc, const float* sw, int nb, int shift,
                                           unsigned cap, unsigned* cursor, slot_t* o_child, float* o_wl, int* overflow) {
    hipLaunchKernelGGL(k_partition<STAGE>, dim3(ceil_div(P, STAGE * 112 / 8192)), dim3(PART_STAGE_T), (size_t)nb * 8 + (size_t)STAGE * 8, st, P, seg, pcnt, sc, sw,
                       nb, shift, cap, cursor, o_child, o_wl, overflow);
}
static inline void launch_partition(hipStream_t st, int staged_ok /* 0 walk, 1 auto, else the stage size to force */, int P, const int64_t* seg, const unsigned* pcnt, const unsigned* sc, const float* sw, int nb,
                                    int shift, unsigned cap, unsigned* cursor, slot_t* o_child, float* o_wl, int* overflow) {
    if ((staged_ok == 1 || staged_ok == 8192) && nb <= 1536) launch_partition_staged<8192>(st, P, seg, pcnt, sc, sw, nb, shift, cap, cursor, o_child, o_wl, overflow);
    else if ((staged_ok == 1 || staged_ok == 6144) && nb <= 3584) launch_partition_staged<6144>(st, P, seg, pcnt, sc, sw, nb, shift, cap, cursor, o_child, o_wl, overflow);
    else if ((staged_ok == 1 || staged_ok == 4096) && nb <= 5632) launch_partition_staged<4096>(st, P, seg, pcnt, sc, sw, nb, shift, cap, cursor, o_child, o_wl, overflow);
    else
        hipLaunchKernelGGL(k_partition<0>, dim3(ceil_div(P, PART_PPW)), dim3(256), (size_t)nb * 4, st, P, seg, pcnt, sc, sw, nb, shift, cap, cursor, o_child,
                           o_wl, overflow);
}

// per-child sum of wL_si, sequential in the (stable) sorted pair order (mixture.cpp:162)
__global__ __launch_bounds__(256) void k_sumlw(int64_t n, const int64_t* __restrict__ cstart,
                                               const float* __restrict__ wl_sorted, float* __restrict__ sumLw,
                                               int* __restrict__ orphan_flag, float* __restrict__ geo_sl) {
    // 8 lanes per child (a child has ~22 pairs, contiguous after the sort): lane s adds elements s, s+8, ... in order and
    // the eight partial sums are folded in a fixed tree -- deterministic, and the wave reads contiguous memory
    // (thread-per-child read 64 scattered segments per load: 0.59 -> 0.2 ms at 5 M).
    const int sub = threadIdx.x & 7;
    const int64_t stride = ((int64_t)gridDim.x * blockDim.x) >> 3;
    for (int64_t j = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3; j < n; j += stride) {
        const int64_t k0 = cstart[j], k1 = cstart[j + 1];
        float s = 0.0f;
        for (int64_t k = k0 + sub; k < k1; k += 8) s += wl_sorted[k];
        s += __shfl_xor(s, 4);
        s += __shfl_xor(s, 2);
        s += __shfl_xor(s, 1);
        if (sub == 0) {
            sumLw[j] = s;
            geo_sl[16 * j] = s;
            orphan_flag[j] = s == 0.0f ? 1 : 0;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Per-child sums of wL without sorting the pairs (the 3-pass radix sort of 10^8 pairs cost 2.4 of a level's 17 ms).
// Children are grouped in buckets of SUM_BUCKET consecutive sorted positions.  k_bucket_hist / k_bucket_scatter
// partition the pairs by bucket (a counting sort; the order INSIDE a bucket is whatever the atomics make it), then one
// workgroup per bucket accumulates its pairs in LDS.  The sums are nevertheless deterministic: every child's terms
// are added as 64-bit integers on a per-child fixed-point scale (2^38 steps of the child's largest term, found by a
// first pass of LDS max operations), and integer addition does not care about order.  Terms below 2^-38 of the
// largest are lost (the float32 sequential sum of the reference loses them below 2^-24); the total is rounded to
// float32 once.  Non-finite terms: the sum is NaN if any term is NaN or both infinities occur, else that infinity.
// ------------------------------------------------------------------------------------------------
#define SUM_BUCKET_SHIFT 12
#define SUM_BUCKET (1 << SUM_BUCKET_SHIFT)        // children per bucket: 4096 x (4 + 8) bytes = 48 KiB of LDS, two workgroups per CU
                                                  // (phase 3 at 5 M: 8192 children 1.53 ms, 4096 1.31, 2048 1.45)
#define SUM_TILE 16384                            // pairs per workgroup of the partition kernels
#define SUM_MAX_BUCKETS 32768                     // 4 bytes of LDS per bucket in the partition kernels: 128 KiB (n <= 268 M components)

__global__ __launch_bounds__(256) void k_bucket_hist(int64_t M, int tile, const unsigned* __restrict__ child, int nb, int shift, unsigned* __restrict__ hist) {
    extern __shared__ unsigned s_h[];
    for (int b = threadIdx.x; b < nb; b += blockDim.x) s_h[b] = 0u;
    __syncthreads();
    const int64_t lo = (int64_t)blockIdx.x * tile, hi = lo + tile < M ? lo + tile : M;
    for (int64_t k = lo + threadIdx.x; k < hi; k += blockDim.x) atomicAdd(&s_h[child[k] >> shift], 1u);
    __syncthreads();
    for (int b = threadIdx.x; b < nb; b += blockDim.x)
        if (s_h[b]) atomicAdd(&hist[b], s_h[b]);
}
// bucket offsets are 64-bit (10^9 pairs at 40 M splats); cursor[b] starts at the bucket's first slot
__global__ __launch_bounds__(256) void k_bucket_scatter(int64_t M, int tile, const unsigned* __restrict__ child, const float* __restrict__ wl, int nb, int shift,
                                                        const unsigned long long* __restrict__ bstart, unsigned long long* __restrict__ cursor,
                                                        slot_t* __restrict__ o_child, float* __restrict__ o_wl) {
    // ONE 32-bit word of LDS per bucket (4 883 buckets at 40 M components: 20 KB; with a count and a 64-bit base per bucket
    // it was 59 KB = two workgroups per CU): first the tile's count, then the next free slot of the tile's run in the bucket,
    // relative to the bucket's first slot bstart[b]
    extern __shared__ unsigned s_h[];
    for (int b = threadIdx.x; b < nb; b += blockDim.x) s_h[b] = 0u;
    __syncthreads();
    const int64_t lo = (int64_t)blockIdx.x * tile, hi = lo + tile < M ? lo + tile : M;
    for (int64_t k = lo + threadIdx.x; k < hi; k += blockDim.x) atomicAdd(&s_h[child[k] >> shift], 1u);
    __syncthreads();
    for (int b = threadIdx.x; b < nb; b += blockDim.x) {
        const unsigned cnt = s_h[b];
        if (cnt) s_h[b] = (unsigned)(atomicAdd(&cursor[b], (unsigned long long)cnt) - bstart[b]);      // this workgroup's run in bucket b
    }
    __syncthreads();
    for (int64_t k = lo + threadIdx.x; k < hi; k += blockDim.x) {
        const unsigned ch = child[k];
        const int b = (int)(ch >> shift);
        const unsigned long long pos = bstart[b] + atomicAdd(&s_h[b], 1u);
        o_child[pos] = (slot_t)(ch & ((1u << shift) - 1u));
        o_wl[pos] = wl[k];
    }
}
// bstart != NULL: bucket b holds the pairs [bstart[b], bstart[b + 1]) (exact partition); else its region [b cap, b cap + cursor[b])
__global__ __launch_bounds__(1024) void k_bucket_sum(int64_t n, int shift, const unsigned long long* __restrict__ bstart, unsigned cap,
                                                     const unsigned* __restrict__ cursor, const slot_t* __restrict__ child,
                                                     const float* __restrict__ wl, float* __restrict__ sumLw, int* __restrict__ orphan_flag,
                                                     float* __restrict__ geo_sl) {
    extern __shared__ unsigned long long s_acc[];  // [bucket] int64 accumulators, then [bucket] max bit patterns
    const int bucket = 1 << shift;
    unsigned* s_max = (unsigned*)(s_acc + bucket);
    const int b = blockIdx.x;
    const int64_t c0 = (int64_t)b << shift;
    const int nc = (int)(n - c0 < bucket ? n - c0 : bucket);
    for (int i = threadIdx.x; i < bucket; i += blockDim.x) { s_acc[i] = 0ull; s_max[i] = 0u; }
    __syncthreads();
    unsigned long long k0, k1;
    if (bstart) { k0 = bstart[b]; k1 = bstart[b + 1]; }
    else { const unsigned cnt = cursor[b]; k0 = (unsigned long long)b * cap; k1 = k0 + (cnt < cap ? cnt : cap); }
    // four loads of a thread in flight (one load per trip kept 16 KB per CU in flight: the pass ran at 1.8 TB/s)
    {
        unsigned long long k = k0 + threadIdx.x;
        const unsigned long long bd = blockDim.x;
        for (; k + 3 * bd < k1; k += 4 * bd) {
            const unsigned c_0 = child[k], c_1 = child[k + bd], c_2 = child[k + 2 * bd], c_3 = child[k + 3 * bd];
            const float w_0 = wl[k], w_1 = wl[k + bd], w_2 = wl[k + 2 * bd], w_3 = wl[k + 3 * bd];
            atomicMax(&s_max[c_0], __float_as_uint(w_0) & 0x7fffffffu);           // |wL| as an ordered integer
            atomicMax(&s_max[c_1], __float_as_uint(w_1) & 0x7fffffffu);
            atomicMax(&s_max[c_2], __float_as_uint(w_2) & 0x7fffffffu);
            atomicMax(&s_max[c_3], __float_as_uint(w_3) & 0x7fffffffu);
        }
        for (; k < k1; k += bd) atomicMax(&s_max[child[k]], __float_as_uint(wl[k]) & 0x7fffffffu);
    }
    __syncthreads();
    const auto add_term = [&](unsigned i, float w) {          // i = the child's slot in this bucket
        const unsigned mb = s_max[i];
        if (mb >= 0x7f800000u) {                   // a non-finite term somewhere: collect flags (1 +inf, 2 -inf, 4 NaN)
            const unsigned wb = __float_as_uint(w);
            const unsigned long long f = (wb & 0x7fffffffu) > 0x7f800000u ? 4ull : (wb == 0x7f800000u ? 1ull : (wb == 0xff800000u ? 2ull : 0ull));
            if (f) atomicOr(&s_acc[i], f);
            return;
        }
        int e = (int)(mb >> 23);
        e = e < 1 ? 1 : e;                         // subnormal maximum: the scale of the smallest normal exponent
        const int k2 = 38 - (e - 127);             // q = w * 2^k2,  |q| <= 2^39
        const double scale = __longlong_as_double((long long)(1023 + k2) << 52);
        const long long q = __double2ll_rn((double)w * scale);
        atomicAdd(&s_acc[i], (unsigned long long)q);
    };
    // (the second walk like the first: four loads of a thread in flight -- an unrolled loop with the LDS work between its loads waited
    // for every pair of loads: one round trip per 1 024 pairs, 88 of them for a bucket of a 5 M level)
    {
        unsigned long long k = k0 + threadIdx.x;
        const unsigned long long bd = blockDim.x;
        for (; k + 3 * bd < k1; k += 4 * bd) {
            unsigned c_0 = child[k], c_1 = child[k + bd], c_2 = child[k + 2 * bd], c_3 = child[k + 3 * bd];
            float w_0 = wl[k], w_1 = wl[k + bd], w_2 = wl[k + 2 * bd], w_3 = wl[k + 3 * bd];
            asm volatile("" : "+v"(c_0), "+v"(c_1), "+v"(c_2), "+v"(c_3), "+v"(w_0), "+v"(w_1), "+v"(w_2), "+v"(w_3));
            add_term(c_0, w_0); add_term(c_1, w_1); add_term(c_2, w_2); add_term(c_3, w_3);
        }
        for (; k < k1; k += bd) add_term(child[k], wl[k]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nc; i += blockDim.x) {
        const unsigned mb = s_max[i];
        float sres;
        if (mb >= 0x7f800000u) {
            const unsigned long long f = s_acc[i];
            sres = (f & 4ull) || ((f & 1ull) && (f & 2ull)) ? __builtin_nanf("") : ((f & 1ull) ? __builtin_inff() : -__builtin_inff());
        } else {
            int e = (int)(mb >> 23);
            e = e < 1 ? 1 : e;
            const int k2 = 38 - (e - 127);
            const double inv = __longlong_as_double((long long)(1023 - k2) << 52);
            sres = (float)((double)(long long)s_acc[i] * inv);
        }
        sumLw[c0 + i] = sres;
        geo_sl[16 * (c0 + i)] = sres;                  // also into the 64-byte geometry record (in place of det, which only the selection used)
        orphan_flag[c0 + i] = sres == 0.0f ? 1 : 0;
    }
}

__global__ __launch_bounds__(256) void k_orphan_flags(int64_t n, const float* __restrict__ sumLw, int* __restrict__ orphan_flag, float* __restrict__ geo_sl) {
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) {
        orphan_flag[j] = sumLw[j] == 0.0f ? 1 : 0;
        geo_sl[16 * j] = sumLw[j];
    }
}

// flags in INPUT order, where the output ranks are defined (mixture.cpp:169,250-253).  The parent flags were drawn in input order
// (Level::is_parent; ghosts of a partitioned level are nobody's output here); only the orphans -- few -- travel back from their
// sorted positions (scattering both flag arrays of all n components cost 0.17 ms and 0.33 GB of partial-line writes at 5 M).
__global__ __launch_bounds__(256) void k_flags_in(int64_t n, int64_t n_own, const uint8_t* __restrict__ is_parent, int* __restrict__ pflag_in,
                                                  int* __restrict__ oflag_in) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        pflag_in[i] = i < n_own && is_parent[i] ? 1 : 0;
        oflag_in[i] = 0;
    }
}
__global__ __launch_bounds__(256) void k_orphans_to_input_order(int64_t n, const unsigned* __restrict__ order, const int* __restrict__ oflag_sorted,
                                                                int* __restrict__ oflag_in) {
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x)
        if (oflag_sorted[j]) oflag_in[order[j]] = 1;
}

// ------------------------------------------------------------------------------------------------
// k_mstep: responsibilities and moment-matching update (mixture.cpp:167-247).
//
// One wavefront handles MSTEP_K consecutive parents of the processing order.  Their headers (pair offset and count,
// output row, parent mean) were laid out in that order by k_mstep_headers, so a wave fetches them with scalar loads.
// Per parent:
//   part 1   lane <-> pair: w = (wL_si / sumLw_i) * weight_i (mixture.cpp:196-197), the 14 moment sums in the
//            reference's expressions; child index and w go to LDS.  The gathers of a pair (the 64-byte geometry record,
//            which also carries sumLw) are issued together and unconditionally.  The sums are folded over the lanes and
//            parked in LDS straight away, so that their registers are free for
//   part 2   lane group <-> child: G lanes x 3 float4 cover one SH row, so one round of three load instructions
//            fetches the rows of 64/G children (16 at SH degree 3); MSTEP_U rounds are in flight together (with a single
//            round the loop was a chain of dependent L2 round trips: five per 66-child parent, measured 1.3 ms of 3.4).
//   sums     across lanes by DPP row rotations and v_permlane16/32_swap (class_sum): no LDS-crossbar traffic.
//   output   the SH row leaves through LDS as one coalesced store.
// (History: one parent per wave with a dependent round trip per phase, SH rows four children per load and butterfly sums by
// ds_bpermute: 4.0 ms at 5 M.)
// ------------------------------------------------------------------------------------------------
struct MstepHeader {           // 32 bytes, one per processing slot
    long long off;             // first pair of the parent in the CSR
    unsigned cnt;              // its pairs
    int oslot;                 // output row (-1: parent of another rank, nothing to do)
    float px, py, pz;          // parent mean
    int js;                    // sorted position of the parent
};
struct MstepArgs {
    const float4* geo;
    const float* shs;          // the children's SH rows: sh_direct = 0 the cell-sorted, padded copy (row j at shs + j RSH); 1 = the level's own
                               // array (row of input index i at shs + i F, unpadded: see load4_unaligned).  Which of the two a level uses is
                               // decided on the device (k_gather_sh); the kernels set these two fields from *sh_mode_p when they start
    int sh_direct;
    const int* sh_mode_p;      // device: 1 = read the level's own array (sh_own), 0 = the cell-sorted copy (sh_sorted)
    const float* sh_own;
    const float* sh_sorted;
    unsigned sh_last;          // sh_direct: the input index of the array's LAST row, which is read from sh_tail instead (0xffffffff: none)
    const float* sh_tail;      // a copy of that row padded to RSH floats (k_gather wrote it): see load4_unaligned
    int RSH;                   // F rounded up to whole float4 (the row stride of the padded copy)
    const MstepHeader* hdr;    // [P], processing order
    int xcd;
    const int* nheavy;
    const unsigned* pair_child;
    const float* pair_wl;
    int P, F;
    int small;                 // serve parents of <= MSTEP_SMALL pairs four at a time (GSR_HEM_MSTEP_SMALL=0: the general path for all)
    int split;                 // k_mstep<.., HEAVY>: 1 = one wave per segment of a heavy parent, 0 = one wave per heavy parent
    const unsigned* hcount;    // device: [0] heavy parents, [1] their segments (work items)
    const uint4* hlist;        // heavy parent -> {slot, first item, segments}
    const uint2* hitems;       // item -> {slot, segment}
    float* hscratch;           // item -> 16 + RSH floats: the segment's 14 moment sums, its SH sums
    float *o_xyz, *o_color, *o_cov6, *o_opacity, *o_weight, *o_sh;
};
__global__ __launch_bounds__(256) void k_mstep_headers(int P, const unsigned* __restrict__ porder, const unsigned* __restrict__ plist,
                                                       const int64_t* __restrict__ poff, const unsigned* __restrict__ pcnt,
                                                       const unsigned* __restrict__ order, const int* __restrict__ prank_in,
                                                       const float4* __restrict__ A, int own_lo, int own_hi, MstepHeader* __restrict__ hdr,
                                                       unsigned* __restrict__ max_cnt, unsigned seg, unsigned* __restrict__ hcount,
                                                       uint4* __restrict__ hlist, uint2* __restrict__ hitems) {
    unsigned mx = 0u;
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < P; s += gridDim.x * blockDim.x) {
        const int p = porder ? (int)porder[s] : s;
        const int js = (int)plist[p];
        const float4 a = A[js];
        MstepHeader h;
        h.off = poff[p]; h.cnt = pcnt[p]; h.js = js;          // poff = first pair of the parent's run (sparse segment or compact CSR)
        h.oslot = (p >= own_lo && p < own_hi) ? prank_in[order[js]] : -1;
        h.px = a.x; h.py = a.y; h.pz = a.z;
        hdr[s] = h;
        if (h.oslot >= 0) mx = h.cnt > mx ? h.cnt : mx;
        // a HEAVY parent (more than one segment of `seg` pairs): its segments become work items of k_mstep<.., HEAVY> (hlist / hitems
        // cannot overflow: a heavy parent holds more than seg of the level's pairs and ceil(cnt / seg) <= 2 cnt / seg)
        if (hcount && h.oslot >= 0 && h.cnt > seg) {
            const unsigned nseg = (h.cnt + seg - 1u) / seg;
            const unsigned hi = atomicAdd(&hcount[0], 1u);
            const unsigned base = atomicAdd(&hcount[1], nseg);
            hlist[hi] = make_uint4((unsigned)s, base, nseg, 0u);
            for (unsigned k = 0; k < nseg; ++k) hitems[base + k] = make_uint2((unsigned)s, k);
        }
    }
    // the largest pair count of a parent of this level (a statistic: gsr_hem_get_stats_ex [5]).  One atomic per WORKGROUP, and only
    // where it would raise the word as the workgroup sees it: an atomicMax per wave was 8 192 same-address atomics at the 5 M level,
    // ~12 ns each one after the other -- 100 of this kernel's 128 us (35 of 38 us at 556 k).
    __shared__ unsigned s_mx[4];
    for (int o = 32; o > 0; o >>= 1) { const unsigned t = (unsigned)__shfl_xor((int)mx, o); mx = t > mx ? t : mx; }
    if ((threadIdx.x & 63) == 0) s_mx[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned bm = s_mx[0];
        for (int w = 1; w < 4; ++w) bm = s_mx[w] > bm ? s_mx[w] : bm;
        if (bm > __hip_atomic_load(max_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(max_cnt, bm);
    }
}

#ifndef MSTEP_CHUNK
#define MSTEP_CHUNK 128       // pairs per chunk: what the wave's LDS holds of a parent at a time (child indices, weights)
#endif
#define MSTEP_SMALL 16        // parents with at most this many pairs are served four at a time, one per DPP row
#define MSTEP_SEG 2048        // pairs per SEGMENT of a parent's sums (a multiple of MSTEP_CHUNK); see mstep_segment
#define MSTEP_NV 3            // float4 per lane and SH row
#define MSTEP_K 4             // parents per wavefront
#ifndef MSTEP_U
#define MSTEP_U 2             // rounds of SH row loads in flight (each: MSTEP_NV float4 per lane, 64/G children): what 96 VGPRs hold (MSTEP_WAVES)
#endif
#ifndef MSTEP_WAVES
#define MSTEP_WAVES 5         // waves per SIMD the kernel is built for: 96 VGPRs and 7 424 bytes of LDS (six granules of 1 280) per one-wave workgroup.
                              // The kernel's time follows 1.66 ms + 17.6 ms / (waves per CU) at the 5 M level (measured at 8 / 12 / 16 waves by padding
                              // the LDS, profiles/r05y_mstep_occupancy.txt): a gather floor and a latency term the waves share out.  16 -> 20 waves with
                              // two rounds in flight instead of four: 2.71 -> 2.61 ms (surfels -6 %, clusters -5 %), profiles/r05z_ab_mstep_waves.txt
#endif

// One SEGMENT of a parent's sums: the pairs [off, off + cnt) (cnt <= MSTEP_SEG), from zero.  Leaves the 14 moment sums in s_mom and
// the SH sums folded over the lane classes: G <= 16 (mstep_packed) in shp -- shp[v] of a lane of row q and class gl = lane mod G is
// the total of float 4 (gl + G v) + packed_slot(q) of the row, the four components of a float4 spread over the wave's four rows
// (two sums per swap, see pack32_sum) --, larger rows in acc (every lane of a class holds the class's total of its float4).
// How a parent's sums are DEFINED (all paths agree bit for bit, test_mstep_heavy_parent_split_changes_nothing): its pairs are cut
// into segments of MSTEP_SEG; inside a segment the sums run in chunks of MSTEP_CHUNK pairs, lane l of a chunk
// taking pairs l, l + 64, ..., the chunk's lane sums folded over the lanes (moment_sums: halves, row pairs, then inside the row) and
// added to the segment's running sums in chunk order, the SH products accumulated per lane group along the CHUNK by fused
// multiply-adds, folded behind it (the same order of levels) and added to the running totals in chunk order -- and the segments'
// totals are added in segment order.  A parent of at most MSTEP_SEG pairs (all but the giants) is one segment: nothing
// changed for it.  The giants' segments are independent, which is the point: a parent with 5 * 10^4 pairs (found on the surfel
// cloud) kept ONE wave busy for 2.5 ms, the whole M-step of that level.
// The 14 moment sums of the parent in flight live in the PADDING of the record stage (s_rec rows are 80 bytes apart for the sake of the
// LDS banks; the fifth float4 of rows 60 .. 63 is nobody's); LDS is allocated in granules of 1 280 bytes: the wave's 7 424 bytes (128 child
// indices, 128 weights, the record stage, 1 280 bytes for the row on its way out) are six of them, 20 waves per CU.  (Until late in round 5:
// chunks of 256 pairs and 3 KB for the per-lane SH products of a parent with more than one chunk, carried from chunk to chunk -- 10 240 bytes,
// 16 waves.  Now a chunk's products are folded behind the chunk and the running totals wait in the row's staging area.)
// One global_load_dwordx4 from an address that is only 4-byte aligned (a row of the level's own SH array: F = 45 floats).  The hardware
// takes it (HSA runs the memory pipeline in unaligned-access mode); the compiler, told the truth about the alignment, splits the load into
// two or three pieces (dwordx2 + dwordx2, dwordx3 + dword), so it is not told.
// A lane's float4 slot q covers floats [4 q, 4 q + 4) of the row; in an unpadded row the last slot runs up to three floats into the NEXT
// row -- sums that are never stored (the output stops at F) -- and behind the array's end for the one row that is the array's last: that
// row is read from a padded copy the library keeps (MstepArgs::sh_tail, written by k_gather), so no load leaves the caller's array.
__device__ __forceinline__ float4 load4_unaligned(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ const float* sh_row_of(const float* rows, const float* tail, unsigned idx, unsigned last, int stride) {
    const float* r = rows + (int64_t)idx * stride;
    return idx == last ? tail : r;
}
__host__ __device__ constexpr bool mstep_packed(int G) { return G >= 1 && G <= 16; }
struct MomRef {
    float4* rec;
    __device__ __forceinline__ float& operator[](int i) const { return reinterpret_cast<float*>(rec + (60 + (i >> 2)) * 5 + 4)[i & 3]; }
};
template <int G>
__device__ __forceinline__ void mstep_segment(const MstepArgs& a, int lane, int gl, int grp, const int (&qi)[MSTEP_NV], float* s_w, unsigned* s_j,
                                              MomRef s_mom, float4* s_rec, float* s_park, long long off, unsigned cnt, const f3 pm,
                                              float4 (&acc)[MSTEP_NV], float (&shp)[MSTEP_NV]) {
    constexpr int GG = G > 0 ? G : 1;
    constexpr int CPR = 64 / GG;                                // children per round
    const int sh_stride = a.sh_direct ? a.F : a.RSH;            // floats between two SH rows (qi: this lane's float offsets inside a row)
    if (lane < 16) s_mom[lane] = 0.0f;
    // SH sums: per chunk of pairs, from zero (the per-lane products occupy no registers during part 1), folded over the lanes behind the
    // chunk and added to the segment's running totals in chunk order (three registers in the packed form)
#pragma unroll
    for (int v = 0; v < MSTEP_NV; ++v) { acc[v] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); shp[v] = 0.0f; }
    float4 tot4[MSTEP_NV];                                      // (rows of more than 16 lanes: the unpacked totals)
#pragma unroll
    for (int v = 0; v < MSTEP_NV; ++v) tot4[v] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    for (unsigned c0 = 0; c0 < cnt; c0 += MSTEP_CHUNK) {
        const unsigned cn = (cnt - c0) < MSTEP_CHUNK ? (cnt - c0) : MSTEP_CHUNK;
        // part 1
        {
            float w_s = 0, smx = 0, smy = 0, smz = 0, scx = 0, scy = 0, scz = 0;
            float v00 = 0, v01 = 0, v02 = 0, v11 = 0, v12 = 0, v22 = 0, so = 0;
            // (the next batch's pairs are requested before this batch's records: one dependent round trip less per further batch)
            unsigned jn = a.pair_child[off + c0 + ((unsigned)lane < cn ? (unsigned)lane : cn - 1)];
            float wn = a.pair_wl[off + c0 + ((unsigned)lane < cn ? (unsigned)lane : cn - 1)];
#pragma nounroll
            for (unsigned k0 = 0; k0 < cn; k0 += 64) {
                const unsigned k = k0 + lane;
                const bool live = k < cn;
                const unsigned j = jn;
                const float wl = wn;
                if (k0 + 64 < cn) {
                    const unsigned kn = k + 64 < cn ? k + 64 : cn - 1;
                    jn = a.pair_child[off + c0 + kn];
                    wn = a.pair_wl[off + c0 + kn];
                }
                // the 64-byte records of the batch's 64 children, fetched by FOUR lanes per record (a wave instruction then
                // touches 16 cache lines instead of 64: one piece of 64 different records per instruction kept the CU's
                // address pipeline busy four times as long) and handed to the pair's lane through LDS
                s_j[k0 + lane] = j;
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int r = (lane >> 2) + 16 * u;
                    const unsigned jr = s_j[k0 + r];
                    s_rec[r * 5 + (lane & 3)] = a.geo[4 * (int64_t)jr + (lane & 3)];
                }
                __builtin_amdgcn_wave_barrier();
                const float4 ca = s_rec[lane * 5], cb = s_rec[lane * 5 + 1], cc = s_rec[lane * 5 + 2], cd = s_rec[lane * 5 + 3];
                __builtin_amdgcn_wave_barrier();
                if (a.sh_direct) s_j[k0 + lane] = __float_as_uint(ca.w) >> 2;      // part 2 wants the child's row in the level's own SH array: its input index
                const float sl = cd.w;                     // sumLw_i: k_bucket_sum stored it in the record (in place of det)
                float w = 0.0f;
                if (live && sl != 0.0f) {                  // sumLw == 0: skipped (mixture.cpp:190)
                    const float r_is = wl / sl;            // mixture.cpp:196
                    w = r_is * cd.z;                       // * child.weight (:197)
                    const f3 cm = {ca.x, ca.y, ca.z};
                    const f3 d = sub3(cm, pm);
                    w_s += w;
                    smx += cm.x * w; smy += cm.y * w; smz += cm.z * w;
                    scx += cc.z * w; scy += cc.w * w; scz += cd.x * w;
                    v00 += (cb.x + d.x * d.x) * w; v01 += (cb.y + d.x * d.y) * w; v02 += (cb.z + d.x * d.z) * w;
                    v11 += (cb.w + d.y * d.y) * w; v12 += (cc.x + d.y * d.z) * w; v22 += (cc.y + d.z * d.z) * w;
                    so += w * cd.y;
                }
                if (live) s_w[k] = w;
            }
            // the chunk's 14 sums over the lanes (moment_sums: two values per swap, four registers left for the row level), then out of
            // the registers (part 2 needs them for its row loads): the first lane of each row adds the row's four values
            const float mom[14] = {w_s, smx, smy, smz, scx, scy, scz, v00, v01, v02, v11, v12, v22, so};
            float mr[4];
            moment_sums(mom, mr);
            if ((lane & 15) == 0) {
                const int q = lane >> 4;
#pragma unroll
                for (int jj = 0; jj < 3; ++jj) s_mom[moment_of(jj, q)] += mr[jj];
                if ((q & 1) == 0) s_mom[moment_of(3, q)] += mr[3];
            }
        }
        __builtin_amdgcn_wave_barrier();
        // part 2: children in pair order, MSTEP_U rounds of row loads in flight; a skipped child has w = 0 (its row is
        // loaded all the same: no branch per load)
        if (G > 0) {
#pragma unroll
            for (int v = 0; v < MSTEP_NV; ++v) acc[v] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            for (unsigned k0 = 0; k0 < cn; k0 += CPR * MSTEP_U) {
                float4 rowv[MSTEP_U][MSTEP_NV];
                float wv_[MSTEP_U];
#pragma unroll
                for (int u = 0; u < MSTEP_U; ++u) {
                    const unsigned k = k0 + CPR * u + grp;
                    const unsigned kc = k < cn ? k : cn - 1;    // unconditional LDS reads and loads
                    const unsigned j = s_j[kc];
                    wv_[u] = k < cn ? s_w[kc] : 0.0f;
                    const float* row = sh_row_of(a.shs, a.sh_tail, j, a.sh_last, sh_stride);
#pragma unroll
                    for (int v = 0; v < MSTEP_NV; ++v) rowv[u][v] = load4_unaligned(row + 4 * qi[v]);
                }
#pragma unroll
                for (int u = 0; u < MSTEP_U; ++u) {
#pragma unroll
                    for (int v = 0; v < MSTEP_NV; ++v) {
                        acc[v].x = __builtin_fmaf(rowv[u][v].x, wv_[u], acc[v].x); acc[v].y = __builtin_fmaf(rowv[u][v].y, wv_[u], acc[v].y);
                        acc[v].z = __builtin_fmaf(rowv[u][v].z, wv_[u], acc[v].z); acc[v].w = __builtin_fmaf(rowv[u][v].w, wv_[u], acc[v].w);
                    }
                }
            }
            // the chunk's products over the lanes; the first chunk's totals ARE the running totals (0 + x would turn a -0 into +0)
#pragma unroll
            for (int v = 0; v < MSTEP_NV; ++v) {
                if constexpr (mstep_packed(G)) {
                    // x, y, z, w of the float4 end in rows 0, 2, 1, 3 (packed_slot), each row holding its component's class totals
                    const float t = row_class_sum<GG>(pack16_sum(pack32_sum(acc[v].x, acc[v].y), pack32_sum(acc[v].z, acc[v].w)));
                    // (between the chunks the running totals wait in LDS -- the row's staging area, idle until the parent's output -- and
                    // not in three registers that would be alive through part 1)
                    const float r = c0 == 0 ? t : s_park[v * 64 + lane] + t;
                    if (c0 + MSTEP_CHUNK < cnt) s_park[v * 64 + lane] = r; else shp[v] = r;
                } else {
                    const float tx = class_sum<GG>(acc[v].x), ty = class_sum<GG>(acc[v].y), tz = class_sum<GG>(acc[v].z), tw = class_sum<GG>(acc[v].w);
                    tot4[v].x = c0 == 0 ? tx : tot4[v].x + tx; tot4[v].y = c0 == 0 ? ty : tot4[v].y + ty;
                    tot4[v].z = c0 == 0 ? tz : tot4[v].z + tz; tot4[v].w = c0 == 0 ? tw : tot4[v].w + tw;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if constexpr (G > 0 && !mstep_packed(G)) {
#pragma unroll
        for (int v = 0; v < MSTEP_NV; ++v) acc[v] = tot4[v];
    }
}

// G = lanes per SH row (power of two, G * MSTEP_NV float4 >= RSH / 4); G == 0: no SH at all.  HEAVY: the work items of the heavy
// parents (k_mstep_headers) instead of the parents themselves -- one wave per segment, the segment's sums out to hscratch;
// k_mstep_heavy_finish adds a parent's segments in order and writes its row
#if MSTEP_WAVES > 0
#define MSTEP_ATTR __attribute__((amdgpu_waves_per_eu(MSTEP_WAVES, MSTEP_WAVES)))
#else
#define MSTEP_ATTR
#endif
template <int G, int WPB, bool HEAVY = false>
__global__ __launch_bounds__(64 * WPB) MSTEP_ATTR void k_mstep(MstepArgs a) {
    __shared__ float s_w_[WPB][MSTEP_CHUNK];
    __shared__ unsigned s_j_[WPB][MSTEP_CHUNK];
#ifndef MSTEP_PAD4
#define MSTEP_PAD4 0            // (occupancy experiments: extra float4 of LDS per wave)
#endif
    // a parent's SH row on its way out (F floats; the small-parent path: four rows 64 floats apart and the four parents' moment sums
    // behind them, 320 floats)
    __shared__ float4 s_acc_[WPB][(G > 16 ? MSTEP_NV * 32 : 80) + MSTEP_PAD4];
    __shared__ float4 s_rec_[WPB][64 * 5];                      // part 1: the batch's geometry records on their way to the pairs' lanes (80-byte stride)
    constexpr int GG = G > 0 ? G : 1;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    a.sh_direct = *a.sh_mode_p;                                 // (uniform: a scalar load)
    a.shs = a.sh_direct ? a.sh_own : a.sh_sorted;
    if (!a.sh_direct) a.sh_last = 0xffffffffu;
    float* s_w = s_w_[wv];
    unsigned* s_j = s_j_[wv];
    float4* s_acc = s_acc_[wv];
    float4* s_rec = s_rec_[wv];
    const MomRef s_mom = {s_rec};                               // the 14 moment sums of the parent (see MomRef)
    float* s_out = reinterpret_cast<float*>(s_acc);             // at the end of a parent: its SH row on the way out
    const int gl = lane & (GG - 1), grp = lane / GG;
    const int nq = a.RSH >> 2;                                  // float4 per SH row
    int qi[MSTEP_NV];                                           // float4 slots of this lane (slots beyond the row: re-read slot 0, discarded)
#pragma unroll
    for (int v = 0; v < MSTEP_NV; ++v) qi[v] = gl + GG * v < nq ? gl + GG * v : 0;
    if constexpr (HEAVY) {
        // a.split: one wave per work item (segment); GSR_HEM_MSTEP_SPLIT=0: one wave per heavy PARENT, its segments one after the
        // other (the schedule of rounds 1-3) -- the same segment records either way, added in order by k_mstep_heavy_finish
        const unsigned nitems = a.hcount[1], nheavy = a.hcount[0];
        const int RS = 16 + a.RSH;
        const unsigned nwork = a.split ? nitems : nheavy;
        for (unsigned wk = blockIdx.x * WPB + wv; wk < nwork; wk += gridDim.x * WPB) {
            const uint4 hp = a.split ? make_uint4(0u, wk, 1u, 0u) : a.hlist[wk];        // {-, first item, items}
            for (unsigned item = hp.y; item < hp.y + hp.z; ++item) {
                const uint2 it = a.hitems[item];
                const MstepHeader h = a.hdr[it.x];
                const f3 pm = {h.px, h.py, h.pz};
                const unsigned first = it.y * MSTEP_SEG;
                const unsigned cn = h.cnt - first < MSTEP_SEG ? h.cnt - first : MSTEP_SEG;
                float4 acc[MSTEP_NV];
                float shp[MSTEP_NV];
                mstep_segment<G>(a, lane, gl, grp, qi, s_w, s_j, s_mom, s_rec, s_out, h.off + first, cn, pm, acc, shp);
                float* rec = a.hscratch + (int64_t)item * RS;
                if (lane < 14) rec[lane] = s_mom[lane];
                if constexpr (mstep_packed(G)) {
                    if ((lane & 15) < GG) {
#pragma unroll
                        for (int v = 0; v < MSTEP_NV; ++v)
                            if (gl + GG * v < nq) rec[16 + 4 * (gl + GG * v) + packed_slot(lane >> 4)] = shp[v];
                    }
                } else if (G > 0 && grp == 0) {
#pragma unroll
                    for (int v = 0; v < MSTEP_NV; ++v)
                        if (gl + GG * v < nq) reinterpret_cast<float4*>(rec + 16)[gl + GG * v] = acc[v];
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        return;
    }
    constexpr int KB = MSTEP_K * WPB;                           // parents per workgroup: consecutive slots, neighbours in space
    const int nblk = (a.P + KB - 1) / KB;
    const int hbk = a.nheavy ? (((*a.nheavy + KB - 1) / KB + 7) & ~7) : 0;
    const int bid = block_slot((int)blockIdx.x, nblk, hbk < nblk ? hbk : nblk, a.xcd);
    if (bid < 0) return;
    const int s0 = bid * KB + wv * MSTEP_K;
    if (s0 >= a.P) return;
    const int ns = a.P - s0 < MSTEP_K ? a.P - s0 : MSTEP_K;     // parents of this wave

    // ---- small parents (at most MSTEP_SMALL = 16 pairs: most parents of a surfel-shaped cloud, 6.5 pairs on average) ----------------
    // The wave's small parents are served TOGETHER, parent q in DPP row q (lanes 16 q .. 16 q + 15), instead of one after the other
    // with 6 of 64 lanes alive: one pair-list load, one record gather, one set of row loads and one output sequence for up to four
    // parents.  Bit for bit the sums of the general path below: there lane l < 16 holds pair l, the lanes above hold +0.0, and
    // moment_sums is two additions of +0.0 (the other half, the other row of the pair) followed by row_sum16; the SH sum of a parent
    // with <= 16 pairs is one product per lane group (16 / G groups per row), and the general path folds its four rows first over the
    // halves, (S0 + S2) and (S1 + S3), then over the row pairs, then inside the row (class G) -- here the four rows of the general path
    // are four ROUNDS of this parent's own row, combined in the same order.  (G <= 4: a round then holds 16 / G >= 4 children per row and
    // <= 16 children never chain two products through one fmaf; larger rows -- SH degree 4 and up -- take the general path.)
    // The wave's (up to four) headers with ONE load: lane l takes dword l of the 128 contiguous bytes, a field is a v_readlane away.
    // (`a.hdr[s0 + it]` per parent came out as vector loads of one header at a time, each waited for: eight dependent round trips at
    // the head of every wave where one does.)
    static_assert(sizeof(MstepHeader) == 32 && MSTEP_K * 8 <= 64, "a wave's headers are 8 dwords each, one per lane");
    unsigned hv = 0u;
    if (lane < ns * 8) hv = reinterpret_cast<const unsigned*>(a.hdr + s0)[lane];
    const auto hfield = [hv](int it, int f) { return (unsigned)__builtin_amdgcn_readlane((int)hv, it * 8 + f); };
    const auto header = [&hfield](int it) {
        MstepHeader h;
        h.off = (long long)(((unsigned long long)hfield(it, 1) << 32) | hfield(it, 0));
        h.cnt = hfield(it, 2); h.oslot = (int)hfield(it, 3);
        h.px = __uint_as_float(hfield(it, 4)); h.py = __uint_as_float(hfield(it, 5)); h.pz = __uint_as_float(hfield(it, 6));
        h.js = (int)hfield(it, 7);
        return h;
    };
    unsigned small_mask = 0u;
    if constexpr (G <= 4) {
#pragma unroll
        for (int it = 0; it < MSTEP_K; ++it)
            if (it < ns) {
                if ((int)hfield(it, 3) >= 0 && hfield(it, 2) <= MSTEP_SMALL) small_mask |= 1u << it;
            }
    }
    if (G <= 4 && a.small && small_mask != 0u) {
        const int q = lane >> 4, i = lane & 15;
        // this lane's parent (row q): the four headers are uniform (scalar loads), the per-lane copy is a chain of selects
        long long off_q = 0; unsigned cnt_q = 0u; int slot_q = -1; float ppx = 0.0f, ppy = 0.0f, ppz = 0.0f;
        unsigned maxcnt = 0u;
#pragma unroll
        for (int it = 0; it < MSTEP_K; ++it)
            if ((small_mask >> it) & 1u) {
                const MstepHeader h = header(it);
                maxcnt = h.cnt > maxcnt ? h.cnt : maxcnt;
                if (q == it) { off_q = h.off; cnt_q = h.cnt; slot_q = h.oslot; ppx = h.px; ppy = h.py; ppz = h.pz; }
            }
        const bool live = (unsigned)i < cnt_q;
        unsigned j = 0xffffffffu;
        float wl = 0.0f;
        if (live) { j = a.pair_child[off_q + i]; wl = a.pair_wl[off_q + i]; }
        s_j[lane] = j;
        __builtin_amdgcn_wave_barrier();
        // records: gather round u serves row u (its 16 slots, four lanes per 64-byte record); rows without a small parent and
        // slots beyond a parent's pairs load nothing
        // (all four rounds' loads first, then the four LDS stores: a load under `if (jr valid)` followed by its store made every round
        // a round trip of its own; a slot without a pair reads record 0 and stores nothing)
        {
            float4 rv[4];
            unsigned jr[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                jr[u] = s_j[(lane >> 2) + 16 * u];
                rv[u] = a.geo[4 * (int64_t)(jr[u] != 0xffffffffu ? jr[u] : 0u) + (lane & 3)];
            }
            // (pinned here: a load whose only use sits behind a branch is sunk into it by the compiler, and waited for there)
            asm volatile("" : "+v"(rv[0].x), "+v"(rv[0].y), "+v"(rv[0].z), "+v"(rv[0].w), "+v"(rv[1].x), "+v"(rv[1].y), "+v"(rv[1].z), "+v"(rv[1].w),
                              "+v"(rv[2].x), "+v"(rv[2].y), "+v"(rv[2].z), "+v"(rv[2].w), "+v"(rv[3].x), "+v"(rv[3].y), "+v"(rv[3].z), "+v"(rv[3].w));
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (jr[u] != 0xffffffffu) s_rec[((lane >> 2) + 16 * u) * 5 + (lane & 3)] = rv[u];
        }
        __builtin_amdgcn_wave_barrier();
        float w = 0.0f;
        float w_s = 0, smx = 0, smy = 0, smz = 0, scx = 0, scy = 0, scz = 0;
        float v00 = 0, v01 = 0, v02 = 0, v11 = 0, v12 = 0, v22 = 0, so = 0;
        if (live) {
            const float4 ca = s_rec[lane * 5], cb = s_rec[lane * 5 + 1], cc = s_rec[lane * 5 + 2], cd = s_rec[lane * 5 + 3];
            if (a.sh_direct) s_j[lane] = __float_as_uint(ca.w) >> 2;          // the child's input index: where its SH row lies
            const float sl = cd.w;
            if (sl != 0.0f) {
                const float r_is = wl / sl;            // mixture.cpp:196
                w = r_is * cd.z;                       // * child.weight (:197)
                const f3 cm = {ca.x, ca.y, ca.z};
                const f3 pmq = {ppx, ppy, ppz};
                const f3 d = sub3(cm, pmq);
                w_s += w;
                smx += cm.x * w; smy += cm.y * w; smz += cm.z * w;
                scx += cc.z * w; scy += cc.w * w; scz += cd.x * w;
                v00 += (cb.x + d.x * d.x) * w; v01 += (cb.y + d.x * d.y) * w; v02 += (cb.z + d.x * d.z) * w;
                v11 += (cb.w + d.y * d.y) * w; v12 += (cc.x + d.y * d.z) * w; v22 += (cc.y + d.z * d.z) * w;
                so += w * cd.y;
            }
        }
        s_w[lane] = w;
        w_s = row_sum16(w_s);
        smx = row_sum16(smx); smy = row_sum16(smy); smz = row_sum16(smz);
        scx = row_sum16(scx); scy = row_sum16(scy); scz = row_sum16(scz);
        v00 = row_sum16(v00); v01 = row_sum16(v01); v02 = row_sum16(v02);
        v11 = row_sum16(v11); v12 = row_sum16(v12); v22 = row_sum16(v22);
        so = row_sum16(so);
        float* s_mom4 = reinterpret_cast<float*>(s_acc) + 256;          // 4 x 16 floats behind the four SH rows on their way out
        if (i == 0) {
            float* m = s_mom4 + 16 * q;
            m[0] = w_s; m[1] = smx; m[2] = smy; m[3] = smz; m[4] = scx; m[5] = scy; m[6] = scz;
            m[7] = v00; m[8] = v01; m[9] = v02; m[10] = v11; m[11] = v12; m[12] = v22; m[13] = so;
        }
        __builtin_amdgcn_wave_barrier();
        const float* mq = s_mom4 + 16 * q;
        const float w_sq = mq[0];
        const float inv_w = 1.0f / w_sq;                       // mixture.cpp:209
        if (slot_q >= 0) {
            const float mx = mq[1] * inv_w, my = mq[2] * inv_w, mz = mq[3] * inv_w;
            const float dx = mx - ppx, dy = my - ppy, dz = mz - ppz;
            const int64_t slot = slot_q;
            float val = w_sq;                                                        // lane 13 of the row: weight
            float* dst = a.o_weight + slot;
            if (i < 3) { val = i == 0 ? mx : (i == 1 ? my : mz); dst = a.o_xyz + 3 * slot + i; }
            else if (i < 6) { val = mq[4 + (i - 3)] * inv_w; dst = a.o_color + 3 * slot + (i - 3); }
            else if (i < 12) {
                const int t = i - 6;                                                 // xx xy xz yy yz zz
                const float da = t < 3 ? dx : (t < 5 ? dy : dz);
                const float db = t == 0 ? dx : (t == 1 || t == 3 ? dy : dz);
                val = mq[7 + t] * inv_w - da * db;                                   // mixture.cpp:211-212,236-238
                dst = a.o_cov6 + 6 * slot + t;
            } else if (i == 12) { val = inv_w * mq[13]; dst = a.o_opacity + slot; }
            if (i < 14) *dst = val;
        }
        if constexpr (G > 0) {
            constexpr int CR = 16 / GG;                        // children per round and row
            const int gi = i / GG;                             // this lane's group inside the row
            float4 rowv[GG][MSTEP_NV];
            float wr[GG];
#pragma unroll
            for (int r = 0; r < GG; ++r) {
                const unsigned k = (unsigned)(CR * r + gi);
                const bool lv = k < cnt_q && (unsigned)(CR * r) < maxcnt;
                wr[r] = 0.0f;
#pragma unroll
                for (int v = 0; v < MSTEP_NV; ++v) rowv[r][v] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (lv) {
                    const unsigned jc = s_j[16 * q + k];
                    wr[r] = s_w[16 * q + k];
                    const float* row = sh_row_of(a.shs, a.sh_tail, jc, a.sh_last, a.sh_direct ? a.F : a.RSH);
#pragma unroll
                    for (int v = 0; v < MSTEP_NV; ++v) rowv[r][v] = load4_unaligned(row + 4 * qi[v]);
                }
            }
            float4 tot[MSTEP_NV];
#pragma unroll
            for (int v = 0; v < MSTEP_NV; ++v) {
                float4 sr[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float4 pr = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    if (r < GG) {       // the general path's first accumulation into a zero: fmaf(row, w, +0.0); an absent child: +0.0
                        pr.x = __builtin_fmaf(rowv[r][v].x, wr[r], 0.0f); pr.y = __builtin_fmaf(rowv[r][v].y, wr[r], 0.0f);
                        pr.z = __builtin_fmaf(rowv[r][v].z, wr[r], 0.0f); pr.w = __builtin_fmaf(rowv[r][v].w, wr[r], 0.0f);
                    }
                    sr[r] = pr;
                }
                // the general path's levels in its order: the halves (rows 0 + 2, 1 + 3), the row pairs, then inside the row
                tot[v].x = row_class_sum<GG>((sr[0].x + sr[2].x) + (sr[1].x + sr[3].x)); tot[v].y = row_class_sum<GG>((sr[0].y + sr[2].y) + (sr[1].y + sr[3].y));
                tot[v].z = row_class_sum<GG>((sr[0].z + sr[2].z) + (sr[1].z + sr[3].z)); tot[v].w = row_class_sum<GG>((sr[0].w + sr[2].w) + (sr[1].w + sr[3].w));
            }
            __builtin_amdgcn_wave_barrier();
            if (gi == 0) {
#pragma unroll
                for (int v = 0; v < MSTEP_NV; ++v) {
                    const int f0 = 64 * q + 4 * (gl + GG * v);
                    s_out[f0] = tot[v].x * inv_w; s_out[f0 + 1] = tot[v].y * inv_w;
                    s_out[f0 + 2] = tot[v].z * inv_w; s_out[f0 + 3] = tot[v].w * inv_w;
                }
            }
            __builtin_amdgcn_wave_barrier();
            // four rows of F floats leave with coalesced stores: lane l of row q takes floats l, l + 16, ... of parent q's row
            if (slot_q >= 0)
                for (int f = i; f < a.F; f += 16) a.o_sh[(int64_t)slot_q * a.F + f] = s_out[64 * q + f];
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (!(G <= 4 && a.small)) small_mask = 0u;

    for (int it = 0; it < ns; ++it) {
        if ((small_mask >> it) & 1u) continue;                  // served above
        const MstepHeader h = header(it);
        if (h.oslot < 0) continue;
        const f3 pm = {h.px, h.py, h.pz};
        const unsigned cnt = h.cnt;
        if (cnt > MSTEP_SEG) continue;                          // a heavy parent: its segments belong to k_mstep<.., HEAVY> + k_mstep_heavy_finish
        float4 acc[MSTEP_NV];
        float shp[MSTEP_NV];
        mstep_segment<G>(a, lane, gl, grp, qi, s_w, s_j, s_mom, s_rec, s_out, h.off, cnt, pm, acc, shp);
        const float w_s = s_mom[0];
        const float inv_w = 1.0f / w_s;                        // mixture.cpp:209
        const int64_t slot = h.oslot;
        {
            // the 14 geometry outputs leave with ONE store instruction: lane t < 14 computes output t and stores it to its own
            // array (a vector-memory instruction costs the CU ~30 cycles whatever its lane count; 14 one-lane stores per
            // parent were a third of this kernel's memory instructions)
            const float mx = s_mom[1] * inv_w, my = s_mom[2] * inv_w, mz = s_mom[3] * inv_w;
            const float dx = mx - pm.x, dy = my - pm.y, dz = mz - pm.z;
            float val = w_s;                                                         // lane 13: weight
            float* dst = a.o_weight + slot;
            if (lane < 3) { val = lane == 0 ? mx : (lane == 1 ? my : mz); dst = a.o_xyz + 3 * slot + lane; }
            else if (lane < 6) { val = s_mom[4 + (lane - 3)] * inv_w; dst = a.o_color + 3 * slot + (lane - 3); }
            else if (lane < 12) {
                const int t = lane - 6;                                              // xx xy xz yy yz zz
                const float da = t < 3 ? dx : (t < 5 ? dy : dz);
                const float db = t == 0 ? dx : (t == 1 || t == 3 ? dy : dz);
                val = s_mom[7 + t] * inv_w - da * db;                                // mixture.cpp:211-212,236-238
                dst = a.o_cov6 + 6 * slot + t;
            } else if (lane == 12) { val = inv_w * s_mom[13]; dst = a.o_opacity + slot; }
            if (lane < 14) *dst = val;
        }
        if (G > 0) {
            // the row leaves through LDS: F consecutive floats, one coalesced store per 64
            __builtin_amdgcn_wave_barrier();
            if constexpr (mstep_packed(G)) {
                if ((lane & 15) < GG) {                          // G lanes of each row: the row's component of every float4
#pragma unroll
                    for (int v = 0; v < MSTEP_NV; ++v) s_out[4 * (gl + GG * v) + packed_slot(lane >> 4)] = shp[v] * inv_w;
                }
            } else if (grp == 0) {
#pragma unroll
                for (int v = 0; v < MSTEP_NV; ++v) {
                    const int f0 = 4 * (gl + GG * v);
                    s_out[f0] = acc[v].x * inv_w; s_out[f0 + 1] = acc[v].y * inv_w;
                    s_out[f0 + 2] = acc[v].z * inv_w; s_out[f0 + 3] = acc[v].w * inv_w;
                }
            }
            __builtin_amdgcn_wave_barrier();
            for (int f = lane; f < a.F; f += 64) a.o_sh[slot * a.F + f] = s_out[f];
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// A heavy parent's row: its segments' sums (k_mstep<.., HEAVY>) added in segment order, then the output expressions of k_mstep.
// One wave per heavy parent; element e of a segment record: 0..13 the moment sums, 16 + f the SH sums.
__global__ __launch_bounds__(64) void k_mstep_heavy_finish(MstepArgs a) {
    __shared__ float s_t[16 + 4 * MSTEP_NV * 32 + 16];         // RSH <= 4 * MSTEP_NV * 32 floats
    const int lane = threadIdx.x;
    const unsigned nheavy = a.hcount[0];
    const int RS = 16 + a.RSH;
    for (unsigned hi = blockIdx.x; hi < nheavy; hi += gridDim.x) {
        const uint4 e = a.hlist[hi];
        const MstepHeader h = a.hdr[e.x];
        const float* rec = a.hscratch + (int64_t)e.y * RS;
        for (int t = lane; t < RS; t += 64) {
            float tot = rec[t];
            for (unsigned k = 1; k < e.z; ++k) tot = tot + rec[(int64_t)k * RS + t];
            s_t[t] = tot;
        }
        __builtin_amdgcn_wave_barrier();
        const float* s_mom = s_t;
        const f3 pm = {h.px, h.py, h.pz};
        const float w_s = s_mom[0];
        const float inv_w = 1.0f / w_s;                        // mixture.cpp:209
        const int64_t slot = h.oslot;
        {
            const float mx = s_mom[1] * inv_w, my = s_mom[2] * inv_w, mz = s_mom[3] * inv_w;
            const float dx = mx - pm.x, dy = my - pm.y, dz = mz - pm.z;
            float val = w_s;
            float* dst = a.o_weight + slot;
            if (lane < 3) { val = lane == 0 ? mx : (lane == 1 ? my : mz); dst = a.o_xyz + 3 * slot + lane; }
            else if (lane < 6) { val = s_mom[4 + (lane - 3)] * inv_w; dst = a.o_color + 3 * slot + (lane - 3); }
            else if (lane < 12) {
                const int t = lane - 6;
                const float da = t < 3 ? dx : (t < 5 ? dy : dz);
                const float db = t == 0 ? dx : (t == 1 || t == 3 ? dy : dz);
                val = s_mom[7 + t] * inv_w - da * db;
                dst = a.o_cov6 + 6 * slot + t;
            } else if (lane == 12) { val = inv_w * s_mom[13]; dst = a.o_opacity + slot; }
            if (lane < 14) *dst = val;
        }
        for (int f = lane; f < a.F; f += 64) a.o_sh[slot * a.F + f] = s_t[16 + f] * inv_w;
        __builtin_amdgcn_wave_barrier();
    }
}

// Orphans: components no parent addressed (sumLw == 0) are copied unchanged after all parents (mixture.cpp:250-253), row P + (rank among the
// orphans in input order).  ONE pass over the level in INPUT order, a wave per 64 components, straight from the level's own arrays (the
// records and the cell-sorted SH copy hold the same bits): an orphan's lane copies its 14 geometry values, the SH rows then leave with
// the whole wave on a row, four rows in flight.  Input order because the ranks ascend with it: the orphans of a wave's 64 components take
// CONSECUTIVE output rows, so the wave writes one contiguous block per array -- in cell order (rounds 1-5a) every orphan's 180-byte SH row
// and its 12- / 24-byte pieces landed at a place of their own, partial cache lines the L2 could not merge before it evicted them
// (0.34 ms for 0.69 GB on the surfel level, a third of whose components are orphans).  Rounds 2-4 had four kernels for this (a pass over
// the components for the records and a slot table, then one of three SH copies chosen by the orphans' number on the host, which an
// asynchronous level does not know).  Rows beyond out_cap are not written (an asynchronous level's arrays were sized before the count
// existed; the abort flag is up then).
__global__ __launch_bounds__(256) void k_orphan_rows(int64_t n, int P, const int* __restrict__ oflag_in, const int* __restrict__ orank_in,
                                                     const float* __restrict__ i_xyz, const float* __restrict__ i_color, const float* __restrict__ i_cov6,
                                                     const float* __restrict__ i_opacity, const float* __restrict__ i_weight, const float* __restrict__ i_sh, int F,
                                                     float* o_xyz, float* o_color, float* o_cov6, float* o_opacity,
                                                     float* o_weight, float* o_sh, int64_t out_cap) {
    const int lane = threadIdx.x & 63;
    for (int64_t base = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) & ~(int64_t)63; base < n; base += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = base + lane;
        int slot = -1;
        if (i < n && oflag_in[i]) {
            const int64_t sl = (int64_t)P + orank_in[i];
            if (sl < out_cap) {
                slot = (int)sl;
                const float x = i_xyz[3 * i], y = i_xyz[3 * i + 1], z = i_xyz[3 * i + 2];
                const float c0 = i_color[3 * i], c1 = i_color[3 * i + 1], c2 = i_color[3 * i + 2];
                const float v0 = i_cov6[6 * i], v1 = i_cov6[6 * i + 1], v2 = i_cov6[6 * i + 2], v3 = i_cov6[6 * i + 3], v4 = i_cov6[6 * i + 4], v5 = i_cov6[6 * i + 5];
                const float op = i_opacity[i], wt = i_weight[i];
                o_xyz[3 * sl] = x; o_xyz[3 * sl + 1] = y; o_xyz[3 * sl + 2] = z;
                o_color[3 * sl] = c0; o_color[3 * sl + 1] = c1; o_color[3 * sl + 2] = c2;
                o_cov6[6 * sl] = v0; o_cov6[6 * sl + 1] = v1; o_cov6[6 * sl + 2] = v2;
                o_cov6[6 * sl + 3] = v3; o_cov6[6 * sl + 4] = v4; o_cov6[6 * sl + 5] = v5;
                o_opacity[sl] = op;
                o_weight[sl] = wt;
            }
        }
        if (F > 0) {
            // the SH rows of the wave's orphans: four rows in flight, the whole wave on a row
            unsigned long long m = __ballot(slot >= 0);
            while (m != 0ull) {
                const float* from[4];
                float* to[4];
                bool on[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    on[u] = m != 0ull;                                   // (uniform)
                    const int b = on[u] ? __builtin_ctzll(m) : 0;
                    m &= m - 1ull;                                       // (0 stays 0)
                    from[u] = i_sh + (base + b) * F;
                    to[u] = o_sh + (int64_t)__builtin_amdgcn_readlane(slot, b) * F;
                }
                for (int f = lane; f < F; f += 64) {
                    float v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) v[u] = on[u] ? from[u][f] : 0.0f;
#pragma unroll
                    for (int u = 0; u < 4; ++u) if (on[u]) to[u][f] = v[u];
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Parent flags from the glibc TYPE_3 rand() stream, generated ON THE DEVICE.
//
// libc rand() (what hem::rand() consumes, base.hpp:44-56) is the additive lagged-Fibonacci sequence
//     y_n = y_{n-3} + y_{n-31}  (mod 2^32),   rand() number k = y_{310+k} >> 1,
// seeded with 31 words by srand().  The recurrence is linear: with a_k = x^k mod (x^31 - x^28 - 1) over
// Z/2^32 and Y the sequence continued from any 31-word state,  y_{k-31+m} = sum_j a_k[j] * Y[m+j].
// Flag i = rand01() < 1/rho with rand01 = float(r)/float(0xffffffff), r = eight successive rand()%16
// nibbles, low nibble first (base.hpp:44-56, mixture.cpp:257-259,330).
//
// Two kernels.  k_rng_block_state: one wavefront per block of RNG_THREADS*RNG_ELEMS flags jumps to the
// block's position k_b by square-and-multiply on precomputed x^(2^b) (lane j holds coefficient j, a
// polynomial product is 31 lane-broadcast multiply-adds) and writes the 61 sequence values around k_b.
// k_flags_glibc: thread t needs the state 992 t draws further on, a_{992 t} is a per-context table, so its
// 31-word state is a 31x31 correlation of that table row with the block's 61 values; it then runs the
// recurrence for its 124 flags with the state ring in registers (the unrolled body covers 31 flags =
// 248 draws = 8 full turns of the ring, so every ring index is a compile-time constant).
// (First version: one thread jumped 48 bits by itself and kept the ring in LDS -- 0.33 ms per call at any
// size, 8 calls per bench step.)
// ------------------------------------------------------------------------------------------------
#define RNG_THREADS 64
#define RNG_ELEMS 124                          // flags per thread = 4 * 31
#define RNG_BLOCK_ELEMS (RNG_THREADS * RNG_ELEMS)
#define RNG_TAB_XPOW 0                         // [48][31]  x^(2^b)
#define RNG_TAB_T (48 * 31)                    // [31][64]  coefficient j of x^(8 * RNG_ELEMS * t)
#define RNG_TAB_YBASE (RNG_TAB_T + 31 * 64)    // [91]      the sequence continued from the seed state: y_{-31} .. y_{59}
#define RNG_TAB_WORDS (RNG_TAB_YBASE + 96)

__global__ __launch_bounds__(64) void k_rng_block_state(unsigned long long first_draw, const unsigned* __restrict__ tab,
                                                        unsigned* __restrict__ blockY /* [nblocks][64] */, const long long* __restrict__ n_dev) {
    if (n_dev && (long long)blockIdx.x * RNG_BLOCK_ELEMS >= *n_dev) return;       // launched for a bound: the blocks beyond the real count
    const int lane = threadIdx.x;
    const unsigned long long k = 310ull + 8ull * (first_draw + (unsigned long long)blockIdx.x * RNG_BLOCK_ELEMS);
    unsigned a = lane == 0 ? 1u : 0u;                       // x^0
    for (int b = 0; b < 48; ++b) {
        if (!((k >> b) & 1ull)) continue;
        unsigned v = lane < 31 ? tab[RNG_TAB_XPOW + b * 31 + lane] : 0u;   // v = x^i * xb mod P, i = 0
        unsigned c = 0;
        for (int i = 0; i < 31; ++i) {
            c += (unsigned)__shfl((int)a, i) * v;
            const unsigned top = (unsigned)__shfl((int)v, 30), up = (unsigned)__shfl_up((int)v, 1);
            v = lane == 0 ? top : up;                       // x * v:  x^31 = x^28 + 1
            if (lane == 28) v += top;
            if (lane > 30) v = 0u;
        }
        a = c;
    }
    unsigned y = 0;                                         // lane m: y_{k-31+m} = sum_j a[j] * Ybase[m+j]
    for (int j = 0; j < 31; ++j) {
        const unsigned aj = (unsigned)__shfl((int)a, j);
        if (lane < 61) y += aj * tab[RNG_TAB_YBASE + lane + j];
    }
    blockY[(size_t)blockIdx.x * 64 + lane] = y;
}

__global__ __launch_bounds__(RNG_THREADS) void k_flags_glibc(int64_t n, float prob, const unsigned* __restrict__ tab,
                                                             const unsigned* __restrict__ blockY, uint8_t* __restrict__ is_parent,
                                                             const long long* __restrict__ n_dev) {
    if (n_dev && *n_dev < n) n = *n_dev;
    const int t = threadIdx.x;
    const int64_t i0 = ((int64_t)blockIdx.x * RNG_THREADS + t) * RNG_ELEMS;
    if (i0 >= n) return;
    unsigned T[31];
#pragma unroll
    for (int j = 0; j < 31; ++j) T[j] = tab[RNG_TAB_T + j * 64 + t];
    const unsigned* Yb = blockY + (size_t)blockIdx.x * 64;  // uniform: scalar loads
    unsigned ring[31];                                      // slot i holds y_{k-31+i}; y_n lives in slot (n-k) mod 31
#pragma unroll
    for (int i = 0; i < 31; ++i) {
        unsigned acc = 0;
#pragma unroll
        for (int j = 0; j < 31; ++j) acc += T[j] * Yb[i + j];
        ring[i] = acc;
    }
    for (int it = 0; it < RNG_ELEMS / 31; ++it) {
#pragma unroll
        for (int e = 0; e < 31; ++e) {
            unsigned x = 0;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int f = (e * 8 + q) % 31, r = (e * 8 + q + 28) % 31;   // slots of y_{n-31} (overwritten by y_n) and y_{n-3}
                const unsigned v = ring[f] + ring[r];
                ring[f] = v;
                x |= ((v >> 1) & 15u) << (4 * q);
            }
            const int64_t i = i0 + it * 31 + e;
            const float r01 = (float)x / 4294967296.0f;
            if (i < n) is_parent[i] = r01 < prob ? 1 : 0;
        }
    }
}
__device__ __forceinline__ unsigned hash32(unsigned long long x) {      // splitmix64 finaliser
    x += 0x9e3779b97f4a7c15ull;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
    x = x ^ (x >> 31);
    return (unsigned)(x >> 32);
}
__global__ __launch_bounds__(256) void k_flags_hash(int64_t n, unsigned seed, unsigned long long first_draw, float prob,
                                                    uint8_t* __restrict__ is_parent, const long long* __restrict__ n_dev) {
    if (n_dev && *n_dev < n) n = *n_dev;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const unsigned r = hash32(((unsigned long long)seed << 40) ^ (first_draw + (unsigned long long)i));
        const float r01 = (float)r / 4294967296.0f;
        is_parent[i] = r01 < prob ? 1 : 0;
    }
}

// Work-sharded level: the merged components of the parents [lo, lo + cnt) of the cell-sorted parent list as packed rows
// {xyz 3, color 3, cov6 6, opacity, weight, sh F} (PACK) -- one all-gather of equal chunks moves them -- and back into the
// output rows of their owners' slots (UNPACK, for every rank's chunk).  slot = rank of the parent in input order.
template <bool PACK>
__global__ __launch_bounds__(256) void k_shard_rows(int lo, int cnt, int F, const unsigned* __restrict__ plist, const unsigned* __restrict__ order,
                                                    const int* __restrict__ prank_in, float* __restrict__ packed, float* o_xyz, float* o_color,
                                                    float* o_cov6, float* o_opacity, float* o_weight, float* o_sh) {
    const int RW = 14 + F;
    const int64_t total = (int64_t)cnt * RW;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(t / RW), f = (int)(t - (int64_t)i * RW);
        const int64_t slot = prank_in[order[plist[lo + i]]];
        float* dst = f < 3 ? o_xyz + 3 * slot + f : f < 6 ? o_color + 3 * slot + (f - 3) : f < 12 ? o_cov6 + 6 * slot + (f - 6)
                     : f == 12 ? o_opacity + slot : f == 13 ? o_weight + slot : o_sh + slot * F + (f - 14);
        if (PACK) packed[t] = *dst;
        else *dst = packed[t];
    }
}

// ------------------------------------------------------------------------------------------------
// Spatially partitioned levels (BASELINE config 5): ONE large cloud over several GPUs.
//
// Every rank OWNS a subset of the level's components (a spatial slab at level 0; the merged components of its parents and its
// orphans afterwards) and carries their GLOBAL indices gid (= positions in the level's global order: parents by ascending index,
// then orphans, mixture.cpp:169-253).  A level then runs on owned + GHOST components:
//   grid      the ranks all-reduce the bounding box and the three axis histograms (integers: exact), so every rank derives the
//             SAME grid as a single GPU would;
//   ghosts    every rank marks, in a bit mask over the grid's cells, the cells the search spheres of its owned parents touch
//             (k_mark_cells); the masks are all-gathered, and a rank sends each of its owned components to the ranks whose mask
//             holds its cell (one personalised exchange of 64 + 4F + 8-byte rows).  A parent then finds, in every cell its
//             rows scan, exactly the components a single GPU holds there, in the same order (cell sort stable over the gid
//             order): same candidates, same survivors, same pairs in the same order -- long-range parents simply mark more cells;
//   sums      a child's wL terms come from parents on several ranks.  The deterministic LDS fixed point of k_bucket_sum is split
//             in three (k_part_max / k_part_acc / k_part_finish) with the ghosts' partial results sent to their owners in between:
//             first the largest |wL| (it sets the child's scale: integer max), then the 64-bit fixed-point sums (integer adds),
//             then the finished float32 sum back -- integers throughout, so the sum is BIT FOR BIT the single-GPU one;
//   output    a parent's output row is its rank among ALL parents by global index: the ranks all-reduce a bit map of parent gids
//             (and one of orphan gids; disjoint bits: an integer sum is an OR) and prefix-count it; the new flags are drawn from
//             the libc stream at those global ranks.  No floating-point value is ever combined across ranks.
// ------------------------------------------------------------------------------------------------
#define PART_ROW_EXTRA 2      // a ghost row: 16 floats of the packed record, gid, index at its owner (72 bytes); its F SH floats travel apart

// Bits of the cells an owned parent can take candidates from -- two masks per rank:
//   mask_all  the box of its search sphere (the rows of select_scan's pass B): where IRREGULAR components are wanted;
//   mask_reg  that box cut down, along y and z, to the box of the parent's pre-reject ellipsoid (make_filter's ey / ez, exactly
//             the rows pass A enumerates): where REGULAR components are wanted -- a regular child outside the ellipsoid is a
//             certain rejection, pass A never looks at its cell, so it need not travel.  (40 M splats on 8 ranks: the halo of a
//             block falls from 40 % to about a quarter of the rank's own components.)
// 16 lanes per parent; a word is tested before the atomic (neighbours mark the same cells).
__device__ __forceinline__ void mark_box(const GridParams& g, int sub, int x0, int x1, int y0, int y1, int z0, int z1, unsigned* __restrict__ mask) {
    const int ny = y1 - y0 + 1, nrows = ny * (z1 - z0 + 1);
    for (int r = sub; r < nrows; r += 16) {
        const int64_t c0 = ((int64_t)(z0 + r / ny) * g.gy + (y0 + r % ny)) * g.gx + x0, c1 = c0 + (x1 - x0);
        for (int64_t w = c0 >> 5; w <= (c1 >> 5); ++w) {
            const int lo = w == (c0 >> 5) ? (int)(c0 & 31) : 0, hi = w == (c1 >> 5) ? (int)(c1 & 31) : 31;
            const unsigned bits = (hi == 31 ? 0xffffffffu : ((1u << (hi + 1)) - 1u)) & ~((1u << lo) - 1u);
            if ((mask[w] & bits) != bits) atomicOr(&mask[w], bits);
        }
    }
}
__global__ __launch_bounds__(256) void k_mark_cells(int64_t n_own, const float4* __restrict__ rec, const GridParams* __restrict__ gpp, float delta, float kldThr,
                                                    int ell, unsigned* __restrict__ mask_reg, unsigned* __restrict__ mask_all) {
    const GridParams g = *gpp;
    const int sub = threadIdx.x & 15;
    const int64_t stride = ((int64_t)gridDim.x * blockDim.x) >> 4;
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4; i < n_own; i += stride) {
        const float4 a = rec[4 * i];
        if (!(__float_as_uint(a.w) & 1u)) continue;
        const float4 b = rec[4 * i + 1], cc = rec[4 * i + 2], d = rec[4 * i + 3];
        const s6 cov = {b.x, b.y, b.z, b.w, cc.x, cc.y};
        const float R = delta * sqrtf(eig_max6(cov));
        const bool finite = fabsf(a.x) <= FLT_MAX && fabsf(a.y) <= FLT_MAX && fabsf(a.z) <= FLT_MAX;
        if (!(R * R > 0.0f) || !finite) continue;                       // no children at all (k_parent_prep: active = 0)
        const float Ra = fabsf(R) * 1.00001f + g.slack;
        const int x0 = cell_of(a.x - Ra, g.ox, g.inv_c, g.gx), x1 = cell_of(a.x + Ra, g.ox, g.inv_c, g.gx);
        const int y0 = cell_of(a.y - Ra, g.oy, g.inv_c, g.gy), y1 = cell_of(a.y + Ra, g.oy, g.inv_c, g.gy);
        const int z0 = cell_of(a.z - Ra, g.oz, g.inv_c, g.gz), z1 = cell_of(a.z + Ra, g.oz, g.inv_c, g.gz);
        mark_box(g, sub, x0, x1, y0, y1, z0, z1, mask_all);
        // the same record k_parent_prep builds for this parent (same float32 inverse, same filter): pass A's rows
        ParentRec pr;
        const float det_p = d.w;
        make_filter(inverse6(cov, det_p), det_p, kldThr, ell, (__float_as_uint(a.w) & 2u) != 0u, pr);
        if (pr.ec.on != 0.0f) {
            const float Ry = fminf(Ra, pr.ey + g.slack), Rz = fminf(Ra, pr.ez + g.slack);
            const int yy0 = cell_of(a.y - Ry, g.oy, g.inv_c, g.gy), yy1 = cell_of(a.y + Ry, g.oy, g.inv_c, g.gy);
            const int zz0 = cell_of(a.z - Rz, g.oz, g.inv_c, g.gz), zz1 = cell_of(a.z + Rz, g.oz, g.inv_c, g.gz);
            mark_box(g, sub, x0, x1, yy0, yy1, zz0, zz1, mask_reg);
        } else {
            mark_box(g, sub, x0, x1, y0, y1, z0, z1, mask_reg);
        }
    }
}
// dflag[q][i] = 1 when owned component i lies in a cell rank q marked (q != self): in its mask for regular components or, an
// irregular one, in its sphere mask.  masks: per rank [mask_reg | mask_all], mask_words words each.
__global__ __launch_bounds__(256) void k_dest_flags(int64_t n_own, const float4* __restrict__ rec, const GridParams* __restrict__ gpp, int world, int self,
                                                    int64_t mask_words, const unsigned* __restrict__ masks, int* __restrict__ dflag) {
    const GridParams g = *gpp;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_own; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 a = rec[4 * i];
        const int64_t c = ((int64_t)cell_of(a.z, g.oz, g.inv_c, g.gz) * g.gy + cell_of(a.y, g.oy, g.inv_c, g.gy)) * g.gx + cell_of(a.x, g.ox, g.inv_c, g.gx);
        const int64_t which = (__float_as_uint(a.w) & 2u) ? 0 : mask_words;          // regular: mask_reg, irregular: mask_all
        for (int q = 0; q < world; ++q)
            dflag[(int64_t)q * n_own + i] = q != self && ((masks[(int64_t)q * 2 * mask_words + which + (c >> 5)] >> (c & 31)) & 1u) ? 1 : 0;
    }
}
// rows to send: for destination q the k-th flagged owned component (pos = exclusive scan of its flags)
__global__ __launch_bounds__(256) void k_pack_rows(int64_t n_own, int F, const int* __restrict__ dflag, const int* __restrict__ dpos,
                                                   const float4* __restrict__ rec, const float* __restrict__ sh, const unsigned* __restrict__ gid,
                                                   float* __restrict__ rows, float* __restrict__ sh_rows, unsigned* __restrict__ sent_idx) {
    constexpr int RW = 16 + PART_ROW_EXTRA;
    const int lane = threadIdx.x & 63;
    const int64_t stride = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; i < n_own; i += stride) {      // a wave per component
        if (!dflag[i]) continue;
        const int64_t k = dpos[i];
        float* row = rows + k * RW;
        const float* r = reinterpret_cast<const float*>(rec + 4 * i);
        if (lane < 16) row[lane] = r[lane];
        for (int f = lane; f < F; f += 64) sh_rows[k * F + f] = sh[i * F + f];
        if (lane == 0) { row[16] = __uint_as_float(gid[i]); row[17] = __uint_as_float((unsigned)i); sent_idx[k] = (unsigned)i; }
    }
}
// received rows -> the ghosts' records behind the owned ones, their global indices and indices at their owners (their SH rows arrive
// by a second exchange straight into ghost_sh, in the same order)
__global__ __launch_bounds__(256) void k_unpack_rows(int64_t n_ghost, int64_t n_own, const float* __restrict__ rows, float4* __restrict__ rec_loc,
                                                     unsigned* __restrict__ gid_loc, unsigned* __restrict__ ghost_src) {
    constexpr int RW = 16 + PART_ROW_EXTRA;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_ghost * 16; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t k = t >> 4;
        const int f = (int)(t & 15);
        const float* row = rows + k * RW;
        reinterpret_cast<float*>(rec_loc + 4 * (n_own + k))[f] = row[f];
        if (f == 0) { gid_loc[n_own + k] = __float_as_uint(row[16]); ghost_src[k] = __float_as_uint(row[17]); }
    }
}
// cell keys of the local components in the order perm (ascending global index): the stable cell sort then keeps that order inside a cell
__global__ __launch_bounds__(256) void k_keys_rec(int64_t n, const float4* __restrict__ rec, const unsigned* __restrict__ perm, const GridParams* __restrict__ gpp,
                                                  unsigned* __restrict__ keys, unsigned* __restrict__ idx) {
    const GridParams g = *gpp;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
        const unsigned i = perm[k];
        const float4 a = rec[4 * (int64_t)i];
        keys[k] = (unsigned)((cell_of(a.z, g.oz, g.inv_c, g.gz) * g.gy + cell_of(a.y, g.oy, g.inv_c, g.gy)) * g.gx + cell_of(a.x, g.ox, g.inv_c, g.gx));
        idx[k] = i;
    }
}
__global__ __launch_bounds__(256) void k_iota(int64_t n, unsigned* __restrict__ p) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = (unsigned)i;
}
// pown[j] = the component at sorted position j is a parent this rank owns;  inv[local index] = sorted position
__global__ __launch_bounds__(256) void k_own_flags(int64_t n, int64_t n_own, const unsigned* __restrict__ order, const int* __restrict__ pflag,
                                                   int* __restrict__ pown, unsigned* __restrict__ inv) {
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) {
        const unsigned i = order[j];
        pown[j] = pflag[j] && (int64_t)i < n_own ? 1 : 0;
        inv[i] = (unsigned)j;
    }
}
// shs rows from two sources: owned components from the level's SH array, ghosts from theirs
__global__ __launch_bounds__(256) void k_gather_sh2(int64_t n, int64_t n_own, int F, int RSH, const unsigned* __restrict__ order,
                                                    const float* __restrict__ sh_own, const float* __restrict__ sh_ghost, float* __restrict__ shs) {
    const int Q = RSH >> 2;
    const int64_t total = n * Q;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t j = t / Q;
        const int q = (int)(t - j * Q);
        const int64_t i = order[j];
        const float* src = (i < n_own ? sh_own + i * F : sh_ghost + (i - n_own) * F) + 4 * q;
        const int left = F - 4 * q;
        float4 v;
        if (left >= 4) {
            const f4u u = *reinterpret_cast<const f4u*>(src);
            v = make_float4(u.x, u.y, u.z, u.w);
        } else {
            v.x = left > 0 ? src[0] : 0.0f; v.y = left > 1 ? src[1] : 0.0f; v.z = left > 2 ? src[2] : 0.0f; v.w = 0.0f;
        }
        reinterpret_cast<float4*>(shs + j * RSH)[q] = v;
    }
}
// k_bucket_sum in three steps with global per-child arrays (sorted positions): the largest |wL| ...
__global__ __launch_bounds__(1024) void k_part_max(int64_t n, int shift, unsigned cap, const unsigned* __restrict__ cursor, const slot_t* __restrict__ child,
                                                   const float* __restrict__ wl, unsigned* __restrict__ gmax) {
    extern __shared__ unsigned s_m[];
    const int bucket = 1 << shift;
    const int b = blockIdx.x;
    const int64_t c0 = (int64_t)b << shift;
    const int nc = (int)(n - c0 < bucket ? n - c0 : bucket);
    for (int i = threadIdx.x; i < bucket; i += blockDim.x) s_m[i] = 0u;
    __syncthreads();
    const unsigned cnt = cursor[b];
    const unsigned long long k0 = (unsigned long long)b * cap, k1 = k0 + (cnt < cap ? cnt : cap);
    for (unsigned long long k = k0 + threadIdx.x; k < k1; k += blockDim.x) atomicMax(&s_m[child[k]], __float_as_uint(wl[k]) & 0x7fffffffu);
    __syncthreads();
    for (int i = threadIdx.x; i < nc; i += blockDim.x) gmax[c0 + i] = s_m[i];
}
// ... the fixed-point sums on the scale of the (now global) maximum: 64-bit integers, or flag bits where a term is not finite ...
__global__ __launch_bounds__(1024) void k_part_acc(int64_t n, int shift, unsigned cap, const unsigned* __restrict__ cursor, const slot_t* __restrict__ child,
                                                   const float* __restrict__ wl, const unsigned* __restrict__ gmax, unsigned long long* __restrict__ gacc) {
    extern __shared__ unsigned long long s_acc[];
    const int bucket = 1 << shift;
    unsigned* s_max = (unsigned*)(s_acc + bucket);
    const int b = blockIdx.x;
    const int64_t c0 = (int64_t)b << shift;
    const int nc = (int)(n - c0 < bucket ? n - c0 : bucket);
    for (int i = threadIdx.x; i < bucket; i += blockDim.x) { s_acc[i] = 0ull; s_max[i] = i < nc ? gmax[c0 + i] : 0u; }
    __syncthreads();
    const unsigned cnt = cursor[b];
    const unsigned long long k0 = (unsigned long long)b * cap, k1 = k0 + (cnt < cap ? cnt : cap);
    for (unsigned long long k = k0 + threadIdx.x; k < k1; k += blockDim.x) {
        const unsigned i = child[k];
        const float w = wl[k];
        const unsigned mb = s_max[i];
        if (mb >= 0x7f800000u) {                   // a non-finite term somewhere: collect flags (1 +inf, 2 -inf, 4 NaN)
            const unsigned wb = __float_as_uint(w);
            const unsigned long long f = (wb & 0x7fffffffu) > 0x7f800000u ? 4ull : (wb == 0x7f800000u ? 1ull : (wb == 0xff800000u ? 2ull : 0ull));
            if (f) atomicOr(&s_acc[i], f);
            continue;
        }
        int e = (int)(mb >> 23);
        e = e < 1 ? 1 : e;
        const int k2 = 38 - (e - 127);
        const double scale = __longlong_as_double((long long)(1023 + k2) << 52);
        atomicAdd(&s_acc[i], (unsigned long long)__double2ll_rn((double)w * scale));
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nc; i += blockDim.x) gacc[c0 + i] = s_acc[i];
}
// ... and the float32 sum of the components this rank owns (the owners' totals go back to the ghosts afterwards)
__global__ __launch_bounds__(256) void k_part_finish(int64_t n, const unsigned* __restrict__ gmax, const unsigned long long* __restrict__ gacc,
                                                     float* __restrict__ sumLw) {
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) {
        const unsigned mb = gmax[j];
        float sres;
        if (mb >= 0x7f800000u) {
            const unsigned long long f = gacc[j];
            sres = (f & 4ull) || ((f & 1ull) && (f & 2ull)) ? __builtin_nanf("") : ((f & 1ull) ? __builtin_inff() : -__builtin_inff());
        } else {
            int e = (int)(mb >> 23);
            e = e < 1 ? 1 : e;
            const int k2 = 38 - (e - 127);
            const double inv = __longlong_as_double((long long)(1023 - k2) << 52);
            sres = (float)((double)(long long)gacc[j] * inv);
        }
        sumLw[j] = sres;
    }
}
// sumLw into the geometry records; orphans are decided by a child's OWNER only
__global__ __launch_bounds__(256) void k_part_orphans(int64_t n, int64_t n_own, const unsigned* __restrict__ order, const float* __restrict__ sumLw,
                                                      int* __restrict__ orphan_flag, float* __restrict__ geo_sl) {
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) {
        const float s = sumLw[j];
        geo_sl[16 * j] = s;
        orphan_flag[j] = s == 0.0f && (int64_t)order[j] < n_own ? 1 : 0;
    }
}
// values of the ghosts (local index n_own + k -> sorted position inv[..]) into a send buffer, and received values applied at
// the owned components the rows went out for (sent_idx, the order they were sent in).  OP: 0 max (u32), 1 add (u64), 2 or (u64,
// where the child's maximum is not finite: flag bits), 3 set
template <typename T>
__global__ __launch_bounds__(256) void k_ghost_gather(int64_t cnt, int64_t first_local, const unsigned* __restrict__ inv, const T* __restrict__ v, T* __restrict__ out) {
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < cnt; k += (int64_t)gridDim.x * blockDim.x) out[k] = v[inv[first_local + k]];
}
template <typename T>
__global__ __launch_bounds__(256) void k_sent_gather(int64_t cnt, const unsigned* __restrict__ sent_idx, const unsigned* __restrict__ inv, const T* __restrict__ v,
                                                     T* __restrict__ out) {
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < cnt; k += (int64_t)gridDim.x * blockDim.x) out[k] = v[inv[sent_idx[k]]];
}
__global__ __launch_bounds__(256) void k_sent_apply_max(int64_t cnt, const unsigned* __restrict__ sent_idx, const unsigned* __restrict__ inv,
                                                        const unsigned* __restrict__ in, unsigned* __restrict__ gmax) {
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < cnt; k += (int64_t)gridDim.x * blockDim.x) atomicMax(&gmax[inv[sent_idx[k]]], in[k]);
}
__global__ __launch_bounds__(256) void k_sent_apply_acc(int64_t cnt, const unsigned* __restrict__ sent_idx, const unsigned* __restrict__ inv,
                                                        const unsigned long long* __restrict__ in, const unsigned* __restrict__ gmax,
                                                        unsigned long long* __restrict__ gacc) {
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < cnt; k += (int64_t)gridDim.x * blockDim.x) {
        const unsigned j = inv[sent_idx[k]];
        if (gmax[j] >= 0x7f800000u) atomicOr(&gacc[j], in[k]);        // flag bits
        else atomicAdd(&gacc[j], in[k]);                               // two's complement fixed point: integer addition
    }
}
template <typename T>
__global__ __launch_bounds__(256) void k_ghost_set(int64_t cnt, int64_t first_local, const unsigned* __restrict__ inv, const T* __restrict__ in, T* __restrict__ v) {
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < cnt; k += (int64_t)gridDim.x * blockDim.x) v[inv[first_local + k]] = in[k];
}
// bit maps over the level's global indices, and ranks out of them
__global__ __launch_bounds__(256) void k_bits_set(int64_t n_own, const unsigned* __restrict__ gid, const int* __restrict__ flag, unsigned* __restrict__ bits) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_own; i += (int64_t)gridDim.x * blockDim.x)
        if (flag[i]) atomicOr(&bits[gid[i] >> 5], 1u << (gid[i] & 31));
}
__global__ __launch_bounds__(256) void k_bits_popc(int64_t words, const unsigned* __restrict__ bits, int* __restrict__ cnt) {
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < words; w += (int64_t)gridDim.x * blockDim.x) cnt[w] = __popc(bits[w]);
}
// global rank of every flagged owned component: set bits below its gid (+ base); others keep what they have
__global__ __launch_bounds__(256) void k_bits_rank(int64_t n_own, const unsigned* __restrict__ gid, const int* __restrict__ flag, const unsigned* __restrict__ bits,
                                                   const int* __restrict__ wpre, unsigned base, unsigned* __restrict__ rank) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_own; i += (int64_t)gridDim.x * blockDim.x)
        if (flag[i]) { const unsigned g = gid[i]; rank[i] = base + (unsigned)wpre[g >> 5] + (unsigned)__popc(bits[g >> 5] & ((1u << (g & 31)) - 1u)); }
}
// new gid of every output row: parents' rows by their local parent rank, orphans' rows behind them by their local orphan rank
// (a component can be BOTH: a parent nobody selects -- not even itself: a degenerate needle whose self pair fails the gates -- has
// sumLw == 0 and is copied as an orphan beside its merged row, mixture.cpp:250-253: two rows, two ranks)
__global__ __launch_bounds__(256) void k_part_new_gid(int64_t n_own, int P_loc, const int* __restrict__ pflag_in, const int* __restrict__ prank_in,
                                                      const int* __restrict__ oflag_in, const int* __restrict__ orank_in, const unsigned* __restrict__ grank_p,
                                                      const unsigned* __restrict__ grank_o, unsigned* __restrict__ new_gid) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_own; i += (int64_t)gridDim.x * blockDim.x) {
        if (pflag_in[i]) new_gid[prank_in[i]] = grank_p[i];
        if (oflag_in[i]) new_gid[P_loc + orank_in[i]] = grank_o[i];
    }
}
__global__ __launch_bounds__(256) void k_gather_bytes(int64_t n, const unsigned* __restrict__ idx, const uint8_t* __restrict__ src, uint8_t* __restrict__ dst) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[idx[i]];
}
// gsr_hem_set_level0_part trusts nothing about the caller's global indices: every later kernel indexes n_global-sized tables with them
__global__ __launch_bounds__(256) void k_check_gids(int64_t n, int64_t n_global, const unsigned* __restrict__ gid, unsigned* __restrict__ bad) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const unsigned g = gid[i];
        if ((int64_t)g >= n_global || (i > 0 && gid[i - 1] >= g)) atomicAdd(bad, 1u);
    }
}
// the erased rows' global ranks leave the numbering: new gid -= erased rows below it
__global__ __launch_bounds__(256) void k_gid_drop(int64_t n, const unsigned* __restrict__ bits, const int* __restrict__ wpre, unsigned* __restrict__ gid) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const unsigned g = gid[i];
        gid[i] = g - ((unsigned)wpre[g >> 5] + (unsigned)__popc(bits[g >> 5] & ((1u << (g & 31)) - 1u)));
    }
}
__global__ __launch_bounds__(256) void k_compact_u32(int64_t n, const int* __restrict__ keep, const int* __restrict__ pos, const unsigned* __restrict__ src,
                                                     unsigned* __restrict__ dst) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        if (keep[i]) dst[pos[i]] = src[i];
}
__global__ __launch_bounds__(256) void k_not_flag(int64_t n, const int* __restrict__ in, int* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = in[i] ? 0 : 1;
}
__global__ void k_last_total(const int* __restrict__ pos_last, const int* __restrict__ flag_last, long long* __restrict__ out) { *out = (long long)*pos_last + *flag_last; }
__global__ void k_flip3(unsigned* __restrict__ w) { if (threadIdx.x < 3) w[threadIdx.x] = ~w[threadIdx.x]; }

// validity (mixture.cpp:262-282): keep iff !(isnan(mean) || isnan(det) || det <= 0)
__global__ __launch_bounds__(256) void k_valid(int64_t n, const float* __restrict__ xyz, const float* __restrict__ cov6,
                                               int* __restrict__ keep, const long long* __restrict__ n_dev, int* __restrict__ dropped,
                                               int* __restrict__ holes /* [ERASE_MAX]: the erased rows, in any order (NULL: not wanted) */) {
    if (n_dev) n = *n_dev;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
        s6 c = {cov6[6 * i], cov6[6 * i + 1], cov6[6 * i + 2], cov6[6 * i + 3], cov6[6 * i + 4], cov6[6 * i + 5]};
        const float d = det6(c);
        const bool bad = (x != x) || (y != y) || (z != z) || (d != d) || (d <= 0.0f);
        keep[i] = bad ? 0 : 1;
        if (bad && dropped) {                            // (erased rows are a handful per level: the counter says whether anything has to move, the list what)
            const int slot = atomicAdd(dropped, 1);
            if (holes && slot < ERASE_MAX) holes[slot] = (int)i;
        }
    }
}

// The validity erase IN PLACE (mixture.cpp:262-282 erases element by element, O(n) each).  A level drops a handful of rows (a surfel level of
// 3.1 M rows: 22), and rounds 1-4 moved ALL rows of all seven arrays through a second buffer for it (and back, when the level lives in the
// caller's arrays: 0.4 ms of a 7 ms level) behind a host round trip.  Here the rows behind the first hole slide down by the number of holes
// below them, in tiles of ERASE_TILE rows: a tile's rows land on its own rows and on the last rows of the tile BELOW it, so those last rows
// (at most as many as there are holes) are saved first (k_erase_save: one small copy per tile), every tile then stages its surviving
// rows in LDS -- the ones the tile above may already have overwritten come from the saved copy -- and writes them out as one contiguous run
// (k_erase_shift).  Tiles in front of the first hole do nothing.  Up to ERASE_MAX holes; a level with more takes the old path (host).
// A tile of T rows x w floats is staged at once: T = min(256, 12288 / widest row), so that 48 KB of LDS hold it.
struct EraseArgs {
    float* arr[6];                  // xyz color cov6 opacity weight sh
    int w[6];                       // floats per row: 3 3 6 1 1 F
    uint8_t* flags;                 // is_parent
    float* halo;                    // [tiles][ERASE_MAX][W] floats, then [tiles][ERASE_MAX] bytes
    int W;                          // 14 + F
    int tile;                       // rows per tile
    int max_tiles;                  // tiles the halo buffer holds (the launch's bound)
    const int* holes;               // k_valid's list
    const int* dropped;             // its count
    const long long* n_pre_p;       // rows before the erase (device)
    long long* n_keep_out;          // rows after it (written by k_erase_save)
};
// the sorted holes of the level into LDS; returns their number (0: nothing to do -- none, or more than this path takes)
__device__ __forceinline__ int erase_load_holes(const EraseArgs& a, int* s_holes) {
    const int d = *a.dropped;
    if (d <= 0 || d > ERASE_MAX) return 0;
    if ((int)threadIdx.x < d) {                     // rank sort: at most 32 values
        const int v = a.holes[threadIdx.x];
        int r = 0;
        for (int k = 0; k < d; ++k) r += a.holes[k] < v ? 1 : 0;
        s_holes[r] = v;
    }
    __syncthreads();
    return d;
}
__device__ __forceinline__ int holes_below(const int* s_holes, int d, int64_t row) {      // holes with index < row
    int k = 0;
    for (int i = 0; i < d; ++i) k += (int64_t)s_holes[i] < row ? 1 : 0;
    return k;
}
__global__ __launch_bounds__(256) void k_erase_save(EraseArgs a) {
    __shared__ int s_holes[ERASE_MAX];
    const long long n_pre = *a.n_pre_p;
    const int d = erase_load_holes(a, s_holes);
    if (blockIdx.x == 0 && threadIdx.x == 0) *a.n_keep_out = n_pre - (d > 0 ? d : 0);      // (more than ERASE_MAX: the host finishes and sets the size)
    if (d == 0) return;
    const int64_t ntiles = (n_pre + a.tile - 1) / a.tile;
    uint8_t* halo_b = reinterpret_cast<uint8_t*>(a.halo + (int64_t)a.max_tiles * ERASE_MAX * a.W);
    for (int64_t t = 1 + blockIdx.x; t < ntiles; t += gridDim.x) {
        const int64_t row0 = t * a.tile;
        const int k0 = holes_below(s_holes, d, row0);
        if (k0 == 0) continue;
        float* h = a.halo + t * ERASE_MAX * a.W;
        int off = 0;
        for (int q = 0; q < 6; ++q) {
            const int w = a.w[q];
            if (w > 0) for (int e = threadIdx.x; e < k0 * w; e += blockDim.x) h[off + e] = a.arr[q][(row0 - k0) * w + e];
            off += ERASE_MAX * w;
        }
        if ((int)threadIdx.x < k0) halo_b[t * ERASE_MAX + threadIdx.x] = a.flags[row0 - k0 + threadIdx.x];
    }
}
__global__ __launch_bounds__(256) void k_erase_shift(EraseArgs a) {
    extern __shared__ float s_stage[];              // tile x widest row floats
    __shared__ int s_holes[ERASE_MAX];
    __shared__ short s_k[256];                      // per row of the tile: holes below it (relative to the tile's first row), -1 = the row is a hole
    const long long n_pre = *a.n_pre_p;
    const int d = erase_load_holes(a, s_holes);
    if (d == 0) return;
    const int64_t ntiles = (n_pre + a.tile - 1) / a.tile;
    const uint8_t* halo_b = reinterpret_cast<const uint8_t*>(a.halo + (int64_t)a.max_tiles * ERASE_MAX * a.W);
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int64_t row0 = t * a.tile;
        const int nrows = (int)(n_pre - row0 < a.tile ? n_pre - row0 : a.tile);
        const int k0 = holes_below(s_holes, d, row0), k1 = holes_below(s_holes, d, row0 + nrows);
        if (k1 == 0) continue;                      // in front of the first hole: nothing moves
        // rows of THIS tile the tile above overwrites (its first rows land on them): read from the saved copy
        const int kn = row0 + nrows < n_pre ? k1 : 0;
        const int tail0 = nrows - kn;               // local index of the first such row
        __syncthreads();
        if ((int)threadIdx.x < nrows) {
            const int64_t r = row0 + threadIdx.x;
            int k = 0; bool hole = false;
            for (int i = 0; i < d; ++i) { k += (int64_t)s_holes[i] < r ? 1 : 0; hole = hole || (int64_t)s_holes[i] == r; }
            s_k[threadIdx.x] = hole ? (short)-1 : (short)(k - k0);
        }
        __syncthreads();
        const int ndst = nrows - (k1 - k0);
        const float* hn = a.halo + (t + 1) * ERASE_MAX * a.W;
        int off = 0;
        for (int q = 0; q < 6; ++q) {
            const int w = a.w[q];
            if (w > 0) {
                float* arr = a.arr[q];
                for (unsigned e = threadIdx.x; e < (unsigned)(nrows * w); e += blockDim.x) {
                    const unsigned rl = e / (unsigned)w, cc = e - rl * (unsigned)w;
                    const int kk = s_k[rl];
                    if (kk < 0) continue;
                    const float v = (int)rl >= tail0 ? hn[off + ((int)rl - tail0) * w + cc] : arr[(row0 + rl) * w + cc];
                    s_stage[((int)rl - kk) * w + cc] = v;
                }
                __syncthreads();
                for (int e = threadIdx.x; e < ndst * w; e += blockDim.x) arr[(row0 - k0) * w + e] = s_stage[e];
                __syncthreads();
            }
            off += ERASE_MAX * w;
        }
        {   // the flag bytes
            uint8_t* sb = reinterpret_cast<uint8_t*>(s_stage);
            if ((int)threadIdx.x < nrows && s_k[threadIdx.x] >= 0) {
                const int rl = threadIdx.x;
                sb[rl - s_k[rl]] = rl >= tail0 ? halo_b[(t + 1) * ERASE_MAX + (rl - tail0)] : a.flags[row0 + rl];
            }
            __syncthreads();
            if ((int)threadIdx.x < ndst) a.flags[row0 - k0 + threadIdx.x] = sb[threadIdx.x];
            __syncthreads();
        }
    }
}

// The counts of an asynchronous level stay on the device.  lvl[0] = rows of the new level before the validity erase (parents + orphans),
// lvl[1] = orphans.  The arrays the rows go to were sized before that number existed: if they are too small the count is clamped (the
// kernels behind this one only touch rows that exist), the abort flag goes up and the host reruns the level the synchronous way.
__global__ void k_level_tail(int64_t n, int P, const int* __restrict__ orank_in, const int* __restrict__ oflag_in, long long out_cap,
                             long long* __restrict__ lvl, int* __restrict__ abort_p) {
    long long n_orph = (long long)orank_in[n - 1] + oflag_in[n - 1];
    if ((long long)P + n_orph > out_cap) { *abort_p = 1; n_orph = out_cap > P ? out_cap - P : 0; }
    lvl[0] = (long long)P + n_orph;
    lvl[1] = n_orph;
}

// Everything the host wants to know about a level, in ONE round trip behind its last kernel (an asynchronous level; the synchronous one
// uses it for the next level's prologue): the totals of the two scans, the counters, the next level's grid.  16 threads, word t of the
// answer each; the values first, a system-scope fence, then the sequence number (see k_collect).
struct LevelCollect {
    const int64_t* coff; const unsigned* pcap;      // candidates = coff[P - 1] + pcap[P - 1]
    const int64_t* poff; const unsigned* pcnt;      // pairs
    int P;
    const int* cnt;                                 // the level's counter block
    const long long* lvl;                           // k_level_tail's counts (NULL: a synchronous level, the host has them)
    const GridParams* gp; const unsigned* bbox;     // the NEXT level's prologue: grid, parents, irregular components
};
enum { LC_CAND = 0, LC_PAIRS, LC_ORPHANS, LC_NPRE, LC_FLAGS, LC_HEAVY, LC_ITEMS, LC_MAXPAIRS, LC_DROPPED, LC_NEXT_P, LC_NEXT_IRR, LC_NEXT_GP, LC_WORDS = 16 };
__global__ void k_level_collect(LevelCollect q, unsigned long long* __restrict__ dst, unsigned long long seq) {
    const int t = threadIdx.x;
    unsigned long long v = 0ull;
    const bool sel = q.P > 0 && q.coff != nullptr;
    if (t == LC_CAND) v = sel ? (unsigned long long)q.coff[q.P - 1] + q.pcap[q.P - 1] : 0ull;
    else if (t == LC_PAIRS) v = sel ? (unsigned long long)q.poff[q.P - 1] + q.pcnt[q.P - 1] : 0ull;
    else if (t == LC_ORPHANS) v = q.lvl ? (unsigned long long)q.lvl[1] : 0ull;
    else if (t == LC_NPRE) v = q.lvl ? (unsigned long long)q.lvl[0] : 0ull;
    else if (t == LC_FLAGS) v = (q.cnt[2] ? 1ull : 0ull) | (q.cnt[12] ? 2ull : 0ull) | (q.cnt[13] ? 4ull : 0ull);
    else if (t == LC_HEAVY) v = (unsigned)q.cnt[8];
    else if (t == LC_ITEMS) v = (unsigned)q.cnt[10];
    else if (t == LC_MAXPAIRS) v = (unsigned)q.cnt[15];
    else if (t == LC_DROPPED) v = (unsigned)q.cnt[3];
    else if (t == LC_NEXT_P) v = q.bbox[6];
    else if (t == LC_NEXT_IRR) v = q.bbox[7];
    else if (t < LC_WORDS) v = reinterpret_cast<const unsigned long long*>(q.gp)[t - LC_NEXT_GP];
    if (t < LC_WORDS) __hip_atomic_store(dst + 16 + t, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    __syncthreads();
    if (t == 0) __hip_atomic_store(dst + 15, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ __launch_bounds__(256) void k_compact_rows(int64_t n, int width, const int* __restrict__ keep,
                                                      const int* __restrict__ pos, const float* __restrict__ src,
                                                      float* __restrict__ dst) {
    const int64_t total = n * width;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = t / width;
        if (keep[i]) dst[(int64_t)pos[i] * width + (t - i * width)] = src[t];
    }
}
// the same in pieces of four floats (rows of `width` floats, 4-byte aligned: unaligned 16-byte loads and stores): the SH rows of a
// level that drops a few components were copied float by float (0.42 ms of a 3 M-row level)
__global__ __launch_bounds__(256) void k_compact_rows4(int64_t n, int width, const int* __restrict__ keep,
                                                       const int* __restrict__ pos, const float* __restrict__ src,
                                                       float* __restrict__ dst) {
    const int Q = (width + 3) >> 2;
    const int64_t total = n * Q;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = total < ((int64_t)1 << 31) ? (int64_t)((unsigned)t / (unsigned)Q) : t / Q;
        if (!keep[i]) continue;
        const int q = (int)(t - i * Q);
        const float* s4 = src + i * width + 4 * q;
        float* d4 = dst + (int64_t)pos[i] * width + 4 * q;
        const int left = width - 4 * q;
        if (left > 3) *reinterpret_cast<f4u*>(d4) = *reinterpret_cast<const f4u*>(s4);
        else for (int k = 0; k < left; ++k) d4[k] = s4[k];
    }
}
__global__ __launch_bounds__(256) void k_compact_bytes(int64_t n, const int* __restrict__ keep, const int* __restrict__ pos,
                                                       const uint8_t* __restrict__ src, uint8_t* __restrict__ dst) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        if (keep[i]) dst[pos[i]] = src[i];
}

// test hooks: the exact device functions the selection kernel uses
__global__ void k_debug_logf(int64_t n, const float* __restrict__ x, float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = glibc_logf(x[i]);
}
__global__ void k_debug_kld(int64_t n, const float* __restrict__ cm, const float* __restrict__ cc, const float* __restrict__ pm,
                            const float* __restrict__ pc, float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const f3 a = {cm[3 * i], cm[3 * i + 1], cm[3 * i + 2]}, b = {pm[3 * i], pm[3 * i + 1], pm[3 * i + 2]};
        const s6 ca = {cc[6 * i], cc[6 * i + 1], cc[6 * i + 2], cc[6 * i + 3], cc[6 * i + 4], cc[6 * i + 5]};
        const s6 cb = {pc[6 * i], pc[6 * i + 1], pc[6 * i + 2], pc[6 * i + 3], pc[6 * i + 4], pc[6 * i + 5]};
        const float dp = det6(cb);
        out[i] = kld6(sub3(a, b), ca, det6(ca), inverse6(cb, dp), dp);
    }
}

// test hook: the regular predicate, the per-parent filter record and the stage-1 decision, by the device functions k_select uses
__global__ void k_debug_stage1(int64_t n, const float* __restrict__ pm, const float* __restrict__ pc, const float* __restrict__ cm,
                               const float* __restrict__ cc, float thr, uint8_t* __restrict__ preg, uint8_t* __restrict__ creg,
                               uint8_t* __restrict__ white, uint8_t* __restrict__ reject, float* __restrict__ T1, uint8_t* __restrict__ clip) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const s6 P = {pc[6 * i], pc[6 * i + 1], pc[6 * i + 2], pc[6 * i + 3], pc[6 * i + 4], pc[6 * i + 5]};
        const s6 Cc = {cc[6 * i], cc[6 * i + 1], cc[6 * i + 2], cc[6 * i + 3], cc[6 * i + 4], cc[6 * i + 5]};
        const float dp = det6(P), dc = det6(Cc);
        const bool pr_ok = is_regular(P, dp, pm[3 * i], pm[3 * i + 1], pm[3 * i + 2]);
        const bool cr_ok = is_regular(Cc, dc, cm[3 * i], cm[3 * i + 1], cm[3 * i + 2]);
        ParentRec pr;
        pr.pinv = inverse6(P, dp);
        make_filter(pr.pinv, dp, thr, 1, pr_ok, pr);
        const float vc[11] = {pm[3 * i], pm[3 * i + 1], pm[3 * i + 2], pr.u00, pr.u01, pr.u02, pr.u11, pr.u12, pr.u22, pr.T1, 0.0f};
        const bool rej = pr.white != 0.0f && cr_ok && white_smd(vc, cm[3 * i], cm[3 * i + 1], cm[3 * i + 2]) > pr.T1;
        preg[i] = pr_ok ? 1 : 0; creg[i] = cr_ok ? 1 : 0; white[i] = pr.white != 0.0f ? 1 : 0; reject[i] = rej ? 1 : 0;
        T1[i] = pr.T1; clip[i] = pr.ec.on != 0.0f ? 1 : 0;
    }
}

__global__ void k_debug_kl_gate(int64_t n, const float* __restrict__ s2, const float* __restrict__ det_c, const float* __restrict__ det_p, float thr,
                                uint8_t* __restrict__ reject, float* __restrict__ lf, uint8_t* __restrict__ need_exact) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float l;
        bool ex;
        reject[i] = kl_gate_rejects(s2[i], det_c[i], det_p[i], 1.0f / det_p[i], thr, k_logf_tab, &l, &ex) ? 1 : 0;
        lf[i] = l;
        need_exact[i] = ex ? 1 : 0;
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
struct Level {
    int64_t n = 0;
    int F = 0;
    DevBuf xyz, color, cov6, opacity, weight, sh, is_parent;
    int32_t reserve(int64_t m, int f) {
        const size_t mm = (size_t)(m > 0 ? m : 1);
        GSR_TRY(xyz.reserve(mm * 3 * 4)); GSR_TRY(color.reserve(mm * 3 * 4)); GSR_TRY(cov6.reserve(mm * 6 * 4));
        GSR_TRY(opacity.reserve(mm * 4)); GSR_TRY(weight.reserve(mm * 4));
        GSR_TRY(sh.reserve(mm * (size_t)(f > 0 ? f : 1) * 4)); GSR_TRY(is_parent.reserve(mm));
        return GSR_OK;
    }
    void release() { xyz.release(); color.release(); cov6.release(); opacity.release(); weight.release(); sh.release(); is_parent.release(); }
    void swap(Level& o) {
        std::swap(n, o.n); std::swap(F, o.F);
        xyz.swap(o.xyz); color.swap(o.color); cov6.swap(o.cov6); opacity.swap(o.opacity);
        weight.swap(o.weight); sh.swap(o.sh); is_parent.swap(o.is_parent);
    }
};

}  // namespace gsr

using namespace gsr;

struct gsr_hem_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t aux = nullptr;      // second stream: the queue of heavy work items runs beside the light parents
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_pre = nullptr;    // the parents' output ranks (flags in input order + their scan), computed beside the grid phase
    hipStream_t aux2 = nullptr;     // third stream: the SH rows are gathered into cell order (an HBM stream only the M-step needs)
    hipEvent_t ev_sh_fork = nullptr, ev_sh_join = nullptr;      // beside the selection (VALU / latency bound)
    hipEvent_t ev_halo = nullptr;   // partitioned level: the ghosts' SH rows have arrived (their exchange runs on aux2 beside the grid phase and the selection)
    hipEvent_t evp[4] = {nullptr, nullptr, nullptr, nullptr};   // brackets of the two halo exchanges (records on the main stream, SH rows on aux2)
    float part_ms[4] = {0, 0, 0, 0};                            // their durations (gsr_hem_get_part_ms)
    DevBuf sh_send;
    float rho = 3.0f, delta = 3.0f, kappa = 2.5f, tau = 1.0f;
    int rng_mode = GSR_RNG_GLIBC;
    uint32_t rng_seed = 1;
    uint64_t rng_pos = 0;           // hem::rand() values consumed so far
    unsigned rng_base[31];          // y_{-31..-1} of the seeded glibc stream
    bool rng_ready = false;
    Level cur, nxt, tmp;
    // level 0 borrowed from the caller (gsr_hem_set_level0, on_device = 2): `spare` keeps cur's own five big buffers meanwhile
    bool cur_borrowed = false;
    DevBuf spare[5];
    // gsr_hem_set_output: the next level is written straight into caller-owned arrays, which then ARE the current level (borrowed)
    void* out_ptr[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    int64_t out_rows = 0;
    bool out_pending = false;
    DevBuf spare_out[5];            // nxt's own big arrays, parked while the caller's stand in for them
    bool have_level = false;
    // workspace
    DevBuf hist, iflag, irank, ipos, rng_blocks, bhist, bstart, bcursor;
    bool split_heavy = true;        // heavy parents are cut into work items of SEL_PART candidates (GSR_HEM_SPLIT=0: one wave per parent)
    bool sum_bucket = true;         // per-child sums by bucket partition + LDS fixed point (GSR_HEM_SUMLW=sort for the radix sort)
    // spatially partitioned levels (gsr_hem_set_comm + gsr_hem_set_level0_part): this rank owns cur.n components of a level of
    // n_global; gid = their global indices (ascending)
    gsr_comm* comm = nullptr;
    int64_t n_global = 0;
    DevBuf gid, gid_next, rec_loc, gid_loc, ghost_sh, ghost_src, perm, pown, ppos_own, inv, gmax, gacc, cmask, dflag, dpos, sent_idx, rows_send, rows_recv,
        xsend, xrecv, gbits, wcnt, wpre, grank, allflags, pcounts, pmatrix;
    int64_t part_stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // ghosts, rows sent, bytes received in the halo exchange, bytes of the other exchanges, ...
    int partition_stage = 0;        // GSR_HEM_PARTITION_STAGE=6144 / 4096: force a smaller stage than the bucket count asks for (tests)
    bool partition_staged = true;   // k_partition<STAGE > 0>: pairs staged by bucket in LDS, read once (GSR_HEM_PARTITION=walk: the two-walk form)
    bool partition_fixed = true;    // bucket regions of fixed capacity filled straight from the segments (GSR_HEM_PARTITION=exact: histogram + scan)
    bool partition_overflowed = false;      // a region overflowed once: this context uses the exact partition from then on
    double partition_factor = 0.0;  // GSR_HEM_PARTITION_FACTOR: region capacity in multiples of the mean (test knob: < 1 forces the overflow path)
    unsigned long long* host_rb = nullptr;      // pinned host memory the device writes read-backs into (16 words; [15] = sequence number)
    unsigned long long rb_seq = 0;
    bool rb_poll = true;            // the host polls the sequence word (GSR_HEM_RB_POLL=0: hipStreamSynchronize)
    DevBuf rec, bbox, bbox_part, gparams, keys, idx, skeys, order, cellStart, A, geo, shs, Rs, pflag, ppos, plist;
    DevBuf pcap, coff, sp_child, sp_wl, porder, pkeys, pkeys2, pidx, mhdr, prec, rowlist;
    int sh_policy = 2;              // the cell-sorted, padded copy of the SH block (k_gather_sh) the M-step reads its children's rows from: 2 (default) = decided
                                    // per level ON THE DEVICE -- made iff the level has at least sh_direct_pairs accepted pairs per component, else the rows are
                                    // read from the level's own array; GSR_HEM_SH_DIRECT=0: always made (rounds 1-4), =1: never; GSR_HEM_SH_DIRECT_PAIRS=x: the
                                    // threshold.  Measured at 5 M (profiles/r05e_ab_sh_direct.txt), level time copy / no copy: isotropic (22 pairs per component)
                                    // 8.54 / 8.50 ms from own buffers but 8.63 / 8.78 in the bench's zero-copy cascade; clustered (7) 6.98 / 6.70; surfels (2) 6.55 / 6.25
    float sh_direct_pairs = 8.0f;
    bool use_rowlist = true;        // GSR_HEM_ROWLIST=0: k_select computes every row span itself instead of taking the non-empty ones from k_spans
    size_t rowlist_max = (size_t)4096 << 20;   // GSR_HEM_ROWLIST_MAX_MB: the row lists are 256 bytes per parent (0.43 GB at the 1.67 M parents of a 5 M level,
                                    // 3.4 GB at 40 M splats), kept by the context; a level whose lists would be larger runs without them
    int timing = 1;                 // gsr_hem_set_timing / GSR_HEM_TIMING: 0 no events, 1 level + k_select + k_mstep, 2 every phase (see GSR_TIME)
    int select_np = 0;              // GSR_HEM_SELECT_NP=1|2|4: light parents per selection wave (the rings are kept across them, see SEL_NP); 0 = by level size
    bool mstep_split = true;        // GSR_HEM_MSTEP_SPLIT=0: a parent of more than MSTEP_SEG pairs keeps ONE wave for all its segments (the schedule of rounds 1-3)
    DevBuf mh_list, mh_items, mh_scratch;
    hipEvent_t ev_mfork = nullptr, ev_mjoin = nullptr;          // the heavy parents' segments run on the second stream beside k_mstep
    bool mstep_small = true;        // GSR_HEM_MSTEP_SMALL=0: no four-at-a-time path for the parents of <= 16 pairs (test knob: nothing may change)
    bool use_ell = true;            // GSR_HEM_ELL=0: no ellipsoid row clipping (test knob: the pair set must not change)
    int shard_rank = 0, shard_world = 1;      // work-sharded level: parents split over ranks, data replicated
    gsr_allreduce_dev_fn shard_allreduce = nullptr;
    gsr_allgather_dev_fn shard_allgather = nullptr;
    void* shard_user = nullptr;
    DevBuf shard_send, shard_recv;
    bool sparse_path = false;
    DevBuf hitem, hfirst, part_cnt, Ac, cellStartC, cellStartI;
    DevBuf pcnt, poff, pair_child, pair_wl, spair_child, spair_wl, cstart, sumLw, oflag, pflag_in, oflag_in, prank_in, orank_in;
    DevBuf keep, kpos, scratch, draws, counters, rocprim_tmp, rocprim_tmp2, sh_tail, holes, erase_halo;
    int64_t stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int64_t stats_ex[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float phase_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    hipEvent_t evk[4] = {nullptr, nullptr, nullptr, nullptr};   // brackets of k_select<COUNT> and k_select<FILL>
    hipEvent_t evm[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // brackets of k_mstep, k_partition, k_bucket_sum
    float kernel_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // The PROLOGUE of a level -- packed records, bounding box, axis histograms, grid geometry, the counts of parents and irregular
    // components: everything the host needs before it can size and launch the level -- is computed when the level's INPUT comes into
    // being (gsr_hem_set_level0, the end of the level before), and its answer rides in that moment's round trip.  pro.valid: `rec`,
    // `gparams` and the counter block pro.cblock hold it for the current level of pro.n components.
    struct Prologue { bool valid = false; GridParams gp; int P = 0, n_irr = 0, cblock = 0; int64_t n = 0; float ms = 0.0f; } pro;
    hipEvent_t ev_pro[2] = {nullptr, nullptr};      // brackets of a prologue: its time counts for the level that consumes it (pro.ms)
    int cblock = 0;                 // which of the two counter blocks the running level uses (the next level's prologue clears the other)
    DevBuf lvl;                     // long long[8]: device-resident counts of an asynchronous level (k_level_tail); [2] = work-item size (k_heavy_items)
    bool async_ok = true;           // GSR_HEM_ASYNC=0: every level sizes its buffers from counts read back on the way (five round trips, the rounds 1-4 schedule)
    bool last_level = false;        // gsr_hem_run_levels: the level being run is the last of its hierarchy -- no prologue of a next level behind it
    hipStream_t side = nullptr;     // gsr_hem_run_levels: the levels' normals beside the next level
    hipEvent_t ev_side = nullptr, ev_side_fork = nullptr;
    int round_trips = 0;            // host round trips of the last gsr_hem_run_level (statistic: gsr_hem_get_stats_ex [6])
    int was_async = 0;              // the last level ran without a round trip between its first and its last kernel ([7])
    float cell_target = 16.0f;      // components per grid cell (GSR_HEM_CELL_TARGET; the result does not depend on it).  Swept at 5 M after the parents left
                                    // the candidate stream: 5 -> 13.15 ms per level, 8 -> 12.87, 12 -> 12.70, 16 -> 12.61, 24 -> 12.60, 32 -> 12.68 (fewer, longer rows)
    int max_cells = 1 << 24;
};

namespace {

// x^(2^b) mod (x^31 - x^28 - 1) over Z/2^32, b = 0..47 (host, once)
void rng_polymul(const unsigned* a, const unsigned* b, unsigned* out) {
    unsigned prod[61];
    for (int d = 0; d < 61; ++d) prod[d] = 0;
    for (int i = 0; i < 31; ++i) for (int j = 0; j < 31; ++j) prod[i + j] += a[i] * b[j];
    for (int d = 60; d >= 31; --d) { const unsigned c = prod[d]; prod[d - 3] += c; prod[d - 31] += c; }
    for (int d = 0; d < 31; ++d) out[d] = prod[d];
}
const unsigned* rng_xpow_table() {
    struct Table {                       // function-local static: initialised once, thread-safely (C++11)
        unsigned tab[48 * 31];
        Table() {
            for (int j = 0; j < 31; ++j) tab[j] = j == 1 ? 1u : 0u;             // x^1
            for (int b = 1; b < 48; ++b) rng_polymul(tab + (b - 1) * 31, tab + (b - 1) * 31, tab + b * 31);
        }
    };
    static const Table t;
    return t.tab;
}

// draw n parent flags in order into dst (consumes n hem::rand() values)
// (n_dev: n is only a bound -- the launch is sized for it -- and the real count lies on the device; the caller then sets the stream
// position itself once it knows the count)
int32_t draw_flags_raw(gsr_hem_ctx* c, int64_t n, uint8_t* dst, hipStream_t on = nullptr, const long long* n_dev = nullptr) {
    const hipStream_t fst = on ? on : c->stream;
    const float prob = 1.0f / c->rho;
    if (n == 0) return GSR_OK;
    if (c->rng_mode == GSR_RNG_HASH) {
        hipLaunchKernelGGL(k_flags_hash, dim3(stride_grid(n)), dim3(256), 0, fst, n, c->rng_seed,
                           (unsigned long long)c->rng_pos, prob, dst, n_dev);
        c->rng_pos += (uint64_t)n;
        return GSR_OK;
    }
    if (!c->rng_ready) {                      // seed words and the jump table, once per context / reseed
        unsigned st[31];
        {   // srand(seed) fills 31 words (Schrage LCG); keep them BEFORE the 310 discarded outputs
            uint32_t s = c->rng_seed ? c->rng_seed : 1u;
            int32_t word = (int32_t)s;
            st[0] = (unsigned)word;
            for (int i = 1; i < 31; ++i) {
                int32_t hi = word / 127773, lo = word % 127773;
                word = 16807 * lo - 2836 * hi;
                if (word < 0) word += 2147483647;
                st[i] = (unsigned)word;
            }
        }
        for (int j = 0; j < 31; ++j) c->rng_base[j] = st[(j + 3) % 31];        // y_{-31+j} = x_{(j+3) mod 31}
        // device tables: x^(2^b), the per-thread jumps x^(8 * RNG_ELEMS * t), and the sequence continued from the seed state
        std::vector<unsigned> tab(RNG_TAB_WORDS, 0u);
        memcpy(tab.data() + RNG_TAB_XPOW, rng_xpow_table(), 48 * 31 * 4);
        {
            unsigned step[31], cur[31], nxt[31];
            for (int j = 0; j < 31; ++j) { step[j] = j == 0 ? 1u : 0u; cur[j] = step[j]; }
            for (int b = 0; b < 48; ++b)
                if (((unsigned long long)(8 * RNG_ELEMS) >> b) & 1ull) { rng_polymul(step, rng_xpow_table() + b * 31, nxt); memcpy(step, nxt, sizeof(step)); }
            for (int t = 0; t < RNG_THREADS; ++t) {
                for (int j = 0; j < 31; ++j) tab[RNG_TAB_T + j * 64 + t] = cur[j];
                rng_polymul(cur, step, nxt);
                memcpy(cur, nxt, sizeof(cur));
            }
        }
        for (int m = 0; m < 91; ++m)
            tab[RNG_TAB_YBASE + m] = m < 31 ? c->rng_base[m] : tab[RNG_TAB_YBASE + m - 3] + tab[RNG_TAB_YBASE + m - 31];
        GSR_TRY(c->draws.reserve(RNG_TAB_WORDS * 4));
        GSR_HIP(hipMemcpy(c->draws.p, tab.data(), RNG_TAB_WORDS * 4, hipMemcpyHostToDevice));
        c->rng_ready = true;
    }
    const int64_t nblocks = (n + RNG_BLOCK_ELEMS - 1) / RNG_BLOCK_ELEMS;
    GSR_TRY(c->rng_blocks.reserve((size_t)nblocks * 64 * 4));
    hipLaunchKernelGGL(k_rng_block_state, dim3((unsigned)nblocks), dim3(64), 0, fst, (unsigned long long)c->rng_pos,
                       c->draws.as<unsigned>(), c->rng_blocks.as<unsigned>(), n_dev);
    hipLaunchKernelGGL(k_flags_glibc, dim3((unsigned)nblocks), dim3(RNG_THREADS), 0, fst, n, prob, c->draws.as<unsigned>(),
                       c->rng_blocks.as<unsigned>(), dst, n_dev);
    c->rng_pos += (uint64_t)n;
    return GSR_OK;
}
int32_t draw_flags(gsr_hem_ctx* c, Level& lv) { return draw_flags_raw(c, lv.n, lv.is_parent.as<uint8_t>()); }

// Small device values the host needs (counts, totals, the grid geometry) are written by one tiny kernel straight
// into pinned, device-visible host memory: one launch + one stream synchronisation per round trip.  (Separate
// hipMemcpyAsync calls into pageable memory cost ~20 us each on top of the synchronisation.)
struct Collect {
    const void* src[8];
    int bytes[8];          // 4 or 8
    int n;
};
// The values first, a system-scope fence, then the sequence number of the round trip into dst[15]: the host does not call
// hipStreamSynchronize (20-75 us until the thread is awake again) but polls that word in its own memory (~5 us).
__global__ void k_collect(Collect q, unsigned long long* __restrict__ dst, unsigned long long seq) {
    const int t = threadIdx.x;
    if (t < q.n) {
        const unsigned long long v = q.bytes[t] == 8 ? *(const unsigned long long*)q.src[t] : (unsigned long long)*(const unsigned*)q.src[t];
        __hip_atomic_store(dst + t, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __threadfence_system();
    __syncthreads();
    if (t == 0) __hip_atomic_store(dst + 15, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// (the wait of a round trip: the kernel that carries sequence number `seq` has been enqueued on the context's stream)
int32_t wait_round_trip(gsr_hem_ctx* c, unsigned long long seq) {
    c->round_trips += 1;
    bool seen = false;
    if (c->rb_poll) {
        (void)hipStreamQuery(c->stream);                        // makes sure the queue is submitted
        (void)hipGetLastError();                                // (hipErrorNotReady is not an error here)
        volatile unsigned long long* flag = c->host_rb + 15;
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spins = 1; !(seen = __atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq); ++spins) {
            if ((spins & 0x3ffu) == 0u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) break;   // a fault upstream: let the
            gsr::cpu_relax(spins);                                                                                        // synchronisation report it
        }
    }
    if (!seen) GSR_HIP(hipStreamSynchronize(c->stream));
    return GSR_OK;
}
int32_t read_back(gsr_hem_ctx* c, const Collect& q, unsigned long long* out) {
    const unsigned long long seq = ++c->rb_seq;
    hipLaunchKernelGGL(k_collect, dim3(1), dim3(8), 0, c->stream, q, c->host_rb, seq);
    GSR_HIP(hipGetLastError());
    GSR_TRY(wait_round_trip(c, seq));
    for (int i = 0; i < q.n; ++i) out[i] = __atomic_load_n(c->host_rb + i, __ATOMIC_RELAXED);
    return GSR_OK;
}
// the level-wide answer (k_level_collect): LC_WORDS words
int32_t read_back_level(gsr_hem_ctx* c, const LevelCollect& q, unsigned long long* out) {
    const unsigned long long seq = ++c->rb_seq;
    hipLaunchKernelGGL(k_level_collect, dim3(1), dim3(64), 0, c->stream, q, c->host_rb, seq);
    GSR_HIP(hipGetLastError());
    GSR_TRY(wait_round_trip(c, seq));
    for (int i = 0; i < LC_WORDS; ++i) out[i] = __atomic_load_n(c->host_rb + 16 + i, __ATOMIC_RELAXED);
    return GSR_OK;
}

// (side = true: on the context's second stream, with its own temporary storage)
// rocPRIM's look-back scan with more items per thread on large inputs (its default for gfx950: 30 us per 5 M ints; 256 threads x 64 items:
// 19 us; int64 sums of 1.67 M counts 19.5 -> 12 us with 32 items: profiles/r05ak_scan_configs.txt) -- on small inputs the default's
// many small workgroups are the better shape
#ifndef GSR_SCAN_BIG_N
#define GSR_SCAN_BIG_N (1 << 20)
#endif
using scan_cfg_big = rocprim::scan_config<256, 64, rocprim::block_load_method::block_load_transpose, rocprim::block_store_method::block_store_transpose,
                                          rocprim::block_scan_algorithm::using_warp_scan>;
using scan_cfg64_big = rocprim::scan_config<256, 32, rocprim::block_load_method::block_load_transpose, rocprim::block_store_method::block_store_transpose,
                                            rocprim::block_scan_algorithm::using_warp_scan>;
template <typename T>
int32_t exclusive_scan(gsr_hem_ctx* c, const T* in, T* out, int64_t n, bool side = false) {
    size_t bytes = 0;
    const hipStream_t s = side ? c->aux : c->stream;
    DevBuf& tmp = side ? c->rocprim_tmp2 : c->rocprim_tmp;
    if (n >= GSR_SCAN_BIG_N) {
        GSR_HIP(rocprim::exclusive_scan<scan_cfg_big>(nullptr, bytes, in, out, (T)0, (size_t)n, rocprim::plus<T>(), s));
        GSR_TRY(tmp.reserve(bytes));
        GSR_HIP(rocprim::exclusive_scan<scan_cfg_big>(tmp.p, bytes, in, out, (T)0, (size_t)n, rocprim::plus<T>(), s));
        return GSR_OK;
    }
    GSR_HIP(rocprim::exclusive_scan(nullptr, bytes, in, out, (T)0, (size_t)n, rocprim::plus<T>(), s));
    GSR_TRY(tmp.reserve(bytes));
    GSR_HIP(rocprim::exclusive_scan(tmp.p, bytes, in, out, (T)0, (size_t)n, rocprim::plus<T>(), s));
    return GSR_OK;
}

// rocPRIM's Onesweep with its gfx950 kernel shapes but 10 bits per pass: the 19-bit cell keys of a 5 M level take two passes
// instead of three (grid phase 1.21 -> 1.15 ms; 11 bits: no further gain).  Up to 2^20 keys rocPRIM's merge sort runs as before.
#ifndef GSR_SORT_RADIX_BITS
#define GSR_SORT_RADIX_BITS 10
#endif
// GSR_SORT_MERGE_LIMIT: up to this many keys rocPRIM sorts by block sort + merge passes (~20 launches at 5 * 10^5 keys), beyond it by Onesweep
#ifndef GSR_SORT_MERGE_LIMIT
#define GSR_SORT_MERGE_LIMIT (256 * 1024)      // measured on the bench levels (profiles/archive/r04i): 1 M -> 256 k takes 0.05 ms off the 1.67 M level (its 556 k parents) and 0.04 off the 556 k level; 128 k, 32 k: the same
#endif
#ifndef GSR_SORT_BS
#define GSR_SORT_BS 1024
#endif
#ifndef GSR_SORT_IPT
#define GSR_SORT_IPT 12        // keys per thread of the Onesweep kernels (1 024 threads): 16 -> 12 takes 44 us off a 5 M level's two sorts (230 -> 186 us;
                               // 10 the same, 8: 198, 4: 243; 512- and 256-thread workgroups 260 ... 600 us), profiles/r05ah_onesweep_configs.txt
#endif
// (the sorts of up to GSR_SORT_MERGE_LIMIT keys: block sort of 1 024 x 8 keys, then merge passes -- 20 us less per 200 k-splat level than rocPRIM's
// default shape, equal at 50 k and 556 k: profiles/r05am_merge_sort_configs.txt)
#ifndef GSR_MERGE_SORT_BS
#define GSR_MERGE_SORT_BS 1024
#define GSR_MERGE_SORT_IPT 8
#define GSR_MERGE_MP_BS 128
#define GSR_MERGE_MP_IPT 4
#endif
using merge_cfg = rocprim::merge_sort_config<512, GSR_MERGE_SORT_BS, GSR_MERGE_SORT_IPT, 128, GSR_MERGE_MP_BS, GSR_MERGE_MP_IPT>;
using sort_cfg = rocprim::radix_sort_config<rocprim::default_config, merge_cfg,
                                            rocprim::radix_sort_onesweep_config<rocprim::kernel_config<GSR_SORT_BS, GSR_SORT_IPT>, rocprim::kernel_config<GSR_SORT_BS, GSR_SORT_IPT>, GSR_SORT_RADIX_BITS,
                                                                                rocprim::block_radix_rank_algorithm::match>,
                                            GSR_SORT_MERGE_LIMIT>;
// the ordering sort (parents by work class + block of space, ORDER_KEY_BITS + 2 bits): when its key fits ONE Onesweep pass the merge sort's
// block sort + log2(n / 8192) merge passes lose from a few ten thousand keys on
#ifndef GSR_ORDER_MERGE_LIMIT
#define GSR_ORDER_MERGE_LIMIT 16384
#endif
using order_cfg = rocprim::radix_sort_config<rocprim::default_config, merge_cfg,
                                             rocprim::radix_sort_onesweep_config<rocprim::kernel_config<GSR_SORT_BS, GSR_SORT_IPT>, rocprim::kernel_config<GSR_SORT_BS, GSR_SORT_IPT>, GSR_SORT_RADIX_BITS,
                                                                                 rocprim::block_radix_rank_algorithm::match>,
                                             GSR_ORDER_MERGE_LIMIT>;
template <typename V, typename CFG = sort_cfg>
int32_t sort_pairs(gsr_hem_ctx* c, const unsigned* kin, unsigned* kout, const V* vin, V* vout, int64_t n, int end_bit) {
    size_t bytes = 0;
    GSR_HIP(rocprim::radix_sort_pairs<CFG>(nullptr, bytes, kin, kout, vin, vout, (size_t)n, 0u, (unsigned)end_bit, c->stream));
    GSR_TRY(c->rocprim_tmp.reserve(bytes));
    GSR_HIP(rocprim::radix_sort_pairs<CFG>(c->rocprim_tmp.p, bytes, kin, kout, vin, vout, (size_t)n, 0u, (unsigned)end_bit, c->stream));
    return GSR_OK;
}

int bits_for(int64_t n) {
    int b = 1;
    while (b < 32 && ((int64_t)1 << b) < n) ++b;
    return b;
}

}  // namespace


namespace {
// cur.{xyz,color,cov6,opacity,sh} <-> the caller's arrays (borrowed) / the context's own buffers (spare)
inline DevBuf* level_big(Level& L, int i) { DevBuf* b[5] = {&L.xyz, &L.color, &L.cov6, &L.opacity, &L.sh}; return b[i]; }
void unborrow_level0(gsr_hem_ctx* c) {
    if (!c->cur_borrowed) return;
    for (int i = 0; i < 5; ++i) {
        DevBuf* b = level_big(c->cur, i);
        b->p = nullptr; b->cap = 0;                 // the caller's memory: never freed here
        b->swap(c->spare[i]);
    }
    c->cur_borrowed = false;
}

// The prologue of the level whose input is L (gsr_hem_ctx::Prologue): k_prep (the packed records, box partials, the counts of parents
// and irregular components), their fold (which also clears the level's counter block and the histograms), the axis histograms, the grid
// geometry.  n_dev != NULL: the level's size lies on the device (the level before has not reported yet) and n is a bound for the grids.
int32_t enqueue_prologue(gsr_hem_ctx* c, Level& L, int64_t n, const long long* n_dev, int cblock) {
    hipStream_t st = c->stream;
    const dim3 blk(256), grd(stride_grid(n));
    GSR_TRY(c->rec.reserve((size_t)n * 64)); GSR_TRY(c->bbox.reserve(64));
    GSR_TRY(c->gparams.reserve(sizeof(GridParams))); GSR_TRY(c->counters.reserve(128));
    GSR_TRY(c->bbox_part.reserve((size_t)grd.x * 8 * 4)); GSR_TRY(c->hist.reserve(3 * HIST_BINS * 4));
    GSR_TRY(c->bcursor.reserve(((size_t)SUM_MAX_BUCKETS + 1) * 8));
    if (c->timing >= 1) GSR_HIP(hipEventRecord(c->ev_pro[0], st));
    hipLaunchKernelGGL(k_prep, grd, blk, 0, st, n, L.xyz.as<float>(), L.color.as<float>(), L.cov6.as<float>(), L.opacity.as<float>(),
                       L.weight.as<float>(), L.is_parent.as<uint8_t>(), c->rec.as<float4>(), c->bbox_part.as<unsigned>(), n_dev);
    hipLaunchKernelGGL(k_bbox_reduce, dim3(1), dim3(256), 0, st, (int)grd.x, c->bbox_part.as<unsigned>(), c->bbox.as<unsigned>(),
                       c->counters.as<unsigned>() + 16 * cblock, 16, c->hist.as<unsigned>(), 3 * HIST_BINS, c->bcursor.as<unsigned>(), SUM_MAX_BUCKETS + 1);
    hipLaunchKernelGGL(k_hist, dim3(stride_grid(n) > 512 ? 512 : stride_grid(n)), blk, 0, st, n, L.xyz.as<float>(), c->bbox.as<unsigned>(), c->hist.as<unsigned>(), n_dev);
    hipLaunchKernelGGL(k_grid_params, dim3(1), dim3(64), 0, st, c->bbox.as<unsigned>(), c->hist.as<unsigned>(), n, c->cell_target,
                       c->max_cells, c->gparams.as<GridParams>(), n_dev);
    if (c->timing >= 1) GSR_HIP(hipEventRecord(c->ev_pro[1], st));
    GSR_HIP(hipGetLastError());
    return GSR_OK;
}
// ... and its answer, out of the words of k_level_collect
void take_prologue(gsr_hem_ctx* c, const unsigned long long* w, int64_t n, int cblock) {
    static_assert(sizeof(GridParams) == 40 && LC_NEXT_GP + 5 == LC_WORDS, "GridParams travels as five 8-byte words");
    memcpy(&c->pro.gp, w + LC_NEXT_GP, sizeof(GridParams));
    c->pro.P = (int)(unsigned)w[LC_NEXT_P]; c->pro.n_irr = (int)(unsigned)w[LC_NEXT_IRR];
    c->pro.n = n; c->pro.cblock = cblock; c->pro.valid = true;
    c->pro.ms = 0.0f;
    if (c->timing >= 1) (void)hipEventElapsedTime(&c->pro.ms, c->ev_pro[0], c->ev_pro[1]);      // (both have completed: the round trip came behind them)
}
// The current level's prologue, now, with its own round trip (gsr_hem_set_level0, gsr_hem_set_state, a level whose prologue is not there).
// Partitioned / work-sharded levels compute theirs inside the level (their box and histograms are all-reduced over the ranks).
int32_t prologue_now(gsr_hem_ctx* c) {
    c->pro.valid = false;
    if (c->cur.n <= 0 || c->comm != nullptr || c->shard_world > 1) return GSR_OK;
    const int cb = c->cblock ^ 1;
    GSR_TRY(enqueue_prologue(c, c->cur, c->cur.n, nullptr, cb));
    LevelCollect q;
    memset(&q, 0, sizeof(q));
    q.cnt = c->counters.as<int>() + 16 * cb; q.gp = c->gparams.as<GridParams>(); q.bbox = c->bbox.as<unsigned>();
    unsigned long long w[LC_WORDS];
    GSR_TRY(read_back_level(c, q, w));
    take_prologue(c, w, c->cur.n, cb);
    return GSR_OK;
}
}  // namespace

extern "C" {

const char* gsr_last_error(void) { return last_error().c_str(); }
const char* gsr_version(void) { return "gsr_hip 0.1 gfx950"; }
int32_t gsr_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int32_t gsr_hem_create(gsr_hem_ctx** out, int32_t device, void* stream) {
    if (!out) return fail(GSR_E_INVALID, "gsr_hem_create: out is NULL");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(GSR_E_NO_DEVICE, "gsr_hem_create: no HIP device visible (this backend has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(GSR_E_INVALID, "gsr_hem_create: device %d out of range (0..%d)", device, ndev - 1);
    GSR_HIP(hipSetDevice(device));
    gsr_hem_ctx* c = new gsr_hem_ctx();
    c->device = device;
    c->stream = (hipStream_t)stream;
    for (int i = 0; i < 8; ++i) {
        hipError_t e = hipEventCreate(&c->ev[i]);
        if (e != hipSuccess) { delete c; return fail(GSR_E_HIP, "hipEventCreate: %s", hipGetErrorString(e)); }
    }
    for (int i = 0; i < 4; ++i) {
        hipError_t e = hipEventCreate(&c->evk[i]);
        if (e != hipSuccess) { delete c; return fail(GSR_E_HIP, "hipEventCreate: %s", hipGetErrorString(e)); }
    }
    for (int i = 0; i < 6; ++i) {
        hipError_t e = hipEventCreate(&c->evm[i]);
        if (e != hipSuccess) { delete c; return fail(GSR_E_HIP, "hipEventCreate: %s", hipGetErrorString(e)); }
    }
    {
        hipError_t e = hipStreamCreateWithFlags(&c->aux, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_pre, hipEventDisableTiming);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->aux2, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_sh_fork, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_sh_join, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_halo, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_mfork, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_mjoin, hipEventDisableTiming);
        for (int i = 0; i < 4 && e == hipSuccess; ++i) e = hipEventCreate(&c->evp[i]);
        for (int i = 0; i < 2 && e == hipSuccess; ++i) e = hipEventCreate(&c->ev_pro[i]);
        if (e != hipSuccess) { delete c; return fail(GSR_E_HIP, "second stream: %s", hipGetErrorString(e)); }
    }
    {
        // pinned, device-mapped, COHERENT: the read-back kernel's system-scope stores must reach the host while the stream is running
        hipError_t e = hipHostMalloc((void**)&c->host_rb, 512, hipHostMallocMapped | hipHostMallocCoherent);      // [0..14] k_collect, [15] sequence, [16..31] k_level_collect
        if (e != hipSuccess) { c->host_rb = nullptr; delete c; return fail(GSR_E_HIP, "hipHostMalloc: %s", hipGetErrorString(e)); }
        memset(c->host_rb, 0, 512);
    }
    // Environment knobs (all of them): none changes a result, each is exercised by a test.
    if (const char* s = getenv("GSR_HEM_ELL")) c->use_ell = atoi(s) != 0;
    if (const char* s = getenv("GSR_HEM_SUMLW")) c->sum_bucket = strcmp(s, "sort") != 0;
    if (const char* s = getenv("GSR_HEM_SPLIT")) c->split_heavy = atoi(s) != 0;
    if (const char* s = getenv("GSR_HEM_MSTEP_SMALL")) c->mstep_small = atoi(s) != 0;
    if (const char* s = getenv("GSR_HEM_MSTEP_SPLIT")) c->mstep_split = atoi(s) != 0;
    if (const char* s = getenv("GSR_HEM_SH_DIRECT")) c->sh_policy = atoi(s) != 0 ? 1 : 0;
    if (const char* s = getenv("GSR_HEM_SH_DIRECT_PAIRS")) c->sh_direct_pairs = (float)atof(s);
    if (const char* s = getenv("GSR_HEM_ROWLIST")) c->use_rowlist = atoi(s) != 0;
    if (const char* s = getenv("GSR_HEM_ROWLIST_MAX_MB")) c->rowlist_max = (size_t)(atof(s) * 1048576.0);
    if (const char* s = getenv("GSR_HEM_TIMING")) { const int v = atoi(s); c->timing = v < 0 ? 0 : (v > 2 ? 2 : v); }
    if (const char* s = getenv("GSR_HEM_SELECT_NP")) { const int v = atoi(s); c->select_np = v < 1 ? 1 : (v > SEL_NP ? SEL_NP : v); }
    if (const char* s = getenv("GSR_HEM_RB_POLL")) c->rb_poll = atoi(s) != 0;
    if (const char* s = getenv("GSR_HEM_ASYNC")) c->async_ok = atoi(s) != 0;
    if (const char* s = getenv("GSR_HEM_PARTITION_STAGE")) { const int v = atoi(s); if (v == 8192 || v == 6144 || v == 4096) c->partition_stage = v; }
    if (const char* s = getenv("GSR_HEM_PARTITION")) { c->partition_fixed = strcmp(s, "exact") != 0; c->partition_staged = strcmp(s, "walk") != 0; }
    if (const char* s = getenv("GSR_HEM_PARTITION_FACTOR")) c->partition_factor = atof(s);
    if (const char* s = getenv("GSR_HEM_CELL_TARGET")) { float v = (float)atof(s); if (v > 0.25f && v < 4096.0f) c->cell_target = v; }
    (void)hipFuncSetAttribute((const void*)k_bucket_sum, hipFuncAttributeMaxDynamicSharedMemorySize, 12 << 13);
    (void)hipFuncSetAttribute((const void*)k_bucket_hist, hipFuncAttributeMaxDynamicSharedMemorySize, SUM_MAX_BUCKETS * 4);
    (void)hipFuncSetAttribute((const void*)k_bucket_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, SUM_MAX_BUCKETS * 4);
    (void)hipFuncSetAttribute((const void*)k_partition<0>, hipFuncAttributeMaxDynamicSharedMemorySize, SUM_MAX_BUCKETS * 4);
    (void)hipFuncSetAttribute((const void*)k_partition<8192>, hipFuncAttributeMaxDynamicSharedMemorySize, 1536 * 8 + 8192 * 8);
    (void)hipFuncSetAttribute((const void*)k_partition<6144>, hipFuncAttributeMaxDynamicSharedMemorySize, 3584 * 8 + 6144 * 8);
    (void)hipFuncSetAttribute((const void*)k_partition<4096>, hipFuncAttributeMaxDynamicSharedMemorySize, 5632 * 8 + 4096 * 8);
    (void)hipFuncSetAttribute((const void*)k_part_max, hipFuncAttributeMaxDynamicSharedMemorySize, 4 << 13);
    (void)hipFuncSetAttribute((const void*)k_part_acc, hipFuncAttributeMaxDynamicSharedMemorySize, 12 << 13);
    (void)hipFuncSetAttribute((const void*)k_erase_shift, hipFuncAttributeMaxDynamicSharedMemorySize, 12288 * 4);
    *out = c;
    return GSR_OK;
}

int32_t gsr_hem_destroy(gsr_hem_ctx* c) {
    if (!c) return GSR_OK;
    (void)hipSetDevice(c->device);
    unborrow_level0(c);
    for (DevBuf& b : c->spare) b.release();
    for (DevBuf& b : c->spare_out) b.release();
    c->cur.release(); c->nxt.release(); c->tmp.release();
    DevBuf* part_bufs[] = {&c->gid, &c->gid_next, &c->rec_loc, &c->gid_loc, &c->ghost_sh, &c->ghost_src, &c->perm, &c->pown, &c->ppos_own, &c->inv, &c->gmax, &c->gacc,
                           &c->cmask, &c->dflag, &c->dpos, &c->sent_idx, &c->rows_send, &c->rows_recv, &c->sh_send, &c->xsend, &c->xrecv, &c->gbits, &c->wcnt, &c->wpre,
                           &c->grank, &c->allflags, &c->pcounts, &c->pmatrix};
    for (DevBuf* b : part_bufs) b->release();
    DevBuf* all[] = {&c->rec, &c->bbox, &c->bbox_part, &c->gparams, &c->keys, &c->idx, &c->skeys, &c->order, &c->cellStart, &c->A, &c->geo, &c->shs,
                     &c->Rs, &c->pflag, &c->ppos, &c->plist, &c->hitem, &c->hfirst, &c->part_cnt, &c->Ac, &c->cellStartC, &c->cellStartI, &c->pcnt, &c->poff, &c->pair_child, &c->pair_wl,
                     &c->spair_child, &c->spair_wl, &c->cstart, &c->sumLw, &c->oflag, &c->pflag_in, &c->oflag_in, &c->prank_in,
                     &c->orank_in, &c->hist, &c->iflag, &c->irank, &c->ipos, &c->rng_blocks, &c->bhist, &c->bstart, &c->bcursor,
                     &c->porder, &c->pkeys, &c->pkeys2, &c->pidx, &c->mhdr, &c->prec, &c->rowlist, &c->shard_send, &c->shard_recv, &c->pcap, &c->coff, &c->sp_child, &c->sp_wl,
                     &c->keep, &c->kpos, &c->scratch, &c->draws, &c->counters, &c->rocprim_tmp, &c->rocprim_tmp2, &c->mh_list, &c->mh_items, &c->mh_scratch, &c->lvl, &c->sh_tail, &c->holes, &c->erase_halo};
    for (DevBuf* b : all) b->release();
    if (c->host_rb) (void)hipHostFree(c->host_rb);
    for (int i = 0; i < 8; ++i) if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    for (int i = 0; i < 4; ++i) if (c->evk[i]) (void)hipEventDestroy(c->evk[i]);
    for (int i = 0; i < 6; ++i) if (c->evm[i]) (void)hipEventDestroy(c->evm[i]);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_pre) (void)hipEventDestroy(c->ev_pre);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->ev_sh_fork) (void)hipEventDestroy(c->ev_sh_fork);
    if (c->ev_sh_join) (void)hipEventDestroy(c->ev_sh_join);
    if (c->ev_halo) (void)hipEventDestroy(c->ev_halo);
    if (c->ev_mfork) (void)hipEventDestroy(c->ev_mfork);
    if (c->ev_mjoin) (void)hipEventDestroy(c->ev_mjoin);
    for (int i = 0; i < 4; ++i) if (c->evp[i]) (void)hipEventDestroy(c->evp[i]);
    for (int i = 0; i < 2; ++i) if (c->ev_pro[i]) (void)hipEventDestroy(c->ev_pro[i]);
    if (c->aux) (void)hipStreamDestroy(c->aux);
    if (c->aux2) (void)hipStreamDestroy(c->aux2);
    if (c->side) (void)hipStreamDestroy(c->side);
    if (c->ev_side) (void)hipEventDestroy(c->ev_side);
    if (c->ev_side_fork) (void)hipEventDestroy(c->ev_side_fork);
    delete c;
    return GSR_OK;
}

int32_t gsr_hem_set_params(gsr_hem_ctx* c, float rho, float delta, float kappa, float tau) {
    if (!c) return fail(GSR_E_INVALID, "gsr_hem_set_params: NULL context");
    c->rho = rho; c->delta = delta; c->kappa = kappa; c->tau = tau;
    return GSR_OK;
}

int32_t gsr_hem_set_rng(gsr_hem_ctx* c, int32_t mode, uint32_t seed, uint64_t skip) {
    if (!c) return fail(GSR_E_INVALID, "gsr_hem_set_rng: NULL context");
    if (mode != GSR_RNG_GLIBC && mode != GSR_RNG_HASH) return fail(GSR_E_INVALID, "gsr_hem_set_rng: unknown mode %d", mode);
    c->rng_mode = mode; c->rng_seed = seed; c->rng_pos = skip; c->rng_ready = false;
    return GSR_OK;
}
int32_t gsr_hem_get_rng_position(gsr_hem_ctx* c, uint64_t* draws) {
    if (!c || !draws) return fail(GSR_E_INVALID, "gsr_hem_get_rng_position: NULL argument");
    *draws = c->rng_pos;
    return GSR_OK;
}

int32_t gsr_hem_set_shard(gsr_hem_ctx* c, int32_t rank, int32_t world, gsr_allreduce_dev_fn allreduce, gsr_allgather_dev_fn allgather, void* user) {
    if (!c) return fail(GSR_E_INVALID, "gsr_hem_set_shard: NULL context");
    if (world < 1 || rank < 0 || rank >= world) return fail(GSR_E_INVALID, "gsr_hem_set_shard: rank %d of %d", rank, world);
    if (world > 1 && (!allreduce || !allgather)) return fail(GSR_E_INVALID, "gsr_hem_set_shard: world > 1 needs the all-reduce and the all-gather callback");
    c->shard_rank = rank; c->shard_world = world; c->shard_allreduce = allreduce; c->shard_allgather = allgather; c->shard_user = user;
    return GSR_OK;
}

int32_t gsr_hem_set_level0(gsr_hem_ctx* c, const float* xyz, const float* color, const float* cov6,
                           const float* opacity, const float* sh, int64_t n, int32_t F, int32_t on_device) {
    if (!c) return fail(GSR_E_INVALID, "gsr_hem_set_level0: NULL context");
    if (n < 0 || F < 0) return fail(GSR_E_INVALID, "gsr_hem_set_level0: negative size (n=%lld, F=%d)", (long long)n, F);
    if (F > 256) return fail(GSR_E_INVALID, "gsr_hem_set_level0: F=%d exceeds the supported 256 feature floats", F);
    if (n >= ((int64_t)1 << 31) - 1) return fail(GSR_E_INVALID, "gsr_hem_set_level0: n=%lld exceeds 2^31-2", (long long)n);
    if (n > 0 && (!xyz || !color || !cov6 || !opacity || (F > 0 && !sh)))
        return fail(GSR_E_INVALID, "gsr_hem_set_level0: NULL array");
    GSR_HIP(hipSetDevice(c->device));
    unborrow_level0(c);
    Level& L = c->cur;
    if (on_device == 2 && n > 0) {
        // borrow: no copy of the five big arrays (1.19 GB at 5 M splats); they must stay valid and unchanged until the
        // next gsr_hem_run_level returns.  cap = SIZE_MAX makes every later reserve() on them a no-op.
        const void* src[5] = {xyz, color, cov6, opacity, F > 0 ? sh : xyz};
        for (int i = 0; i < 5; ++i) {
            DevBuf* b = level_big(L, i);
            b->swap(c->spare[i]);
            b->p = const_cast<void*>(src[i]); b->cap = (size_t)-1;
        }
        c->cur_borrowed = true;
        const size_t mm = (size_t)n;
        GSR_TRY(L.weight.reserve(mm * 4)); GSR_TRY(L.is_parent.reserve(mm));
        L.n = n; L.F = F;
        hipLaunchKernelGGL(k_fill_const<float>, dim3(stride_grid(n)), dim3(256), 0, c->stream, n, L.weight.as<float>(), 1.0f);
        GSR_TRY(draw_flags(c, L));
        c->have_level = true;
        GSR_TRY(prologue_now(c));                               // (its round trip is this call's synchronisation)
        if (!c->pro.valid) GSR_HIP(hipStreamSynchronize(c->stream));
        return GSR_OK;
    }
    GSR_TRY(L.reserve(n, F));
    L.n = n; L.F = F;
    const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    if (n > 0) {
        GSR_HIP(hipMemcpyAsync(L.xyz.p, xyz, (size_t)n * 12, kind, c->stream));
        GSR_HIP(hipMemcpyAsync(L.color.p, color, (size_t)n * 12, kind, c->stream));
        GSR_HIP(hipMemcpyAsync(L.cov6.p, cov6, (size_t)n * 24, kind, c->stream));
        GSR_HIP(hipMemcpyAsync(L.opacity.p, opacity, (size_t)n * 4, kind, c->stream));
        if (F > 0) GSR_HIP(hipMemcpyAsync(L.sh.p, sh, (size_t)n * F * 4, kind, c->stream));
        hipLaunchKernelGGL(k_fill_const<float>, dim3(stride_grid(n)), dim3(256), 0, c->stream, n, L.weight.as<float>(), 1.0f);   // mixture.cpp:315
    }
    GSR_TRY(draw_flags(c, L));                                  // mixture.cpp:330
    c->have_level = true;
    GSR_TRY(prologue_now(c));                                   // (its round trip is this call's synchronisation: the caller's arrays are free again)
    if (!c->pro.valid) GSR_HIP(hipStreamSynchronize(c->stream));
    return GSR_OK;
}

int32_t gsr_hem_set_comm(gsr_hem_ctx* c, gsr_comm* comm) {
    if (!c) return fail(GSR_E_INVALID, "gsr_hem_set_comm: NULL context");
    c->comm = comm;
    return GSR_OK;
}

int32_t gsr_hem_set_level0_part(gsr_hem_ctx* c, const float* xyz, const float* color, const float* cov6, const float* opacity, const float* sh,
                                const uint32_t* gid, int64_t n_own, int64_t n_global, int32_t F, int32_t on_device) {
    if (!c) return fail(GSR_E_INVALID, "gsr_hem_set_level0_part: NULL context");
    if (!c->comm) return fail(GSR_E_INVALID, "gsr_hem_set_level0_part: no communicator (gsr_hem_set_comm first)");
    if (n_own <= 0 || n_global < n_own || !gid) return fail(GSR_E_INVALID, "gsr_hem_set_level0_part: every rank must own at least one component (n_own=%lld of %lld)", (long long)n_own, (long long)n_global);
    if (n_global >= ((int64_t)1 << 31) - 1) return fail(GSR_E_INVALID, "gsr_hem_set_level0_part: n_global=%lld exceeds 2^31-2", (long long)n_global);
    // the owned arrays as an ordinary (copied) level 0; its flags are replaced below by the flags at the global indices
    const uint64_t pos0 = c->rng_pos;
    GSR_TRY(gsr_hem_set_level0(c, xyz, color, cov6, opacity, sh, n_own, F, on_device ? 1 : 0));
    c->rng_pos = pos0;
    hipStream_t st = c->stream;
    GSR_TRY(c->gid.reserve((size_t)n_own * 4));
    GSR_HIP(hipMemcpyAsync(c->gid.p, gid, (size_t)n_own * 4, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
    {   // strictly ascending and < n_global, checked on the device BEFORE anything is indexed with them
        GSR_TRY(c->scratch.reserve(64));
        GSR_HIP(hipMemsetAsync(c->scratch.p, 0, 4, st));
        hipLaunchKernelGGL(k_check_gids, dim3(stride_grid(n_own)), dim3(256), 0, st, n_own, n_global, c->gid.as<unsigned>(), c->scratch.as<unsigned>());
        unsigned bad = 0;
        GSR_HIP(hipMemcpyAsync(&bad, c->scratch.p, 4, hipMemcpyDeviceToHost, st));
        GSR_HIP(hipStreamSynchronize(st));
        if (bad) {
            c->have_level = false;
            return fail(GSR_E_INVALID, "gsr_hem_set_level0_part: %u of %lld global indices are not strictly ascending or not below n_global=%lld", bad,
                        (long long)n_own, (long long)n_global);
        }
    }
    GSR_TRY(c->allflags.reserve((size_t)n_global));
    GSR_TRY(draw_flags_raw(c, n_global, c->allflags.as<uint8_t>()));          // Mixture::initMixture draws one flag per component of the WHOLE cloud
    hipLaunchKernelGGL(k_gather_bytes, dim3(stride_grid(n_own)), dim3(256), 0, st, n_own, c->gid.as<unsigned>(), c->allflags.as<uint8_t>(), c->cur.is_parent.as<uint8_t>());
    GSR_HIP(hipStreamSynchronize(st));
    c->n_global = n_global;
    c->pro.valid = false;
    return GSR_OK;
}

int32_t gsr_hem_set_output(gsr_hem_ctx* c, float* xyz, float* color, float* cov6, float* opacity, float* sh, int64_t capacity_rows) {
    if (!c) return fail(GSR_E_INVALID, "gsr_hem_set_output: NULL context");
    if (capacity_rows <= 0 || !xyz) { c->out_pending = false; return GSR_OK; }          // (clears a pending request)
    if (!color || !cov6 || !opacity) return fail(GSR_E_INVALID, "gsr_hem_set_output: NULL array");
    c->out_ptr[0] = xyz; c->out_ptr[1] = color; c->out_ptr[2] = cov6; c->out_ptr[3] = opacity; c->out_ptr[4] = sh;
    c->out_rows = capacity_rows;
    c->out_pending = true;
    return GSR_OK;
}

int32_t gsr_hem_get_gids(gsr_hem_ctx* c, uint32_t* gid, int32_t on_device) {
    if (!c || !c->have_level || !gid) return fail(GSR_E_INVALID, "gsr_hem_get_gids: no level set");
    if (!c->comm) return fail(GSR_E_INVALID, "gsr_hem_get_gids: not a partitioned level");
    GSR_HIP(hipSetDevice(c->device));
    if (c->cur.n > 0) {
        GSR_HIP(hipMemcpyAsync(gid, c->gid.p, (size_t)c->cur.n * 4, on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->stream));
        GSR_HIP(hipStreamSynchronize(c->stream));
    }
    return GSR_OK;
}

int32_t gsr_hem_get_part_stats(gsr_hem_ctx* c, int64_t* out8) {
    if (!c || !out8) return fail(GSR_E_INVALID, "gsr_hem_get_part_stats: NULL argument");
    memcpy(out8, c->part_stats, sizeof(c->part_stats));
    out8[7] = c->n_global;
    return GSR_OK;
}

int32_t gsr_hem_set_state(gsr_hem_ctx* c, const uint8_t* parent_mask, const float* weight) {
    if (!c || !c->have_level) return fail(GSR_E_INVALID, "gsr_hem_set_state: no level set");
    GSR_HIP(hipSetDevice(c->device));
    if (c->cur.n > 0) {
        if (parent_mask) GSR_HIP(hipMemcpyAsync(c->cur.is_parent.p, parent_mask, (size_t)c->cur.n, hipMemcpyHostToDevice, c->stream));
        if (weight) GSR_HIP(hipMemcpyAsync(c->cur.weight.p, weight, (size_t)c->cur.n * 4, hipMemcpyHostToDevice, c->stream));
        GSR_TRY(prologue_now(c));                               // the records carry the flags and the weights: again
        GSR_HIP(hipStreamSynchronize(c->stream));
    }
    return GSR_OK;
}

int32_t gsr_hem_level_size(gsr_hem_ctx* c, int64_t* n, int32_t* F) {
    if (!c || !c->have_level) return fail(GSR_E_INVALID, "gsr_hem_level_size: no level set");
    if (n) *n = c->cur.n;
    if (F) *F = c->cur.F;
    return GSR_OK;
}

int32_t gsr_hem_get_level(gsr_hem_ctx* c, float* xyz, float* color, float* cov6, float* opacity, float* sh,
                          float* weight, uint8_t* is_parent, int32_t on_device) {
    if (!c || !c->have_level) return fail(GSR_E_INVALID, "gsr_hem_get_level: no level set");
    GSR_HIP(hipSetDevice(c->device));
    const Level& L = c->cur;
    const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    const size_t n = (size_t)L.n;
    if (n > 0) {
        if (xyz) GSR_HIP(hipMemcpyAsync(xyz, L.xyz.p, n * 12, kind, c->stream));
        if (color) GSR_HIP(hipMemcpyAsync(color, L.color.p, n * 12, kind, c->stream));
        if (cov6) GSR_HIP(hipMemcpyAsync(cov6, L.cov6.p, n * 24, kind, c->stream));
        if (opacity) GSR_HIP(hipMemcpyAsync(opacity, L.opacity.p, n * 4, kind, c->stream));
        if (sh && L.F > 0) GSR_HIP(hipMemcpyAsync(sh, L.sh.p, n * (size_t)L.F * 4, kind, c->stream));
        if (weight) GSR_HIP(hipMemcpyAsync(weight, L.weight.p, n * 4, kind, c->stream));
        if (is_parent) GSR_HIP(hipMemcpyAsync(is_parent, L.is_parent.p, n, kind, c->stream));
        GSR_HIP(hipStreamSynchronize(c->stream));
    }
    return GSR_OK;
}

int32_t gsr_hem_get_part_ms(gsr_hem_ctx* c, float* out4) {
    if (!c || !out4) return fail(GSR_E_INVALID, "gsr_hem_get_part_ms: NULL argument");
    memcpy(out4, c->part_ms, sizeof(c->part_ms));
    return GSR_OK;
}

int32_t gsr_hem_get_stats(gsr_hem_ctx* c, int64_t* out8) {
    if (!c || !out8) return fail(GSR_E_INVALID, "gsr_hem_get_stats: NULL argument");
    memcpy(out8, c->stats, sizeof(c->stats));
    return GSR_OK;
}
int32_t gsr_hem_get_stats_ex(gsr_hem_ctx* c, int64_t* out8) {
    if (!c || !out8) return fail(GSR_E_INVALID, "gsr_hem_get_stats_ex: NULL argument");
    memcpy(out8, c->stats_ex, sizeof(c->stats_ex));
    return GSR_OK;
}
int32_t gsr_hem_get_kernel_ms(gsr_hem_ctx* c, float* out8) {
    if (!c || !out8) return fail(GSR_E_INVALID, "gsr_hem_get_kernel_ms: NULL argument");
    memcpy(out8, c->kernel_ms, sizeof(c->kernel_ms));
    return GSR_OK;
}
int32_t gsr_hem_set_timing(gsr_hem_ctx* c, int32_t level) {
    if (!c || level < 0 || level > 2) return fail(GSR_E_INVALID, "gsr_hem_set_timing: NULL context or level outside 0..2");
    c->timing = level;
    return GSR_OK;
}
int32_t gsr_hem_get_phase_ms(gsr_hem_ctx* c, float* out8) {
    if (!c || !out8) return fail(GSR_E_INVALID, "gsr_hem_get_phase_ms: NULL argument");
    memcpy(out8, c->phase_ms, sizeof(c->phase_ms));
    return GSR_OK;
}

// GSR_HEM_DEBUG_SYNC=1: synchronise and report after every stage of a level (localises a device fault)
// Always: a failed LAUNCH of the stage just enqueued (bad configuration, too much LDS) is reported with the stage's name.
// The events between the phases (gsr_hem_get_phase_ms / _kernel_ms).  An event record between two kernels is a barrier packet of its
// own, 22 per level: 0.1 ms of a 5 M level, 6 % of a 556 k one.  gsr_hem_set_timing: 0 = none, 1 (default) = the level and the two
// dominant kernels (GSR_TIME1), 2 = every phase.
#define GSR_TIME(ev, stream) do { if (c->timing >= 2) GSR_HIP(hipEventRecord(ev, stream)); } while (0)
#define GSR_TIME1(ev, stream) do { if (c->timing >= 1) GSR_HIP(hipEventRecord(ev, stream)); } while (0)
#define GSR_CHECKPOINT(label)                                                                       \
    do {                                                                                            \
        {                                                                                           \
            const hipError_t _l = hipGetLastError();                                                \
            if (_l != hipSuccess) return fail(GSR_E_HIP, "%s: launch failed: %s", label, hipGetErrorString(_l)); \
        }                                                                                           \
        if (dbg_sync) {                                                                             \
            hipError_t _e = hipStreamSynchronize(st);                                               \
            fprintf(stderr, "[gsr_hem] %s: %s\n", label, hipGetErrorString(_e));                    \
            fflush(stderr);                                                                         \
            if (_e != hipSuccess) return fail(GSR_E_HIP, "%s: %s", label, hipGetErrorString(_e));   \
        }                                                                                           \
    } while (0)

namespace {

// Host-side state of ONE spatially partitioned level (gsr_hem_set_comm + gsr_hem_set_level0_part): everything
// the plain level does not have -- who sends which rows to whom, the exchanges along those lists, the global output ranks.  The
// level itself (gsr_hem_run_level) reads as the single-GPU level with five calls into this for `part`.
struct PartLevel {
    gsr_hem_ctx* c;
    hipStream_t st;
    int W = 1, me = 0;
    int64_t n_own = 0;                          // owned components = the first n_own local indices; the ghosts follow
    // halo bookkeeping: rows sent to / received from every rank, and their offsets in the concatenated buffers
    int64_t send_cnt[8] = {0}, recv_cnt[8] = {0}, soff[8] = {0}, roff[8] = {0}, n_sent = 0, n_ghost = 0;
    bool halo_sh_pending = false;               // the ghosts' SH rows are still on their way (third stream)

    // Ownership follows the parents, so a rank CAN run out of components on a later level: that is data, not a local error.  The
    // ranks agree on the level's preconditions before its first data collective (one all-reduce of a status word), so that
    // every rank returns the error instead of one returning and its peers waiting in the next collective for ever.
    int32_t agree_on_preconditions(int64_t n, unsigned local_code) {
        GSR_TRY(c->pcounts.reserve(64));
        // (every rank-local precondition rides in the word: the largest code over the ranks is everybody's answer)
        const unsigned status = local_code ? local_code : (n == 0 ? 1u : 0u);
        GSR_HIP(hipMemcpyAsync(c->pcounts.p, &status, 4, hipMemcpyHostToDevice, st));
        GSR_HIP(hipStreamSynchronize(st));               // (status lives on this stack frame)
        GSR_TRY(gsr_comm_allreduce(c->comm, c->pcounts.p, 1, GSR_DT_U32, GSR_OP_MAX, (void*)st));
        unsigned agreed = 0;
        GSR_HIP(hipMemcpyAsync(&agreed, c->pcounts.p, 4, hipMemcpyDeviceToHost, st));
        GSR_HIP(hipStreamSynchronize(st));
        if (agreed == 1u) return fail(GSR_E_INVALID, "gsr_hem_run_level: a rank of the partitioned level owns no component (reported on every rank; use fewer ranks)");
        if (agreed == 2u) return fail(GSR_E_INVALID, "gsr_hem_run_level: a rank holds 2^30 or more components (reported on every rank)");
        if (agreed == 3u) return fail(GSR_E_INVALID, "gsr_hem_run_level: more than 8 ranks");
        if (agreed == 4u) return fail(GSR_E_INVALID, "gsr_hem_run_level: spatial partition and work sharding are exclusive (reported on every rank)");
        if (agreed == 5u) return fail(GSR_E_HIP, "gsr_hem_run_level: a rank could not allocate the buffers of its halo (reported on every rank)");
        if (agreed) return fail(GSR_E_INVALID, "gsr_hem_run_level: a rank failed a precondition of the partitioned level (code %u)", agreed);
        return GSR_OK;
    }

    // one typed exchange along the halo's lists: to_owner = the ghosts' values go to their owners (roles of the lists reversed)
    int32_t exchange(const void* sendbuf, void* recvbuf, size_t elem, bool to_owner, hipStream_t xs = nullptr) {
        int64_t so[8], sb[8], ro[8], rb[8];
        for (int q = 0; q < W; ++q) {
            so[q] = (to_owner ? roff[q] : soff[q]) * (int64_t)elem; sb[q] = (to_owner ? recv_cnt[q] : send_cnt[q]) * (int64_t)elem;
            ro[q] = (to_owner ? soff[q] : roff[q]) * (int64_t)elem; rb[q] = (to_owner ? send_cnt[q] : recv_cnt[q]) * (int64_t)elem;
            if (q != me) c->part_stats[3] += rb[q];
        }
        return gsr_comm_exchange(c->comm, sendbuf, so, sb, recvbuf, ro, rb, (void*)(xs ? xs : st));
    }

    // The halo: which cells do my parents' search regions touch -> the masks of all ranks -> my components they need -> rows.
    // On return n = owned + ghosts, and c->rec_loc / c->gid_loc hold the working set's records and global indices.
    int32_t halo(const GridParams& gp, Level& L, int F, int64_t& n) {
        const dim3 blk(256), grd(stride_grid(n_own));
        // ---- halo: which cells do my parents' search spheres touch -> masks of all ranks -> my components they need -> rows
        const int64_t mwords = ((int64_t)gp.ncells + 31) / 32 + 1;
        GSR_TRY(c->cmask.reserve((size_t)W * 2 * mwords * 4));
        unsigned* my_mask = c->cmask.as<unsigned>() + (int64_t)me * 2 * mwords;      // [cells wanted of regular components | of irregular ones]
        GSR_HIP(hipMemsetAsync(my_mask, 0, (size_t)2 * mwords * 4, st));
        hipLaunchKernelGGL(k_mark_cells, dim3(stride_grid(n_own * 16)), blk, 0, st, n_own, c->rec.as<float4>(), c->gparams.as<GridParams>(), c->delta,
                           c->delta * c->delta * 0.5f, c->use_ell ? 1 : 0, my_mask, my_mask + mwords);
        GSR_TRY(gsr_comm_allgather(c->comm, my_mask, c->cmask.p, 2 * mwords * 4, (void*)st));
        GSR_TRY(c->dflag.reserve((size_t)W * n_own * 4)); GSR_TRY(c->dpos.reserve((size_t)W * n_own * 4));
        hipLaunchKernelGGL(k_dest_flags, grd, blk, 0, st, n_own, c->rec.as<float4>(), c->gparams.as<GridParams>(), W, me, mwords, c->cmask.as<unsigned>(),
                           c->dflag.as<int>());
        GSR_TRY(c->pcounts.reserve(64)); GSR_TRY(c->pmatrix.reserve(64 * 8));
        GSR_HIP(hipMemsetAsync(c->pcounts.p, 0, 64, st));
        for (int q = 0; q < W; ++q) {
            if (q == me) continue;
            GSR_TRY(exclusive_scan<int>(c, c->dflag.as<int>() + (int64_t)q * n_own, c->dpos.as<int>() + (int64_t)q * n_own, n_own));
            hipLaunchKernelGGL(k_last_total, dim3(1), dim3(1), 0, st, c->dpos.as<int>() + (int64_t)q * n_own + (n_own - 1),
                               c->dflag.as<int>() + (int64_t)q * n_own + (n_own - 1), c->pcounts.as<long long>() + q);
        }
        // (the rank's own slot of its row is free: its owned count rides there, so that every rank knows every rank's working set)
        const long long own_ll = (long long)n_own;
        GSR_HIP(hipMemcpyAsync(c->pcounts.as<long long>() + me, &own_ll, 8, hipMemcpyHostToDevice, st));
        GSR_TRY(gsr_comm_allgather(c->comm, c->pcounts.p, c->pmatrix.p, 64, (void*)st));
        long long mat[64];
        GSR_HIP(hipMemcpyAsync(mat, c->pmatrix.p, (size_t)W * 64, hipMemcpyDeviceToHost, st));
        GSR_HIP(hipStreamSynchronize(st));
        for (int q = 0; q < W; ++q) { send_cnt[q] = q == me ? 0 : mat[me * 8 + q]; recv_cnt[q] = q == me ? 0 : mat[q * 8 + me]; }
        for (int q = 0; q < W; ++q) { soff[q] = n_sent; n_sent += send_cnt[q]; roff[q] = n_ghost; n_ghost += recv_cnt[q]; }
        // Errors past this point would be rank-local with the collectives already under way (a peer would wait in the next one for
        // ever): the size limit is evaluated for EVERY rank from the matrix all of them hold, and the allocations'
        // outcome is agreed on before the first exchange
        for (int r = 0; r < W; ++r) {
            long long tot = 0;
            for (int q = 0; q < W; ++q) tot += mat[q * 8 + r];                   // owned (q == r) + what every peer sends
            if (tot >= (1ll << 30))
                return fail(GSR_E_INVALID, "gsr_hem_run_level: rank %d would hold %lld local components (owned + ghosts; the limit is 2^30; reported on every rank)", r, tot);
        }
        // TWO exchanges along the same lists: the 72-byte rows {record, global index, index at the owner} -- what the grid, the sort
        // and the selection need -- on the level's stream, and the SH rows (4 F bytes: 71 % of a ghost at SH degree 3), which only the
        // M-step reads, on the third stream beside the rest of the grid phase and the selection, received straight into ghost_sh
        constexpr int RW = 16 + PART_ROW_EXTRA;
        const size_t Fm = (size_t)(F > 0 ? F : 1);
        n = n_own + n_ghost;
        {
            int32_t rs = c->rows_send.reserve((size_t)(n_sent > 0 ? n_sent : 1) * RW * 4);
            const auto also = [&rs](int32_t r) { if (rs == GSR_OK) rs = r; };
            also(c->rows_recv.reserve((size_t)(n_ghost > 0 ? n_ghost : 1) * RW * 4));
            also(c->sh_send.reserve((size_t)(n_sent > 0 ? n_sent : 1) * Fm * 4));
            also(c->sent_idx.reserve((size_t)(n_sent > 0 ? n_sent : 1) * 4));
            also(c->ghost_sh.reserve((size_t)(n_ghost > 0 ? n_ghost : 1) * Fm * 4));
            also(c->ghost_src.reserve((size_t)(n_ghost > 0 ? n_ghost : 1) * 4));
            also(c->rec_loc.reserve((size_t)n * 64));
            also(c->gid_loc.reserve((size_t)n * 4));
            GSR_TRY(agree_on_preconditions(1, rs != GSR_OK ? 5u : 0u));
        }
        for (int q = 0; q < W; ++q)
            if (send_cnt[q] > 0)
                hipLaunchKernelGGL(k_pack_rows, dim3(stride_grid(n_own * 64)), blk, 0, st, n_own, F, c->dflag.as<int>() + (int64_t)q * n_own,
                                   c->dpos.as<int>() + (int64_t)q * n_own, c->rec.as<float4>(), L.sh.as<float>(), c->gid.as<unsigned>(),
                                   c->rows_send.as<float>() + soff[q] * RW, c->sh_send.as<float>() + soff[q] * F, c->sent_idx.as<unsigned>() + soff[q]);
        GSR_TIME(c->evp[0], st);
        GSR_TRY(exchange(c->rows_send.p, c->rows_recv.p, (size_t)RW * 4, false));
        GSR_TIME(c->evp[1], st);
        if (F > 0) {        // every rank issues it (the same order of communicator calls everywhere), whatever its own counts
            GSR_HIP(hipEventRecord(c->ev_sh_fork, st)); GSR_HIP(hipStreamWaitEvent(c->aux2, c->ev_sh_fork, 0));
            GSR_TIME(c->evp[2], c->aux2);
            GSR_TRY(exchange(c->sh_send.p, c->ghost_sh.p, (size_t)F * 4, false, c->aux2));
            GSR_TIME(c->evp[3], c->aux2);
            GSR_HIP(hipEventRecord(c->ev_halo, c->aux2));
            halo_sh_pending = true;
        }
        c->part_stats[0] = n_ghost; c->part_stats[1] = n_sent; c->part_stats[2] = c->part_stats[3]; c->part_stats[3] = 0;
        GSR_HIP(hipMemcpyAsync(c->rec_loc.p, c->rec.p, (size_t)n_own * 64, hipMemcpyDeviceToDevice, st));
        GSR_HIP(hipMemcpyAsync(c->gid_loc.p, c->gid.p, (size_t)n_own * 4, hipMemcpyDeviceToDevice, st));
        if (n_ghost > 0)
            hipLaunchKernelGGL(k_unpack_rows, dim3(stride_grid(n_ghost * 16)), blk, 0, st, n_ghost, n_own, c->rows_recv.as<float>(), c->rec_loc.as<float4>(),
                               c->gid_loc.as<unsigned>(), c->ghost_src.as<unsigned>());
        c->stats[6] = n_own;
        return GSR_OK;
    }

    // k_bucket_sum's three steps with the ghosts' partial results sent to their owners in between -- integers only (maximum,
    // 64-bit fixed-point sums), then the owners' finished float32 sums back to the ghosts
    int32_t sums(int64_t n, int P, int64_t M, const int64_t* seg, const unsigned* pc, const float* pw, int nbuckets, int bshift, int* overflow_flag) {
        const dim3 blk(256), grd(stride_grid(n));
        // the communicator's calls in ONE order on the device too: the exchanges below (this stream) behind the SH rows' (third stream)
        if (halo_sh_pending) { GSR_HIP(hipStreamWaitEvent(st, c->ev_halo, 0)); halo_sh_pending = false; }
        if (nbuckets > SUM_MAX_BUCKETS) return fail(GSR_E_INVALID, "gsr_hem_run_level: level too large for the partitioned sums");
        GSR_TRY(c->gmax.reserve((size_t)n * 4)); GSR_TRY(c->gacc.reserve((size_t)n * 8)); GSR_TRY(c->bcursor.reserve(((size_t)nbuckets + 1) * 8));
        unsigned cap = 0;
        for (double factor = 8.0;; factor *= 2.0) {            // bucket regions of fixed capacity; doubled until nothing overflows
            const double capd = (double)M / (double)nbuckets * factor + 4096.0;
            if (capd > 4.0e9) return fail(GSR_E_INVALID, "gsr_hem_run_level: pair partition capacity");
            cap = (unsigned)capd;
            GSR_TRY(c->spair_child.reserve((size_t)nbuckets * cap * sizeof(slot_t))); GSR_TRY(c->spair_wl.reserve((size_t)nbuckets * cap * 4));
            GSR_HIP(hipMemsetAsync(c->bcursor.p, 0, ((size_t)nbuckets + 1) * 4, st));
            GSR_HIP(hipMemsetAsync(overflow_flag, 0, 4, st));
            if (M > 0 && P > 0)
                launch_partition(st, c->partition_staged ? (c->partition_stage ? c->partition_stage : 1) : 0, P, seg, c->pcnt.as<unsigned>(), pc, pw, nbuckets, bshift, cap, c->bcursor.as<unsigned>(),
                                 c->spair_child.as<slot_t>(), c->spair_wl.as<float>(), overflow_flag);
            GSR_HIP(hipGetLastError());
            Collect q;
            q.n = 1; q.src[0] = overflow_flag; q.bytes[0] = 4;
            unsigned long long w[8];
            GSR_TRY(read_back(c, q, w));
            if (w[0] == 0ull) break;
        }
        const dim3 bblk(bshift >= 10 ? 1024 : 256);
        const dim3 gs(stride_grid(n_sent > 0 ? n_sent : 1)), gg(stride_grid(n_ghost > 0 ? n_ghost : 1));
        GSR_TRY(c->xsend.reserve((size_t)(n_sent + n_ghost + 1) * 8)); GSR_TRY(c->xrecv.reserve((size_t)(n_sent + n_ghost + 1) * 8));
        // 1. the largest |wL| of every child: local, then the ghosts' maxima to their owners, then the owners' result back
        hipLaunchKernelGGL(k_part_max, dim3(nbuckets), bblk, (size_t)4 << bshift, st, n, bshift, cap, c->bcursor.as<unsigned>(), c->spair_child.as<slot_t>(),
                           c->spair_wl.as<float>(), c->gmax.as<unsigned>());
        GSR_HIP(hipGetLastError());
        if (n_ghost > 0) hipLaunchKernelGGL(k_ghost_gather<unsigned>, gg, blk, 0, st, n_ghost, n_own, c->inv.as<unsigned>(), c->gmax.as<unsigned>(), c->xsend.as<unsigned>());
        GSR_TRY(exchange(c->xsend.p, c->xrecv.p, 4, true));
        if (n_sent > 0) hipLaunchKernelGGL(k_sent_apply_max, gs, blk, 0, st, n_sent, c->sent_idx.as<unsigned>(), c->inv.as<unsigned>(), c->xrecv.as<unsigned>(), c->gmax.as<unsigned>());
        if (n_sent > 0) hipLaunchKernelGGL(k_sent_gather<unsigned>, gs, blk, 0, st, n_sent, c->sent_idx.as<unsigned>(), c->inv.as<unsigned>(), c->gmax.as<unsigned>(), c->xsend.as<unsigned>());
        GSR_TRY(exchange(c->xsend.p, c->xrecv.p, 4, false));
        if (n_ghost > 0) hipLaunchKernelGGL(k_ghost_set<unsigned>, gg, blk, 0, st, n_ghost, n_own, c->inv.as<unsigned>(), c->xrecv.as<unsigned>(), c->gmax.as<unsigned>());
        // 2. the fixed-point sums on that scale: local, then the ghosts' partial sums to their owners (integer addition)
        hipLaunchKernelGGL(k_part_acc, dim3(nbuckets), bblk, (size_t)12 << bshift, st, n, bshift, cap, c->bcursor.as<unsigned>(), c->spair_child.as<slot_t>(),
                           c->spair_wl.as<float>(), c->gmax.as<unsigned>(), c->gacc.as<unsigned long long>());
        GSR_HIP(hipGetLastError());
        if (n_ghost > 0) hipLaunchKernelGGL(k_ghost_gather<unsigned long long>, gg, blk, 0, st, n_ghost, n_own, c->inv.as<unsigned>(), c->gacc.as<unsigned long long>(), c->xsend.as<unsigned long long>());
        GSR_TRY(exchange(c->xsend.p, c->xrecv.p, 8, true));
        if (n_sent > 0) hipLaunchKernelGGL(k_sent_apply_acc, gs, blk, 0, st, n_sent, c->sent_idx.as<unsigned>(), c->inv.as<unsigned>(), c->xrecv.as<unsigned long long>(),
                                           c->gmax.as<unsigned>(), c->gacc.as<unsigned long long>());
        // 3. the float32 sums (correct for the owned components), the owners' values back to the ghosts, orphans among the owned
        hipLaunchKernelGGL(k_part_finish, grd, blk, 0, st, n, c->gmax.as<unsigned>(), c->gacc.as<unsigned long long>(), c->sumLw.as<float>());
        if (n_sent > 0) hipLaunchKernelGGL(k_sent_gather<float>, gs, blk, 0, st, n_sent, c->sent_idx.as<unsigned>(), c->inv.as<unsigned>(), c->sumLw.as<float>(), c->xsend.as<float>());
        GSR_TRY(exchange(c->xsend.p, c->xrecv.p, 4, false));
        if (n_ghost > 0) hipLaunchKernelGGL(k_ghost_set<float>, gg, blk, 0, st, n_ghost, n_own, c->inv.as<unsigned>(), c->xrecv.as<float>(), c->sumLw.as<float>());
        hipLaunchKernelGGL(k_part_orphans, grd, blk, 0, st, n, n_own, c->order.as<unsigned>(), c->sumLw.as<float>(), c->oflag.as<int>(), c->geo.as<float>() + 15);
        return GSR_OK;
    }

    // The rows' GLOBAL ranks (a parent's = parents below it in the level's global order, an orphan's = all parents + orphans below
    // it): bit maps of the parents' and orphans' global indices, summed over the ranks (disjoint bits)
    int32_t global_ranks(int P, int64_t n_pre, int64_t& P_glob, int64_t& O_glob) {
        const dim3 blk(256);
        const int64_t words = (c->n_global + 31) / 32 + 1;
        GSR_TRY(c->gbits.reserve((size_t)2 * words * 4)); GSR_TRY(c->wcnt.reserve((size_t)2 * words * 4)); GSR_TRY(c->wpre.reserve((size_t)2 * words * 4));
        GSR_TRY(c->grank.reserve((size_t)2 * n_own * 4)); GSR_TRY(c->gid_next.reserve((size_t)(n_pre > 0 ? n_pre : 1) * 4));     // [ranks as parents | as orphans]
        GSR_HIP(hipMemsetAsync(c->gbits.p, 0, (size_t)2 * words * 4, st));
        const dim3 go(stride_grid(n_own));
        hipLaunchKernelGGL(k_bits_set, go, blk, 0, st, n_own, c->gid.as<unsigned>(), c->pflag_in.as<int>(), c->gbits.as<unsigned>());
        hipLaunchKernelGGL(k_bits_set, go, blk, 0, st, n_own, c->gid.as<unsigned>(), c->oflag_in.as<int>(), c->gbits.as<unsigned>() + words);
        GSR_TRY(gsr_comm_allreduce(c->comm, c->gbits.p, 2 * words, GSR_DT_U32, GSR_OP_SUM, (void*)st));
        hipLaunchKernelGGL(k_bits_popc, dim3(stride_grid(2 * words)), blk, 0, st, 2 * words, c->gbits.as<unsigned>(), c->wcnt.as<int>());
        GSR_TRY(exclusive_scan<int>(c, c->wcnt.as<int>(), c->wpre.as<int>(), words));
        GSR_TRY(exclusive_scan<int>(c, c->wcnt.as<int>() + words, c->wpre.as<int>() + words, words));
        {
            Collect q;
            q.n = 4;
            q.src[0] = c->wpre.as<int>() + (words - 1); q.src[1] = c->wcnt.as<int>() + (words - 1);
            q.src[2] = c->wpre.as<int>() + (2 * words - 1); q.src[3] = c->wcnt.as<int>() + (2 * words - 1);
            for (int i = 0; i < 4; ++i) q.bytes[i] = 4;
            unsigned long long w[8];
            GSR_TRY(read_back(c, q, w));
            P_glob = (int64_t)w[0] + (int64_t)w[1]; O_glob = (int64_t)w[2] + (int64_t)w[3];
        }
        hipLaunchKernelGGL(k_bits_rank, go, blk, 0, st, n_own, c->gid.as<unsigned>(), c->pflag_in.as<int>(), c->gbits.as<unsigned>(), c->wpre.as<int>(), 0u,
                           c->grank.as<unsigned>());
        hipLaunchKernelGGL(k_bits_rank, go, blk, 0, st, n_own, c->gid.as<unsigned>(), c->oflag_in.as<int>(), c->gbits.as<unsigned>() + words, c->wpre.as<int>() + words,
                           (unsigned)P_glob, c->grank.as<unsigned>() + n_own);
        hipLaunchKernelGGL(k_part_new_gid, go, blk, 0, st, n_own, P, c->pflag_in.as<int>(), c->prank_in.as<int>(), c->oflag_in.as<int>(), c->orank_in.as<int>(),
                           c->grank.as<unsigned>(), c->grank.as<unsigned>() + n_own, c->gid_next.as<unsigned>());
        c->part_stats[4] = P_glob; c->part_stats[5] = O_glob;
        return GSR_OK;
    }

    // Erased rows leave the GLOBAL numbering too: how many over all ranks, and (rarely more than none) which
    int32_t drop_erased(Level& O, int64_t n_pre, int64_t n_pre_glob, int64_t dropped, int64_t& n_glob_next) {
        const dim3 blk(256);
        // erased rows leave the GLOBAL numbering too: how many over all ranks, and (rarely more than none) which
        GSR_TRY(c->pcounts.reserve(64));
        const long long dl = dropped;
        GSR_HIP(hipMemcpyAsync(c->pcounts.p, &dl, 8, hipMemcpyHostToDevice, st));
        GSR_HIP(hipStreamSynchronize(st));                      // (dl lives on this stack frame)
        GSR_TRY(gsr_comm_allreduce(c->comm, c->pcounts.p, 1, GSR_DT_U64, GSR_OP_SUM, (void*)st));
        long long dg = 0;
        GSR_HIP(hipMemcpyAsync(&dg, c->pcounts.p, 8, hipMemcpyDeviceToHost, st));
        GSR_HIP(hipStreamSynchronize(st));
        if (dg > 0) {
            const int64_t words = (n_pre_glob + 31) / 32 + 1;
            GSR_TRY(c->gbits.reserve((size_t)words * 4)); GSR_TRY(c->wcnt.reserve((size_t)words * 4)); GSR_TRY(c->wpre.reserve((size_t)words * 4));
            GSR_HIP(hipMemsetAsync(c->gbits.p, 0, (size_t)words * 4, st));
            if (n_pre > 0) {
                GSR_TRY(c->scratch.reserve((size_t)n_pre * 4));
                hipLaunchKernelGGL(k_not_flag, dim3(stride_grid(n_pre)), blk, 0, st, n_pre, c->keep.as<int>(), c->scratch.as<int>());
                hipLaunchKernelGGL(k_bits_set, dim3(stride_grid(n_pre)), blk, 0, st, n_pre, c->gid_next.as<unsigned>(), c->scratch.as<int>(), c->gbits.as<unsigned>());
            }
            GSR_TRY(gsr_comm_allreduce(c->comm, c->gbits.p, words, GSR_DT_U32, GSR_OP_SUM, (void*)st));
            hipLaunchKernelGGL(k_bits_popc, dim3(stride_grid(words)), blk, 0, st, words, c->gbits.as<unsigned>(), c->wcnt.as<int>());
            GSR_TRY(exclusive_scan<int>(c, c->wcnt.as<int>(), c->wpre.as<int>(), words));
            if (dropped > 0) {                                  // my own erased rows out of my list first
                GSR_TRY(c->grank.reserve((size_t)n_pre * 4));
                hipLaunchKernelGGL(k_compact_u32, dim3(stride_grid(n_pre)), blk, 0, st, n_pre, c->keep.as<int>(), c->kpos.as<int>(), c->gid_next.as<unsigned>(), c->grank.as<unsigned>());
                c->gid_next.swap(c->grank);
            }
            if (O.n > 0) hipLaunchKernelGGL(k_gid_drop, dim3(stride_grid(O.n)), blk, 0, st, O.n, c->gbits.as<unsigned>(), c->wpre.as<int>(), c->gid_next.as<unsigned>());
            n_glob_next = n_pre_glob - dg;
        }
        c->part_stats[6] = dg;
        return GSR_OK;
    }
};

}  // namespace

namespace {

// internal status of LevelRun::run: the buffers an asynchronous level ran on were too small (or one of its other assumptions did not hold);
// nothing of the level's input has been touched -- gsr_hem_run_level runs it again the synchronous way, which sizes everything exactly
constexpr int32_t GSR_RETRY_SYNC = -1000;

// One level (gsr_hem_run_level), stage by stage.  Two schedules through the same stages:
//  * SYNCHRONOUS (rounds 1-4; still what a partitioned / work-sharded level, a fresh context and every fallback path run): the host sizes each
//    stage's buffers from counts it reads back on the way -- candidates, pairs, orphans, surviving rows: four round trips inside the level;
//  * ASYNCHRONOUS (`spec`, the default once the context's buffers exist): NO round trip between the level's first and its last kernel.
//    What the host knows when it starts is the level's prologue (grid, parents, irregular components: computed when the level's input came
//    into being); everything else stays on the device -- the heavy threshold and the work-item size come from the capacities' scan
//    (k_heavy_keys, k_heavy_items), the pair buffers are the ones the context has and every segment write is clamped to them (k_select),
//    the bucket regions take their capacity from the buffers, the new level's size is k_level_tail's, launches are sized for bounds and
//    their surplus blocks leave on the device count -- and ONE answer comes back behind the last kernel (k_level_collect) together with
//    the NEXT level's prologue.  A raised abort / overflow flag there means the level is run again synchronously (GSR_RETRY_SYNC);
//    its input was never written.
// The reference's level is one function without such a boundary (src/cpp_ext/src/mixture.cpp:25-35, 66-285).
struct LevelRun {
    gsr_hem_ctx* c;
    hipStream_t st;
    Level& L;
    Level& O;
    const bool dbg_sync, part, sharded;
    bool spec = false;
    const int sh_policy;                    // the cell-sorted copy of the SH block: 0 always made, 1 never (rows read from the level's own array), 2 decided on
                                            // the device by the level's pairs per component (k_gather_sh); a partitioned level: 0 (its ghosts' rows lie elsewhere)
    const int64_t n_own;
    int64_t n;
    const int F, RSH;
    PartLevel pl;
    const dim3 blk{256};
    dim3 grd;
    GridParams gp;
    int P_all = 0, P = 0, n_irr = 0;
    int* cnt = nullptr;                     // this level's counter block: [0] [1] heavy parents / segments of the M-step, [2] abort, [3] erased rows,
                                            // [8] heavy parents of the selection, [10] [11] their queue, [12] bucket overflow, [13] item table overflow, [15] max pairs
    long long* lvl = nullptr;               // device: [0] rows of the new level before the erase, [1] orphans, [2] (as unsigned) work-item size
    const float4* rec_src = nullptr;
    bool ranks_forked = false, sh_pending = false, sh_launched = false, flags_forked = false;
    // selection
    SelectArgs sa;
    size_t Pm = 1;
    int own_lo = 0, own_hi = 0;
    int64_t M = 0;
    unsigned long long cand = 0, cap_pairs = 0;
    // per-child sums
    const int64_t* seg = nullptr;
    const unsigned* pc = nullptr;
    const float* pw = nullptr;
    int bshift = 0, nbuckets = 0;
    int* overflow_flag = nullptr;
    bool fixed_tried = false;
    // output
    int64_t n_orph = 0, n_pre = 0, out_cap = 0;
    bool out_active = false;                // the new level is being written into the caller's arrays (gsr_hem_set_output)
    uint64_t rng_pos0 = 0;
    int64_t dropped = 0, P_glob = 0, O_glob = 0, n_pre_glob = 0, n_glob_next = 0;
    float pro_ms = 0.0f;                    // the time of this level's prologue, which ran when its input came into being

    LevelRun(gsr_hem_ctx* ctx, bool allow_async)
        : c(ctx), st(ctx->stream), L(ctx->cur), O(ctx->nxt), dbg_sync(getenv("GSR_HEM_DEBUG_SYNC") != nullptr), part(ctx->comm != nullptr),
          sharded(ctx->shard_world > 1 && ctx->shard_allreduce != nullptr), sh_policy(ctx->comm == nullptr ? ctx->sh_policy : 0), n_own(ctx->cur.n), n(ctx->cur.n), F(ctx->cur.F), RSH((ctx->cur.F + 3) & ~3) {
        spec = allow_async;
        pl.c = c; pl.st = st; pl.n_own = n_own;
        memset(&sa, 0, sizeof(sa));
    }
    ~LevelRun() {       // an error return hands nxt its own buffers back
        if (out_active) for (int i = 0; i < 5; ++i) { DevBuf* b = level_big(c->nxt, i); b->p = nullptr; b->cap = 0; b->swap(c->spare_out[i]); }
    }

    int32_t run(int64_t* n_out, int64_t* n_dropped);
    int32_t grid_phase();
    int32_t select_phase();
    int32_t sums_phase();
    int32_t sums_fixed();
    int32_t sums_exact();
    int32_t compact_pairs();
    int32_t output_ranks();
    int32_t open_output();
    int32_t mstep_phase();
    int32_t flags_and_validity();
    int32_t compact_erased(int64_t n_keep);
    int32_t erase_in_place(int64_t rows_bound, bool may_allocate, int64_t alloc_rows = 0);
    bool erase_on_device = false;           // the erase kernels were enqueued (an asynchronous level without the tails' buffer leaves the erase to the host)
    int32_t launch_gather_sh(bool fork);
    int32_t widen_scan(const unsigned* cnt_in, int64_t* off, int64_t count);
    int32_t total_of(const int64_t* off, const unsigned* cnt_in, int64_t count, int64_t* out);
};

struct WidenU32 { __device__ __host__ int64_t operator()(unsigned v) const { return (int64_t)v; } };
int32_t LevelRun::widen_scan(const unsigned* cnt_in, int64_t* off, int64_t count) {      // off = exclusive scan of cnt_in (int64)
    // (the counts are widened on the fly by the scan's input iterator: a separate transform pass was a launch and 12 bytes per count)
    auto in = rocprim::make_transform_iterator(cnt_in, WidenU32());
    size_t bytes = 0;
    if (count >= GSR_SCAN_BIG_N) {
        GSR_HIP(rocprim::exclusive_scan<scan_cfg64_big>(nullptr, bytes, in, off, (int64_t)0, (size_t)count, rocprim::plus<int64_t>(), st));
        GSR_TRY(c->rocprim_tmp.reserve(bytes));
        GSR_HIP(rocprim::exclusive_scan<scan_cfg64_big>(c->rocprim_tmp.p, bytes, in, off, (int64_t)0, (size_t)count, rocprim::plus<int64_t>(), st));
        return GSR_OK;
    }
    GSR_HIP(rocprim::exclusive_scan(nullptr, bytes, in, off, (int64_t)0, (size_t)count, rocprim::plus<int64_t>(), st));
    GSR_TRY(c->rocprim_tmp.reserve(bytes));
    GSR_HIP(rocprim::exclusive_scan(c->rocprim_tmp.p, bytes, in, off, (int64_t)0, (size_t)count, rocprim::plus<int64_t>(), st));
    return GSR_OK;
}
int32_t LevelRun::total_of(const int64_t* off, const unsigned* cnt_in, int64_t count, int64_t* out) {
    Collect q;
    q.n = 2;
    q.src[0] = off + (count - 1); q.bytes[0] = 8;
    q.src[1] = cnt_in + (count - 1); q.bytes[1] = 4;
    unsigned long long w[8];
    GSR_TRY(read_back(c, q, w));
    *out = (int64_t)w[0] + (int64_t)(unsigned)w[1];
    return GSR_OK;
}

// only the M-step reads the sorted SH rows: the gather (1.9 GB of HBM traffic at 5 M) can run on its own stream beside the
// selection, which is bound by VALU issue and load latency, and is joined in front of the M-step.  sh_overlap: 0 = in line
// here, 1 = forked here (beside the rest of the grid phase), 2 = forked just in front of k_select
int32_t LevelRun::launch_gather_sh(bool fork) {
    if (F <= 0) return GSR_OK;
    hipStream_t sst = st;
    if (fork) {
        GSR_HIP(hipEventRecord(c->ev_sh_fork, st)); GSR_HIP(hipStreamWaitEvent(c->aux2, c->ev_sh_fork, 0));
        sst = c->aux2;
    }
    const int shg = stride_grid(n * (RSH >> 2));
    if (part)
        hipLaunchKernelGGL(k_gather_sh2, dim3(shg), blk, 0, sst, n, n_own, F, RSH, c->order.as<unsigned>(), L.sh.as<float>(),
                           c->ghost_sh.as<float>(), c->shs.as<float>());
    else
        hipLaunchKernelGGL(k_gather_sh, dim3(shg), blk, 0, sst, n, F, RSH, c->order.as<unsigned>(), L.sh.as<float>(), c->shs.as<float>(),
                           P > 0 ? c->poff.as<int64_t>() : (const int64_t*)nullptr, c->pcnt.as<unsigned>(), P, sh_policy, c->sh_direct_pairs,
                           reinterpret_cast<int*>(lvl + 3));
    if (fork) { GSR_HIP(hipEventRecord(c->ev_sh_join, c->aux2)); sh_pending = true; }
    return GSR_OK;
}

// ---- 1. the level's prologue (det, packed records, bounding box, grid) if it is not there yet; sort by cell; the cell-sorted working set ----
int32_t LevelRun::grid_phase() {
    GSR_TRY(c->counters.reserve(128)); GSR_TRY(c->lvl.reserve(64));
    lvl = c->lvl.as<long long>();
    if (!part && !sharded) {
        // the prologue came with the level's input (gsr_hem_set_level0, the level before); if not -- a caller changed the flags, an error
        // path -- it is computed now, with a round trip of its own
        if (!(c->pro.valid && c->pro.n == n)) GSR_TRY(prologue_now(c));
        if (!c->pro.valid) return fail(GSR_E_INVALID, "gsr_hem_run_level: no prologue for the level");
        gp = c->pro.gp; P_all = c->pro.P; n_irr = c->pro.n_irr; c->cblock = c->pro.cblock; pro_ms = c->pro.ms;
        c->pro.valid = false;                   // consumed: `rec` and the counter block belong to this level now
        cnt = c->counters.as<int>() + 16 * c->cblock;
    } else {
        c->pro.valid = false;
        c->cblock ^= 1;
        cnt = c->counters.as<int>() + 16 * c->cblock;
        GSR_TRY(c->rec.reserve(n * 64)); GSR_TRY(c->bbox.reserve(64));
        GSR_TRY(c->gparams.reserve(sizeof(GridParams)));
        GSR_TRY(c->bbox_part.reserve((size_t)grd.x * 8 * 4)); GSR_TRY(c->hist.reserve(3 * HIST_BINS * 4));
        hipLaunchKernelGGL(k_prep, grd, blk, 0, st, n, L.xyz.as<float>(), L.color.as<float>(), L.cov6.as<float>(), L.opacity.as<float>(),
                           L.weight.as<float>(), L.is_parent.as<uint8_t>(), c->rec.as<float4>(), c->bbox_part.as<unsigned>(), (const long long*)nullptr);
        hipLaunchKernelGGL(k_bbox_reduce, dim3(1), dim3(256), 0, st, (int)grd.x, c->bbox_part.as<unsigned>(), c->bbox.as<unsigned>(),
                           (unsigned*)cnt, 16, c->hist.as<unsigned>(), 3 * HIST_BINS, (unsigned*)nullptr, 0);
        if (part) {     // the box of ALL ranks' components: maximum of the (order-preserving) codes, the minima complemented
            hipLaunchKernelGGL(k_flip3, dim3(1), dim3(64), 0, st, c->bbox.as<unsigned>());
            GSR_TRY(gsr_comm_allreduce(c->comm, c->bbox.p, 6, GSR_DT_U32, GSR_OP_MAX, (void*)st));
            hipLaunchKernelGGL(k_flip3, dim3(1), dim3(64), 0, st, c->bbox.as<unsigned>());
        }
        hipLaunchKernelGGL(k_hist, dim3(stride_grid(n) > 512 ? 512 : stride_grid(n)), blk, 0, st, n, L.xyz.as<float>(), c->bbox.as<unsigned>(), c->hist.as<unsigned>(),
                           (const long long*)nullptr);
        if (part) GSR_TRY(gsr_comm_allreduce(c->comm, c->hist.p, 3 * HIST_BINS, GSR_DT_U32, GSR_OP_SUM, (void*)st));      // integer counts: exact
        hipLaunchKernelGGL(k_grid_params, dim3(1), dim3(64), 0, st, c->bbox.as<unsigned>(), c->hist.as<unsigned>(), part ? c->n_global : n, c->cell_target,
                           c->max_cells, c->gparams.as<GridParams>(), (const long long*)nullptr);
        static_assert(sizeof(GridParams) == 40, "GridParams is read back as five 8-byte words");
        Collect q;
        q.n = 7;
        for (int i = 0; i < 5; ++i) { q.src[i] = (const char*)c->gparams.p + 8 * i; q.bytes[i] = 8; }
        q.src[5] = c->bbox.as<unsigned>() + 6; q.src[6] = c->bbox.as<unsigned>() + 7;      // k_prep's counts: parents, irregular components
        q.bytes[5] = q.bytes[6] = 4;
        unsigned long long w[8];
        GSR_TRY(read_back(c, q, w));
        memcpy(&gp, w, sizeof(gp));
        P_all = (int)(unsigned)w[5]; n_irr = (int)(unsigned)w[6];
    }
    overflow_flag = cnt + 12;
    c->stats[5] = gp.ncells;
    // The parents' output ranks depend on nothing but the level's flags (input order): flags as ints + their scan on the second
    // stream, beside the grid phase, instead of between the sums and the M-step (three launches off the critical path).
    GSR_TRY(c->pflag_in.reserve(n * 4)); GSR_TRY(c->oflag_in.reserve(n * 4)); GSR_TRY(c->prank_in.reserve(n * 4)); GSR_TRY(c->orank_in.reserve(n * 4));
    if (!part && c->aux && c->ev_pre) {
        GSR_HIP(hipEventRecord(c->ev_fork, st)); GSR_HIP(hipStreamWaitEvent(c->aux, c->ev_fork, 0));
        hipLaunchKernelGGL(k_flags_in, grd, blk, 0, c->aux, n, n_own, L.is_parent.as<uint8_t>(), c->pflag_in.as<int>(), c->oflag_in.as<int>());
        GSR_TRY(exclusive_scan<int>(c, c->pflag_in.as<int>(), c->prank_in.as<int>(), n, true));
        GSR_HIP(hipEventRecord(c->ev_pre, c->aux));
        ranks_forked = true;
    }

    rec_src = c->rec.as<float4>();              // the packed records of the working set, by local index
    if (part) {
        GSR_TRY(pl.halo(gp, L, F, n));
        grd = dim3(stride_grid(n));
        rec_src = c->rec_loc.as<float4>();
    }
    GSR_TRY(c->keys.reserve(n * 4)); GSR_TRY(c->idx.reserve(n * 4)); GSR_TRY(c->skeys.reserve(n * 4)); GSR_TRY(c->order.reserve(n * 4));
    if (part) {
        // local components in ascending GLOBAL index (what a single GPU's input order is), then the stable sort by cell
        GSR_TRY(c->perm.reserve(n * 4));
        hipLaunchKernelGGL(k_iota, grd, blk, 0, st, n, c->idx.as<unsigned>());
        GSR_TRY(sort_pairs<unsigned>(c, c->gid_loc.as<unsigned>(), c->skeys.as<unsigned>(), c->idx.as<unsigned>(), c->perm.as<unsigned>(), n, bits_for(c->n_global + 1)));
        hipLaunchKernelGGL(k_keys_rec, grd, blk, 0, st, n, rec_src, c->perm.as<unsigned>(), c->gparams.as<GridParams>(), c->keys.as<unsigned>(), c->idx.as<unsigned>());
    } else
        hipLaunchKernelGGL(k_keys, grd, blk, 0, st, n, L.xyz.as<float>(), c->gparams.as<GridParams>(), c->keys.as<unsigned>(), c->idx.as<unsigned>());
    GSR_TRY(sort_pairs<unsigned>(c, c->keys.as<unsigned>(), c->skeys.as<unsigned>(), c->idx.as<unsigned>(), c->order.as<unsigned>(), n,
                                 bits_for(gp.ncells)));
    GSR_TRY(c->cellStart.reserve(((size_t)gp.ncells + 1) * 4));
    hipLaunchKernelGGL(k_run_starts<int>, grd, blk, 0, st, n, c->skeys.as<unsigned>(), (int64_t)gp.ncells, c->cellStart.as<int>());

    GSR_TRY(c->A.reserve((n + SEL_PAD) * 16)); GSR_TRY(c->geo.reserve((size_t)n * 64));
    if (sh_policy != 1) GSR_TRY(c->shs.reserve((size_t)n * (RSH > 0 ? RSH : 1) * 4));
    GSR_TRY(c->Rs.reserve(n * 4)); GSR_TRY(c->pflag.reserve(n * 4));
    GSR_TRY(c->iflag.reserve((n + 1) * 4)); GSR_TRY(c->irank.reserve((n + 1) * 4)); GSR_TRY(c->ipos.reserve(n * 4)); GSR_TRY(c->ppos.reserve(n * 4)); GSR_TRY(c->plist.reserve(n * 4));
    const bool tail = !part && F > 0;
    if (tail) GSR_TRY(c->sh_tail.reserve((size_t)(RSH + 4) * 4));
    hipLaunchKernelGGL(k_gather, dim3((unsigned)((n + 255) / 256)), blk, 0, st, n, c->order.as<unsigned>(), rec_src, c->delta, c->A.as<float4>(), c->geo.as<float4>(),
                       c->Rs.as<float>(), c->pflag.as<int>(), c->iflag.as<int>(), L.sh.as<float>(), F, tail ? c->sh_tail.as<float>() : (float*)nullptr);
    // (a partitioned level: forked here -- the third stream carries the ghosts' SH rows, the gather queues up behind them; one GPU: behind
    // the selection, when the pair count that decides about the copy exists)
    if (part && F > 0) {
        hipLaunchKernelGGL(k_fill_const<int>, dim3(1), dim3(1), 0, st, (int64_t)1, reinterpret_cast<int*>(lvl + 3), 0);
        GSR_TRY(launch_gather_sh(true)); sh_launched = true;
    }
    GSR_TRY(exclusive_scan<int>(c, c->pflag.as<int>(), c->ppos.as<int>(), n));
    if (part) {     // the parents this rank works on are the ones it owns; the ghosts' parent flags still keep them out of the children's stream
        GSR_TRY(c->pown.reserve(n * 4)); GSR_TRY(c->ppos_own.reserve(n * 4)); GSR_TRY(c->inv.reserve(n * 4));
        hipLaunchKernelGGL(k_own_flags, grd, blk, 0, st, n, n_own, c->order.as<unsigned>(), c->pflag.as<int>(), c->pown.as<int>(), c->inv.as<unsigned>());
        GSR_TRY(exclusive_scan<int>(c, c->pown.as<int>(), c->ppos_own.as<int>(), n));
        hipLaunchKernelGGL(k_scatter_list, grd, blk, 0, st, n, c->pown.as<int>(), c->ppos_own.as<int>(), c->plist.as<unsigned>());
    } else
        hipLaunchKernelGGL(k_scatter_list, grd, blk, 0, st, n, c->pflag.as<int>(), c->ppos.as<int>(), c->plist.as<unsigned>());
    // the irregular components (never pre-rejected): their sorted positions, and their rank at every position
    // (a level without any -- k_prep counted them -- builds no list: pass B does not run then)
    if (part || n_irr > 0) {
        GSR_TRY(exclusive_scan<int>(c, c->iflag.as<int>(), c->irank.as<int>(), n + 1));
        hipLaunchKernelGGL(k_scatter_list, grd, blk, 0, st, n, c->iflag.as<int>(), c->irank.as<int>(), c->ipos.as<unsigned>());
    }
    P = P_all;                                  // (one GPU: the two counts came with the prologue, k_prep counted them)
    if (part) {
        Collect q;
        q.n = 5;
        q.src[0] = c->irank.as<int>() + n; q.src[1] = c->ppos.as<int>() + (n - 1); q.src[2] = c->pflag.as<int>() + (n - 1);
        q.src[3] = c->ppos_own.as<int>() + (n - 1); q.src[4] = c->pown.as<int>() + (n - 1);
        for (int i = 0; i < 5; ++i) q.bytes[i] = 4;
        unsigned long long w[8];
        GSR_TRY(read_back(c, q, w));
        n_irr = (int)w[0];
        P_all = (int)w[1] + (int)w[2];          // parents among the local components (they are no candidates)
        P = (int)w[3] + (int)w[4];              // parents this rank evaluates
    }
    c->stats[0] = P;
    c->stats_ex[0] = n_irr;
    // the candidate stream of pass A (non-parents only) and its prefix table
    GSR_TRY(c->Ac.reserve(((size_t)(n - P_all) + SEL_PAD) * 16)); GSR_TRY(c->cellStartC.reserve(((size_t)gp.ncells + 1) * 4));
    const bool have_irr = part || n_irr > 0;
    if (have_irr) GSR_TRY(c->cellStartI.reserve(((size_t)gp.ncells + 1) * 4));
    hipLaunchKernelGGL(k_child_stream, grd, blk, 0, st, n, (int64_t)gp.ncells, P_all, c->A.as<float4>(), c->pflag.as<int>(), c->ppos.as<int>(),
                       c->cellStart.as<int>(), c->Ac.as<float4>(), c->cellStartC.as<int>(), (int)SEL_PAD,
                       have_irr ? c->irank.as<int>() : (const int*)nullptr, c->cellStartI.as<int>());
    GSR_CHECKPOINT("grid + gather");
    GSR_TIME(c->ev[1], st);
    return GSR_OK;
}

// ---- 2. selection ------------------------------------------------------------------------------
// Fast path (SPARSE): one evaluation pass.  k_spans sums the span lengths per parent (an upper bound of
// its child count); every parent writes its pairs at the head of a segment of that capacity.  The sparse buffers cost 8 bytes per
// candidate scanned; when that exceeds the budget (GSR_HEM_SPARSE_GB, default 45 % of free HBM) the two-pass COUNT + FILL fallback
// runs instead (same device code, evaluates every candidate twice).  An asynchronous level takes the buffers as they are.
int32_t LevelRun::select_phase() {
    Pm = (size_t)(P > 0 ? P : 1);
    GSR_TRY(c->pcnt.reserve(Pm * 4)); GSR_TRY(c->pcap.reserve(Pm * 4)); GSR_TRY(c->poff.reserve((Pm + 1) * 8));
    GSR_TRY(c->scratch.reserve((Pm * 2 + 128) * 8));
    sa.A = c->A.as<float4>(); sa.geo = c->geo.as<float4>();
    sa.Rs = c->Rs.as<float>(); sa.plist = c->plist.as<unsigned>(); sa.cellStart = c->cellStart.as<int>();
    sa.gp = c->gparams.as<GridParams>(); sa.P = P;
    sa.Ac = c->Ac.as<float4>(); sa.cellStartC = c->cellStartC.as<int>(); sa.cellStartI = c->cellStartI.as<int>();
    sa.irank = c->irank.as<int>(); sa.ipos = c->ipos.as<unsigned>(); sa.n_irr = n_irr; sa.ell = c->use_ell ? 1 : 0;
    // work sharding: rank r of W evaluates the contiguous run [P r / W, P (r+1) / W) of the cell-sorted parents
    own_lo = sharded ? (int)((int64_t)P * c->shard_rank / c->shard_world) : 0;
    own_hi = sharded ? (int)((int64_t)P * (c->shard_rank + 1) / c->shard_world) : P;
    sa.own_lo = own_lo; sa.own_hi = own_hi;
    {   // colour gate  sqrtf(x) > kappa^2/2  (mixture.cpp:123) as a test on x: sqrtf is correctly rounded and monotone, so the
        // gate is  x > x*,  x* = the largest float whose square root does not exceed the threshold
        const float t = c->kappa * c->kappa * 0.5f;
        float x;
        if (t != t) x = t;                               // NaN threshold: never rejects
        else if (t < 0.0f) x = -1.0f;                    // every distance exceeds it
        else if (t > FLT_MAX) x = t;                     // +inf: never rejects
        else {
            x = t * t;
            if (x > FLT_MAX) x = FLT_MAX;
            while (x < FLT_MAX && sqrtf(nextafterf(x, INFINITY)) <= t) x = nextafterf(x, INFINITY);
            while (x > 0.0f && sqrtf(x) > t) x = nextafterf(x, -INFINITY);
        }
        sa.colorThr2 = x;
    }
    sa.kldThr = c->delta * c->delta * 0.5f;       // mixture.cpp:128
    sa.tau2 = c->tau * c->tau;                    // mixture.cpp:58,61
    sa.pcnt = c->pcnt.as<unsigned>();
    sa.pcap = c->pcap.as<unsigned>();
    sa.part_p = reinterpret_cast<const unsigned*>(lvl + 2);
    sa.abort_p = cnt + 2;
    // parents per selection wave: SEL_NP, but no fewer waves than the chip holds at once (256 CUs x 28)
    sa.np = c->select_np > 0 ? c->select_np : (P >= SEL_NP * 7168 ? SEL_NP : (P >= 2 * 7168 ? 2 : 1));
    M = 0;
    constexpr int WPB = SEL_WPB;
    // (the queue-serving kernel runs beside the light parents' on the context's second stream: launch_select forks / joins by events)
#define GSR_LAUNCH_SELECT(MODE) GSR_TRY(launch_select(MODE, sa, st, c->aux, c->ev_fork, c->ev_join))
    c->sparse_path = false;
    if (P > 0) {
        GSR_TRY(c->prec.reserve(Pm * sizeof(ParentRec)));
        launch_parent_prep(st, P, c->plist.as<unsigned>(), c->geo.as<float4>(), c->Rs.as<float>(), sa.kldThr, sa.ell, c->prec.as<ParentRec>());
        sa.prec = c->prec.as<ParentRec>();
        if (c->use_rowlist && Pm * 2 * SEL_ROWS * sizeof(int2) <= c->rowlist_max) {
            GSR_TRY(c->rowlist.reserve(Pm * 2 * SEL_ROWS * sizeof(int2)));
            sa.rowlist = c->rowlist.as<int2>();
        }
        launch_spans(st, sa);                                                         // candidates scanned per parent
        GSR_TRY(c->coff.reserve((Pm + 1) * 8));
        GSR_TRY(widen_scan(c->pcap.as<unsigned>(), c->coff.as<int64_t>(), P));
        // what the pair buffers hold today: an asynchronous level runs on that (k_select clamps every segment to it)
        cap_pairs = (unsigned long long)(std::min(c->sp_child.cap, c->sp_wl.cap) / 4);
        if (!spec) {
            int64_t cand_i = 0;
            GSR_TRY(total_of(c->coff.as<int64_t>(), c->pcap.as<unsigned>(), P, &cand_i));
            cand = (unsigned long long)cand_i;
            c->stats[4] = (int64_t)cand;
        }
        // processing order: heavy parents first (LPT), the light ones along a Z-order curve
        GSR_TRY(c->porder.reserve(Pm * 4)); GSR_TRY(c->pkeys.reserve(Pm * 4)); GSR_TRY(c->pkeys2.reserve(Pm * 4)); GSR_TRY(c->pidx.reserve(Pm * 4));
        {
            // "heavy" = 16x the mean capacity: k_heavy_keys takes the total from the scan itself
            hipLaunchKernelGGL(k_heavy_keys, dim3(stride_grid(P)), blk, 0, st, P, c->pcap.as<unsigned>(), c->coff.as<int64_t>(), c->plist.as<unsigned>(),
                               c->A.as<float4>(), c->gparams.as<GridParams>(), c->pkeys.as<unsigned>(), c->pidx.as<unsigned>());
            GSR_TRY((sort_pairs<unsigned, order_cfg>(c, c->pkeys.as<unsigned>(), c->pkeys2.as<unsigned>(), c->pidx.as<unsigned>(), c->porder.as<unsigned>(), P, order_key_bits(P) + 2)));
            if (!c->split_heavy)
                hipLaunchKernelGGL(k_count_heavy, dim3(1), dim3(1), 0, st, P, c->pkeys2.as<unsigned>(), cnt + 8);
            sa.porder = c->porder.as<unsigned>();
            sa.xcd = 1;
            sa.nheavy = cnt + 8;
            // work items of the heavy parents: sum(ceil(cap / part)) <= candidates / part + P, part >= 2048 (k_heavy_items chooses it)
            if (c->split_heavy) {
                const unsigned long long cand_bound = spec ? cap_pairs : cand;
                const int max_items = (int)std::min<unsigned long long>(cand_bound / 2048ull + (unsigned long long)P + 1ull, 0x7fffffffull);
                GSR_TRY(c->hitem.reserve((size_t)max_items * sizeof(uint2))); GSR_TRY(c->hfirst.reserve(Pm * 4));
                GSR_TRY(c->part_cnt.reserve((size_t)max_items * 4));
                sa.heavy_blocks = SEL_HEAVY_BLOCKS;
                sa.hitem = c->hitem.as<uint2>(); sa.hfirst = c->hfirst.as<int>(); sa.part_cnt = c->part_cnt.as<unsigned>();
                sa.hq = cnt + 10;
                hipLaunchKernelGGL(k_heavy_items, dim3(1), dim3(1024), 0, st, P, c->pkeys2.as<unsigned>(), cnt + 8, sa.porder,
                                   c->pcap.as<unsigned>(), c->coff.as<int64_t>(), reinterpret_cast<unsigned*>(lvl + 2), own_lo, own_hi,
                                   SEL_HEAVY_BLOCKS * WPB, max_items, c->hitem.as<uint2>(), c->hfirst.as<int>(), c->pcnt.as<unsigned>(), sa.hq,
                                   cnt + 13);
            }
        }
        bool sparse = true;
        if (!spec) {
            // the budget question needs the driver (hipMemGetInfo: a system call per level) only when the buffers would have to grow
            const char* sparse_env = getenv("GSR_HEM_SPARSE_GB");
            sparse = cand < (1ull << 40);
            if (sparse && (sparse_env || (size_t)cand * 4 > c->sp_child.cap || (size_t)cand * 4 > c->sp_wl.cap)) {
                size_t free_b = 0, total_b = 0;
                (void)hipMemGetInfo(&free_b, &total_b);
                size_t budget = (free_b + c->sp_child.cap + c->sp_wl.cap) / 20 * 9;      // 45 % of what is free (a 40 M-splat level needs 79 GB)
                if (sparse_env) budget = (size_t)(atof(sparse_env) * 1073741824.0);
                sparse = (double)cand * 8.0 <= (double)budget;
            }
        }
        if (sparse) {
            if (!spec) {
                const size_t Cm = (size_t)(cand > 0 ? cand : 1);
                GSR_TRY(c->sp_child.reserve(Cm * 4)); GSR_TRY(c->sp_wl.reserve(Cm * 4));
                cap_pairs = (unsigned long long)(std::min(c->sp_child.cap, c->sp_wl.cap) / 4);
            }
            sa.cap_pairs = cap_pairs;
            sa.poff = c->coff.as<int64_t>(); sa.pair_child = c->sp_child.as<unsigned>(); sa.pair_wl = c->sp_wl.as<float>();
            GSR_TIME1(c->evk[2], st);
            GSR_CHECKPOINT("spans + ordering");
            GSR_LAUNCH_SELECT(SEL_SPARSE);
            GSR_CHECKPOINT("k_select<SPARSE>");
            GSR_TIME1(c->evk[3], st);
        } else {
            GSR_TIME1(c->evk[0], st);
            GSR_LAUNCH_SELECT(SEL_COUNT);
            GSR_TIME1(c->evk[1], st);
        }
        GSR_TRY(widen_scan(c->pcnt.as<unsigned>(), c->poff.as<int64_t>(), P));
        if (!spec) GSR_TRY(total_of(c->poff.as<int64_t>(), c->pcnt.as<unsigned>(), P, &M));
        // the pairs stay where the selection wrote them (segments of capacity pcap[p] at coff[p]): the partition pass and the
        // M-step walk the segments.  Only the parts of the split parents are slid together (in place) -- by a handful of workgroups
        // in a chain of dependent copies, BEFORE the SH copy is let loose beside it: under that kernel's 5 TB/s the chain took 0.13 ms
        // of the 5 M level's critical path.
        if ((spec || M > 0) && sparse && sa.heavy_blocks)
            hipLaunchKernelGGL(k_join_parts, dim3(1024), blk, 0, st, sa.nheavy, sa.porder, c->coff.as<int64_t>(), sa.hfirst, sa.part_cnt,
                               c->pcap.as<unsigned>(), sa.part_p, c->sp_child.as<unsigned>(), c->sp_wl.as<float>());
        // the cell-sorted copy of the SH block -- if the level's pair count says it pays (k_gather_sh) -- on the third stream, beside the
        // pair partition and the per-child sums; joined in front of the M-step
        if (!sh_launched) { GSR_TRY(launch_gather_sh(c->aux2 != nullptr)); sh_launched = true; }
        if (spec || M > 0) {
            if (!sparse) {
                const size_t Mm = (size_t)(M > 0 ? M : 1);
                GSR_TRY(c->pair_child.reserve(Mm * 4)); GSR_TRY(c->pair_wl.reserve(Mm * 4));
                sa.poff = c->poff.as<int64_t>(); sa.pair_child = c->pair_child.as<unsigned>(); sa.pair_wl = c->pair_wl.as<float>();
                if (sa.heavy_blocks)                         // the queue cursor back behind the statically assigned items
                    hipLaunchKernelGGL(k_fill_const<int>, dim3(1), dim3(1), 0, st, (int64_t)1, sa.hq + 1, (int)(SEL_HEAVY_BLOCKS * WPB));
                GSR_TIME1(c->evk[2], st);
                GSR_LAUNCH_SELECT(SEL_FILL);
                GSR_TIME1(c->evk[3], st);
            }
        }
        c->sparse_path = sparse;
        c->stats_ex[1] = sparse ? 1 : 0;
    } else if (spec) {
        return GSR_RETRY_SYNC;                  // (a level without a parent: nothing to speculate about)
    }
#undef GSR_LAUNCH_SELECT
    c->stats[1] = M;
    GSR_CHECKPOINT("selection");
    GSR_TIME(c->ev[2], st);
    return GSR_OK;
}

// a compact copy of the pair list: only the exact partition (histogram + scan) and the sort path read one
int32_t LevelRun::compact_pairs() {
    if (!c->sparse_path) return GSR_OK;
    const size_t Mm = (size_t)(M > 0 ? M : 1);
    GSR_TRY(c->pair_child.reserve(Mm * 4)); GSR_TRY(c->pair_wl.reserve(Mm * 4));
    hipLaunchKernelGGL(k_compact_pairs, dim3(ceil_div(P, 8)), blk, 0, st, P, c->coff.as<int64_t>(), c->pcnt.as<unsigned>(), c->poff.as<int64_t>(),
                       (const int*)nullptr, (const unsigned*)nullptr, c->pcap.as<unsigned>(), (const unsigned*)nullptr, c->sp_child.as<unsigned>(), c->sp_wl.as<float>(),
                       c->pair_child.as<unsigned>(), c->pair_wl.as<float>());
    return GSR_OK;
}
// fixed-capacity partition: bucket regions of `cap` pairs (6x the mean, 8x on small levels; an asynchronous level: what the buffers hold),
// straight from the segments
int32_t LevelRun::sums_fixed() {
    unsigned cap;
    if (spec) {
        const size_t have = std::min(c->spair_child.cap / sizeof(slot_t), c->spair_wl.cap / 4) / (size_t)nbuckets;
        if (have < 4096) return GSR_RETRY_SYNC;
        cap = (unsigned)std::min<size_t>(have, 0xfffff000u);
    } else {
        const double mean = (double)M / (double)nbuckets;
        double capd = mean * (c->partition_factor > 0.0 ? c->partition_factor : (M < (1 << 24) ? 8.0 : 6.0)) + (c->partition_factor > 0.0 ? 64.0 : 4096.0);
        if (capd > 4.0e9) return GSR_E_INVALID;                  // (not an error: the caller takes the exact path)
        cap = (unsigned)capd;
        GSR_TRY(c->spair_child.reserve((size_t)nbuckets * cap * sizeof(slot_t))); GSR_TRY(c->spair_wl.reserve((size_t)nbuckets * cap * 4));
    }
    GSR_TRY(c->bcursor.reserve(((size_t)nbuckets + 1) * 8));
    if (sharded) GSR_HIP(hipMemsetAsync(c->bcursor.p, 0, ((size_t)nbuckets + 1) * 4, st));      // (one GPU: the level's prologue cleared the cursors, k_bbox_reduce)
    (void)hipGetLastError();
    GSR_TIME(c->evm[2], st);
    launch_partition(st, c->partition_staged ? (c->partition_stage ? c->partition_stage : 1) : 0, P, seg, c->pcnt.as<unsigned>(), pc, pw, nbuckets, bshift, cap, c->bcursor.as<unsigned>(),
                     c->spair_child.as<slot_t>(), c->spair_wl.as<float>(), overflow_flag);
    GSR_HIP(hipGetLastError());
    GSR_TIME(c->evm[3], st);
    GSR_CHECKPOINT("pair partition (fixed capacity)");
    GSR_TIME(c->evm[4], st);
    hipLaunchKernelGGL(k_bucket_sum, dim3(nbuckets), dim3(bshift >= 10 ? 1024 : 256), (size_t)12 << bshift, st, n, bshift, (const unsigned long long*)nullptr, cap,
                       c->bcursor.as<unsigned>(), c->spair_child.as<slot_t>(), c->spair_wl.as<float>(), c->sumLw.as<float>(), c->oflag.as<int>(),
                       c->geo.as<float>() + 15);
    GSR_TIME(c->evm[5], st);
    GSR_HIP(hipGetLastError());
    fixed_tried = true;
    return GSR_OK;
}
int32_t LevelRun::sums_exact() {
    // histogram + scan + scatter of a compact pair list, then one workgroup per bucket sums in LDS on a fixed-point scale.  The
    // partition kernels keep per-bucket counters in dynamic LDS (raised above the 64 KiB default in gsr_hem_create)
    const int tile = SUM_TILE;      // (x4 on levels with > 2000 buckets: measured, no gain)
    const int ntiles = (int)((M + tile - 1) / tile);
    GSR_TRY(compact_pairs());
    const size_t Mm = (size_t)(M > 0 ? M : 1);
    GSR_TRY(c->spair_child.reserve(Mm * 4)); GSR_TRY(c->spair_wl.reserve(Mm * 4));
    GSR_TRY(c->bhist.reserve(((size_t)nbuckets + 1) * 4)); GSR_TRY(c->bstart.reserve(((size_t)nbuckets + 1) * 8));
    GSR_TRY(c->bcursor.reserve(((size_t)nbuckets + 1) * 8));
    GSR_HIP(hipMemsetAsync(c->bhist.p, 0, ((size_t)nbuckets + 1) * 4, st));
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_bucket_hist, dim3(ntiles), blk, (size_t)nbuckets * 4, st, M, tile, c->pair_child.as<unsigned>(), nbuckets, bshift, c->bhist.as<unsigned>());
    GSR_HIP(hipGetLastError());
    GSR_TRY(widen_scan(c->bhist.as<unsigned>(), (int64_t*)c->bstart.p, nbuckets + 1));
    GSR_HIP(hipMemcpyAsync(c->bcursor.p, c->bstart.p, ((size_t)nbuckets + 1) * 8, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(k_bucket_scatter, dim3(ntiles), blk, (size_t)nbuckets * 4, st, M, tile, c->pair_child.as<unsigned>(),
                       c->pair_wl.as<float>(), nbuckets, bshift, c->bstart.as<unsigned long long>(), c->bcursor.as<unsigned long long>(),
                       c->spair_child.as<slot_t>(), c->spair_wl.as<float>());
    GSR_HIP(hipGetLastError());
    GSR_CHECKPOINT("pair partition");
    hipLaunchKernelGGL(k_bucket_sum, dim3(nbuckets), dim3(bshift >= 10 ? 1024 : 256), (size_t)12 << bshift, st, n, bshift, c->bstart.as<unsigned long long>(), 0u,
                       (const unsigned*)nullptr, c->spair_child.as<slot_t>(), c->spair_wl.as<float>(), c->sumLw.as<float>(), c->oflag.as<int>(),
                       c->geo.as<float>() + 15);
    GSR_HIP(hipGetLastError());
    return GSR_OK;
}

// ---- 3. per-child sums of wL (deterministic whatever the order: LDS fixed point, k_bucket_sum) ----------
// The pairs of parent p are the run [seg[p], seg[p] + pcnt[p]) of (pc, pw): the sparse segments of the one-pass selection or
// the compact CSR of the two-pass fallback.
int32_t LevelRun::sums_phase() {
    seg = c->sparse_path ? c->coff.as<int64_t>() : c->poff.as<int64_t>();
    pc = c->sparse_path ? c->sp_child.as<unsigned>() : c->pair_child.as<unsigned>();
    pw = c->sparse_path ? c->sp_wl.as<float>() : c->pair_wl.as<float>();
    GSR_TRY(c->cstart.reserve(((size_t)n + 1) * 8)); GSR_TRY(c->sumLw.reserve(n * 4)); GSR_TRY(c->oflag.reserve(n * 4));
    // children per bucket: SUM_BUCKET on large levels; on small ones fewer, so that the bucket kernel still has ~2 workgroups per CU
    bshift = SUM_BUCKET_SHIFT;
    while (bshift > 6 && (n >> bshift) < 512) --bshift;
    while (bshift < 13 && (n >> bshift) > 2500) ++bshift;       // very large levels: too many buckets scatter the partition's writes
    nbuckets = (int)((n + (1 << bshift) - 1) >> bshift);
    if (spec) {
        if (nbuckets > SUM_MAX_BUCKETS) return GSR_RETRY_SYNC;
        GSR_TRY(sums_fixed());
    } else if (part) {
        GSR_TRY(pl.sums(n, P, M, seg, pc, pw, nbuckets, bshift, overflow_flag));
    } else if (c->sum_bucket && M > 0 && nbuckets <= SUM_MAX_BUCKETS) {
        if (c->partition_fixed && !c->partition_overflowed) {
            const int32_t r = sums_fixed();
            if (r == GSR_E_INVALID && !fixed_tried) GSR_TRY(sums_exact()); else GSR_TRY(r);
        } else {
            if (c->partition_overflowed) c->stats_ex[2] = 1;
            GSR_TRY(sums_exact());
        }
    } else {
        if (M > 0) {
            GSR_TRY(compact_pairs());
            const size_t Mm = (size_t)M;
            GSR_TRY(c->spair_child.reserve(Mm * 4)); GSR_TRY(c->spair_wl.reserve(Mm * 4));
            GSR_TRY(sort_pairs<float>(c, c->pair_child.as<unsigned>(), c->spair_child.as<unsigned>(), c->pair_wl.as<float>(),
                                      c->spair_wl.as<float>(), M, bits_for(n)));
            GSR_CHECKPOINT("pair sort");
            hipLaunchKernelGGL(k_run_starts<int64_t>, dim3(stride_grid(M)), blk, 0, st, M, c->spair_child.as<unsigned>(), n, c->cstart.as<int64_t>());
        } else {
            hipLaunchKernelGGL(k_fill_const<int64_t>, grd, blk, 0, st, n + 1, c->cstart.as<int64_t>(), (int64_t)0);
        }
        GSR_TRY(c->spair_wl.reserve(4));
        hipLaunchKernelGGL(k_sumlw, dim3(stride_grid(n * 8)), blk, 0, st, n, c->cstart.as<int64_t>(), c->spair_wl.as<float>(), c->sumLw.as<float>(), c->oflag.as<int>(), c->geo.as<float>() + 15);
    }
    if (sharded) {
        // exchange 1: every rank holds the sums over ITS parents; the total decides responsibilities and orphans
        GSR_HIP(hipStreamSynchronize(st));
        if (c->shard_allreduce(c->sumLw.p, n, c->shard_user) != 0) return fail(GSR_E_INVALID, "gsr_hem_run_level: all-reduce callback failed (sumLw)");
        hipLaunchKernelGGL(k_orphan_flags, grd, blk, 0, st, n, c->sumLw.as<float>(), c->oflag.as<int>(), c->geo.as<float>() + 15);
    }
    GSR_CHECKPOINT("per-child sums");
    GSR_TIME(c->ev[3], st);
    return GSR_OK;
}

// ---- 4a. output ranks in input order: the orphans' flags back to input order, their scan, the level's row count ----------
int32_t LevelRun::output_ranks() {
    GSR_TRY(c->pflag_in.reserve(n * 4)); GSR_TRY(c->oflag_in.reserve(n * 4)); GSR_TRY(c->prank_in.reserve(n * 4)); GSR_TRY(c->orank_in.reserve(n * 4));
    int o_last = 0, o_flag = 0;
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (ranks_forked && attempt == 0) {
            GSR_HIP(hipStreamWaitEvent(st, c->ev_pre, 0));
        } else {
            hipLaunchKernelGGL(k_flags_in, grd, blk, 0, st, n, n_own, L.is_parent.as<uint8_t>(), c->pflag_in.as<int>(), c->oflag_in.as<int>());
            GSR_TRY(exclusive_scan<int>(c, c->pflag_in.as<int>(), c->prank_in.as<int>(), n));
        }
        hipLaunchKernelGGL(k_orphans_to_input_order, grd, blk, 0, st, n, c->order.as<unsigned>(), c->oflag.as<int>(), c->oflag_in.as<int>());
        GSR_TRY(exclusive_scan<int>(c, c->oflag_in.as<int>(), c->orank_in.as<int>(), n));
        if (spec) return GSR_OK;                    // (the counts stay on the device: k_level_tail, once the output's capacity is known)
        Collect q;
        q.n = 6;
        q.src[0] = c->orank_in.as<int>() + (n - 1); q.src[1] = c->oflag_in.as<int>() + (n - 1); q.src[2] = overflow_flag;
        q.src[3] = cnt + 13;
        q.src[4] = cnt + 8; q.src[5] = cnt + 10;      // heavy parents, their work items (statistics)
        for (int i = 0; i < 6; ++i) q.bytes[i] = 4;
        unsigned long long w[8];
        GSR_TRY(read_back(c, q, w));
        o_last = (int)w[0]; o_flag = (int)w[1];
        if (P > 0 && M >= 0 && c->stats[4] > 0) { c->stats_ex[3] = (int64_t)(unsigned)w[4]; c->stats_ex[4] = c->split_heavy ? (int64_t)(unsigned)w[5] : 0; }
        if (w[3] != 0ull) return fail(GSR_E_INVALID, "gsr_hem_run_level: the work-item table of the heavy parents overflowed (rerun with GSR_HEM_SPLIT=0)");
        if (!(fixed_tried && w[2] != 0ull) || attempt == 1) break;
        // a bucket region overflowed (pairs far more clustered than 6x the mean): the sums are incomplete.  Exact partition, and
        // this context stays with it
        c->partition_overflowed = true;
        c->stats_ex[2] = 1;
        GSR_HIP(hipMemsetAsync(overflow_flag, 0, 4, st));
        GSR_TRY(sums_exact());
        if (sharded) return fail(GSR_E_INVALID, "gsr_hem_run_level: bucket overflow on a sharded level");     // (sharded levels use the exact partition)
    }
    n_orph = (int64_t)o_last + o_flag;
    c->stats[2] = n_orph;
    n_pre = (int64_t)P + n_orph;
    return GSR_OK;
}

// ---- 4b. where the new level goes: the context's own arrays or the caller's (gsr_hem_set_output) ----------
int32_t LevelRun::open_output() {
    // rows the arrays must hold: n_pre -- or, for an asynchronous level, which does not know n_pre yet, the size of the level being reduced
    // (parents + orphans exceed it only when parents are orphans too: k_level_tail checks, the level then reruns)
    const int64_t need = spec ? n : n_pre;
    if (c->out_pending) {
        c->out_pending = false;                                 // one request, one level
        if (spec && P > c->out_rows) return GSR_RETRY_SYNC;     // (the synchronous level reports it)
        if (!spec && n_pre > c->out_rows) return fail(GSR_E_INVALID, "gsr_hem_run_level: the output arrays hold %lld rows, the level has %lld", (long long)c->out_rows, (long long)n_pre);
        if (F > 0 && !c->out_ptr[4]) return fail(GSR_E_INVALID, "gsr_hem_run_level: the output has no SH array but the level has F = %d", F);
        if (need > 0) {
            for (int i = 0; i < 5; ++i) {
                DevBuf* b = level_big(O, i);
                b->swap(c->spare_out[i]);
                b->p = c->out_ptr[i] ? c->out_ptr[i] : c->out_ptr[0]; b->cap = (size_t)-1;       // reserve() on them is a no-op
            }
            out_active = true;
        }
    }
    GSR_TRY(O.reserve(need, F));
    out_cap = out_active ? std::min<int64_t>(c->out_rows, n + (int64_t)P) : need;     // (parents + orphans <= n + P)
    // an asynchronous level: O.weight / is_parent are the library's own arrays, reserved for `need` = n rows -- no row beyond them may be
    // written (a degenerate level whose parents are orphans too has more than n rows: k_level_tail raises the flag, the level reruns)
    if (spec) out_cap = std::min(out_cap, need);
    O.n = spec ? 0 : n_pre; O.F = F;
    if (spec)
        hipLaunchKernelGGL(k_level_tail, dim3(1), dim3(1), 0, st, n, P, c->orank_in.as<int>(), c->oflag_in.as<int>(), (long long)out_cap, lvl, cnt + 2);
    return GSR_OK;
}

// ---- 4c. M-step; orphans -----------------------------------------
int32_t LevelRun::mstep_phase() {
    const long long* n_pre_dev = spec ? lvl : nullptr;
    const int* sh_mode = reinterpret_cast<const int*>(lvl + 3);                     // where a child's SH row is read from: k_gather_sh decided
    // the new level's parent flags depend on nothing but the stream position and n_pre: drawn on the second stream beside the
    // M-step (the jump to the stream position is a fixed ~40 us chain, 5 % of a 200 k-splat level)
    rng_pos0 = c->rng_pos;
    const int64_t n_draw = spec ? out_cap : n_pre;
    if (!part && c->aux && n_draw > 0) {
        GSR_HIP(hipEventRecord(c->ev_fork, st)); GSR_HIP(hipStreamWaitEvent(c->aux, c->ev_fork, 0));
        GSR_TRY(O.is_parent.reserve((size_t)n_draw));
        GSR_TRY(draw_flags_raw(c, n_draw, O.is_parent.as<uint8_t>(), c->aux, n_pre_dev));
        GSR_HIP(hipEventRecord(c->ev_join, c->aux));
        flags_forked = true;
    }
    if (!sh_launched) { GSR_TRY(launch_gather_sh(false)); sh_launched = true; }                       // (a level without parents)
    if (sh_pending) { GSR_HIP(hipStreamWaitEvent(st, c->ev_sh_join, 0)); sh_pending = false; }      // the sorted SH rows are needed from here on
    if (P > 0) {
        MstepArgs ma;
        memset(&ma, 0, sizeof(ma));
        ma.geo = c->geo.as<float4>(); ma.RSH = RSH;
        ma.sh_mode_p = sh_mode; ma.sh_own = L.sh.as<float>(); ma.sh_sorted = c->shs.as<float>();
        ma.sh_last = part ? 0xffffffffu : (unsigned)(n - 1); ma.sh_tail = c->sh_tail.as<float>();
        ma.pair_child = pc; ma.pair_wl = pw;
        ma.P = P; ma.F = F;
        ma.small = c->mstep_small ? 1 : 0;
        // processing order: the selection's (heavy parents by candidates scanned first, then Z-order) -- parents with many
        // candidates are the ones with many pairs.  The per-parent headers are laid out in that order.
        const unsigned* mporder = (spec || M > 0) ? c->porder.as<unsigned>() : nullptr;
        ma.xcd = mporder ? 1 : 0;
        ma.nheavy = mporder ? cnt + 8 : nullptr;
        GSR_TRY(c->mhdr.reserve(Pm * sizeof(MstepHeader)));
        // heavy parents (more than MSTEP_SEG pairs): their segments are work items of a second launch (mstep_segment).  An asynchronous
        // level does not know whether there is one: the launch is always made (beside k_mstep, on the second stream) and finds its list empty
        const bool msplit = spec || M > MSTEP_SEG;              // (a parent of more than MSTEP_SEG pairs can exist)
        if (msplit) {
            const size_t Mb = (size_t)(spec ? cap_pairs : (unsigned long long)M);      // pairs <= candidates <= what the buffers hold
            const size_t cap_heavy = Mb / MSTEP_SEG + 2, cap_items = 2 * (Mb / MSTEP_SEG) + 4;
            GSR_TRY(c->mh_list.reserve(cap_heavy * sizeof(uint4))); GSR_TRY(c->mh_items.reserve(cap_items * sizeof(uint2)));
            GSR_TRY(c->mh_scratch.reserve(cap_items * (size_t)(16 + RSH) * 4));
        }
        unsigned* hcount = (unsigned*)cnt;                       // [0] heavy parents, [1] their segments (cleared with the level's counters)
        hipLaunchKernelGGL(k_mstep_headers, dim3(stride_grid(P)), blk, 0, st, P, mporder, c->plist.as<unsigned>(), seg,
                           c->pcnt.as<unsigned>(), c->order.as<unsigned>(), c->prank_in.as<int>(), c->A.as<float4>(), own_lo, own_hi,
                           c->mhdr.as<MstepHeader>(), (unsigned*)cnt + 15, (unsigned)MSTEP_SEG, msplit ? hcount : (unsigned*)nullptr,
                           c->mh_list.as<uint4>(), c->mh_items.as<uint2>());
        ma.hdr = c->mhdr.as<MstepHeader>();
        ma.split = c->mstep_split ? 1 : 0;
        ma.hcount = hcount; ma.hlist = c->mh_list.as<uint4>(); ma.hitems = c->mh_items.as<uint2>(); ma.hscratch = c->mh_scratch.as<float>();
        ma.o_xyz = O.xyz.as<float>(); ma.o_color = O.color.as<float>(); ma.o_cov6 = O.cov6.as<float>();
        ma.o_opacity = O.opacity.as<float>(); ma.o_weight = O.weight.as<float>(); ma.o_sh = O.sh.as<float>();
        // one wavefront per workgroup: consecutive parents on one CU share no cache lines in time (2 / 4 / 8 waves per workgroup
        // were measured 1 / 10 / 24 % slower)
        const int nq = RSH >> 2;                                // float4 per SH row; a lane covers MSTEP_NV of them
        // the heavy parents' segments: 2 048 waves pull them from the item table, on the second stream beside k_mstep when there is
        // one (the finish kernel adds a parent's segments in order); joined behind k_mstep
        hipStream_t hst = st;
        if (msplit && c->aux) {
            GSR_HIP(hipEventRecord(c->ev_mfork, st)); GSR_HIP(hipStreamWaitEvent(c->aux, c->ev_mfork, 0));
            hst = c->aux;
        }
#define GSR_LAUNCH_MSTEP(G)                                                                                              \
    {                                                                                                                    \
        if (msplit && hst != st) {                                                                                       \
            hipLaunchKernelGGL((k_mstep<G, 1, true>), dim3(2048), dim3(64), 0, hst, ma);                                 \
            hipLaunchKernelGGL(k_mstep_heavy_finish, dim3(256), dim3(64), 0, hst, ma);                                   \
            GSR_HIP(hipEventRecord(c->ev_mjoin, hst));                                                                   \
        }                                                                                                                \
        hipLaunchKernelGGL((k_mstep<G, 1>), dim3(8 * ceil_div(ceil_div(P, MSTEP_K), 8)), dim3(64), 0, st, ma);           \
        if (msplit && hst == st) {                                                                                       \
            hipLaunchKernelGGL((k_mstep<G, 1, true>), dim3(2048), dim3(64), 0, st, ma);                                  \
            hipLaunchKernelGGL(k_mstep_heavy_finish, dim3(256), dim3(64), 0, st, ma);                                    \
        }                                                                                                                \
    }
        GSR_TIME1(c->evm[0], st);
        if (nq == 0) { GSR_LAUNCH_MSTEP(0) }
        else if (nq <= 1 * MSTEP_NV) { GSR_LAUNCH_MSTEP(1) }
        else if (nq <= 2 * MSTEP_NV) { GSR_LAUNCH_MSTEP(2) }
        else if (nq <= 4 * MSTEP_NV) { GSR_LAUNCH_MSTEP(4) }
        else if (nq <= 8 * MSTEP_NV) { GSR_LAUNCH_MSTEP(8) }
        else if (nq <= 16 * MSTEP_NV) { GSR_LAUNCH_MSTEP(16) }
        else { GSR_LAUNCH_MSTEP(32) }      // F <= 384
#undef GSR_LAUNCH_MSTEP
        if (msplit && hst != st) GSR_HIP(hipStreamWaitEvent(st, c->ev_mjoin, 0));
        GSR_TIME1(c->evm[1], st);
    }
    hipLaunchKernelGGL(k_orphan_rows, dim3(stride_grid(n_own)), blk, 0, st, n_own, P, c->oflag_in.as<int>(), c->orank_in.as<int>(), L.xyz.as<float>(), L.color.as<float>(),
                       L.cov6.as<float>(), L.opacity.as<float>(), L.weight.as<float>(), L.sh.as<float>(), F, O.xyz.as<float>(), O.color.as<float>(),
                       O.cov6.as<float>(), O.opacity.as<float>(), O.weight.as<float>(), O.sh.as<float>(), (int64_t)(spec ? out_cap : n_pre));
    if (sharded && P > 0) {
        // exchange 2: the merged components.  Every rank packs the rows of ITS parents, ONE all-gather of equal chunks
        // (ceil(P / world) rows of 14 + F floats) moves them, and every rank scatters every chunk into the output rows
        // of the owners' slots: P (14 + F) 4 bytes per rank received in all, no floating-point arithmetic in the exchange.
        const int W = c->shard_world, RW = 14 + F;
        const int chunk = (P + W - 1) / W + 1;                    // rows per rank (the ranges differ by at most one)
        const size_t chunk_bytes = (size_t)chunk * RW * 4;
        GSR_TRY(c->shard_send.reserve(chunk_bytes)); GSR_TRY(c->shard_recv.reserve(chunk_bytes * W));
        const int own_cnt = own_hi - own_lo;
        if (own_cnt > 0)
            hipLaunchKernelGGL((k_shard_rows<true>), dim3(stride_grid((int64_t)own_cnt * RW)), blk, 0, st, own_lo, own_cnt, F, c->plist.as<unsigned>(),
                               c->order.as<unsigned>(), c->prank_in.as<int>(), c->shard_send.as<float>(), O.xyz.as<float>(), O.color.as<float>(),
                               O.cov6.as<float>(), O.opacity.as<float>(), O.weight.as<float>(), O.sh.as<float>());
        GSR_HIP(hipStreamSynchronize(st));
        if (c->shard_allgather(c->shard_send.p, c->shard_recv.p, (int64_t)chunk_bytes, c->shard_user) != 0)
            return fail(GSR_E_INVALID, "gsr_hem_run_level: all-gather callback failed (merged components)");
        for (int r = 0; r < W; ++r) {
            if (r == c->shard_rank) continue;
            const int lo = (int)((int64_t)P * r / W), hi = (int)((int64_t)P * (r + 1) / W);
            if (hi > lo)
                hipLaunchKernelGGL((k_shard_rows<false>), dim3(stride_grid((int64_t)(hi - lo) * RW)), blk, 0, st, lo, hi - lo, F, c->plist.as<unsigned>(),
                                   c->order.as<unsigned>(), c->prank_in.as<int>(), c->shard_recv.as<float>() + (size_t)r * chunk * RW, O.xyz.as<float>(),
                                   O.color.as<float>(), O.cov6.as<float>(), O.opacity.as<float>(), O.weight.as<float>(), O.sh.as<float>());
        }
    }
    P_glob = P; O_glob = n_orph;
    if (part) GSR_TRY(pl.global_ranks(P, n_pre, P_glob, O_glob));
    n_pre_glob = P_glob + O_glob;
    GSR_CHECKPOINT("M-step + orphans");
    GSR_TIME(c->ev[4], st);
    return GSR_OK;
}

// The validity erase in place, on the device (k_erase_save + k_erase_shift): lvl[0] = rows before it, cnt[3] / c->holes = what k_valid found;
// lvl[4] = rows after it.  Both kernels leave at once when there is nothing to erase -- or more than ERASE_MAX rows (the host's path then).
int32_t LevelRun::erase_in_place(int64_t rows_bound, bool may_allocate, int64_t alloc_rows) {
    // (the saved tails need 32 rows per tile of the bound -- 150 MB at 5 M rows -- and most clouds never erase a row: the buffer is allocated
    // the first time a level of this context does, by the host's path; until then an asynchronous level only COUNTS the erased rows.
    // alloc_rows: that first allocation covers the bound the ASYNCHRONOUS levels of this size will ask for -- the level's input size, not the
    // n_pre the host's path knows (sized for n_pre, every later asynchronous level found the buffer too small and took the
    // host's path again, a second round trip per surfel level))
    const int tile0 = std::min(256, 12288 / std::max(F, 6));
    const auto halo_bytes = [&](int64_t rows) { return (size_t)((rows + tile0 - 1) / tile0 + 1) * ERASE_MAX * ((size_t)(14 + F) * 4 + 1) + 64; };
    {
        const size_t need = halo_bytes(rows_bound);
        if (!may_allocate && c->erase_halo.cap < need) {
            hipLaunchKernelGGL(k_fill_const<long long>, dim3(1), dim3(1), 0, st, (int64_t)1, lvl + 4, (long long)0);     // (unused: the prologue reads lvl[0])
            return GSR_OK;
        }
    }
    erase_on_device = true;
    EraseArgs ea;
    memset(&ea, 0, sizeof(ea));
    float* arrs[6] = {O.xyz.as<float>(), O.color.as<float>(), O.cov6.as<float>(), O.opacity.as<float>(), O.weight.as<float>(), O.sh.as<float>()};
    const int ws[6] = {3, 3, 6, 1, 1, F};
    for (int q = 0; q < 6; ++q) { ea.arr[q] = arrs[q]; ea.w[q] = ws[q]; }
    ea.flags = O.is_parent.as<uint8_t>();
    ea.W = 14 + F;
    ea.tile = std::min(256, 12288 / std::max(F, 6));
    ea.max_tiles = (int)((rows_bound + ea.tile - 1) / ea.tile) + 1;
    GSR_TRY(c->erase_halo.reserve(halo_bytes(std::max(rows_bound, alloc_rows))));
    ea.halo = c->erase_halo.as<float>();
    ea.holes = c->holes.as<int>(); ea.dropped = cnt + 3; ea.n_pre_p = lvl; ea.n_keep_out = lvl + 4;
    const int g = std::min(ea.max_tiles, 2048);
    hipLaunchKernelGGL(k_erase_save, dim3(g), blk, 0, st, ea);
    hipLaunchKernelGGL(k_erase_shift, dim3(g), blk, (size_t)12288 * 4, st, ea);
    GSR_HIP(hipGetLastError());
    return GSR_OK;
}

// the validity erase of a level that drops MANY rows (mixture.cpp:262-282): the surviving rows of every array move up, in order, through a second buffer
int32_t LevelRun::compact_erased(int64_t n_keep) {
    Level& T = c->tmp;
    const dim3 g2(stride_grid(n_pre));
    GSR_TRY(T.reserve(n_keep, F));
    T.n = n_keep; T.F = F;
    const int* keep = c->keep.as<int>();
    const int* pos = c->kpos.as<int>();
    hipLaunchKernelGGL(k_compact_rows, dim3(stride_grid(n_pre * 3)), blk, 0, st, n_pre, 3, keep, pos, O.xyz.as<float>(), T.xyz.as<float>());
    hipLaunchKernelGGL(k_compact_rows, dim3(stride_grid(n_pre * 3)), blk, 0, st, n_pre, 3, keep, pos, O.color.as<float>(), T.color.as<float>());
    hipLaunchKernelGGL(k_compact_rows, dim3(stride_grid(n_pre * 6)), blk, 0, st, n_pre, 6, keep, pos, O.cov6.as<float>(), T.cov6.as<float>());
    hipLaunchKernelGGL(k_compact_rows, g2, blk, 0, st, n_pre, 1, keep, pos, O.opacity.as<float>(), T.opacity.as<float>());
    hipLaunchKernelGGL(k_compact_rows, g2, blk, 0, st, n_pre, 1, keep, pos, O.weight.as<float>(), T.weight.as<float>());
    if (F > 0)
        hipLaunchKernelGGL(k_compact_rows4, dim3(stride_grid(n_pre * ((F + 3) / 4))), blk, 0, st, n_pre, F, keep, pos, O.sh.as<float>(), T.sh.as<float>());
    hipLaunchKernelGGL(k_compact_bytes, g2, blk, 0, st, n_pre, keep, pos, O.is_parent.as<uint8_t>(), T.is_parent.as<uint8_t>());
    if (out_active) {     // the caller's arrays stay the level's home: the compacted rows are copied back into them
        const size_t row_bytes[5] = {12, 12, 24, 4, (size_t)F * 4};
        for (int i = 0; i < 5; ++i)
            if (row_bytes[i] > 0 && n_keep > 0)
                GSR_HIP(hipMemcpyAsync(level_big(O, i)->p, level_big(T, i)->p, (size_t)n_keep * row_bytes[i], hipMemcpyDeviceToDevice, st));
        O.weight.swap(T.weight); O.is_parent.swap(T.is_parent);
        O.n = n_keep;
    } else
        O.swap(T);
    return GSR_OK;
}

// ---- 5. new parent flags (one draw per component, before the erase), validity erase (synchronous schedule) ---------
int32_t LevelRun::flags_and_validity() {
    if (part) {     // the libc stream is drawn for the GLOBAL level; a row takes the flag at its global rank
        GSR_TRY(c->allflags.reserve((size_t)(n_pre_glob > 0 ? n_pre_glob : 1)));
        GSR_TRY(draw_flags_raw(c, n_pre_glob, c->allflags.as<uint8_t>()));
        if (n_pre > 0)
            hipLaunchKernelGGL(k_gather_bytes, dim3(stride_grid(n_pre)), blk, 0, st, n_pre, c->gid_next.as<unsigned>(), c->allflags.as<uint8_t>(), O.is_parent.as<uint8_t>());
    } else if (flags_forked) {
        GSR_HIP(hipStreamWaitEvent(st, c->ev_join, 0));
    } else {
        GSR_TRY(draw_flags(c, O));
    }
    dropped = 0;
    if (n_pre > 0) {
        GSR_TRY(c->keep.reserve(n_pre * 4)); GSR_TRY(c->kpos.reserve(n_pre * 4));
        const dim3 g2(stride_grid(n_pre));
        GSR_TRY(c->holes.reserve(ERASE_MAX * 4));
        hipLaunchKernelGGL(k_valid, g2, blk, 0, st, n_pre, O.xyz.as<float>(), O.cov6.as<float>(), c->keep.as<int>(), (const long long*)nullptr, cnt + 3, c->holes.as<int>());
        if (part) GSR_TRY(exclusive_scan<int>(c, c->keep.as<int>(), c->kpos.as<int>(), n_pre));        // (drop_erased renumbers with it)
        {
            Collect q;
            q.n = 2;
            q.src[0] = cnt + 3; q.src[1] = cnt + 15;
            q.bytes[0] = q.bytes[1] = 4;
            unsigned long long w[8];
            GSR_TRY(read_back(c, q, w));
            dropped = (int64_t)(unsigned)w[0];
            if (P > 0) c->stats_ex[5] = (int64_t)(unsigned)w[1];
        }
        if (dropped > 0 && dropped <= ERASE_MAX && !part) {         // a handful of rows: in place, on the device
            hipLaunchKernelGGL(k_fill_const<long long>, dim3(1), dim3(1), 0, st, (int64_t)1, lvl, (long long)n_pre);
            GSR_TRY(erase_in_place(n_pre, true, n));
            O.n = n_pre - dropped;
        } else if (dropped > 0) {
            if (!part) GSR_TRY(exclusive_scan<int>(c, c->keep.as<int>(), c->kpos.as<int>(), n_pre));
            GSR_TRY(compact_erased(n_pre - dropped));
        }
    }
    n_glob_next = n_pre_glob;
    if (part) GSR_TRY(pl.drop_erased(O, n_pre, n_pre_glob, dropped, n_glob_next));
    GSR_CHECKPOINT("flags + validity");
    return GSR_OK;
}

int32_t LevelRun::run(int64_t* n_out, int64_t* n_dropped) {
    memset(c->stats, 0, sizeof(c->stats));
    memset(c->stats_ex, 0, sizeof(c->stats_ex));
    memset(c->part_stats, 0, sizeof(c->part_stats));
    c->stats[6] = n;
    if (part) {
        // the rank-local preconditions of a partitioned level are AGREED ON before its first data collective (one all-reduce of a status
        // word): a rank that returned by itself would leave its peers in the next collective for ever
        pl.W = gsr_comm_world(c->comm); pl.me = gsr_comm_rank(c->comm);
        const unsigned code = c->shard_world > 1 ? 4u : (pl.W > 8 ? 3u : (n >= (1ll << 30) ? 2u : 0u));
        GSR_TRY(pl.agree_on_preconditions(n, code));
    }
    if (n >= (1ll << 30)) return fail(GSR_E_INVALID, "gsr_hem_run_level: %lld components (the candidate records carry the sorted position in 30 bits)", (long long)n);
    if (n == 0) {
        if (n_out) *n_out = 0;
        if (n_dropped) *n_dropped = 0;
        return GSR_OK;
    }
    // An asynchronous level: one GPU, the default path of every stage, and buffers to run on (a fresh context's first level sizes them the
    // synchronous way).  What it cannot know beforehand it checks on the device (GSR_RETRY_SYNC).
    spec = spec && !part && !sharded && !dbg_sync && c->sum_bucket && c->partition_fixed && !c->partition_overflowed && c->partition_factor == 0.0 &&
           c->split_heavy && getenv("GSR_HEM_SPARSE_GB") == nullptr && c->aux != nullptr &&
           std::min(c->sp_child.cap, c->sp_wl.cap) >= (size_t)4096 && std::min(c->spair_child.cap / sizeof(slot_t), c->spair_wl.cap / 4) >= (size_t)4096;
    grd = dim3(stride_grid(n));
    GSR_TIME1(c->ev[0], st);
    GSR_TRY(grid_phase());
    GSR_TRY(select_phase());
    GSR_TRY(sums_phase());
    GSR_TRY(output_ranks());
    GSR_TRY(open_output());
    GSR_TRY(mstep_phase());

    unsigned long long w[LC_WORDS];
    bool have_next = false;             // w holds the next level's prologue
    const bool want_pro = !c->last_level;       // (the last level of a gsr_hem_run_levels hierarchy: nobody will read it)
    const int cb_next = c->cblock ^ 1;
    LevelCollect q;
    memset(&q, 0, sizeof(q));
    q.cnt = cnt; q.gp = c->gparams.as<GridParams>(); q.bbox = c->bbox.as<unsigned>();
    if (P > 0) { q.coff = c->coff.as<int64_t>(); q.pcap = c->pcap.as<unsigned>(); q.poff = c->poff.as<int64_t>(); q.pcnt = c->pcnt.as<unsigned>(); q.P = P; }
    if (spec) {
        // ---- 5 (asynchronous). flags joined, validity counted, the NEXT level's prologue on the rows as they are, ONE answer ----------
        if (flags_forked) GSR_HIP(hipStreamWaitEvent(st, c->ev_join, 0));
        GSR_TRY(c->keep.reserve((size_t)out_cap * 4)); GSR_TRY(c->kpos.reserve((size_t)out_cap * 4));
        GSR_TRY(c->holes.reserve(ERASE_MAX * 4));
        hipLaunchKernelGGL(k_valid, dim3(stride_grid(out_cap)), blk, 0, st, out_cap, O.xyz.as<float>(), O.cov6.as<float>(), c->keep.as<int>(), (const long long*)lvl, cnt + 3,
                           c->holes.as<int>());
        GSR_TRY(erase_in_place(out_cap, false));        // (leaves at once when nothing is erased; lvl[4] = the rows that remain)
        GSR_TIME1(c->ev[5], st);
        if (want_pro) GSR_TRY(enqueue_prologue(c, O, out_cap, erase_on_device ? lvl + 4 : lvl, cb_next));
        q.lvl = lvl;
        GSR_TRY(read_back_level(c, q, w));
        if (w[LC_FLAGS] != 0ull) return GSR_RETRY_SYNC;        // a clamped segment, a full bucket region or item table, an output too small
        cand = w[LC_CAND]; M = (int64_t)w[LC_PAIRS]; n_orph = (int64_t)w[LC_ORPHANS]; n_pre = (int64_t)w[LC_NPRE];
        c->stats[4] = (int64_t)cand; c->stats[1] = M; c->stats[2] = n_orph;
        c->stats_ex[3] = (int64_t)w[LC_HEAVY]; c->stats_ex[4] = (int64_t)w[LC_ITEMS]; c->stats_ex[5] = (int64_t)w[LC_MAXPAIRS];
        c->rng_pos = rng_pos0 + (uint64_t)n_pre;                // one draw per row of the new level (the launch covered a bound)
        O.n = n_pre;
        P_glob = P; O_glob = n_orph; n_pre_glob = n_pre; n_glob_next = n_pre;
        dropped = (int64_t)w[LC_DROPPED];
        have_next = dropped == 0 || (erase_on_device && dropped <= ERASE_MAX);      // (erased in place on the device: the prologue saw the level as it is now)
        if (have_next) O.n = n_pre - dropped;
        else if (dropped <= ERASE_MAX) {                // the first level of this context that erases rows: in place, from here (the buffer exists from now on)
            GSR_TRY(erase_in_place(n_pre, true, out_cap));
            O.n = n_pre - dropped;
        } else {                        // many rows to erase: the host finishes the level (scan + compaction), and the prologue is taken again
            GSR_TRY(exclusive_scan<int>(c, c->keep.as<int>(), c->kpos.as<int>(), n_pre));
            GSR_TRY(compact_erased(n_pre - dropped));
        }
    } else {
        GSR_TRY(flags_and_validity());
        GSR_TIME1(c->ev[5], st);
    }
    // the next level's prologue with this level's last round trip (one GPU); a partitioned / sharded level just waits for the stream
    if (!part && !sharded && !have_next && O.n > 0 && want_pro) {
        GSR_TRY(enqueue_prologue(c, O, O.n, nullptr, cb_next));
        GSR_TRY(read_back_level(c, q, w));
        have_next = true;
    } else if (!have_next) {
        GSR_HIP(hipStreamSynchronize(st));
    }
    unborrow_level0(c);                 // a borrowed level 0 goes back to the caller; cur gets its own buffers again
    c->cur.swap(c->nxt);
    if (out_active) {                   // the new current level lives in the caller's arrays: borrowed, its own buffers parked in spare
        out_active = false;
        for (int i = 0; i < 5; ++i) c->spare[i].swap(c->spare_out[i]);
        c->cur_borrowed = true;
    }
    if (have_next && want_pro && c->cur.n > 0) take_prologue(c, w, c->cur.n, cb_next); else c->pro.valid = false;
    if (part) { c->gid.swap(c->gid_next); c->n_global = n_glob_next; }
    c->stats[3] = dropped;
    c->stats[7] = c->cur.n;
    c->stats_ex[6] = c->round_trips;
    c->stats_ex[7] = spec ? 1 : 0;
    memset(c->phase_ms, 0, sizeof(c->phase_ms));
    memset(c->part_ms, 0, sizeof(c->part_ms));
    memset(c->kernel_ms, 0, sizeof(c->kernel_ms));
    if (c->timing >= 1) {
        if (c->timing >= 2) for (int i = 0; i < 5; ++i) (void)hipEventElapsedTime(&c->phase_ms[i], c->ev[i], c->ev[i + 1]);
        (void)hipEventElapsedTime(&c->phase_ms[5], c->ev[0], c->ev[5]);
        c->phase_ms[5] += pro_ms;           // the level's time includes its prologue, wherever that ran
        if (c->timing >= 2) c->phase_ms[0] += pro_ms;
        c->phase_ms[6] = c->phase_ms[7] = 0.0f;
        if (P > 0 && !c->sparse_path) (void)hipEventElapsedTime(&c->phase_ms[6], c->evk[0], c->evk[1]);
        if (P > 0 && (M > 0 || c->sparse_path)) (void)hipEventElapsedTime(&c->phase_ms[7], c->evk[2], c->evk[3]);
        if (part && c->timing >= 2) {
            (void)hipEventElapsedTime(&c->part_ms[0], c->evp[0], c->evp[1]);
            if (F > 0) { GSR_HIP(hipStreamSynchronize(c->aux2)); (void)hipEventElapsedTime(&c->part_ms[1], c->evp[2], c->evp[3]); }
        }
        c->kernel_ms[0] = c->phase_ms[7];
        if (P > 0) (void)hipEventElapsedTime(&c->kernel_ms[1], c->evm[0], c->evm[1]);
        if (c->timing >= 2 && fixed_tried && !c->partition_overflowed) {
            (void)hipEventElapsedTime(&c->kernel_ms[2], c->evm[2], c->evm[3]);
            (void)hipEventElapsedTime(&c->kernel_ms[3], c->evm[4], c->evm[5]);
        }
    }
    if (n_out) *n_out = c->cur.n;
    if (n_dropped) *n_dropped = dropped;
    return GSR_OK;
}

}  // namespace

int32_t gsr_hem_run_level(gsr_hem_ctx* c, int64_t* n_out, int64_t* n_dropped) {
    if (!c || !c->have_level) return fail(GSR_E_INVALID, "gsr_hem_run_level: no level set");
    GSR_HIP(hipSetDevice(c->device));
    c->round_trips = 0;
    const uint64_t rng_pos0 = c->rng_pos;
    const bool out0 = c->out_pending;
    int32_t r;
    {
        LevelRun run(c, c->async_ok);
        r = run.run(n_out, n_dropped);
    }                                   // (its destructor hands nxt its own buffers back)
    if (r == GSR_RETRY_SYNC) {
        // The asynchronous level ran on buffers that turned out too small (or met another case it leaves to the synchronous schedule).
        // Its input is untouched: the stream is drained, the stream position and the output request are put back, and the level runs again
        // with every buffer sized from the counts -- the context keeps those sizes, the next level of this size is asynchronous again.
        GSR_HIP(hipStreamSynchronize(c->stream));
        if (c->aux) GSR_HIP(hipStreamSynchronize(c->aux));
        if (c->aux2) GSR_HIP(hipStreamSynchronize(c->aux2));
        c->rng_pos = rng_pos0;
        c->out_pending = out0;
        c->pro.valid = false;
        LevelRun run(c, false);
        r = run.run(n_out, n_dropped);
        c->stats_ex[7] = 2;             // (statistic: an asynchronous attempt was rerun)
    }
    return r;
}

// the normals of a level leave with the level (gsr_hem_run_levels): the arithmetic of gsr_normals_from_cov (icp.hip), gsr_normals.h
__global__ __launch_bounds__(256) void k_level_normals(int64_t n, const float* __restrict__ cov6, double* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double v[3];
        gsr::normal_of_cov_d(cov6[6 * i], cov6[6 * i + 1], cov6[6 * i + 2], cov6[6 * i + 3], cov6[6 * i + 4], cov6[6 * i + 5], v);
        out[3 * i] = v[0]; out[3 * i + 1] = v[1]; out[3 * i + 2] = v[2];
    }
}

int32_t gsr_hem_run_levels(gsr_hem_ctx* c, int32_t n_levels, float* xyz, float* color, float* cov6, float* opacity, float* sh, double* normals,
                           double* normals0, int64_t arena_rows, gsr_hem_level_report* reports) {
    if (!c || !c->have_level) return fail(GSR_E_INVALID, "gsr_hem_run_levels: no level set");
    if (n_levels < 0 || (n_levels > 0 && !reports)) return fail(GSR_E_INVALID, "gsr_hem_run_levels: bad argument (n_levels = %d)", n_levels);
    if (n_levels > 0 && (!xyz || !color || !cov6 || !opacity || arena_rows <= 0)) return fail(GSR_E_INVALID, "gsr_hem_run_levels: NULL arena");
    if (c->cur.F > 0 && !sh && n_levels > 0) return fail(GSR_E_INVALID, "gsr_hem_run_levels: the levels have F = %d but there is no SH arena", c->cur.F);
    if (c->comm || c->shard_world > 1) return fail(GSR_E_INVALID, "gsr_hem_run_levels: not for partitioned / sharded levels (run them level by level)");
    GSR_HIP(hipSetDevice(c->device));
    const int F = c->cur.F;
    const bool want_normals = normals != nullptr || normals0 != nullptr;
    if (want_normals && !c->side) {
        GSR_HIP(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
        GSR_HIP(hipEventCreateWithFlags(&c->ev_side, hipEventDisableTiming));
        GSR_HIP(hipEventCreateWithFlags(&c->ev_side_fork, hipEventDisableTiming));
    }
    bool side_used = false;
    // the side stream starts behind whatever the caller has enqueued on the context's stream (the arrays of level 0 may still be written)
    if (want_normals) { GSR_HIP(hipEventRecord(c->ev_side_fork, c->stream)); GSR_HIP(hipStreamWaitEvent(c->side, c->ev_side_fork, 0)); }
    if (normals0 && c->cur.n > 0) {
        hipLaunchKernelGGL(k_level_normals, dim3(stride_grid(c->cur.n)), dim3(256), 0, c->side, c->cur.n, c->cur.cov6.as<float>(), normals0);
        side_used = true;
    }
    int64_t off = 0;
    int32_t r = GSR_OK;
    for (int k = 0; k < n_levels; ++k) {
        const int64_t n_in = c->cur.n;
        if (off + n_in > arena_rows) {
            r = fail(GSR_E_INVALID, "gsr_hem_run_levels: level %d needs rows [%lld, %lld) of arenas that hold %lld rows", k + 1, (long long)off, (long long)(off + n_in),
                     (long long)arena_rows);
            break;
        }
        gsr_hem_level_report& R = reports[k];
        memset(&R, 0, sizeof(R));
        R.offset_rows = off;
        if (n_in > 0) {
            r = gsr_hem_set_output(c, xyz + 3 * off, color + 3 * off, cov6 + 6 * off, opacity + off, F > 0 ? sh + (size_t)F * off : nullptr, n_in);
            if (r != GSR_OK) break;
        }
        int64_t n_out = 0, dropped = 0;
        c->last_level = k == n_levels - 1;
        r = gsr_hem_run_level(c, &n_out, &dropped);
        c->last_level = false;
        if (r != GSR_OK) break;
        R.rows = n_out; R.dropped = dropped; R.rng_position = c->rng_pos;
        memcpy(R.stats, c->stats, sizeof(R.stats)); memcpy(R.stats_ex, c->stats_ex, sizeof(R.stats_ex));
        memcpy(R.phase_ms, c->phase_ms, sizeof(R.phase_ms)); memcpy(R.kernel_ms, c->kernel_ms, sizeof(R.kernel_ms));
        // the level is complete (its answer came back behind its last kernel, the validity erase included) and from here on it is only
        // read: its normals beside the next level, on the side stream
        if (normals && n_out > 0) {
            hipLaunchKernelGGL(k_level_normals, dim3(stride_grid(n_out)), dim3(256), 0, c->side, n_out, cov6 + 6 * off, normals + 3 * off);
            side_used = true;
        }
        off += (n_out + 63) / 64 * 64;
    }
    if (side_used) {            // whatever is enqueued on the context's stream after this call sees the normals
        GSR_HIP(hipEventRecord(c->ev_side, c->side));
        GSR_HIP(hipStreamWaitEvent(c->stream, c->ev_side, 0));
        GSR_HIP(hipGetLastError());
    }
    return r;
}

#ifdef GSR_SELECT_PROFILE
int32_t gsr_debug_select_profile(unsigned long long* out16, int32_t reset) { return select_profile(out16, reset); }
#endif

int32_t gsr_debug_logf(const float* x, int64_t n, float* out, int32_t device) {
    if (n < 0 || (n > 0 && (!x || !out))) return fail(GSR_E_INVALID, "gsr_debug_logf: bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(GSR_E_NO_DEVICE, "gsr_debug_logf: no HIP device visible");
    if (n == 0) return GSR_OK;
    GSR_HIP(hipSetDevice(device));
    DevBuf a, b;
    int32_t r = a.reserve((size_t)n * 4);
    if (r == GSR_OK) r = b.reserve((size_t)n * 4);
    hipError_t e = hipSuccess;
    if (r == GSR_OK) {
        e = hipMemcpy(a.p, x, (size_t)n * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_debug_logf, dim3(stride_grid(n)), dim3(256), 0, nullptr, n, a.as<float>(), b.as<float>());
            e = hipMemcpy(out, b.p, (size_t)n * 4, hipMemcpyDeviceToHost);
        }
    }
    a.release(); b.release();
    if (r != GSR_OK) return r;
    if (e != hipSuccess) return fail(GSR_E_HIP, "gsr_debug_logf: %s", hipGetErrorString(e));
    return GSR_OK;
}

int32_t gsr_debug_kld(const float* cm, const float* cc, const float* pm, const float* pc, int64_t n, float* out, int32_t device) {
    if (n < 0 || (n > 0 && (!cm || !cc || !pm || !pc || !out))) return fail(GSR_E_INVALID, "gsr_debug_kld: bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(GSR_E_NO_DEVICE, "gsr_debug_kld: no HIP device visible");
    if (n == 0) return GSR_OK;
    GSR_HIP(hipSetDevice(device));
    DevBuf b[5];
    const size_t sz[5] = {(size_t)n * 12, (size_t)n * 24, (size_t)n * 12, (size_t)n * 24, (size_t)n * 4};
    const float* src[4] = {cm, cc, pm, pc};
    int32_t r = GSR_OK;
    hipError_t e = hipSuccess;
    for (int i = 0; i < 5 && r == GSR_OK; ++i) r = b[i].reserve(sz[i]);
    for (int i = 0; i < 4 && r == GSR_OK && e == hipSuccess; ++i) e = hipMemcpy(b[i].p, src[i], sz[i], hipMemcpyHostToDevice);
    if (r == GSR_OK && e == hipSuccess) {
        hipLaunchKernelGGL(k_debug_kld, dim3(stride_grid(n)), dim3(256), 0, nullptr, n, b[0].as<float>(), b[1].as<float>(), b[2].as<float>(),
                           b[3].as<float>(), b[4].as<float>());
        e = hipMemcpy(out, b[4].p, sz[4], hipMemcpyDeviceToHost);
    }
    for (int i = 0; i < 5; ++i) b[i].release();
    if (r != GSR_OK) return r;
    if (e != hipSuccess) return fail(GSR_E_HIP, "gsr_debug_kld: %s", hipGetErrorString(e));
    return GSR_OK;
}

int32_t gsr_debug_stage1(const float* parent_mean, const float* parent_cov6, const float* child_mean, const float* child_cov6, int64_t n,
                         float kld_thr, uint8_t* parent_regular, uint8_t* child_regular, uint8_t* white, uint8_t* reject, float* T1,
                         uint8_t* clip_on, int32_t device) {
    if (n < 0 || (n > 0 && (!parent_mean || !parent_cov6 || !child_mean || !child_cov6 || !parent_regular || !child_regular || !white || !reject || !T1 || !clip_on)))
        return fail(GSR_E_INVALID, "gsr_debug_stage1: bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(GSR_E_NO_DEVICE, "gsr_debug_stage1: no HIP device visible");
    if (n == 0) return GSR_OK;
    GSR_HIP(hipSetDevice(device));
    DevBuf b[10];
    const size_t sz[10] = {(size_t)n * 12, (size_t)n * 24, (size_t)n * 12, (size_t)n * 24, (size_t)n, (size_t)n, (size_t)n, (size_t)n, (size_t)n * 4, (size_t)n};
    const float* src[4] = {parent_mean, parent_cov6, child_mean, child_cov6};
    void* dst[6] = {parent_regular, child_regular, white, reject, T1, clip_on};
    int32_t r = GSR_OK;
    hipError_t e = hipSuccess;
    for (int i = 0; i < 10 && r == GSR_OK; ++i) r = b[i].reserve(sz[i]);
    for (int i = 0; i < 4 && r == GSR_OK && e == hipSuccess; ++i) e = hipMemcpy(b[i].p, src[i], sz[i], hipMemcpyHostToDevice);
    if (r == GSR_OK && e == hipSuccess) {
        hipLaunchKernelGGL(k_debug_stage1, dim3(stride_grid(n)), dim3(256), 0, nullptr, n, b[0].as<float>(), b[1].as<float>(), b[2].as<float>(), b[3].as<float>(),
                           kld_thr, b[4].as<uint8_t>(), b[5].as<uint8_t>(), b[6].as<uint8_t>(), b[7].as<uint8_t>(), b[8].as<float>(), b[9].as<uint8_t>());
        for (int i = 0; i < 6 && e == hipSuccess; ++i) e = hipMemcpy(dst[i], b[4 + i].p, sz[4 + i], hipMemcpyDeviceToHost);
    }
    for (int i = 0; i < 10; ++i) b[i].release();
    if (r != GSR_OK) return r;
    if (e != hipSuccess) return fail(GSR_E_HIP, "gsr_debug_stage1: %s", hipGetErrorString(e));
    return GSR_OK;
}

int32_t gsr_debug_kl_gate(const float* s2, const float* det_c, const float* det_p, int64_t n, float thr, uint8_t* reject, float* fast_log,
                          uint8_t* need_exact, int32_t device) {
    if (n < 0 || (n > 0 && (!s2 || !det_c || !det_p || !reject || !fast_log || !need_exact))) return fail(GSR_E_INVALID, "gsr_debug_kl_gate: bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(GSR_E_NO_DEVICE, "gsr_debug_kl_gate: no HIP device visible");
    if (n == 0) return GSR_OK;
    GSR_HIP(hipSetDevice(device));
    DevBuf b[6];
    const size_t sz[6] = {(size_t)n * 4, (size_t)n * 4, (size_t)n * 4, (size_t)n, (size_t)n * 4, (size_t)n};
    const float* src[3] = {s2, det_c, det_p};
    int32_t r = GSR_OK;
    hipError_t e = hipSuccess;
    for (int i = 0; i < 6 && r == GSR_OK; ++i) r = b[i].reserve(sz[i]);
    for (int i = 0; i < 3 && r == GSR_OK && e == hipSuccess; ++i) e = hipMemcpy(b[i].p, src[i], sz[i], hipMemcpyHostToDevice);
    if (r == GSR_OK && e == hipSuccess) {
        hipLaunchKernelGGL(k_debug_kl_gate, dim3(stride_grid(n)), dim3(256), 0, nullptr, n, b[0].as<float>(), b[1].as<float>(), b[2].as<float>(), thr,
                           b[3].as<uint8_t>(), b[4].as<float>(), b[5].as<uint8_t>());
        e = hipMemcpy(reject, b[3].p, sz[3], hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(fast_log, b[4].p, sz[4], hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(need_exact, b[5].p, sz[5], hipMemcpyDeviceToHost);
    }
    for (int i = 0; i < 6; ++i) b[i].release();
    if (r != GSR_OK) return r;
    if (e != hipSuccess) return fail(GSR_E_HIP, "gsr_debug_kl_gate: %s", hipGetErrorString(e));
    return GSR_OK;
}

}  // extern "C"
