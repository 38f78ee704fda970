// hem_select.h -- what the child selection (hem_select.hip: k_parent_prep, k_spans, k_select) shares with the rest of the level (hem.hip):
// the argument block, the per-parent record, the row-span and filter geometry (also used by the partitioned level's halo marking and
// by the test hooks), and the launchers.  hem_select.hip is the one translation unit built without MachineLICM.
#pragma once
#include "hem_device.h"

namespace gsr {

// ------------------------------------------------------------------------------------------------
// k_select: child selection (mixture.cpp:102-137) and wL_si (mixture.cpp:140-164), one wavefront per parent
//
//   rows      the grid rows (fixed y,z cell) that meet the parent's search region; lane r of a 64-row batch computes
//             the contiguous span [s, s+len) of sorted components of its row (x-clipped to the sphere, and to the
//             pre-reject ellipsoid for a regular parent)
//   stream    the spans of a batch form one flattened index space [0, total); lane l of chunk c0 handles candidate
//             c0 + l.  Its row comes from a BIT MASK in LDS (bit p set <=> a row starts at flat position p): one
//             broadcast 8-byte LDS read per chunk and two v_mbcnt give every lane the number of row starts at or
//             before it, one ds_bpermute fetches that row's (start - prefix).  (The six-step binary search by lane
//             shuffles this replaces was a chain of six DEPENDENT ds_bpermute per chunk: 24 cycles of LDS-pipe issue
//             each, scripts/micro/valu_issue.hip.)
//   stage 1   a conservative filter, not a decision: for a regular parent the squared Mahalanobis distance in
//             whitened form |U d|^2 (U = Cholesky factor of P^-1, nine fused multiply-adds) against the pre-reject
//             bound + 1 %; for an irregular parent, and for the irregular children (pass B), the reference's own
//             radius test.  Survivors are compacted (ballot + mbcnt) into a per-wave LDS ring.
//   stage 2   on full batches of 64 survivors, the reference's float32 expressions bit for bit: radius test
//             d2 < R^2 (pointindex.cpp:137), colour gate, KL gate, parent rule (mixture.cpp:122-133).  The KL gate's
//             logf is decided with the hardware v_log_f32 when the result is farther from the threshold than its
//             error bound, and with glibc's own algorithm (gsr_math.h) otherwise -- the decision is the reference's
//             in both cases.  Accepted pairs go to a second LDS queue.
//   stage 3   likelihood (mixture.cpp:54-64) on full batches of 64 accepted pairs, pair records out.
//   modes     COUNT  only counts accepted pairs (first pass of the two-pass fallback)
//             FILL   writes pairs at poff[p]  (second pass of the fallback)
//             SPARSE single pass: writes pairs at coff[p] (capacity = candidates scanned, from k_spans), count to
//                    pcnt[p]; k_compact_pairs then packs them
// ------------------------------------------------------------------------------------------------
struct ParentRec;
struct SelectArgs {
    const float4* A;                // compact {x, y, z, flags}
    const float4* geo;              // 64-byte records {A, B, C, D}
    const float* Rs;
    const unsigned* plist;
    const unsigned* porder;         // processing order of the parents (heavy ones first), or NULL = natural order
    int xcd;                        // 1 = light parents are dealt to the XCDs in contiguous chunks (block_slot)
    const int* nheavy;              // device: number of heavy parents at the head of porder
    int own_lo, own_hi;             // sharded level: this rank evaluates parents [own_lo, own_hi) of plist only
    const int* cellStart;
    // pass A streams the NON-PARENT components only (a parent can be claimed by no parent but itself, mixture.cpp:131-133, and
    // that pair is queued directly): Ac = their {x, y, z, (sorted position << 2) | flags} in cell order, cellStartC = the grid's
    // prefix table counted over them.  A third of the components are parents.
    const float4* Ac;
    const int* cellStartC;
    const int* cellStartI;          // the same table counted over the irregular components (the list pass B scans); only when n_irr > 0
    const double* logtab;           // glibc logf table (LDS copy)
    const int* irank;               // irank[j] = number of irregular components among sorted positions [0, j)   (n + 1 entries)
    const unsigned* ipos;           // sorted positions of the irregular components, ascending
    int n_irr;
    int ell;                        // 1 = clip the grid rows of a regular parent to its Mahalanobis ellipsoid
    const GridParams* gp;
    int P;
    float colorThr2;                // largest float x with sqrtf(x) <= kappa^2 / 2: the colour gate on the squared colour distance
    float kldThr, tau2;
    const ParentRec* prec;          // [P] per-parent records (k_parent_prep)
    unsigned* pcnt;                 // COUNT / SPARSE out: accepted pairs per parent
    unsigned* pcap;                 // k_spans out: candidates scanned per parent
    const int64_t* poff;            // FILL: compact offsets;  SPARSE: capacity offsets
    unsigned* pair_child;
    float* pair_wl;
    // heavy parents (the first *nheavy slots of porder) are cut into work items of SEL_PART candidates of their flat
    // candidate space; a fixed number of workgroups at the head of the launch pulls the items from a queue
    const uint2* hitem;             // item -> (slot in porder, part)
    const int* hfirst;              // [P] first item of a heavy parent, -1 for the others
    unsigned* part_cnt;             // accepted pairs per item (COUNT / SPARSE out, FILL in)
    int* hq;                        // hq[0] = number of items, hq[1] = queue cursor
    int heavy_blocks;               // workgroups at the head of the launch that serve the queue (0 = no splitting)
    const unsigned* part_p;         // device: candidates per work item (<= SEL_PART; smaller on small levels, where an item is the critical path) -- k_heavy_items
                                    // derives it from the level's candidate total, which the host of an asynchronous level never sees
    // SPARSE: the pair buffers hold cap_pairs entries.  A segment that would end beyond them is SKIPPED (its parent gets no pairs) and
    // *abort_p is raised: an asynchronous level sizes nothing from the candidate total -- it runs on the buffers the context has and
    // reruns synchronously when they were too small -- so every write has to be clamped on the device.  0 = unchecked
    unsigned long long cap_pairs;
    int* abort_p;
    int np;                         // light parents per wave (1 ... SEL_NP), see SEL_NP
    int2* rowlist;                  // [P][2][SEL_ROWS]: the non-empty row spans {first position, length} of pass A / pass B in scan order (k_spans), or NULL
};

enum { SEL_COUNT = 0, SEL_FILL = 1, SEL_SPARSE = 2 };
#define SEL_PART 8192         // candidates per work item of a heavy parent, at most (SelectArgs::part)
#define SEL_HEAVY_BLOCKS 2048  // workgroups at the head of k_select that serve the queue of heavy work items (a multiple of 8)
#define SEL_QCAP 256          // survivor ring (power of two >= 64 + SEL_U*64: the rest of a batch -- or 63 entries and a parent's own -- plus a group of chunks)
#define SEL_U 3               // chunks whose candidate loads are in flight together (2 or 3: equal, 4: +1 %, 6: +2 %, 8: +14 % -- registers)
#define SEL_MCAP 1024         // flat positions covered by the row-start bit mask at a time (a parent scans ~500 candidates; LDS is allocated in granules of 1 280 bytes and the workgroup sits just under eight)
#define SEL_ROWS 16           // non-empty row spans per parent and pass that k_spans hands to k_select (a parent with more recomputes them)
#define SEL_PAD (64 * SEL_U)  // entries the sorted A array is padded by: the inactive lanes of a batch's last chunks read past the last row

// Everything k_select / k_spans need to know about a parent, computed ONCE per parent by k_parent_prep (one thread each)
// and fetched by the selection waves with scalar loads: the cofactor inverse, the clipping and whitening constants cost
// ~300 vector instructions, which every one of the 64 lanes of a wave used to repeat for its parent (17 % of k_select).
struct EllClip {
    float on;                 // 1.0f = clip (regular parent with a sane Schur complement)
    float k11, k12, k22, kr, im00, m01, m02, T;
};
struct ParentRec {            // 40 dwords
    f3 pm, pcol;
    s6 pinv;
    float det_p, inv_det_p, pweight, R, R2;
    float white;              // 1.0f = stage 1 uses the whitened Mahalanobis filter (regular parent)
    float u00, u01, u02, u11, u12, u22;   // upper Cholesky factor of pinv
    float T1;                 // filter bound on |U d|^2: the pre-reject bound + 1 %
    EllClip ec;
    int js;
    int active;               // 0: zero / NaN radius or non-finite mean -> no children at all
    int selfq;                // 1: regular parent -- it is not in the stream of pass A and queues itself (flat candidate 0)
    float ey, ez;             // half extents of the pre-reject ellipsoid along y and z (+0.1 %): no row of pass A lies beyond them
    int rows;                 // written by k_spans: bit 31 = the row lists of this parent are valid; bits 0-7 / 8-15 = non-empty rows of pass A / B
                              // (0xff = more than SEL_ROWS: that pass recomputes its spans)
};
static_assert(sizeof(ParentRec) == 160, "ParentRec is fetched as 40 dwords");

__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }       // v_sqrt_f32, 1 ulp: clipping margins only

// popc(mask & lanes below this one) + base, two VALU instructions
__device__ __forceinline__ int mbcnt64(unsigned long long mask, int base) {
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, (unsigned)base));
}
__device__ __forceinline__ unsigned long long uniform64(unsigned long long v) {
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
           (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}
// inclusive prefix sum over the 64 lanes by DPP row shifts / row broadcasts (six VALU instructions, no LDS traffic);
// every lane must be active
__device__ __forceinline__ int wave_incl_scan(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);      // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);      // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);      // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);      // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);      // row_bcast:15 into rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);      // row_bcast:31 into rows 2, 3
    return v;
}

// GSR_SELECT_PROFILE (variant builds only, scripts/select_profile.py): where a selection wave spends its clock -- s_memtime deltas
// of the phases, summed over all waves with one atomic per phase and parent
#ifdef GSR_SELECT_PROFILE
extern __device__ unsigned long long g_sel_prof[1024 * 16];       // 1024 copies (by workgroup): atomics on ONE address from 10^6 waves serialise
#define SEL_PROF_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define SEL_PROF_ADD(slot, t0, lane) do { if ((lane) == 0) atomicAdd(&g_sel_prof[(blockIdx.x & 1023u) * 16 + (slot)], __builtin_amdgcn_s_memtime() - (t0)); } while (0)
#define SEL_PROF_CNT(slot, v, lane) do { if ((lane) == 0) atomicAdd(&g_sel_prof[(blockIdx.x & 1023u) * 16 + (slot)], (unsigned long long)(v)); } while (0)
#else
#define SEL_PROF_T(var)
#define SEL_PROF_ADD(slot, t0, lane)
#define SEL_PROF_CNT(slot, v, lane)
#endif

// KL gate decision  KLD(child, parent) > thr  (gaussian.hpp:106-109, mixture.cpp:126-129) with
//     k = 0.5f * (((smd + tr) - 3.0f) - logf(q)),   q = det_c / det_p,   s2 = (smd + tr) - 3.0f  as the reference rounds it.
// Fast path: q' = det_c * (1 / det_p) (within 1.2e-7 of q) and v_log_f32 (1 ulp of log2 q', plus the rounding of the
// product with ln 2) put lf within 4.2e-7 (1 + |ln q|) of ln q, glibc's logf is within 1 ulp of it, so k computed with lf
// differs from the reference's k by less than 0.5 (4.8e-7 (1 + |lf|) + 2 ulp(s2 - lf)) < 1e-6 (1 + |lf| + |s2|) / 2:
// outside that margin around thr both give the same decision; inside it (and for q' not a comfortably normal positive
// number, or s2 not finite) the exact expression runs: IEEE division and glibc's own logf algorithm.
// tests/test_hem_gpu.py::test_fast_log_margin checks the bound and the decisions on the device.
__device__ __forceinline__ bool kl_gate_rejects(float s2, float det_c, float det_p, float inv_det_p, float thr, const double* logtab,
                                                float* lf_out = nullptr, bool* exact_out = nullptr) {
    const float qf = det_c * inv_det_p;
    const float lf = __builtin_amdgcn_logf(qf) * 0.6931471805599453f;
    const float kf = 0.5f * (s2 - lf);
    const float margin = 1e-6f * (1.0f + fabsf(lf) + fabsf(s2));
    bool reject = kf > thr;
    const bool need_exact = !(qf >= 4.0f * FLT_MIN && qf <= 0.25f * FLT_MAX) || !(fabsf(s2) <= FLT_MAX) || !(fabsf(kf - thr) > margin);
    if (need_exact) {
        const float k = 0.5f * (s2 - glibc_logf_tab(det_c / det_p, logtab));
        reject = k > thr;
    }
    if (lf_out) *lf_out = lf;
    if (exact_out) *exact_out = need_exact;
    return reject;
}

// Row clipping by the parent's filter ellipsoid E = { d : d^T M d <= T_clip } (struct EllClip, filled by make_filter, which
// also carries the argument): a regular child outside E fails the stage-1 filter anyway, so only the grid cells E touches need
// scanning (E is inscribed in the query sphere; for a flat disc it holds a few percent of its volume).

// The span (first sorted position, length) of one grid row (ry, rz) for a parent at pm: the row's cells within the
// sphere of radius sqrt(Ra2), clipped to the pre-reject ellipsoid when `clip`.  IRR: positions in the irregular list.
// ONE definition shared by k_select (every mode) and k_spans: the capacities must equal what the passes scan.
// (The square roots are the hardware's 1-ulp v_sqrt_f32: every one of them sits behind a margin of 1e-5 or more.)
template <bool IRR>
__device__ __forceinline__ void select_row_span(const SelectArgs& a, const GridParams& g, const f3& pm, const EllClip& ec, bool clip,
                                                float Ra2, int x0, int x1, int ry, int rz, int& s, int& len) {
    // distance from the parent to the row's y/z slab (widened by the rounding slack)
    // (the first / last row of the grid also holds every centre clamped in from outside: half-infinite)
    const bool edge = ry == 0 || ry == g.gy - 1 || rz == 0 || rz == g.gz - 1;
    const float ylo = ry == 0 ? -FLT_MAX : g.oy + ry * g.c - g.slack, yhi = ry == g.gy - 1 ? FLT_MAX : g.oy + (ry + 1) * g.c + g.slack;
    const float zlo = rz == 0 ? -FLT_MAX : g.oz + rz * g.c - g.slack, zhi = rz == g.gz - 1 ? FLT_MAX : g.oz + (rz + 1) * g.c + g.slack;
    const float dy = fmaxf(0.0f, fmaxf(ylo - pm.y, pm.y - yhi));
    const float dz = fmaxf(0.0f, fmaxf(zlo - pm.z, pm.z - zhi));
    const float rem = Ra2 - dy * dy - dz * dz;
    if (rem >= 0.0f) {
        const float hx = fast_sqrt(rem) * 1.00001f + g.slack;
        float lo = -hx, hi = hx;                                   // x interval relative to the parent
        if (clip && !edge) {
            const float cy = 0.5f * (ylo + yhi) - pm.y, cz = 0.5f * (zlo + zhi) - pm.z;
            const float hy = 0.5f * (yhi - ylo), hz = 0.5f * (zhi - zlo);
            const float sc = fast_sqrt(fmaxf(0.0f, ec.k11 * cy * cy + 2.0f * ec.k12 * cy * cz + ec.k22 * cz * cz));
            const float smin = fmaxf(0.0f, sc * 0.999f - ec.kr * fast_sqrt(hy * hy + hz * hz));
            const float rem2 = ec.T - smin * smin;
            if (rem2 < 0.0f) {
                lo = 1.0f; hi = -1.0f;                             // the row misses the ellipsoid
            } else {
                const float w = fast_sqrt(rem2 * ec.im00) * 1.001f;
                const float t1 = ec.m01 * cy, t2 = ec.m02 * cz;
                const float xc = -(t1 + t2) * ec.im00;
                const float dl = (fabsf(ec.m01) * hy + fabsf(ec.m02) * hz) * ec.im00 * 1.001f;
                // the two terms of xc may cancel (a thin disc tilted against the axes): its rounding error is relative to them, not to xc
                const float pad = g.slack + 1e-5f * fabsf(xc) + 1e-6f * (fabsf(t1) + fabsf(t2)) * ec.im00;
                lo = fmaxf(lo, xc - dl - w - pad);
                hi = fminf(hi, xc + dl + w + pad);
            }
        }
        if (lo <= hi) {
            int xa = cell_of(pm.x + lo, g.ox, g.inv_c, g.gx), xb = cell_of(pm.x + hi, g.ox, g.inv_c, g.gx);
            xa = xa < x0 ? x0 : xa;
            xb = xb > x1 ? x1 : xb;
            const int rowbase = (rz * g.gy + ry) * g.gx;
            int e;
            if (IRR) { s = a.cellStartI[rowbase + xa]; e = a.cellStartI[rowbase + xb + 1]; }      // positions in the irregular list
            else { s = a.cellStartC[rowbase + xa]; e = a.cellStartC[rowbase + xb + 1]; }      // positions in the children's stream
            len = e - s;
        }
    }
}

// The stage-1 filter of a regular parent, and the row clipping that goes with it (one thread per parent, float64).
//
// What the reference decides (gaussian.hpp:106-109, mixture.cpp:126-129), with M = its float32 cofactor inverse of the
// parent's covariance (pr.pinv, the very numbers stage 2 uses), d = the float32 mean difference, C the child:
//     reject  <=>  fl(0.5 (s2 - lf)) > thr,   s2 = fl(fl(smd + tr) - 3),  smd = fl(d^T M d),  tr = fl(tr(M C)),  lf = logf(fl(det_c / det_p)).
// Stage 1 may drop a pair only when that is CERTAIN.  With u = 2^-24, S = d^T M d and Tr = tr(M C) in real arithmetic
// on the float32 data, K >= tr(M) / lambda_min(M), and a regular child (is_regular: C positive definite, its float32
// determinant within DET_TOL of det C, quotient of determinants a normal number):
//     smd >= S (1 - 6.1 u K),  tr >= Tr (1 - 5.1 u K)                       (dot products of length 3 + 3 / 3 + 2, |M|:|C| <= K Tr)
//     s2 - lf >= (S + Tr)(1 - theta) - 3 - ln(det_c / det_p) - 1.5e-5,     theta = 6.1 u K + 2 u
//     Tr (1 - theta) - ln det C >= 3 + ln det((1 - theta) M)               (M positive definite: minimum over all SPD C)
//  => s2 - lf >= S (1 - theta) + 3 ln(1 - theta) - G - DET_TOL - 2e-5,     G = -ln(det M * det_p)   (0 for an exact inverse)
// so the pair is rejected for certain when S > S_min = (2 thr + G + DET_TOL + 3.2 theta + 4e-5) / (1 - theta).  The filter
// evaluates S as |U d|^2 with the float32-rounded Cholesky factor U of M (nine fused multiply-adds): that value is below
// S (1 + 4.1 u sqrt(K))^2 (1 + 3.1 u), which the factor (1 + theta)^2 on the bound covers.  Everything a parent needs
// for it -- M positive definite, K, G -- is computed HERE from M's float32 entries in float64; a parent for which it
// cannot be certified (theta > 1/4: condition number beyond ~3e5; M not positive definite; |G| > 1) scans its search
// sphere with the reference's radius test instead (white = 0), like every parent did before round 1's pre-reject.
// tests/test_hem_gpu.py::test_stage1_filter_never_rejects_an_accepted_pair attacks the bound on the device.
//
// Row clipping: only the grid rows the ellipsoid E = { d : S <= T_clip } meets need scanning, T_clip = T1 (1 + theta)^2 (1 + 1e-4)
// (beyond it the filter's own float32 value exceeds T1).  For a row = the slab dy in [cy - hy, cy + hy], dz in [cz - hz, cz + hz]:
// S = M00 (dx - xc(dy, dz))^2 + q(dy, dz), q the quadratic form of the Schur complement Ks of M; sqrt(q) is a norm, so over the
// slab sqrt(q) >= sqrt(q(c)) - sqrt(lmax(Ks)) |h|, lmax(Ks) <= tr Ks, and xc is linear: a conservative x interval in ~45
// flops (select_row_span).  The constants are float64 values rounded to float32 with the margins folded in: Ks scaled
// down by (1 - 4 theta - 2e-3) -- the float32 evaluation of q loses up to 12 u tr(Ks) / lmin(Ks) <= 4 theta of it --
// T_clip and 1 / M00 rounded up.
__device__ __forceinline__ void make_filter(const s6& Mf, float det_p, float kldThr, int ell, bool parent_regular, ParentRec& pr) {
    EllClip ec;
    ec.on = 0.0f; ec.k11 = ec.k12 = ec.k22 = ec.kr = ec.im00 = ec.m01 = ec.m02 = 0.0f; ec.T = __builtin_inff();
    pr.white = 0.0f;
    pr.u00 = pr.u01 = pr.u02 = pr.u11 = pr.u12 = pr.u22 = 0.0f;
    pr.T1 = __builtin_inff();
    pr.ey = pr.ez = __builtin_inff();
    const double thr2 = 2.0 * (double)kldThr;
    if (parent_regular && thr2 >= 0.0 && thr2 < 1e30) {
        const double m00 = Mf.e00, m01 = Mf.e01, m02 = Mf.e02, m11 = Mf.e11, m12 = Mf.e12, m22 = Mf.e22;
        double detM;
        const bool finite = fabs(m00) < 1e30 && fabs(m01) < 1e30 && fabs(m02) < 1e30 && fabs(m11) < 1e30 && fabs(m12) < 1e30 && fabs(m22) < 1e30;
        if (finite && spd_det64(m00, m01, m02, m11, m12, m22, detM)) {
            const double trM = m00 + m11 + m22;
            const double e2 = (m00 * m11 - m01 * m01) + (m00 * m22 - m02 * m02) + (m11 * m22 - m12 * m12);   // >= lmax * lmid
            const double K = trM * e2 / detM * 1.000001;                 // >= tr(M) / lambda_min(M)
            const double u = 5.9604644775390625e-8;
            const double theta = 6.1 * u * K + 2.0 * u;
            const double G = -(log(detM) + log((double)det_p)) + 2e-6;
            if (theta <= 0.25 && fabs(G) <= 1.0) {
                const double smin = (thr2 * (1.0 + 2.0 * u) + G + (double)GSR_DET_TOL + 3.2 * theta / (1.0 - theta) + 4e-5) / (1.0 - theta);
                const double T1 = smin * (1.0 + theta) * (1.0 + theta) * (1.0 + 1e-5);
                if (T1 > 0.0 && T1 < 1e30) {
                    // Cholesky factor of M (upper), float64, rounded to float32
                    const double u00 = sqrt(m00), u01 = m01 / u00, u02 = m02 / u00;
                    const double t11 = m11 - u01 * u01, u11 = sqrt(t11), u12 = (m12 - u01 * u02) / u11;
                    const double t22 = m22 - u02 * u02 - u12 * u12, u22 = sqrt(t22);
                    if (t11 > 0.0 && t22 > 0.0) {
                        pr.u00 = (float)u00; pr.u01 = (float)u01; pr.u02 = (float)u02; pr.u11 = (float)u11; pr.u12 = (float)u12; pr.u22 = (float)u22;
                        pr.T1 = (float)(T1 * (1.0 + 2.0 * u));
                        pr.white = 1.0f;
                        if (ell) {
                            const double Tc = T1 * (1.0 + theta) * (1.0 + theta) * (1.0 + 1e-4);
                            const double fk = 1.0 - 4.0 * theta - 2e-3;                       // > 0 for theta <= 1/4... only just: see `ok`
                            const double im00 = 1.0 / m00;
                            const double k11 = (m11 - m01 * m01 * im00) * fk, k12 = (m12 - m01 * m02 * im00) * fk, k22 = (m22 - m02 * m02 * im00) * fk;
                            ec.k11 = (float)k11; ec.k12 = (float)k12; ec.k22 = (float)k22;
                            ec.kr = (float)(sqrt((k11 + k22) / fk) * 1.001);
                            ec.im00 = (float)(im00 * (1.0 + 4.0 * u));
                            ec.m01 = (float)m01; ec.m02 = (float)m02;
                            ec.T = (float)(Tc * (1.0 + 2.0 * u));
                            const bool ok = fk > 0.5 && k11 > 0.0 && k22 > 0.0 && k11 * k22 > k12 * k12 && ec.kr < FLT_MAX && ec.im00 < FLT_MAX && ec.T < FLT_MAX;
                            ec.on = ok ? 1.0f : 0.0f;
                            // extent of E along y / z = sqrt(T_clip (M^-1)_yy / zz), M^-1 from M itself
                            pr.ey = (float)(sqrt(fmax(0.0, Tc * (m00 * m22 - m02 * m02) / detM)) * 1.001);
                            pr.ez = (float)(sqrt(fmax(0.0, Tc * (m00 * m11 - m01 * m01) / detM)) * 1.001);
                        }
                    }
                }
            }
        }
    }
    pr.ec = ec;
}

// The stage-1 filter value |U d|^2 (make_filter): vc = {parent mean, u00, u01, u02, u11, u12, u22, T1}; nine fused multiply-adds
__device__ __forceinline__ float white_smd(const float (&vc)[11], float cx, float cy, float cz) {
    const float dx = cx - vc[0], dy = cy - vc[1], dz = cz - vc[2];
    const float y2 = vc[8] * dz;
    const float y1 = __builtin_fmaf(vc[6], dy, vc[7] * dz);
    const float y0 = __builtin_fmaf(vc[3], dx, __builtin_fmaf(vc[4], dy, vc[5] * dz));
    return __builtin_fmaf(y0, y0, __builtin_fmaf(y1, y1, y2 * y2));
}

// Parents per wave.  A wave takes up to SEL_NP consecutive slots of the processing order (neighbours on the Z-order curve) ONE AFTER
// THE OTHER and keeps the survivor ring and the third-stage queue ACROSS them: stage 2 and stage 3 run on full batches of 64
// whatever parent the entries belong to (a surfel-like parent leaves 57 survivors and 7 pairs -- alone it runs stage 2 at 52 % and
// stage 3 at 10 % of the lanes), and only the wave's last batches are partial.  A ring entry carries its parent's number k in the
// bits above the sorted position (positions are < 2^30: the stream packs `position << 2 | flags` into 32 bits), and what stages
// 2 / 3 need of parent k they read per lane from LDS (struct ParLds) instead of the scalar registers -- which also takes 17 dwords
// of every parent out of the SGPR file.  The rings are FIFO and the parents are scanned in turn, so every parent's pairs come out in
// exactly the order (and at exactly the places) the one-parent-per-wave form wrote them.
#define SEL_NP 4
#define SEL_TAG_SHIFT 30
#define SEL_TAG_MASK 0x3fffffffu

// ---- launchers (defined in hem_select.hip) ----------------------------------------------------------------------------------------
void launch_parent_prep(hipStream_t st, int P, const unsigned* plist, const float4* geo, const float* Rs, float kldThr, int ell, ParentRec* prec);
void launch_spans(hipStream_t st, const SelectArgs& sa);
// the light parents' launch on `st` and -- sa.heavy_blocks > 0 -- the heavy parents' queue beside it on `aux` (forked / joined by the two events)
int32_t launch_select(int mode, const SelectArgs& sa, hipStream_t st, hipStream_t aux, hipEvent_t ev_fork, hipEvent_t ev_join);
constexpr int SEL_WPB = 2;      // parents per workgroup of k_select (work per parent is heavy-tailed: small workgroups free their CU slot sooner)
void select_set_attributes();
int32_t select_profile(unsigned long long* out16, int32_t reset);

}  // namespace gsr
