// hem_select.hip -- the child selection of a HEM level: k_parent_prep (per-parent records), k_spans (capacities + row lists), k_select
// (candidate stream -> stage 1 -> stage 2 -> stage 3).  Its own translation unit since round 5: it is the one that is built WITHOUT
// MachineLICM (-mllvm -disable-machine-licm, __graft_entry__.EXTRA_FLAGS) -- k_select's body runs in a loop over the wave's parents,
// and the pass hoists a dozen constants and uniform float values (which live in VECTOR registers) out of it: 84 VGPRs instead of 72.
// Reference: src/cpp_ext/src/mixture.cpp:102-137 (selection), :140-164 (wL), include/gaussian.hpp:82-114.
#include "hem_select.h"

namespace gsr {

#ifdef GSR_SELECT_PROFILE
__device__ unsigned long long g_sel_prof[1024 * 16];       // 1024 copies (by workgroup): atomics on ONE address from 10^6 waves serialise
#endif
struct __attribute__((aligned(16))) ParLds {      // 20 dwords; the first 16 are the head of the parent's ParentRec as it lies in memory
    float4 r0;                                    // pm.x, pm.y, pm.z, pcol.x
    float4 r1;                                    // pcol.y, pcol.z, pinv.e00, pinv.e01
    float4 r2;                                    // pinv.e02, pinv.e11, pinv.e12, pinv.e22
    float4 r3;                                    // det_p, inv_det_p, pweight, R
    float R2;
    int js;
    int64_t base;                                 // FILL / SPARSE: where the parent's (this work item's) pairs go
};
static_assert(sizeof(ParLds) == 80, "ParLds is read as four float4 and two 8-byte words");

// Third-stage queue (per wave, in LDS): the accepted pairs wait here until 64 of them fill a wavefront, so that
// the likelihood (two expf, two sqrtf, two IEEE divisions) runs on full waves instead of the ~26 % of the lanes
// that pass the KL gate.  rel = the pair's rank among its parent's accepted pairs: it goes to ParLds::base + rel.
#define SEL_Q3CAP 128
struct Q3 {
    unsigned *j, *rel;             // j keeps the parent tag
    float *d2, *cd, *op, *det;
    int h, n;                      // head, fill (wave-uniform)
};

// wL_si = w_s * clamp(hemLikelihoodOpacity, FLT_MIN, 1e8)   (mixture.cpp:54-64,155-158) for `cnt` queued pairs
__device__ __forceinline__ void select_stage3(const SelectArgs& a, const ParLds* par, int lane, int cnt, Q3& q3) {
    if (lane < cnt) {
        const int k = (q3.h + lane) & (SEL_Q3CAP - 1);
        const unsigned e = q3.j[k];
        const ParLds* pl = par + (e >> SEL_TAG_SHIFT);
        const float distanceDiff = sqrtf(q3.d2[k]);
        const float cdiff = sqrtf(q3.cd[k]);                  // ColorDelta (gaussian.hpp:111-114); the queue holds its square
        const float distWeight = expf(-distanceDiff * distanceDiff / a.tau2);
        const float colorInfluence = expf(-cdiff * cdiff / a.tau2);
        const float L = distWeight * q3.op[k] * colorInfluence * sqrtf(q3.det[k]);
        const int64_t dst = pl->base + (int64_t)q3.rel[k];
        a.pair_child[dst] = e & SEL_TAG_MASK;
        a.pair_wl[dst] = pl->r3.z * ref_clamp(L, FLT_MIN, 1e8f);
    }
    q3.h = (q3.h + cnt) & (SEL_Q3CAP - 1);
    q3.n -= cnt;
}

// stage 2 on up to 64 queued survivors (lane < cnt holds one): radius test, colour gate, KL gate, parent rule -- the
// reference's decisions, every expression in its operand order; accepted pairs go to the third-stage queue.
// count_v: lane k holds the number of pairs parent k has had accepted so far (read and advanced here).
template <int MODE>
__device__ __forceinline__ void select_stage2(const SelectArgs& a, const ParLds* par, int lane, int cnt, const unsigned* q, int qh,
                                              unsigned& count_v, Q3& q3) {
    SEL_PROF_T(tp2);
    SEL_PROF_CNT(8, 1, lane); SEL_PROF_CNT(9, cnt, lane);
    bool acc = false;
    unsigned e = 0u;
    int k = 0;
    float d2 = 0.0f, cdiff = 0.0f, op = 0.0f, det_c = 0.0f;
    if (lane < cnt) {
        e = q[(qh + lane) & (SEL_QCAP - 1)];
        const int j = (int)(e & SEL_TAG_MASK);
        k = (int)(e >> SEL_TAG_SHIFT);
        const ParLds* pl = par + k;
        const float4 p0 = pl->r0, p1 = pl->r1;
        const float R2 = pl->R2;
        const float4* row = a.geo + 4 * (int64_t)j;
        float4 ca = row[0], cb = row[1], cc = row[2], cd = row[3];   // 64 contiguous bytes, all four up front: one round trip
        // (an empty statement the compiler cannot see through: without it the loads of the covariance and the determinant sink into the
        // branch behind the radius / colour gates -- a SECOND round trip per batch, +2 800 cycles, found in the phase profile)
        asm volatile("" : "+v"(ca.x), "+v"(ca.y), "+v"(ca.z), "+v"(ca.w), "+v"(cb.x), "+v"(cb.y), "+v"(cb.z), "+v"(cb.w),
                          "+v"(cc.x), "+v"(cc.y), "+v"(cc.z), "+v"(cc.w), "+v"(cd.x), "+v"(cd.y), "+v"(cd.z), "+v"(cd.w));
        const f3 cm = {ca.x, ca.y, ca.z};
        const f3 ccol = {cc.z, cc.w, cd.x};
        const f3 pm = {p0.x, p0.y, p0.z}, pcol = {p0.w, p1.x, p1.y};
        const f3 d = sub3(cm, pm);
        d2 = dot3(d, d);                                      // == dot(pm - cm, pm - cm) bit for bit (pointindex.cpp:137)
        const f3 dc = sub3(ccol, pcol);                       // ColorDelta(child, parent), gaussian.hpp:111-114
        cdiff = dot3(dc, dc);                                 // its square; sqrtf(x) > colorThr  <=>  x > colorThr2 (a.colorThr2, host)
        if (d2 < R2 && !(cdiff > a.colorThr2)) {              // radiusSearch (strict), mixture.cpp:122-124
            const float4 p2 = pl->r2, p3 = pl->r3;
            const s6 ccov = {cb.x, cb.y, cb.z, cb.w, cc.x, cc.y};
            const s6 pinv = {p1.z, p1.w, p2.x, p2.y, p2.z, p2.w};
            det_c = cd.w;
            op = cd.y;
            const float smd = dot3(d, mul6(pinv, d));         // gaussian.hpp:82-85
            const float tr = trace_prod6(pinv, ccov);
            const float s2 = smd + tr - 3.0f;                 // gaussian.hpp:106-109: 0.5f * (smd + tr - 3.0f - log(q))
            if (!kl_gate_rejects(s2, det_c, p3.x, p3.y, a.kldThr, a.logtab)) {      // mixture.cpp:126-129 (NaN passes)
                const bool child_is_parent = (__float_as_uint(ca.w) & 1u) != 0u;
                acc = !(child_is_parent && j != pl->js);      // mixture.cpp:131-133
            }
        }
    }
    const unsigned long long m = __ballot(acc);
    const int na = __popcll(m);
    SEL_PROF_CNT(15, na, lane);
    if (na == 0) { SEL_PROF_ADD(2, tp2, lane); return; }
    // the entries are in parent order (FIFO): the batch holds the parents k_lo .. k_hi, every one a contiguous run of lanes
    const int k_lo = __builtin_amdgcn_readfirstlane(k);
    const int k_hi = __builtin_amdgcn_readlane(k, __builtin_amdgcn_readfirstlane(cnt - 1));
    unsigned rel = 0u;
#pragma unroll
    for (int kk = 0; kk < SEL_NP; ++kk) {
        if (kk < k_lo || kk > k_hi) continue;                 // (uniform)
        const unsigned long long mk = __ballot(acc && k == kk);
        const unsigned ck = (unsigned)__builtin_amdgcn_readlane((int)count_v, kk);
        if (acc && k == kk) rel = ck + (unsigned)mbcnt64(mk, 0);
        if (lane == kk) count_v += (unsigned)__popcll(mk);
    }
    if (MODE == SEL_COUNT) { SEL_PROF_ADD(2, tp2, lane); return; }
    if (acc) {
        const int t = mbcnt64(m, q3.h + q3.n) & (SEL_Q3CAP - 1);
        q3.j[t] = e; q3.rel[t] = rel; q3.d2[t] = d2; q3.cd[t] = cdiff; q3.op[t] = op; q3.det[t] = det_c;
    }
    q3.n += na;
    __builtin_amdgcn_wave_barrier();
    if (q3.n >= 64) select_stage3(a, par, lane, 64, q3);
    __builtin_amdgcn_wave_barrier();
    SEL_PROF_ADD(2, tp2, lane);
}

// One thread per parent: the record the selection kernels read (see struct ParentRec).
__global__ __launch_bounds__(256) void k_parent_prep(int P, const unsigned* __restrict__ plist, const float4* __restrict__ geo,
                                                     const float* __restrict__ Rs, float kldThr, int ell, ParentRec* __restrict__ prec) {
    // a wave's 64 records leave as 10 KiB of contiguous memory (through LDS): a lane storing its own 160-byte record wrote 16
    // bytes of 64 different records per instruction
    __shared__ ParentRec s_pr[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int base = blockIdx.x * blockDim.x + (threadIdx.x & ~63); base < P; base += gridDim.x * blockDim.x) {
        const int p = base + lane;
        if (p < P) {
        ParentRec pr;
        pr.js = (int)plist[p];
        const float4* prow = geo + 4 * (int64_t)pr.js;
        const float4 pa = prow[0], pb = prow[1], pc = prow[2], pd = prow[3];
        pr.pm = {pa.x, pa.y, pa.z};
        pr.det_p = pd.w;
        pr.inv_det_p = 1.0f / pd.w;
        const s6 pcov = {pb.x, pb.y, pb.z, pb.w, pc.x, pc.y};
        pr.pcol = {pc.z, pc.w, pd.x};
        pr.pweight = pd.z;
        pr.pinv = inverse6(pcov, pr.det_p);
        pr.R = Rs[pr.js];
        pr.R2 = pr.R * pr.R;
        make_filter(pr.pinv, pr.det_p, kldThr, ell, (__float_as_uint(pa.w) & 2u) != 0u, pr);
        // R2 is NaN for a NaN radius and 0 for R = 0: `d2 < R2` is then never true -> no children.
        const bool pm_finite = fabsf(pr.pm.x) <= FLT_MAX && fabsf(pr.pm.y) <= FLT_MAX && fabsf(pr.pm.z) <= FLT_MAX;
        pr.active = (pr.R2 > 0.0f && pm_finite) ? 1 : 0;
        pr.selfq = (pr.active && (__float_as_uint(pa.w) & 2u)) ? 1 : 0;
        pr.rows = 0;
        s_pr[wv][lane] = pr;
        }
        __builtin_amdgcn_wave_barrier();
        const int nrec = P - base < 64 ? P - base : 64;
        const float4* src = reinterpret_cast<const float4*>(s_pr[wv]);
        float4* dst = reinterpret_cast<float4*>(prec + base);
        for (int t = lane; t < nrec * (int)(sizeof(ParentRec) / 16); t += 64) dst[t] = src[t];
        __builtin_amdgcn_wave_barrier();
    }
}

// One pass of a parent over its grid rows: IRR = false scans the cell-sorted components themselves and keeps the
// REGULAR ones; IRR = true scans the list of irregular components (ipos, addressed through irank at the cell
// boundaries).  Survivors of the stage-1 filter go to the LDS ring.
// Full batches of 64 survivors go through stage 2 from inside the scan (the ring holds the rest of a batch plus one group of chunks).
// (Tried on the way to four parents per wave: the scan only FILLING a 512-entry ring, suspended when it is full and resumed through
// the work items' [lo, hi) mechanism, the drain behind it -- it kept stage 2's registers out of the scan's when the kernel held 96
// VGPRs.  With the parent record in scalar registers both forms need 65, and this one is 5 % faster: no suspended scans (6 % of the
// parents recomputed a batch of row spans), half the ring.  profiles/archive/r04o_*.)
template <int MODE, bool IRR>
__device__ __forceinline__ void select_scan(const SelectArgs& a, const GridParams& g, const ParentRec& pr, const ParLds* par, unsigned ktag,
                                            const float (&vc)[11], int lane, unsigned long long* bits, unsigned* q, int& qh, int& qn, unsigned& count_v,
                                            Q3& q3, unsigned& cum, unsigned lo, unsigned hi, const int2* rl, int lcnt) {
    const f3 pm = pr.pm;
    const EllClip& ec = pr.ec;
    // rl != NULL: the spans of the parent's lcnt non-empty rows, in scan order, as k_spans left them (it computes every span anyway,
    // for the capacities): one 8-byte load per row instead of the box, ~80 instructions per row and two dependent look-ups in the
    // prefix table.  The candidates, their order and with it the pairs are the same either way (select_row_span made both).
    const bool listed = rl != nullptr;
    bool clip = false;
    int x0 = 0, x1 = 0, y0 = 0, z0 = 0, ny = 1, nrows = lcnt;
    float Ra2 = 0.0f;
    if (!listed) {
        // (uniform values, but float arithmetic is VALU work and its results would sit in vector registers for the whole scan: the box
        // goes back to the scalar file.  The compiler knows the values are uniform and folds a plain readfirstlane away -- leaving them
        // where they are; a zero it cannot see through, added to the bit pattern, keeps the instruction)
        int vz;
        asm volatile("v_mov_b32 %0, 0" : "=v"(vz));
        const auto uni = [vz](int v) { return __builtin_amdgcn_readfirstlane(v + vz); };
        const auto unif = [vz](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v) + vz)); };
        const float Ra = unif(fabsf(pr.R) * 1.00001f + g.slack);       // conservative search extent
        clip = !IRR && ec.on != 0.0f;
        // rows: the sphere's box, cut down to the pre-reject ellipsoid's box when the rows are clipped to it (rows beyond its y / z
        // extent would come out empty one by one)
        const float Ry = clip ? fminf(Ra, pr.ey + g.slack) : Ra, Rz = clip ? fminf(Ra, pr.ez + g.slack) : Ra;
        x0 = uni(cell_of(pm.x - Ra, g.ox, g.inv_c, g.gx)); x1 = uni(cell_of(pm.x + Ra, g.ox, g.inv_c, g.gx));
        y0 = uni(cell_of(pm.y - Ry, g.oy, g.inv_c, g.gy));
        const int y1 = uni(cell_of(pm.y + Ry, g.oy, g.inv_c, g.gy));
        z0 = uni(cell_of(pm.z - Rz, g.oz, g.inv_c, g.gz));
        const int z1 = uni(cell_of(pm.z + Rz, g.oz, g.inv_c, g.gz));
        ny = y1 - y0 + 1;
        nrows = ny * (z1 - z0 + 1);
        Ra2 = unif(Ra * Ra);
    }
    const bool white = !IRR && pr.white != 0.0f;                // wave-uniform
    SEL_PROF_CNT(10, nrows, lane);
    for (int rb = 0; rb < nrows; rb += 64) {
        SEL_PROF_T(tpr);
        SEL_PROF_CNT(11, 1, lane);
        const int r = rb + lane;
        int s = 0, len = 0;
        if (r < nrows) {
            if (listed) { const int2 e = rl[r]; s = e.x; len = e.y; }
            else select_row_span<IRR>(a, g, pm, ec, clip, Ra2, x0, x1, y0 + r % ny, z0 + r / ny, s, len);
        }
        len = len > 0 ? len : 0;
        const unsigned long long nz_m = __ballot(len > 0);
        if (nz_m == 0ull) continue;
        // flattened candidate space of the batch: row r covers flat positions [pre, pre + len)
        const int incl = wave_incl_scan(len);
        const int pre = incl - len;
        const int total = __builtin_amdgcn_readlane(incl, 63);
        // this work item's part [lo, hi) of the parent's flat candidate space (cum = candidates of the batches before this
        // one; a parent that is not split has lo = 0, hi = 2^32 - 1)
        const unsigned cum0 = cum;
        cum += (unsigned)total;
        if (hi <= cum0 || lo >= cum) continue;
        const int b_lo = lo > cum0 ? (int)(lo - cum0) : 0;
        const int b_hi = hi - cum0 < (unsigned)total ? (int)(hi - cum0) : total;
        // (start - prefix) of the q-th non-empty row goes to lane q (the empty rows take the lanes behind them: a permutation)
        const int nrb = __popcll(nz_m);
        const int below = mbcnt64(nz_m, 0);
        const int dst = len > 0 ? below : nrb + (lane - below);
        const int delta = __builtin_amdgcn_ds_permute(dst << 2, s - pre);
        SEL_PROF_ADD(1, tpr, lane);
        SEL_PROF_CNT(12, nrb, lane); SEL_PROF_CNT(13, total, lane);
        for (int seg0 = b_lo; seg0 < b_hi; seg0 += SEL_MCAP) {
            const int rel = pre - seg0;
            const bool mine = len > 0 && rel >= 0 && rel < SEL_MCAP;
            if (mine) atomicOr(&bits[rel >> 6], 1ull << (rel & 63));
            int rows_before = __popcll(__ballot(len > 0 && rel < 0));      // rows that start before this segment (uniform)
            __builtin_amdgcn_wave_barrier();
            const int seg_end = b_hi < seg0 + SEL_MCAP ? b_hi : seg0 + SEL_MCAP;
            for (int t0 = seg0; t0 < seg_end; t0 += 64 * SEL_U) {
                SEL_PROF_CNT(14, 1, lane);
                SEL_PROF_T(tps);
                float4 ca[SEL_U];
                int jj[SEL_U];
                int left[SEL_U];                                              // active lanes of each chunk (uniform)
#pragma unroll
                for (int u = 0; u < SEL_U; ++u) {
                    const int c0 = t0 + 64 * u;                               // uniform
                    // row of candidate c0 + lane = (row starts at flat positions <= c0 + lane) - 1
                    const unsigned long long word = uniform64(bits[(c0 - seg0) >> 6]);     // broadcast read; the words behind the segment are zero
                    const int row = mbcnt64(word >> 1, rows_before - 1 + (int)(word & 1ull));
                    rows_before += __popcll(word);
                    left[u] = seg_end - c0;
                    const int k = c0 + lane + __builtin_amdgcn_ds_bpermute(row << 2, delta);
                    // the inactive lanes behind the batch's last candidate read on past the last row: the sorted A array is
                    // padded by SEL_PAD entries, so the load stays unconditional (a load under a branch would make hipcc
                    // wait vmcnt(0) after each one instead of overlapping the SEL_U loads)
                    if (IRR) { jj[u] = lane < left[u] ? (int)a.ipos[k] : pr.js; ca[u] = a.A[jj[u]]; }
                    else { jj[u] = k; ca[u] = a.Ac[k]; }                      // pass A: jj becomes the sorted position below
                }
#pragma unroll
                for (int u = 0; u < SEL_U; ++u) {
                    if (left[u] <= 0) continue;
                    const f3 cm = {ca[u].x, ca[u].y, ca[u].z};
                    bool in;
                    if (white) {                                              // regular parent, regular children
                        // vc = {pm, U, T1} in VECTOR registers: a VALU instruction with an SGPR operand issues at half rate
                        in = !(white_smd(vc, cm.x, cm.y, cm.z) > vc[9]);
                    } else {                                                  // the reference's radius test (pointindex.cpp:137)
                        const f3 dq = sub3(pm, cm);
                        in = dot3(dq, dq) < pr.R2;
                    }
                    if (!IRR) {
                        in = in && (__float_as_uint(ca[u].w) & 2u);           // irregular children belong to pass B
                        jj[u] = (int)(__float_as_uint(ca[u].w) >> 2);         // the stream carries the sorted position
                    } else {
                        // the parent rule (mixture.cpp:131-133): a component that is a parent itself is claimed by no parent but
                        // itself.  Pass A's stream holds no parents at all; the irregular list does
                        in = in && (!(__float_as_uint(ca[u].w) & 1u) || jj[u] == pr.js);
                    }
                    in = in && lane < left[u];
                    const unsigned long long m = __ballot(in);
                    if (m == 0ull) continue;
                    if (in) q[mbcnt64(m, qh + qn) & (SEL_QCAP - 1)] = (unsigned)jj[u] | ktag;
                    qn += __popcll(m);
                }
                __builtin_amdgcn_wave_barrier();
                SEL_PROF_ADD(4, tps, lane);
                // drain: stage 2 appears ONCE in the code (not once per unrolled u), on full batches of 64
                while (qn >= 64) {
                    select_stage2<MODE>(a, par, lane, 64, q, qh, count_v, q3);
                    qh = (qh + 64) & (SEL_QCAP - 1);
                    qn -= 64;
                }
                __builtin_amdgcn_wave_barrier();
            }
            if (mine) bits[rel >> 6] = 0ull;                                  // leave the mask clean for the next segment / batch
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// What stages 2 / 3 need of parent p, from its record in memory to the wave's LDS slot (ONE lane calls this per parent; the slot is
// free: a wave's parents take different slots, a work item's rings are flushed before the next one starts).  (Lane 0 writing it from
// the scalar copy of the record inside select_parent, measured: +5 % at four parents per wave -- 17 more live SGPRs and 20 v_mov.)
__device__ __forceinline__ void select_fill_par(const SelectArgs& a, int p, int64_t base, ParLds* slot) {
    const float4* r = reinterpret_cast<const float4*>(a.prec + p);
    const float4 r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3];
    slot->r0 = r0; slot->r1 = r1; slot->r2 = r2; slot->r3 = r3;
    slot->R2 = a.prec[p].R2; slot->js = a.prec[p].js; slot->base = base;
}

// One work item: parent p (the wave's k-th), part [lo, hi) of its flat candidate space (the whole parent: 0, 2^32 - 1); its pairs
// go to par[k].base on (FILL / SPARSE; select_fill_par has filled par[k]).  Survivors and accepted pairs may stay behind in the
// rings: select_flush ends a wave's (a work item's) run.  The number of accepted pairs accumulates in lane k of count_v.
template <int MODE>
__device__ __forceinline__ void select_parent(const SelectArgs& a, const GridParams& g, int p, int k, unsigned lo, unsigned hi, int lane,
                                              const ParLds* par, unsigned* q, int& qh, int& qn, unsigned long long* bits, unsigned& count_v, Q3& q3) {
    SEL_PROF_T(tp0);
    // The record lives in SGPRs.  As inline assembly: this body runs in a loop behind the wave's own stores (pcnt, the pairs), the
    // compiler cannot tell that they leave prec[] alone, and a load that "may be clobbered" is not made a scalar load however uniform
    // its address -- it became ten global_load_dwordx4 with 64 lanes reading the same 160 bytes, and 40 v_readfirstlane behind them
    // (SQ_INSTS_VMEM_RD +16 M per 5 M level, profiles/archive/r04o_*).
    ParentRec pr;
    {
        typedef unsigned u16v __attribute__((ext_vector_type(16)));
        typedef unsigned u8v __attribute__((ext_vector_type(8)));
        u16v w0, w1; u8v w2;
        const ParentRec* rp = a.prec + p;
        asm volatile("s_load_dwordx16 %0, %3, 0x0\n\ts_load_dwordx16 %1, %3, 0x40\n\ts_load_dwordx8 %2, %3, 0x80\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(w0), "=&s"(w1), "=&s"(w2) : "s"(rp) : "memory");
        unsigned raw[40];
        static_assert(sizeof(raw) == sizeof(ParentRec), "the three scalar loads cover the record");
#pragma unroll
        for (int i = 0; i < 16; ++i) { raw[i] = w0[i]; raw[16 + i] = w1[i]; }
#pragma unroll
        for (int i = 0; i < 8; ++i) raw[32 + i] = w2[i];
        __builtin_memcpy(&pr, raw, sizeof(pr));
    }
    // the constants of the stage-1 filter in vector registers (v_mov from the SGPRs once per parent)
    float vc[11];
    {
        const float src[11] = {pr.pm.x, pr.pm.y, pr.pm.z, pr.u00, pr.u01, pr.u02, pr.u11, pr.u12, pr.u22, pr.T1, 0.0f};
#pragma unroll
        for (int i = 0; i < 11; ++i) asm volatile("v_mov_b32 %0, %1" : "=v"(vc[i]) : "s"(src[i]));
    }
    const unsigned ktag = (unsigned)k << SEL_TAG_SHIFT;
    unsigned cum = 0;                   // flat candidates of the batches behind the scan
    if (pr.selfq) {                     // the parent itself: flat candidate 0, straight into the survivor ring (stage 2 decides)
        cum = 1u;
        if (lo == 0u) {
            if (lane == 0) q[(qh + qn) & (SEL_QCAP - 1)] = (unsigned)pr.js | ktag;
            qn += 1;
        }
        __builtin_amdgcn_wave_barrier();
        // The ring holds 64 + one group of chunks (SEL_QCAP): the scan's drain keeps qn < 64 behind every group, but a parent WITHOUT a
        // non-empty candidate row never gets there -- several isolated parents in a row (SEL_NP per wave) each add their own entry, and
        // a following parent's first group of SEL_U * 64 survivors would wrap onto the oldest entries (an accepted pair lost silently,
        // ADVICE r04).  So a full batch goes through stage 2 right here: qn <= 63 again in front of every scan.
        if (qn >= 64) {
            select_stage2<MODE>(a, par, lane, 64, q, qh, count_v, q3);
            qh = (qh + 64) & (SEL_QCAP - 1);
            qn -= 64;
        }
    }
    if (pr.active) {
        // pass A: the regular children (rows clipped to the parent's Mahalanobis ellipsoid when it is regular);
        // pass B: the irregular children, which the pre-reject never applies to (rows clipped to the sphere only)
        // the row lists k_spans left for this parent (bit 31 of pr.rows), per pass: a count of at most SEL_ROWS is a complete list
        const int cA = pr.rows & 0xff, cB = (pr.rows >> 8) & 0xff;
        const int2* rlp = (a.rowlist != nullptr && pr.rows < 0) ? a.rowlist + (int64_t)p * (2 * SEL_ROWS) : nullptr;
        select_scan<MODE, false>(a, g, pr, par, ktag, vc, lane, bits, q, qh, qn, count_v, q3, cum, lo, hi,
                                 rlp != nullptr && cA <= SEL_ROWS ? rlp : nullptr, cA);
        if (a.n_irr > 0) select_scan<MODE, true>(a, g, pr, par, ktag, vc, lane, bits, q, qh, qn, count_v, q3, cum, lo, hi,
                                                 rlp != nullptr && cB <= SEL_ROWS ? rlp + SEL_ROWS : nullptr, cB);
    }
    SEL_PROF_ADD(0, tp0, lane);
    SEL_PROF_CNT(7, 1, lane);
}

// The partial batches at the end of a wave's run: the survivors left in the ring, then the accepted pairs left in the queue.
template <int MODE>
__device__ __forceinline__ void select_flush(const SelectArgs& a, const ParLds* par, int lane, const unsigned* q, int& qh, int& qn,
                                             unsigned& count_v, Q3& q3) {
    while (qn > 0) {                    // (the parent's own entry may have made it 64)
        const int c = qn < 64 ? qn : 64;
        select_stage2<MODE>(a, par, lane, c, q, qh, count_v, q3);
        qh = (qh + c) & (SEL_QCAP - 1);
        qn -= c;
    }
    SEL_PROF_T(tp3);
    if ((MODE == SEL_FILL || MODE == SEL_SPARSE) && q3.n > 0) select_stage3(a, par, lane, q3.n, q3);
    SEL_PROF_ADD(3, tp3, lane);
    __builtin_amdgcn_wave_barrier();
}

// WPB = wavefronts per workgroup.  QUEUE = false: a.np (1 ... SEL_NP) consecutive light parents of the processing order per wave,
// one after the other with the rings kept across them (the heavy slots at the order's head are skipped when a.heavy_blocks > 0).
// QUEUE = true, launched beside it on a second stream with
// a.heavy_blocks workgroups: every wave serves the queue of heavy work items (item <its index> first, then it pulls).
// Two kernels rather than one: the item loop's uniform state does not fit the scalar registers beside the parent record,
// and the spills would cost the light parents, 99.7 % of the work, two waves per SIMD.
template <int MODE, int WPB, bool QUEUE>
__global__ __launch_bounds__(64 * WPB) void k_select(SelectArgs a) {
    __shared__ unsigned s_q[WPB][SEL_QCAP];
    __shared__ unsigned s_q3u[WPB][2][SEL_Q3CAP];
    __shared__ float s_q3f[WPB][4][SEL_Q3CAP];
    __shared__ unsigned long long s_bits[WPB][SEL_MCAP / 64 + SEL_U];
    __shared__ ParLds s_par[WPB][SEL_NP];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    a.logtab = k_logf_tab;                                  // (the exact logf is a rare path: its table stays in constant memory, 256 bytes of LDS less)
    if (lane < SEL_MCAP / 64 + SEL_U) s_bits[wv][lane] = 0ull;
    unsigned* q = s_q[wv];
    ParLds* par = s_par[wv];
    Q3 q3 = {s_q3u[wv][0], s_q3u[wv][1], s_q3f[wv][0], s_q3f[wv][1], s_q3f[wv][2], s_q3f[wv][3], 0, 0};
    __builtin_amdgcn_wave_barrier();
    const GridParams g = *a.gp;
    int qh = 0, qn = 0;                 // survivor ring: head and fill (uniform)

    if constexpr (QUEUE) {
        const int n_items = a.hq[0];
        const unsigned part = *a.part_p;
        int item = (int)blockIdx.x * WPB + wv;
        while (item < n_items) {
            const uint2 it = a.hitem[item];
            const int p = __builtin_amdgcn_readfirstlane((int)a.porder[it.x]);
            const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(it.y * part));
            const unsigned hi = lo + part > lo ? lo + part : 0xffffffffu;
            int64_t base = 0;
            if (MODE == SEL_SPARSE) base = a.poff[p] + lo;          // accepted <= candidates of the part: the parts cannot collide
            if (MODE == SEL_SPARSE && a.cap_pairs && (unsigned long long)base + part > a.cap_pairs) {      // beyond the buffers: no pairs, the level reruns
                if (lane == 0) { a.part_cnt[item] = 0u; *a.abort_p = 1; }
                int nxt0 = 0;
                if (lane == 0) nxt0 = atomicAdd(&a.hq[1], 1);
                item = __builtin_amdgcn_readfirstlane(nxt0);
                continue;
            }
            if (MODE == SEL_FILL) {
                base = a.poff[p];
                for (int k = a.hfirst[p]; k < item; ++k) base += a.part_cnt[k];
            }
            unsigned count_v = 0u;
            if (lane == 0) select_fill_par(a, p, base, par);
            __builtin_amdgcn_wave_barrier();
            select_parent<MODE>(a, g, p, 0, lo, hi, lane, par, q, qh, qn, s_bits[wv], count_v, q3);
            select_flush<MODE>(a, par, lane, q, qh, qn, count_v, q3);
            if (lane == 0 && (MODE == SEL_COUNT || MODE == SEL_SPARSE)) {
                a.part_cnt[item] = count_v;
                atomicAdd(&a.pcnt[p], count_v);                     // integer sum: the order of the parts does not matter
            }
            int nxt = 0;
            if (lane == 0) nxt = atomicAdd(&a.hq[1], 1);
            item = __builtin_amdgcn_readfirstlane(nxt);
            __builtin_amdgcn_wave_barrier();
        }
    } else {
        SEL_PROF_T(tpk);
        const int np = a.np;
        const int ppb = WPB * np;                                   // parents per workgroup
        const int nheavy = a.nheavy ? *a.nheavy : 0;
        const int nblk = (a.P + ppb - 1) / ppb;
        const int hb = ((nheavy + ppb - 1) / ppb + 7) & ~7;
        const int bid = block_slot((int)blockIdx.x, nblk, hb < nblk ? hb : nblk, a.xcd);
        if (bid < 0) return;
        const int slot0 = (bid * WPB + wv) * np;
        // lane k < np looks at the wave's k-th slot: whose parent it is, whether it is this launch's, and fills its LDS record --
        // nothing of this stays in the scalar registers while the parents are scanned
        int p_v = -1;                                               // lane k: the wave's k-th parent (-1: none, or not this launch's)
        if (lane < np) {
            const int slot = slot0 + lane;
            if (slot < a.P && !(a.heavy_blocks > 0 && slot < nheavy)) {      // (a heavy parent: the queue has it)
                const int p = a.porder ? (int)a.porder[slot] : slot;
                if (p < a.own_lo || p >= a.own_hi) {                // another rank's parent: no work, no pairs
                    if (MODE == SEL_COUNT || MODE == SEL_SPARSE) a.pcnt[p] = 0u;
                } else {
                    const int64_t base = (MODE == SEL_FILL || MODE == SEL_SPARSE) ? a.poff[p] : 0;
                    if (MODE == SEL_SPARSE && a.cap_pairs && (unsigned long long)base + a.pcap[p] > a.cap_pairs) {
                        a.pcnt[p] = 0u; *a.abort_p = 1;             // the segment would end beyond the buffers: no pairs, the level reruns
                    } else {
                        p_v = p;
                        select_fill_par(a, p, base, par + lane);
                    }
                }
            }
        }
        unsigned long long todo = __ballot(p_v >= 0);
        unsigned count_v = 0u;
        __builtin_amdgcn_wave_barrier();
        SEL_PROF_ADD(5, tpk, lane);
        while (todo != 0ull) {
            const int k = __builtin_ctzll(todo);
            todo &= todo - 1ull;
            const int p = __builtin_amdgcn_readlane(p_v, k);
            select_parent<MODE>(a, g, p, k, 0u, 0xffffffffu, lane, par, q, qh, qn, s_bits[wv], count_v, q3);
        }
        SEL_PROF_T(tpf);
        select_flush<MODE>(a, par, lane, q, qh, qn, count_v, q3);
        SEL_PROF_ADD(6, tpf, lane);
        if (p_v >= 0 && (MODE == SEL_COUNT || MODE == SEL_SPARSE)) a.pcnt[p_v] = count_v;      // no global atomics: totals come from the scans
    }
}

// Capacities (candidates every parent's passes will scan), with 16 lanes per parent instead of a wavefront: the pass has no
// candidate work, so its cost is the per-parent set-up, which four parents per wavefront share.  Same row spans as k_select
// by construction (select_row_span).  (Measured in round 4: three rows per lane with their table look-ups issued together -- 0.30
// against 0.25 ms on the isotropic 5 M level, 0.58 against 0.58 on the surfel one: the pass is bound by the rows' arithmetic, ~80
// instructions each for 15 / 68 rows per parent, not by the look-ups.)
__global__ __launch_bounds__(256) void k_spans(SelectArgs a) {
    const int sub = threadIdx.x & 15;
    const int p = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4);
    if (p >= a.P) return;
    unsigned long long scanned = 0;
    int selfq = 0;
    int nA = 0, nB = 0;                                         // non-empty rows of the two passes
    if (p >= a.own_lo && p < a.own_hi) {
        const GridParams g = *a.gp;
        // the record's fields with six 16-byte loads issued together (field by field the compiler fetched `active` first, waited, then
        // the rest one by one behind the branches that use them: four dependent round trips in a kernel that is nothing but latency)
        const float4* rq = reinterpret_cast<const float4*>(a.prec + p);
        float4 q0 = rq[0], q3 = rq[3], q6 = rq[6], q7 = rq[7], q8 = rq[8], q9 = rq[9];
        asm volatile("" : "+v"(q0.x), "+v"(q0.y), "+v"(q0.z), "+v"(q3.w), "+v"(q6.y), "+v"(q6.z), "+v"(q6.w), "+v"(q7.x), "+v"(q7.y), "+v"(q7.z), "+v"(q7.w),
                          "+v"(q8.x), "+v"(q8.y), "+v"(q8.w), "+v"(q9.x), "+v"(q9.y), "+v"(q9.z));
        static_assert(offsetof(ParentRec, R) == 60 && offsetof(ParentRec, ec) == 100 && offsetof(ParentRec, active) == 140 && offsetof(ParentRec, ey) == 148,
                      "k_spans reads the record by offsets");
        const f3 pm = {q0.x, q0.y, q0.z};
        const EllClip ec = {q6.y, q6.z, q6.w, q7.x, q7.y, q7.z, q7.w, q8.x, q8.y};
        struct { float R, ey, ez; int active; } rv = {q3.w, q9.y, q9.z, __float_as_int(q8.w)};
        const auto* rp = &rv;
        selfq = __float_as_int(q9.x);
        // The non-empty spans go to k_select in scan order (ascending row): in one trip of the loops below the parent's 16 lanes hold 16
        // consecutive rows, so a span's place is the count so far plus the non-empty rows on the lanes below it.
        const unsigned gsh = (unsigned)(threadIdx.x & 48);          // this parent's 16 lanes inside the wave's ballot
        const auto keep_span = [&](int pass, int s, int len, int& n) {
            const unsigned gm = (unsigned)(__ballot(len > 0) >> gsh) & 0xffffu;
            if (len > 0 && a.rowlist) {
                const int pos = n + __popc(gm & ((1u << sub) - 1u));
                if (pos < SEL_ROWS) a.rowlist[((int64_t)p * 2 + pass) * SEL_ROWS + pos] = make_int2(s, len);
            }
            n += __popc(gm);
        };
        if (rp->active) {
            const float Ra = fabsf(rp->R) * 1.00001f + g.slack;
            const float Ra2 = Ra * Ra;
            const int x0 = cell_of(pm.x - Ra, g.ox, g.inv_c, g.gx), x1 = cell_of(pm.x + Ra, g.ox, g.inv_c, g.gx);
            {   // pass A (the same rows as select_scan<.., false>)
                const bool clip = ec.on != 0.0f;
                const float Ry = clip ? fminf(Ra, rp->ey + g.slack) : Ra, Rz = clip ? fminf(Ra, rp->ez + g.slack) : Ra;
                const int y0 = cell_of(pm.y - Ry, g.oy, g.inv_c, g.gy), y1 = cell_of(pm.y + Ry, g.oy, g.inv_c, g.gy);
                const int z0 = cell_of(pm.z - Rz, g.oz, g.inv_c, g.gz), z1 = cell_of(pm.z + Rz, g.oz, g.inv_c, g.gz);
                const int ny = y1 - y0 + 1, nrows = ny * (z1 - z0 + 1);
                for (int r = sub; r < nrows; r += 16) {
                    int s = 0, len = 0;
                    select_row_span<false>(a, g, pm, ec, clip, Ra2, x0, x1, y0 + r % ny, z0 + r / ny, s, len);
                    scanned += (unsigned long long)(len > 0 ? len : 0);
                    keep_span(0, s, len, nA);
                }
            }
            if (a.n_irr > 0) {      // pass B: the irregular list over the sphere's rows
                const int y0 = cell_of(pm.y - Ra, g.oy, g.inv_c, g.gy), y1 = cell_of(pm.y + Ra, g.oy, g.inv_c, g.gy);
                const int z0 = cell_of(pm.z - Ra, g.oz, g.inv_c, g.gz), z1 = cell_of(pm.z + Ra, g.oz, g.inv_c, g.gz);
                const int ny = y1 - y0 + 1, nrows = ny * (z1 - z0 + 1);
                for (int r = sub; r < nrows; r += 16) {
                    int s = 0, len = 0;
                    select_row_span<true>(a, g, pm, ec, false, Ra2, x0, x1, y0 + r % ny, z0 + r / ny, s, len);
                    scanned += (unsigned long long)(len > 0 ? len : 0);
                    keep_span(1, s, len, nB);
                }
            }
        }
    }
    for (int o = 8; o > 0; o >>= 1) scanned += __shfl_xor(scanned, o);
    if (selfq) scanned += 1ull;                                 // the parent itself is flat candidate 0
    if (sub == 0) {
        a.pcap[p] = (unsigned)(scanned > 0xffffffffull ? 0xffffffffull : scanned);
        // (the record is this kernel's input and k_select's; the count rides in it so that k_select gets it with the record's scalar loads)
        if (a.rowlist) const_cast<ParentRec*>(a.prec)[p].rows = (int)(0x80000000u | (unsigned)(nA > SEL_ROWS ? 0xff : nA) | ((unsigned)(nB > SEL_ROWS ? 0xff : nB) << 8));
    }
}

void launch_parent_prep(hipStream_t st, int P, const unsigned* plist, const float4* geo, const float* Rs, float kldThr, int ell, ParentRec* prec) {
    hipLaunchKernelGGL(k_parent_prep, dim3(stride_grid(P)), dim3(256), 0, st, P, plist, geo, Rs, kldThr, ell, prec);
}
void launch_spans(hipStream_t st, const SelectArgs& sa) {
    hipLaunchKernelGGL(k_spans, dim3(ceil_div(sa.P, 16)), dim3(256), 0, st, sa);      // candidates scanned per parent
}
template <int MODE>
static int32_t launch_select_mode(const SelectArgs& sa, hipStream_t st, hipStream_t aux, hipEvent_t ev_fork, hipEvent_t ev_join) {
    constexpr int WPB = SEL_WPB;
    if (sa.heavy_blocks) {          // the queue-serving kernel runs beside the light parents' on the context's second stream
        GSR_HIP(hipEventRecord(ev_fork, st)); GSR_HIP(hipStreamWaitEvent(aux, ev_fork, 0));
        hipLaunchKernelGGL((k_select<MODE, WPB, true>), dim3(sa.heavy_blocks), dim3(64 * WPB), 0, aux, sa);
        GSR_HIP(hipEventRecord(ev_join, aux));
    }
    hipLaunchKernelGGL((k_select<MODE, WPB, false>), dim3(8 * ceil_div(ceil_div(sa.P, WPB * sa.np), 8)), dim3(64 * WPB), 0, st, sa);
    if (sa.heavy_blocks) GSR_HIP(hipStreamWaitEvent(st, ev_join, 0));
    return GSR_OK;
}
int32_t launch_select(int mode, const SelectArgs& sa, hipStream_t st, hipStream_t aux, hipEvent_t ev_fork, hipEvent_t ev_join) {
    if (mode == SEL_SPARSE) return launch_select_mode<SEL_SPARSE>(sa, st, aux, ev_fork, ev_join);
    if (mode == SEL_COUNT) return launch_select_mode<SEL_COUNT>(sa, st, aux, ev_fork, ev_join);
    return launch_select_mode<SEL_FILL>(sa, st, aux, ev_fork, ev_join);
}
void select_set_attributes() {}
int32_t select_profile(unsigned long long* out16, int32_t reset) {
#ifdef GSR_SELECT_PROFILE
    static unsigned long long h[1024 * 16];
    if (out16) {
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_sel_prof), sizeof(h)) != hipSuccess) return GSR_E_HIP;
        for (int k = 0; k < 16; ++k) { out16[k] = 0; for (int b = 0; b < 1024; ++b) out16[k] += h[b * 16 + k]; }
    }
    if (reset) { memset(h, 0, sizeof(h)); if (hipMemcpyToSymbol(HIP_SYMBOL(g_sel_prof), h, sizeof(h)) != hipSuccess) return GSR_E_HIP; }
    return GSR_OK;
#else
    (void)out16; (void)reset;
    return GSR_E_INVALID;
#endif
}

}  // namespace gsr
