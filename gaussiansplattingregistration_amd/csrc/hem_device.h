// hem_device.h -- device-side structures and lane helpers shared by the translation units of the HEM level (hem.hip, hem_select.hip).
#pragma once
#include "gsr_common.h"
#include "gsr_math.h"

#include <float.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

namespace gsr {

// ------------------------------------------------------------------------------------------------
// device-side structures
// ------------------------------------------------------------------------------------------------
struct GridParams {
    float ox, oy, oz;      // origin (bbox min of the finite points)
    float c, inv_c;        // cell edge and its reciprocal
    float slack;           // absolute slack used when culling rows (covers float rounding of cell_of)
    int gx, gy, gz;        // grid dimensions
    int ncells;
};

__device__ __forceinline__ unsigned enc_f(float f) {      // order-preserving float -> uint
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __host__ inline float dec_f(unsigned u) {
    unsigned v = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    float f;
    memcpy(&f, &v, 4);
    return f;
}

// Monotone non-decreasing map coordinate -> cell index in [0, g-1]; NaN -> 0.
__device__ __forceinline__ int cell_of(float v, float o, float inv_c, int g) {
    float t = (v - o) * inv_c;
    t = fminf(fmaxf(t, 0.0f), (float)(g - 1));
    return (int)t;
}

// XCD-aware block remap (MI355X: 8 XCDs, each with its own 4 MiB L2; blocks are dealt round-robin over
// the XCDs).  Blocks b, b+8, b+16, ... share an XCD, so give them CONSECUTIVE logical tiles: every XCD
// then sweeps its own contiguous run of cell-sorted parents and the children they share stay in that
// XCD's L2.  Launch with a grid of 8*ceil(nblk/8) blocks; returns -1 for the padding blocks.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int chunk = (nblk + 7) >> 3;
    const int t = (bid & 7) * chunk + (bid >> 3);
    return t < nblk ? t : -1;
}

// Processing slot of a block.  The first `hb` blocks (heavy parents, launched first) keep the natural
// order; the remaining blocks (light parents in Z-order) are dealt to the XCDs in contiguous chunks so
// that every XCD's L2 serves one compact 3-D region.  hb is a multiple of 8, so (bid - hb) & 7 is still
// the XCD group of the block.  Returns -1 for padding blocks.
__device__ __forceinline__ int block_slot(int bid, int nblk, int hb, int xcd) {
    if (!xcd) return bid < nblk ? bid : -1;
    if (bid < hb) return bid < nblk ? bid : -1;
    const int t = xcd_remap(bid - hb, nblk - hb);
    return t < 0 ? -1 : hb + t;
}

__device__ __forceinline__ float wave_min(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// Sums across lanes WITHOUT the LDS crossbar (a ds_bpermute costs a SIMD 24 cycles of LDS-pipe issue, scripts/micro/
// valu_issue.hip; the 14 butterfly sums of the first M-step were 84 of them per parent): rotations inside a row of 16
// lanes by DPP, rows by gfx950's v_permlane16_swap / v_permlane32_swap.  Every lane must be active.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float swap16_sum(float v) {          // v[l] + v[l ^ 16]
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}
__device__ __forceinline__ float swap32_sum(float v) {          // v[l] + v[l ^ 32]
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}
// the row-local part of class_sum<1>: every lane ends with the total of its row of 16 lanes (the same rotations in the same order)
__device__ __forceinline__ float row_sum16(float v) {
    v += dpp_f<0x128>(v);
    v += dpp_f<0x124>(v);
    v += dpp_f<0x122>(v);
    v += dpp_f<0x121>(v);
    return v;
}
// the row-local part of class_sum<G>: the total of the lanes l' == l (mod G) of the lane's own row
template <int G>
__device__ __forceinline__ float row_class_sum(float v) {
    if (G <= 8) v += dpp_f<0x128>(v);
    if (G <= 4) v += dpp_f<0x124>(v);
    if (G <= 2) v += dpp_f<0x122>(v);
    if (G <= 1) v += dpp_f<0x121>(v);
    return v;
}
// sum over the lanes l' == l (mod G), G a power of two: every lane ends with the total of its residue class
template <int G>
__device__ __forceinline__ float class_sum(float v) {
    if (G <= 8) v += dpp_f<0x128>(v);        // row_ror:8
    if (G <= 4) v += dpp_f<0x124>(v);        // row_ror:4
    if (G <= 2) v += dpp_f<0x122>(v);        // row_ror:2
    if (G <= 1) v += dpp_f<0x121>(v);        // row_ror:1
    if (G <= 16) v = swap16_sum(v);
    if (G <= 32) v = swap32_sum(v);
    return v;
}

// TWO sums per swap: the swap instructions exchange halves (rows) between two registers, so the addition behind one swap folds a
// DIFFERENT value in each half of the wave.  pack32_sum: lanes 0..31 end with a[l] + a[l + 32], lanes 32..63 with b[l - 32] + b[l];
// pack16_sum: row 0 ends with a's rows 0 + 1, row 1 with b's rows 0 + 1, row 2 with a's rows 2 + 3, row 3 with b's rows 2 + 3.
// (v_permlane32_swap: the upper half of the first operand <-> the lower half of the second; v_permlane16_swap: the odd rows of the first
// <-> the even rows of the second.)  Folding N values over the wave this way costs N + N / 2 instructions for the two cross-row levels
// where swap16_sum / swap32_sum cost 6 N (a copy, a swap, an addition each), and leaves a quarter of the registers for the row level.
__device__ __forceinline__ float pack32_sum(float a, float b) {
    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}
__device__ __forceinline__ float pack16_sum(float a, float b) {
    const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}
// which of four values x0..x3 row q holds after pack16_sum(pack32_sum(x0, x1), pack32_sum(x2, x3)): 0, 2, 1, 3
__device__ __forceinline__ int packed_slot(int q) { return ((q & 1) << 1) | (q >> 1); }
// The 14 moment sums of the M-step over the wave's 64 lanes: 7 + 4 swaps with their additions and 16 row rotations (39 instructions;
// class_sum<1> of each was 140).  The tree of every sum: (v[l] + v[l + 32]) over the halves, then over the row pairs (+ 16), then
// row_sum16 -- a value that lives in lanes 0..15 only (the others +0.0) comes out as row_sum16 of it.  r[j], row q, every lane of the
// row: moment 4 j + packed_slot(q) for j < 3; r[3]: moment 12 in rows 0 and 1, moment 13 in rows 2 and 3.
__device__ __forceinline__ void moment_sums(const float (&m)[14], float (&r)[4]) {
    const float a0 = pack32_sum(m[0], m[1]), a1 = pack32_sum(m[2], m[3]), a2 = pack32_sum(m[4], m[5]), a3 = pack32_sum(m[6], m[7]);
    const float a4 = pack32_sum(m[8], m[9]), a5 = pack32_sum(m[10], m[11]), a6 = pack32_sum(m[12], m[13]);
    r[0] = row_sum16(pack16_sum(a0, a1));
    r[1] = row_sum16(pack16_sum(a2, a3));
    r[2] = row_sum16(pack16_sum(a4, a5));
    r[3] = row_sum16(swap16_sum(a6));
}
__device__ __forceinline__ int moment_of(int j, int q) { return j < 3 ? 4 * j + packed_slot(q) : 12 + (q >> 1); }

#define GSR_DET_TOL 0.04          // |det_float32 / det - 1| allowed for a regular component (enters the filter bound as is)
__device__ __forceinline__ bool spd_det64(double a00, double a01, double a02, double a11, double a12, double a22, double& det) {
    const double m2 = a00 * a11 - a01 * a01;
    const double t1 = a00 * (a11 * a22 - a12 * a12), t2 = a01 * (a01 * a22 - a12 * a02), t3 = a02 * (a01 * a12 - a11 * a02);
    det = t1 - t2 + t3;
    const double mag = fabs(a00) * (fabs(a11 * a22) + a12 * a12) + fabs(a01) * (fabs(a01 * a22) + fabs(a12 * a02)) +
                       fabs(a02) * (fabs(a01 * a12) + fabs(a11 * a02));
    return a00 > 0.0 && a11 > 0.0 && a22 > 0.0 && m2 > 1e-12 * a00 * a11 && det > 1e-9 * mag && mag < 1e300;
}

}  // namespace gsr
