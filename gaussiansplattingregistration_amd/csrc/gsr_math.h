// gsr_math.h -- device-side float algebra for the HEM kernels (gfx950).
//
// The discrete decisions of a HEM level (radius test, colour gate, KL gate) are comparisons of
// float32 expressions against thresholds, so they only match the reference CPU extension if every
// expression keeps the reference's operand order and rounding:
//   * this translation unit is compiled with -ffp-contract=off (no FMA contraction) and hipcc's
//     default correctly-rounded fp32 divide / sqrt;
//   * f32 subnormals are kept (hipcc default for gfx9);
//   * the one transcendental inside a gate, log(det_c / det_p) of the KL divergence
//     (reference gaussian.hpp:106-109), goes through glibc_logf() below -- the table-driven
//     double-precision algorithm glibc's logf uses (ARM optimized-routines, MIT licensed,
//     glibc sysdeps/ieee754/flt-32/e_logf.c), whose result is identical to the host libm's on
//     every input tested (tests/test_oracle.py checks 4e8 samples of it against libm through
//     the CPU oracle).
// Function comments cite the reference (paths relative to src/cpp_ext/).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gsr {

struct f3 { float x, y, z; };
struct s6 { float e00, e01, e02, e11, e12, e22; };   // xx xy xz yy yz zz (include/vec.hpp:458)

__device__ __forceinline__ f3 sub3(const f3& a, const f3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
// include/vec.hpp:156-159: a.x*b.x + a.y*b.y + a.z*b.z, left to right
__device__ __forceinline__ float dot3(const f3& a, const f3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// include/vec.hpp:863-866
__device__ __forceinline__ float det6(const s6& c) {
    return -c.e02 * c.e02 * c.e11 + 2.0f * c.e01 * c.e02 * c.e12 - c.e00 * c.e12 * c.e12
           - c.e01 * c.e01 * c.e22 + c.e00 * c.e11 * c.e22;
}

// include/vec.hpp:868-872 with smat3::operator/ (:516-520): cofactors times (1/det)
__device__ __forceinline__ s6 inverse6(const s6& c, float detc) {
    s6 r = {c.e11 * c.e22 - c.e12 * c.e12, c.e02 * c.e12 - c.e01 * c.e22, c.e01 * c.e12 - c.e02 * c.e11,
            c.e00 * c.e22 - c.e02 * c.e02, c.e02 * c.e01 - c.e00 * c.e12, c.e00 * c.e11 - c.e01 * c.e01};
    float inv_s = 1.0f / detc;
    return {r.e00 * inv_s, r.e01 * inv_s, r.e02 * inv_s, r.e11 * inv_s, r.e12 * inv_s, r.e22 * inv_s};
}

// include/vec.hpp:540-543
__device__ __forceinline__ f3 mul6(const s6& m, const f3& v) {
    return {m.e00 * v.x + m.e01 * v.y + m.e02 * v.z, m.e01 * v.x + m.e11 * v.y + m.e12 * v.z,
            m.e02 * v.x + m.e12 * v.y + m.e22 * v.z};
}

// trace(a * c): diagonal of smat3::operator*(const smat3&) (include/vec.hpp:544-551), a = *this
__device__ __forceinline__ float trace_prod6(const s6& a, const s6& c) {
    float m00 = c.e00 * a.e00 + c.e01 * a.e01 + c.e02 * a.e02;
    float m11 = c.e01 * a.e01 + c.e11 * a.e11 + c.e12 * a.e12;
    float m22 = c.e02 * a.e02 + c.e12 * a.e12 + c.e22 * a.e22;
    return m00 + m11 + m22;
}

// glibc logf (see file header).  Table = __logf_data of glibc 2.35 / ARM optimized-routines logf_data.c.
__device__ __constant__ static const double k_logf_tab[32] = {
    0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2, 0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2,
    0x1.49539f0f010b0p+0, -0x1.01eae7f513a67p-2, 0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3,
    0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3, 0x1.25e227b0b8ea0p+0, -0x1.1aa2bc79c8100p-3,
    0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4, 0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4,
    0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5, 0x1p+0,                0x0p+0,
    0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5,  0x1.ca4b31f026aa0p-1, 0x1.c5e53aa362eb4p-4,
    0x1.b2036576afce6p-1, 0x1.526e57720db08p-3,  0x1.9c2d163a1aa2dp-1, 0x1.bc2860d224770p-3,
    0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2,  0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2};

// `tab` = the 32 doubles of k_logf_tab (k_select keeps a copy in LDS: a per-lane lookup in __constant__ memory is a
// vector-memory gather with its full round-trip latency in the middle of the KL gate)
__device__ __forceinline__ float glibc_logf_tab(float x, const double* tab) {
    uint32_t ix = __float_as_uint(x);
    if (ix == 0x3f800000u) return 0.0f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
        if (ix * 2u == 0u) return -__builtin_inff();                                   // log(+-0) = -inf
        if (ix == 0x7f800000u) return x;                                               // log(inf) = inf
        if ((ix & 0x80000000u) || ix * 2u >= 0xff000000u) return __builtin_nanf("");   // x < 0 or NaN
        ix = __float_as_uint(x * 0x1p23f);                                             // subnormal: normalise
        ix -= 23u << 23;
    }
    uint32_t tmp = ix - 0x3f330000u;
    int i = (tmp >> 19) & 15;
    int k = (int32_t)tmp >> 23;
    uint32_t iz = ix - (tmp & 0xff800000u);
    double invc = tab[2 * i], logc = tab[2 * i + 1];
    double z = (double)__uint_as_float(iz);
    double r = z * invc - 1.0;
    double y0 = logc + (double)k * 0x1.62e42fefa39efp-1;
    double r2 = r * r;
    double y = 0x1.5575b0be00b6ap-2 * r + -0x1.ffffef20a4123p-2;
    y = -0x1.00ea348b88334p-2 * r2 + y;
    y = y * r2 + (y0 + r);
    return (float)y;
}
__device__ __forceinline__ float glibc_logf(float x) { return glibc_logf_tab(x, k_logf_tab); }

// Kullback-Leibler gate value, include/gaussian.hpp:106-109 (child c against parent p).
//   d = mu_c - mu_p;  pinv = inverse(cov_p)
__device__ __forceinline__ float kld6(const f3& d, const s6& cov_c, float det_c, const s6& pinv, float det_p) {
    float smd = dot3(d, mul6(pinv, d));                                                // gaussian.hpp:82-85
    return 0.5f * (smd + trace_prod6(pinv, cov_c) - 3.0f - glibc_logf(det_c / det_p));
}

// Largest eigenvalue by the trigonometric closed form, include/vec.hpp:736-768: coefficients in
// float32, trig in float64, result narrowed to float32 (.z of the ascending triple).
__device__ __forceinline__ float eig_max6(const s6& m) {
    const double inv3 = 0.33333333333333333333333333333333;
    const double root3 = 1.7320508075688772935274463415059;
    double c0 = m.e00 * m.e11 * m.e22 + 2.0f * m.e01 * m.e02 * m.e12 - m.e00 * m.e12 * m.e12
                - m.e11 * m.e02 * m.e02 - m.e22 * m.e01 * m.e01;
    double c1 = m.e00 * m.e11 - m.e01 * m.e01 + m.e00 * m.e22 - m.e02 * m.e02 + m.e11 * m.e22 - m.e12 * m.e12;
    double c2 = m.e00 + m.e11 + m.e22;
    double c2Div3 = c2 * inv3;
    double aDiv3 = c1 * inv3 - c2Div3 * c2Div3;
    if (aDiv3 > 0.0) aDiv3 = 0.0;
    double mbDiv2 = 0.5 * c0 + c2Div3 * c2Div3 * c2Div3 - 0.5 * c2Div3 * c1;
    double q = mbDiv2 * mbDiv2 + aDiv3 * aDiv3 * aDiv3;
    if (q > 0.0) q = 0.0;
    double magnitude = sqrt(-aDiv3);
    double angle = atan2(sqrt(-q), mbDiv2) * inv3;
    if (angle != angle) angle = 0.0;
    double sn = sin(angle), cs = cos(angle);
    double e0 = c2Div3 + 2 * magnitude * cs;
    double e1 = c2Div3 - magnitude * (cs + root3 * sn);
    double e2 = c2Div3 - magnitude * (cs - root3 * sn);
    double h;
    if (e2 < e1) { h = e1; e1 = e2; e2 = h; }
    if (e1 < e0) { h = e0; e0 = e1; e1 = h; }
    if (e2 < e1) { h = e1; e1 = e2; e2 = h; }
    return (float)e2;
}

// hem::clamp with hem::fminf / hem::fmaxf (include/base.hpp:24-27): a NaN argument clamps to `hi`.
__device__ __forceinline__ float ref_clamp(float f, float lo, float hi) {
    float m = f < hi ? f : hi;
    return lo > m ? lo : m;
}

}  // namespace gsr
