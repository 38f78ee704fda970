// gsr_normals.h -- normal of a splat from its covariance (Open3D estimate_normals() on a cloud whose covariances are set:
// src/utils/point_cloud_converter.py:40-43 of the reference).  Shared by csrc/icp.hip (gsr_normals_from_cov: a cloud handed to the ICP
// without normals) and csrc/hem.hip (gsr_hem_run_levels: the normals of a level leave with the level -- SURVEY.md section 7 step 6),
// so that both write the same bits.
#pragma once
#include <hip/hip_runtime.h>

namespace gsr {

// ---- normals from splat covariances (Open3D FastEigen3x3, Eberly's robust 3x3 eigensolver) -------
static __device__ __forceinline__ void cross3d(const double a[3], const double b[3], double o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
static __device__ void eigvec0_d(const double A[3][3], double ev, double out[3]) {
    double r0[3] = {A[0][0] - ev, A[0][1], A[0][2]};
    double r1[3] = {A[0][1], A[1][1] - ev, A[1][2]};
    double r2[3] = {A[0][2], A[1][2], A[2][2] - ev};
    double c01[3], c02[3], c12[3];
    cross3d(r0, r1, c01); cross3d(r0, r2, c02); cross3d(r1, r2, c12);
    double d0 = c01[0] * c01[0] + c01[1] * c01[1] + c01[2] * c01[2];
    double d1 = c02[0] * c02[0] + c02[1] * c02[1] + c02[2] * c02[2];
    double d2 = c12[0] * c12[0] + c12[1] * c12[1] + c12[2] * c12[2];
    double dmax = d0;
    int imax = 0;
    if (d1 > dmax) { dmax = d1; imax = 1; }
    if (d2 > dmax) { dmax = d2; imax = 2; }
    const double inv = 1.0 / sqrt(dmax);
    for (int k = 0; k < 3; ++k) out[k] = (imax == 0 ? c01[k] : (imax == 1 ? c02[k] : c12[k])) * inv;
}
static __device__ void eigvec1_d(const double A[3][3], const double e0[3], double ev1, double out[3]) {
    double U[3], V[3];
    if (fabs(e0[0]) > fabs(e0[1])) {
        double inv = 1.0 / sqrt(e0[0] * e0[0] + e0[2] * e0[2]);
        U[0] = -e0[2] * inv; U[1] = 0; U[2] = e0[0] * inv;
    } else {
        double inv = 1.0 / sqrt(e0[1] * e0[1] + e0[2] * e0[2]);
        U[0] = 0; U[1] = e0[2] * inv; U[2] = -e0[1] * inv;
    }
    cross3d(e0, U, V);
    double AU[3], AV[3];
    for (int i = 0; i < 3; ++i) {
        AU[i] = A[i][0] * U[0] + A[i][1] * U[1] + A[i][2] * U[2];
        AV[i] = A[i][0] * V[0] + A[i][1] * V[1] + A[i][2] * V[2];
    }
    double m00 = U[0] * AU[0] + U[1] * AU[1] + U[2] * AU[2] - ev1;
    double m01 = U[0] * AV[0] + U[1] * AV[1] + U[2] * AV[2];
    double m11 = V[0] * AV[0] + V[1] * AV[1] + V[2] * AV[2] - ev1;
    const double a00 = fabs(m00), a01 = fabs(m01), a11 = fabs(m11);
    if (a00 >= a11) {
        if (fmax(a00, a01) > 0) {
            if (a00 >= a01) { m01 /= m00; m00 = 1 / sqrt(1 + m01 * m01); m01 *= m00; }
            else { m00 /= m01; m01 = 1 / sqrt(1 + m00 * m00); m00 *= m01; }
            for (int i = 0; i < 3; ++i) out[i] = m01 * U[i] - m00 * V[i];
        } else for (int i = 0; i < 3; ++i) out[i] = U[i];
    } else {
        if (fmax(a11, a01) > 0) {
            if (a11 >= a01) { m01 /= m11; m11 = 1 / sqrt(1 + m01 * m01); m01 *= m11; }
            else { m11 /= m01; m01 = 1 / sqrt(1 + m11 * m11); m11 *= m01; }
            for (int i = 0; i < 3; ++i) out[i] = m11 * U[i] - m01 * V[i];
        } else for (int i = 0; i < 3; ++i) out[i] = U[i];
    }
}
// normal of a covariance: eigenvector of its smallest eigenvalue (Open3D ComputeNormal with FastEigen3x3), (0, 0, 1) when
// that comes out as the zero vector or NaN
static __device__ void normal_of_cov_d(double c00, double c01, double c02, double c11, double c12, double c22, double v[3]) {
    v[0] = v[1] = v[2] = 0;
    double mx = fmax(fmax(fmax(c00, c01), fmax(c02, c11)), fmax(c12, c22));
    if (mx != 0 && mx == mx) {
        double A[3][3] = {{c00 / mx, c01 / mx, c02 / mx}, {c01 / mx, c11 / mx, c12 / mx}, {c02 / mx, c12 / mx, c22 / mx}};
        const double norm = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
        if (norm > 0) {
            const double q = (A[0][0] + A[1][1] + A[2][2]) / 3;
            const double b00 = A[0][0] - q, b11 = A[1][1] - q, b22 = A[2][2] - q;
            const double p = sqrt((b00 * b00 + b11 * b11 + b22 * b22 + norm * 2) / 6);
            const double k00 = b11 * b22 - A[1][2] * A[1][2];
            const double k01 = A[0][1] * b22 - A[1][2] * A[0][2];
            const double k02 = A[0][1] * A[1][2] - b11 * A[0][2];
            const double det = (b00 * k00 - A[0][1] * k01 + A[0][2] * k02) / (p * p * p);
            const double half_det = fmin(fmax(det * 0.5, -1.0), 1.0);
            const double angle = acos(half_det) / 3.0;
            const double two_thirds_pi = 2.09439510239319549;
            const double beta2 = cos(angle) * 2, beta0 = cos(angle + two_thirds_pi) * 2, beta1 = -(beta0 + beta2);
            const double ev[3] = {q + p * beta0, q + p * beta1, q + p * beta2};
            double e0[3], e1[3], e2[3];
            if (half_det >= 0) {
                eigvec0_d(A, ev[2], e2);
                if (ev[2] < ev[0] && ev[2] < ev[1]) { v[0] = e2[0]; v[1] = e2[1]; v[2] = e2[2]; }
                else {
                    eigvec1_d(A, e2, ev[1], e1);
                    if (ev[1] < ev[0] && ev[1] < ev[2]) { v[0] = e1[0]; v[1] = e1[1]; v[2] = e1[2]; }
                    else cross3d(e1, e2, v);
                }
            } else {
                eigvec0_d(A, ev[0], e0);
                if (ev[0] < ev[1] && ev[0] < ev[2]) { v[0] = e0[0]; v[1] = e0[1]; v[2] = e0[2]; }
                else {
                    eigvec1_d(A, e0, ev[1], e1);
                    if (ev[1] < ev[0] && ev[1] < ev[2]) { v[0] = e1[0]; v[1] = e1[1]; v[2] = e1[2]; }
                    else cross3d(e0, e1, v);
                }
            }
        } else {
            if (A[0][0] < A[1][1] && A[0][0] < A[2][2]) v[0] = 1;
            else if (A[1][1] < A[0][0] && A[1][1] < A[2][2]) v[1] = 1;
            else v[2] = 1;
        }
    }
    const double nn = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    if (nn == 0.0 || nn != nn) { v[0] = 0; v[1] = 0; v[2] = 1; }
}
}  // namespace gsr
