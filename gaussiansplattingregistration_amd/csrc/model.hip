// model.hip -- level export glue on the device: scaling / rotation of every component of a mixture level
// from its covariance, behind gsr_decompose_cov (include/gsr_hip.h).
//
// Replaces GaussianModel.decompose_covariance_matrix + matrices_to_quaternions of the reference
// (src/models/gaussian_model.py:151-153,242-265, src/utils/general_utils.py:94-100), which run a batched
// torch.linalg.eigh, three gathers / scatters and a stack of element-wise kernels on "cuda:0" after every HEM level.
// One thread per component: float64 cyclic Jacobi on the symmetric 3x3 in registers (the covariances are float32; the
// decomposition is exact to float32 output precision), then
//
//   GSR_DECOMP_REFERENCE   the reference's arithmetic, bug for bug: eigenvalues ascending; eigenpair k claims the axis
//       its eigenvector is most aligned with (arg-max of |v_k|, first maximum on ties); the eigenVALUE k goes to that
//       slot of `scaling` and ROW k of the eigenvector matrix V (columns = eigenvectors; the reference scatters rows of
//       the tensor eigh returns, gaussian_model.py:262) goes to that row of the "rotation" matrix; two claims of one
//       slot overwrite in eigenvalue order (torch's CPU scatter_: the last one wins), an unclaimed slot stays zero;
//       quaternion (w, x, y, z) by the trace formula w = sqrt(1 + tr) / 2 without any branch (NaN when 1 + tr < 0, as
//       there).  The "scaling" is the eigenvalue itself -- not its square root, not a logarithm.
//   GSR_DECOMP_EXACT       what save_ply of a down-sampled model actually needs (the reference's own comment calls its
//       version unused): scaling = log of the standard deviations (0.5 log lambda_k, lambda clamped at 1e-30), rotation =
//       the proper rotation [v_0 v_1 v_2] (third column flipped if the determinant is negative) as a unit quaternion by
//       Shepperd's branch on the largest diagonal term, so that R diag(exp(scaling))^2 R^T reproduces the covariance.
//
// Sign convention of the eigenvectors (torch.linalg.eigh leaves it to the LAPACK / rocSOLVER build): the component of
// largest magnitude of every eigenvector is positive (first maximum on ties).  Everything the reference path derives from
// V is invariant under column sign flips except the signs inside the scattered rows; tests compare up to those.
#include "gsr_common.h"

#include <float.h>
#include <math.h>
#include <string.h>

#include <vector>

namespace gsr {

// cyclic Jacobi for a symmetric 3x3: A = V diag(lam) V^T, eigenvalues ascending, columns of V = eigenvectors
__device__ void sym_eig3_d(const double Ain[3][3], double V[3][3], double lam[3]) {
    double A[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { A[i][j] = Ain[i][j]; V[i][j] = i == j ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 30; ++sweep) {
        const double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
        const double diag = A[0][0] * A[0][0] + A[1][1] * A[1][1] + A[2][2] * A[2][2];
        if (!(off > 1e-34 * diag) || !(off == off)) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (A[p][q] == 0.0) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) {           // A <- A J
                    const double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - s * akq; A[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {           // A <- J^T A
                    const double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - s * aqk; A[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k) {           // V <- V J
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq; V[k][q] = s * vkp + c * vkq;
                }
            }
    }
    lam[0] = A[0][0]; lam[1] = A[1][1]; lam[2] = A[2][2];
    // ascending order (three compare-exchanges on the columns)
    for (int pass = 0; pass < 3; ++pass) {
        const int a = pass == 1 ? 1 : 0, b = a + 1;             // (0,1) (1,2) (0,1)
        if (lam[b] < lam[a]) {
            const double t = lam[a]; lam[a] = lam[b]; lam[b] = t;
            for (int k = 0; k < 3; ++k) { const double v = V[k][a]; V[k][a] = V[k][b]; V[k][b] = v; }
        }
    }
    // sign convention: the component of largest magnitude of every eigenvector is positive
    for (int c = 0; c < 3; ++c) {
        int m = 0;
        if (fabs(V[1][c]) > fabs(V[m][c])) m = 1;
        if (fabs(V[2][c]) > fabs(V[m][c])) m = 2;
        if (V[m][c] < 0.0) for (int k = 0; k < 3; ++k) V[k][c] = -V[k][c];
    }
}

__global__ __launch_bounds__(256) void k_decompose_cov(int64_t n, const float* __restrict__ cov6, int mode, float* __restrict__ scaling,
                                                       float* __restrict__ quat, float* __restrict__ mat) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double c00 = cov6[6 * i], c01 = cov6[6 * i + 1], c02 = cov6[6 * i + 2], c11 = cov6[6 * i + 3], c12 = cov6[6 * i + 4], c22 = cov6[6 * i + 5];
        const double A[3][3] = {{c00, c01, c02}, {c01, c11, c12}, {c02, c12, c22}};
        double V[3][3], lam[3];
        sym_eig3_d(A, V, lam);
        float M[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
        float sc[3] = {0, 0, 0};
        float q[4];
        if (mode == GSR_DECOMP_REFERENCE) {
            for (int k = 0; k < 3; ++k) {                       // ascending eigenvalue: the later claim of a slot wins
                int slot = 0;                                   // arg-max of |v_k| = row k of |V^T| (gaussian_model.py:251-254)
                if (fabs(V[1][k]) > fabs(V[slot][k])) slot = 1;
                if (fabs(V[2][k]) > fabs(V[slot][k])) slot = 2;
                sc[slot] = (float)lam[k];
                for (int c = 0; c < 3; ++c) M[slot][c] = (float)V[k][c];      // ROW k of the eigenvector matrix, as the reference scatters it
            }
            const float w = sqrtf(1.0f + (M[0][0] + M[1][1] + M[2][2])) / 2.0f;      // general_utils.py:94-100
            q[0] = w;
            q[1] = (M[2][1] - M[1][2]) / (4.0f * w);
            q[2] = (M[0][2] - M[2][0]) / (4.0f * w);
            q[3] = (M[1][0] - M[0][1]) / (4.0f * w);
        } else {
            double R[3][3];
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) R[r][c] = V[r][c];
            const double det = R[0][0] * (R[1][1] * R[2][2] - R[1][2] * R[2][1]) - R[0][1] * (R[1][0] * R[2][2] - R[1][2] * R[2][0]) +
                               R[0][2] * (R[1][0] * R[2][1] - R[1][1] * R[2][0]);
            if (det < 0.0) for (int r = 0; r < 3; ++r) R[r][2] = -R[r][2];
            for (int k = 0; k < 3; ++k) sc[k] = (float)(0.5 * log(lam[k] > 1e-30 ? lam[k] : 1e-30));
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) M[r][c] = (float)R[r][c];
            // Shepperd: pick the largest of w^2, x^2, y^2, z^2
            const double tr = R[0][0] + R[1][1] + R[2][2];
            double w, x, y, z;
            if (tr > 0.0) {
                const double s = sqrt(tr + 1.0) * 2.0;
                w = 0.25 * s; x = (R[2][1] - R[1][2]) / s; y = (R[0][2] - R[2][0]) / s; z = (R[1][0] - R[0][1]) / s;
            } else if (R[0][0] > R[1][1] && R[0][0] > R[2][2]) {
                const double s = sqrt(1.0 + R[0][0] - R[1][1] - R[2][2]) * 2.0;
                w = (R[2][1] - R[1][2]) / s; x = 0.25 * s; y = (R[0][1] + R[1][0]) / s; z = (R[0][2] + R[2][0]) / s;
            } else if (R[1][1] > R[2][2]) {
                const double s = sqrt(1.0 + R[1][1] - R[0][0] - R[2][2]) * 2.0;
                w = (R[0][2] - R[2][0]) / s; x = (R[0][1] + R[1][0]) / s; y = 0.25 * s; z = (R[1][2] + R[2][1]) / s;
            } else {
                const double s = sqrt(1.0 + R[2][2] - R[0][0] - R[1][1]) * 2.0;
                w = (R[1][0] - R[0][1]) / s; x = (R[0][2] + R[2][0]) / s; y = (R[1][2] + R[2][1]) / s; z = 0.25 * s;
            }
            const double nq = sqrt(w * w + x * x + y * y + z * z);
            if (w < 0.0) { w = -w; x = -x; y = -y; z = -z; }
            q[0] = (float)(w / nq); q[1] = (float)(x / nq); q[2] = (float)(y / nq); q[3] = (float)(z / nq);
        }
        scaling[3 * i] = sc[0]; scaling[3 * i + 1] = sc[1]; scaling[3 * i + 2] = sc[2];
        quat[4 * i] = q[0]; quat[4 * i + 1] = q[1]; quat[4 * i + 2] = q[2]; quat[4 * i + 3] = q[3];
        if (mat)
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) mat[9 * i + 3 * r + c] = M[r][c];
    }
}

// ---- RANSAC plane fitting: the data-parallel inner part of _fit_single_plane of the reference
// (src/utils/plane_fitting_util.py:38-69).  The reference evaluates its `iterations` candidate planes one after the
// other, each with five full-size torch kernels on the CPU; here ALL candidates of a plane search are scored in one
// pass over the points: a thread per point, the candidates broadcast from LDS, inliers counted per candidate by
// ballot + popcount (one LDS atomic per wave and candidate).
//   distance  = (x n'_0 + y n'_1 + z n'_2 + d) / |n'|      n' = the plane normal re-normalised in float32 (:91-96)
//   inlier   <=> |distance| < distance_threshold  and  |<point normal, plane normal>| > normal_threshold   (:54-61)
// float32 throughout.  The reference's dot products come out of torch.mm / torch.matmul on the CPU, i.e. MKL's sgemv, whose
// K = 3 rows round as  fl(z n2) + fma(y, n1, fl(x n0))  -- measured in the build container: that expression reproduces
// torch.mm bit for bit on 10^6 random rows, EXCEPT the few rows in the tail of each OpenMP thread's share (266 of 2.3 M rows
// with 8 threads; which rows depends on the thread count of the machine), which take another path.  So the host result
// cannot be reproduced exactly in general; the kernel uses the dominant rounding (tests/test_planes_gpu.py).
struct PlaneCand { float n0, n1, n2, d, m0, m1, m2, nn; };      // n' (re-normalised), d, plane normal as sampled, |n'|

__device__ __forceinline__ bool plane_inlier(const PlaneCand& c, float x, float y, float z, float nx, float ny, float nz, float dist_thr, float nrm_thr) {
    const float dot = z * c.n2 + __builtin_fmaf(y, c.n1, x * c.n0);
    const float dist = (dot + c.d) / c.nn;
    const float al = nz * c.m2 + __builtin_fmaf(ny, c.m1, nx * c.m0);
    return fabsf(dist) < dist_thr && fabsf(al) > nrm_thr;
}

#define PLANE_CHUNK 256
__global__ __launch_bounds__(256) void k_plane_score(int64_t n, const float* __restrict__ xyz, const float* __restrict__ nrm, int P,
                                                     const PlaneCand* __restrict__ cand, float dist_thr, float nrm_thr, unsigned* __restrict__ counts) {
    __shared__ PlaneCand s_c[PLANE_CHUNK];
    __shared__ unsigned s_n[PLANE_CHUNK];
    const int lane = threadIdx.x & 63;
    for (int p0 = 0; p0 < P; p0 += PLANE_CHUNK) {
        const int pn = P - p0 < PLANE_CHUNK ? P - p0 : PLANE_CHUNK;
        __syncthreads();
        if ((int)threadIdx.x < pn) { s_c[threadIdx.x] = cand[p0 + threadIdx.x]; s_n[threadIdx.x] = 0u; }
        __syncthreads();
        for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x; i0 < n; i0 += (int64_t)gridDim.x * blockDim.x) {
            const int64_t i = i0 + threadIdx.x;
            const bool live = i < n;
            const int64_t ii = live ? i : 0;
            const float x = xyz[3 * ii], y = xyz[3 * ii + 1], z = xyz[3 * ii + 2];
            const float nx = nrm[3 * ii], ny = nrm[3 * ii + 1], nz = nrm[3 * ii + 2];
            for (int p = 0; p < pn; ++p) {
                const bool in = live && plane_inlier(s_c[p], x, y, z, nx, ny, nz, dist_thr, nrm_thr);
                const unsigned long long m = __ballot(in);
                if (m != 0ull && lane == 0) atomicAdd(&s_n[p], (unsigned)__popcll(m));
            }
        }
        __syncthreads();
        if ((int)threadIdx.x < pn && s_n[threadIdx.x]) atomicAdd(&counts[p0 + threadIdx.x], s_n[threadIdx.x]);
    }
}
__global__ __launch_bounds__(256) void k_plane_mask(int64_t n, const float* __restrict__ xyz, const float* __restrict__ nrm, PlaneCand c,
                                                    float dist_thr, float nrm_thr, uint8_t* __restrict__ mask) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        mask[i] = plane_inlier(c, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2], dist_thr, nrm_thr) ? 1 : 0;
}

}  // namespace gsr

using namespace gsr;

extern "C" int32_t gsr_plane_score(const float* xyz, const float* normals, int64_t n, const float* candidates, int32_t n_candidates,
                                   float distance_threshold, float normal_threshold, uint32_t* counts, uint8_t* best_mask, int32_t* best,
                                   int32_t on_device, int32_t device, void* stream) {
    if (n < 0 || n_candidates < 0 || (n > 0 && (!xyz || !normals)) || (n_candidates > 0 && (!candidates || !counts)) || !best)
        return fail(GSR_E_INVALID, "gsr_plane_score: bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(GSR_E_NO_DEVICE, "gsr_plane_score: no HIP device visible (this backend has no CPU fallback)");
    *best = -1;
    if (n == 0 || n_candidates == 0) return GSR_OK;
    GSR_HIP(hipSetDevice(device));
    hipStream_t st = (hipStream_t)stream;
    static_assert(sizeof(PlaneCand) == 32, "candidates are 8 floats");
    DevBuf dx, dn, dc, dcnt, dmask;
    const float* px = xyz;
    const float* pn = normals;
    int32_t r = dc.reserve((size_t)n_candidates * 32);
    if (r == GSR_OK) r = dcnt.reserve((size_t)n_candidates * 4);
    if (r == GSR_OK && !on_device) { r = dx.reserve((size_t)n * 12); if (r == GSR_OK) r = dn.reserve((size_t)n * 12); }
    if (r == GSR_OK && best_mask && !on_device) r = dmask.reserve((size_t)n);
    hipError_t e = hipSuccess;
    std::vector<uint32_t> hc((size_t)n_candidates);
    if (r == GSR_OK) {
        if (!on_device) {
            e = hipMemcpyAsync(dx.p, xyz, (size_t)n * 12, hipMemcpyHostToDevice, st);
            if (e == hipSuccess) e = hipMemcpyAsync(dn.p, normals, (size_t)n * 12, hipMemcpyHostToDevice, st);
            px = dx.as<float>(); pn = dn.as<float>();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(dc.p, candidates, (size_t)n_candidates * 32, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipMemsetAsync(dcnt.p, 0, (size_t)n_candidates * 4, st);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_plane_score, dim3(stride_grid(n)), dim3(256), 0, st, n, px, pn, (int)n_candidates, dc.as<PlaneCand>(), distance_threshold,
                               normal_threshold, dcnt.as<unsigned>());
            e = hipMemcpyAsync(hc.data(), dcnt.p, (size_t)n_candidates * 4, hipMemcpyDeviceToHost, st);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(st);
    }
    if (r == GSR_OK && e == hipSuccess) {
        // the reference keeps the FIRST candidate with the strictly largest count (plane_fitting_util.py:63-66)
        uint32_t mx = 0;
        for (int32_t p = 0; p < n_candidates; ++p) {
            counts[p] = hc[(size_t)p];
            if (hc[(size_t)p] > mx) { mx = hc[(size_t)p]; *best = p; }
        }
        if (*best >= 0 && best_mask) {
            PlaneCand c;
            memcpy(&c, candidates + 8 * (size_t)*best, sizeof(c));
            uint8_t* dm = on_device ? best_mask : dmask.as<uint8_t>();
            hipLaunchKernelGGL(k_plane_mask, dim3(stride_grid(n)), dim3(256), 0, st, n, px, pn, c, distance_threshold, normal_threshold, dm);
            if (!on_device) e = hipMemcpyAsync(best_mask, dm, (size_t)n, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
        }
    }
    dx.release(); dn.release(); dc.release(); dcnt.release(); dmask.release();
    if (r != GSR_OK) return r;
    if (e != hipSuccess) return fail(GSR_E_HIP, "gsr_plane_score: %s", hipGetErrorString(e));
    return GSR_OK;
}

// ---- 3DGS .ply rows -> device SoA ------------------------------------------------------------------------
// The reference loads a .ply straight onto cuda:0 (src/models/gaussian_model.py:98-139: plyfile -> numpy -> torch.tensor(device=
// "cuda"), the f_rest block transposed from the file's channel-major (P, 3, K) to coefficient-major (P, K, 3), the covariance
// built there from scale and rotation, :34-38,139).  Here the file's vertex rows (little-endian float32 properties, any order:
// `off` holds each wanted property's byte offset inside a row) arrive in HBM as they are on disk -- chunked through pinned
// memory by the host -- and ONE kernel scatters a chunk into the level-0 arrays the HEM boundary takes.
struct PlyLayout {
    int row_bytes;
    int K;                      // SH-rest coefficients per channel (15 at degree 3)
    int xyz[3], dc[3], opacity, scale[3], rot[4];
    int rest0;                  // byte offset of f_rest_0 ... f_rest_(3K-1), consecutive float32
};

__device__ __forceinline__ float ply_f32(const unsigned char* row, int off) {
    float v;
    if ((reinterpret_cast<uintptr_t>(row + off) & 3u) == 0u) v = *reinterpret_cast<const float*>(row + off);
    else memcpy(&v, row + off, 4);              // rows of files with uchar / double properties in front need not be 4-byte aligned
    return v;
}

__global__ __launch_bounds__(256) void k_ply_unpack(int64_t n, const unsigned char* __restrict__ rows, PlyLayout L, float* __restrict__ xyz,
                                                    float* __restrict__ color, float* __restrict__ sh, float* __restrict__ opacity,
                                                    float* __restrict__ scale, float* __restrict__ rot, float* __restrict__ cov6) {
    const int F = 3 * L.K;
    // geometry: a thread per splat
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const unsigned char* r = rows + i * L.row_bytes;
        xyz[3 * i] = ply_f32(r, L.xyz[0]); xyz[3 * i + 1] = ply_f32(r, L.xyz[1]); xyz[3 * i + 2] = ply_f32(r, L.xyz[2]);
        color[3 * i] = ply_f32(r, L.dc[0]); color[3 * i + 1] = ply_f32(r, L.dc[1]); color[3 * i + 2] = ply_f32(r, L.dc[2]);
        opacity[i] = ply_f32(r, L.opacity);
        const float s0 = ply_f32(r, L.scale[0]), s1 = ply_f32(r, L.scale[1]), s2 = ply_f32(r, L.scale[2]);
        const float qw = ply_f32(r, L.rot[0]), qx = ply_f32(r, L.rot[1]), qy = ply_f32(r, L.rot[2]), qz = ply_f32(r, L.rot[3]);
        scale[3 * i] = s0; scale[3 * i + 1] = s1; scale[3 * i + 2] = s2;
        rot[4 * i] = qw; rot[4 * i + 1] = qx; rot[4 * i + 2] = qy; rot[4 * i + 3] = qz;
        // build_rotation (general_utils.py:43-66) on the normalised quaternion, L = R diag(exp(scale)), covariance = L L^T (:69-80), float32
        const float nrm = sqrtf(qw * qw + qx * qx + qy * qy + qz * qz);
        const float w = qw / nrm, x = qx / nrm, y = qy / nrm, z = qz / nrm;
        const float e0 = expf(s0), e1 = expf(s1), e2 = expf(s2);
        const float R00 = 1.0f - 2.0f * (y * y + z * z), R01 = 2.0f * (x * y - w * z), R02 = 2.0f * (x * z + w * y);
        const float R10 = 2.0f * (x * y + w * z), R11 = 1.0f - 2.0f * (x * x + z * z), R12 = 2.0f * (y * z - w * x);
        const float R20 = 2.0f * (x * z - w * y), R21 = 2.0f * (y * z + w * x), R22 = 1.0f - 2.0f * (x * x + y * y);
        const float l00 = R00 * e0, l01 = R01 * e1, l02 = R02 * e2, l10 = R10 * e0, l11 = R11 * e1, l12 = R12 * e2, l20 = R20 * e0, l21 = R21 * e1, l22 = R22 * e2;
        cov6[6 * i] = (l00 * l00 + l01 * l01) + l02 * l02; cov6[6 * i + 1] = (l00 * l10 + l01 * l11) + l02 * l12;
        cov6[6 * i + 2] = (l00 * l20 + l01 * l21) + l02 * l22; cov6[6 * i + 3] = (l10 * l10 + l11 * l11) + l12 * l12;
        cov6[6 * i + 4] = (l10 * l20 + l11 * l21) + l12 * l22; cov6[6 * i + 5] = (l20 * l20 + l21 * l21) + l22 * l22;
    }
    // SH rest: a thread per output float, consecutive threads on consecutive output addresses; sh[i][k][c] = file f_rest[c * K + k]
    if (F > 0) {
        const int64_t total = n * F;
        for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
            const int64_t i = t / F;
            const int f = (int)(t - i * F);
            const int k = f / 3, c = f - 3 * k;
            sh[t] = ply_f32(rows + i * L.row_bytes, L.rest0 + 4 * (c * L.K + k));
        }
    }
}

extern "C" int32_t gsr_ply_unpack(const void* rows_dev, int64_t n, int32_t row_bytes, const int32_t* offsets, int32_t K, float* xyz, float* color,
                                  float* sh, float* opacity, float* scale, float* rot, float* cov6, int32_t device, void* stream) {
    if (n < 0 || K < 0 || row_bytes <= 0 || !offsets || (n > 0 && (!rows_dev || !xyz || !color || !opacity || !scale || !rot || !cov6 || (K > 0 && !sh))))
        return fail(GSR_E_INVALID, "gsr_ply_unpack: bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(GSR_E_NO_DEVICE, "gsr_ply_unpack: no HIP device visible (this backend has no CPU fallback)");
    PlyLayout L;
    L.row_bytes = row_bytes; L.K = K;
    for (int i = 0; i < 3; ++i) { L.xyz[i] = offsets[i]; L.dc[i] = offsets[3 + i]; L.scale[i] = offsets[7 + i]; }
    L.opacity = offsets[6];
    for (int i = 0; i < 4; ++i) L.rot[i] = offsets[10 + i];
    L.rest0 = offsets[14];
    for (int i = 0; i < 15; ++i)
        if (offsets[i] < 0 || offsets[i] + (i == 14 ? 12 * K : 4) > row_bytes) {
            if (i == 14 && K == 0) continue;
            return fail(GSR_E_INVALID, "gsr_ply_unpack: property offset %d outside the %d-byte row", offsets[i], row_bytes);
        }
    if (n == 0) return GSR_OK;
    GSR_HIP(hipSetDevice(device));
    hipLaunchKernelGGL(k_ply_unpack, dim3(stride_grid(n * (K > 0 ? 3 * K : 1))), dim3(256), 0, (hipStream_t)stream, n, (const unsigned char*)rows_dev, L, xyz, color, sh,
                       opacity, scale, rot, cov6);
    GSR_HIP(hipGetLastError());
    return GSR_OK;
}

extern "C" int32_t gsr_decompose_cov(const float* cov6, int64_t n, int32_t mode, float* scaling, float* rotation, float* matrix,
                                     int32_t on_device, int32_t device, void* stream) {
    if (n < 0 || (n > 0 && (!cov6 || !scaling || !rotation))) return fail(GSR_E_INVALID, "gsr_decompose_cov: bad argument");
    if (mode != GSR_DECOMP_REFERENCE && mode != GSR_DECOMP_EXACT) return fail(GSR_E_INVALID, "gsr_decompose_cov: unknown mode %d", mode);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(GSR_E_NO_DEVICE, "gsr_decompose_cov: no HIP device visible (this backend has no CPU fallback)");
    if (n == 0) return GSR_OK;
    GSR_HIP(hipSetDevice(device));
    hipStream_t st = (hipStream_t)stream;
    if (on_device) {
        hipLaunchKernelGGL(k_decompose_cov, dim3(stride_grid(n)), dim3(256), 0, st, n, cov6, (int)mode, scaling, rotation, matrix);
        GSR_HIP(hipGetLastError());
        GSR_HIP(hipStreamSynchronize(st));
        return GSR_OK;
    }
    DevBuf in, sc, q, m;
    int32_t r = in.reserve((size_t)n * 24);
    if (r == GSR_OK) r = sc.reserve((size_t)n * 12);
    if (r == GSR_OK) r = q.reserve((size_t)n * 16);
    if (r == GSR_OK && matrix) r = m.reserve((size_t)n * 36);
    hipError_t e = hipSuccess;
    if (r == GSR_OK) {
        e = hipMemcpyAsync(in.p, cov6, (size_t)n * 24, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_decompose_cov, dim3(stride_grid(n)), dim3(256), 0, st, n, in.as<float>(), (int)mode, sc.as<float>(), q.as<float>(),
                               matrix ? m.as<float>() : (float*)nullptr);
            e = hipMemcpyAsync(scaling, sc.p, (size_t)n * 12, hipMemcpyDeviceToHost, st);
        }
        if (e == hipSuccess) e = hipMemcpyAsync(rotation, q.p, (size_t)n * 16, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess && matrix) e = hipMemcpyAsync(matrix, m.p, (size_t)n * 36, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
    }
    in.release(); sc.release(); q.release(); m.release();
    if (r != GSR_OK) return r;
    if (e != hipSuccess) return fail(GSR_E_HIP, "gsr_decompose_cov: %s", hipGetErrorString(e));
    return GSR_OK;
}
