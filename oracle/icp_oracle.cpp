// oracle/icp_oracle.cpp -- CPU restatement of the ICP leaf the reference reaches through Open3D.
//
// TEST INFRASTRUCTURE ONLY (see oracle_api.h).
//
// Parity: UNPINNED.  The arithmetic of this half lives in a third-party dependency that is absent
// from /root/reference and from this machine: open3d == 0.16.0 (requirements.txt:3), called at
// src/utils/local_registration_util.py:88-90 (registration_icp) with estimators chosen at :39-51
// and robust losses at :58-73; clouds are built at src/utils/point_cloud_converter.py:31-49.
// The reference holds no test or golden vector for this boundary.  What follows restates Open3D
// 0.16.0's published algorithm (cpp/open3d/pipelines/registration/{Registration,
// TransformationEstimation,RobustKernel}.cpp, cpp/open3d/geometry/EstimateNormals.cpp,
// cpp/open3d/utility/Eigen.cpp, Eigen::umeyama) from its documented behaviour; it is cross-checked
// in tests/test_icp_oracle.py against an independent SciPy cKDTree + NumPy SVD/solve restatement
// and against known ground-truth rigid motions.
//
//   RegistrationICP loop     : T = init; evaluate; for it < max_iter { update = estimate(corr);
//                              T = update*T; source.Transform(update); evaluate;
//                              stop if |dfitness| < rel_fitness && |drmse| < rel_rmse }
//   evaluate                 : 1-NN per source point (KD-tree), kept iff d^2 < max_corr^2 (strict);
//                              fitness = |corr|/|source|, inlier_rmse = sqrt(sum d^2/|corr|)
//   point-to-point estimate  : Eigen::umeyama(src, dst, with_scaling=false)
//   point-to-plane estimate  : r = (p-q).n, w = kernel(r), J = [p x n, n]; JTJ += w J J^T, JTr += w r J;
//                              x = solve(JTJ, -JTr); update = Rz(x2) Ry(x1) Rx(x0), t = x3..5
//   all in float64; source points are transformed in place, incrementally, as Open3D does.
#include "oracle_api.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>
#include <map>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

struct P3 { double x, y, z; };

// ---- exact 1-NN KD-tree (median split, leaves of <= 16 points) -----------------------------------
struct KdTree {
    struct Node { int lo, hi, axis, left, right; double split; };
    std::vector<Node> nodes;
    std::vector<int64_t> idx;
    const P3* pts = nullptr;

    static double coord(const P3& p, int a) { return a == 0 ? p.x : (a == 1 ? p.y : p.z); }

    int build(int lo, int hi) {
        Node nd{lo, hi, -1, -1, -1, 0.0};
        int id = (int)nodes.size();
        nodes.push_back(nd);
        if (hi - lo <= 16) return id;
        double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
        for (int k = lo; k < hi; ++k) {
            const P3& p = pts[idx[k]];
            mn[0] = std::min(mn[0], p.x); mx[0] = std::max(mx[0], p.x);
            mn[1] = std::min(mn[1], p.y); mx[1] = std::max(mx[1], p.y);
            mn[2] = std::min(mn[2], p.z); mx[2] = std::max(mx[2], p.z);
        }
        int ax = 0;
        if (mx[1] - mn[1] > mx[ax] - mn[ax]) ax = 1;
        if (mx[2] - mn[2] > mx[ax] - mn[ax]) ax = 2;
        if (!(mx[ax] - mn[ax] > 0)) return id;   // all points coincide (or NaN): keep as leaf
        int mid = (lo + hi) / 2;
        std::nth_element(idx.begin() + lo, idx.begin() + mid, idx.begin() + hi,
                         [&](int64_t a, int64_t b) { return coord(pts[a], ax) < coord(pts[b], ax); });
        double split = coord(pts[idx[mid]], ax);
        int l = build(lo, mid);
        int r = build(mid, hi);
        nodes[id].axis = ax; nodes[id].split = split; nodes[id].left = l; nodes[id].right = r;
        return id;
    }

    void create(const P3* p, int64_t n) {
        pts = p;
        idx.resize(n);
        std::iota(idx.begin(), idx.end(), (int64_t)0);
        nodes.clear();
        nodes.reserve((size_t)(n / 4 + 16));
        if (n > 0) build(0, (int)n);
    }

    // nearest neighbour; ties resolved towards the lowest target index
    void nearest(const P3& q, int node, int64_t& best, double& bestd) const {
        const Node& nd = nodes[node];
        if (nd.axis < 0) {
            for (int k = nd.lo; k < nd.hi; ++k) {
                int64_t j = idx[k];
                double dx = q.x - pts[j].x, dy = q.y - pts[j].y, dz = q.z - pts[j].z;
                double d = dx * dx + dy * dy + dz * dz;
                if (d < bestd || (d == bestd && j < best)) { bestd = d; best = j; }
            }
            return;
        }
        double diff = coord(q, nd.axis) - nd.split;
        int first = diff < 0 ? nd.left : nd.right;
        int second = diff < 0 ? nd.right : nd.left;
        nearest(q, first, best, bestd);
        if (diff * diff <= bestd) nearest(q, second, best, bestd);
    }

    // k nearest neighbours: `heap` is a max-heap on (distance, index) holding the k best so far
    typedef std::pair<double, int64_t> DI;
    void knn(const P3& q, int node, size_t k, std::vector<DI>& heap) const {
        const Node& nd = nodes[node];
        if (nd.axis < 0) {
            for (int t = nd.lo; t < nd.hi; ++t) {
                int64_t j = idx[t];
                double dx = q.x - pts[j].x, dy = q.y - pts[j].y, dz = q.z - pts[j].z;
                DI c(dx * dx + dy * dy + dz * dz, j);
                if (heap.size() < k) { heap.push_back(c); std::push_heap(heap.begin(), heap.end()); }
                else if (c < heap.front()) { std::pop_heap(heap.begin(), heap.end()); heap.back() = c; std::push_heap(heap.begin(), heap.end()); }
            }
            return;
        }
        double diff = coord(q, nd.axis) - nd.split;
        int first = diff < 0 ? nd.left : nd.right;
        int second = diff < 0 ? nd.right : nd.left;
        knn(q, first, k, heap);
        if (heap.size() < k || diff * diff <= heap.front().first) knn(q, second, k, heap);
    }
};

// ---- small dense linear algebra in float64 --------------------------------------------------------
struct M3 { double a[3][3]; };
struct M4 { double a[4][4]; };

M4 identity4() { M4 m; std::memset(&m, 0, sizeof(m)); for (int i = 0; i < 4; ++i) m.a[i][i] = 1; return m; }
M4 mul4(const M4& A, const M4& B) {
    M4 C;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0;
            for (int k = 0; k < 4; ++k) s += A.a[i][k] * B.a[k][j];
            C.a[i][j] = s;
        }
    return C;
}
double det3(const M3& m) {
    return m.a[0][0] * (m.a[1][1] * m.a[2][2] - m.a[1][2] * m.a[2][1])
         - m.a[0][1] * (m.a[1][0] * m.a[2][2] - m.a[1][2] * m.a[2][0])
         + m.a[0][2] * (m.a[1][0] * m.a[2][1] - m.a[1][1] * m.a[2][0]);
}

// One-sided (Hestenes) Jacobi SVD of a 3x3: A = U diag(s) V^T, U and V orthogonal (full).
void svd3(const M3& A, M3& U, double s[3], M3& V) {
    double B[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { B[i][j] = A.a[i][j]; V.a[i][j] = i == j; }
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int i = 0; i < 3; ++i) { alpha += B[i][p] * B[i][p]; beta += B[i][q] * B[i][q]; gamma += B[i][p] * B[i][q]; }
                if (gamma == 0) continue;
                off = std::max(off, std::fabs(gamma) / std::sqrt(alpha * beta + 1e-300));
                double zeta = (beta - alpha) / (2 * gamma);
                double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1 + zeta * zeta));
                double c = 1 / std::sqrt(1 + t * t), sn = c * t;
                for (int i = 0; i < 3; ++i) {
                    double bp = B[i][p], bq = B[i][q];
                    B[i][p] = c * bp - sn * bq; B[i][q] = sn * bp + c * bq;
                    double vp = V.a[i][p], vq = V.a[i][q];
                    V.a[i][p] = c * vp - sn * vq; V.a[i][q] = sn * vp + c * vq;
                }
            }
        if (off < 1e-17) break;
    }
    // column norms are the singular values; sort descending
    int order[3] = {0, 1, 2};
    double nrm[3];
    for (int j = 0; j < 3; ++j) nrm[j] = std::sqrt(B[0][j] * B[0][j] + B[1][j] * B[1][j] + B[2][j] * B[2][j]);
    std::sort(order, order + 3, [&](int a, int b) { return nrm[a] > nrm[b]; });
    M3 Vs;
    double Bs[3][3];
    for (int j = 0; j < 3; ++j) {
        s[j] = nrm[order[j]];
        for (int i = 0; i < 3; ++i) { Vs.a[i][j] = V.a[i][order[j]]; Bs[i][j] = B[i][order[j]]; }
    }
    V = Vs;
    // U columns: normalised B columns; complete a rank-deficient basis by cross products
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i) U.a[i][j] = s[j] > 0 ? Bs[i][j] / s[j] : 0.0;
    auto colnorm = [&](int j) { return std::sqrt(U.a[0][j] * U.a[0][j] + U.a[1][j] * U.a[1][j] + U.a[2][j] * U.a[2][j]); };
    const double tiny = 1e-12 * (s[0] > 0 ? 1.0 : 0.0);
    if (!(s[0] > 0)) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) U.a[i][j] = i == j; return; }
    if (s[1] <= tiny * s[0] || colnorm(1) < 0.5) {
        // pick any unit vector orthogonal to u0
        double u0[3] = {U.a[0][0], U.a[1][0], U.a[2][0]};
        int k = std::fabs(u0[0]) < std::fabs(u0[1]) ? (std::fabs(u0[0]) < std::fabs(u0[2]) ? 0 : 2)
                                                     : (std::fabs(u0[1]) < std::fabs(u0[2]) ? 1 : 2);
        double e[3] = {0, 0, 0}; e[k] = 1;
        double d = e[0] * u0[0] + e[1] * u0[1] + e[2] * u0[2];
        double v[3] = {e[0] - d * u0[0], e[1] - d * u0[1], e[2] - d * u0[2]};
        double n = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        for (int i = 0; i < 3; ++i) U.a[i][1] = v[i] / n;
    }
    if (s[2] <= 1e-12 * s[0] || colnorm(2) < 0.5) {
        U.a[0][2] = U.a[1][0] * U.a[2][1] - U.a[2][0] * U.a[1][1];
        U.a[1][2] = U.a[2][0] * U.a[0][1] - U.a[0][0] * U.a[2][1];
        U.a[2][2] = U.a[0][0] * U.a[1][1] - U.a[1][0] * U.a[0][1];
    }
}

// Eigen::umeyama(src, dst, with_scaling = false) on the matched pairs
M4 umeyama(const std::vector<P3>& src, const std::vector<P3>& dst) {
    const size_t n = src.size();
    const double one_over_n = 1.0 / (double)n;
    double ms[3] = {0, 0, 0}, md[3] = {0, 0, 0};
    for (size_t i = 0; i < n; ++i) {
        ms[0] += src[i].x; ms[1] += src[i].y; ms[2] += src[i].z;
        md[0] += dst[i].x; md[1] += dst[i].y; md[2] += dst[i].z;
    }
    for (int k = 0; k < 3; ++k) { ms[k] *= one_over_n; md[k] *= one_over_n; }
    M3 sigma;
    std::memset(&sigma, 0, sizeof(sigma));
    for (size_t i = 0; i < n; ++i) {
        double a[3] = {dst[i].x - md[0], dst[i].y - md[1], dst[i].z - md[2]};
        double b[3] = {src[i].x - ms[0], src[i].y - ms[1], src[i].z - ms[2]};
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) sigma.a[r][c] += a[r] * b[c];
    }
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) sigma.a[r][c] *= one_over_n;
    M3 U, V;
    double s[3];
    svd3(sigma, U, s, V);
    double S[3] = {1, 1, 1};
    if (det3(U) * det3(V) < 0) S[2] = -1;
    M4 T = identity4();
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            double v = 0;
            for (int k = 0; k < 3; ++k) v += U.a[r][k] * S[k] * V.a[c][k];
            T.a[r][c] = v;
        }
    for (int r = 0; r < 3; ++r)
        T.a[r][3] = md[r] - (T.a[r][0] * ms[0] + T.a[r][1] * ms[1] + T.a[r][2] * ms[2]);
    return T;
}

// Robust kernel weights (Open3D RobustKernel.cpp)
double kernel_weight(int loss, double k, double r) {
    switch (loss) {
        case 1: { double e = std::fabs(r); double t = std::min(1.0, e / k); double u = 1.0 - t * t; return u * u; }   // Tukey
        case 2: { double t = r / k; return 1.0 / (1.0 + t * t); }                                                  // Cauchy
        case 3: { double t = k + r * r; return k / (t * t); }                                                      // GM
        case 4: { double e = std::fabs(r); return k / std::max(e, k); }                                            // Huber
        default: return 1.0;                                                                                       // L2
    }
}

// Solve the symmetric 6x6 system A x = b by LDL^T with diagonal pivoting (Eigen's A.ldlt().solve(b)).
void solve6(const double A_[6][6], const double b_[6], double x[6]) {
    double A[6][6], b[6];
    int perm[6];
    for (int i = 0; i < 6; ++i) { perm[i] = i; b[i] = b_[i]; for (int j = 0; j < 6; ++j) A[i][j] = A_[i][j]; }
    double L[6][6] = {{0}}, D[6];
    for (int k = 0; k < 6; ++k) {
        int piv = k;
        for (int i = k + 1; i < 6; ++i) if (std::fabs(A[i][i]) > std::fabs(A[piv][piv])) piv = i;
        if (piv != k) {
            for (int j = 0; j < 6; ++j) std::swap(A[k][j], A[piv][j]);
            for (int i = 0; i < 6; ++i) std::swap(A[i][k], A[i][piv]);
            for (int j = 0; j < k; ++j) std::swap(L[k][j], L[piv][j]);
            std::swap(perm[k], perm[piv]);
        }
        D[k] = A[k][k];
        L[k][k] = 1;
        for (int i = k + 1; i < 6; ++i) L[i][k] = A[i][k] / D[k];
        for (int i = k + 1; i < 6; ++i)
            for (int j = k + 1; j < 6; ++j) A[i][j] -= L[i][k] * D[k] * L[j][k];
    }
    double y[6], z[6];
    for (int i = 0; i < 6; ++i) { double s = b[perm[i]]; for (int j = 0; j < i; ++j) s -= L[i][j] * y[j]; y[i] = s; }
    for (int i = 0; i < 6; ++i) y[i] /= D[i];
    for (int i = 5; i >= 0; --i) { double s = y[i]; for (int j = i + 1; j < 6; ++j) s -= L[j][i] * z[j]; z[i] = s; }
    for (int i = 0; i < 6; ++i) x[perm[i]] = z[i];
}

// TransformVector6dToMatrix4d (Open3D utility/Eigen.cpp): R = Rz(x2) * Ry(x1) * Rx(x0), t = x3..5
M4 vec6_to_mat4(const double x[6]) {
    double ca = std::cos(x[0]), sa = std::sin(x[0]);
    double cb = std::cos(x[1]), sb = std::sin(x[1]);
    double cg = std::cos(x[2]), sg = std::sin(x[2]);
    M4 T = identity4();
    T.a[0][0] = cg * cb; T.a[0][1] = cg * sb * sa - sg * ca; T.a[0][2] = cg * sb * ca + sg * sa;
    T.a[1][0] = sg * cb; T.a[1][1] = sg * sb * sa + cg * ca; T.a[1][2] = sg * sb * ca - cg * sa;
    T.a[2][0] = -sb;     T.a[2][1] = cb * sa;                T.a[2][2] = cb * ca;
    T.a[0][3] = x[3]; T.a[1][3] = x[4]; T.a[2][3] = x[5];
    return T;
}

void transform_points(std::vector<P3>& p, const M4& T) {
    for (P3& q : p) {
        double x = T.a[0][0] * q.x + T.a[0][1] * q.y + T.a[0][2] * q.z + T.a[0][3];
        double y = T.a[1][0] * q.x + T.a[1][1] * q.y + T.a[1][2] * q.z + T.a[1][3];
        double z = T.a[2][0] * q.x + T.a[2][1] * q.y + T.a[2][2] * q.z + T.a[2][3];
        q = {x, y, z};
    }
}

struct Eval { double fitness = 0, rmse = 0; std::vector<int64_t> si, ti; };

Eval evaluate(const std::vector<P3>& src, const KdTree& tree, int64_t nt, double max_corr, int threads) {
    const int64_t ns = (int64_t)src.size();
    std::vector<int64_t> nn(ns, -1);
    std::vector<double> d2(ns, 0.0);
    const double r2 = max_corr * max_corr;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < ns; ++i) {
        int64_t best = -1;
        double bd = 1e300;
        if (nt > 0) tree.nearest(src[i], 0, best, bd);
        if (best >= 0 && bd < r2) { nn[i] = best; d2[i] = bd; }
    }
    Eval e;
    double err = 0;
    for (int64_t i = 0; i < ns; ++i)
        if (nn[i] >= 0) { e.si.push_back(i); e.ti.push_back(nn[i]); err += d2[i]; }
    if (!e.si.empty()) {
        e.fitness = (double)e.si.size() / (double)ns;
        e.rmse = std::sqrt(err / (double)e.si.size());
    }
    return e;
}

}  // namespace

extern "C" {

int32_t gsr_oracle_icp(const double* src_, int64_t ns, const double* tgt_, const double* tgt_normals,
                       int64_t nt, const double* init4x4, int32_t kind, int32_t loss, double k,
                       double max_corr, double rel_fitness, double rel_rmse, int32_t max_iter,
                       int32_t threads, double* out_T, double* out_fitness, double* out_rmse, double* trace) {
    if (!(max_corr > 0)) return -1;
    if (kind == 1 && !tgt_normals) return -2;
    if (ns <= 0 || nt <= 0) return -3;
    std::vector<P3> src(ns), tgt(nt);
    std::memcpy(src.data(), src_, sizeof(P3) * ns);
    std::memcpy(tgt.data(), tgt_, sizeof(P3) * nt);
    const P3* nrm = reinterpret_cast<const P3*>(tgt_normals);
    KdTree tree;
    tree.create(tgt.data(), nt);

    M4 T;
    std::memcpy(&T, init4x4, sizeof(M4));
    bool is_identity = true;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) if (T.a[i][j] != (i == j ? 1.0 : 0.0)) is_identity = false;
    if (!is_identity) transform_points(src, T);

    Eval res = evaluate(src, tree, nt, max_corr, threads);
    auto record = [&](int slot) {
        if (!trace) return;
        double* t = trace + (size_t)slot * 18;
        t[0] = res.fitness; t[1] = res.rmse;
        std::memcpy(t + 2, &T, sizeof(M4));
    };
    record(0);
    int it = 0;
    for (; it < max_iter; ++it) {
        M4 update = identity4();
        const size_t nc = res.si.size();
        if (nc > 0) {
            if (kind == 0) {
                std::vector<P3> a(nc), b(nc);
                for (size_t c = 0; c < nc; ++c) { a[c] = src[res.si[c]]; b[c] = tgt[res.ti[c]]; }
                update = umeyama(a, b);
            } else {
                double JTJ[6][6] = {{0}}, JTr[6] = {0};
                for (size_t c = 0; c < nc; ++c) {
                    const P3& vs = src[res.si[c]];
                    const P3& vt = tgt[res.ti[c]];
                    const P3& n = nrm[res.ti[c]];
                    double r = (vs.x - vt.x) * n.x + (vs.y - vt.y) * n.y + (vs.z - vt.z) * n.z;
                    double w = kernel_weight(loss, k, r);
                    double J[6] = {vs.y * n.z - vs.z * n.y, vs.z * n.x - vs.x * n.z, vs.x * n.y - vs.y * n.x, n.x, n.y, n.z};
                    for (int p = 0; p < 6; ++p) {
                        for (int q = 0; q < 6; ++q) JTJ[p][q] += J[p] * w * J[q];
                        JTr[p] += J[p] * w * r;
                    }
                }
                double nb[6], x[6];
                for (int p = 0; p < 6; ++p) nb[p] = -JTr[p];
                solve6(JTJ, nb, x);
                update = vec6_to_mat4(x);
            }
        }
        T = mul4(update, T);
        transform_points(src, update);
        Eval backup = std::move(res);
        res = evaluate(src, tree, nt, max_corr, threads);
        record(it + 1);
        if (std::fabs(backup.fitness - res.fitness) < rel_fitness && std::fabs(backup.rmse - res.rmse) < rel_rmse) { ++it; break; }
    }
    std::memcpy(out_T, &T, sizeof(M4));
    *out_fitness = res.fitness;
    *out_rmse = res.rmse;
    return it;
}

// Generalized ICP (Open3D 0.16.0 cpp/open3d/pipelines/registration/GeneralizedICP.cpp), as the reference reaches it
// at src/utils/local_registration_util.py:96-98 with clouds that carry per-point covariances
// (point_cloud_converter.py:38), so InitializePointCloudForGeneralizedICP keeps them ("pre-computed covariances").
//   loop      : RegistrationICP (above); PointCloud::Transform also rotates the source covariances, C <- R C R^T
//   estimate  : d = vs - vt, M = Ct + Cs, W = M^-1/2 (principal root), J = W [-skew(vs) | I]  (3 rows),
//               r_i = W.row(i).d, w_i = kernel(r_i); JTJ += w_i J_i J_i^T, JTr += w_i r_i J_i; x = solve(JTJ, -JTr)
// cov arguments: n x 9 float64 (row-major 3x3).
static void sym_eig3(const double A_[3][3], double V[3][3], double lam[3]) {      // cyclic Jacobi, float64
    double A[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { A[i][j] = A_[i][j]; V[i][j] = i == j ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 60; ++sweep) {
        const double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
        const double dg = A[0][0] * A[0][0] + A[1][1] * A[1][1] + A[2][2] * A[2][2];
        if (!(off > 1e-40 * dg)) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (A[p][q] == 0.0) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
                for (int k = 0; k < 3; ++k) {           // A <- A G
                    const double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - sn * akq; A[k][q] = sn * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {           // A <- G^T A
                    const double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - sn * aqk; A[q][k] = sn * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - sn * vkq; V[k][q] = sn * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < 3; ++i) lam[i] = A[i][i];
}
static void inv_sqrt3(const double M[3][3], double W[3][3]) {
    double V[3][3], lam[3];
    sym_eig3(M, V, lam);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += V[i][k] * (1.0 / std::sqrt(lam[k])) * V[j][k];
            W[i][j] = s;
        }
}

int32_t gsr_oracle_gicp(const double* src_, const double* src_cov, int64_t ns, const double* tgt_, const double* tgt_cov,
                        int64_t nt, const double* init4x4, int32_t loss, double k, double max_corr, double rel_fitness,
                        double rel_rmse, int32_t max_iter, int32_t threads, double* out_T, double* out_fitness,
                        double* out_rmse) {
    if (!(max_corr > 0)) return -1;
    if (!src_cov || !tgt_cov) return -2;
    if (ns <= 0 || nt <= 0) return -3;
    std::vector<P3> src(ns), tgt(nt);
    std::memcpy(src.data(), src_, sizeof(P3) * ns);
    std::memcpy(tgt.data(), tgt_, sizeof(P3) * nt);
    std::vector<double> cs(src_cov, src_cov + 9 * ns);
    KdTree tree;
    tree.create(tgt.data(), nt);
    auto rotate_covs = [&](const M4& U) {
        for (int64_t i = 0; i < ns; ++i) {
            double* C = cs.data() + 9 * i;
            double RC[3][3], out[3][3];
            for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) { double s = 0; for (int m = 0; m < 3; ++m) s += U.a[a][m] * C[3 * m + b]; RC[a][b] = s; }
            for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) { double s = 0; for (int m = 0; m < 3; ++m) s += RC[a][m] * U.a[b][m]; out[a][b] = s; }
            for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) C[3 * a + b] = out[a][b];
        }
    };
    M4 T;
    std::memcpy(&T, init4x4, sizeof(M4));
    bool is_identity = true;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) if (T.a[i][j] != (i == j ? 1.0 : 0.0)) is_identity = false;
    if (!is_identity) { transform_points(src, T); rotate_covs(T); }
    Eval res = evaluate(src, tree, nt, max_corr, threads);
    int it = 0;
    for (; it < max_iter; ++it) {
        M4 update = identity4();
        const size_t nc = res.si.size();
        if (nc > 0) {
            double JTJ[6][6] = {{0}}, JTr[6] = {0};
            for (size_t c = 0; c < nc; ++c) {
                const P3& vs = src[res.si[c]];
                const P3& vt = tgt[res.ti[c]];
                const double* Cs = cs.data() + 9 * res.si[c];
                const double* Ct = tgt_cov + 9 * res.ti[c];
                double M[3][3], W[3][3];
                for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) M[a][b] = Ct[3 * a + b] + Cs[3 * a + b];
                inv_sqrt3(M, W);
                const double d[3] = {vs.x - vt.x, vs.y - vt.y, vs.z - vt.z};
                const double nsk[3][3] = {{0, vs.z, -vs.y}, {-vs.z, 0, vs.x}, {vs.y, -vs.x, 0}};      // -skew(vs)
                for (int i = 0; i < 3; ++i) {
                    double J[6];
                    for (int b = 0; b < 3; ++b) { double s = 0; for (int m = 0; m < 3; ++m) s += W[i][m] * nsk[m][b]; J[b] = s; }
                    for (int b = 0; b < 3; ++b) J[3 + b] = W[i][b];
                    const double r = W[i][0] * d[0] + W[i][1] * d[1] + W[i][2] * d[2];
                    const double w = kernel_weight(loss, k, r);
                    for (int p = 0; p < 6; ++p) {
                        for (int q = 0; q < 6; ++q) JTJ[p][q] += J[p] * w * J[q];
                        JTr[p] += J[p] * w * r;
                    }
                }
            }
            double nb[6], x[6];
            for (int p = 0; p < 6; ++p) nb[p] = -JTr[p];
            solve6(JTJ, nb, x);
            update = vec6_to_mat4(x);
        }
        T = mul4(update, T);
        transform_points(src, update);
        rotate_covs(update);
        Eval backup = std::move(res);
        res = evaluate(src, tree, nt, max_corr, threads);
        if (std::fabs(backup.fitness - res.fitness) < rel_fitness && std::fabs(backup.rmse - res.rmse) < rel_rmse) { ++it; break; }
    }
    std::memcpy(out_T, &T, sizeof(M4));
    *out_fitness = res.fitness;
    *out_rmse = res.rmse;
    return it;
}

// PointCloud::VoxelDownSample (Open3D 0.16.0 cpp/open3d/geometry/PointCloud.cpp), the voxel multiscale path of the
// reference (qt_multiscale_registrator.py:127-128): voxel_min_bound = min_bound - voxel_size / 2,
// index = floor((p - voxel_min_bound) / voxel_size) per axis (float64), every voxel averages its points, colours and
// covariances (sum in insertion order / count).  Open3D emits the voxels in unordered_map order; here they are
// emitted in ascending (ix, iy, iz) order -- the defined order of the build's own output.
// xyz n x 3, color n x 3 or NULL, cov n x 9 or NULL (all float64).  Returns the number of voxels; call with
// out_* = NULL to size the outputs (out_xyz V x 3, out_color V x 3, out_cov V x 9).
int64_t gsr_oracle_voxel_down_sample(const double* xyz, const double* color, const double* cov, int64_t n, double voxel_size,
                                     double* out_xyz, double* out_color, double* out_cov) {
    if (!(voxel_size > 0.0)) return -1;
    if (n <= 0) return 0;
    double mn[3] = {1e300, 1e300, 1e300};
    for (int64_t i = 0; i < n; ++i) for (int a = 0; a < 3; ++a) mn[a] = std::min(mn[a], xyz[3 * i + a]);
    for (int a = 0; a < 3; ++a) mn[a] -= voxel_size * 0.5;
    struct Key { int v[3]; bool operator<(const Key& o) const { return std::lexicographical_compare(v, v + 3, o.v, o.v + 3); } };
    struct Acc { double p[3] = {0, 0, 0}, c[3] = {0, 0, 0}, C[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; int64_t num = 0; };
    std::map<Key, Acc> vox;
    for (int64_t i = 0; i < n; ++i) {
        Key k;
        for (int a = 0; a < 3; ++a) k.v[a] = (int)std::floor((xyz[3 * i + a] - mn[a]) / voxel_size);
        Acc& acc = vox[k];
        for (int a = 0; a < 3; ++a) acc.p[a] += xyz[3 * i + a];
        if (color) for (int a = 0; a < 3; ++a) acc.c[a] += color[3 * i + a];
        if (cov) for (int a = 0; a < 9; ++a) acc.C[a] += cov[9 * i + a];
        acc.num++;
    }
    if (out_xyz) {
        int64_t v = 0;
        for (const auto& kv : vox) {
            const Acc& acc = kv.second;
            const double d = (double)acc.num;
            for (int a = 0; a < 3; ++a) out_xyz[3 * v + a] = acc.p[a] / d;
            if (color && out_color) for (int a = 0; a < 3; ++a) out_color[3 * v + a] = acc.c[a] / d;
            if (cov && out_cov) for (int a = 0; a < 9; ++a) out_cov[9 * v + a] = acc.C[a] / d;
            ++v;
        }
    }
    return (int64_t)vox.size();
}

// Colored ICP (Open3D 0.16.0 cpp/open3d/pipelines/registration/ColoredICP.cpp), reference call site
// src/utils/local_registration_util.py:92-94.
//   target preparation (InitializePointCloudForColoredICP, search = Hybrid(radius 2 max_corr, max_nn 30)):
//       KDTreeFlann::SearchHybrid = the 30 nearest neighbours (the point itself first), cut at d^2 < radius^2;
//       with nn >= 4 found: intensity it = mean(rgb); for neighbours 1..nn-1: row = (proj(v_adj) - vt), rhs = it_adj - it,
//       proj = v_adj - ((v_adj - vt).nt) nt; last row = (nn-1) nt, rhs 0; gradient = solve(A^T A, A^T b) (LDLT)
//   per pair, two rows (lambda_geometric = 0.968):
//       r0 = sqrt(lg) (vs - vt).nt,                       J0 = sqrt(lg) [vs x nt, nt]
//       r1 = sqrt(1-lg) (is - (dit.(vs_proj - vt) + it)), J1 = sqrt(1-lg) [vs x ditM, ditM],  ditM = -dit^T (I - nt nt^T)
//   weights kernel(r), 6x6 solve, loop = RegistrationICP.
static void solve3(const double A_[3][3], const double b_[3], double x[3]) {       // Gaussian elimination, partial pivoting
    double A[3][4];
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) A[i][j] = A_[i][j]; A[i][3] = b_[i]; }
    for (int c = 0; c < 3; ++c) {
        int piv = c;
        for (int r = c + 1; r < 3; ++r) if (std::fabs(A[r][c]) > std::fabs(A[piv][c])) piv = r;
        for (int j = 0; j < 4; ++j) std::swap(A[c][j], A[piv][j]);
        for (int r = c + 1; r < 3; ++r) {
            const double f = A[r][c] / A[c][c];
            for (int j = c; j < 4; ++j) A[r][j] -= f * A[c][j];
        }
    }
    for (int i = 2; i >= 0; --i) {
        double v = A[i][3];
        for (int j = i + 1; j < 3; ++j) v -= A[i][j] * x[j];
        x[i] = v / A[i][i];
    }
}

// color gradient of every target point; out n x 3
void gsr_oracle_color_gradient(const double* tgt_, const double* tgt_normals, const double* tgt_colors, int64_t nt, double radius,
                               int32_t max_nn, int32_t threads, double* out) {
    std::vector<P3> tgt(nt);
    std::memcpy(tgt.data(), tgt_, sizeof(P3) * nt);
    const P3* nrm = reinterpret_cast<const P3*>(tgt_normals);
    KdTree tree;
    tree.create(tgt.data(), nt);
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t k = 0; k < nt; ++k) {
        out[3 * k] = out[3 * k + 1] = out[3 * k + 2] = 0.0;
        const P3& vt = tgt[k];
        const P3& n = nrm[k];
        const double it = (tgt_colors[3 * k] + tgt_colors[3 * k + 1] + tgt_colors[3 * k + 2]) / 3.0;
        std::vector<KdTree::DI> heap;
        tree.knn(vt, 0, (size_t)max_nn, heap);
        std::sort(heap.begin(), heap.end());
        size_t nn = 0;
        while (nn < heap.size() && heap[nn].first < radius * radius) ++nn;
        if (nn < 4) continue;
        double AtA[3][3] = {{0}}, Atb[3] = {0};
        for (size_t i = 1; i < nn; ++i) {
            const int64_t a = heap[i].second;
            const P3& va = tgt[a];
            const double dn = (va.x - vt.x) * n.x + (va.y - vt.y) * n.y + (va.z - vt.z) * n.z;
            const double row[3] = {va.x - dn * n.x - vt.x, va.y - dn * n.y - vt.y, va.z - dn * n.z - vt.z};
            const double rhs = (tgt_colors[3 * a] + tgt_colors[3 * a + 1] + tgt_colors[3 * a + 2]) / 3.0 - it;
            for (int p = 0; p < 3; ++p) { for (int q = 0; q < 3; ++q) AtA[p][q] += row[p] * row[q]; Atb[p] += row[p] * rhs; }
        }
        const double f = (double)(nn - 1);
        const double row[3] = {f * n.x, f * n.y, f * n.z};
        for (int p = 0; p < 3; ++p) for (int q = 0; q < 3; ++q) AtA[p][q] += row[p] * row[q];
        solve3(AtA, Atb, out + 3 * k);
    }
}

int32_t gsr_oracle_colored_icp(const double* src_, const double* src_colors, int64_t ns, const double* tgt_, const double* tgt_normals,
                               const double* tgt_colors, int64_t nt, const double* init4x4, int32_t loss, double k, double lambda_geometric,
                               double max_corr, double rel_fitness, double rel_rmse, int32_t max_iter, int32_t threads, double* out_T,
                               double* out_fitness, double* out_rmse) {
    if (!(max_corr > 0)) return -1;
    if (!tgt_normals) return -2;
    if (ns <= 0 || nt <= 0) return -3;
    if (!src_colors || !tgt_colors) return -4;
    std::vector<P3> src(ns), tgt(nt);
    std::memcpy(src.data(), src_, sizeof(P3) * ns);
    std::memcpy(tgt.data(), tgt_, sizeof(P3) * nt);
    const P3* nrm = reinterpret_cast<const P3*>(tgt_normals);
    std::vector<double> grad((size_t)nt * 3);
    gsr_oracle_color_gradient(tgt_, tgt_normals, tgt_colors, nt, max_corr * 2.0, 30, threads, grad.data());
    KdTree tree;
    tree.create(tgt.data(), nt);
    const double sg = std::sqrt(lambda_geometric), sp = std::sqrt(1.0 - lambda_geometric);
    M4 T;
    std::memcpy(&T, init4x4, sizeof(M4));
    bool is_identity = true;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) if (T.a[i][j] != (i == j ? 1.0 : 0.0)) is_identity = false;
    if (!is_identity) transform_points(src, T);
    Eval res = evaluate(src, tree, nt, max_corr, threads);
    int it = 0;
    for (; it < max_iter; ++it) {
        M4 update = identity4();
        const size_t nc = res.si.size();
        if (nc > 0) {
            double JTJ[6][6] = {{0}}, JTr[6] = {0};
            for (size_t c = 0; c < nc; ++c) {
                const int64_t cs = res.si[c], ct = res.ti[c];
                const P3& vs = src[cs];
                const P3& vt = tgt[ct];
                const P3& n = nrm[ct];
                const double dn = (vs.x - vt.x) * n.x + (vs.y - vt.y) * n.y + (vs.z - vt.z) * n.z;
                double J[2][6], r[2];
                J[0][0] = sg * (vs.y * n.z - vs.z * n.y); J[0][1] = sg * (vs.z * n.x - vs.x * n.z); J[0][2] = sg * (vs.x * n.y - vs.y * n.x);
                J[0][3] = sg * n.x; J[0][4] = sg * n.y; J[0][5] = sg * n.z;
                r[0] = sg * dn;
                const double pj[3] = {vs.x - dn * n.x - vt.x, vs.y - dn * n.y - vt.y, vs.z - dn * n.z - vt.z};     // vs_proj - vt
                const double is = (src_colors[3 * cs] + src_colors[3 * cs + 1] + src_colors[3 * cs + 2]) / 3.0;
                const double itn = (tgt_colors[3 * ct] + tgt_colors[3 * ct + 1] + tgt_colors[3 * ct + 2]) / 3.0;
                const double* d = grad.data() + 3 * ct;
                const double is0 = d[0] * pj[0] + d[1] * pj[1] + d[2] * pj[2] + itn;
                const double dd = d[0] * n.x + d[1] * n.y + d[2] * n.z;
                const double m[3] = {-(d[0] - dd * n.x), -(d[1] - dd * n.y), -(d[2] - dd * n.z)};                 // -dit^T (I - n n^T)
                J[1][0] = sp * (vs.y * m[2] - vs.z * m[1]); J[1][1] = sp * (vs.z * m[0] - vs.x * m[2]); J[1][2] = sp * (vs.x * m[1] - vs.y * m[0]);
                J[1][3] = sp * m[0]; J[1][4] = sp * m[1]; J[1][5] = sp * m[2];
                r[1] = sp * (is - is0);
                for (int row = 0; row < 2; ++row) {
                    const double w = kernel_weight(loss, k, r[row]);
                    for (int p = 0; p < 6; ++p) {
                        for (int q = 0; q < 6; ++q) JTJ[p][q] += J[row][p] * w * J[row][q];
                        JTr[p] += J[row][p] * w * r[row];
                    }
                }
            }
            double nb[6], x[6];
            for (int p = 0; p < 6; ++p) nb[p] = -JTr[p];
            solve6(JTJ, nb, x);
            update = vec6_to_mat4(x);
        }
        T = mul4(update, T);
        transform_points(src, update);
        Eval backup = std::move(res);
        res = evaluate(src, tree, nt, max_corr, threads);
        if (std::fabs(backup.fitness - res.fitness) < rel_fitness && std::fabs(backup.rmse - res.rmse) < rel_rmse) { ++it; break; }
    }
    std::memcpy(out_T, &T, sizeof(M4));
    *out_fitness = res.fitness;
    *out_rmse = res.rmse;
    return it;
}

int gsr_oracle_icp_correspond(const double* src_, int64_t ns, const double* tgt_, int64_t nt,
                              const double* T4x4, double max_corr, int32_t threads,
                              int64_t* out_idx, double* out_d2) {
    if (ns < 0 || nt < 0) return -1;
    std::vector<P3> src(ns), tgt(nt);
    std::memcpy(src.data(), src_, sizeof(P3) * ns);
    std::memcpy(tgt.data(), tgt_, sizeof(P3) * nt);
    M4 T;
    std::memcpy(&T, T4x4, sizeof(M4));
    transform_points(src, T);
    KdTree tree;
    tree.create(tgt.data(), nt);
    const double r2 = max_corr * max_corr;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < ns; ++i) {
        int64_t best = -1;
        double bd = 1e300;
        if (nt > 0) tree.nearest(src[i], 0, best, bd);
        if (best >= 0 && bd < r2) { out_idx[i] = best; out_d2[i] = bd; }
        else { out_idx[i] = -1; out_d2[i] = 0; }
    }
    return 0;
}

// Open3D EstimateNormals with covariances present: ComputeNormal(cov, fast=true) = FastEigen3x3,
// the eigenvector of the smallest eigenvalue by the robust closed form (Eberly, "A Robust
// Eigensolver for 3x3 Symmetric Matrices"); zero result -> (0,0,1).
namespace {
void cross3(const double a[3], const double b[3], double o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
void eigvec0(const double A[3][3], double ev, double out[3]) {
    double r0[3] = {A[0][0] - ev, A[0][1], A[0][2]};
    double r1[3] = {A[0][1], A[1][1] - ev, A[1][2]};
    double r2[3] = {A[0][2], A[1][2], A[2][2] - ev};
    double c01[3], c02[3], c12[3];
    cross3(r0, r1, c01); cross3(r0, r2, c02); cross3(r1, r2, c12);
    double d0 = c01[0] * c01[0] + c01[1] * c01[1] + c01[2] * c01[2];
    double d1 = c02[0] * c02[0] + c02[1] * c02[1] + c02[2] * c02[2];
    double d2 = c12[0] * c12[0] + c12[1] * c12[1] + c12[2] * c12[2];
    double dmax = d0; const double* best = c01;
    if (d1 > dmax) { dmax = d1; best = c02; }
    if (d2 > dmax) { dmax = d2; best = c12; }
    double inv = 1.0 / std::sqrt(dmax);
    out[0] = best[0] * inv; out[1] = best[1] * inv; out[2] = best[2] * inv;
}
void eigvec1(const double A[3][3], const double e0[3], double ev1, double out[3]) {
    double U[3], V[3];
    if (std::fabs(e0[0]) > std::fabs(e0[1])) {
        double inv = 1.0 / std::sqrt(e0[0] * e0[0] + e0[2] * e0[2]);
        U[0] = -e0[2] * inv; U[1] = 0; U[2] = e0[0] * inv;
    } else {
        double inv = 1.0 / std::sqrt(e0[1] * e0[1] + e0[2] * e0[2]);
        U[0] = 0; U[1] = e0[2] * inv; U[2] = -e0[1] * inv;
    }
    cross3(e0, U, V);
    double AU[3], AV[3];
    for (int i = 0; i < 3; ++i) {
        AU[i] = A[i][0] * U[0] + A[i][1] * U[1] + A[i][2] * U[2];
        AV[i] = A[i][0] * V[0] + A[i][1] * V[1] + A[i][2] * V[2];
    }
    double m00 = U[0] * AU[0] + U[1] * AU[1] + U[2] * AU[2] - ev1;
    double m01 = U[0] * AV[0] + U[1] * AV[1] + U[2] * AV[2];
    double m11 = V[0] * AV[0] + V[1] * AV[1] + V[2] * AV[2] - ev1;
    double a00 = std::fabs(m00), a01 = std::fabs(m01), a11 = std::fabs(m11);
    if (a00 >= a11) {
        if (std::max(a00, a01) > 0) {
            if (a00 >= a01) { m01 /= m00; m00 = 1 / std::sqrt(1 + m01 * m01); m01 *= m00; }
            else { m00 /= m01; m01 = 1 / std::sqrt(1 + m00 * m00); m00 *= m01; }
            for (int i = 0; i < 3; ++i) out[i] = m01 * U[i] - m00 * V[i];
        } else for (int i = 0; i < 3; ++i) out[i] = U[i];
    } else {
        if (std::max(a11, a01) > 0) {
            if (a11 >= a01) { m01 /= m11; m11 = 1 / std::sqrt(1 + m01 * m01); m01 *= m11; }
            else { m11 /= m01; m01 = 1 / std::sqrt(1 + m11 * m11); m11 *= m01; }
            for (int i = 0; i < 3; ++i) out[i] = m11 * U[i] - m01 * V[i];
        } else for (int i = 0; i < 3; ++i) out[i] = U[i];
    }
}
void fast_eigen_min(const double C[3][3], double out[3]) {
    double A[3][3];
    double mx = C[0][0];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) mx = std::max(mx, C[i][j]);
    if (mx == 0) { out[0] = out[1] = out[2] = 0; return; }
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) A[i][j] = C[i][j] / mx;
    double norm = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
    if (norm > 0) {
        double q = (A[0][0] + A[1][1] + A[2][2]) / 3;
        double b00 = A[0][0] - q, b11 = A[1][1] - q, b22 = A[2][2] - q;
        double p = std::sqrt((b00 * b00 + b11 * b11 + b22 * b22 + norm * 2) / 6);
        double c00 = b11 * b22 - A[1][2] * A[1][2];
        double c01 = A[0][1] * b22 - A[1][2] * A[0][2];
        double c02 = A[0][1] * A[1][2] - b11 * A[0][2];
        double det = (b00 * c00 - A[0][1] * c01 + A[0][2] * c02) / (p * p * p);
        double half_det = std::min(std::max(det * 0.5, -1.0), 1.0);
        double angle = std::acos(half_det) / 3.0;
        const double two_thirds_pi = 2.09439510239319549;
        double beta2 = std::cos(angle) * 2;
        double beta0 = std::cos(angle + two_thirds_pi) * 2;
        double beta1 = -(beta0 + beta2);
        double ev[3] = {q + p * beta0, q + p * beta1, q + p * beta2};
        double e0[3], e1[3], e2[3];
        if (half_det >= 0) {
            eigvec0(A, ev[2], e2);
            if (ev[2] < ev[0] && ev[2] < ev[1]) { std::memcpy(out, e2, 24); return; }
            eigvec1(A, e2, ev[1], e1);
            if (ev[1] < ev[0] && ev[1] < ev[2]) { std::memcpy(out, e1, 24); return; }
            cross3(e1, e2, out);
        } else {
            eigvec0(A, ev[0], e0);
            if (ev[0] < ev[1] && ev[0] < ev[2]) { std::memcpy(out, e0, 24); return; }
            eigvec1(A, e0, ev[1], e1);
            if (ev[1] < ev[0] && ev[1] < ev[2]) { std::memcpy(out, e1, 24); return; }
            cross3(e0, e1, out);
        }
    } else {
        out[0] = out[1] = out[2] = 0;
        if (A[0][0] < A[1][1] && A[0][0] < A[2][2]) out[0] = 1;
        else if (A[1][1] < A[0][0] && A[1][1] < A[2][2]) out[1] = 1;
        else out[2] = 1;
    }
}
}  // namespace

void gsr_oracle_normals_from_cov(const double* cov3x3, int64_t n, double* out) {
    for (int64_t i = 0; i < n; ++i) {
        double C[3][3];
        std::memcpy(C, cov3x3 + 9 * i, sizeof(C));
        double v[3];
        fast_eigen_min(C, v);
        double nn = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        if (nn == 0.0 || nn != nn) { v[0] = 0; v[1] = 0; v[2] = 1; }
        out[3 * i] = v[0]; out[3 * i + 1] = v[1]; out[3 * i + 2] = v[2];
    }
}

// Open3D PointCloud::EstimateNormals(KDTreeSearchParamKNN(knn), fast_normal_computation = true) on a cloud WITHOUT
// covariances -- what convert_input_pc_to_open3d_pc does to a sparse (COLMAP) input cloud
// (src/utils/point_cloud_converter.py:26, default knn = 30) [upstream-recall of Open3D 0.16 EstimateNormals.cpp]:
// per point the knn nearest points (itself included, ordered by (distance, index)); with >= 3 of them the covariance by
// cumulants (ComputeCovariance: means of x, y, z, xx, xy, ... then E[xy] - E[x] E[y]) else the identity; normal = the
// FastEigen3x3 eigenvector of the smallest eigenvalue, (0, 0, 1) if that is the zero vector.  No orientation step.
void gsr_oracle_normals_knn(const double* pts_, int64_t n, int32_t knn, int32_t threads, double* out) {
    std::vector<P3> pts(n);
    std::memcpy(pts.data(), pts_, sizeof(P3) * n);
    KdTree tree;
    tree.create(pts.data(), n);
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = 0; i < n; ++i) {
        std::vector<KdTree::DI> heap;
        tree.knn(pts[i], 0, (size_t)knn, heap);
        std::sort(heap.begin(), heap.end());
        double C[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
        if (heap.size() >= 3) {
            double c[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            for (const KdTree::DI& e : heap) {
                const P3& p = pts[e.second];
                c[0] += p.x; c[1] += p.y; c[2] += p.z;
                c[3] += p.x * p.x; c[4] += p.x * p.y; c[5] += p.x * p.z; c[6] += p.y * p.y; c[7] += p.y * p.z; c[8] += p.z * p.z;
            }
            for (int k = 0; k < 9; ++k) c[k] /= (double)heap.size();
            C[0][0] = c[3] - c[0] * c[0]; C[1][1] = c[6] - c[1] * c[1]; C[2][2] = c[8] - c[2] * c[2];
            C[0][1] = C[1][0] = c[4] - c[0] * c[1]; C[0][2] = C[2][0] = c[5] - c[0] * c[2]; C[1][2] = C[2][1] = c[7] - c[1] * c[2];
        }
        double v[3];
        fast_eigen_min(C, v);
        const double nn = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        if (nn == 0.0 || nn != nn) { v[0] = 0; v[1] = 0; v[2] = 1; }
        out[3 * i] = v[0]; out[3 * i + 1] = v[1]; out[3 * i + 2] = v[2];
    }
}

}  // extern "C"
