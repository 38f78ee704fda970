// oracle/hem_oracle.cpp -- CPU restatement of the reference's Hierarchical-EM mixture downsampler.
//
// TEST INFRASTRUCTURE ONLY: loaded by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
// leg, as the checker / reported baseline.  Never part of the product path.
//
// Parity: PINNED.  tests/test_oracle_vs_ref.py checks this file bit-for-bit against the reference's
// own compiled extension (oracle/_ref, `make -C oracle ref`) and tests/golden/*.npz hold the
// reference's outputs.  To be bit-equal every float expression below keeps the reference's operand
// order and this file is compiled like the reference (-O2, SSE2 scalar, -ffp-contract=off).
//
// What is restated (all paths relative to /root/reference/src/cpp_ext):
//   initial flags / weights        src/mixture.cpp:287-333        (nvar omitted: output-inert)
//   hem::rand, rand01              include/base.hpp:44-56          -> glibc_rand.h
//   eigenvalues (trig closed form) include/vec.hpp:736-768
//   det / inverse / smat3 products include/vec.hpp:544-551,863-872
//   KLD, SMD, ColorDelta           include/gaussian.hpp:82-85,106-114
//   hash grid + 27-cell search     src/pointindex.cpp:55-108,120-143, include/pointindex.hpp:28-38,101-104
//   level: select / E-step / M-step / orphans / flags / validity   src/mixture.cpp:66-285
//   likelihood                     src/mixture.cpp:54-64, clamp include/base.hpp:24-27
//
// Structure deliberately mirrors the reference so that its timing is a fair CPU stand-in: one grid
// whose cell is the LARGEST parent radius, 27-cell scans, OpenMP on the selection loop only, serial
// E-step and M-step.  Data layout is this repo's own (SoA, no per-component heap vectors).
#include "oracle_api.h"
#include "glibc_rand.h"

#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstring>
#include <unordered_map>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

// ---- scalar helpers with the reference's exact semantics (base.hpp:24-27) ------------------------
inline float ref_min(float a, float b) { return a < b ? a : b; }
inline float ref_max(float a, float b) { return a > b ? a : b; }
inline float ref_clamp(float f, float a, float b) { return ref_max(a, ref_min(f, b)); }
inline int   ref_imin(int a, int b) { return a < b ? a : b; }

struct V3 { float x, y, z; };
struct S6 { float e00, e01, e02, e11, e12, e22; };   // xx xy xz yy yz zz  (vec.hpp:458)

inline V3 sub(const V3& a, const V3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline float dot(const V3& a, const V3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float sqdist(const V3& a, const V3& b) { V3 d = sub(a, b); return dot(d, d); }
// vec.hpp:165-168 calls ::sqrt(double) on a float and narrows; for sqrt that equals sqrtf.
inline float dist(const V3& a, const V3& b) { return (float)std::sqrt((double)sqdist(a, b)); }

// vec.hpp:863-866
inline float det(const S6& c) {
    return -c.e02 * c.e02 * c.e11 + 2 * c.e01 * c.e02 * c.e12 - c.e00 * c.e12 * c.e12
           - c.e01 * c.e01 * c.e22 + c.e00 * c.e11 * c.e22;
}
// vec.hpp:868-872 (cofactors, then times 1/det as smat3::operator/ does, vec.hpp:516-520)
inline S6 inverse(const S6& c) {
    S6 r = {c.e11 * c.e22 - c.e12 * c.e12, c.e02 * c.e12 - c.e01 * c.e22, c.e01 * c.e12 - c.e02 * c.e11,
            c.e00 * c.e22 - c.e02 * c.e02, c.e02 * c.e01 - c.e00 * c.e12, c.e00 * c.e11 - c.e01 * c.e01};
    float inv_s = 1.0f / det(c);
    return {r.e00 * inv_s, r.e01 * inv_s, r.e02 * inv_s, r.e11 * inv_s, r.e12 * inv_s, r.e22 * inv_s};
}
inline V3 mul(const S6& m, const V3& v) {   // vec.hpp:540-543
    return {m.e00 * v.x + m.e01 * v.y + m.e02 * v.z, m.e01 * v.x + m.e11 * v.y + m.e12 * v.z,
            m.e02 * v.x + m.e12 * v.y + m.e22 * v.z};
}
// trace( a * c ) with a = *this, c = argument of smat3::operator*(const smat3&) (vec.hpp:544-551);
// the mat3 result reaches trace() through the implicit mat3 -> smat3 conversion (vec.hpp:468-470).
inline float trace_prod(const S6& a, const S6& c) {
    float m00 = c.e00 * a.e00 + c.e01 * a.e01 + c.e02 * a.e02;
    float m11 = c.e01 * a.e01 + c.e11 * a.e11 + c.e12 * a.e12;
    float m22 = c.e02 * a.e02 + c.e12 * a.e12 + c.e22 * a.e22;
    return m00 + m11 + m22;
}

// vec.hpp:736-768: coefficients in float, trig in double, ascending eigenvalues narrowed to float
inline void eigenvalues(const S6& m, float out[3]) {
    const double inv3 = 0.33333333333333333333333333333333;
    const double root3 = 1.7320508075688772935274463415059;
    double c0 = m.e00 * m.e11 * m.e22 + 2 * m.e01 * m.e02 * m.e12 - m.e00 * m.e12 * m.e12
                - m.e11 * m.e02 * m.e02 - m.e22 * m.e01 * m.e01;
    double c1 = m.e00 * m.e11 - m.e01 * m.e01 + m.e00 * m.e22 - m.e02 * m.e02 + m.e11 * m.e22 - m.e12 * m.e12;
    double c2 = m.e00 + m.e11 + m.e22;
    double c2Div3 = c2 * inv3;
    double aDiv3 = c1 * inv3 - c2Div3 * c2Div3;
    if (aDiv3 > 0.0) aDiv3 = 0.0;
    double mbDiv2 = 0.5 * c0 + c2Div3 * c2Div3 * c2Div3 - 0.5 * c2Div3 * c1;
    double q = mbDiv2 * mbDiv2 + aDiv3 * aDiv3 * aDiv3;
    if (q > 0.0) q = 0.0;
    double magnitude = std::sqrt(-aDiv3);
    double angle = std::atan2(std::sqrt(-q), mbDiv2) * inv3;
    if (angle != angle) angle = 0.0;
    double sn = std::sin(angle), cs = std::cos(angle);
    double ev[3];
    ev[0] = c2Div3 + 2 * magnitude * cs;
    ev[1] = c2Div3 - magnitude * (cs + root3 * sn);
    ev[2] = c2Div3 - magnitude * (cs - root3 * sn);
    double h;
    if (ev[2] < ev[1]) { h = ev[1]; ev[1] = ev[2]; ev[2] = h; }
    if (ev[1] < ev[0]) { h = ev[0]; ev[0] = ev[1]; ev[1] = h; }
    if (ev[2] < ev[1]) { h = ev[1]; ev[1] = ev[2]; ev[2] = h; }
    out[0] = (float)ev[0]; out[1] = (float)ev[1]; out[2] = (float)ev[2];
}

// gaussian.hpp:106-109 (gc = child, gp = parent)
inline float kld(const V3& cm, const S6& cc, const V3& pm, const S6& pc) {
    S6 pinv = inverse(pc);
    V3 d = sub(cm, pm);
    float smd = dot(d, mul(pinv, d));
    return 0.5f * (smd + trace_prod(inverse(pc), cc) - 3.0f - std::log(det(cc) / det(pc)));
}

// One mixture level, SoA.
struct Level {
    int64_t n = 0;
    int F = 0;
    std::vector<V3> mean, color;
    std::vector<S6> cov;
    std::vector<float> opacity, weight, sh;      // sh is n*F
    std::vector<uint8_t> is_parent;
    void resize(int64_t m, int f) {
        n = m; F = f;
        mean.resize(m); color.resize(m); cov.resize(m); opacity.resize(m); weight.resize(m);
        sh.resize((size_t)m * f); is_parent.resize(m);
    }
    void push_from(const Level& o, int64_t i) {
        mean.push_back(o.mean[i]); color.push_back(o.color[i]); cov.push_back(o.cov[i]);
        opacity.push_back(o.opacity[i]); weight.push_back(o.weight[i]);
        sh.insert(sh.end(), o.sh.begin() + (size_t)i * F, o.sh.begin() + (size_t)(i + 1) * F);
        is_parent.push_back(0); ++n;
    }
};

struct I3 { int x, y, z; };
struct I3Hash {
    size_t operator()(const I3& c) const {   // pointindex.cpp:9-12
        return std::hash<unsigned>()(c.x) ^ std::hash<unsigned>()(c.y) ^ std::hash<unsigned>()(c.z);
    }
};
struct I3Eq { bool operator()(const I3& a, const I3& b) const { return a.x == b.x && a.y == b.y && a.z == b.z; } };

// The reference's 3-D hash grid (pointindex.cpp:55-108).  Lists are kept as [start,end) runs of the
// sorted index array; a later run with the same coordinate overwrites an earlier one, as
// `mGrid[coord] = currentList` does.
struct Grid {
    const std::vector<V3>* pts = nullptr;
    V3 bbmin{}, bbmax{};
    I3 gsize{};
    float cell = 0;
    std::vector<unsigned> sorted;
    std::unordered_map<I3, std::pair<unsigned, unsigned>, I3Hash, I3Eq> cells;

    // pointindex.hpp:101-104.  vec3/float multiplies by the reciprocal (vec.hpp:129-133); the vec3i
    // min clamps z against the *y* bound (vec.hpp:76) -- kept, it is what the reference computes.
    inline I3 coord(const V3& p) const {
        float is = 1.0f / cell;
        V3 d = sub(p, bbmin);
        I3 c = {(int)(d.x * is), (int)(d.y * is), (int)(d.z * is)};
        I3 m = {gsize.x - 1, gsize.y - 1, gsize.z - 1};
        return {ref_imin(c.x, m.x), ref_imin(c.y, m.y), ref_imin(c.z, m.y)};
    }

    void create(const std::vector<V3>& points, float maxSearchRadius) {
        pts = &points;
        cells.clear();
        bbmin = {FLT_MAX, FLT_MAX, FLT_MAX};
        bbmax = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        for (const V3& p : points) {
            bbmin = {ref_min(bbmin.x, p.x), ref_min(bbmin.y, p.y), ref_min(bbmin.z, p.z)};
            bbmax = {ref_max(bbmax.x, p.x), ref_max(bbmax.y, p.y), ref_max(bbmax.z, p.z)};
        }
        cell = maxSearchRadius;
        V3 bbsize = sub(bbmax, bbmin);
        float is = 1.0f / cell;
        gsize = {(int)(bbsize.x * is) + 1, (int)(bbsize.y * is) + 1, (int)(bbsize.z * is) + 1};
        bbsize = {(float)gsize.x * cell, (float)gsize.y * cell, (float)gsize.z * cell};
        V3 half = {bbsize.x * 0.5f, bbsize.y * 0.5f, bbsize.z * 0.5f};
        V3 center = {(bbmax.x + bbmin.x) * 0.5f, (bbmax.y + bbmin.y) * 0.5f, (bbmax.z + bbmin.z) * 0.5f};
        bbmin = sub(center, half);
        bbmax = {center.x + half.x, center.y + half.y, center.z + half.z};

        sorted.resize(points.size());
        for (unsigned i = 0; i < sorted.size(); ++i) sorted[i] = i;
        // same algorithm (libstdc++ introsort) and the same strict-weak order on cell coordinates as
        // pointindex.cpp:86, so the within-cell order of indices -- and with it the float summation
        // order of the M-step -- comes out identical.
        std::sort(sorted.begin(), sorted.end(), [this](const unsigned& a, const unsigned& b) {
            I3 ca = coord((*pts)[a]), cb = coord((*pts)[b]);
            if (ca.x != cb.x) return ca.x < cb.x;
            if (ca.y != cb.y) return ca.y < cb.y;
            return ca.z < cb.z;
        });
        if (sorted.empty()) return;
        I3 cur = coord(points[sorted[0]]);
        unsigned start = 0;
        for (unsigned k = 0; k < sorted.size(); ++k) {
            I3 c = coord(points[sorted[k]]);
            if (!(c.x == cur.x && c.y == cur.y && c.z == cur.z)) {
                cells[cur] = {start, k};
                start = k;
                cur = c;
            }
        }
        cells[cur] = {start, (unsigned)sorted.size()};
    }

    // ---- accelerated search (gsr_oracle_hem_set_fast_search): the SAME result list, found through a finer grid ------------
    // The reference's list is { i : coord(p_i) - coord(q) in {-1,0,1}^3  and  sqdist(q, p_i) < R*R }, in the order of the 27-cell
    // scan (dz, dy, dx ascending) and, inside a cell, of the sorted index array.  With cell = the LARGEST radius a cell of the
    // 5 M-splat bench cloud holds ~60 000 points and a parent tests 1.6 M of them for the ~200 it keeps.  The fast path finds the
    // points with sqdist < R*R (the same float expression) through a uniform grid of ~8 points per cell, keeps those whose
    // REFERENCE cell is one of the 27 (the reference's own coord(), clamp quirk included), and orders them by (scan position of
    // that cell, position in the reference's sorted array): element for element the list radius_search() returns.  Only for
    // clouds of finite coordinates (anything else keeps the plain scan).  tests/test_oracle_golden.py checks both searches
    // against the golden vectors; tests/golden/make_golden_5m.py shows equality with oracle/_ref at 1 M before it is used at 5 M.
    std::vector<unsigned> pos;        // pos[sorted[k]] = k
    std::vector<I3> pcoord;           // coord() of every point
    std::vector<unsigned> fstart, fidx;
    double fmin[3] = {0, 0, 0}, finv = 0;
    int fdim[3] = {0, 0, 0};
    bool fast_ready = false;

    void build_fast() {
        fast_ready = false;
        const std::vector<V3>& P = *pts;
        const size_t n = P.size();
        if (n == 0 || n >= 0xffffffffull) return;
        double mn[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, mx[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
        for (const V3& p : P) {
            if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) return;
            const double c[3] = {p.x, p.y, p.z};
            for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], c[a]); mx[a] = std::max(mx[a], c[a]); }
        }
        const double ex = mx[0] - mn[0], ey = mx[1] - mn[1], ez = mx[2] - mn[2];
        const double emax = std::max(ex, std::max(ey, ez));
        if (!(emax > 0)) return;
        double c = std::cbrt(std::max(ex, emax * 1e-3) * std::max(ey, emax * 1e-3) * std::max(ez, emax * 1e-3) * 8.0 / (double)n);
        for (;;) {
            double tot = 1;
            for (int a = 0; a < 3; ++a) { fdim[a] = (int)std::floor((mx[a] - mn[a]) / c) + 1; tot *= fdim[a]; }
            if (tot <= 6.4e7) break;
            c *= 1.26;
        }
        finv = 1.0 / c;
        for (int a = 0; a < 3; ++a) fmin[a] = mn[a];
        const size_t nc = (size_t)fdim[0] * fdim[1] * fdim[2];
        fstart.assign(nc + 1, 0);
        std::vector<unsigned> cell_of(n);
        for (size_t i = 0; i < n; ++i) {
            const int cx = fcell(P[i].x, 0), cy = fcell(P[i].y, 1), cz = fcell(P[i].z, 2);
            cell_of[i] = (unsigned)(((size_t)cz * fdim[1] + cy) * fdim[0] + cx);
            ++fstart[cell_of[i] + 1];
        }
        for (size_t k = 0; k < nc; ++k) fstart[k + 1] += fstart[k];
        fidx.resize(n);
        std::vector<unsigned> cur(fstart.begin(), fstart.end() - 1);
        for (size_t i = 0; i < n; ++i) fidx[cur[cell_of[i]]++] = (unsigned)i;
        pos.resize(n);
        for (size_t k = 0; k < n; ++k) pos[sorted[k]] = (unsigned)k;
        pcoord.resize(n);
        for (size_t i = 0; i < n; ++i) pcoord[i] = coord(P[i]);
        fast_ready = true;
    }
    inline int fcell(double v, int a) const {
        double t = std::floor((v - fmin[a]) * finv);
        if (t < 0) t = 0;
        if (t > fdim[a] - 1) t = fdim[a] - 1;
        return (int)t;
    }
    void radius_search_fast(const V3& q, float radius, std::vector<unsigned>& out, std::vector<uint64_t>& keys) const {
        out.clear();
        keys.clear();
        if (!(radius > 0) || !std::isfinite(q.x) || !std::isfinite(q.y) || !std::isfinite(q.z)) return;   // every `d2 < R*R` fails / no cell holds q
        const float r2 = radius * radius;
        // sqdist < fl(R*R) in float32 implies a true distance below R (1 + 1e-6): the pad covers it with room to spare
        const double R = (double)radius * (1.0 + 1e-4) + 1e-30;
        const I3 cq = coord(q);
        const int x0 = fcell(q.x - R, 0), x1 = fcell(q.x + R, 0), y0 = fcell(q.y - R, 1), y1 = fcell(q.y + R, 1);
        const int z0 = fcell(q.z - R, 2), z1 = fcell(q.z + R, 2);
        for (int z = z0; z <= z1; ++z)
            for (int y = y0; y <= y1; ++y) {
                const size_t row = ((size_t)z * fdim[1] + y) * fdim[0];
                for (unsigned k = fstart[row + x0]; k < fstart[row + x1 + 1]; ++k) {
                    const unsigned i = fidx[k];
                    if (!(sqdist(q, (*pts)[i]) < r2)) continue;
                    const I3& ci = pcoord[i];
                    // the same wrapped integer arithmetic as radius_search(): n = c + d  <=>  d = ci - cq
                    const unsigned dx = (unsigned)ci.x - (unsigned)cq.x + 1u, dy = (unsigned)ci.y - (unsigned)cq.y + 1u,
                                   dz = (unsigned)ci.z - (unsigned)cq.z + 1u;
                    if (dx > 2u || dy > 2u || dz > 2u) continue;
                    keys.push_back(((uint64_t)(dz * 9u + dy * 3u + dx) << 32) | pos[i]);
                }
            }
        std::sort(keys.begin(), keys.end());
        out.reserve(keys.size());
        for (uint64_t k : keys) out.push_back(sorted[(unsigned)(k & 0xffffffffu)]);
    }

    // pointindex.cpp:120-143; offsets in the order of pointindex.hpp:28-38 (x fastest, then y, then z)
    void radius_search(const V3& q, float radius, std::vector<unsigned>& out) const {
        out.clear();
        I3 c = coord(q);
        for (int dz = -1; dz <= 1; ++dz)
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    // a NaN query coordinate converts to INT_MIN (cvttss2si), and the reference's vec3i addition then wraps; the
                    // same wrap here in unsigned arithmetic (defined behaviour: found by UBSan, scripts/sanitize_oracle.sh)
                    I3 n = {(int)((unsigned)c.x + (unsigned)dx), (int)((unsigned)c.y + (unsigned)dy), (int)((unsigned)c.z + (unsigned)dz)};
                    auto it = cells.find(n);
                    if (it == cells.end()) continue;
                    for (unsigned k = it->second.first; k < it->second.second; ++k) {
                        unsigned i = sorted[k];
                        if (sqdist(q, (*pts)[i]) < radius * radius) out.push_back(i);
                    }
                }
    }
};

inline bool isnan3(const V3& v) { return v.x != v.x || v.y != v.y || v.z != v.z; }

double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

struct gsr_oracle_hem {
    float rho, delta, kappa, tau;
    int F;
    GlibcRand rng;
    uint64_t draws = 0;
    std::vector<Level> levels;
    int64_t stats[6] = {0, 0, 0, 0, 0, 0};
    double margins[2] = {0, 0};
    double phase[5] = {0, 0, 0, 0, 0};
    bool fast_search = false;          // gsr_oracle_hem_set_fast_search: the same lists through a finer grid (Grid::radius_search_fast)
    bool used_fast = false;

    // mixture.cpp:54-64
    float likelihood(const Level& L, int64_t s, int64_t i) const {
        float distanceDiff = dist(L.mean[s], L.mean[i]);
        float distWeight = std::exp(-distanceDiff * distanceDiff / (tau * tau));
        float colorDiff = dist(L.color[s], L.color[i]);
        float colorInfluence = std::exp(-colorDiff * colorDiff / (tau * tau));
        return distWeight * L.opacity[i] * colorInfluence * sqrtf(det(L.cov[i]));
    }

    int64_t run_level(int threads) {
        const Level& P = levels.back();
        const int64_t nC = P.n;
        double t0 = now_s();

        // 1. centres, parents, radii (mixture.cpp:70-95)
        std::vector<unsigned> parents;
        std::vector<float> radii;
        float maxR = 0;
        for (int64_t i = 0; i < nC; ++i) {
            if (!P.is_parent[i]) continue;
            parents.push_back((unsigned)i);
            float ev[3];
            eigenvalues(P.cov[i], ev);
            float R = delta * sqrtf(ev[2]);
            radii.push_back(R);
            if (R > maxR) maxR = R;
        }
        // 2. grid over all centres, cell = largest parent radius (mixture.cpp:98-99)
        // With no parent or no positive radius every radius test `d2 < R*R` fails (R is 0 or NaN), so
        // all result sets are empty whatever the (degenerate, cell = 0) grid would look like.
        Grid grid;
        const bool searchable = nC > 0 && maxR > 0;
        if (searchable) grid.create(P.mean, maxR);
        if (searchable && fast_search) grid.build_fast();
        used_fast = searchable && fast_search && grid.fast_ready;
        const bool fast = used_fast;
        double t1 = now_s();

        // 3. child selection (mixture.cpp:102-137) -- the reference's only OpenMP loop
        const int64_t nP = (int64_t)parents.size();
        std::vector<std::vector<unsigned>> child(nP);
        const float colorThr = kappa * kappa * 0.5f;
        const float kldThr = delta * delta * 0.5f;
        int64_t nCand = 0;
        double mKld = 1e300, mCol = 1e300;
#ifdef _OPENMP
        if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel for schedule(static) reduction(+ : nCand) reduction(min : mKld) reduction(min : mCol)
        for (int64_t s_ = 0; s_ < nP; ++s_) {
            const int64_t s = parents[s_];
            std::vector<unsigned> result;
            std::vector<uint64_t> keys;
            if (fast) grid.radius_search_fast(P.mean[s], radii[s_], result, keys);
            else if (searchable) grid.radius_search(P.mean[s], radii[s_], result);
            nCand += (int64_t)result.size();
            for (unsigned i : result) {
                float colorDiff = dist(P.color[i], P.color[s]);          // gaussian.hpp:111-114
                {
                    double m = std::fabs((double)colorDiff - (double)colorThr) / (double)colorThr;
                    if (m < mCol) mCol = m;
                }
                if (colorDiff > colorThr) continue;
                float k = kld(P.mean[i], P.cov[i], P.mean[s], P.cov[s]);
                {
                    double m = std::fabs((double)k - (double)kldThr) / (double)kldThr;
                    if (m < mKld) mKld = m;                               // NaN never updates the min
                }
                if (k > kldThr) continue;
                if (P.is_parent[i] && s != (int64_t)i) continue;
                child[s_].push_back(i);
            }
        }
        double t2 = now_s();

        // 4. wL_si and their per-child sums, in parent order (mixture.cpp:140-164)
        std::vector<std::vector<float>> wL(nP);
        std::vector<float> sumLw(nC, 0.0f);
        int64_t nPairs = 0;
        for (int64_t s_ = 0; s_ < nP; ++s_) {
            const int64_t s = parents[s_];
            const std::vector<unsigned>& I = child[s_];
            wL[s_].resize(I.size());
            nPairs += (int64_t)I.size();
            for (size_t i_ = 0; i_ < I.size(); ++i_) {
                unsigned i = I[i_];
                float w = P.weight[s] * ref_clamp(likelihood(P, s, i), FLT_MIN, 1e8f);
                wL[s_][i_] = w;
                sumLw[i] += w;
            }
        }
        double t3 = now_s();

        // 5. responsibilities and moment-matching update (mixture.cpp:167-247)
        Level N;
        N.F = F;
        std::vector<float> sumF(F);
        for (int64_t s_ = 0; s_ < nP; ++s_) {
            const int64_t s = parents[s_];
            const std::vector<unsigned>& I = child[s_];
            const V3 pm = P.mean[s];
            float w_s = 0.0f;
            V3 sm = {0, 0, 0}, sc = {0, 0, 0};
            S6 sv = {0, 0, 0, 0, 0, 0};
            float so = 0.0f;
            std::fill(sumF.begin(), sumF.end(), 0.0f);
            for (size_t i_ = 0; i_ < I.size(); ++i_) {
                unsigned i = I[i_];
                if (sumLw[i] == 0.0f) continue;
                float r_is = wL[s_][i_] / sumLw[i];
                float w = r_is * P.weight[i];
                const V3& cm = P.mean[i];
                const V3& cc = P.color[i];
                const S6& cv = P.cov[i];
                w_s += w;
                sm = {sm.x + cm.x * w, sm.y + cm.y * w, sm.z + cm.z * w};
                sc = {sc.x + cc.x * w, sc.y + cc.y * w, sc.z + cc.z * w};
                V3 d = sub(cm, pm);
                sv.e00 += (cv.e00 + d.x * d.x) * w;
                sv.e01 += (cv.e01 + d.x * d.y) * w;
                sv.e02 += (cv.e02 + d.x * d.z) * w;
                sv.e11 += (cv.e11 + d.y * d.y) * w;
                sv.e12 += (cv.e12 + d.y * d.z) * w;
                sv.e22 += (cv.e22 + d.z * d.z) * w;
                so += w * P.opacity[i];
                const float* f = P.sh.data() + (size_t)i * F;      // (F may be 0: no element to take the address of -- UBSan)
                for (int k = 0; k < F; ++k) sumF[k] += f[k] * w;
            }
            float inv_w = 1.0f / w_s;
            V3 mean_s = {sm.x * inv_w, sm.y * inv_w, sm.z * inv_w};
            V3 dm = sub(mean_s, pm);
            S6 cov_s = {sv.e00 * inv_w - dm.x * dm.x, sv.e01 * inv_w - dm.x * dm.y, sv.e02 * inv_w - dm.x * dm.z,
                        sv.e11 * inv_w - dm.y * dm.y, sv.e12 * inv_w - dm.y * dm.z, sv.e22 * inv_w - dm.z * dm.z};
            N.mean.push_back(mean_s);
            N.color.push_back({sc.x * inv_w, sc.y * inv_w, sc.z * inv_w});
            N.cov.push_back(cov_s);
            N.opacity.push_back(inv_w * so);
            N.weight.push_back(w_s);
            for (int k = 0; k < F; ++k) N.sh.push_back(sumF[k] * inv_w);
            N.is_parent.push_back(0);
            ++N.n;
        }
        double t4 = now_s();

        // 6. orphans, in input order (mixture.cpp:250-253)
        int64_t nOrph = 0;
        for (int64_t i = 0; i < nC; ++i)
            if (sumLw[i] == 0.0f) { N.push_from(P, i); ++nOrph; }

        // 7. new parent flags, one hem::rand01() per component in output order (mixture.cpp:256-259)
        const float parentProbability = 1.0f / rho;
        for (int64_t i = 0; i < N.n; ++i) { N.is_parent[i] = rng.hem_rand01() < parentProbability; ++draws; }

        // 8. validity erase, order preserving (mixture.cpp:262-282)
        Level V;
        V.F = F;
        int64_t dropped = 0;
        for (int64_t i = 0; i < N.n; ++i) {
            float d = det(N.cov[i]);
            if (isnan3(N.mean[i]) || d != d || d <= 0) { ++dropped; continue; }
            V.push_from(N, i);
            V.is_parent.back() = N.is_parent[i];
        }
        levels.push_back(std::move(V));
        double t5 = now_s();

        stats[0] = nP; stats[1] = nPairs; stats[2] = nOrph; stats[3] = dropped; stats[4] = nCand;
        stats[5] = (int64_t)draws;
        margins[0] = mKld; margins[1] = mCol;
        phase[0] = t1 - t0; phase[1] = t2 - t1; phase[2] = t3 - t2; phase[3] = t4 - t3; phase[4] = t5 - t4;
        return levels.back().n;
    }
};

extern "C" {

gsr_oracle_hem* gsr_oracle_hem_create(const float* xyz, const float* color, const float* cov6,
                                      const float* opacity, const float* sh, int64_t n, int32_t F,
                                      float rho, float delta, float kappa, float tau,
                                      uint32_t rng_seed, uint64_t rng_skip) {
    if (n < 0 || F < 0) return nullptr;
    gsr_oracle_hem* h = new gsr_oracle_hem();
    h->rho = rho; h->delta = delta; h->kappa = kappa; h->tau = tau; h->F = F;
    h->rng.seed(rng_seed);
    for (uint64_t i = 0; i < rng_skip; ++i) h->rng.hem_rand();
    h->draws = rng_skip;
    Level L;
    L.resize(n, F);
    const float parentProbability = 1.0f / rho;
    for (int64_t i = 0; i < n; ++i) {
        L.mean[i] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
        L.color[i] = {color[3 * i], color[3 * i + 1], color[3 * i + 2]};
        L.cov[i] = {cov6[6 * i], cov6[6 * i + 1], cov6[6 * i + 2], cov6[6 * i + 3], cov6[6 * i + 4], cov6[6 * i + 5]};
        L.opacity[i] = opacity[i];
        L.weight[i] = 1.0f;
        L.is_parent[i] = h->rng.hem_rand01() < parentProbability;   // mixture.cpp:330
        ++h->draws;
    }
    if (F > 0 && n > 0) std::memcpy(L.sh.data(), sh, sizeof(float) * (size_t)n * F);
    h->levels.push_back(std::move(L));
    return h;
}

void gsr_oracle_hem_destroy(gsr_oracle_hem* h) { delete h; }

int gsr_oracle_hem_set_parent_mask(gsr_oracle_hem* h, const uint8_t* mask) {
    if (!h || !mask) return -1;
    Level& L = h->levels.back();
    for (int64_t i = 0; i < L.n; ++i) L.is_parent[i] = mask[i] ? 1 : 0;
    return 0;
}

int gsr_oracle_hem_set_weights(gsr_oracle_hem* h, const float* w) {
    if (!h || !w) return -1;
    Level& L = h->levels.back();
    for (int64_t i = 0; i < L.n; ++i) L.weight[i] = w[i];
    return 0;
}

int64_t gsr_oracle_hem_level(gsr_oracle_hem* h, int32_t threads) {
    if (!h) return -1;
    return h->run_level(threads);
}

int gsr_oracle_hem_set_fast_search(gsr_oracle_hem* h, int32_t on) {
    if (!h) return -1;
    h->fast_search = on != 0;
    return 0;
}
int gsr_oracle_hem_used_fast_search(const gsr_oracle_hem* h) { return h && h->used_fast ? 1 : 0; }

int32_t gsr_oracle_hem_num_levels(const gsr_oracle_hem* h) { return h ? (int32_t)h->levels.size() : -1; }

int64_t gsr_oracle_hem_level_size(const gsr_oracle_hem* h, int32_t level) {
    if (!h || level < 0 || level >= (int32_t)h->levels.size()) return -1;
    return h->levels[level].n;
}

int gsr_oracle_hem_get_level(const gsr_oracle_hem* h, int32_t level, float* xyz, float* color,
                             float* cov6, float* opacity, float* sh, float* weight, uint8_t* is_parent) {
    if (!h || level < 0 || level >= (int32_t)h->levels.size()) return -1;
    const Level& L = h->levels[level];
    if (xyz) std::memcpy(xyz, L.mean.data(), sizeof(V3) * L.n);
    if (color) std::memcpy(color, L.color.data(), sizeof(V3) * L.n);
    if (cov6) std::memcpy(cov6, L.cov.data(), sizeof(S6) * L.n);
    if (opacity) std::memcpy(opacity, L.opacity.data(), sizeof(float) * L.n);
    if (sh && L.F > 0) std::memcpy(sh, L.sh.data(), sizeof(float) * (size_t)L.n * L.F);
    if (weight) std::memcpy(weight, L.weight.data(), sizeof(float) * L.n);
    if (is_parent) std::memcpy(is_parent, L.is_parent.data(), L.n);
    return 0;
}

int gsr_oracle_hem_stats(const gsr_oracle_hem* h, int64_t* out6) {
    if (!h) return -1;
    std::memcpy(out6, h->stats, sizeof(h->stats));
    return 0;
}
int gsr_oracle_hem_margins(const gsr_oracle_hem* h, double* out2) {
    if (!h) return -1;
    out2[0] = h->margins[0]; out2[1] = h->margins[1];
    return 0;
}
int gsr_oracle_hem_phase_times(const gsr_oracle_hem* h, double* out5) {
    if (!h) return -1;
    std::memcpy(out5, h->phase, sizeof(h->phase));
    return 0;
}

void gsr_oracle_rand_stream(uint32_t seed, uint64_t skip, int64_t n, uint32_t* out) {
    GlibcRand g;
    g.seed(seed);
    for (uint64_t i = 0; i < skip; ++i) g.hem_rand();
    for (int64_t i = 0; i < n; ++i) out[i] = g.hem_rand();
}

void gsr_oracle_parent_flags(uint32_t seed, uint64_t skip, float rho, int64_t n, uint8_t* out) {
    GlibcRand g;
    g.seed(seed);
    for (uint64_t i = 0; i < skip; ++i) g.hem_rand();
    const float p = 1.0f / rho;
    for (int64_t i = 0; i < n; ++i) out[i] = g.hem_rand01() < p;
}

void gsr_oracle_eigenvalues(const float* cov6, int64_t n, float* out3) {
    for (int64_t i = 0; i < n; ++i) {
        S6 m = {cov6[6 * i], cov6[6 * i + 1], cov6[6 * i + 2], cov6[6 * i + 3], cov6[6 * i + 4], cov6[6 * i + 5]};
        eigenvalues(m, out3 + 3 * i);
    }
}
void gsr_oracle_det(const float* cov6, int64_t n, float* out) {
    for (int64_t i = 0; i < n; ++i) {
        S6 m = {cov6[6 * i], cov6[6 * i + 1], cov6[6 * i + 2], cov6[6 * i + 3], cov6[6 * i + 4], cov6[6 * i + 5]};
        out[i] = det(m);
    }
}
void gsr_oracle_kld(const float* cm, const float* cc, const float* pm, const float* pc, int64_t n, float* out) {
    for (int64_t i = 0; i < n; ++i) {
        V3 a = {cm[3 * i], cm[3 * i + 1], cm[3 * i + 2]}, b = {pm[3 * i], pm[3 * i + 1], pm[3 * i + 2]};
        S6 ca = {cc[6 * i], cc[6 * i + 1], cc[6 * i + 2], cc[6 * i + 3], cc[6 * i + 4], cc[6 * i + 5]};
        S6 cb = {pc[6 * i], pc[6 * i + 1], pc[6 * i + 2], pc[6 * i + 3], pc[6 * i + 4], pc[6 * i + 5]};
        out[i] = kld(a, ca, b, cb);
    }
}
float gsr_oracle_logf(float x) { return std::log(x); }

}  // extern "C"
