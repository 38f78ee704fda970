// oracle/sanitize_main.cpp -- TEST INFRASTRUCTURE ONLY: drives the CPU oracle through its C ABI on small synthetic
// inputs so that the restatement can run under AddressSanitizer / UBSan / ThreadSanitizer in the build container
// ("test CPU restatement under -fsanitize=address,undefined and TSan").  GPU sanitizers are not
// available on the pool; this covers the CPU half.  Built by `make -C oracle asan|tsan` together with the oracle's two
// sources; exits non-zero when a result is implausible (the sanitizers abort on their own findings).
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "oracle_api.h"

static uint64_t g_s = 0x9E3779B97F4A7C15ull;
static double urand() { g_s ^= g_s << 13; g_s ^= g_s >> 7; g_s ^= g_s << 17; return (double)(g_s >> 11) / 9007199254740992.0; }
static double nrand() { double u = urand() + 1e-300, v = urand(); return sqrt(-2.0 * log(u)) * cos(6.283185307179586 * v); }

int main(int argc, char** argv) {
    const int threads = argc > 1 ? atoi(argv[1]) : 4;
    const int64_t n = argc > 2 ? atoll(argv[2]) : 6000;
    const int F = 9;
    std::vector<float> xyz(n * 3), col(n * 3), cov(n * 6), op(n), sh(n * F);
    for (int64_t i = 0; i < n; ++i) {
        for (int k = 0; k < 3; ++k) { xyz[3 * i + k] = (float)((urand() * 2 - 1) * 0.8); col[3 * i + k] = (float)(0.5 * nrand()); }
        double s[3], q[4], nq = 0;
        for (int k = 0; k < 3; ++k) s[k] = exp(-2.5 + 0.5 * nrand());
        if (i % 7 == 0) s[2] *= 0.01;                                       // some discs
        for (int k = 0; k < 4; ++k) { q[k] = nrand(); nq += q[k] * q[k]; }
        nq = sqrt(nq);
        const double w = q[0] / nq, x = q[1] / nq, y = q[2] / nq, z = q[3] / nq;
        const double R[3][3] = {{1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)},
                                {2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)},
                                {2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)}};
        int t = 0;
        for (int a = 0; a < 3; ++a)
            for (int b = a; b < 3; ++b) {
                double v = 0;
                for (int k = 0; k < 3; ++k) v += R[a][k] * s[k] * s[k] * R[b][k];
                cov[6 * i + t++] = (float)v;
            }
        op[i] = (float)(2.0 * nrand());
        for (int k = 0; k < F; ++k) sh[i * F + k] = (float)(0.1 * nrand());
    }
    // edge cases the reference drops or clamps: NaN mean, non-PD covariance, negative opacity
    xyz[3 * 5] = NAN;
    cov[6 * 11] = -1.0f;
    op[17] = -3.0f;
    gsr_oracle_hem* h = gsr_oracle_hem_create(xyz.data(), col.data(), cov.data(), op.data(), sh.data(), n, F, 3.0f, 3.0f, 2.5f, 1.0f, 1, 0);
    if (!h) { fprintf(stderr, "create failed\n"); return 2; }
    int64_t sizes[3] = {n, 0, 0};
    for (int l = 0; l < 2; ++l) {
        sizes[l + 1] = gsr_oracle_hem_level(h, threads);
        if (sizes[l + 1] <= 0 || sizes[l + 1] > sizes[l]) { fprintf(stderr, "level %d: %lld components\n", l, (long long)sizes[l + 1]); return 3; }
    }
    int64_t st[6];
    gsr_oracle_hem_stats(h, st);
    const int64_t m = sizes[2];
    std::vector<float> oxyz(m * 3), ocol(m * 3), ocov(m * 6), oop(m), osh(m * F), ow(m);
    std::vector<uint8_t> opar(m);
    gsr_oracle_hem_get_level(h, 2, oxyz.data(), ocol.data(), ocov.data(), oop.data(), osh.data(), ow.data(), opar.data());
    gsr_oracle_hem_destroy(h);
    // ICP: point-to-point and point-to-plane on a moved copy
    const int64_t np = 3000;
    std::vector<double> src(np * 3), tgt(np * 3), nrm(np * 3), c33(np * 9);
    const double ang = 0.03, ca = cos(ang), sa = sin(ang);
    for (int64_t i = 0; i < np; ++i) {
        const double p[3] = {urand() * 2 - 1, urand() * 2 - 1, 0.2 * sin(3 * urand())};
        for (int k = 0; k < 3; ++k) tgt[3 * i + k] = p[k];
        src[3 * i] = ca * p[0] + sa * p[1] + 0.01; src[3 * i + 1] = -sa * p[0] + ca * p[1] - 0.02; src[3 * i + 2] = p[2] + 0.005;
        for (int k = 0; k < 9; ++k) c33[9 * i + k] = 0;
        c33[9 * i] = 0.01; c33[9 * i + 4] = 0.02; c33[9 * i + 8] = 1e-4;
    }
    gsr_oracle_normals_from_cov(c33.data(), np, nrm.data());
    gsr_oracle_normals_knn(tgt.data(), np, 30, threads, nrm.data());
    const double I4[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    double T[16], fit = 0, rmse = 0;
    for (int kind = 0; kind < 2; ++kind) {
        const int it = gsr_oracle_icp(src.data(), np, tgt.data(), nrm.data(), np, I4, kind, kind ? 4 : 0, 0.1, 0.3, 1e-6, 1e-6, 20, threads, T, &fit, &rmse, nullptr);
        if (it < 0 || !(fit > 0.5)) { fprintf(stderr, "icp kind %d: it %d fitness %g\n", kind, it, fit); return 4; }
    }
    if (gsr_oracle_gicp(src.data(), c33.data(), np, tgt.data(), c33.data(), np, I4, 0, 0.0, 0.3, 1e-6, 1e-6, 10, threads, T, &fit, &rmse) < 0) return 5;
    std::vector<double> vx(np * 3);
    const int64_t nv = gsr_oracle_voxel_down_sample(tgt.data(), nullptr, c33.data(), np, 0.1, nullptr, nullptr, nullptr);
    if (nv <= 0 || nv > np) return 6;
    printf("sanitize_main ok: levels %lld -> %lld -> %lld, pairs %lld, icp fitness %.3f, voxels %lld (threads %d)\n", (long long)sizes[0],
           (long long)sizes[1], (long long)sizes[2], (long long)st[1], fit, (long long)nv, threads);
    return 0;
}
