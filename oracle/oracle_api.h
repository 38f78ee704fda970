// oracle/oracle_api.h -- C ABI of the CPU oracle.  TEST INFRASTRUCTURE ONLY.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and
// only as the checker / reported baseline.  The product path (gaussiansplattingregistration_amd)
// never imports it and fails loudly when its HIP extension is missing.
//
// HEM half:  parity PINNED  -- bit-for-bit equal to the reference's compiled cpp_ext
//            (oracle/_ref, built from /root/reference/src/cpp_ext) on every committed fixture.
// ICP half:  parity UNPINNED -- Open3D 0.16.0 (requirements.txt:3) is an un-vendored wheel that is
//            not installable here; icp_oracle.cpp restates its published algorithm.
#pragma once
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gsr_oracle_hem gsr_oracle_hem;

// Create a mixture from level-0 arrays (row-major float32, copied).  Draws the level-0 parent flags
// from the context's glibc-compatible stream (seed 1, `rng_skip` hem::rand() values discarded first)
// exactly as Mixture::initMixture does (mixture.cpp:287-333).
gsr_oracle_hem* gsr_oracle_hem_create(const float* xyz, const float* color, const float* cov6,
                                      const float* opacity, const float* sh, int64_t n, int32_t F,
                                      float rho, float delta, float kappa, float tau,
                                      uint32_t rng_seed, uint64_t rng_skip);
void gsr_oracle_hem_destroy(gsr_oracle_hem* h);

// Override the parent flags of the CURRENT (latest) level (n bytes, 0/1).  Does not touch the RNG.
int gsr_oracle_hem_set_parent_mask(gsr_oracle_hem* h, const uint8_t* mask);
// Override the weights of the CURRENT level (n floats).  Used for single-level, cascade-free checks.
int gsr_oracle_hem_set_weights(gsr_oracle_hem* h, const float* w);

// Run one clustering level on the latest level (Mixture::createClusterLevel, mixture.cpp:66-285).
// Returns the new component count (after the validity erase), or -1 on error.
int64_t gsr_oracle_hem_level(gsr_oracle_hem* h, int32_t threads);

// Accelerated neighbour search for LARGE clouds (the 5 M-splat digest, tests/golden/make_golden_5m.py): the result list of every
// parent is element for element the reference's 27-cell scan (same members, same order), found through a finer grid.  Off by
// default; applies to the following levels.  used_fast_search: whether the most recent level took it (finite coordinates only).
int gsr_oracle_hem_set_fast_search(gsr_oracle_hem* h, int32_t on);
int gsr_oracle_hem_used_fast_search(const gsr_oracle_hem* h);

int32_t gsr_oracle_hem_num_levels(const gsr_oracle_hem* h);           // including level 0
int64_t gsr_oracle_hem_level_size(const gsr_oracle_hem* h, int32_t level);
// Copy a level out (any pointer may be NULL).  weight/is_parent are internal state the reference
// keeps across levels but never exports (mixture.hpp:33-44); exposed here for single-level checks.
int gsr_oracle_hem_get_level(const gsr_oracle_hem* h, int32_t level, float* xyz, float* color,
                             float* cov6, float* opacity, float* sh, float* weight, uint8_t* is_parent);

// Diagnostics of the most recent gsr_oracle_hem_level call:
//  out[0] parents, out[1] accepted (parent,child) pairs, out[2] orphans, out[3] dropped (validity erase),
//  out[4] candidates that passed the radius test, out[5] hem::rand() values drawn so far
int gsr_oracle_hem_stats(const gsr_oracle_hem* h, int64_t* out6);
// Smallest |gate - threshold| seen in the most recent level, relative: out[0] KLD gate, out[1] colour gate
int gsr_oracle_hem_margins(const gsr_oracle_hem* h, double* out2);
// Phase wall times of the most recent level (seconds): radii+grid, selection, likelihood, m-step, rest
int gsr_oracle_hem_phase_times(const gsr_oracle_hem* h, double* out5);

// ---- glibc rand model, for tests ----
void gsr_oracle_rand_stream(uint32_t seed, uint64_t skip_hem_rands, int64_t n, uint32_t* out_hem_rand);
void gsr_oracle_parent_flags(uint32_t seed, uint64_t skip_hem_rands, float rho, int64_t n, uint8_t* out);

// ---- smat3 helpers exposed for unit tests ----
void gsr_oracle_eigenvalues(const float* cov6, int64_t n, float* out3);      // vec.hpp:736-768
void gsr_oracle_det(const float* cov6, int64_t n, float* out);               // vec.hpp:863-866
void gsr_oracle_kld(const float* child_mean, const float* child_cov6, const float* parent_mean,
                    const float* parent_cov6, int64_t n, float* out);        // gaussian.hpp:106-109
float gsr_oracle_logf(float x);                                               // libm logf as the reference calls it

// ---- ICP (icp_oracle.cpp) ----
// kind: 0 point-to-point (Umeyama, no scaling), 1 point-to-plane (6x6 Gauss-Newton, robust weight)
// loss: 0 L2, 1 Tukey, 2 Cauchy, 3 GM, 4 Huber (Open3D RobustKernel.cpp semantics), k = loss parameter
// Returns the number of iterations executed (>=0) or a negative error code:
//  -1 max_corr <= 0, -2 point-to-plane without target normals, -3 empty input.
int32_t gsr_oracle_icp(const double* src, int64_t ns, const double* tgt, const double* tgt_normals,
                       int64_t nt, const double* init4x4, int32_t kind, int32_t loss, double k,
                       double max_corr, double rel_fitness, double rel_rmse, int32_t max_iter,
                       int32_t threads, double* out_T4x4, double* out_fitness, double* out_rmse,
                       double* trace /* optional, (max_iter+1) x 18: fitness, rmse, T(16) per evaluation */);
// Generalized ICP (Open3D GeneralizedICP.cpp) with the clouds' own covariances (n x 9 float64, row-major 3x3).
// Same loop, losses and return convention as gsr_oracle_icp; -2 = a covariance array is missing.
int32_t gsr_oracle_gicp(const double* src, const double* src_cov3x3, int64_t ns, const double* tgt, const double* tgt_cov3x3,
                        int64_t nt, const double* init4x4, int32_t loss, double k, double max_corr, double rel_fitness,
                        double rel_rmse, int32_t max_iter, int32_t threads, double* out_T4x4, double* out_fitness,
                        double* out_rmse);
// PointCloud::VoxelDownSample (Open3D PointCloud.cpp) with the voxels emitted in ascending (ix, iy, iz) order; float64
// arrays, color / cov (n x 9) optional.  Returns the voxel count (out_* NULL = count only), -1 for voxel_size <= 0.
int64_t gsr_oracle_voxel_down_sample(const double* xyz, const double* color, const double* cov3x3, int64_t n, double voxel_size,
                                     double* out_xyz, double* out_color, double* out_cov3x3);
// Colored ICP (Open3D ColoredICP.cpp): colour gradient of every target point over its 30 nearest neighbours within
// `radius` (n x 3 out), and the registration itself (lambda_geometric = 0.968 in Open3D; -4 = colours missing).
void gsr_oracle_color_gradient(const double* tgt, const double* tgt_normals, const double* tgt_colors, int64_t nt, double radius,
                               int32_t max_nn, int32_t threads, double* out_gradient);
int32_t gsr_oracle_colored_icp(const double* src, const double* src_colors, int64_t ns, const double* tgt, const double* tgt_normals,
                               const double* tgt_colors, int64_t nt, const double* init4x4, int32_t loss, double k,
                               double lambda_geometric, double max_corr, double rel_fitness, double rel_rmse, int32_t max_iter,
                               int32_t threads, double* out_T4x4, double* out_fitness, double* out_rmse);
// One correspondence evaluation: nearest target index (or -1) and squared distance for every source point.
int gsr_oracle_icp_correspond(const double* src, int64_t ns, const double* tgt, int64_t nt,
                              const double* T4x4, double max_corr, int32_t threads,
                              int64_t* out_idx, double* out_d2);
// Smallest-eigenvalue eigenvector of each 3x3 covariance (Open3D EstimateNormals with covariances set).
void gsr_oracle_normals_from_cov(const double* cov3x3, int64_t n, double* out_normals);
// Open3D EstimateNormals(KDTreeSearchParamKNN(knn)) on a cloud without covariances (a sparse input cloud,
// src/utils/point_cloud_converter.py:9-28).
void gsr_oracle_normals_knn(const double* pts, int64_t n, int32_t knn, int32_t threads, double* out_normals);

#ifdef __cplusplus
}
#endif
