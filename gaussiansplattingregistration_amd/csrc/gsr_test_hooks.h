/* gsr_test_hooks.h -- PRIVATE test hooks of libgsr_hip.so (not part of the drop-in boundary, include/gsr_hip.h).
 * They run the exact device functions the selection kernel uses on caller-supplied values, so that tests/ can pin
 * them against the host libm / the oracle.  Bound by tests through gaussiansplattingregistration_amd/_lib.py:TEST_HOOKS. */
#ifndef GSR_TEST_HOOKS_H
#define GSR_TEST_HOOKS_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* out[i] = the kernels' logf (glibc-compatible, gsr_math.h) of x[i]; host pointers. */
int32_t gsr_debug_logf(const float* x, int64_t n, float* out, int32_t device);
/* out[i] = KL gate value of child i against parent i as the selection kernel computes it; host pointers,
 * child_mean/parent_mean [n*3], child_cov6/parent_cov6 [n*6]. */
int32_t gsr_debug_kld(const float* child_mean, const float* child_cov6, const float* parent_mean,
                      const float* parent_cov6, int64_t n, float* out, int32_t device);
/* The KL gate decision of k_select for s2[i] = (smd + tr) - 3 and q = det_c[i] / det_p[i] against thr: reject[i] = the
 * kernel's decision (1 = KLD > thr), fast_log[i] = v_log_f32(det_c * (1 / det_p)) * ln 2, need_exact[i] = 1 where the decision
 * fell back to the IEEE division + glibc logf.  Host pointers. */
int32_t gsr_debug_kl_gate(const float* s2, const float* det_c, const float* det_p, int64_t n, float thr, uint8_t* reject,
                          float* fast_log, uint8_t* need_exact, int32_t device);
/* The stage-1 filter of k_select on n independent (parent, child) pairs, by the device functions the kernels use: the
 * "regular" predicate of both (is_regular), the parent's filter record (make_filter: white = filter certified, T1 = its bound,
 * clip_on = grid rows clipped to the ellipsoid) and reject[i] = 1 where stage 1 would drop the pair without the exact gates.
 * Host pointers; means [n*3], cov6 [n*6]. */
int32_t gsr_debug_stage1(const float* parent_mean, const float* parent_cov6, const float* child_mean, const float* child_cov6, int64_t n,
                         float kld_thr, uint8_t* parent_regular, uint8_t* child_regular, uint8_t* white, uint8_t* reject, float* T1,
                         uint8_t* clip_on, int32_t device);
#ifdef __cplusplus
}
#endif
#endif /* GSR_TEST_HOOKS_H */
