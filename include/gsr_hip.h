/* include/gsr_hip.h -- C ABI of the MI355X (gfx950) registration backend, libgsr_hip.so.
 *
 * Drop-in boundary for the ONE data-parallel hot path of erikszasz/GaussianSplattingRegistration:
 * the Hierarchical-EM Gaussian-mixture downsampler and the per-iteration ICP step.  Plain pointers
 * and sizes only -- no torch, no C++ types.  Every entry point returns 0 on success or a negative
 * GSR_E_* code; gsr_last_error() returns the thread-local message of the last failure.
 *
 * Which reference interface each group replaces (paths relative to the reference repository):
 *
 *   gsr_hem_*   replaces the pybind11 module `mixture_bind` (src/cpp_ext/mixture_bind.cpp:11-61):
 *                 MixtureLevel.CreateMixtureLevel(xyz, colors, opacities, covariance, features)
 *                     src/cpp_ext/include/mixturelevel.hpp:17-22, src/mixturelevel.cpp:14-28  -> gsr_hem_set_level0
 *                 MixtureCreator.CreateMixture(clusterLevel, hemReduction, distanceDelta, colorDelta, decayRate, level)
 *                     src/cpp_ext/mixture_wrapper.hpp:10, mixture_wrapper.cpp:10-18              -> gsr_hem_create + gsr_hem_run_level x clusterLevel,
 *                                                                                                    or gsr_hem_run_levels (all levels in one call)
 *                 MixtureLevel.CreatePythonLists(level)  src/mixturelevel.cpp:30-70             -> gsr_hem_get_level
 *               (arithmetic: src/cpp_ext/src/mixture.cpp:54-64,66-285,287-333; include/gaussian.hpp:82-114;
 *                include/vec.hpp:736-768,863-872; src/pointindex.cpp:55-143; include/base.hpp:24-27,44-56)
 *
 *   gsr_icp_*   replaces what src/utils/local_registration_util.py:76-100 (do_icp_registration) reaches
 *               through open3d==0.16.0 (requirements.txt:3): registration_icp with
 *               TransformationEstimationPointToPoint / PointToPlane(loss) (:39-51, :58-73) and
 *               ICPConvergenceCriteria (:54-55).
 *
 *   gsr_normals_from_cov  replaces the estimate_normals() call on a cloud whose covariances were set
 *               from the splat covariances (src/utils/point_cloud_converter.py:40-43).
 *
 * Memory: every array argument is row-major and contiguous.  `on_device != 0` means the pointer is a
 * HIP device pointer on the context's device (e.g. a PyTorch-ROCm tensor's data_ptr()); otherwise it
 * is host memory and the library stages it.  The caller owns everything it passes and receives; the
 * library owns only its context and workspace, released by the matching *_destroy.
 * Contexts are single-owner and not re-entrant; different contexts may run concurrently on
 * different streams / GPUs.  `stream` is a hipStream_t (NULL = the default stream).
 */
#ifndef GSR_HIP_H
#define GSR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSR_OK              0
#define GSR_E_INVALID      -1   /* bad argument (NULL handle, negative size, wrong state) */
#define GSR_E_HIP          -2   /* a HIP runtime call failed (message carries hipGetErrorString) */
#define GSR_E_NO_DEVICE    -3   /* no gfx950 device visible: the product path has no CPU fallback */
#define GSR_E_PRECONDITION -4   /* ICP preconditions: max_corr <= 0, point-to-plane without normals, empty cloud */

const char* gsr_last_error(void);
/* "gsr_hip <version> gfx950" */
const char* gsr_version(void);
/* Number of visible HIP devices (0 when none; never fails). */
int32_t gsr_device_count(void);

/* ------------------------------------------------------------------------------------ communicator */

/* Multi-GPU runs (BASELINE configs 4 and 5; the reference has no distributed code): one process per GPU,
 * RCCL over xGMI called FROM THE LIBRARY on the context's own stream -- nothing crosses into the host language per ICP
 * iteration or per HEM level.  Rank 0 calls gsr_comm_get_unique_id and hands the GSR_COMM_ID_BYTES bytes to every rank by any
 * means (the Python side broadcasts them with torch.distributed); every rank then calls gsr_comm_create (ncclCommInitRank:
 * collective, blocks until all ranks have called it).  librccl.so.1 is opened lazily: a single-GPU user never loads it.
 * gsr_comm_create_callbacks builds the same object over host-language collectives on DEVICE buffers (test boxes where two
 * ranks share one GPU, which RCCL refuses): a callback is called after the library has synchronised the stream and must have
 * completed when it returns 0. */
typedef struct gsr_comm gsr_comm;
#define GSR_COMM_ID_BYTES 128
#define GSR_DT_F64 0
#define GSR_DT_F32 1
#define GSR_DT_I32 2
#define GSR_DT_U32 3
#define GSR_DT_U64 4
#define GSR_OP_SUM 0
#define GSR_OP_MAX 1
typedef struct gsr_comm_callbacks {
    /* replace dev_buf[count] (elements of GSR_DT_*) by its element-wise GSR_OP_* over the ranks */
    int32_t (*allreduce)(void* dev_buf, int64_t count, int32_t dtype, int32_t op, void* user);
    /* dev_recv[r * bytes_per_rank ...] = rank r's dev_send[0 .. bytes_per_rank) */
    int32_t (*allgather)(const void* dev_send, void* dev_recv, int64_t bytes_per_rank, void* user);
    /* personalised exchange: send_bytes[r] bytes at dev_send + send_off[r] go to rank r, which receives them at its
     * dev_recv + recv_off[this rank]; the arrays have one entry per rank (the entry of the calling rank is handled by the library) */
    int32_t (*exchange)(const void* dev_send, const int64_t* send_off, const int64_t* send_bytes, void* dev_recv,
                        const int64_t* recv_off, const int64_t* recv_bytes, void* user);
    void* user;
} gsr_comm_callbacks;
int32_t gsr_comm_get_unique_id(void* id128);
int32_t gsr_comm_create(gsr_comm** out, const void* id128, int32_t rank, int32_t world, int32_t device);
int32_t gsr_comm_create_callbacks(gsr_comm** out, int32_t rank, int32_t world, int32_t device, const gsr_comm_callbacks* cb);
int32_t gsr_comm_destroy(gsr_comm* comm);
int32_t gsr_comm_rank(const gsr_comm* comm);
int32_t gsr_comm_world(const gsr_comm* comm);
/* The operations themselves (device buffers, enqueued on `stream` with the RCCL transport); exposed for tests and for host code
 * that shares the communicator. */
int32_t gsr_comm_allreduce(gsr_comm* comm, void* dev_buf, int64_t count, int32_t dtype, int32_t op, void* stream);
int32_t gsr_comm_allgather(gsr_comm* comm, const void* dev_send, void* dev_recv, int64_t bytes_per_rank, void* stream);
int32_t gsr_comm_exchange(gsr_comm* comm, const void* dev_send, const int64_t* send_off, const int64_t* send_bytes, void* dev_recv,
                          const int64_t* recv_off, const int64_t* recv_bytes, void* stream);

/* ------------------------------------------------------------------------------------------- HEM */

typedef struct gsr_hem_ctx gsr_hem_ctx;

/* RNG that draws the parent flags (mixture.cpp:256-259,330):
 *   GSR_RNG_GLIBC  glibc TYPE_3 rand() model, eight rand()%16 nibbles per flag (base.hpp:44-56);
 *                  seed 1 + skip 0 replays a fresh reference process.  Parity mode (default).
 *   GSR_RNG_HASH   counter-based hash of (seed, draw index): same distribution, not the same stream. */
#define GSR_RNG_GLIBC 0
#define GSR_RNG_HASH  1

int32_t gsr_hem_create(gsr_hem_ctx** out, int32_t device, void* stream);
int32_t gsr_hem_destroy(gsr_hem_ctx* ctx);

/* hemReduction, distanceDelta, colorDelta, decayRate -- src/params/merge_parameters.py:5-10 */
int32_t gsr_hem_set_params(gsr_hem_ctx* ctx, float hem_reduction, float distance_delta,
                           float color_delta, float decay_rate);
/* rng_skip = number of hem::rand() values already consumed from the stream (lets a second cloud
 * continue the first cloud's stream as qt_gaussian_mixture.py:55,79 does). */
int32_t gsr_hem_set_rng(gsr_hem_ctx* ctx, int32_t mode, uint32_t seed, uint64_t rng_skip);
int32_t gsr_hem_get_rng_position(gsr_hem_ctx* ctx, uint64_t* draws);

/* Level 0: xyz[n*3], color[n*3] (SH DC), cov6[n*6] (xx,xy,xz,yy,yz,zz), opacity[n] (RAW logit),
 * sh[n*F] (SH rest, coefficient-major); float32.  Sets weight = 1 and draws the n initial parent
 * flags (Mixture::initMixture, mixture.cpp:287-333).  F may be 0 (sh may then be NULL).
 * on_device: 0 = host arrays (copied), 1 = device arrays (copied), 2 = device arrays BORROWED without a copy: they must
 * stay valid and unchanged until the next gsr_hem_run_level (or gsr_hem_set_level0 / destroy) returns. */
int32_t gsr_hem_set_level0(gsr_hem_ctx* ctx, const float* xyz, const float* color, const float* cov6,
                           const float* opacity, const float* sh, int64_t n, int32_t F, int32_t on_device);
/* Override internal per-component state of the CURRENT level (either may be NULL):
 * parent_mask[n] (0/1 bytes, consumes no RNG draws) and weight[n].  Host pointers. */
int32_t gsr_hem_set_state(gsr_hem_ctx* ctx, const uint8_t* parent_mask, const float* weight);

/* Work-sharded levels for ONE large cloud on several GPUs (the reference has no such
 * thing).  Every rank holds the full level (replicated data, deterministic replicated grid); rank r of
 * `world` evaluates the contiguous run [P r/world, P (r+1)/world) of the cell-sorted parents -- a spatial
 * slab -- and the level makes two exchanges:
 *   allreduce(dev_f32, count, user)   the per-child sums of wL: replace the float32 DEVICE buffer (n values) by its
 *                                     element-wise sum over the ranks (RCCL all-reduce in the host language);
 *   allgather(send, recv, bytes, user) the merged components: every rank contributes `bytes` bytes at `send` (device),
 *                                     `recv` (device, world * bytes) receives rank r's contribution at offset r * bytes.
 * Both must have completed (or be ordered on the context's stream) when they return 0.
 * world == 1 / NULL callbacks restore the single-GPU level. */
typedef int32_t (*gsr_allreduce_dev_fn)(void* dev_f32, int64_t count, void* user);
typedef int32_t (*gsr_allgather_dev_fn)(const void* dev_send, void* dev_recv, int64_t bytes_per_rank, void* user);
int32_t gsr_hem_set_shard(gsr_hem_ctx* ctx, int32_t rank, int32_t world, gsr_allreduce_dev_fn allreduce,
                          gsr_allgather_dev_fn allgather, void* user);

/* SPATIALLY PARTITIONED levels for one large cloud on several GPUs (BASELINE config 5; the reference has no
 * such thing).  Every rank owns a subset of the cloud -- compact blocks keep the halo small (parallel.block_of), but any disjoint
 * cover works -- and passes its components with their GLOBAL indices (ascending) to gsr_hem_set_level0_part; gsr_hem_run_level
 * then runs the level on owned + ghost components and the result is BIT FOR BIT the single-GPU level, distributed: per level
 *   - integer all-reduces of the bounding box (24 bytes), the axis histograms (12 KB) and two bit maps over the level's global
 *     indices (n_global / 4 bytes): every rank derives the same grid and the same global output ranks;
 *   - one all-gather of two bit masks over the grid's cells (which cells can my parents take regular / irregular candidates from:
 *     the box of the pre-reject ellipsoid / of the search sphere) and the personalised exchange of the halo, in two messages per
 *     neighbour along the same lists: 72-byte rows {packed record, global index, index at the owner} on the context's stream, and
 *     the 4F-byte SH rows -- which only the M-step reads -- on a second stream, overlapped with the grid phase and the selection;
 *   - five small exchanges along the same halo lists for the per-child sums: maxima (u32), the owners' maxima back, 64-bit
 *     fixed-point partial sums, the owners' float32 sums back -- no floating-point value is ever combined across ranks.
 * The new level stays distributed (gsr_hem_get_level returns the owned rows, gsr_hem_get_gids their global indices; ownership
 * follows the parents).  All of it goes through the communicator: RCCL enqueued on the context's stream, or callbacks.
 * A level's rank-local PRECONDITIONS (a rank that owns nothing, 2^30 components, more than 8 ranks) are agreed on with one all-reduce
 * of a status word before its first data collective: every rank returns the same error.  Beyond that ERRORS ARE LOCAL: a rank whose
 * call fails (an allocation, a callback returning non-zero) returns its error code while the
 * other ranks are inside or in front of the next collective -- as with any RCCL program the caller must then abort the process
 * group (ncclCommAbort / tear the job down); the library does not try to agree on a status across ranks.  The same holds for
 * gsr_icp_register with a communicator or an all-reduce callback. */
int32_t gsr_hem_set_comm(gsr_hem_ctx* ctx, gsr_comm* comm);
int32_t gsr_hem_set_level0_part(gsr_hem_ctx* ctx, const float* xyz, const float* color, const float* cov6, const float* opacity,
                                const float* sh, const uint32_t* gid, int64_t n_own, int64_t n_global, int32_t F, int32_t on_device);
int32_t gsr_hem_get_gids(gsr_hem_ctx* ctx, uint32_t* gid, int32_t on_device);
/* of the most recent partitioned level: [0] ghost components received  [1] halo rows sent  [2] bytes received in the halo exchange
 * [3] bytes received in the five sum exchanges  [4] parents of the level over all ranks  [5] orphans over all ranks
 * [6] components erased over all ranks  [7] components of the CURRENT level over all ranks */
int32_t gsr_hem_get_part_stats(gsr_hem_ctx* ctx, int64_t* out8);
/* durations (ms, device events) of the most recent partitioned level's halo exchanges: [0] the 72-byte rows (on the critical path)
 * [1] the SH rows (second stream, overlapped)  [2], [3] reserved */
int32_t gsr_hem_get_part_ms(gsr_hem_ctx* ctx, float* out4);

/* Zero-copy level output ("returning torch tensors, zero-copy on device"): the NEXT gsr_hem_run_level writes its level
 * straight into these caller-owned DEVICE arrays -- xyz[rows*3], color[rows*3], cov6[rows*6], opacity[rows], sh[rows*F] (sh may be
 * NULL when F = 0) -- instead of the context's own buffers, and that memory then IS the current level (borrowed exactly like an
 * on_device = 2 level 0): nothing is copied on the way out, nothing on the way into the level after.  capacity_rows >= the size of
 * the level being reduced is always enough (a level never grows); the call fails cleanly if the new level does not fit.  The
 * arrays hold the level when run_level returns (n_out rows) and must stay valid and unchanged until the run_level AFTER that one
 * (or gsr_hem_set_level0 / destroy) has returned.  One request covers one level; capacity_rows <= 0 withdraws it.
 * (The reference copies every level by value, mixture.cpp:342, and then into Python lists, mixturelevel.cpp:30-70.) */
int32_t gsr_hem_set_output(gsr_hem_ctx* ctx, float* xyz, float* color, float* cov6, float* opacity, float* sh, int64_t capacity_rows);

/* One clustering level on the current level (Mixture::createClusterLevel, mixture.cpp:66-285).
 * n_out = components of the new level (after the validity erase); n_dropped = components erased by
 * it (the reference prints these to cerr, mixture.cpp:270-274).  The new level becomes current.
 * The call returns when the level is complete.  On one GPU it makes ONE host round trip, behind the level's last kernel (the level's
 * own prologue -- grid geometry, parent count -- was computed with its input and travelled with the round trip of gsr_hem_set_level0 /
 * of the level before; every other count stays on the device and every write is clamped to the context's buffers).  When those buffers
 * turn out too small -- a context's first level, a much larger cloud than before -- the level is run again with every buffer sized from
 * counts read back on the way (gsr_hem_get_stats_ex [6], [7]); its input is never written, the result is the same.  GSR_HEM_ASYNC=0
 * always takes that second schedule.  (The reference's level has no such boundary: one function, mixture.cpp:25-35.) */
int32_t gsr_hem_run_level(gsr_hem_ctx* ctx, int64_t* n_out, int64_t* n_dropped);

/* MixtureCreator::CreateMixture(clusterLevel, ...) in ONE call (mixture_wrapper.cpp:10-18: the reference runs every level inside one
 * function and returns the list of levels): `n_levels` clustering levels on the current level, each written straight into caller-owned
 * DEVICE arenas -- xyz[arena_rows*3], color[arena_rows*3], cov6[arena_rows*6], opacity[arena_rows], sh[arena_rows*F] (NULL when
 * F = 0) -- one level behind the other: level k occupies rows [reports[k].offset_rows, + reports[k].rows) of every arena (offsets are
 * multiples of 64 rows, so every level's arrays start 256-byte aligned).  A level needs room for as many rows as its INPUT has while it
 * runs: the call fails cleanly (GSR_E_INVALID, nothing of that level written) when offset + input rows > arena_rows; arena_rows =
 * n_levels x (rows of level 0) always suffices; 1.5 x is enough for a reduction by 3 per level, a cloud of discs and needles (levels keep 60 - 90 % of their input) needs 1.7 x for three levels.  Equivalent to gsr_hem_set_output +
 * gsr_hem_run_level per level -- the same bits -- without a return to the host language between the levels (a Python caller spent
 * ~0.15 ms per level there, a tenth of a 556 k-splat level); reports[k] carries what gsr_hem_get_stats / _stats_ex / _phase_ms /
 * _kernel_ms / _rng_position would have returned after level k.
 * normals (optional, DEVICE, arena_rows*3 float64): the normal of every component of every new level -- the eigenvector of the smallest
 * eigenvalue of its covariance, what `estimate_normals()` gives a cloud whose covariances are set (point_cloud_converter.py:40-43),
 * bit for bit gsr_normals_from_cov -- at the level's row offset: the normals leave with the level, computed
 * on a side stream beside the next level and joined into the context's stream before the call returns.
 * normals0 (optional, DEVICE, rows of the current level * 3 float64): the same for the level the call starts from.
 * The arenas must stay valid and unchanged until the next gsr_hem_run_level(s) / gsr_hem_set_level0 / destroy has returned (the last
 * level of the call is the context's current level, borrowed).  Not for partitioned / sharded levels (gsr_hem_set_comm, _set_shard). */
typedef struct gsr_hem_level_report {
    int64_t offset_rows;        /* first row of the level in the arenas */
    int64_t rows;               /* components of the level (after the validity erase) */
    int64_t dropped;            /* components the validity erase removed */
    uint64_t rng_position;      /* gsr_hem_get_rng_position after the level */
    int64_t stats[8];           /* gsr_hem_get_stats */
    int64_t stats_ex[8];        /* gsr_hem_get_stats_ex */
    float phase_ms[8];          /* gsr_hem_get_phase_ms */
    float kernel_ms[8];         /* gsr_hem_get_kernel_ms */
} gsr_hem_level_report;
int32_t gsr_hem_run_levels(gsr_hem_ctx* ctx, int32_t n_levels, float* xyz, float* color, float* cov6, float* opacity, float* sh,
                           double* normals, double* normals0, int64_t arena_rows,
                           gsr_hem_level_report* reports);

int32_t gsr_hem_level_size(gsr_hem_ctx* ctx, int64_t* n, int32_t* F);
/* Copy the current level into caller buffers (any may be NULL).  weight / is_parent are internal
 * state the reference never exports (mixture.hpp:33-44); offered for single-level checks. */
int32_t gsr_hem_get_level(gsr_hem_ctx* ctx, float* xyz, float* color, float* cov6, float* opacity,
                          float* sh, float* weight, uint8_t* is_parent, int32_t on_device);

/* Counters of the most recent gsr_hem_run_level:
 *  [0] parents  [1] accepted (parent,child) pairs  [2] orphans  [3] dropped  [4] candidates scanned
 *  [5] grid cells  [6] components in  [7] components out */
int32_t gsr_hem_get_stats(gsr_hem_ctx* ctx, int64_t* out8);
/* More counters of the most recent level:  [0] components outside the stage-1 filter's precondition ("irregular": not
 * verified symmetric positive definite with an accurate float32 determinant -- they take the exact gates only)
 * [1] 1 = the one-pass selection ran, 0 = the COUNT + FILL fallback  [2] 1 = a bucket region of the pair partition overflowed (in this level or an
 * earlier one of the context) and the level's sums took the exact partition  [3] heavy parents (candidates scanned > 16 x the mean: cut
 * into work items)  [4] their work items  [5] the largest number of accepted pairs of one parent
 * [6] host round trips of the level (1: the asynchronous schedule; 4-6: the synchronous one)
 * [7] schedule: 1 = no round trip between the level's first and last kernel, 0 = synchronous (buffers sized from counts read back on the way: a
 *     context's first level, partitioned / sharded levels, GSR_HEM_ASYNC=0), 2 = an asynchronous attempt whose buffers were too small, rerun. */
int32_t gsr_hem_get_stats_ex(gsr_hem_ctx* ctx, int64_t* out8);
/* Device time of the phases of the most recent level, in milliseconds (hipEvent pairs on the
 * context's stream):  [0] prep+grid  [1] selection (count+scan+fill)  [2] per-child sums
 * [3] M-step + orphans  [4] flags+validity  [5] whole level
 * [6] the k_select<COUNT> launch alone (two-pass fallback only, else 0)
 * [7] the k_select<SPARSE> launch alone (or k_select<FILL> on the fallback) */
int32_t gsr_hem_get_phase_ms(gsr_hem_ctx* ctx, float* out8);
/* Device time of single kernel launches of the most recent level (hipEvent pairs on the context's stream), milliseconds:
 *  [0] k_select (the light parents' launch; the heavy work items run beside it)  [1] k_mstep  [2] k_partition  [3] k_bucket_sum
 *  [4..7] reserved (0).  What bench.py prices against the roofline. */
int32_t gsr_hem_get_kernel_ms(gsr_hem_ctx* ctx, float* out8);
/* How much of the above a level records (the reference has no counterpart: its only trace is a `cout` per level, mixture.cpp:32).
 * An event between two kernels is a packet of its own on the stream -- 22 per level, 0.1 ms of a 5 M-splat level, 6 % of a 556 k one:
 *   0  nothing (every figure of gsr_hem_get_phase_ms / _kernel_ms reads 0)
 *   1  the whole level and the launches of k_select and k_mstep: phase [5] [6] [7], kernel [0] [1]   (the default)
 *   2  every phase and kernel listed above
 * The environment variable GSR_HEM_TIMING sets the value a new context starts with. */
int32_t gsr_hem_set_timing(gsr_hem_ctx* ctx, int32_t level);

/* ------------------------------------------------------------------------------------------- ICP */

typedef struct gsr_icp_ctx gsr_icp_ctx;

#define GSR_ICP_POINT_TO_POINT 0   /* LocalRegistrationType.ICP_Point_To_Point, local_registration_util.py:33 */
#define GSR_ICP_POINT_TO_PLANE 1   /* LocalRegistrationType.ICP_Point_To_Plane, :34 */
#define GSR_ICP_GENERALIZED    2   /* LocalRegistrationType.ICP_General, :36 (registration_generalized_icp, :96-98) */
#define GSR_ICP_COLORED        3   /* LocalRegistrationType.ICP_Color, :35 (registration_colored_icp, :92-94) */

#define GSR_LOSS_L2     0          /* KernelLossFunctionType.Loss_None or k == 0, :63-64 */
#define GSR_LOSS_TUKEY  1
#define GSR_LOSS_CAUCHY 2
#define GSR_LOSS_GM     3
#define GSR_LOSS_HUBER  4

int32_t gsr_icp_create(gsr_icp_ctx** out, int32_t device, void* stream);
int32_t gsr_icp_destroy(gsr_icp_ctx* ctx);

/* Target cloud: xyz[n*3] float32 (the reference widens the float32 splat positions to float64,
 * point_cloud_converter.py:33), normals[n*3] float64 or NULL (Open3D normals are float64).  Builds
 * the uniform-grid index used for the exact nearest-neighbour search (expanding rings of cells, at
 * most ceil(max_corr / cell) of them).  Call it BEFORE gsr_icp_set_source: the source is kept sorted by
 * this grid, and a new target invalidates the source. */
int32_t gsr_icp_set_target(gsr_icp_ctx* ctx, const float* xyz, const double* normals, int64_t n,
                           double max_corr, int32_t on_device);
int32_t gsr_icp_set_source(gsr_icp_ctx* ctx, const float* xyz, int64_t n, int32_t on_device);
/* Per-point covariances for GSR_ICP_GENERALIZED: cov6[n*6] float64 (xx, xy, xz, yy, yz, zz), in the order of the
 * points last given to gsr_icp_set_target / gsr_icp_set_source (call these after them).  The reference's clouds
 * carry the splats' own covariances (point_cloud_converter.py:38), which Open3D's generalized ICP then uses as
 * they are; the source covariances follow the source under the current transform (C <- R C R^T). */
int32_t gsr_icp_set_target_cov(gsr_icp_ctx* ctx, const double* cov6, int32_t on_device);
int32_t gsr_icp_set_source_cov(gsr_icp_ctx* ctx, const double* cov6, int32_t on_device);
/* Colours for GSR_ICP_COLORED: rgb[n*3] float64 in the order of the points last given to gsr_icp_set_target /
 * gsr_icp_set_source (call after them; the target needs normals).  The target call also prepares the cloud as Open3D's
 * InitializePointCloudForColoredICP does: per point, the 30 nearest neighbours within 2 * max_corr (ordered by
 * distance, then index), a least-squares colour gradient in the tangent plane.  lambda_geometric defaults to 0.968
 * (TransformationEstimationForColoredICP).  gsr_icp_get_color_gradient: the gradients [n*3], host memory, caller order. */
int32_t gsr_icp_set_target_color(gsr_icp_ctx* ctx, const double* rgb, int32_t on_device);
int32_t gsr_icp_set_source_color(gsr_icp_ctx* ctx, const double* rgb, int32_t on_device);
int32_t gsr_icp_set_lambda_geometric(gsr_icp_ctx* ctx, double lambda_geometric);
int32_t gsr_icp_get_color_gradient(gsr_icp_ctx* ctx, double* out);
/* Multi-GPU source split: this rank owns source points, the target is replicated.  `allreduce` is
 * called once per correspondence evaluation with the rank-local accumulator vector (float64[len],
 * host memory) and must replace it by the element-wise sum over ranks (RCCL/gloo all-reduce in the
 * host language).  NULL = single rank.  n_source_global = source points over all ranks. */
typedef int32_t (*gsr_allreduce_fn)(double* buf, int32_t len, void* user);
int32_t gsr_icp_set_allreduce(gsr_icp_ctx* ctx, gsr_allreduce_fn fn, void* user, int64_t n_source_global);
/* The same with a DEVICE buffer: `fn(dev_f64, count, user)` must replace the float64 device vector by its sum over the
 * ranks, ordered on the context's stream (an RCCL all-reduce enqueued on that stream, or a synchronous one).  The
 * iteration loop then stays device resident: per iteration one reduction kernel, the collective on 32 doubles, one
 * kernel that tests convergence and solves on EVERY rank from the identical reduced vector.  A rank may hold an empty
 * shard (gsr_icp_set_source with n = 0).  Installing one kind of callback removes the other; NULL restores single-GPU. */
typedef int32_t (*gsr_allreduce_dev64_fn)(void* dev_f64, int64_t count, void* user);
int32_t gsr_icp_set_allreduce_dev(gsr_icp_ctx* ctx, gsr_allreduce_dev64_fn fn, void* user, int64_t n_source_global);
/* The same through a communicator: per iteration the accumulate kernel (its last workgroup folds the block partials), ONE
 * all-reduce of 32 float64 enqueued on the context's stream (ncclAllReduce with the RCCL transport: no host involvement), and
 * the solve kernel.  comm = NULL restores single-GPU.  The communicator must outlive the context's use of it. */
int32_t gsr_icp_set_comm(gsr_icp_ctx* ctx, gsr_comm* comm, int64_t n_source_global);

/* One correspondence evaluation + accumulator reduction at transform T (row-major 4x4 float64):
 * acc[0]=count, acc[1]=sum d^2, then for point-to-point acc[2..4]=sum p, [5..7]=sum q, [8..16]=sum p q^T
 * (p = transformed source, q = matched target, both relative to the target-bbox centre);
 * for point-to-plane and generalized ICP acc[2..22]=upper triangle of J^T w J (row-major), [23..28]=J^T w r,
 * [29]=sum r^2 (generalized: three residual rows per pair, J = W [-skew(p) | I], W = (Ct + R Cs R^T)^-1/2;
 * colored: a geometric and a photometric row per pair).
 * len(acc) = GSR_ICP_ACC_LEN.  This is the "hot loop" exposed for tests and for RCCL all-reduce. */
#define GSR_ICP_ACC_LEN 32
int32_t gsr_icp_accumulate(gsr_icp_ctx* ctx, const double* T, int32_t kind, int32_t loss, double k,
                           double* acc);

/* registration_icp: iterate until |dfitness| < rel_fitness && |drmse| < rel_rmse or max_iter.
 * out_T row-major 4x4 float64.  iterations = estimator updates applied. */
int32_t gsr_icp_register(gsr_icp_ctx* ctx, const double* init_T, int32_t kind, int32_t loss, double k,
                         double rel_fitness, double rel_rmse, int32_t max_iter,
                         double* out_T, double* fitness, double* inlier_rmse, int32_t* iterations);
/* The same with the two CLOUDS as arguments -- Open3D's own signature, registration_icp(source, target, max_correspondence_distance,
 * init, estimation_method, criteria) (local_registration_util.py:88-90): gsr_icp_set_target + gsr_icp_set_source + gsr_icp_register in
 * one call, without a return to the host language or a stream synchronisation between them; same result, bit for bit.  Point-to-point
 * and point-to-plane (tgt_normals[nt*3] float64, required for the latter; ignored for the former); one process, one GPU (an installed
 * communicator / all-reduce callback is removed).  on_device: 0 = host arrays, 1 = device arrays (read in place). */
int32_t gsr_icp_register_clouds(gsr_icp_ctx* ctx, const float* src_xyz, int64_t ns, const float* tgt_xyz, const double* tgt_normals,
                                int64_t nt, int32_t on_device, double max_corr, const double* init_T, int32_t kind, int32_t loss,
                                double k, double rel_fitness, double rel_rmse, int32_t max_iter, double* out_T, double* fitness,
                                double* inlier_rmse, int32_t* iterations);
/* The coarse-to-fine schedule in one call (MultiScaleRegistratorMixture._register_main_point_clouds, qt_multiscale_registrator.py:197-236: entry k registers
 * the k-th coarsest level of the two clouds, starting from the transform entry k - 1 ended with): gsr_icp_register_clouds per entry without a return to
 * the host language between the entries.  entries[] coarsest first; results[k] = entry k's outcome (its start, its transform, fitness, RMSE, iterations, the
 * device milliseconds of its index build and of its iterations); out_T = the last entry's transform (init_T when n_entries = 0). */
typedef struct gsr_icp_entry {
    const float* src_xyz; int64_t ns;
    const float* tgt_xyz; const double* tgt_normals; int64_t nt;
    double max_corr;            /* max_correspondence_distance of the entry */
    int32_t max_iter; int32_t reserved;
} gsr_icp_entry;
typedef struct gsr_icp_entry_result {
    double init_T[16], T[16];
    double fitness, inlier_rmse;
    int32_t iterations, evaluations;
    float ms_build, ms_iters;
} gsr_icp_entry_result;
int32_t gsr_icp_register_multiscale(gsr_icp_ctx* ctx, int32_t n_entries, const gsr_icp_entry* entries, int32_t on_device, const double* init_T,
                                    int32_t kind, int32_t loss, double k, double rel_fitness, double rel_rmse, gsr_icp_entry_result* results,
                                    double* out_T);
/* Nearest target index (or -1) and squared distance for every source point at transform T. */
int32_t gsr_icp_correspondences(gsr_icp_ctx* ctx, const double* T, int64_t* idx, double* d2);
/* Device milliseconds: [0] target index build, [1] all correspondence/accumulate kernels of the last
 * gsr_icp_register, [2] their count. */
int32_t gsr_icp_get_timing(gsr_icp_ctx* ctx, float* out3);

/* Normals = unit eigenvector of the smallest eigenvalue of each 3x3 splat covariance, computed in
 * float64 from the float32 covariance widened to float64 (as Open3D does on the converted cloud);
 * a zero vector becomes (0,0,1).  cov6[n*6] float32 in, normals[n*3] float64 out. */
int32_t gsr_normals_from_cov(const float* cov6, int64_t n, double* normals, int32_t on_device,
                             int32_t device, void* stream);

/* Estimator solve on the HOST from a reduced accumulator vector (no GPU involved): the 3x3 Jacobi
 * SVD of Eigen::umeyama (point-to-point) or the 6x6 LDL^T solve + Rz*Ry*Rx of Open3D's point-to-plane
 * estimator.  centre[3] = the point the point-to-point sums are relative to (ignored for
 * point-to-plane).  update = row-major 4x4.  Every rank of a multi-GPU run calls this on the same
 * all-reduced vector and so gets the identical update. */
int32_t gsr_icp_solve(const double* acc, int32_t kind, const double* centre, double* update);
/* The centre used by the context's point-to-point sums (target bounding-box centre). */
int32_t gsr_icp_get_centre(gsr_icp_ctx* ctx, double* centre3);

/* Normals of a cloud that has NO covariances -- a sparse (COLMAP) input cloud of the multiscale worker's sparse
 * pre-registration (src/gui/workers/registration/qt_multiscale_registrator.py:74-90, src/utils/file_loader.py:20-30):
 * replaces the estimate_normals() call of convert_input_pc_to_open3d_pc (src/utils/point_cloud_converter.py:9-28), i.e.
 * Open3D's default KDTreeSearchParamKNN(knn = 30) + covariance of the neighbourhood + FastEigen3x3.
 * xyz[n*3] float32, normals[n*3] float64 (host or device as on_device says), knn in [1, 30]. */
int32_t gsr_normals_knn(const float* xyz, int64_t n, int32_t knn, double* normals, int32_t on_device, int32_t device, void* stream);

/* Covariances for generalized ICP on a cloud that carries none: Open3D's InitializePointCloudForGeneralizedICP
 * (GeneralizedICP.cpp) -- C_i = Rx diag(epsilon, 1, 1) Rx^T, Rx = the rotation taking e1 to the point's normal.  The reference
 * reaches it through registration_generalized_icp on its SPARSE input clouds (local_registration_util.py:96-98 called from
 * qt_multiscale_registrator.py:82-85), which have KNN-30 normals (point_cloud_converter.py:26) and no covariances; a cloud
 * without normals gets gsr_normals_knn(knn = 20) first, as Open3D does.  epsilon = 1e-3 is Open3D's default.
 * normals[n*3] float64 in, cov6[n*6] float64 (xx, xy, xz, yy, yz, zz) out; host or device as on_device says. */
int32_t gsr_cov_from_normals(const double* normals, int64_t n, double epsilon, double* cov6, int32_t on_device,
                             int32_t device, void* stream);

/* ------------------------------------------------------------------------------------ level export */

/* Scaling / rotation of every component from its covariance, on the device: replaces
 * GaussianModel.decompose_covariance_matrix + matrices_to_quaternions, which GaussianModel.from_mixture runs on every
 * HEM level (src/models/gaussian_model.py:141-153,242-265; src/utils/general_utils.py:94-100).
 *   GSR_DECOMP_REFERENCE  the reference's arithmetic, bug for bug: scaling[k] = the eigenVALUE whose eigenvector is most
 *                         aligned with axis k (0 if none claims it, the larger eigenvalue if two do), rotation = the
 *                         trace-formula quaternion (w, x, y, z) of the matrix whose row k is the claiming ROW of eigh's
 *                         eigenvector matrix (csrc/model.hip spells it out).
 *   GSR_DECOMP_EXACT      scaling = log standard deviations, rotation = unit quaternion of a proper rotation, such that
 *                         R diag(exp(scaling))^2 R^T reproduces the covariance (what save_ply of a level needs).
 * cov6[n*6] (xx,xy,xz,yy,yz,zz), scaling[n*3], rotation[n*4], matrix[n*9] or NULL (the 3x3 the quaternion was taken from,
 * row-major); float32, all host or all device as on_device says. */
#define GSR_DECOMP_REFERENCE 0
#define GSR_DECOMP_EXACT     1
int32_t gsr_decompose_cov(const float* cov6, int64_t n, int32_t mode, float* scaling, float* rotation, float* matrix,
                          int32_t on_device, int32_t device, void* stream);

/* 3DGS .ply vertex rows -> the level-0 arrays, on the device (replaces the plyfile -> numpy -> torch.tensor(device=
 * "cuda") chain of GaussianModel.from_ply, src/models/gaussian_model.py:98-139, and the covariance it builds, :34-38 +
 * src/utils/general_utils.py:43-80).  rows_dev: n rows of row_bytes bytes as they are in the file (binary little endian), already in
 * HBM -- the host reads the file in chunks into pinned memory and copies them asynchronously; offsets[15] = byte offsets inside a
 * row of  x y z  f_dc_0..2  opacity  scale_0..2  rot_0..3  f_rest_0  (float32 properties; f_rest_0 .. f_rest_(3K-1) consecutive);
 * K = SH-rest coefficients per channel.  Outputs (device, float32): xyz[n*3], color[n*3] (SH DC), sh[n*3K] coefficient-major
 * (the file is channel-major), opacity[n] (raw), scale[n*3] (log), rot[n*4] (w,x,y,z as stored), cov6[n*6] = R diag(exp(scale))^2 R^T.
 * Enqueued on `stream`; does not synchronise. */
int32_t gsr_ply_unpack(const void* rows_dev, int64_t n, int32_t row_bytes, const int32_t* offsets, int32_t K, float* xyz, float* color,
                       float* sh, float* opacity, float* scale, float* rot, float* cov6, int32_t device, void* stream);

/* RANSAC plane search, the data-parallel part: scores ALL candidate planes of one
 * _fit_single_plane call of the reference (src/utils/plane_fitting_util.py:38-69) in one pass over the points.
 * candidates[n_candidates*8] = {n'_0, n'_1, n'_2, d, m_0, m_1, m_2, |n'|}: the plane normal re-normalised as
 * project_point_onto_plane does (:91-96), the offset d, the normal as sampled (used for the alignment test :57-58).
 * counts[c] = points with |distance| < distance_threshold and |<normal_i, m>| > normal_threshold; *best = the first candidate
 * with the strictly largest count (-1 if none has an inlier); best_mask[n] (or NULL) = its inlier mask.
 * xyz / normals[n*3] float32 and best_mask on the host or the device as on_device says; candidates, counts, best: host. */
int32_t gsr_plane_score(const float* xyz, const float* normals, int64_t n, const float* candidates, int32_t n_candidates,
                        float distance_threshold, float normal_threshold, uint32_t* counts, uint8_t* best_mask, int32_t* best,
                        int32_t on_device, int32_t device, void* stream);

/* ------------------------------------------------------------------------------ voxel down-sampling */

/* PointCloud::VoxelDownSample (Open3D 0.16.0 PointCloud.cpp), the first step of the reference's voxel multiscale
 * registration (src/gui/workers/registration/qt_multiscale_registrator.py:127-128): voxel_min_bound = min_bound -
 * voxel_size / 2, index = floor((p - voxel_min_bound) / voxel_size) in float64, every voxel averages its points,
 * covariances and colours (float64 sums in input order / count).  Voxels come out in ascending (ix, iy, iz) order
 * (Open3D: unordered_map order).  xyz[n*3], cov6[n*6] or NULL, color[n*3] or NULL, float32.  The result object
 * holds the float64 means on the device until gsr_voxel_free. */
typedef struct gsr_voxel_result gsr_voxel_result;
int32_t gsr_voxel_down_sample(int32_t device, void* stream, const float* xyz, const float* cov6, const float* color,
                              int64_t n, double voxel_size, int32_t on_device, gsr_voxel_result** out,
                              int64_t* n_voxels);
/* xyz[V*3], cov6[V*6] or NULL, color[V*3] or NULL, float64, host or device memory. */
int32_t gsr_voxel_fetch(gsr_voxel_result* r, double* xyz, double* cov6, double* color, int32_t on_device);
int32_t gsr_voxel_free(gsr_voxel_result* r);

#ifdef __cplusplus
}
#endif
#endif /* GSR_HIP_H */
