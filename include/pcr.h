/*
 * pcr.h -- C ABI of libpcr.so, the MI355X (gfx950) device library for the ICP
 * fine-registration hot path of fufesou/PointCloudRegistration.
 *
 * This is the drop-in boundary: every entry point below is what a binding on the
 * reference side would call in place of the reference function cited next to it
 * (paths relative to /root/reference; see INTEGRATION.md for the C++ stubs).
 *
 * Conventions
 *   - return 0 (PCR_OK) on success, a negative pcr_status otherwise; never throws,
 *     no C++/STL/torch types cross the boundary; handles are opaque.
 *   - pointers are HOST pointers unless the parameter name ends in _d (device).
 *   - points are AoS double[3n] (x,y,z), exactly CPointCloud::getPosPointer()
 *     (PointCloudRegistration/PointCloud.h:19); indices are uint32_t; R is 3x3
 *     COLUMN-major (Eigen::Matrix3d::data()), T is 3 doubles.
 *   - all arithmetic that decides an index is IEEE fp64 without FMA contraction, in
 *     the reference's operation order (PointCloud.cpp:368-370, Common/KNNSearch.h:137-143).
 *   - neighbour selection is the exact K nearest under the total order (d2, index):
 *     identical to the reference whenever no two candidates are exactly equidistant
 *     from the query; on exact ties the reference keeps the first one visited in
 *     KD-tree order (Common/KNNSearch.h:75) and this library keeps the lowest index.
 *   - a context is used by one host thread at a time; distinct contexts are independent.
 *     A built index is read-only: any context ON THE SAME DEVICE may search it, step on it or fit on it, and any
 *     number of them at the same time (an index must outlive every fit created on it, and is freed by the context
 *     that built it being idle).
 *     Entry points without a _d suffix are synchronous; _d entry points enqueue on the
 *     context's stream and return (call pcr_ctx_synchronize, or order later work on the
 *     same stream).
 */
#ifndef PCR_H
#define PCR_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    PCR_OK = 0,
    PCR_E_INVALID = -1,    /* bad argument (NULL, k == 0, k > n, dim != 3, ...)        */
    PCR_E_HIP = -2,        /* a HIP runtime call failed; pcr_last_error() has the text */
    PCR_E_NOMEM = -3,      /* device or host allocation failed                        */
    PCR_E_UNSUPPORTED = -4,/* k above PCR_MAX_K, index kind not built, ...            */
    PCR_E_EMPTY = -5,      /* no correspondence survived (reference divides by zero:  */
                           /*   RegUtilityFunctions.cpp:71)                           */
    PCR_E_NO_DEVICE = -6   /* no HIP device / wrong architecture                      */
} pcr_status;

#define PCR_NO_INDEX 0xFFFFFFFFu
#define PCR_MAX_K 32u
#define PCR_NUM_PARTIALS 17u

typedef struct pcr_ctx pcr_ctx;
typedef struct pcr_cloud pcr_cloud;
typedef struct pcr_index pcr_index;

const char *pcr_strerror(int status);
/* text of the last failure on this thread (HIP error string, argument that was rejected) */
const char *pcr_last_error(void);
/* library version string; also names the target ISA it was compiled for */
const char *pcr_version(void);
/* Tuning knobs for experiments (no reference counterpart).  Results do not depend on them.
 *   key 2: log2 of the brick edge (in cells) used to order the samples of a step
 *   key 3: 1 = leave K-NN query batches in the caller's order
 *   key 4: (unused: the transformed copy of the source it chose is gone -- the reverse search always moves candidates as it reads them)
 *   key 5: 1 = neither the ICP step nor the K-NN entry points use the fp32-filtered kernels (every candidate goes
 *          through the fp64 walk)
 *   key 6: overrides pcr_index_params.compact_tables for indices built from now on: 0 = no override, 1 = dense tables
 *          only, 2 = every level of a nested index compact, 3 = the levels under 1/8 occupancy compact */
int pcr_tune(int key, int value);

/* ---- context -----------------------------------------------------------------------
 * No reference counterpart (the reference is single-threaded CPU code).  One context =
 * one device + one HIP stream + reusable scratch.  A new context owns a private non-blocking
 * stream; pcr_ctx_set_stream makes it enqueue on a caller's hipStream_t instead (for example
 * torch.cuda.current_stream().cuda_stream; NULL selects the legacy default stream), so that
 * _d entry points are ordered with the caller's own work on that stream. */
int pcr_ctx_create(int device, pcr_ctx **out);
int pcr_ctx_set_stream(pcr_ctx *ctx, void *hip_stream);
int pcr_ctx_destroy(pcr_ctx *ctx);
int pcr_ctx_synchronize(pcr_ctx *ctx);
/* Optional measurement hooks (no reference counterpart; the reference times whole fits with
 * GetTickCount, unit test/FineRegistration/TestFineRegistration.cpp:50-68).  When on, every kernel
 * launch of the classes below is bracketed by a HIP event pair on the context's stream.
 * pcr_ctx_read_kernel_times synchronises, returns per class the summed kernel time in ms and the
 * number of launches since the previous read, and resets the counters.
 * class 0: source transform, 1: fused ICP step, 2: partial fold, 3: K-NN query.
 * on: 0 = off, 1 = every class, otherwise (mask << 1) | (stride << 8) with bit c of mask selecting class c and, for
 * stride > 1, only every stride-th launch of a class bracketed, the first included (an event pair costs microseconds of
 * queue time per launch -- 7 us of a 153 us iteration when every launch is bracketed: a timed loop brackets only the
 * kernel it reports, and only a sample of its launches). */
int pcr_ctx_set_profiling(pcr_ctx *ctx, int on);
int pcr_ctx_read_kernel_times(pcr_ctx *ctx, double sum_ms[4], uint32_t launches[4]);
/* the bracketed launches of one class one by one, in launch order (ms); synchronises, resets nothing; *count = how
 * many there are (the first min(cap, *count) are written) */
int pcr_ctx_read_kernel_launches(pcr_ctx *ctx, int kernel_class, double *ms_out, uint32_t cap, uint32_t *count);

/* ---- clouds ------------------------------------------------------------------------
 * Replaces the borrowed `const double*` of hiveCommon::CKNNSearch::initKNNSearch
 * (Common/KNNSearch.cpp:17-26) and hiveSearch::CBaseSearch::setInputData
 * (PointCloudRegistration/BaseSearch.cpp:27-33).  pcr_cloud_upload COPIES (the caller keeps
 * its memory); pcr_cloud_wrap_d borrows device memory that must outlive the cloud. */
int pcr_cloud_upload(pcr_ctx *ctx, const double *xyz, uint32_t n, pcr_cloud **out);
int pcr_cloud_wrap_d(pcr_ctx *ctx, const double *xyz_d, uint32_t n, pcr_cloud **out);
int pcr_cloud_free(pcr_cloud *cloud);
uint32_t pcr_cloud_size(const pcr_cloud *cloud);
/* axis-aligned bounds lo[3], hi[3] -- hiveCommon::CKNNSearch::__computeBoundingBox,
 * Common/KNNSearch.cpp:96-116 */
int pcr_cloud_bounds(const pcr_cloud *cloud, double lo[3], double hi[3]);

/* ---- index build -------------------------------------------------------------------
 * Replaces hiveCommon::CKNNSearch::initKNNSearch -> __buildSearchIndex
 * (Common/KNNSearch.cpp:17,79-92) and hiveSearch::CKNNSearch::_buildAccelerationStructureV
 * (PointCloudRegistration/NewKNNSearch.cpp:22-28).  The structure is a uniform grid built on
 * the device (cell histogram -> exclusive scan -> scatter into cell order), or nothing at
 * all for the tiled brute-force kernel used on small clouds. */
typedef enum { PCR_INDEX_AUTO = 0, PCR_INDEX_GRID = 1, PCR_INDEX_BRUTE = 2 } pcr_index_kind;
typedef struct {
    int kind;                 /* pcr_index_kind; AUTO = BRUTE for n <= brute_max_points    */
    double target_occupancy;  /* mean points per OCCUPIED cell the build aims at; <=0: 4.5 */
    double cell_size;         /* > 0 fixes the cell edge and skips the occupancy search   */
    uint64_t max_cells;       /* cap on grid cells (dense cell table); 0: 2^30            */
    uint32_t brute_max_points;/* AUTO threshold; 0: 2048                                  */
    uint32_t max_levels;      /* nested grids for clouds of very uneven density (a dense
                               * region gets a finer grid of its own); 0: 6, 1: one grid  */
    uint32_t compact_tables;  /* nested grids only.  0: every level keeps a start per cell
                               * (fastest); 1: a level under 1/8 occupancy keeps occupancy
                               * bits, ranks and a start per OCCUPIED cell instead (a 16 M
                               * LiDAR sweep's index: 2.4 -> 0.9 GB, the iteration 8 % slower);
                               * 2: every level does.  Results do not depend on it.        */
} pcr_index_params;
typedef struct {
    int kind;
    uint32_t n, dims[3];
    uint64_t cells, occupied_cells;
    double cell_size, mean_occupancy;
    uint32_t max_occupancy;
    uint32_t levels;          /* grids in the index; dims and cell_size describe level 0,
                               * cells / occupied_cells / max_occupancy all levels        */
    uint64_t device_bytes;
} pcr_index_info;
int pcr_index_build(pcr_ctx *ctx, const pcr_cloud *cloud, const pcr_index_params *params /*nullable*/,
                    pcr_index **out);
int pcr_index_free(pcr_index *index);
int pcr_index_get_info(const pcr_index *index, pcr_index_info *info);

/* ---- exact K-NN --------------------------------------------------------------------
 * Replaces hiveCommon::CKNNSearch::executeKNN (Common/KNNSearch.cpp:33-43,65-75; ascending
 * distance = PCR_ORDER_ASCENDING) and hiveSearch::CBaseSearch::executeSearch ->
 * CKNNSearch::_executeSearchV (BaseSearch.cpp:37-60, NewKNNSearch.cpp:32-68; row-major
 * [nq x k], farthest first = PCR_ORDER_DESCENDING).  out_d2 (nullable) receives the squared
 * distances in the same order.  max_d2 bounds the search: neighbours with d2 >= max_d2 are
 * not reported (their slots hold PCR_NO_INDEX / +inf); pass a negative value or +inf for an
 * unbounded search (the reference behaviour). */
typedef enum { PCR_ORDER_ASCENDING = 0, PCR_ORDER_DESCENDING = 1 } pcr_order;
int pcr_knn(pcr_ctx *ctx, const pcr_index *index, const double *q, uint32_t nq, uint32_t k, int order,
            double max_d2, uint32_t *out_idx, double *out_d2 /*nullable*/);
int pcr_knn_d(pcr_ctx *ctx, const pcr_index *index, const double *q_d, uint32_t nq, uint32_t k, int order,
              double max_d2, uint32_t *out_idx_d, double *out_d2_d /*nullable*/);

/* ---- CUniqueData::setPointCloud (PointCloudRegistration/UniqueData.cpp:25-53) ----------
 * Mean squared distance to the 4 nearest neighbours (self included) over the ~20 evenly
 * strided points of the cloud the index was built on; reuses the index instead of building a
 * throw-away tree. */
int pcr_unique_sq_dist(pcr_ctx *ctx, const pcr_index *index, double *out);

/* ---- samplers: the step before the loop ---------------------------------------------------------
 * pcr_sample_random  CSamplerRandom::_doSampleV (Sampler/SamplerRandom.cpp:25-51): point i is kept iff u_i < ratio,
 *                    u_i a counter-based hash of (seed, i) where the reference calls an unseeded rand().
 * pcr_sample_grid    one point per occupied cell of the index's grid (the sampler the reference only stubs,
 *                    Sampler/SamplerUniformGrid.cpp:25-28).
 * out_idx (host, `cap` slots) receives ascending source indices, *out_count their number; PCR_E_INVALID if they
 * do not fit. */
int pcr_sample_random(pcr_ctx *ctx, uint32_t n, double ratio, uint64_t seed, uint32_t cap, uint32_t *out_idx,
                      uint32_t *out_count);
/* CSamplerRandom's rule on a rand() stream of the CALLER's (SamplerRandom.cpp:45-48 with ICPMacros.h:4): point i is kept
 * iff (double)draws[i] / (rand_max + 1) < ratio, draws[i] = the i-th value rand() returned, rand_max = that generator's
 * RAND_MAX (32767 for the reference's MSVC runtime).  Given the stream, the reference's sample set index for index. */
int pcr_sample_random_draws(pcr_ctx *ctx, uint32_t n, double ratio, const int32_t *draws, uint32_t rand_max, uint32_t cap,
                            uint32_t *out_idx, uint32_t *out_count);
int pcr_sample_grid(pcr_ctx *ctx, const pcr_index *index, uint32_t cap, uint32_t *out_idx, uint32_t *out_count);

/* ---- one ICRP iteration ------------------------------------------------------------
 * Replaces, fused into device work:
 *   CCorrespondenceEstimationClosestPoints::_determineCorrespondencesV
 *        (FineRegistrationClosestPoints/CorrespondenceEstimationClosestPoints.cpp:54-117)
 *   CPointCloud::transform of the whole source        (PointCloud.cpp:358-371; :73-74)
 *   the per-iteration source KD-tree rebuild           (:77-78) -- replaced by ONE source index
 *        built in the source frame and searched with transformed candidates
 *   CCorrespondenceRejectionClosestPoints (pass-through, CorrespondenceRejectionClosestPoints.cpp:18-22)
 *   compute3DCentroid x2 + __deMean + __buildPointPairMatrix + H = A*B^T
 *        (RegUtilityFunctions.cpp:44-72, TransformationEstimationSVD2.cpp:33-103)
 *   CDefaultConvergenceCriteria::__calculateMSE         (DefaultConvergenceCriterial.cpp:46-65)
 *
 * For each sample i (sample_idx NULL = every source point, in index order):
 *   p = R*s + T;  j = K-th nearest target point of p;  q = tgt[j];
 *   r = nearest transformed source point of q;  p' = R*src[r] + T;
 *   keep = |p-q|^2 < thr_fwd_sq  &&  |p-p'|^2 < thr_rev_sq            (strict <, :97-98)
 * and over the kept samples the 17 partial sums
 *   partials[0]      n
 *   partials[1..3]   sum (p - pivot_p)
 *   partials[4..6]   sum (q - pivot_q)
 *   partials[7..15]  sum (p - pivot_p)(q - pivot_q)^T   row-major, row = p component
 *   partials[16]     sum |p-q|^2
 * with pivot_p = R*c_src + T and pivot_q = c_tgt, c_* = centre of the cloud's bounds
 * (pcr_icp_pivots returns them).  The pivots only condition the sums; pcr_solve_rigid removes them.
 *
 * flags: PCR_STEP_EXACT_INDICES makes out_fwd/out_rev the reference's values for EVERY sample;
 * without it a sample whose forward distance already fails the gate may report PCR_NO_INDEX in
 * out_fwd/out_rev (keep and partials are identical either way).
 * Outputs out_fwd, out_rev, out_keep are nullable.  WHERE a sample's result goes: with a sample list
 * (sample_idx != NULL) the arrays hold m entries and entry t belongs to sample_idx[t]; without one
 * (sample_idx NULL: the samples are source points themselves) the arrays are indexed by ORIGINAL source
 * index and must hold ALL n source entries -- also for the sub-range [sample_begin, sample_begin+m) of
 * pcr_icp_step_d, which is a range of the index's internal processing order, not of original indices:
 * only the entries of the points in that range are written, the others are left untouched. */
#define PCR_STEP_EXACT_INDICES 1u
int pcr_icp_step(pcr_ctx *ctx, const pcr_index *src_index, const pcr_index *tgt_index,
                 const uint32_t *sample_idx /*nullable*/, uint32_t m, const double R[9], const double T[3],
                 uint32_t k, double thr_fwd_sq, double thr_rev_sq, uint32_t flags, double partials[17],
                 uint32_t *out_fwd, uint32_t *out_rev, uint8_t *out_keep);
/* ---- f4, OPT-IN: the normal-aware pair decision CCorrespondenceEstimationClosestPoints::__insertFitPairs
 *      (CorrespondenceEstimationClosestPoints.cpp:128-175; its call is commented out in the reference, :105) ----
 * In place of the reciprocal gate: the K forward neighbours of p = R*s + T are tried nearest first, and the first one
 * whose direction from p is (anti)parallel to its own normal to within nvdt -- |1 - |cos|| < NVDT (0.0125) -- makes the
 * pair; with compare_normals != 0 its normal must also agree with the sample's rotated normal, |n . R n_s| > ndt (0.85;
 * the reference hard-wires that test off).  Normals are attached to the clouds with pcr_cloud_set_normals (host
 * double[3n], original order; copied) or pcr_cloud_set_normals_d (borrowed device pointer); NULL detaches.
 * NOT the reference's observable behaviour: there the function reads a vector it never assigns (TmpTgtPoint, :139,
 * :146, :166) -- here that vector is the neighbour's position.  Grid indices only.  partials/out_keep as in
 * pcr_icp_step; out_tgt (nullable) = the chosen target index or PCR_NO_INDEX, laid out like out_fwd. */
int pcr_cloud_set_normals(pcr_cloud *cloud, const double *normals /*nullable*/);
int pcr_cloud_set_normals_d(pcr_cloud *cloud, const double *normals_d /*nullable*/);
int pcr_icp_step_normal_fit(pcr_ctx *ctx, const pcr_index *src_index, const pcr_index *tgt_index,
                            const uint32_t *sample_idx /*nullable*/, uint32_t m, const double R[9], const double T[3],
                            uint32_t k, double nvdt, double ndt, int compare_normals, double partials[17],
                            uint32_t *out_tgt, uint8_t *out_keep);

/* device-resident variant for the multi-GPU loop: sample_idx_d [m] (nullable) on the device,
 * partials_d [17] on the device (ready for an RCCL all-reduce on the same stream).  With sample_idx_d NULL the
 * output arrays need n entries (see above), whatever m is. */
int pcr_icp_step_d(pcr_ctx *ctx, const pcr_index *src_index, const pcr_index *tgt_index,
                   const uint32_t *sample_idx_d /*nullable*/, uint32_t sample_begin, uint32_t m,
                   const double R[9], const double T[3], uint32_t k, double thr_fwd_sq, double thr_rev_sq,
                   uint32_t flags, double *partials_d, uint32_t *out_fwd_d, uint32_t *out_rev_d,
                   uint8_t *out_keep_d);
/* name of the kernel the calling thread's last ICP step (single, fit or batch) was launched as -- "icp_step_fast_kernel<4>",
 * "icp_step_grid_kernel", ... -- so that a harness can say what it measured (no reference counterpart) */
const char *pcr_last_step_kernel(void);
int pcr_icp_pivots(const pcr_index *src_index, const pcr_index *tgt_index, const double R[9],
                   const double T[3], double pivot_p[3], double pivot_q[3]);
/* pcr_icp_step_d for callers that CANNOT ALLOCATE: code under HIP graph capture, tight serving loops, harnesses
 * that forbid hipMalloc.  No counterpart in the reference.  Same arguments, same results bit for bit, but the
 * step's scratch (per-iteration constants + per-block partial sums) lives in a fixed device array inside the
 * library (about 2.1 MiB, held by every GPU the library runs on until the process ends) instead of context-owned
 * allocations: the call performs no hipMalloc / hipFree, whatever its arguments.  Limits: m <= PCR_NOALLOC_MAX_SAMPLES
 * (larger m is rejected with PCR_E_INVALID: split the samples into ranges and add the partials); all calls on one
 * GPU share that GPU's array, so two calls that overlap in time on one GPU (different contexts or streams) race --
 * serialise them.  The call is legal under HIP stream capture (kernel and memset nodes only; tested in global capture
 * mode, where any allocation would invalidate the capture).  Inputs and outputs are the caller's, as for pcr_icp_step_d. */
#define PCR_NOALLOC_MAX_SAMPLES 4194304u
int pcr_icp_step_noalloc_d(pcr_ctx *ctx, const pcr_index *src_index, const pcr_index *tgt_index,
                           const uint32_t *sample_idx_d /*nullable*/, uint32_t sample_begin, uint32_t m,
                           const double R[9], const double T[3], uint32_t k, double thr_fwd_sq, double thr_rev_sq,
                           uint32_t flags, double *partials_d, uint32_t *out_fwd_d, uint32_t *out_rev_d,
                           uint8_t *out_keep_d);
/* TEST HOOK of the entry point above (no reference counterpart): leaves the fixed scratch's fold tickets of the
 * context's GPU non-zero, as a launch that died half-way would -- the next pcr_icp_step_noalloc_d must still be right,
 * because it clears them itself at the start of every call (tests/test_zz_no_alloc.py). */
int pcr_noalloc_debug_poison_tickets(pcr_ctx *ctx);

/* ---- the same 17 sums over an EXPLICIT pair list ------------------------------------------
 * Replaces compute3DCentroid x2 + __deMean + __buildPointPairMatrix + H = A*B^T
 * (RegUtilityFunctions.cpp:44-72, TransformationEstimationSVD2.cpp:33-103) and __calculateMSE
 * (DefaultConvergenceCriterial.cpp:46-65) when the pairs already exist on the host -- the public
 * CTransformationEstimationSVD2::computeRotationAndTranslation(A, B, R, T) overload
 * (TransformationEstimationSVD2.cpp:33-49), or the output of a rejection stage.  pair_p, pair_q: AoS
 * double[3P]; pivots as in pcr_icp_step (any point near the data, e.g. the first pair). */
int pcr_reduce_pairs(pcr_ctx *ctx, const double *pair_p, const double *pair_q, uint32_t P, const double pivot_p[3],
                     const double pivot_q[3], double partials[17]);

/* ---- the whole fit loop, device-resident -----------------------------------------------------
 * Replaces CIterativeFit::__fitLoop / _fitStepV (IterativeFit.cpp:119-175) with its stages fused as above AND the
 * 3x3 solve (TransformationEstimationSVD2.cpp:98-117), the composition (BaseTransformationEstimation.h:28-32) and
 * CDefaultConvergenceCriteria::_isConvergedV (DefaultConvergenceCriterial.cpp:69-207) run by a one-thread kernel,
 * so that (R,T) never leaves the device between iterations: a caller enqueues
 *       pcr_fit_enqueue_step;  [all-reduce pcr_fit_partials_d in place on the same stream];  pcr_fit_enqueue_solve
 * as many times as it allows iterations and reads the result once.  After the criteria report convergence (or no
 * pair survives) every kernel still in the queue returns immediately.  The transforms are bit-identical to the
 * host path (pcr_icp_step_d + pcr_solve_rigid): same source compiled for both.
 * conv NULL: never converges by itself (a fixed number of iterations, e.g. for benchmarking).
 * traj_cap: rows of the device-side trajectory log, each {R[9], T[3], mse, pairs} (0: none). */
typedef struct {
    uint32_t max_iterations;           /* BaseConvergenceCriteria.MI   (BaseConvergenceCriteria.cpp:20-27,33)   */
    int failure_after_max_iterations;  /* DefaultConvergenceCriteria.FAMI (DefaultConvergenceCriterial.cpp:86-99) */
    double rotation_threshold;         /* .RT  on |R_prev - R|_F^2            (:103-133)                         */
    double translation_threshold;      /* .TT  on |T|^2, T accumulated        (:106)                             */
    double abs_mse;                    /* .AMSE, scaled by the target's unit square distance (:137-170)          */
    double rel_mse;                    /* .RMSE, likewise                     (:174-207)                         */
    uint32_t max_similar;              /* .MIST                               (:111-123)                         */
} pcr_convergence_params;
typedef enum {                         /* EConvergenceState (DefaultConvergenceCriterial.h:32-40) */
    PCR_FIT_NOT_CONVERGED = 0,
    PCR_FIT_ITERATIONS = 1,
    PCR_FIT_TRANSFORM = 2,
    PCR_FIT_ABS_MSE = 3,
    PCR_FIT_REL_MSE = 4,
    PCR_FIT_NO_CORRESPONDENCES = 5,
    PCR_FIT_DEGENERATE = -2            /* the accumulated R stopped being a rotation (no reference counterpart) */
} pcr_fit_state;
typedef struct pcr_fit pcr_fit;
int pcr_fit_create(pcr_ctx *ctx, const pcr_index *src_index, const pcr_index *tgt_index,
                   const uint32_t *sample_idx_d /*nullable*/, uint32_t sample_begin, uint32_t m, uint32_t k,
                   double thr_fwd_sq, double thr_rev_sq, const pcr_convergence_params *conv /*nullable*/,
                   double unique_sq_dist, uint32_t traj_cap, pcr_fit **out);
/* replace the sample set by a HOST list of source indices (copied; the fit owns the device copy).  An empty list
 * (m = 0) yields zero sums.  Synchronises the stream. */
int pcr_fit_set_samples(pcr_fit *fit, const uint32_t *sample_idx, uint32_t m);
/* DEBUG / parity runs only: let every iteration of the device-resident loop also write what the reference's loop hands
 * from stage to stage each iteration (the pair set of IterativeFit.cpp:138-158, CorrespondenceEstimationClosestPoints.cpp:
 * 84-106) -- per sample the forward index, the reverse index and the keep flag, with pcr_icp_step_d's slot convention
 * (sample list: slot = position in the list, m entries; sample range: slot = source index, n entries).  Device
 * pointers, each nullable; all NULL switches the outputs off again (the default: a benchmark leaves them off).  Each
 * iteration overwrites the previous one's: enqueue one iteration, copy, enqueue the next.  Takes effect from the next
 * enqueued iteration (also inside a pcr_fit_batch of which the fit is a member); drops a captured graph. */
int pcr_fit_set_debug_outputs(pcr_fit *fit, uint32_t *out_fwd_d, uint32_t *out_rev_d, uint8_t *out_keep_d);
/* (re)start from the initial transform: iteration 0, criteria state as freshly constructed */
int pcr_fit_reset(pcr_fit *fit, const double R[9], const double T[3]);
int pcr_fit_enqueue_step(pcr_fit *fit);
double *pcr_fit_partials_d(pcr_fit *fit);
int pcr_fit_enqueue_solve(pcr_fit *fit);
/* pcr_fit_enqueue_step + pcr_fit_enqueue_solve for a fit that needs no exchange in between (one GPU): where the step
 * runs as the filtered kernel, its last block folds the partials and runs the solve -- ONE launch per iteration. */
int pcr_fit_enqueue_iteration(pcr_fit *fit);
/* `count` of those as ONE HIP graph launch: captured on the first call with that count (no kernel argument depends on
 * the iteration), replayed afterwards; falls back to `count` plain launches while kernel timing is on.  Surplus
 * iterations after convergence are no-ops, as always. */
int pcr_fit_enqueue_iterations(pcr_fit *fit, uint32_t count);
/* ---- many small fits, ONE launch per iteration ---------------------------------------------------
 * The caller of the loop (CRegistartion::fineFit per scan pair, Registration.cpp:148-158; unit test/FineRegistration/
 * TestFineRegistration.cpp:26-92) registers a scan set pair by pair: independent problems, each too small to fill
 * the GPU (BASELINE config C4).  A batch ties existing fits of one context together: every pcr_fit_batch_enqueue_iteration
 * advances ALL of them by one iteration in a single kernel launch (each fit folds and solves for itself in the last
 * of its own blocks; a fit whose criteria have fired costs nothing in later launches).  The fits keep their own state:
 * pcr_fit_reset before, pcr_fit_read after, exactly as for a fit run alone -- and with bit-identical results.
 * Members must be on single-level grid indices with K <= 8, every member's K in the same class (1, 2..4 or 5..8); a member's sample set must not
 * change while the batch exists.  The batch is destroyed before its members. */
typedef struct pcr_fit_batch pcr_fit_batch;
int pcr_fit_batch_create(pcr_ctx *ctx, pcr_fit *const *fits, uint32_t count, pcr_fit_batch **out);
int pcr_fit_batch_enqueue_iteration(pcr_fit_batch *batch);
int pcr_fit_batch_destroy(pcr_fit_batch *batch);
/* synchronises the stream.  All outputs nullable; traj receives min(iterations, traj_cap, traj_rows) rows. */
int pcr_fit_read(pcr_fit *fit, double R[9], double T[3], uint32_t *iterations, int *state, double *mse,
                 double *pairs, double *traj, uint32_t traj_rows);
int pcr_fit_destroy(pcr_fit *fit);

/* ---- the exchange step of the multi-GPU loop ---------------------------------------------------
 * No reference counterpart (the reference is a single-threaded CPU program).  One process per GPU, samples
 * sharded, clouds and indices replicated; the only exchange per iteration is the sum of the 17 partials.
 * pcr_allreduce_partials enqueues an in-place RCCL all-reduce of 17 doubles on the context's stream (between
 * pcr_fit_enqueue_step and pcr_fit_enqueue_solve, or after pcr_icp_step_d).  RCCL is bound at run time; without
 * it these return PCR_E_UNSUPPORTED.  Rank 0 creates the id and distributes it out of band (a file, MPI, a socket);
 * pcr_comm_create is collective.  A caller that already owns a communicator may all-reduce the buffer itself. */
#define PCR_COMM_ID_BYTES 128
typedef struct pcr_comm pcr_comm;
int pcr_comm_unique_id(char id[PCR_COMM_ID_BYTES]);
int pcr_comm_create(pcr_ctx *ctx, int rank, int world, const char id[PCR_COMM_ID_BYTES], pcr_comm **out);
int pcr_allreduce_partials(pcr_comm *comm, double *partials_d);
int pcr_comm_destroy(pcr_comm *comm);

/* ---- host: CTransformationEstimationSVD2::__computeRelativeRotationAndTranslation
 *      (TransformationEstimationSVD2.cpp:98-117) + accumulateTransformation
 *      (BaseTransformationEstimation.h:28-32), fed by the 17 partials instead of pair lists.
 * R,T in/out = accumulated transform; Rc,Tc (nullable) = this iteration's increment;
 * mse (nullable) = partials[16]/n.  PCR_E_EMPTY if n == 0. */
int pcr_solve_rigid(const double partials[17], const double pivot_p[3], const double pivot_q[3],
                    double R[9], double T[3], double Rc[9], double Tc[3], double *mse);

/* ---- CRejectionCentroidDist::_rejectInvalidPairsV (RejectionCentroidDist.cpp:23-153) ----
 * pair_p = TRANSFORMED source points, queried against the UNTRANSFORMED source index
 * (reference quirk, :52,:75); pair_q = target points.  out_keep[P]; out_d[P] (nullable) the
 * per-pair statistic; out_stat[2] (nullable) = {mean, sigma}.  msn = MSN (25), msd = MSD (8.0,
 * compared with a squared distance as the reference does), cdf = CDF. */
int pcr_reject_centroid_dist(pcr_ctx *ctx, const pcr_index *src_index, const pcr_index *tgt_index,
                             const double *pair_p, const double *pair_q, uint32_t P, uint32_t msn, double msd,
                             double cdf, uint8_t *out_keep, double *out_d, double *out_stat);


/* The same with the PAIRS sharded over ranks ("a second tiny all-reduce of (sum d, P) then
 * sum (d - mean)^2"; RejectionCentroidDist.cpp:34-37 computes both over all pairs).  Each rank passes its own P
 * pairs (P = 0 allowed) and a hook that sums a 17-double DEVICE buffer over all ranks, in place, ordered on the
 * context's stream -- called exactly twice on every rank.  The buffer has the shape of the fit loop's partials, so
 * pcr_allreduce_partials serves (pcr_reduce17_with_comm below, user = the pcr_comm), and so does any communicator
 * the caller owns.  The hook returns PCR_OK or a negative status.  out_stat = {mean, sigma} over ALL ranks' pairs. */
typedef int (*pcr_reduce17_fn)(void *user, double *buf17_d);
int pcr_reduce17_with_comm(void *comm /* pcr_comm* */, double *buf17_d);
int pcr_reject_centroid_dist_sharded(pcr_ctx *ctx, const pcr_index *src_index, const pcr_index *tgt_index,
                                     const double *pair_p, const double *pair_q, uint32_t P, uint32_t msn, double msd,
                                     double cdf, uint8_t *out_keep, double *out_d, double *out_stat,
                                     pcr_reduce17_fn reduce, void *reduce_user);

#ifdef __cplusplus
}
#endif
#endif /* PCR_H */
