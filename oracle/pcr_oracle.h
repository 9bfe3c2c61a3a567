/*
 * pcr_oracle.h -- C API of the CPU ORACLE.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-CPU restatement of the reference's
 * ICP fine-registration hot path (fufesou/PointCloudRegistration), written from
 * the text of the reference sources.  It is the checker for the HIP product
 * path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load it; nothing under pointcloudregistration_amd/ links or imports it.
 *
 * PARITY UNPINNED.  The reference holds no golden vectors, known-answer tests or
 * fixtures for this path, and none of its sources can be
 * built in this image: the one self-contained file, Common/KNNSearch.cpp, needs
 * the un-vendored Hive framework's CommonExport.h, everything above the tree
 * needs Eigen/Boost/Hive.  The restatement follows the source text only; the
 * vectors under tests/golden/ are its own frozen outputs (make_golden.py).
 *
 * All arithmetic is IEEE fp64 compiled with -ffp-contract=off.  Points are AoS
 * double[3n]; R is 3x3 column-major (Eigen::Matrix3d::data()), T is 3.
 */
#ifndef PCR_ORACLE_H
#define PCR_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NO_INDEX 0xFFFFFFFFu

/* ---- hiveCommon::CKNNSearch (Common/KNNSearch.{h,cpp}) -- the tree ICP uses ---- */
/* initKNNSearch, Common/KNNSearch.cpp:17.  The oracle copies the points. */
void *orc_kd_old_build(const double *xyz, uint32_t n, uint32_t k);
/* executeKNN, Common/KNNSearch.cpp:65.  out_idx [nq*k] ascending distance, first
 * visited wins on ties (KNNSearch.h:75).  out_d2 may be NULL.  threads>1 runs an
 * OpenMP loop over queries (mirrors NewKNNSearch.cpp:39), results identical. */
void orc_kd_old_query(const void *tree, const double *q, uint32_t nq, uint32_t *out_idx,
                      double *out_d2, int threads);
void orc_kd_old_free(void *tree);

/* ---- hiveSearch::CKNNSearch behind CBaseSearch (NewKNNSearch.cpp, DistanceQueue.cpp) ---- */
void *orc_kd_new_build(const double *data, uint32_t n, uint32_t dim, int leaf_max);
/* _executeSearchV, NewKNNSearch.cpp:32: out [nq*k], per query farthest first
 * (DistanceQueue.cpp:37-46). */
void orc_kd_new_query(const void *tree, uint32_t k, const double *q, uint32_t nq, uint32_t *out_idx);
void orc_kd_new_free(void *tree);

/* ---- CPointCloud::transform, PointCloud.cpp:358-371 (positions only) ---- */
void orc_transform(double *xyz, uint32_t n, const double R[9], const double T[3]);

/* ---- CUniqueData::setPointCloud, UniqueData.cpp:25-53 ---- */
double orc_uniq_sq_dist(const double *xyz, uint32_t n);

/* ---- CCorrespondenceEstimationClosestPoints::_determineCorrespondencesV,
 *      CorrespondenceEstimationClosestPoints.cpp:54-117.
 * tgt_tree: orc_kd_old_build(tgt, nt, k).  sample_idx NULL = every source point.
 * Outputs (all nullable): fwd[m], rev[m], keep[m]; pair_p / pair_q [3*P] in sample
 * order.  Returns P (number of kept pairs).  threads>1: OpenMP over samples
 * (the "generous" CPU baseline); results identical to threads==1. */
uint32_t orc_icrp_correspondences(const double *src, uint32_t ns, const double *tgt, uint32_t nt,
                                  const void *tgt_tree, const uint32_t *sample_idx, uint32_t m,
                                  const double R[9], const double T[3], uint32_t k,
                                  double thr_fwd_sq, double thr_rev_sq, uint32_t *fwd,
                                  uint32_t *rev, uint8_t *keep, double *pair_p, double *pair_q,
                                  int threads);

/* CSamplerRandom::_doSampleV (Sampler/SamplerRandom.cpp:25-51) on a caller-supplied stream of rand() values. */
uint32_t orc_sampler_random(const int32_t *draws, uint32_t n, uint32_t rand_max, double ratio, uint32_t *out_idx);

/* f4, opt-in: __insertFitPairs (CorrespondenceEstimationClosestPoints.cpp:128-175) as the pair decision, with
 * TmpTgtPoint := the neighbour's position (the reference reads it unassigned: undefined behaviour there).
 * out_tgt[m]: the chosen target index or ORC_NO_INDEX.  Returns the number of pairs. */
uint32_t orc_icrp_normal_fit(const double *src, const double *src_normals, uint32_t ns, const double *tgt,
                             const double *tgt_normals, uint32_t nt, const void *tgt_tree,
                             const uint32_t *sample_idx, uint32_t m, const double R[9], const double T[3],
                             uint32_t k, double nvdt, double ndt, int compare_normals, uint32_t *out_tgt,
                             uint8_t *keep, double *pair_p, double *pair_q, int threads);

/* ---- CTransformationEstimationSVD2, TransformationEstimationSVD2.cpp:17-117 +
 *      compute3DCentroid (RegUtilityFunctions.cpp:44-72, single slot = Release build) +
 *      accumulateTransformation (BaseTransformationEstimation.h:28-32).
 * R,T in/out (accumulated); Rc,Tc out (increment; nullable).  Returns 0, or -1 if P==0. */
int orc_svd2(const double *pair_p, const double *pair_q, uint32_t P, double R[9], double T[3],
             double Rc[9], double Tc[3]);

/* 3x3 SVD used by orc_svd2 (two-sided Jacobi as published for Eigen::JacobiSVD):
 * H (col-major) = U diag(S) V^T, S descending, all col-major. */
void orc_svd3(const double H[9], double U[9], double S[3], double V[9]);

/* ---- CDefaultConvergenceCriteria, DefaultConvergenceCriterial.cpp:21-207 ---- */
typedef struct {
    uint32_t max_iterations;       /* BaseConvergenceCriteria.MI   (default 1000, RegCfg.txt 30) */
    int failure_after_max_iter;    /* FAMI default false */
    double rotation_threshold;     /* RT  default 1e-4, RegCfg.txt 1e-6 */
    double translation_threshold;  /* TT  default 3e-8, RegCfg.txt 1e-7 */
    double rel_mse;                /* RMSE default 0.01 */
    double abs_mse;                /* AMSE default 0.01 */
    uint32_t max_similar;          /* MIST default 0 */
} orc_conv_params;
void *orc_conv_create(const orc_conv_params *p);
/* isConverged(R,T,iter) with the pair set of this iteration (setCorrespondencePairSet).
 * uniq = target CUniqueData value.  state_out: 0 NotConverged, 1 Iterations, 2 Transform,
 * 3 AbsMSE, 4 RelMSE.  mse_out: m_CurMSE after the call. */
int orc_conv_step(void *conv, const double R[9], const double T[3], uint32_t iter,
                  const double *pair_p, const double *pair_q, uint32_t P, double uniq,
                  int *state_out, double *mse_out);
void orc_conv_free(void *conv);

/* ---- CIterativeFit::fit, IterativeFit.cpp:74-175, closest-point plugin quartet ----
 * traj (nullable) receives per iteration 14 doubles: R(9) T(3) MSE P.  Returns the
 * number of iterations run; R,T in/out; state_out as orc_conv_step. */
typedef struct {
    uint32_t k;        /* CorrespondenceEstimationClosestPoints.NN */
    double sdft, ndft; /* .SDFT, .NDFT */
    orc_conv_params conv;
} orc_fit_params;
uint32_t orc_fit(const double *src, uint32_t ns, const double *tgt, uint32_t nt,
                 const uint32_t *sample_idx, uint32_t m, const orc_fit_params *prm, double R[9],
                 double T[3], int *state_out, double *traj, uint32_t traj_cap, int threads);

/* ---- CRejectionCentroidDist, RejectionCentroidDist.cpp:23-208 (quirks kept) ----
 * pair_p are TRANSFORMED source points, queried against the UNTRANSFORMED source tree (:52,:75).
 * d_out[P] (nullable) = per-pair statistic; keep[P]; stat_out[2] = {mean, sigma}. Returns P'. */
uint32_t orc_reject_centroid_dist(const double *src, uint32_t ns, const double *tgt, uint32_t nt,
                                  const double *pair_p, const double *pair_q, uint32_t P,
                                  uint32_t msn, double msd, double cdf, double *d_out,
                                  uint8_t *keep, double *stat_out);

/* brute-force exact K-NN, ascending (d2, idx) -- the definitional check for tie-free inputs. */
void orc_knn_brute(const double *xyz, uint32_t n, const double *q, uint32_t nq, uint32_t k,
                   uint32_t *out_idx, double *out_d2);

#ifdef __cplusplus
}
#endif
#endif
