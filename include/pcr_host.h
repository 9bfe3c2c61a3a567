/*
 * pcr_host.h -- C entry points of libpcr_host.so, the C++ plugin layer that mirrors the reference's
 * registration interfaces (hiveSearch::CBaseSearch, CBaseCorrespondenceEstimation,
 * CBaseCorrespondenceRejection, CBaseTransformationEstimation, CDefaultConvergenceCriteria,
 * CIterativeFit) on top of the device library declared in pcr.h.
 *
 * C++ callers use the classes directly (the headers under pointcloudregistration_amd/host/); these functions exist for
 * the Python test-suite and other non-C++ callers.  Every one drives the plugin classes through the
 * factory and the config map, the way the reference's callers do.  All pointers are host pointers;
 * points are AoS double[3n]; R is 3x3 column-major, T is 3 doubles.  Return 0 on success.
 */
#ifndef PCR_HOST_H
#define PCR_HOST_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* drop every factory product, every device cloud/index and the device context (fresh singletons next call;
 * the reference keeps plugin and criteria state for the life of the process) */
void pcrh_reset(void);
void pcrh_set_device(int device);
void pcrh_set_verbose(int on);
/* pcr_index_params.compact_tables (include/pcr.h) for every index the plugin layer builds from now on: 0 (default)
 * dense cell tables, 1 compact tables for the sparse levels of nested grids, 2 for every level of a nested grid.
 * No counterpart in the reference (its KD-tree has no such choice); results do not depend on it. */
void pcrh_set_compact_tables(uint32_t mode);

/* sample sharding over ranks: this process runs samples [begin,begin+count) of every step and
 * `fn` sums the 17 partials over the ranks in place (return 0 on success). world <= 1 disables it. */
typedef int (*pcrh_allreduce_fn)(double *partials17, void *user);
void pcrh_set_sharding(uint32_t rank, uint32_t world, pcrh_allreduce_fn fn, void *user);

/* CControlParameters (ControlParameters.h:15-86): `Key = Value` lines, `/` starts a comment, keys are
 * case-insensitive.  pcrh_config_get: -1 if the key is absent. */
int pcrh_config_parse_file(const char *path);
void pcrh_config_parse_text(const char *text);
void pcrh_config_set(const char *key, const char *value);
int pcrh_config_get(const char *key, char *buf, uint32_t cap);

/* 1 if a plugin with this class signature is registered with the factory
 * (hiveCommon::CProductFactory<T> TheCreator(T::getClassSig()), e.g. TransformationEstimationSVD2.cpp:10) */
int pcrh_is_registered(const char *sig);

/* CIterativeFit::fit (IterativeFit.cpp:74-93) with plugins chosen by signature (NULL = the GPU default of
 * each stage; rejection default is the pass-through CorrespondenceRejectionClosestPoints).  sample_idx NULL
 * = every source point.  R,T in: initial transform, out: result.  traj (nullable): per iteration 14 doubles
 * {R[9], T[3], mse, pairs}, at most traj_cap rows.  msg receives the convergence message
 * (BaseConvergenceCriteria.cpp:42-52).  -2: a signature is not registered. */
int pcrh_fit(const double *src, const double *src_normals, uint32_t ns, const double *tgt, const double *tgt_normals, uint32_t nt,
             const uint32_t *sample_idx, uint32_t m, const char *sampler_sig, const char *estimation_sig, const char *rejection_sig,
             const char *transformation_sig, double R[9], double T[3], double *traj, uint32_t traj_cap, uint32_t *iterations,
             char *msg, uint32_t msg_cap);

/* determineCorrespondences + rejectInvalidPairs for one (R,T), pair lists copied back to the host
 * (CorrespondenceEstimationClosestPoints.cpp:54-117): out_pair_p = transformed source points, out_pair_q =
 * target points, in sample order.  -3 if more than `cap` pairs survive (out_pairs still holds the count). */
int pcrh_correspondences(const double *src, uint32_t ns, const double *tgt, uint32_t nt, const uint32_t *sample_idx, uint32_t m,
                         const double R[9], const double T[3], const char *rejection_sig, double *out_pair_p, double *out_pair_q,
                         uint32_t cap, uint32_t *out_pairs);

/* CTransformationEstimationSVD2::computeRotationAndTranslation(A, B, R, T) (TransformationEstimationSVD2.cpp:33-49) */
int pcrh_svd2(const double *pair_a, const double *pair_b, uint32_t P, double R[9], double T[3]);

/* hiveSearch::CBaseSearch::setInputData + executeSearch on CGridSearchGPU (BaseSearch.cpp:27-60).  Row-major
 * [nq x k]; farthest first unless nearest_first.  -1 when executeSearch returns NULL (dim != 3, k > n, no GPU). */
int pcrh_search(const double *points, uint32_t n, uint32_t dim, const double *queries, uint32_t nq, uint32_t k, int nearest_first,
                int multi_threaded, uint32_t *out_idx);

/* ASCII PLY as the reference reads and writes it (PLYLoad.cpp:76-111, unit test/TestUnitity.cpp:278-339): `x y z`
 * (+ `nx ny nz` with_normals) per vertex after `element vertex N` ... `end_header`; faces ignored.
 * pcrh_ply_load: -1 unreadable, -3 more than `cap` vertices (out_n still holds the count: call again). */
int pcrh_ply_load(const char *path, int with_normals, uint32_t cap, double *out_xyz, double *out_normals, uint32_t *out_n);
int pcrh_ply_save(const char *path, const double *xyz, const double *normals /*nullable*/, uint32_t n);

/* CRegistartion::fineFit (Registration.cpp:144-158) over a scan sequence: scan i+1 is registered to scan i from
 * init_RT[i] = {R[9], T[3]} (NULL: identity).  out_relative_RT[i] maps scan i+1 into scan i's frame,
 * out_absolute_RT[i] into scan 0's; out_iterations[i], msgs + i*msg_stride: per pair.  num_scans >= 2. */
int pcrh_register_sequence(const double *const *scans, const uint32_t *counts, uint32_t num_scans, const double *init_RT,
                           double *out_relative_RT, double *out_absolute_RT, uint32_t *out_iterations, char *msgs, uint32_t msg_stride);

/* CDefaultConvergenceCriteria on explicit pair lists (host only; DefaultConvergenceCriterial.cpp:69-207).
 * pcrh_conv_reset clears the criteria's carried state and sets the target's unit square distance. */
void pcrh_conv_reset(double target_unique_sq_dist);
int pcrh_conv_step(const double R[9], const double T[3], uint32_t iteration, const double *pair_p, const double *pair_q, uint32_t P,
                   int *state, double *mse, char *msg, uint32_t msg_cap);

#ifdef __cplusplus
}
#endif
#endif /* PCR_HOST_H */
