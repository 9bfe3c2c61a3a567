// ORACLE (test infrastructure only -- see pcr_oracle.h).
//
// CPU restatement of CRejectionCentroidDist, written from the text of
//   /root/reference/PointCloudRegistration/RejectionCentroidDist.cpp:23-208
//   /root/reference/PointCloudRegistration/RegMath.h:11-22,42-55,64-67
// The class is orphan code in the reference (no factory registration); its
// quirks are reproduced on purpose:
//   * the "centroid" divides component 0 three times and never components 1,2 (:113-115);
//   * the source-side query is the TRANSFORMED source point against the UNTRANSFORMED
//     source tree (:52,:75);
//   * MSD is compared with a SQUARED distance (:200-208), lower_bound on the ascending list.
#include <cmath>
#include <vector>
#include "pcr_oracle.h"

extern "C" void orc_kd_old_query_one(const void *tree, const double *q, uint32_t *oi, double *od);

namespace {

// __comSquaredDist2Centroid (:81-92) = __searchNeighbors + __computeCentroid + distance
double sq_dist_to_centroid(const double *x, const double *cloud, const void *tree, uint32_t msn, double msd) {
    std::vector<uint32_t> nb(msn);
    orc_kd_old_query_one(tree, x, nb.data(), nullptr);  // __searchDesignedNumNeighbors (:180-189)
    // __searchDesignedRadiusNeighbors (:165-176): std::lower_bound(nb, MSD, DistCmp) on the list that
    // is ascending in distance => first position whose squared distance is NOT < MSD; erase from there.
    // DistCmp (:200-208) computes (n-x)^2 summed x,y,z.
    size_t lo = 0, hi = nb.size();
    while (lo < hi) {  // same bisection as std::lower_bound
        size_t mid = lo + (hi - lo) / 2;
        const double *p = cloud + (size_t)nb[mid] * 3;
        double d = (p[0] - x[0]) * (p[0] - x[0]) + (p[1] - x[1]) * (p[1] - x[1]) + (p[2] - x[2]) * (p[2] - x[2]);
        if (d < msd)
            lo = mid + 1;
        else
            hi = mid;
    }
    const size_t cnt = lo;
    // __computeCentroid (:96-116)
    double c[3] = {x[0], x[1], x[2]};
    for (size_t i = 0; i < cnt; ++i) {
        const double *p = cloud + (size_t)nb[i] * 3;
        c[0] += p[0];
        c[1] += p[1];
        c[2] += p[2];
    }
    c[0] /= (double)(cnt + 1);
    c[0] /= (double)(cnt + 1);
    c[0] /= (double)(cnt + 1);
    return (x[0] - c[0]) * (x[0] - c[0]) + (x[1] - c[1]) * (x[1] - c[1]) + (x[2] - c[2]) * (x[2] - c[2]);
}

}  // namespace

extern "C" uint32_t orc_reject_centroid_dist(const double *src, uint32_t ns, const double *tgt, uint32_t nt,
                                             const double *pair_p, const double *pair_q, uint32_t P, uint32_t msn,
                                             double msd, double cdf, double *d_out, uint8_t *keep, double *stat_out) {
    if (P == 0 || msn == 0 || msn > ns || msn > nt) return 0;
    void *stree = orc_kd_old_build(src, ns, msn);  // __initKNN (:42-60)
    void *ttree = orc_kd_old_build(tgt, nt, msn);
    std::vector<double> d(P);
    for (uint32_t i = 0; i < P; ++i) {  // __comDiffSquaredDist2CentroidVec (:64-79)
        const double t = sq_dist_to_centroid(pair_q + (size_t)i * 3, tgt, ttree, msn, msd);
        const double s = sq_dist_to_centroid(pair_p + (size_t)i * 3, src, stree, msn, msd);
        d[i] = std::abs(t - s);
    }
    orc_kd_old_free(stree);
    orc_kd_old_free(ttree);
    double sum = 0;  // computeMeanValue (RegMath.h:11-22)
    for (uint32_t i = 0; i < P; ++i) sum += d[i];
    const double mean = sum / P;
    double var = 0;  // computeVariance(..., mean, POPULATION) (RegMath.h:42-55)
    for (uint32_t i = 0; i < P; ++i) var += (d[i] - mean) * (d[i] - mean);
    var /= P;
    const double sigma = std::sqrt(var);
    const double tol = cdf * sigma;  // __reject (:134-142)
    uint32_t kept = 0;
    for (uint32_t i = 0; i < P; ++i) {
        const uint8_t k = std::abs(d[i] - mean) <= tol;
        if (keep) keep[i] = k;
        if (d_out) d_out[i] = d[i];
        kept += k;
    }
    if (stat_out) {
        stat_out[0] = mean;
        stat_out[1] = sigma;
    }
    return kept;
}
