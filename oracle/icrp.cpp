// ORACLE (test infrastructure only -- see pcr_oracle.h).
//
// CPU restatement of one ICRP iteration and of the loop around it, written from the
// text of (all under /root/reference/PointCloudRegistration/):
//   FineRegistrationClosestPoints/CorrespondenceEstimationClosestPoints.cpp:54-117
//   PointCloud.cpp:358-371                      (transform)
//   UniqueData.cpp:25-53, RegUtilityFunctions.h:57-74
//   TransformationEstimationSVD2.cpp:17-117, RegUtilityFunctions.cpp:44-72
//   BaseTransformationEstimation.h:28-32
//   DefaultConvergenceCriterial.cpp:21-207, BaseConvergenceCriteria.cpp:20-38
//   IterativeFit.cpp:74-175
// Eigen (JacobiSVD, squaredNorm, Matrix3d*Vector3d) is a third-party dependency that is
// absent here and unpinned in the reference ($(EIGEN), PointCloudRegistration.vcproj:44):
// its arithmetic is restated as plain left-to-right fp64, the SVD as the published
// two-sided Jacobi method.  PARITY UNPINNED for these parts (no golden numbers exist in
// the reference); the polar factor R = V diag(1,1,+-1) U^T does not depend
// on the SVD algorithm for non-degenerate H.
#include <cfloat>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "pcr_oracle.h"

extern "C" {
const double *orc_kd_old_points(const void *tree);
uint32_t orc_kd_old_k(const void *tree);
void orc_kd_old_query_one(const void *tree, const double *q, uint32_t *oi, double *od);
}

namespace {

// PointCloud.cpp:368-370 : ((r0*x + r3*y) + r6*z) + t0, column-major R, no FMA.
inline void xform(const double R[9], const double T[3], const double s[3], double o[3]) {
    const double x = s[0], y = s[1], z = s[2];
    o[0] = R[0] * x + R[3] * y + R[6] * z + T[0];
    o[1] = R[1] * x + R[4] * y + R[7] * z + T[1];
    o[2] = R[2] * x + R[5] * y + R[8] * z + T[2];
}

// (a-b).squaredNorm(), CorrespondenceEstimationClosestPoints.cpp:97-98
inline double sqdist(const double a[3], const double b[3]) {
    double d = 0, t;
    t = a[0] - b[0];
    d += t * t;
    t = a[1] - b[1];
    d += t * t;
    t = a[2] - b[2];
    d += t * t;
    return d;
}

// column-major 3x3 helpers
inline double &M(double *A, int r, int c) { return A[c * 3 + r]; }
inline double M(const double *A, int r, int c) { return A[c * 3 + r]; }

void matmul3(const double *A, const double *B, double *C) {  // C = A*B
    double t[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) t[c * 3 + r] = M(A, r, 0) * M(B, 0, c) + M(A, r, 1) * M(B, 1, c) + M(A, r, 2) * M(B, 2, c);
    std::memcpy(C, t, sizeof t);
}

double det3(const double *A) {
    return M(A, 0, 0) * (M(A, 1, 1) * M(A, 2, 2) - M(A, 1, 2) * M(A, 2, 1)) -
           M(A, 0, 1) * (M(A, 1, 0) * M(A, 2, 2) - M(A, 1, 2) * M(A, 2, 0)) +
           M(A, 0, 2) * (M(A, 1, 0) * M(A, 2, 1) - M(A, 1, 1) * M(A, 2, 0));
}

// rows p,q of A <- G^T-style plane rotation: x' = c x + s y ; y' = -s x + c y
void rot_rows(double *A, int p, int q, double c, double s) {
    for (int j = 0; j < 3; ++j) {
        double x = M(A, p, j), y = M(A, q, j);
        M(A, p, j) = c * x + s * y;
        M(A, q, j) = -s * x + c * y;
    }
}
void rot_cols(double *A, int p, int q, double c, double s) {  // x' = c x - s y ; y' = s x + c y
    for (int i = 0; i < 3; ++i) {
        double x = M(A, i, p), y = M(A, i, q);
        M(A, i, p) = c * x - s * y;
        M(A, i, q) = s * x + c * y;
    }
}

}  // namespace

extern "C" {

void orc_transform(double *xyz, uint32_t n, const double R[9], const double T[3]) {
    for (uint32_t i = 0; i < n; ++i) {
        double o[3];
        xform(R, T, xyz + (size_t)i * 3, o);
        xyz[(size_t)i * 3 + 0] = o[0];
        xyz[(size_t)i * 3 + 1] = o[1];
        xyz[(size_t)i * 3 + 2] = o[2];
    }
}

double orc_uniq_sq_dist(const double *xyz, uint32_t n) {
    // UniqueData.cpp:29-52: ~20 evenly strided points, K=4 (self included), mean of the 4 d2.
    uint32_t stride = n / 20;
    if (stride < 1) stride = 1;
    void *tree = orc_kd_old_build(xyz, n, 4);
    double sum = 0;
    uint32_t count = 0;
    for (uint32_t i = 0; i < n; i += stride) {
        uint32_t nb[4];
        orc_kd_old_query_one(tree, xyz + (size_t)i * 3, nb, nullptr);
        for (int j = 0; j < 4; ++j) {
            // comSquaredDist(pPos, SampleItr, pPos, nb): (a-b)*(a-b) summed x,y,z from 0.0
            const double *a = xyz + (size_t)i * 3, *b = xyz + (size_t)nb[j] * 3;
            double d = 0.0;
            for (int c = 0; c < 3; ++c) d += ((a[c] - b[c]) * (a[c] - b[c]));
            sum += d;
        }
        ++count;
    }
    orc_kd_old_free(tree);
    return sum / (count * 4);
}

uint32_t orc_icrp_correspondences(const double *src, uint32_t ns, const double *tgt, uint32_t nt, const void *tgt_tree,
                                  const uint32_t *sample_idx, uint32_t m, const double R[9], const double T[3],
                                  uint32_t k, double thr_fwd_sq, double thr_rev_sq, uint32_t *fwd, uint32_t *rev,
                                  uint8_t *keep, double *pair_p, double *pair_q, int threads) {
    (void)nt;
    if (orc_kd_old_k(tgt_tree) != k) return 0;
    // :73-78  whole source copied, transformed, fresh K=1 tree on it -- EVERY iteration
    std::vector<double> tsrc(src, src + (size_t)ns * 3);
    orc_transform(tsrc.data(), ns, R, T);
    void *rtree = orc_kd_old_build(tsrc.data(), ns, 1);

    std::vector<uint32_t> fw(m), rv(m);
    std::vector<uint8_t> kp(m);
    const int nth = threads > 1 ? threads : 1;
#pragma omp parallel for schedule(dynamic, 2048) num_threads(nth)
    for (long i = 0; i < (long)m; ++i) {
        const uint32_t si = sample_idx ? sample_idx[i] : (uint32_t)i;
        double p[3];
        xform(R, T, src + (size_t)si * 3, p);  // :86  CurSrcPoint = R*s + T
        uint32_t nb[64];
        std::vector<uint32_t> nbh;
        uint32_t *pn = nb;
        if (k > 64) {
            nbh.resize(k);
            pn = nbh.data();
        }
        orc_kd_old_query_one(tgt_tree, p, pn, nullptr);  // :89
        const uint32_t j = pn[k - 1];                      // NeighborSet.back() = K-th nearest (:91-92)
        const double *q = tgt + (size_t)j * 3;
        uint32_t r1;
        orc_kd_old_query_one(rtree, q, &r1, nullptr);  // :94
        const double *pp = &tsrc[(size_t)r1 * 3];      // :95
        fw[i] = j;
        rv[i] = r1;
        kp[i] = (sqdist(p, q) < thr_fwd_sq) && (sqdist(p, pp) < thr_rev_sq);  // :97-98 strict <
    }
    orc_kd_old_free(rtree);

    uint32_t P = 0;
    for (uint32_t i = 0; i < m; ++i) {
        if (fwd) fwd[i] = fw[i];
        if (rev) rev[i] = rv[i];
        if (keep) keep[i] = kp[i];
        if (kp[i]) {
            if (pair_p) {
                const uint32_t si = sample_idx ? sample_idx[i] : i;
                xform(R, T, src + (size_t)si * 3, pair_p + (size_t)P * 3);
            }
            if (pair_q) std::memcpy(pair_q + (size_t)P * 3, tgt + (size_t)fw[i] * 3, 24);
            ++P;
        }
    }
    return P;
}

// CSamplerRandom::_doSampleV (Sampler/SamplerRandom.cpp:25-51) on an explicit stream of rand() values: the loop
// `for i: if (ICP_RAND(0.0, 1.0) < SampleRatio) push_back(i)` with ICP_RAND(min, max) = (min) + (double)rand()/(RAND_MAX+1)
// *((max)-(min)) (ICPMacros.h:4).  The reference never seeds rand(); whoever supplies the stream decides it.
uint32_t orc_sampler_random(const int32_t *draws, uint32_t n, uint32_t rand_max, double ratio, uint32_t *out_idx) {
    uint32_t c = 0;
    for (uint32_t i = 0; i < n; ++i)
        if ((0.0) + (double)draws[i] / ((double)rand_max + 1.0) * ((1.0) - (0.0)) < ratio) out_idx[c++] = i;
    return c;
}

// Eigen's Vector3d::normalize() of the reference's vintage: *this /= sqrt(squaredNorm()), no zero test (a zero
// vector becomes NaN); squaredNorm and dot sum left to right.
static inline void normalize3(double v[3]) {
    const double n = std::sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
    v[0] /= n;
    v[1] /= n;
    v[2] /= n;
}

// ---- f4: CCorrespondenceEstimationClosestPoints::__insertFitPairs (:128-175) as the pair decision of the loop at
// :84-106, in place of the reciprocal gate (:94-104) -- its call is commented out at :105.  For every sample the K
// forward neighbours are tried nearest first (:140); the first one whose direction from the sample is (anti)parallel to
// its own normal to within NVDT (:145-155) -- and, if normals are compared, whose normal agrees with the sample's
// rotated normal to within NDT (:157-164; the reference hard-wires NoNormalCompare = 1, :133) -- makes the pair (:166-171).
// ONE unavoidable departure, and the reason this variant is opt-in and "parity unpinned" by construction: the reference
// reads TmpTgtPoint (:146, :166) without ever assigning it -- line 142 streams the neighbour's POSITION into
// TmpTgtNormal, which line 143 then overwrites with the normal -- i.e. it computes with an uninitialised vector.  Here
// TmpTgtPoint is the neighbour's position, which is what line 142 evidently meant to write.
// vSrcCorNormal is normalised IN PLACE every time a candidate passes the NVDT test (:157), as there.
uint32_t orc_icrp_normal_fit(const double *src, const double *src_normals, uint32_t ns, const double *tgt,
                             const double *tgt_normals, uint32_t nt, const void *tgt_tree, const uint32_t *sample_idx,
                             uint32_t m, const double R[9], const double T[3], uint32_t k, double nvdt, double ndt,
                             int compare_normals, uint32_t *out_tgt, uint8_t *keep, double *pair_p, double *pair_q,
                             int threads) {
    (void)ns;
    (void)nt;
    if (orc_kd_old_k(tgt_tree) != k || !tgt_normals || (compare_normals && !src_normals)) return 0;
    std::vector<uint32_t> chosen(m, ORC_NO_INDEX);
    const int nth = threads > 1 ? threads : 1;
#pragma omp parallel for schedule(dynamic, 2048) num_threads(nth)
    for (long i = 0; i < (long)m; ++i) {
        const uint32_t si = sample_idx ? sample_idx[i] : (uint32_t)i;
        double p[3], sn[3] = {0, 0, 0};
        xform(R, T, src + (size_t)si * 3, p);  // :86
        if (src_normals) {                      // :87  CurSrcNormal = R * n
            const double *n = src_normals + (size_t)si * 3;
            for (int r = 0; r < 3; ++r) sn[r] = (R[r] * n[0] + R[3 + r] * n[1]) + R[6 + r] * n[2];
        }
        std::vector<uint32_t> nb(k);
        orc_kd_old_query_one(tgt_tree, p, nb.data(), nullptr);  // :89, ascending
        for (uint32_t c = 0; c < k; ++c) {                      // :140
            const uint32_t j = nb[c];
            double tn[3] = {tgt_normals[(size_t)j * 3], tgt_normals[(size_t)j * 3 + 1], tgt_normals[(size_t)j * 3 + 2]};  // :143
            normalize3(tn);                                                                                              // :145
            const double *q = tgt + (size_t)j * 3;  // TmpTgtPoint: see above
            double v[3] = {q[0] - p[0], q[1] - p[1], q[2] - p[2]};  // :146
            normalize3(v);                                          // :147
            const double nvd = std::fabs(((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]) -
                                         std::fabs((v[0] * tn[0] + v[1] * tn[1]) + v[2] * tn[2]));  // :149
            if (nvd < nvdt) {                                                                       // :155
                double nd = 0;
                if (src_normals) {
                    normalize3(sn);                                                    // :157
                    nd = std::fabs((tn[0] * sn[0] + tn[1] * sn[1]) + tn[2] * sn[2]);  // :158
                }
                if (!compare_normals || nd > ndt) {  // :164
                    chosen[i] = j;
                    break;  // :171
                }
            }
        }
    }
    uint32_t P = 0;
    for (uint32_t i = 0; i < m; ++i) {
        if (out_tgt) out_tgt[i] = chosen[i];
        if (keep) keep[i] = chosen[i] != ORC_NO_INDEX;
        if (chosen[i] != ORC_NO_INDEX) {
            if (pair_p) {
                const uint32_t si = sample_idx ? sample_idx[i] : i;
                xform(R, T, src + (size_t)si * 3, pair_p + (size_t)P * 3);
            }
            if (pair_q) std::memcpy(pair_q + (size_t)P * 3, tgt + (size_t)chosen[i] * 3, 24);
            ++P;
        }
    }
    return P;
}

void orc_svd3(const double H[9], double U[9], double S[3], double V[9]) {
    // Two-sided Jacobi (the method published for Eigen::JacobiSVD on real square input):
    // sweep the (p,q) pairs; each 2x2 block is first made symmetric by a left rotation, then
    // diagonalised by a Jacobi rotation applied on both sides.
    double W[9];
    double scale = 0;
    for (int i = 0; i < 9; ++i) scale = std::fmax(scale, std::fabs(H[i]));
    if (scale == 0) scale = 1;
    for (int i = 0; i < 9; ++i) W[i] = H[i] / scale;
    const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    std::memcpy(U, I3, sizeof I3);
    std::memcpy(V, I3, sizeof I3);
    const double eps = 2 * DBL_EPSILON, tiny = DBL_MIN;
    double maxdiag = std::fmax(std::fabs(M(W, 0, 0)), std::fmax(std::fabs(M(W, 1, 1)), std::fabs(M(W, 2, 2))));
    for (int sweep = 0; sweep < 64; ++sweep) {
        bool done = true;
        for (int p = 1; p < 3; ++p)
            for (int q = 0; q < p; ++q) {
                const double thr = std::fmax(tiny, eps * maxdiag);
                if (std::fabs(M(W, p, q)) <= thr && std::fabs(M(W, q, p)) <= thr) continue;
                done = false;
                // block [[a b],[c d]] on rows/cols (p,q)
                double a = M(W, p, p), b = M(W, p, q), c = M(W, q, p), d = M(W, q, q);
                // 1) left rotation making the block symmetric
                double c1 = 1, s1 = 0;
                const double tt = a + d, dd = c - b;
                if (std::fabs(dd) > tiny) {
                    const double u = tt / dd, h = std::sqrt(1 + u * u);
                    s1 = 1 / h;
                    c1 = u / h;
                }
                // rows: x' = c1 x + s1 y ; y' = -s1 x + c1 y
                rot_rows(W, p, q, c1, s1);
                rot_cols(U, p, q, c1, -s1);  // U <- U * G1 (G1^T applied to W on the left)
                a = M(W, p, p);
                b = M(W, p, q);
                d = M(W, q, q);
                // 2) symmetric Jacobi on [[a b],[b d]]
                double c2 = 1, s2 = 0;
                if (std::fabs(b) > tiny) {
                    const double tau = (d - a) / (2 * b);
                    const double t = (tau >= 0 ? 1.0 : -1.0) / (std::fabs(tau) + std::sqrt(1 + tau * tau));
                    c2 = 1 / std::sqrt(1 + t * t);
                    s2 = t * c2;
                }
                // W <- J^T W J with J = [[c2 s2],[-s2 c2]]
                rot_rows(W, p, q, c2, -s2);
                rot_cols(W, p, q, c2, s2);
                rot_cols(U, p, q, c2, s2);
                rot_cols(V, p, q, c2, s2);
                maxdiag = std::fmax(maxdiag, std::fmax(std::fabs(M(W, p, p)), std::fabs(M(W, q, q))));
            }
        if (done) break;
    }
    for (int i = 0; i < 3; ++i) {
        double s = M(W, i, i);
        if (s < 0) {
            s = -s;
            for (int r = 0; r < 3; ++r) M(U, r, i) = -M(U, r, i);
        }
        S[i] = s * scale;
    }
    for (int i = 0; i < 3; ++i) {  // sort descending, swapping columns of U and V
        int best = i;
        for (int j = i + 1; j < 3; ++j)
            if (S[j] > S[best]) best = j;
        if (best != i) {
            std::swap(S[i], S[best]);
            for (int r = 0; r < 3; ++r) {
                std::swap(M(U, r, i), M(U, r, best));
                std::swap(M(V, r, i), M(V, r, best));
            }
        }
    }
}

int orc_svd2(const double *pair_p, const double *pair_q, uint32_t P, double R[9], double T[3], double Rc[9],
             double Tc[3]) {
    if (P == 0) return -1;  // reference divides by zero here (RegUtilityFunctions.cpp:71)
    // compute3DCentroid, Release build: one thread => one sequential slot, then /= N
    double ca[3] = {0, 0, 0}, cb[3] = {0, 0, 0};
    for (uint32_t i = 0; i < P; ++i)
        for (int c = 0; c < 3; ++c) {
            ca[c] += pair_p[(size_t)i * 3 + c];
            cb[c] += pair_q[(size_t)i * 3 + c];
        }
    for (int c = 0; c < 3; ++c) {
        ca[c] /= (double)(int)P;
        cb[c] /= (double)(int)P;
    }
    // __deMean (:53-65) then H = A * B^T (:103): H(r,c) = sum_i a_i[r]*b_i[c], summed in i order
    double H[9] = {0};
    for (uint32_t i = 0; i < P; ++i) {
        double a[3], b[3];
        for (int c = 0; c < 3; ++c) {
            a[c] = pair_p[(size_t)i * 3 + c] - ca[c];
            b[c] = pair_q[(size_t)i * 3 + c] - cb[c];
        }
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) M(H, r, c) += a[r] * b[c];
    }
    double U[9], S[3], V[9];
    orc_svd3(H, U, S, V);
    if (det3(U) * det3(V) < 0)  // (:108-113)
        for (int r = 0; r < 3; ++r) M(V, r, 2) *= -1;
    double Ut[9], cr[9], ct[3];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) M(Ut, r, c) = M(U, c, r);
    matmul3(V, Ut, cr);  // (:115)
    for (int r = 0; r < 3; ++r)
        ct[r] = cb[r] - (M(cr, r, 0) * ca[0] + M(cr, r, 1) * ca[1] + M(cr, r, 2) * ca[2]);  // (:116)
    // accumulateTransformation: R <- Rc*R ; T <- Rc*T + Tc
    double nr[9], nt[3];
    matmul3(cr, R, nr);
    for (int r = 0; r < 3; ++r) nt[r] = (M(cr, r, 0) * T[0] + M(cr, r, 1) * T[1] + M(cr, r, 2) * T[2]) + ct[r];
    std::memcpy(R, nr, sizeof nr);
    std::memcpy(T, nt, sizeof nt);
    if (Rc) std::memcpy(Rc, cr, sizeof cr);
    if (Tc) std::memcpy(Tc, ct, sizeof ct);
    return 0;
}

struct OrcConv {
    orc_conv_params p;
    uint32_t similar;
    double pre_mse, cur_mse;
    double pre_rot[9];
};

void *orc_conv_create(const orc_conv_params *p) {
    OrcConv *c = new OrcConv;
    c->p = *p;
    c->similar = 0;  // __init, DefaultConvergenceCriterial.cpp:23-25
    c->pre_mse = std::numeric_limits<double>::max();
    c->cur_mse = std::numeric_limits<double>::max();
    // m_PreRot is never initialised in the reference (DefaultConvergenceCriterial.h:46);
    // NaN makes the first transformation test false, which is what any garbage far from a
    // rotation matrix does.
    for (double &v : c->pre_rot) v = std::numeric_limits<double>::quiet_NaN();
    return c;
}

void orc_conv_free(void *conv) { delete static_cast<OrcConv *>(conv); }

static double calc_mse(const double *pp, const double *pq, uint32_t P) {  // __calculateMSE (:46-65)
    double s = 0;
    for (uint32_t i = 0; i != P; ++i) s += sqdist(pp + (size_t)i * 3, pq + (size_t)i * 3);
    return s / (double)P;
}

int orc_conv_step(void *conv, const double R[9], const double T[3], uint32_t iter, const double *pair_p,
                  const double *pair_q, uint32_t P, double uniq, int *state_out, double *mse_out) {
    OrcConv &c = *static_cast<OrcConv *>(conv);
    int state = 0;
    bool converged = false;
    auto similar_gate = [&](int st) {  // the MIST counter shared by the three tests
        if (c.similar < c.p.max_similar) {
            ++c.similar;
            return false;
        }
        c.similar = 0;
        state = st;
        return true;
    };
    do {
        // (1) __isIterationEndedV (:86-99)
        if (iter >= c.p.max_iterations) {
            state = 1;
            if (!c.p.failure_after_max_iter) {
                converged = true;
                break;
            }
        }
        // (2) __isTransformationEnded (:103-133); T is the ACCUMULATED translation
        double rot = 0;
        for (int i = 0; i < 9; ++i) rot += (c.pre_rot[i] - R[i]) * (c.pre_rot[i] - R[i]);
        const double tr = T[0] * T[0] + T[1] * T[1] + T[2] * T[2];
        if (rot < c.p.rotation_threshold && tr < c.p.translation_threshold && similar_gate(2)) {
            converged = true;
            break;
        }
        // (3) __isMSEAbsoluteEnded (:137-170)
        c.cur_mse = calc_mse(pair_p, pair_q, P);
        if (std::fabs(c.cur_mse - c.pre_mse) < c.p.abs_mse * uniq && similar_gate(3)) {
            converged = true;
            break;
        }
        // (4) __isMSERelativeEnded (:174-207)
        c.cur_mse = calc_mse(pair_p, pair_q, P);
        if (std::fabs(c.cur_mse - c.pre_mse) / c.pre_mse < c.p.rel_mse * uniq && similar_gate(4)) {
            converged = true;
            break;
        }
    } while (false);
    c.pre_mse = c.cur_mse;  // (:77-79)
    std::memcpy(c.pre_rot, R, sizeof c.pre_rot);
    if (state_out) *state_out = state;
    if (mse_out) *mse_out = c.cur_mse;
    return converged ? 1 : 0;
}

uint32_t orc_fit(const double *src, uint32_t ns, const double *tgt, uint32_t nt, const uint32_t *sample_idx,
                 uint32_t m, const orc_fit_params *prm, double R[9], double T[3], int *state_out, double *traj,
                 uint32_t traj_cap, int threads) {
    // __setPointCloud (:179-191): target unit; __setMethodPointCloud (:202): target tree with K=NN
    const double U = orc_uniq_sq_dist(tgt, nt);
    void *ttree = orc_kd_old_build(tgt, nt, prm->k);
    void *conv = orc_conv_create(&prm->conv);
    std::vector<double> pp((size_t)m * 3), pq((size_t)m * 3);
    uint32_t iter = 0;
    int state = 0;
    if (m < 5) {  // _isCorrespondencePointSetValidV (:108-115)
        if (state_out) *state_out = -1;
        orc_kd_old_free(ttree);
        orc_conv_free(conv);
        return 0;
    }
    for (;;) {  // __fitLoop (:119-129), _fitStepV (:133-175)
        ++iter;
        const uint32_t P = orc_icrp_correspondences(src, ns, tgt, nt, ttree, sample_idx, m, R, T, prm->k, U * prm->sdft,
                                                    U * prm->ndft, nullptr, nullptr, nullptr, pp.data(), pq.data(),
                                                    threads);
        // CCorrespondenceRejectionClosestPoints: pass-through
        if (orc_svd2(pp.data(), pq.data(), P, R, T, nullptr, nullptr) != 0) {
            state = 5;  // reference: division by zero -> NaN transform; oracle stops instead
            break;
        }
        double mse;
        const int conv_now = orc_conv_step(conv, R, T, iter, pp.data(), pq.data(), P, U, &state, &mse);
        if (traj && iter <= traj_cap) {
            double *t = traj + (size_t)(iter - 1) * 14;
            std::memcpy(t, R, 72);
            std::memcpy(t + 9, T, 24);
            t[12] = mse;
            t[13] = (double)P;
        }
        if (conv_now) break;
    }
    if (state_out) *state_out = state;
    orc_kd_old_free(ttree);
    orc_conv_free(conv);
    return iter;
}
}
