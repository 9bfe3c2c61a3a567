// The 3x3 problem the 17 partial sums reduce to, as host+device code: the same source is compiled into
// pcr_solve_rigid (host, pcr_solve.cpp) and into the device-resident fit loop (fit_solve_kernel, pcr_icp.hip), so
// the two give bit-identical transforms (IEEE fp64, no FMA contraction, correctly rounded sqrt and division).
//
// Replaces CTransformationEstimationSVD2::__computeRelativeRotationAndTranslation
// (TransformationEstimationSVD2.cpp:98-117) and CBaseTransformationEstimation::accumulateTransformation
// (BaseTransformationEstimation.h:28-32).  The reference forms H = sum (a_i - abar)(b_i - bbar)^T from explicit
// de-meaned copies; here H = sum a' b'^T - n abar' bbar'^T with a' = a - pivot_p, b' = b - pivot_q (pivots near
// the centroids, so the subtraction does not cancel), which is the same matrix.
// The SVD is a one-sided (Hestenes) Jacobi iteration -- Eigen is not a dependency.  For non-degenerate H the
// result R = V diag(1,1,+-1) U^T is the unique polar factor, independent of the SVD algorithm.
#pragma once
#include <math.h>
#include "pcr_internal.h"

#define PCR_HD __host__ __device__

namespace pcr {

// all matrices column-major: A(r,c) = A[c*3+r]
PCR_HD inline double &el(double *A, int r, int c) { return A[c * 3 + r]; }
PCR_HD inline double el(const double *A, int r, int c) { return A[c * 3 + r]; }

PCR_HD inline double det3(const double *A) {
    return el(A, 0, 0) * (el(A, 1, 1) * el(A, 2, 2) - el(A, 1, 2) * el(A, 2, 1)) -
           el(A, 0, 1) * (el(A, 1, 0) * el(A, 2, 2) - el(A, 1, 2) * el(A, 2, 0)) +
           el(A, 0, 2) * (el(A, 1, 0) * el(A, 2, 1) - el(A, 1, 1) * el(A, 2, 0));
}

PCR_HD inline void cross(const double *a, const double *b, double *o) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

PCR_HD inline double norm3(const double *a) { return sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }

// One Jacobi rotation of columns I < J of A (and of V), every index a compile-time constant so that A and V live in
// registers on the device (a run-time column index sends them to scratch memory, and the one thread that solves
// then waits on memory for most of its life).  false: the columns already are orthogonal to within two units of
// roundoff -- a stricter test is not reachable in general (gamma then hovers at rounding noise and the sweeps run
// to the limit).
template <int I, int J>
PCR_HD inline bool jacobi_pair(double *A, double *V) {
    double alpha = 0, beta = 0, gamma = 0;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        alpha += A[I * 3 + r] * A[I * 3 + r];
        beta += A[J * 3 + r] * A[J * 3 + r];
        gamma += A[I * 3 + r] * A[J * 3 + r];
    }
    // (one thread runs this, and on the device every fp64 square root or division is a dependent chain of two dozen
    // instructions -- the rotations ARE the solve's time.  The test is taken on squares.  With d = beta - alpha,
    // r = sqrt(d^2 + 4 gamma^2) and u = |d| + r the textbook t = sign(d) 2 gamma / u has 1 + t^2 = 2 r / u, hence
    // c = u / w and s = sign(d) 2 gamma / w with w = sqrt(2 r u): two square roots and ONE division per rotation (round
    // 2: two and two, the textbook three and three), and c, s come from one reciprocal instead of one after the other.)
    const double g2 = gamma * gamma;
    if (g2 <= 1.9721522630525295e-31 * (alpha * beta)) return false;  // |gamma| <= 2^-51 sqrt(alpha beta)
    const double d = beta - alpha;
    const double r = sqrt(d * d + 4.0 * g2), u = fabs(d) + r;
    const double iw = 1.0 / sqrt(2.0 * r * u);
    const double c = u * iw, s = (d >= 0 ? 2.0 * gamma : -2.0 * gamma) * iw;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const double ai = A[I * 3 + r], aj = A[J * 3 + r];
        A[I * 3 + r] = c * ai - s * aj;
        A[J * 3 + r] = s * ai + c * aj;
        const double vi = V[I * 3 + r], vj = V[J * 3 + r];
        V[I * 3 + r] = c * vi - s * vj;
        V[J * 3 + r] = s * vi + c * vj;
    }
    return true;
}

// columns I < J of A and V, and their singular values, into descending order (a stable exchange: only on strict >)
template <int I, int J>
PCR_HD inline void order_pair(double *sig, double *A, double *V) {
    if (sig[J] > sig[I]) {
        double t = sig[I];
        sig[I] = sig[J];
        sig[J] = t;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            t = A[I * 3 + r];
            A[I * 3 + r] = A[J * 3 + r];
            A[J * 3 + r] = t;
            t = V[I * 3 + r];
            V[I * 3 + r] = V[J * 3 + r];
            V[J * 3 + r] = t;
        }
    }
}

// H = U diag(S) V^T, S descending, U and V orthogonal.
PCR_HD inline void svd3_hestenes(const double *H, double *U, double *S, double *V) {
    double A[9];
    double scale = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) scale = fmax(scale, fabs(H[i]));
    const double inv = scale > 0 ? 1.0 / scale : 1.0;
#pragma unroll
    for (int i = 0; i < 9; ++i) A[i] = H[i] * inv;
#pragma unroll
    for (int i = 0; i < 9; ++i) V[i] = (i == 0 || i == 4 || i == 8) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool rotated = jacobi_pair<0, 1>(A, V);
        rotated |= jacobi_pair<0, 2>(A, V);
        rotated |= jacobi_pair<1, 2>(A, V);
        if (!rotated) break;
    }
    // columns of A are now orthogonal: sigma_i = |col_i|; descending (the selection order of the exchanges below is
    // the one a selection sort over an index array makes: (0,1), (0,2), (1,2))
    double sig[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) sig[i] = norm3(&A[i * 3]);
    order_pair<0, 1>(sig, A, V);
    order_pair<0, 2>(sig, A, V);
    order_pair<1, 2>(sig, A, V);
    int rank = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        S[i] = sig[i] * scale;
        if (sig[i] > 1e-300 && sig[i] > 1e-15 * sig[0]) {
#pragma unroll
            for (int r = 0; r < 3; ++r) U[i * 3 + r] = A[i * 3 + r] / sig[i];
            rank = i + 1;
        }
    }
    // complete U to an orthonormal basis when H is rank deficient (planar / collinear pair sets)
    if (rank == 0) {
#pragma unroll
        for (int i = 0; i < 9; ++i) U[i] = (i == 0 || i == 4 || i == 8) ? 1.0 : 0.0;
    } else if (rank == 1) {
        double e[3] = {0, 0, 0};
        // axis least aligned with u0
        const double a0 = fabs(U[0]), a1 = fabs(U[1]), a2 = fabs(U[2]);
        int m = 0;
        double am = a0;
        if (a1 < am) {
            m = 1;
            am = a1;
        }
        if (a2 < am) m = 2;
        e[0] = m == 0 ? 1.0 : 0.0;
        e[1] = m == 1 ? 1.0 : 0.0;
        e[2] = m == 2 ? 1.0 : 0.0;
        double u1[3], u2[3];
        cross(&U[0], e, u1);
        const double n1 = norm3(u1);
#pragma unroll
        for (int r = 0; r < 3; ++r) u1[r] /= n1;
        cross(&U[0], u1, u2);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            U[3 + r] = u1[r];
            U[6 + r] = u2[r];
        }
    } else if (rank == 2) {
        double u2[3];
        cross(&U[0], &U[3], u2);
        const double n2 = norm3(u2);
#pragma unroll
        for (int r = 0; r < 3; ++r) U[6 + r] = u2[r] / n2;
    }
}


// n = partials[0] >= 1 is the caller's business.  R,T in/out = accumulated transform; Rc,Tc,mse nullable.
PCR_HD inline void solve_rigid_core(const double *partials, const double *pivot_p, const double *pivot_q, double *R,
                                    double *T, double *Rc, double *Tc, double *mse) {
    const double n = partials[0];
    const double *Sa = partials + 1, *Sb = partials + 4, *Sab = partials + 7;
    double ma[3], mb[3], ca[3], cb[3];
    for (int a = 0; a < 3; ++a) {
        ma[a] = Sa[a] / n;  // centroid offsets from the pivots
        mb[a] = Sb[a] / n;
        ca[a] = pivot_p[a] + ma[a];  // compute3DCentroid results (RegUtilityFunctions.cpp:44-72)
        cb[a] = pivot_q[a] + mb[a];
    }
    double H[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) el(H, r, c) = Sab[r * 3 + c] - n * ma[r] * mb[c];  // (:103)
    double U[9], S[3], V[9];
    svd3_hestenes(H, U, S, V);  // (:104-106)
    if (det3(U) * det3(V) < 0)  // (:108-113)
        for (int r = 0; r < 3; ++r) el(V, r, 2) = -el(V, r, 2);
    double cr[9], ct[3];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)  // Rc = V * U^T (:115)
            el(cr, r, c) = el(V, r, 0) * el(U, c, 0) + el(V, r, 1) * el(U, c, 1) + el(V, r, 2) * el(U, c, 2);
    for (int r = 0; r < 3; ++r)  // Tc = bbar - Rc * abar (:116)
        ct[r] = cb[r] - (el(cr, r, 0) * ca[0] + el(cr, r, 1) * ca[1] + el(cr, r, 2) * ca[2]);
    double nr[9], nt[3];  // accumulateTransformation: R <- Rc*R ; T <- Rc*T + Tc
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            el(nr, r, c) = el(cr, r, 0) * el(R, 0, c) + el(cr, r, 1) * el(R, 1, c) + el(cr, r, 2) * el(R, 2, c);
    for (int r = 0; r < 3; ++r) nt[r] = (el(cr, r, 0) * T[0] + el(cr, r, 1) * T[1] + el(cr, r, 2) * T[2]) + ct[r];
    for (int i = 0; i < 9; ++i) R[i] = nr[i];
    for (int i = 0; i < 3; ++i) T[i] = nt[i];
    if (Rc)
        for (int i = 0; i < 9; ++i) Rc[i] = cr[i];
    if (Tc)
        for (int i = 0; i < 3; ++i) Tc[i] = ct[i];
    if (mse) *mse = partials[16] / n;
}

// Inverse of a near-rotation (column-major) and its orthogonality defect max |R^T R - I|.  false if singular.
PCR_HD inline bool invert3(const double *R, double *inv, double &defect) {
    const double c00 = el(R, 1, 1) * el(R, 2, 2) - el(R, 1, 2) * el(R, 2, 1);
    const double c01 = el(R, 1, 2) * el(R, 2, 0) - el(R, 1, 0) * el(R, 2, 2);
    const double c02 = el(R, 1, 0) * el(R, 2, 1) - el(R, 1, 1) * el(R, 2, 0);
    const double det = el(R, 0, 0) * c00 + el(R, 0, 1) * c01 + el(R, 0, 2) * c02;
    if (!(fabs(det) > 0.5)) return false;
    const double id = 1.0 / det;
    double m[3][3];
    m[0][0] = c00 * id;
    m[0][1] = (el(R, 0, 2) * el(R, 2, 1) - el(R, 0, 1) * el(R, 2, 2)) * id;
    m[0][2] = (el(R, 0, 1) * el(R, 1, 2) - el(R, 0, 2) * el(R, 1, 1)) * id;
    m[1][0] = c01 * id;
    m[1][1] = (el(R, 0, 0) * el(R, 2, 2) - el(R, 0, 2) * el(R, 2, 0)) * id;
    m[1][2] = (el(R, 0, 2) * el(R, 1, 0) - el(R, 0, 0) * el(R, 1, 2)) * id;
    m[2][0] = c02 * id;
    m[2][1] = (el(R, 0, 1) * el(R, 2, 0) - el(R, 0, 0) * el(R, 2, 1)) * id;
    m[2][2] = (el(R, 0, 0) * el(R, 1, 1) - el(R, 0, 1) * el(R, 1, 0)) * id;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) inv[c * 3 + r] = m[r][c];
    defect = 0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int r = 0; r < 3; ++r) s += el(R, r, i) * el(R, r, j);
            defect = fmax(defect, fabs(s - (i == j ? 1.0 : 0.0)));
        }
    return true;
}

}  // namespace pcr
