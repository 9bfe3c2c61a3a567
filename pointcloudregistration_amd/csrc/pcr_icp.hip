// libpcr.so -- one ICRP iteration as fused device work (gfx950).
//
// Reference dataflow (SURVEY.md Appendix C;
// FineRegistrationClosestPoints/CorrespondenceEstimationClosestPoints.cpp:54-117,
// TransformationEstimationSVD2.cpp:33-103, DefaultConvergenceCriterial.cpp:46-65):
//   copy + transform the whole source, build a KD-tree on it, then serially per sample:
//   forward K-NN, reverse 1-NN, two gates, push_back; later 5 more passes for centroids and H.
// Here: one streaming transform kernel, then ONE kernel in which each lane owns a sample and does
// transform -> forward K-NN (target grid) -> reverse 1-NN (source grid, transformed candidates)
// -> gates -> masked fp64 accumulation, followed by a wave/block tree reduction; a second tiny
// kernel folds the block partials in a fixed order (bitwise reproducible run to run).
#include <math.h>
#include <string.h>
#include "knn_device.h"
#ifndef PCR_STEP_WAVES
#define PCR_STEP_WAVES 4
#endif

namespace pcr {

__global__ void __launch_bounds__(256) transform_kernel(const Pt4 *__restrict__ in, uint32_t n, Rigid M,
                                                        Pt4 *__restrict__ out) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const Pt4 s = in[i];
        Pt4 o;
        xform(M, s.x, s.y, s.z, o.x, o.y, o.z);
        o.w = s.w;
        out[i] = o;
    }
}

struct StepArgs {
    GridView src, tgt;
    const Pt4 *tsrc;             // transformed source, parallel to src.pts
    const uint32_t *src_pos_of;  // original index -> position in src.pts
    const uint32_t *src_order;   // positions of src.pts in brick order (processing order of "all samples")
    const uint32_t *sample_idx;  // NULL: samples are positions [sample_begin, sample_begin+m)
    uint32_t sample_begin, m;
    Rigid M;         // current (R,T)
    double rinv[9];  // inverse of R (column-major), for locating q in the source grid
    int k;
    double thr_fwd, thr_rev;
    uint32_t exact;  // PCR_STEP_EXACT_INDICES
    double rev_bound_scale, rev_extra_margin;
    double pivot_p[3], pivot_q[3];
    double *block_partials;  // [17][gridDim.x]
    uint32_t *out_fwd, *out_rev;
    uint8_t *out_keep;
};

// 17 per-lane values -> one value per block and column, stored column-major [17][nblocks] so that the fold
// reads coalesced.  Wave reduction by shuffles, then LDS across the 4 waves.
__device__ __forceinline__ void block_reduce17(double (&v)[17], double *__restrict__ out, uint32_t block,
                                               uint32_t nblocks) {
    __shared__ double sm[4][17];
#pragma unroll
    for (int j = 0; j < 17; ++j) {
        double x = v[j];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        v[j] = x;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < 17; ++j) sm[wave][j] = v[j];
    }
    __syncthreads();
    if (threadIdx.x < 17)
        out[(size_t)threadIdx.x * nblocks + block] =
            (sm[0][threadIdx.x] + sm[1][threadIdx.x]) + (sm[2][threadIdx.x] + sm[3][threadIdx.x]);
}

__device__ __forceinline__ void accumulate(double (&acc)[17], bool keep, double px, double py, double pz, double qx,
                                           double qy, double qz, double dF, const StepArgs &A) {
    const double ax = px - A.pivot_p[0], ay = py - A.pivot_p[1], az = pz - A.pivot_p[2];
    const double bx = qx - A.pivot_q[0], by = qy - A.pivot_q[1], bz = qz - A.pivot_q[2];
    const double w = keep ? 1.0 : 0.0;
    acc[0] = w;
    acc[1] = keep ? ax : 0.0;
    acc[2] = keep ? ay : 0.0;
    acc[3] = keep ? az : 0.0;
    acc[4] = keep ? bx : 0.0;
    acc[5] = keep ? by : 0.0;
    acc[6] = keep ? bz : 0.0;
    acc[7] = keep ? ax * bx : 0.0;
    acc[8] = keep ? ax * by : 0.0;
    acc[9] = keep ? ax * bz : 0.0;
    acc[10] = keep ? ay * bx : 0.0;
    acc[11] = keep ? ay * by : 0.0;
    acc[12] = keep ? ay * bz : 0.0;
    acc[13] = keep ? az * bx : 0.0;
    acc[14] = keep ? az * by : 0.0;
    acc[15] = keep ? az * bz : 0.0;
    acc[16] = keep ? dF : 0.0;
}

template <int KC>
__global__ void __launch_bounds__(256, PCR_STEP_WAVES) icp_step_grid_kernel(StepArgs A) {
    const uint32_t lb = xcd_remap(blockIdx.x, gridDim.x);
    const uint32_t t = lb * 256u + threadIdx.x;
    const bool live = t < A.m;
    double acc[17];
#pragma unroll
    for (int j = 0; j < 17; ++j) acc[j] = 0.0;
    uint32_t pos = 0, orig = 0, slot = 0;
    double px = 0, py = 0, pz = 0;
    if (live) {
        if (A.sample_idx) {
            orig = A.sample_idx[t];
            pos = A.src_pos_of[orig];
            slot = t;
        } else {
            pos = A.src_order[A.sample_begin + t];  // brick order: a wave's samples are neighbours in space
            orig = (uint32_t)A.src.pts[pos].w;
            slot = orig;
        }
        const Pt4 s = A.src.pts[pos];
        xform(A.M, s.x, s.y, s.z, px, py, pz);  // CurSrcPoint = R*s + T (:86)
    }
    // ---- forward: K-th nearest target point of p (:89-92) ----
    const double fmax_d2 = A.exact ? PCR_INF : A.thr_fwd;
    const bool fin = live && isfinite(px) && isfinite(py) && isfinite(pz);
    TopK<KC> top;
    top.init(fmax_d2);
    __shared__ uint2 rowtab[9][256];  // per-lane row table of the ring walk (knn_device.h)
    uint2 *my_rows = &rowtab[0][threadIdx.x];
    if (fin)
        lane_search<KC>(A.tgt, A.tgt.pts, px, py, pz, px, py, pz, fmax_d2, 1.0 - 1.8189894035458565e-12, 0.0, top, my_rows);
    double dF;
    uint32_t j;
    top.slot(A.k, dF, j);  // NeighborSet.back() = K-th nearest (:91-92)
    const bool found = fin && j != PCR_NO_INDEX && dF < fmax_d2;
    const bool passF = found && dF < A.thr_fwd;  // strict < (:97)

    // ---- reverse: nearest TRANSFORMED source point of q (:94-95).  The sample's own image p is a member
    // of that set at distance exactly dF: seeding with it changes nothing and bounds the search. ----
    const bool need_rev = found && (passF || A.exact);
    double qx = 0, qy = 0, qz = 0, sx = 0, sy = 0, sz = 0;
    if (need_rev) {
        const Pt4 qpt = A.tgt.pts[A.tgt.pos_of[j]];
        qx = qpt.x;
        qy = qpt.y;
        qz = qpt.z;
        const double ux = qx - A.M.t[0], uy = qy - A.M.t[1], uz = qz - A.M.t[2];
        sx = A.rinv[0] * ux + A.rinv[3] * uy + A.rinv[6] * uz;
        sy = A.rinv[1] * ux + A.rinv[4] * uy + A.rinv[7] * uz;
        sz = A.rinv[2] * ux + A.rinv[5] * uy + A.rinv[8] * uz;
    }
    TopK<1> rt;
    rt.d[0] = dF;
    rt.idx[0] = orig;
    const bool rev_active = need_rev && isfinite(sx) && isfinite(sy) && isfinite(sz);
    if (rev_active)
        lane_search<1>(A.src, A.tsrc, qx, qy, qz, sx, sy, sz, PCR_INF, A.rev_bound_scale, A.rev_extra_margin, rt, my_rows);
    uint32_t r = PCR_NO_INDEX;
    bool keep = false;
    if (need_rev) {
        r = rt.idx[0];
        const Pt4 pp = A.tsrc[r == orig ? pos : A.src_pos_of[r]];
        const double dR = sqdist3(px, py, pz, pp.x, pp.y, pp.z);  // (CurSrcPoint - TmpPoint).squaredNorm()
        keep = passF && dR < A.thr_rev;                           // strict < (:98)
    }
    if (live) {
        accumulate(acc, keep, px, py, pz, qx, qy, qz, dF, A);
        if (A.out_fwd) A.out_fwd[slot] = found ? j : PCR_NO_INDEX;
        if (A.out_rev) A.out_rev[slot] = r;
        if (A.out_keep) A.out_keep[slot] = keep ? 1 : 0;
    }
    block_reduce17(acc, A.block_partials, lb, gridDim.x);
}

// Small clouds: both searches as tiled brute force (LDS-staged tiles, broadcast reads).
template <int KC>
__global__ void __launch_bounds__(256) icp_step_brute_kernel(StepArgs A) {
    __shared__ Pt4 tile[512];
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    const bool live = t < A.m;
    double acc[17];
#pragma unroll
    for (int j = 0; j < 17; ++j) acc[j] = 0.0;
    uint32_t pos = 0, orig = 0, slot = 0;
    double px = 0, py = 0, pz = 0;
    if (live) {
        if (A.sample_idx) {
            orig = A.sample_idx[t];
            pos = A.src_pos_of[orig];
            slot = t;
        } else {
            pos = A.sample_begin + t;
            orig = (uint32_t)A.src.pts[pos].w;
            slot = orig;
        }
        const Pt4 s = A.src.pts[pos];
        xform(A.M, s.x, s.y, s.z, px, py, pz);
    }
    const bool fin = live && isfinite(px) && isfinite(py) && isfinite(pz);
    TopK<KC> top;
    top.init();
    for (uint32_t base = 0; base < A.tgt.n; base += 512) {
        const int cnt = (int)min(512u, A.tgt.n - base);
        for (int u = threadIdx.x; u < cnt; u += 256) tile[u] = A.tgt.pts[base + u];
        __syncthreads();
        if (fin) tile_scan<KC>(tile, cnt, px, py, pz, top);
        __syncthreads();
    }
    double dF;
    uint32_t j;
    top.slot(A.k, dF, j);
    const bool found = fin && j != PCR_NO_INDEX;
    const bool passF = found && dF < A.thr_fwd;
    double qx = 0, qy = 0, qz = 0;
    if (found) {
        const Pt4 qpt = A.tgt.pts[A.tgt.pos_of[j]];
        qx = qpt.x;
        qy = qpt.y;
        qz = qpt.z;
    }
    TopK<1> rt;
    rt.d[0] = found ? dF : PCR_INF;
    rt.idx[0] = found ? orig : PCR_NO_INDEX;
    for (uint32_t base = 0; base < A.src.n; base += 512) {
        const int cnt = (int)min(512u, A.src.n - base);
        for (int u = threadIdx.x; u < cnt; u += 256) tile[u] = A.tsrc[base + u];
        __syncthreads();
        if (found) tile_scan<1>(tile, cnt, qx, qy, qz, rt);
        __syncthreads();
    }
    bool keep = false;
    uint32_t r = PCR_NO_INDEX;
    if (found) {
        r = rt.idx[0];
        const Pt4 pp = A.tsrc[A.src_pos_of[r]];
        const double dR = sqdist3(px, py, pz, pp.x, pp.y, pp.z);
        keep = passF && dR < A.thr_rev;
    }
    if (live) {
        accumulate(acc, keep, px, py, pz, qx, qy, qz, dF, A);
        if (A.out_fwd) A.out_fwd[slot] = found ? j : PCR_NO_INDEX;
        if (A.out_rev) A.out_rev[slot] = r;
        if (A.out_keep) A.out_keep[slot] = keep ? 1 : 0;
    }
    block_reduce17(acc, A.block_partials, blockIdx.x, gridDim.x);
}

// Fold the block partials in a fixed order: block j of the grid owns column j (one of the 17 sums); its 256
// threads stride over the blocks' values (coalesced), then a tree.  Same order every run => the sums are
// bitwise reproducible run to run.
__global__ void __launch_bounds__(256) fold_partials_kernel(const double *__restrict__ cols, uint32_t nblocks,
                                                            double *__restrict__ out /*[17]*/) {
    __shared__ double sm[256];
    const double *col = cols + (size_t)blockIdx.x * nblocks;
    double s = 0.0;
    for (uint32_t r = threadIdx.x; r < nblocks; r += 256) s += col[r];
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sm[threadIdx.x] += sm[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = sm[0];
}

// Pair lists that already exist (a host-side rejection stage produced them): the same 17 sums over P explicit
// pairs, same block reduction and fold.  pairs are AoS double[3P].
__global__ void __launch_bounds__(256) reduce_pairs_kernel(const double *__restrict__ pp, const double *__restrict__ pq,
                                                           uint32_t P, StepArgs A) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    double acc[17];
#pragma unroll
    for (int j = 0; j < 17; ++j) acc[j] = 0.0;
    if (t < P) {
        const double px = pp[3 * (size_t)t], py = pp[3 * (size_t)t + 1], pz = pp[3 * (size_t)t + 2];
        const double qx = pq[3 * (size_t)t], qy = pq[3 * (size_t)t + 1], qz = pq[3 * (size_t)t + 2];
        accumulate(acc, true, px, py, pz, qx, qy, qz, sqdist3(px, py, pz, qx, qy, qz), A);
    }
    block_reduce17(acc, A.block_partials, blockIdx.x, gridDim.x);
}

static bool invert3(const double R[9], double inv[9], double &defect) {
    // column-major: R(r,c) = R[c*3+r]
    auto at = [&](int r, int c) { return R[c * 3 + r]; };
    const double c00 = at(1, 1) * at(2, 2) - at(1, 2) * at(2, 1);
    const double c01 = at(1, 2) * at(2, 0) - at(1, 0) * at(2, 2);
    const double c02 = at(1, 0) * at(2, 1) - at(1, 1) * at(2, 0);
    const double det = at(0, 0) * c00 + at(0, 1) * c01 + at(0, 2) * c02;
    if (!(fabs(det) > 0.5)) return false;
    const double id = 1.0 / det;
    double m[3][3];
    m[0][0] = c00 * id;
    m[0][1] = (at(0, 2) * at(2, 1) - at(0, 1) * at(2, 2)) * id;
    m[0][2] = (at(0, 1) * at(1, 2) - at(0, 2) * at(1, 1)) * id;
    m[1][0] = c01 * id;
    m[1][1] = (at(0, 0) * at(2, 2) - at(0, 2) * at(2, 0)) * id;
    m[1][2] = (at(0, 2) * at(1, 0) - at(0, 0) * at(1, 2)) * id;
    m[2][0] = c02 * id;
    m[2][1] = (at(0, 1) * at(2, 0) - at(0, 0) * at(2, 1)) * id;
    m[2][2] = (at(0, 0) * at(1, 1) - at(0, 1) * at(1, 0)) * id;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) inv[c * 3 + r] = m[r][c];
    defect = 0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int r = 0; r < 3; ++r) s += at(r, i) * at(r, j);
            defect = fmax(defect, fabs(s - (i == j ? 1.0 : 0.0)));
        }
    return true;
}

static void cloud_scale(const pcr_cloud *c, double &scale) {
    for (int a = 0; a < 3; ++a) scale = fmax(scale, fmax(fabs(c->lo[a]), fabs(c->hi[a])));
}

}  // namespace pcr

using namespace pcr;

extern "C" {

int pcr_icp_pivots(const pcr_index *src, const pcr_index *tgt, const double R[9], const double T[3],
                   double pivot_p[3], double pivot_q[3]) {
    PCR_REQUIRE(src && tgt && R && T && pivot_p && pivot_q, "pcr_icp_pivots: NULL argument");
    double c[3];
    for (int a = 0; a < 3; ++a) {
        c[a] = 0.5 * (src->cloud->lo[a] + src->cloud->hi[a]);
        pivot_q[a] = 0.5 * (tgt->cloud->lo[a] + tgt->cloud->hi[a]);
    }
    for (int r = 0; r < 3; ++r) pivot_p[r] = R[r] * c[0] + R[3 + r] * c[1] + R[6 + r] * c[2] + T[r];
    return PCR_OK;
}

int pcr_icp_step_d(pcr_ctx *ctx, const pcr_index *src, const pcr_index *tgt, const uint32_t *sample_idx_d,
                   uint32_t sample_begin, uint32_t m, const double R[9], const double T[3], uint32_t k,
                   double thr_fwd_sq, double thr_rev_sq, uint32_t flags, double *partials_d, uint32_t *out_fwd_d,
                   uint32_t *out_rev_d, uint8_t *out_keep_d) {
    PCR_REQUIRE(ctx && src && tgt && R && T && partials_d, "pcr_icp_step_d: NULL argument");
    PCR_REQUIRE(src->ctx == ctx && tgt->ctx == ctx, "pcr_icp_step_d: index belongs to another context");
    PCR_REQUIRE(src->kind == tgt->kind, "pcr_icp_step_d: source and target index must be of the same kind");
    PCR_REQUIRE(k >= 1 && k <= tgt->n, "pcr_icp_step_d: k=%u out of range for %u target points", k, tgt->n);
    if (k > PCR_MAX_K) {
        set_last_error("pcr_icp_step_d: k=%u above PCR_MAX_K=%u", k, PCR_MAX_K);
        return PCR_E_UNSUPPORTED;
    }
    if (!sample_idx_d)
        PCR_REQUIRE((uint64_t)sample_begin + m <= src->n, "pcr_icp_step_d: sample range [%u,%u+%u) exceeds %u source points",
                    sample_begin, sample_begin, m, src->n);
    for (int i = 0; i < 9; ++i) PCR_REQUIRE(isfinite(R[i]), "pcr_icp_step_d: R is not finite");
    for (int i = 0; i < 3; ++i) PCR_REQUIRE(isfinite(T[i]), "pcr_icp_step_d: T is not finite");
    PCR_HIP(hipSetDevice(ctx->device));

    StepArgs A;
    memset(&A, 0, sizeof A);
    double defect = 0;
    PCR_REQUIRE(invert3(R, A.rinv, defect) && defect <= 1e-3,
                "pcr_icp_step_d: R must be a rotation to within 1e-3 (orthogonality defect %g)", defect);
    A.src = src->view;
    A.tgt = tgt->view;
    A.tsrc = src->tpts_d;
    A.src_pos_of = src->pos_of_d;
    A.src_order = src->order_d;
    A.sample_idx = sample_idx_d;
    A.sample_begin = sample_idx_d ? 0 : sample_begin;
    A.m = m;
    memcpy(A.M.r, R, sizeof A.M.r);
    memcpy(A.M.t, T, sizeof A.M.t);
    A.k = (int)k;
    A.thr_fwd = thr_fwd_sq;
    A.thr_rev = thr_rev_sq;
    A.exact = (flags & PCR_STEP_EXACT_INDICES) ? 1u : 0u;
    // |R v| >= sigma_min |v| with sigma_min^2 >= 1 - 3*defect (Gershgorin on R^T R)
    A.rev_bound_scale = (1.0 - 3.0 * defect) * (1.0 - 1.8189894035458565e-12);
    double scale = fabs(T[0]) + fabs(T[1]) + fabs(T[2]);
    cloud_scale(src->cloud, scale);
    cloud_scale(tgt->cloud, scale);
    A.rev_extra_margin = 3.637978807091713e-12 * scale;  // 2^-38
    pcr_icp_pivots(src, tgt, R, T, A.pivot_p, A.pivot_q);
    A.out_fwd = out_fwd_d;
    A.out_rev = out_rev_d;
    A.out_keep = out_keep_d;

    const uint32_t nb = m ? (m + 255) / 256 : 0;
    if (nb == 0) {
        PCR_HIP(hipMemsetAsync(partials_d, 0, sizeof(double) * 17, ctx->stream));
        return PCR_OK;
    }
    if ((size_t)nb * 17 > ctx->block_partials_cap) {
        if (ctx->block_partials_d) {
            PCR_HIP(hipStreamSynchronize(ctx->stream));
            hipFree(ctx->block_partials_d);
            ctx->block_partials_d = nullptr;
            ctx->block_partials_cap = 0;
        }
        PCR_HIP(hipMalloc(&ctx->block_partials_d, sizeof(double) * 17 * (size_t)nb));
        ctx->block_partials_cap = (size_t)nb * 17;
    }
    A.block_partials = ctx->block_partials_d;

    // CPointCloud::transform of the whole source (:73-74), in index order
    {
        ProfScope ps(ctx, PCR_PROF_TRANSFORM);
        hipLaunchKernelGGL(transform_kernel, dim3(min(2048u, (src->n + 255) / 256)), dim3(256), 0, ctx->stream,
                           (const Pt4 *)src->pts_d, src->n, A.M, src->tpts_d);
    }
#define LAUNCH_STEP(KC)                                                                                \
    do {                                                                                               \
        if (src->kind == PCR_INDEX_GRID)                                                               \
            hipLaunchKernelGGL(icp_step_grid_kernel<KC>, dim3(nb), dim3(256), 0, ctx->stream, A);      \
        else                                                                                           \
            hipLaunchKernelGGL(icp_step_brute_kernel<KC>, dim3(nb), dim3(256), 0, ctx->stream, A);     \
    } while (0)
    {
        ProfScope ps(ctx, PCR_PROF_STEP);
        if (k == 1)
            LAUNCH_STEP(1);
        else if (k <= 4)
            LAUNCH_STEP(4);
        else if (k <= 8)
            LAUNCH_STEP(8);
        else if (k <= 16)
            LAUNCH_STEP(16);
        else
            LAUNCH_STEP(32);
    }
#undef LAUNCH_STEP
    {
        ProfScope ps(ctx, PCR_PROF_FOLD);
        hipLaunchKernelGGL(fold_partials_kernel, dim3(17), dim3(256), 0, ctx->stream,
                           (const double *)ctx->block_partials_d, nb, partials_d);
    }
    PCR_HIP(hipGetLastError());
    return PCR_OK;
}

int pcr_icp_step(pcr_ctx *ctx, const pcr_index *src, const pcr_index *tgt, const uint32_t *sample_idx, uint32_t m,
                 const double R[9], const double T[3], uint32_t k, double thr_fwd_sq, double thr_rev_sq,
                 uint32_t flags, double partials[17], uint32_t *out_fwd, uint32_t *out_rev, uint8_t *out_keep) {
    PCR_REQUIRE(ctx && src && tgt && partials, "pcr_icp_step: NULL argument");
    if (!sample_idx) PCR_REQUIRE(m == src->n, "pcr_icp_step: sample_idx NULL means every source point: m must be %u", src->n);
    PCR_HIP(hipSetDevice(ctx->device));
    const uint32_t nout = sample_idx ? m : src->n;
    uint32_t *s_d = nullptr, *f_d = nullptr, *r_d = nullptr;
    uint8_t *k_d = nullptr;
    hipError_t e = hipSuccess;
    int rc = PCR_OK;
    if (sample_idx && m) {
        for (uint32_t i = 0; i < m; ++i)
            if (sample_idx[i] >= src->n) {
                set_last_error("pcr_icp_step: sample_idx[%u]=%u out of range", i, sample_idx[i]);
                return PCR_E_INVALID;
            }
        e = hipMalloc(&s_d, sizeof(uint32_t) * m);
        if (e == hipSuccess) e = hipMemcpyAsync(s_d, sample_idx, sizeof(uint32_t) * m, hipMemcpyHostToDevice, ctx->stream);
    }
    if (e == hipSuccess && out_fwd && nout) e = hipMalloc(&f_d, sizeof(uint32_t) * nout);
    if (e == hipSuccess && out_rev && nout) e = hipMalloc(&r_d, sizeof(uint32_t) * nout);
    if (e == hipSuccess && out_keep && nout) e = hipMalloc(&k_d, nout);
    if (e == hipSuccess) {
        rc = pcr_icp_step_d(ctx, src, tgt, s_d, 0, m, R, T, k, thr_fwd_sq, thr_rev_sq, flags, ctx->result_d, f_d, r_d,
                            k_d);
        if (rc == PCR_OK) {
            e = hipMemcpyAsync(ctx->result_h, ctx->result_d, sizeof(double) * 17, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess && f_d) e = hipMemcpyAsync(out_fwd, f_d, sizeof(uint32_t) * nout, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess && r_d) e = hipMemcpyAsync(out_rev, r_d, sizeof(uint32_t) * nout, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess && k_d) e = hipMemcpyAsync(out_keep, k_d, nout, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            if (e == hipSuccess) memcpy(partials, ctx->result_h, sizeof(double) * 17);
        }
    }
    if (s_d) hipFree(s_d);
    if (f_d) hipFree(f_d);
    if (r_d) hipFree(r_d);
    if (k_d) hipFree(k_d);
    if (rc != PCR_OK) return rc;
    PCR_HIP(e);
    return PCR_OK;
}

int pcr_reduce_pairs(pcr_ctx *ctx, const double *pair_p, const double *pair_q, uint32_t P, const double pivot_p[3],
                     const double pivot_q[3], double partials[17]) {
    PCR_REQUIRE(ctx && partials && pivot_p && pivot_q, "pcr_reduce_pairs: NULL argument");
    PCR_REQUIRE(P == 0 || (pair_p && pair_q), "pcr_reduce_pairs: NULL pair list");
    PCR_HIP(hipSetDevice(ctx->device));
    if (P == 0) {
        memset(partials, 0, sizeof(double) * 17);
        return PCR_OK;
    }
    const uint32_t nb = (P + 255) / 256;
    double *buf = nullptr;  // [3P | 3P | 17*nb | 17]
    const size_t n3 = (size_t)P * 3;
    if (hipMalloc(&buf, sizeof(double) * (2 * n3 + 17 * (size_t)nb + 17)) != hipSuccess) {
        set_last_error("pcr_reduce_pairs: device allocation for %u pairs failed", P);
        return PCR_E_NOMEM;
    }
    StepArgs A;
    memset(&A, 0, sizeof A);
    memcpy(A.pivot_p, pivot_p, sizeof A.pivot_p);
    memcpy(A.pivot_q, pivot_q, sizeof A.pivot_q);
    A.block_partials = buf + 2 * n3;
    double *res_d = A.block_partials + 17 * (size_t)nb;
    hipError_t e = hipMemcpyAsync(buf, pair_p, sizeof(double) * n3, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(buf + n3, pair_q, sizeof(double) * n3, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(reduce_pairs_kernel, dim3(nb), dim3(256), 0, ctx->stream, (const double *)buf,
                           (const double *)(buf + n3), P, A);
        hipLaunchKernelGGL(fold_partials_kernel, dim3(17), dim3(256), 0, ctx->stream, (const double *)A.block_partials, nb,
                           res_d);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(partials, res_d, sizeof(double) * 17, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    hipFree(buf);
    PCR_HIP(e);
    return PCR_OK;
}

}  // extern "C"

#ifdef PCR_DIAG
extern "C" int pcr_diag_read_icp(unsigned long long out[16], int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(pcr::g_diag), sizeof(unsigned long long) * 16) != hipSuccess) return -2;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(pcr::g_diag), z, sizeof z) != hipSuccess) return -2;
    }
    return 0;
}
#endif
