// libpcr.so -- CRejectionCentroidDist on the device (gfx950).
//
// Replaces CRejectionCentroidDist::_rejectInvalidPairsV and its helpers
// (PointCloudRegistration/RejectionCentroidDist.cpp:23-208; RegMath.h:11-22,42-55,64-67).
// The class is orphan code in the reference (compiled, never registered: SURVEY.md 0-7); its
// behaviour is reproduced INCLUDING its quirks, because "results identical to the reference" is
// the bar:
//   * "centroid" = (x + sum of neighbours) with ONLY component 0 divided, three times, by
//     (count+1); components 1 and 2 stay undivided sums                        (:103-115)
//   * the source-side query uses the TRANSFORMED source point against the index of the
//     UNTRANSFORMED source cloud                                               (:52,:75)
//   * MSD is compared with a SQUARED distance; the neighbour list is cut at the first entry with
//     d2 >= MSD (std::lower_bound on the ascending list)                       (:165-176,200-208)
// Per pair: two exact K-NN searches with K = MSN (one lane per pair, top-K in registers), the
// statistic d_i = |s_tgt - s_src|; then mean, population sigma (two fixed-order tree reductions)
// and the mask |d_i - mean| <= CDF * sigma.
#include <math.h>
#include <vector>
#include "knn_device.h"
#include "fast_search.h"

namespace pcr {

// MULTI: the index has nested levels; the single-level instance contains none of the level logic (and never indexes
// the levels' descriptions at run time, which would send the whole description to every lane's scratch memory)
template <int KC, bool MULTI>
__device__ __forceinline__ double sq_dist_to_centroid(const MultiView &g, double x, double y, double z, int k,
                                                      double msd, uint2 *rowtab) {
    TopK<KC> top;
    top.init();
    {
        const bool fin = isfinite(x) && isfinite(y) && isfinite(z);
        const double bs = 1.0 - 1.8189894035458565e-12;
        if constexpr (MULTI)  // (lists of up to 32 entries: no wave-cooperative pass, whose merge is sized for a handful)
            multi_lane_search<KC, false, false>(g, nullptr, g.all, fin, x, y, z, x, y, z, PCR_INF, bs, 0.0, top, rowtab);
        else if (fin)
            lane_search<KC>(g.v[0], g.all, x, y, z, x, y, z, PCR_INF, bs, 0.0, top, rowtab);
    }
    double c0 = x, c1 = y, c2 = z;
    int cnt = 0;
    bool open = true;  // prefix of the ascending list with d2 < MSD
#pragma unroll
    for (int j = 0; j < KC; ++j) {
        if (j < k) {
            open = open && top.idx[j] != PCR_NO_INDEX && top.d[j] < msd;
            if (open) {
                const Pt4 p = g.all[g.pos_of[top.idx[j]]];
                c0 += p.x;
                c1 += p.y;
                c2 += p.z;
                ++cnt;
            }
        }
    }
    const double den = (double)(cnt + 1);
    c0 /= den;
    c0 /= den;
    c0 /= den;
    return sqdist3(x, y, z, c0, c1, c2);
}

template <int KC, bool MS, bool MT>
__global__ void __launch_bounds__(256) reject_stat_kernel(MultiView src, MultiView tgt, const double *__restrict__ pp,
                                                          const double *__restrict__ pq, const uint32_t *__restrict__ order,
                                                          uint32_t P, int k, double msd, double *__restrict__ d_out) {
    // order (nullable): the pairs grouped by where their target point lies, so that a wave's 64 pairs search the same
    // neighbourhood (pairs arrive in sample order, which says nothing about space: unsorted, every lane of a wave
    // gathers from a different corner of both indices)
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= P) return;
    const uint32_t i = order ? order[t] : t;
    double s[2];
    __shared__ uint2 rowtab[9][256];  // per-lane row table of the ring walk (knn_device.h), nested indices only
    // target side first, as the reference does (:74-75).  Two call sites on purpose: a run-time choice between the
    // two index arguments would make the compiler copy both (1.4 KB) into every lane's scratch memory.
    uint2 *my_rows = &rowtab[0][threadIdx.x];
    s[0] = sq_dist_to_centroid<KC, MT>(tgt, pq[(size_t)i * 3], pq[(size_t)i * 3 + 1], pq[(size_t)i * 3 + 2], k, msd, my_rows);
    s[1] = sq_dist_to_centroid<KC, MS>(src, pp[(size_t)i * 3], pp[(size_t)i * 3 + 1], pp[(size_t)i * 3 + 2], k, msd, my_rows);
    d_out[i] = fabs(s[0] - s[1]);  // (:77)
}

// brute-force flavour for BRUTE indices: one lane per pair, plain global scan (small clouds only)
template <int KC>
__device__ __forceinline__ double sq_dist_to_centroid_brute(const MultiView &g, double x, double y, double z, int k,
                                                            double msd) {
    TopK<KC> top;
    top.init();
    double wd = PCR_INF;
    uint32_t wi = PCR_NO_INDEX;
    for (uint32_t j = 0; j < g.n; ++j) {
        const Pt4 c = g.all[j];
        const double d = sqdist3(x, y, z, c.x, c.y, c.z);
        if (better(d, (uint32_t)c.w, wd, wi)) {
            top.insert(d, (uint32_t)c.w);
            wd = top.worst_d();
            wi = top.worst_i();
        }
    }
    double c0 = x, c1 = y, c2 = z;
    int cnt = 0;
    bool open = true;
#pragma unroll
    for (int j = 0; j < KC; ++j) {
        if (j < k) {
            open = open && top.idx[j] != PCR_NO_INDEX && top.d[j] < msd;
            if (open) {
                const Pt4 p = g.all[g.pos_of[top.idx[j]]];
                c0 += p.x;
                c1 += p.y;
                c2 += p.z;
                ++cnt;
            }
        }
    }
    const double den = (double)(cnt + 1);
    c0 /= den;
    c0 /= den;
    c0 /= den;
    return sqdist3(x, y, z, c0, c1, c2);
}

template <int KC>
__global__ void __launch_bounds__(256) reject_stat_brute_kernel(MultiView src, MultiView tgt,
                                                                const double *__restrict__ pp,
                                                                const double *__restrict__ pq, uint32_t P, int k,
                                                                double msd, double *__restrict__ d_out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= P) return;
    const double st = sq_dist_to_centroid_brute<KC>(tgt, pq[(size_t)i * 3], pq[(size_t)i * 3 + 1], pq[(size_t)i * 3 + 2], k, msd);
    const double ss = sq_dist_to_centroid_brute<KC>(src, pp[(size_t)i * 3], pp[(size_t)i * 3 + 1], pp[(size_t)i * 3 + 2], k, msd);
    d_out[i] = fabs(st - ss);
}

// out[0] = sum_i f(d_i), f = identity (center == NULL) or (d - *center)^2; out[1] = count_slot ? P : untouched;
// out[2] = failed (1.0: this rank could not do its share -- it contributes no pair, and the flag, summed with the rest,
// tells EVERY rank so); out[3..16] = 0.  One block, fixed order.  `out` is a 17-double exchange buffer: in a sharded
// run every rank's is summed (the all-reduce the fit loop uses for its 17 partials) before finish_stat_kernel divides.
__global__ void __launch_bounds__(1024) sum_kernel(const double *__restrict__ d, uint32_t P,
                                                   const double *__restrict__ center, int count_slot, int failed,
                                                   double *__restrict__ out) {
    __shared__ double sm[1024];
    const double c = center ? *center : 0.0;
    double s = 0.0;
    for (uint32_t i = threadIdx.x; i < P; i += 1024) {
        const double v = d[i];
        s += center ? (v - c) * (v - c) : v;
    }
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sm[threadIdx.x] += sm[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sm[0];
    if (threadIdx.x == 1 && count_slot) out[1] = (double)P;
    if (threadIdx.x == 2) out[2] = failed ? 1.0 : 0.0;
    if (threadIdx.x >= 3 && threadIdx.x < 17) out[threadIdx.x] = 0.0;
}

// stat[which] = x[0] / count: the mean (:34-35), then the population variance (:36-37), over ALL ranks' pairs.
// total_out (nullable) keeps the pair count for the second call, whose exchange buffer no longer carries it;
// failed_io accumulates the ranks' failure flags of both exchanges.
__global__ void finish_stat_kernel(const double *__restrict__ x, const double *__restrict__ count, double *__restrict__ stat,
                                   int which, double *__restrict__ total_out, double *__restrict__ failed_io) {
    const double n = *count;
    stat[which] = x[0] / n;  // (no pair on any rank: NaN here, PCR_E_INVALID on the host)
    if (total_out) *total_out = n;
    *failed_io = (which == 0 ? 0.0 : *failed_io) + x[2];
}

__global__ void __launch_bounds__(256) reject_mask_kernel(const double *__restrict__ d, uint32_t P,
                                                          const double *__restrict__ stat /*mean, variance*/,
                                                          double cdf, uint8_t *__restrict__ keep) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= P) return;
    const double tol = cdf * sqrt(stat[1]);            // Tolerance = CDF * sigma (:134)
    keep[i] = fabs(d[i] - stat[0]) <= tol ? 1 : 0;     // (:137)
}

}  // namespace pcr

using namespace pcr;

static int reject_impl(pcr_ctx *ctx, const pcr_index *src, const pcr_index *tgt, const double *pair_p,
                       const double *pair_q, uint32_t P, uint32_t msn, double msd, double cdf, uint8_t *out_keep,
                       double *out_d, double *out_stat, pcr_reduce17_fn reduce, void *reduce_user) {
    PCR_REQUIRE(ctx && src && tgt && (out_keep || P == 0), "pcr_reject_centroid_dist: NULL argument");
    PCR_REQUIRE(src->ctx->device == ctx->device && tgt->ctx->device == ctx->device, "pcr_reject_centroid_dist: index lives on another device");
    PCR_REQUIRE(src->kind == tgt->kind, "pcr_reject_centroid_dist: indices must be of the same kind");
    // (a rank of a sharded run may hold no pair at all: it still takes part in both reductions)
    PCR_REQUIRE((P > 0 && pair_p && pair_q) || (P == 0 && reduce), "pcr_reject_centroid_dist: empty pair set (reference asserts, :27)");
    PCR_REQUIRE(msn >= 1 && msn <= src->n && msn <= tgt->n, "pcr_reject_centroid_dist: MSN=%u out of range", msn);
    if (msn > PCR_MAX_K) {
        set_last_error("pcr_reject_centroid_dist: MSN=%u above PCR_MAX_K=%u", msn, PCR_MAX_K);
        return PCR_E_UNSUPPORTED;
    }
    PCR_HIP(hipSetDevice(ctx->device));
    double *pp_d = nullptr, *pq_d = nullptr, *d_d = nullptr;
    uint8_t *k_d = nullptr;
    const size_t bytes = sizeof(double) * 3 * (size_t)(P ? P : 1);
    hipError_t e = hipMalloc(&pp_d, bytes);
    if (e == hipSuccess) e = hipMalloc(&pq_d, bytes);
    if (e == hipSuccess) e = hipMalloc(&d_d, sizeof(double) * (P ? P : 1));
    if (e == hipSuccess) e = hipMalloc(&k_d, P ? P : 1);
    if (e == hipSuccess && P) e = hipMemcpyAsync(pp_d, pair_p, bytes, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && P) e = hipMemcpyAsync(pq_d, pair_q, bytes, hipMemcpyHostToDevice, ctx->stream);
    // A sharded run promises every rank that the hook is entered EXACTLY TWICE on every rank (the others wait in the
    // collective otherwise).  A rank that cannot do its share -- an allocation or a copy failed -- therefore still goes
    // through both exchanges: it contributes no pair and a failure flag (slot 2), which, summed over the ranks, makes
    // EVERY rank return the same error.  Without a hook a local failure simply returns.
    const bool local_fail = e != hipSuccess;
    const hipError_t local_err = e;
    if (local_fail && !reduce) {
        if (pp_d) hipFree(pp_d);
        if (pq_d) hipFree(pq_d);
        if (d_d) hipFree(d_d);
        if (k_d) hipFree(k_d);
        PCR_HIP(local_err);
    }
    const uint32_t Pw = local_fail ? 0u : P;  // pairs this rank works on
    int reduce_rc = PCR_OK;
    {
        const uint32_t nb = (Pw + 255) / 256;
        const uint32_t *qorder = nullptr;
        if (src->kind == PCR_INDEX_GRID && Pw >= 2048) {
            uint32_t *o = (uint32_t *)ctx_scratch(ctx, 4, sizeof(uint32_t) * (size_t)Pw);
            if (o && coarse_query_order(ctx, tgt->view.v[0], pq_d, Pw, o) == PCR_OK) qorder = o;
        }
#define LAUNCH_STAT(KC)                                                                                              \
    do {                                                                                                             \
        if (src->kind == PCR_INDEX_GRID && src->view.levels > 1 && tgt->view.levels > 1)                            \
            hipLaunchKernelGGL((reject_stat_kernel<KC, true, true>), dim3(nb), dim3(256), 0, ctx->stream, src->view, \
                               tgt->view, (const double *)pp_d, (const double *)pq_d, qorder, Pw, (int)msn, msd, d_d); \
        else if (src->kind == PCR_INDEX_GRID && src->view.levels > 1)                                                \
            hipLaunchKernelGGL((reject_stat_kernel<KC, true, false>), dim3(nb), dim3(256), 0, ctx->stream, src->view, \
                               tgt->view, (const double *)pp_d, (const double *)pq_d, qorder, Pw, (int)msn, msd, d_d); \
        else if (src->kind == PCR_INDEX_GRID && tgt->view.levels > 1)                                                \
            hipLaunchKernelGGL((reject_stat_kernel<KC, false, true>), dim3(nb), dim3(256), 0, ctx->stream, src->view, \
                               tgt->view, (const double *)pp_d, (const double *)pq_d, qorder, Pw, (int)msn, msd, d_d); \
        else if (src->kind == PCR_INDEX_GRID)                                                                        \
            hipLaunchKernelGGL((reject_stat_kernel<KC, false, false>), dim3(nb), dim3(256), 0, ctx->stream, src->view, \
                               tgt->view, (const double *)pp_d, (const double *)pq_d, qorder, Pw, (int)msn, msd, d_d); \
        else                                                                                                         \
            hipLaunchKernelGGL(reject_stat_brute_kernel<KC>, dim3(nb), dim3(256), 0, ctx->stream, src->view,         \
                               tgt->view, (const double *)pp_d, (const double *)pq_d, Pw, (int)msn, msd, d_d);      \
    } while (0)
        if (Pw) {
            ProfScope ps(ctx, PCR_PROF_KNN);  // (timed with the K-NN class: two K = MSN searches per pair)
            if (msn <= 4)
                LAUNCH_STAT(4);
            else if (msn <= 8)
                LAUNCH_STAT(8);
            else if (msn <= 16)
                LAUNCH_STAT(16);
            else
                LAUNCH_STAT(32);
        }
#undef LAUNCH_STAT
        double *stat_d = ctx->result_d + 20;  // [mean, variance, total pair count, failed ranks]
        double *xbuf_d = ctx->result_d;       // [17] exchange buffer: {sum, count, failed, 0...}
        hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const double *)d_d, Pw,
                           (const double *)nullptr, 1, local_fail ? 1 : 0, xbuf_d);
        if (reduce) reduce_rc = reduce(reduce_user, xbuf_d);  // sum over ranks of {sum d, P, failed} (enqueued on the stream)
        hipLaunchKernelGGL(finish_stat_kernel, dim3(1), dim3(1), 0, ctx->stream, (const double *)xbuf_d,
                           (const double *)(xbuf_d + 1), stat_d, 0, stat_d + 2, stat_d + 3);
        hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const double *)d_d, Pw,
                           (const double *)stat_d, 0, (local_fail || reduce_rc != PCR_OK) ? 1 : 0, xbuf_d);
        if (reduce) {  // sum over ranks of sum (d - mean)^2 -- entered even if the first exchange failed HERE: the other
                       // ranks are on their way into it
            const int rc2 = reduce(reduce_user, xbuf_d);
            if (reduce_rc == PCR_OK) reduce_rc = rc2;
        }
        hipLaunchKernelGGL(finish_stat_kernel, dim3(1), dim3(1), 0, ctx->stream, (const double *)xbuf_d,
                           (const double *)(stat_d + 2), stat_d, 1, (double *)nullptr, stat_d + 3);
        if (Pw)
            hipLaunchKernelGGL(reject_mask_kernel, dim3(nb), dim3(256), 0, ctx->stream, (const double *)d_d, Pw,
                               (const double *)stat_d, cdf, k_d);
        e = hipGetLastError();
        if (e == hipSuccess && Pw) e = hipMemcpyAsync(out_keep, k_d, Pw, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess && out_d && Pw) e = hipMemcpyAsync(out_d, d_d, sizeof(double) * Pw, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess)
            e = hipMemcpyAsync(ctx->result_h + 20, stat_d, sizeof(double) * 4, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e == hipSuccess && out_stat) {
            out_stat[0] = ctx->result_h[20];
            out_stat[1] = sqrt(ctx->result_h[21]);
        }
    }
    if (pp_d) hipFree(pp_d);
    if (pq_d) hipFree(pq_d);
    if (d_d) hipFree(d_d);
    if (k_d) hipFree(k_d);
    if (local_fail) PCR_HIP(local_err);
    PCR_HIP(e);
    if (reduce_rc != PCR_OK) {
        set_last_error("pcr_reject_centroid_dist_sharded: the reduction hook failed (%d)", reduce_rc);
        return reduce_rc < 0 ? reduce_rc : PCR_E_INVALID;
    }
    if (ctx->result_h[23] > 0.0) {
        set_last_error("pcr_reject_centroid_dist_sharded: %g rank(s) could not do their share (allocation, copy or exchange "
                       "failed there); every rank returns this error", ctx->result_h[23]);
        return PCR_E_HIP;
    }
    if (!(ctx->result_h[22] >= 1.0)) {
        set_last_error("pcr_reject_centroid_dist: no pair on any rank (the reference asserts, RejectionCentroidDist.cpp:27)");
        return PCR_E_INVALID;
    }
    return PCR_OK;
}

extern "C" int pcr_reject_centroid_dist(pcr_ctx *ctx, const pcr_index *src, const pcr_index *tgt, const double *pair_p,
                                        const double *pair_q, uint32_t P, uint32_t msn, double msd, double cdf,
                                        uint8_t *out_keep, double *out_d, double *out_stat) {
    return reject_impl(ctx, src, tgt, pair_p, pair_q, P, msn, msd, cdf, out_keep, out_d, out_stat, nullptr, nullptr);
}

extern "C" int pcr_reject_centroid_dist_sharded(pcr_ctx *ctx, const pcr_index *src, const pcr_index *tgt,
                                                const double *pair_p, const double *pair_q, uint32_t P, uint32_t msn,
                                                double msd, double cdf, uint8_t *out_keep, double *out_d,
                                                double *out_stat, pcr_reduce17_fn reduce, void *reduce_user) {
    PCR_REQUIRE(reduce, "pcr_reject_centroid_dist_sharded: NULL reduction hook");
    return reject_impl(ctx, src, tgt, pair_p, pair_q, P, msn, msd, cdf, out_keep, out_d, out_stat, reduce, reduce_user);
}
