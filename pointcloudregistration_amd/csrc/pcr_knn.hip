// libpcr.so -- batched exact K-NN entry points (gfx950).
//
// Replaces hiveCommon::CKNNSearch::executeKNN (Common/KNNSearch.cpp:65-75, one query per call,
// serial tree descent) and hiveSearch::CKNNSearch::_executeSearchV (NewKNNSearch.cpp:32-54,
// OpenMP over queries) with one launch over all queries: one lane per query walking the
// uniform grid, or -- for small clouds -- a tiled brute-force scan with LDS-staged tiles.
#include <math.h>
#include <vector>
#include "knn_device.h"
#include "fast_search.h"
#ifndef PCR_KNN_U
#define PCR_KNN_U 6  // shadow records per trip of the free-query search
#endif

namespace pcr {

template <int KC>
__device__ __forceinline__ void write_result(const TopK<KC> &top, int k, int order, double max_d2, uint32_t *oi,
                                             double *od) {
#pragma unroll
    for (int j = 0; j < KC; ++j) {
        if (j < k) {
            const bool ok = top.d[j] < max_d2;
            const int o = order == PCR_ORDER_DESCENDING ? k - 1 - j : j;
            oi[o] = ok ? top.idx[j] : PCR_NO_INDEX;
            if (od) od[o] = ok ? top.d[j] : PCR_INF;
        }
    }
}

// One lane per query, pruned ring walk (knn_device.h).
// order (nullable): processing order of the queries (by grid brick), results go to the query's own row.
template <int KC, bool MULTI>
__global__ void __launch_bounds__(256) knn_grid_kernel(MultiView g, const MultiView *__restrict__ g_mem, const double *__restrict__ q,
                                                       const uint32_t *__restrict__ order, uint32_t nq, int k,
                                                       int order_out, double max_d2, uint32_t *__restrict__ out_idx,
                                                       double *__restrict__ out_d2) {
    const uint32_t t = (MULTI ? xcd_remap_chunked(blockIdx.x, gridDim.x) : xcd_remap(blockIdx.x, gridDim.x)) * 256u + threadIdx.x;
    const bool live = t < nq;
    const uint32_t i = live ? (order ? order[t] : t) : 0u;
    double qx = 0, qy = 0, qz = 0;
    if (live) {
        qx = q[(size_t)i * 3];
        qy = q[(size_t)i * 3 + 1];
        qz = q[(size_t)i * 3 + 2];
    }
    const bool active = live && isfinite(qx) && isfinite(qy) && isfinite(qz);
    TopK<KC> top;
    top.init(max_d2);
    __shared__ uint2 rowtab[9][256];
    if constexpr (MULTI) {  // (every lane calls: see multi_lane_search)
        multi_lane_search<KC, false, (KC <= 8)>(g, g_mem, g.all, active, qx, qy, qz, qx, qy, qz, max_d2, 1.0 - 1.8189894035458565e-12, 0.0, top,
                              &rowtab[0][threadIdx.x]);
    } else if (active) {
        lane_search<KC>(g.v[0], g.all, qx, qy, qz, qx, qy, qz, max_d2, 1.0 - 1.8189894035458565e-12, 0.0, top,
                        &rowtab[0][threadIdx.x]);
    }
    if (live)
        write_result<KC>(top, k, order_out, max_d2, out_idx + (size_t)i * k, out_d2 ? out_d2 + (size_t)i * k : nullptr);
}

// The same on a single-level grid that carries fp32 shadows: the filtered walk (fast_search.h: fp32 keys select, fp64
// decides and verifies) and, for the rare lane it cannot vouch for, the fp64 walk above.  The grid's description is
// read from device memory through the constant address space (scalar loads at the point of use).
#define PCR_CONST_AS __attribute__((address_space(4)))
#ifndef PCR_KNN_BLOCK
#define PCR_KNN_BLOCK 256  // lanes per block of the filtered K-NN kernel (1 M queries, K = 1 / 4, e9 queries/s: 64: 8.4 / 6.8, 256: 9.1 / 7.3)
#endif
template <int KC>
__global__ void __launch_bounds__(PCR_KNN_BLOCK, 4) knn_fast_kernel(const MultiView *__restrict__ gp, const double *__restrict__ q,
                                                          const uint32_t *__restrict__ order, uint32_t nq, int k, int order_out,
                                                          double max_d2, uint32_t *__restrict__ out_idx,
                                                          double *__restrict__ out_d2) {
    const uint32_t t = xcd_remap(blockIdx.x, gridDim.x) * (uint32_t)PCR_KNN_BLOCK + threadIdx.x;
    const bool live = t < nq;
    const uint32_t i = live ? (order ? order[t] : t) : 0u;
    double qx = 0, qy = 0, qz = 0;
    if (live) {
        qx = q[(size_t)i * 3];
        qy = q[(size_t)i * 3 + 1];
        qz = q[(size_t)i * 3 + 2];
    }
    const bool active = live && isfinite(qx) && isfinite(qy) && isfinite(qz);
    __shared__ uint2 rowtab[9][PCR_KNN_BLOCK];
    uint2 *my_rows = &rowtab[0][threadIdx.x];
    const PCR_CONST_AS GridView &G = ((const PCR_CONST_AS MultiView *)(uintptr_t)gp)->v[0];
    const double scale = 1.0 - 1.8189894035458565e-12;
    FastExact<KC> ex;
    double dk;
    uint32_t ik;
    bool done = !active;
    if (active)
        done = fast_ring1<KC, false, false, 3, PCR_KNN_U, PCR_KNN_BLOCK>(G, G.pts, qx, qy, qz, qx, qy, qz, max_d2, PCR_INF, scale, 0.0,
                                                                    G.key_delta, k, 0.0, 0u, dk, ik, my_rows, (const Rigid *)nullptr, &ex);
    uint32_t *oi = out_idx + (size_t)i * k;
    double *od = out_d2 ? out_d2 + (size_t)i * k : nullptr;
    if (active & done) {  // rank r of the exact candidates goes to slot r (or k-1-r): the first k ranks fill the row
#pragma unroll
        for (int j = 0; j < KC + 1; ++j) {
            if (j < ex.n && ex.rank[j] < (uint32_t)k) {
                const uint32_t o = order_out == PCR_ORDER_DESCENDING ? (uint32_t)k - 1u - ex.rank[j] : ex.rank[j];
                oi[o] = ex.i[j];  // (max_d2, PCR_NO_INDEX) where fewer than k candidates lie below max_d2
                if (od) od[o] = ex.i[j] == PCR_NO_INDEX ? PCR_INF : ex.d[j];
            }
        }
    }
    if (__any(live & !(active & done))) {
        if (live & !(active & done)) {
            TopK<KC> top;
            top.init(max_d2);
            if (active) {
                const GridView Gc = gp->v[0];
                lane_search<KC, false, PCR_KNN_BLOCK>(Gc, gp->all, qx, qy, qz, qx, qy, qz, max_d2, scale, 0.0, top, my_rows);
            }
            write_result<KC>(top, k, order_out, max_d2, oi, od);
        }
    }
}

constexpr int BRUTE_TILE = 512;  // points per LDS tile: 16 KiB

template <int KC>
__global__ void __launch_bounds__(256) knn_brute_kernel(const Pt4 *__restrict__ pts, uint32_t n,
                                                        const double *__restrict__ q, uint32_t nq, int k, int order,
                                                        double max_d2, uint32_t *__restrict__ out_idx,
                                                        double *__restrict__ out_d2) {
    __shared__ Pt4 tile[BRUTE_TILE];
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const bool live = i < nq;
    double qx = 0, qy = 0, qz = 0;
    if (live) {
        qx = q[(size_t)i * 3];
        qy = q[(size_t)i * 3 + 1];
        qz = q[(size_t)i * 3 + 2];
    }
    const bool fin = isfinite(qx) && isfinite(qy) && isfinite(qz);
    TopK<KC> top;
    top.init(max_d2);
    for (uint32_t base = 0; base < n; base += BRUTE_TILE) {
        const int cnt = (int)min((uint32_t)BRUTE_TILE, n - base);
        for (int t = threadIdx.x; t < cnt; t += 256) tile[t] = pts[base + t];  // coalesced 32-B rows
        __syncthreads();
        if (live && fin) tile_scan<KC>(tile, cnt, qx, qy, qz, top);
        __syncthreads();
    }
    if (live)
        write_result<KC>(top, k, order, max_d2, out_idx + (size_t)i * k, out_d2 ? out_d2 + (size_t)i * k : nullptr);
}

__global__ void gather_rows_kernel(const double *__restrict__ xyz, uint32_t first, uint32_t stride, uint32_t count,
                                   double *__restrict__ out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < count * 3) out[t] = xyz[(size_t)(first + (t / 3) * stride) * 3 + t % 3];
}

template <int KC>
static void launch_knn(pcr_ctx *ctx, const pcr_index *ix, const double *q_d, const uint32_t *qorder, uint32_t nq,
                       uint32_t k, int order, double max_d2, uint32_t *oi_d, double *od_d) {
    const uint32_t nb = (nq + 255) / 256;
    if constexpr (KC <= 8) {
        if (ix->kind == PCR_INDEX_GRID && ix->view.levels == 1 && ix->view.v[0].shadow && ix->view_d && g_tune[5] != 1) {
            hipLaunchKernelGGL(knn_fast_kernel<KC>, dim3((nq + PCR_KNN_BLOCK - 1) / PCR_KNN_BLOCK), dim3(PCR_KNN_BLOCK), 0, ctx->stream,
                               (const MultiView *)ix->view_d, q_d, qorder, nq, (int)k, order, max_d2, oi_d, od_d);
            return;
        }
    }
    if (ix->kind == PCR_INDEX_GRID && ix->view.levels > 1)
        hipLaunchKernelGGL((knn_grid_kernel<KC, true>), dim3(nb), dim3(256), 0, ctx->stream, ix->view, (const MultiView *)ix->view_d, q_d, qorder, nq,
                           (int)k, order, max_d2, oi_d, od_d);
    else if (ix->kind == PCR_INDEX_GRID)
        hipLaunchKernelGGL((knn_grid_kernel<KC, false>), dim3(nb), dim3(256), 0, ctx->stream, ix->view, (const MultiView *)ix->view_d, q_d, qorder, nq,
                           (int)k, order, max_d2, oi_d, od_d);
    else
        hipLaunchKernelGGL(knn_brute_kernel<KC>, dim3(nb), dim3(256), 0, ctx->stream, ix->pts_d, ix->n, q_d, nq,
                           (int)k, order, max_d2, oi_d, od_d);
}

}  // namespace pcr

using namespace pcr;

extern "C" {

int pcr_knn_d(pcr_ctx *ctx, const pcr_index *ix, const double *q_d, uint32_t nq, uint32_t k, int order, double max_d2,
              uint32_t *out_idx_d, double *out_d2_d) {
    PCR_REQUIRE(ctx && ix && out_idx_d, "pcr_knn_d: NULL argument");
    PCR_REQUIRE(ix->ctx->device == ctx->device, "pcr_knn_d: index lives on another device");
    PCR_REQUIRE(k >= 1, "pcr_knn_d: k must be >= 1");
    PCR_REQUIRE(k <= ix->n, "pcr_knn_d: k=%u exceeds the %u indexed points (Common/KNNSearch.cpp:19)", k, ix->n);
    if (k > PCR_MAX_K) {
        set_last_error("pcr_knn_d: k=%u above PCR_MAX_K=%u", k, PCR_MAX_K);
        return PCR_E_UNSUPPORTED;
    }
    PCR_REQUIRE(order == PCR_ORDER_ASCENDING || order == PCR_ORDER_DESCENDING, "pcr_knn_d: bad order %d", order);
    if (nq == 0) return PCR_OK;
    PCR_REQUIRE(q_d, "pcr_knn_d: q_d is NULL");
    PCR_HIP(hipSetDevice(ctx->device));
    if (!(max_d2 >= 0) || isinf(max_d2)) max_d2 = PCR_INF_H;  // negative / NaN / inf: unbounded
    // Order the queries by grid brick so that each wave's 64 queries share a small box (the callers'
    // query order is arbitrary); a handful of queries is not worth the extra launches.
    const uint32_t *qorder = nullptr;
    if (ix->kind == PCR_INDEX_GRID && nq >= 2048 && g_tune[3] == 0) {
        uint32_t *o = (uint32_t *)ctx_scratch(ctx, 4, sizeof(uint32_t) * (size_t)nq);
        if (!o) return PCR_E_NOMEM;
        int rc = coarse_query_order(ctx, ix->view.v[0], q_d, nq, o);
        if (rc != PCR_OK) return rc;
        qorder = o;
    }
    ProfScope ps(ctx, PCR_PROF_KNN);
    if (k == 1)
        launch_knn<1>(ctx, ix, q_d, qorder, nq, k, order, max_d2, out_idx_d, out_d2_d);
    else if (k <= 4)
        launch_knn<4>(ctx, ix, q_d, qorder, nq, k, order, max_d2, out_idx_d, out_d2_d);
    else if (k <= 8)
        launch_knn<8>(ctx, ix, q_d, qorder, nq, k, order, max_d2, out_idx_d, out_d2_d);
    else if (k <= 16)
        launch_knn<16>(ctx, ix, q_d, qorder, nq, k, order, max_d2, out_idx_d, out_d2_d);
    else
        launch_knn<32>(ctx, ix, q_d, qorder, nq, k, order, max_d2, out_idx_d, out_d2_d);
    PCR_HIP(hipGetLastError());
    return PCR_OK;
}

int pcr_knn(pcr_ctx *ctx, const pcr_index *ix, const double *q, uint32_t nq, uint32_t k, int order, double max_d2,
            uint32_t *out_idx, double *out_d2) {
    PCR_REQUIRE(ctx && ix && out_idx, "pcr_knn: NULL argument");
    if (nq == 0) return pcr_knn_d(ctx, ix, nullptr, 0, k, order, max_d2, out_idx, nullptr);
    PCR_REQUIRE(q, "pcr_knn: q is NULL");
    PCR_HIP(hipSetDevice(ctx->device));
    double *q_d = nullptr, *od_d = nullptr;
    uint32_t *oi_d = nullptr;
    int rc = PCR_OK;
    hipError_t e = hipMalloc(&q_d, sizeof(double) * 3 * (size_t)nq);
    if (e == hipSuccess) e = hipMalloc(&oi_d, sizeof(uint32_t) * (size_t)nq * k);
    if (e == hipSuccess && out_d2) e = hipMalloc(&od_d, sizeof(double) * (size_t)nq * k);
    if (e == hipSuccess) e = hipMemcpyAsync(q_d, q, sizeof(double) * 3 * (size_t)nq, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        rc = pcr_knn_d(ctx, ix, q_d, nq, k, order, max_d2, oi_d, od_d);
        if (rc == PCR_OK) {
            e = hipMemcpyAsync(out_idx, oi_d, sizeof(uint32_t) * (size_t)nq * k, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess && out_d2)
                e = hipMemcpyAsync(out_d2, od_d, sizeof(double) * (size_t)nq * k, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        }
    }
    if (q_d) hipFree(q_d);
    if (oi_d) hipFree(oi_d);
    if (od_d) hipFree(od_d);
    if (rc != PCR_OK) return rc;
    PCR_HIP(e);
    return PCR_OK;
}

int pcr_unique_sq_dist(pcr_ctx *ctx, const pcr_index *ix, double *out) {
    PCR_REQUIRE(ctx && ix && out, "pcr_unique_sq_dist: NULL argument");
    PCR_REQUIRE(ix->n >= 4, "pcr_unique_sq_dist: needs >= 4 points (K=4, UniqueData.cpp:34)");
    PCR_HIP(hipSetDevice(ctx->device));
    // UniqueData.cpp:29-31,40-50: SampleDist = max(1, N/20); SampleItr = 0, SampleDist, ... < N
    uint32_t stride = ix->n / 20;
    if (stride < 1) stride = 1;
    const uint32_t count = (ix->n + stride - 1) / stride;
    double *q_d = nullptr, *d_d = nullptr;
    uint32_t *i_d = nullptr;
    hipError_t e = hipMalloc(&q_d, sizeof(double) * 3 * count);
    if (e == hipSuccess) e = hipMalloc(&d_d, sizeof(double) * 4 * count);
    if (e == hipSuccess) e = hipMalloc(&i_d, sizeof(uint32_t) * 4 * count);
    int rc = PCR_OK;
    std::vector<double> d2((size_t)count * 4);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(gather_rows_kernel, dim3((count * 3 + 255) / 256), dim3(256), 0, ctx->stream,
                           ix->cloud->xyz_d, 0u, stride, count, q_d);
        rc = pcr_knn_d(ctx, ix, q_d, count, 4, PCR_ORDER_ASCENDING, -1.0, i_d, d_d);
        if (rc == PCR_OK) {
            e = hipMemcpyAsync(d2.data(), d_d, sizeof(double) * 4 * count, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        }
    }
    if (q_d) hipFree(q_d);
    if (d_d) hipFree(d_d);
    if (i_d) hipFree(i_d);
    if (rc != PCR_OK) return rc;
    PCR_HIP(e);
    double sum = 0;  // same order as the reference: sample by sample, neighbour 0..3 (:43-46)
    for (uint32_t s = 0; s < count; ++s)
        for (int j = 0; j < 4; ++j) sum += d2[(size_t)s * 4 + j];
    *out = sum / (count * 4);
    return PCR_OK;
}

}  // extern "C"

#ifdef PCR_DIAG
extern "C" int pcr_diag_read_knn(unsigned long long out[48], int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(pcr::g_diag), sizeof(unsigned long long) * 48) != hipSuccess) return -2;
    if (reset) {
        unsigned long long z[48] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(pcr::g_diag), z, sizeof z) != hipSuccess) return -2;
    }
    return 0;
}
#endif
