// libpcr.so -- index build, second part: x-order inside every grid cell (gfx950).
//
// A uniform grid meets clouds whose density varies by orders of magnitude (LiDAR sweeps: density ~ 1/r^2 from the
// sensor): a few cells then hold thousands of points and a per-cell scan degenerates.
// The search therefore needs a second criterion INSIDE a heavy cell.  Every cell's points are kept sorted by x
// (ties by original index).  Cells are x-intervals, and the three x-adjacent cells of a (y,z) row are stored back
// to back, so a whole row range is sorted by x as well: the search can enter a long row at the query's x by
// binary search and walk outwards until the x-gap alone exceeds the current K-th distance (knn_device.h,
// windowed_scan).  Small cells are ordered by one thread each (cell_order_kernel, pcr_core.hip); cells above
// PCR_HEAVY_CELL points are ordered here by two stable segmented radix sorts (by index, then by x) -- rocPRIM's
// segmented sort is the one library primitive of the build path; the hot loop uses none.
#include <cstring>
#include <rocprim/device/device_segmented_radix_sort.hpp>
#include "pcr_internal.h"

namespace pcr {

__global__ void __launch_bounds__(256) heavy_list_kernel(const uint32_t *__restrict__ cs, uint64_t cells, uint32_t cap,
                                                         uint32_t *__restrict__ seg_begin, uint32_t *__restrict__ seg_end,
                                                         uint32_t *__restrict__ counter) {
    const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cells) return;
    const uint32_t b = cs[c], e = cs[c + 1];
    if (e - b <= PCR_HEAVY_CELL) return;
    const uint32_t slot = atomicAdd(counter, 1u);
    if (slot < cap) {
        seg_begin[slot] = b;
        seg_end[slot] = e;
    }
}

// one block per segment (grid-stride), threads stride over the segment
template <typename F>
__global__ void __launch_bounds__(256) for_segments_kernel(const uint32_t *__restrict__ seg_begin,
                                                           const uint32_t *__restrict__ seg_end, uint32_t segments, F f) {
    for (uint32_t s = blockIdx.x; s < segments; s += gridDim.x) {
        const uint32_t b = seg_begin[s], e = seg_end[s];
        for (uint32_t i = b + threadIdx.x; i < e; i += 256) f(i);
    }
}

struct InitIdx {
    const Pt4 *pts;
    uint32_t *key, *val;
    __device__ void operator()(uint32_t i) const {
        key[i] = (uint32_t)pts[i].w;
        val[i] = i;
    }
};
struct GatherX {
    const Pt4 *pts;
    const uint32_t *perm;
    double *key;
    __device__ void operator()(uint32_t i) const { key[i] = pts[perm[i]].x; }
};
struct GatherPts {
    const Pt4 *pts;
    const uint32_t *perm;
    Pt4 *out;
    __device__ void operator()(uint32_t i) const { out[i] = pts[perm[i]]; }
};
struct CopyPts {
    const Pt4 *in;
    Pt4 *out;
    __device__ void operator()(uint32_t i) const { out[i] = in[i]; }
};

#define SORT_HIP(call)                                                                                         \
    do {                                                                                                       \
        hipError_t e__ = (call);                                                                               \
        if (e__ != hipSuccess) {                                                                               \
            set_last_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__);       \
            rc = e__ == hipErrorOutOfMemory ? PCR_E_NOMEM : PCR_E_HIP;                                         \
            goto done;                                                                                         \
        }                                                                                                      \
    } while (0)

// pts: [n] in cell order; tmp: [n] scratch of the same size (the index's transformed-copy buffer is free during
// the build).  Synchronous.
int sort_heavy_cells(pcr_ctx *ctx, const uint32_t *cs_d, uint64_t cells, uint32_t n, Pt4 *pts, Pt4 *tmp) {
    int rc = PCR_OK;
    const uint32_t cap = n / (PCR_HEAVY_CELL + 1) + 1;
    uint32_t *seg_begin = nullptr, *seg_end = nullptr, *counter = nullptr;
    uint32_t *key_i = nullptr, *key_i2 = nullptr, *val_a = nullptr, *val_b = nullptr, *val_c = nullptr;
    double *key_x = nullptr, *key_x2 = nullptr;
    void *temp = nullptr;
    size_t temp_bytes = 0, need = 0;
    uint32_t segments = 0;
    hipStream_t st = ctx->stream;
    SORT_HIP(hipMalloc(&seg_begin, sizeof(uint32_t) * cap));
    SORT_HIP(hipMalloc(&seg_end, sizeof(uint32_t) * cap));
    SORT_HIP(hipMalloc(&counter, sizeof(uint32_t)));
    SORT_HIP(hipMemsetAsync(counter, 0, sizeof(uint32_t), st));
    hipLaunchKernelGGL(heavy_list_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, st, cs_d, cells, cap, seg_begin,
                       seg_end, counter);
    SORT_HIP(hipMemcpyAsync(&segments, counter, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    SORT_HIP(hipStreamSynchronize(st));
    if (segments == 0) goto done;
    if (segments > cap) segments = cap;  // cannot happen: every heavy cell holds more than PCR_HEAVY_CELL points
    SORT_HIP(hipMalloc(&key_i, sizeof(uint32_t) * (size_t)n));
    SORT_HIP(hipMalloc(&key_i2, sizeof(uint32_t) * (size_t)n));
    SORT_HIP(hipMalloc(&val_a, sizeof(uint32_t) * (size_t)n));
    SORT_HIP(hipMalloc(&val_b, sizeof(uint32_t) * (size_t)n));
    SORT_HIP(hipMalloc(&val_c, sizeof(uint32_t) * (size_t)n));
    SORT_HIP(hipMalloc(&key_x, sizeof(double) * (size_t)n));
    SORT_HIP(hipMalloc(&key_x2, sizeof(double) * (size_t)n));
    {
        const unsigned grid = segments < 65535u ? segments : 65535u;
        hipLaunchKernelGGL(for_segments_kernel<InitIdx>, dim3(grid), dim3(256), 0, st, (const uint32_t *)seg_begin,
                           (const uint32_t *)seg_end, segments, InitIdx{pts, key_i, val_a});
        // pass 1: by original index (makes the layout independent of the scatter's atomic order)
        SORT_HIP(rocprim::segmented_radix_sort_pairs(nullptr, need, key_i, key_i2, val_a, val_b, n, segments, seg_begin, seg_end, 0,
                                                     32, st));
        temp_bytes = need;
        SORT_HIP(rocprim::segmented_radix_sort_pairs(nullptr, need, key_x, key_x2, val_b, val_c, n, segments, seg_begin, seg_end, 0,
                                                     64, st));
        if (need > temp_bytes) temp_bytes = need;
        SORT_HIP(hipMalloc(&temp, temp_bytes ? temp_bytes : 16));
        need = temp_bytes;
        SORT_HIP(rocprim::segmented_radix_sort_pairs(temp, need, key_i, key_i2, val_a, val_b, n, segments, seg_begin, seg_end, 0, 32,
                                                     st));
        // pass 2 (stable): by x
        hipLaunchKernelGGL(for_segments_kernel<GatherX>, dim3(grid), dim3(256), 0, st, (const uint32_t *)seg_begin,
                           (const uint32_t *)seg_end, segments, GatherX{pts, val_b, key_x});
        need = temp_bytes;
        SORT_HIP(rocprim::segmented_radix_sort_pairs(temp, need, key_x, key_x2, val_b, val_c, n, segments, seg_begin, seg_end, 0, 64,
                                                     st));
        hipLaunchKernelGGL(for_segments_kernel<GatherPts>, dim3(grid), dim3(256), 0, st, (const uint32_t *)seg_begin,
                           (const uint32_t *)seg_end, segments, GatherPts{pts, val_c, tmp});
        hipLaunchKernelGGL(for_segments_kernel<CopyPts>, dim3(grid), dim3(256), 0, st, (const uint32_t *)seg_begin,
                           (const uint32_t *)seg_end, segments, CopyPts{tmp, pts});
        SORT_HIP(hipGetLastError());
        SORT_HIP(hipStreamSynchronize(st));
    }
done:
    if (seg_begin) hipFree(seg_begin);
    if (seg_end) hipFree(seg_end);
    if (counter) hipFree(counter);
    if (key_i) hipFree(key_i);
    if (key_i2) hipFree(key_i2);
    if (val_a) hipFree(val_a);
    if (val_b) hipFree(val_b);
    if (val_c) hipFree(val_c);
    if (key_x) hipFree(key_x);
    if (key_x2) hipFree(key_x2);
    if (temp) hipFree(temp);
    return rc;
}

}  // namespace pcr
