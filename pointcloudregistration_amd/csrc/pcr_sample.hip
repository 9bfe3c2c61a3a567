// libpcr.so -- samplers: the step before the loop.
//
//   pcr_sample_random  CSamplerRandom::_doSampleV (Sampler/SamplerRandom.cpp:25-51): keep point i iff u_i < ratio.
//                      The reference draws u_i from an unseeded rand() (and overflows RAND_MAX+1 on glibc); here
//                      u_i is a counter-based hash of (seed, i), so a sample set is a pure function of its seed --
//                      reproducible, order-independent, and parallel.  Same distribution, not the same stream:
//                      there is no stream to reproduce ("parity unpinned" by construction of the reference).
//   pcr_sample_random_draws  the same rule on a stream of rand() values the caller drew: given the stream, the sample set
//                      is the reference's index for index (checked against the oracle's restatement).
//   pcr_sample_grid    the uniform-grid sampler the reference only stubs (Sampler/SamplerUniformGrid.cpp:25-28):
//                      one point per occupied cell of the index's grid -- the cell's first point in layout order
//                      (smallest x, ties by index), free once the index exists.
// Both return ascending source indices (sample order = summation order downstream).
#include <algorithm>
#include <vector>
#include "pcr_internal.h"

namespace pcr {

__device__ __forceinline__ double hash_uniform(uint64_t seed, uint64_t i) {  // splitmix64 -> [0,1)
    uint64_t z = seed + 0x9e3779b97f4a7c15ull * (i + 1);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    z ^= z >> 31;
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

struct RandomPick {
    uint64_t seed;
    double ratio;
    __device__ bool keep(uint64_t i) const { return hash_uniform(seed, i) < ratio; }
    __device__ uint32_t value(uint64_t i) const { return (uint32_t)i; }
};
// CSamplerRandom's rule on a stream of rand() values the CALLER drew (ICPMacros.h:4: (double)rand() / (RAND_MAX + 1)):
// with the stream given, the sample set is the reference's, index for index
struct DrawPick {
    const int32_t *draws;  // device, [n]
    double denom;          // RAND_MAX + 1
    double ratio;
    __device__ bool keep(uint64_t i) const { return 0.0 + (double)draws[i] / denom * (1.0 - 0.0) < ratio; }
    __device__ uint32_t value(uint64_t i) const { return (uint32_t)i; }
};
struct CellPick {
    const uint32_t *cs;
    const Pt4 *pts;
    __device__ bool keep(uint64_t c) const { return cs[c + 1] > cs[c]; }
    __device__ uint32_t value(uint64_t c) const { return (uint32_t)pts[cs[c]].w; }
};

// the same over a level's COMPACT cell table (GridView::cstart): every entry is an occupied cell, in cell order
struct OccupiedPick {
    const uint32_t *cstart;
    const Pt4 *pts;
    __device__ bool keep(uint64_t) const { return true; }
    __device__ uint32_t value(uint64_t j) const { return (uint32_t)pts[cstart[j]].w; }
};

constexpr int PICK_TILE = 1024;  // items per block (4 per thread)

template <typename P>
__global__ void __launch_bounds__(256) pick_count_kernel(P pick, uint64_t n, uint32_t *__restrict__ block_count) {
    __shared__ uint32_t wave_sum[4];
    const uint64_t base = (uint64_t)blockIdx.x * PICK_TILE;
    uint32_t c = 0;
    for (int u = 0; u < 4; ++u) {
        const uint64_t i = base + (uint64_t)u * 256 + threadIdx.x;
        c += (i < n && pick.keep(i)) ? 1u : 0u;
    }
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_count[blockIdx.x] = wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
}

// stable within the block: item order is (u, thread), ranks by ballot prefix
template <typename P>
__global__ void __launch_bounds__(256) pick_scatter_kernel(P pick, uint64_t n, const uint32_t *__restrict__ block_offset,
                                                           uint32_t *__restrict__ out) {
    __shared__ uint32_t wave_cnt[4];
    const uint64_t base = (uint64_t)blockIdx.x * PICK_TILE;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t run = block_offset[blockIdx.x];
    for (int u = 0; u < 4; ++u) {
        const uint64_t i = base + (uint64_t)u * 256 + threadIdx.x;
        const bool k = i < n && pick.keep(i);
        const unsigned long long m = __ballot(k);
        if (lane == 0) wave_cnt[wave] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t before = 0;
        for (int w = 0; w < wave; ++w) before += wave_cnt[w];
        const uint32_t total = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        if (k) out[run + before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = pick.value(i);
        run += total;
        __syncthreads();
    }
}

template <typename P>
static int pick_compact(pcr_ctx *ctx, P pick, uint64_t n, uint32_t cap, uint32_t *out_idx, uint32_t *out_count, bool sort_result) {
    *out_count = 0;
    if (n == 0) return PCR_OK;
    const uint64_t blocks64 = (n + PICK_TILE - 1) / PICK_TILE;
    PCR_REQUIRE(blocks64 <= 0x7fffffffull, "sampler: too many items");
    const uint32_t blocks = (uint32_t)blocks64;
    uint32_t *cnt_d = nullptr, *out_d = nullptr;
    PCR_HIP(hipMalloc(&cnt_d, sizeof(uint32_t) * blocks));
    hipLaunchKernelGGL(pick_count_kernel<P>, dim3(blocks), dim3(256), 0, ctx->stream, pick, n, cnt_d);
    std::vector<uint32_t> cnt(blocks);
    hipError_t e = hipMemcpyAsync(cnt.data(), cnt_d, sizeof(uint32_t) * blocks, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    uint64_t total = 0;
    for (uint32_t b = 0; b < blocks && e == hipSuccess; ++b) {
        const uint32_t c = cnt[b];
        cnt[b] = (uint32_t)total;
        total += c;
    }
    int rc = PCR_OK;
    if (e == hipSuccess && total > cap) {
        set_last_error("sampler: %llu samples do not fit the caller's %u slots", (unsigned long long)total, cap);
        rc = PCR_E_INVALID;
    }
    if (e == hipSuccess && rc == PCR_OK && total) {
        e = hipMalloc(&out_d, sizeof(uint32_t) * total);
        if (e == hipSuccess) e = hipMemcpyAsync(cnt_d, cnt.data(), sizeof(uint32_t) * blocks, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(pick_scatter_kernel<P>, dim3(blocks), dim3(256), 0, ctx->stream, pick, n, (const uint32_t *)cnt_d, out_d);
            e = hipMemcpyAsync(out_idx, out_d, sizeof(uint32_t) * total, hipMemcpyDeviceToHost, ctx->stream);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    }
    hipFree(cnt_d);
    if (out_d) hipFree(out_d);
    if (rc != PCR_OK) return rc;
    PCR_HIP(e);
    *out_count = (uint32_t)total;
    if (sort_result) std::sort(out_idx, out_idx + total);
    return PCR_OK;
}

}  // namespace pcr

using namespace pcr;

extern "C" {

int pcr_sample_random(pcr_ctx *ctx, uint32_t n, double ratio, uint64_t seed, uint32_t cap, uint32_t *out_idx, uint32_t *out_count) {
    PCR_REQUIRE(ctx && out_idx && out_count, "pcr_sample_random: NULL argument");
    PCR_REQUIRE(ratio >= 0.0 && ratio <= 1.0, "pcr_sample_random: ratio %g outside [0,1]", ratio);
    PCR_HIP(hipSetDevice(ctx->device));
    return pick_compact(ctx, RandomPick{seed, ratio}, n, cap, out_idx, out_count, false);
}

int pcr_sample_random_draws(pcr_ctx *ctx, uint32_t n, double ratio, const int32_t *draws, uint32_t rand_max, uint32_t cap,
                            uint32_t *out_idx, uint32_t *out_count) {
    PCR_REQUIRE(ctx && out_idx && out_count && (draws || n == 0), "pcr_sample_random_draws: NULL argument");
    PCR_REQUIRE(ratio >= 0.0 && ratio <= 1.0, "pcr_sample_random_draws: ratio %g outside [0,1]", ratio);
    PCR_REQUIRE(rand_max >= 1, "pcr_sample_random_draws: rand_max must be RAND_MAX of the generator that drew the stream");
    PCR_HIP(hipSetDevice(ctx->device));
    *out_count = 0;
    if (n == 0) return PCR_OK;
    int32_t *d = nullptr;
    PCR_HIP(hipMalloc(&d, sizeof(int32_t) * (size_t)n));
    hipError_t e = hipMemcpyAsync(d, draws, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, ctx->stream);
    int rc = PCR_OK;
    if (e == hipSuccess) rc = pick_compact(ctx, DrawPick{d, (double)rand_max + 1.0, ratio}, n, cap, out_idx, out_count, false);
    hipFree(d);
    PCR_HIP(e);
    return rc;
}

int pcr_sample_grid(pcr_ctx *ctx, const pcr_index *index, uint32_t cap, uint32_t *out_idx, uint32_t *out_count) {
    PCR_REQUIRE(ctx && index && out_idx && out_count, "pcr_sample_grid: NULL argument");
    PCR_REQUIRE(index->ctx->device == ctx->device, "pcr_sample_grid: index lives on another device");
    PCR_REQUIRE(index->kind == PCR_INDEX_GRID, "pcr_sample_grid: needs a grid index");
    PCR_HIP(hipSetDevice(ctx->device));
    // every level of the index has cells of its own; the levels partition the cloud, so the picks are distinct
    uint32_t total = 0;
    for (int l = 0; l < index->view.levels; ++l) {
        uint32_t got = 0;
        int rc = index->cstart_d[l]
                     ? pick_compact(ctx, OccupiedPick{index->cstart_d[l], index->view.v[l].pts}, index->occupied[l], cap - total,
                                    out_idx + total, &got, false)
                     : pick_compact(ctx, CellPick{index->cs_d[l], index->view.v[l].pts}, index->cells[l], cap - total, out_idx + total,
                                    &got, false);
        if (rc != PCR_OK) return rc;
        total += got;
    }
    std::sort(out_idx, out_idx + total);
    *out_count = total;
    return PCR_OK;
}

}  // extern "C"
