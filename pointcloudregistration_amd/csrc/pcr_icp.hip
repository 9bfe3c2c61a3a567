// libpcr.so -- one ICRP iteration as fused device work (gfx950).
//
// Reference dataflow (SURVEY.md Appendix C;
// FineRegistrationClosestPoints/CorrespondenceEstimationClosestPoints.cpp:54-117,
// TransformationEstimationSVD2.cpp:33-103, DefaultConvergenceCriterial.cpp:46-65):
//   copy + transform the whole source, build a KD-tree on it, then serially per sample:
//   forward K-NN, reverse 1-NN, two gates, push_back; later 5 more passes for centroids and H.
// Here: ONE kernel per iteration in which each lane owns a sample and does
// transform -> forward K-NN (target grid) -> reverse 1-NN (source grid; candidates moved by (R,T) as they are read: no
// transformed copy of the source exists) -> gates -> masked fp64 accumulation, followed by a reduction per WAVE (a tile
// of 64 samples) through LDS; the waves that draw the last tickets fold the tile partials in a fixed order (bitwise
// reproducible run to run) and, in a device-resident fit on one GPU, run the 3x3 solve and the convergence criteria on
// the spot.
#include <math.h>
#include <string.h>
#include <mutex>
#include <vector>
#include "knn_device.h"
#include "fast_search.h"
#include "solve_core.h"
#ifndef PCR_STEP_WAVES
#define PCR_STEP_WAVES 4
#endif
#ifndef PCR_MULTI_WAVES
#define PCR_MULTI_WAVES 3  // nested levels: 168 registers spill far less than 128 (16 M LiDAR iteration: 26.7 -> 17.5 ms)
#endif
#ifndef PCR_FAST_WAVES
#define PCR_FAST_WAVES 4
#endif
#ifndef PCR_FAST_WAVES8
#define PCR_FAST_WAVES8 3  // K = 5..8: 168 registers hold the longer list without spills (1 M, K = 5: 236 -> 215 us; 2 waves: the same)
#endif

namespace pcr {

// What changes from one iteration to the next.  It lives in DEVICE memory: written by set_dyn_kernel (single-step
// entry points: the host computed it) or by fit_solve_kernel (device-resident loop: the previous iteration's
// solve computed it), read -- as wave-uniform scalars -- by the transform and step kernels.  No kernel argument
// depends on the iteration, so a whole fit can be enqueued without the host ever waiting for the device.
struct StepDyn {
    Rigid M;         // current (R,T)
    double rinv[9];  // inverse of R (column-major), for locating q in the source grid
    double pivot_p[3], pivot_q[3];
    double rev_bound_scale, rev_extra_margin;
    // key-space error bound of the filtered REVERSE search (fast_search.h) on the source's level-0 grid: the grid's own
    // key_delta, plus |R v|^2 vs |v|^2 (1 - rev_bound_scale >= 3 * the orthogonality defect) over the key range, plus the
    // rounding of the inverse map at the largest distance of interest.  Uniform over a launch: worked out here, once, by
    // whoever makes the transform, not by every lane.
    uint32_t rev_delta;
    uint32_t done;  // device-resident loop only: converged/stopped -- the remaining enqueued kernels return at once
    // device-resident loop only: no source point's image moved farther than this since the previous iteration
    // (+inf: unknown).  The K-th-neighbour distance is 1-Lipschitz in the query, so last iteration's distance plus this
    // bounds this iteration's before a single candidate is looked at (StepArgs::dk_io).
    float disp_bound;
};

// (R,T) -> everything else in StepDyn.  Host and device run this same code (bit-identical results).
//   c_src, c_tgt : centres of the clouds' bounds (the pivots only condition the sums)
//   cloud_scale  : largest |coordinate| of either cloud
// false: R is not a rotation to within 1e-3.
//   src_key_sc, src_key_delta : GridView::key_sc / key_delta of the source index's level 0 (0: no shadows)
PCR_HD inline bool make_dyn(const double *R, const double *T, const double *c_src, const double *c_tgt,
                            double cloud_scale, double src_key_sc, uint32_t src_key_delta, StepDyn &D, double *defect_out) {
    for (int i = 0; i < 9; ++i) D.M.r[i] = R[i];
    for (int i = 0; i < 3; ++i) D.M.t[i] = T[i];
    double defect = 0;
    const bool ok = invert3(R, D.rinv, defect) && defect <= 1e-3;
    if (defect_out) *defect_out = defect;
    for (int r = 0; r < 3; ++r) {
        D.pivot_p[r] = R[r] * c_src[0] + R[3 + r] * c_src[1] + R[6 + r] * c_src[2] + T[r];
        D.pivot_q[r] = c_tgt[r];
    }
    // |R v| >= sigma_min |v| with sigma_min^2 >= 1 - 3*defect (Gershgorin on R^T R)
    D.rev_bound_scale = (1.0 - 3.0 * defect) * (1.0 - 1.8189894035458565e-12);
    const double scale = fmax(fabs(T[0]) + fabs(T[1]) + fabs(T[2]), 0.0) + cloud_scale;
    D.rev_extra_margin = 3.637978807091713e-12 * scale;  // 2^-38: rounding of the inverse map
    D.rev_delta = src_key_delta + (uint32_t)fmin((1.0 - D.rev_bound_scale) * 1.1 * KEY_RANGE +
                                                     2.0 * KEY_MAX_ROOT * 1.01 * D.rev_extra_margin * src_key_sc + 2.0,
                                                 268435456.0);
    D.done = 0;
    D.disp_bound = __builtin_huge_valf();
    return ok;
}

__global__ void set_dyn_kernel(StepDyn value, StepDyn *dst) { *dst = value; }
__global__ void __launch_bounds__(256) fill_u32_kernel(uint32_t *dst, uint32_t n, uint32_t value) {
    for (uint32_t i = threadIdx.x; i < n; i += 256u) dst[i] = value;
}


struct FitDev;
struct StepArgs {
    MultiView src, tgt;
    const MultiView *src_mem, *tgt_mem;  // the indices' own device copies of the same (pcr_index::view_d)
    const uint32_t *src_pos_of;  // original index -> position in src.pts
    const uint32_t *src_order;   // positions of src.pts in brick order (processing order of "all samples")
    const uint32_t *sample_idx;  // NULL: samples are positions [sample_begin, sample_begin+m)
    uint32_t sample_begin, m;
    const StepDyn *dyn;          // per-iteration values (device memory)
    int k;
    double thr_fwd, thr_rev;
    uint32_t exact;  // PCR_STEP_EXACT_INDICES
    double *tile_partials;  // [17][ntiles], ntiles = ceil(m / 64): one partial per column and wave (wave_reduce17)
    // device-resident loop only (else NULL): per sample slot t, the forward search's exact K-th squared distance of the
    // previous iteration, rounded up to float (+inf: none) -- read as a bound, overwritten with this iteration's
    float *dk_io;
    // In-kernel fold (NULL tickets: fold_partials_kernel follows instead).  Tiles are grouped by 64; the wave that hands
    // in a group's last tile folds the group's partials, the wave that finishes the last group folds the groups into
    // partials_out -- fixed grouping, fixed order: bitwise reproducible -- and sets every counter back to zero for the
    // next launch.  fit_inline (device-resident loop on one GPU): that same wave then runs the solve.
    uint32_t *tickets;       // [1 + ngroups], zero between launches
    double *group_partials;  // [17][ngroups]
    double *partials_out;    // [17]
    struct FitDev *fit_inline;
    double *fit_traj;
    uint32_t *out_fwd, *out_rev;
    uint8_t *out_keep;
    // the normal-aware variant only (icp_step_normal_kernel): per-point normals in ORIGINAL order, the two thresholds
    const double *src_normals, *tgt_normals;
    double nvdt, ndt;
    uint32_t compare_normals;
};

// The filtered kernel reads its arguments from DEVICE memory (a copy of StepArgs), not from the kernel-argument
// segment: by-value arguments are all fetched at kernel entry and stay live (or spilled) in scalar registers for the
// whole kernel -- two grids' worth of geometry plus the transform is far more than the scalar file holds.  Behind a
// pointer every phase loads what it needs where it needs it.
__global__ void __launch_bounds__(256) set_args_kernel(StepArgs value, StepArgs *dst) {
    const uint32_t *s = (const uint32_t *)&value;
    uint32_t *d = (uint32_t *)dst;
    for (uint32_t i = threadIdx.x; i < sizeof(StepArgs) / 4; i += 256) d[i] = s[i];
}
static_assert(sizeof(StepArgs) % 4 == 0 && sizeof(StepArgs) <= 3584, "StepArgs travels as a kernel argument of set_args_kernel");

// An opaque copy of a (wave-uniform) pointer: loads through it cannot be merged with, or hoisted to, loads through
// the original -- values are re-read where they are used instead of living in scalar registers across the searches.
// The copy points into the CONSTANT address space: the memory is not written while the kernel runs (an earlier kernel
// on the stream wrote it), and a load from there at a uniform address is a scalar load.
#define PCR_CONST_AS __attribute__((address_space(4)))
template <class T>
__device__ __forceinline__ const PCR_CONST_AS T *reload_from(const T *p) {
    const uint64_t v = (uint64_t)p;
    uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    asm volatile("" : "+s"(lo), "+s"(hi));
    return (const PCR_CONST_AS T *)(((uint64_t)hi << 32) | lo);
}
template <class T>
__device__ __forceinline__ const PCR_CONST_AS T *reload_from(const PCR_CONST_AS T *p) {
    return reload_from((const T *)p);
}

// Relaxed atomics at DEVICE scope: performed at the memory side, past the per-XCD L2s.  Data that blocks hand to each
// other inside one kernel travels through these; a device-scope release FENCE instead would write the whole L2 back
// (measured: one per block doubles the step kernel's time).
__device__ __forceinline__ void st_agent(double *p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_agent(const double *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// A lane that has issued st_agent stores and is about to let ANOTHER lane publish them (a ticket drawn after a block
// barrier) waits here until memory has acknowledged them.  A workgroup barrier does not do that on gfx950: s_barrier
// waits for waves, not for their outstanding stores, and the relaxed ticket atomic travels on another channel -- without
// this wait the ticket can be seen by a block on another XCD while a partial sum is still in flight, and that block folds
// whatever the slot held before (last launch's value: close enough, in a converging fit, to pass a 1e-9 test).  The
// stores being write-through (sc1) themselves, nothing needs writing back: no release fence, only the wait.
// tests/test_isa_fold_order.py checks the shipped code object for it.
__device__ __forceinline__ void wait_stores_acked() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// ---- the 17 sums: per TILE (the 64 samples of one wave), per GROUP of 64 tiles, total -----------------------------
// Every step kernel ends the same way, so that their sums are bit-identical to each other: a wave reduces its 64 lanes'
// 17 values to one partial per column (wave_reduce17: fixed order), stores it at [column][tile]; the wave that hands in
// the last tile of a group of 64 folds the group (wave_fold17), the wave that hands in the last group folds the groups
// into the 17 sums -- fixed grouping, fixed order: bitwise reproducible run to run -- and, in a device-resident fit on
// one GPU, runs the solve.  The unit is the WAVE, not the block: a wave of the filtered kernel (64-lane blocks) frees
// its slot when IT is done, not when the slowest of four is (round 2: 6 000 of a block's 61 500 cycles).
constexpr int WR_STRIDE = 68;  // doubles per column of a wave's scratch: 8 columns x 4 parts of a half-wave read land on 32 distinct bank pairs
constexpr int WAVE_LDS_BYTES = 17 * WR_STRIDE * 8;      // 9 248
constexpr int REDUCE17_LDS_BYTES = 4 * WAVE_LDS_BYTES;  // a 256-lane block: one slice per wave (36 992)

// LDS written by some lanes of a wave and read by others of the SAME wave: the hardware executes a wave's LDS
// instructions in order; this only keeps the compiler from reordering them.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// 17 per-lane values -> one value per column, through the wave's own LDS slice, not shuffles: every lane parks its 17
// values (17 conflict-free writes), four lanes per column add 16 values each and a 2-step shuffle finishes columns
// 0..15; column 16 takes 16 lanes x 4 values and a 4-step shuffle.  On return lane 4c holds column c (c < 16) in s,
// lane 0 holds column 16 in s16.  Fixed order: reproducible bit for bit.
//   lds: WAVE_LDS_BYTES that the WAVE no longer needs for anything else (it hands over its row table)
__device__ __forceinline__ void wave_reduce17(const double (&v)[17], double *__restrict__ lds, double &s, double &s16) {
    const uint32_t lane = threadIdx.x & 63u;
    wave_lds_sync();
#pragma unroll
    for (int j = 0; j < 17; ++j) lds[j * WR_STRIDE + lane] = v[j];
    wave_lds_sync();
    const uint32_t col = lane >> 2, part = lane & 3u;
    s = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += lds[col * WR_STRIDE + i * 4 + part];
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 1, 64);
    s16 = 0.0;
    if (lane < 16u) {
#pragma unroll
        for (int i = 0; i < 4; ++i) s16 += lds[16 * WR_STRIDE + i * 16 + lane];
    }
    s16 += __shfl_xor(s16, 8, 64);
    s16 += __shfl_xor(s16, 4, 64);
    s16 += __shfl_xor(s16, 2, 64);
    s16 += __shfl_xor(s16, 1, 64);
    wave_lds_sync();  // (the slice may be written again at once: a fold right after a tile)
}
// ... stored at out[column * stride + at]: device-scope (write-through) stores -- the fold of a wave on another XCD reads them
__device__ __forceinline__ void store17(double s, double s16, double *__restrict__ out, size_t stride, size_t at) {
    const uint32_t lane = threadIdx.x & 63u;
    if ((lane & 3u) == 0u) st_agent(&out[(size_t)(lane >> 2) * stride + at], s);
    if (lane == 0u) st_agent(&out[(size_t)16 * stride + at], s16);
}

template <class DY>
__device__ __forceinline__ void accumulate(double (&acc)[17], bool keep, double px, double py, double pz, double qx,
                                           double qy, double qz, double dF, const DY &A) {
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

// one thread of the solve (defined with the device-resident loop below)
__device__ void fit_solve_body(FitDev *F, const double *partials, double *traj);

// 17 columns x `count` values (column j at x + j * stride + i) -> out[j * out_stride]: ONE wave; lane l adds the
// values i = l, l + 64, ... in that order, then wave_reduce17.  Device-scope loads (they miss the XCD's L2 by
// construction); the caller has made sure the values are there (a ticket).
__device__ __forceinline__ void wave_fold17(const double *x, size_t stride, uint32_t count, double *__restrict__ lds,
                                            double *out, size_t out_stride) {
    const uint32_t lane = threadIdx.x & 63u;
    double v[17];
#pragma unroll
    for (int j = 0; j < 17; ++j) v[j] = 0.0;
    // (two rounds of loads in flight per trip -- the loads are device-scope atomics, which the compiler keeps in program
    // order: one round per trip is one trip to memory per 64 values of a column, and the fold of 245 groups waited for
    // four of them; the order of the additions is unchanged)
    uint32_t i = lane;
    for (; i + 64u < count; i += 128u) {
        double a[17], b[17];
#pragma unroll
        for (int j = 0; j < 17; ++j) a[j] = ld_agent(&x[(size_t)j * stride + i]);
#pragma unroll
        for (int j = 0; j < 17; ++j) b[j] = ld_agent(&x[(size_t)j * stride + i + 64u]);
#pragma unroll
        for (int j = 0; j < 17; ++j) v[j] = (v[j] + a[j]) + b[j];
    }
    if (i < count) {
#pragma unroll
        for (int j = 0; j < 17; ++j) v[j] += ld_agent(&x[(size_t)j * stride + i]);
    }
    double s, s16;
    wave_reduce17(v, lds, s, s16);
    store17(s, s16, out, out_stride, 0);
}

#ifdef PCR_BLOCK_TIMES
__device__ unsigned long long g_solve_times[4];  // timing build: last fold's start, solve start, solve end (100 MHz ticks)
#endif
// The hand-over.  A wave that has stored tile partials -- and whose storing lanes have waited for memory's
// acknowledgement (wait_stores_acked) -- draws a ticket of the tiles' group: a device-scope atomic add by lane 0 of as
// many tiles as it hands in.  The wave that draws the group's last ticket folds the group, draws the launch's ticket
// and, if that was the last group, folds the groups, sets every counter back to zero for the next launch and runs the
// solve.  Wave-uniform control flow throughout; no block barrier anywhere in here.
// (Measured and dropped, round 3: (i) an ARRIVAL ticket drawn before the closing arithmetic plus a fire-and-forget
// completion count, so that 63 waves of 64 end one round trip earlier -- 141.3 us against 136.2: a device-scope atomic
// takes longer to come back than the closing arithmetic lasts, and there are twice as many of them; (ii) several tiles
// per wave, a tile's ticket drawn and redeemed while the wave works on its next tile -- any loop around the tile's body
// takes the register allocator from 62 spilled registers to 96-147, some inside the candidate loops: 173-192 us.)
__device__ __forceinline__ uint32_t ticket_draw(uint32_t *tickets, uint32_t g, uint32_t handed_in) {
    uint32_t old = 0;
    if ((threadIdx.x & 63u) == 0u)
        old = __hip_atomic_fetch_add(&tickets[1u + g], handed_in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return __builtin_amdgcn_readfirstlane(old);
}
template <class AP>
__device__ __forceinline__ void ticket_redeem(AP A, uint32_t old, uint32_t g, uint32_t handed_in, uint32_t ntiles,
                                              double *__restrict__ lds) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t ngroups = (ntiles + 63u) >> 6, gsize = min(64u, ntiles - (g << 6));
    if (old + handed_in != gsize) return;
    uint32_t *tickets = A->tickets;
    wave_fold17(A->tile_partials + ((size_t)g << 6), ntiles, gsize, lds, A->group_partials + g, ngroups);
    wait_stores_acked();
    uint32_t o2 = 0;
    if (lane == 0u) o2 = __hip_atomic_fetch_add(&tickets[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (__builtin_amdgcn_readfirstlane(o2) != ngroups - 1u) return;
#ifdef PCR_BLOCK_TIMES
    if (lane == 0u) g_solve_times[0] = wall_clock64();
#endif
    double *out = A->partials_out;
    wave_fold17(A->group_partials, ngroups, ngroups, lds, out, 1);
    for (uint32_t i = lane; i < ngroups + 1u; i += 64u) tickets[i] = 0u;
    FitDev *fit = A->fit_inline;
    if (fit) {
        wait_stores_acked();
#ifdef PCR_BLOCK_TIMES
        if (lane == 0u) g_solve_times[1] = wall_clock64();
#endif
        if (lane == 0u) fit_solve_body(fit, out, A->fit_traj);
#ifdef PCR_BLOCK_TIMES
        if (lane == 0u) g_solve_times[2] = wall_clock64();
#endif
    }
}

// The end of a 256-lane block (the fp64 walk, brute force, the normal-aware variant): its four waves are tiles
// 4 lb .. 4 lb + 3; each reduces and stores its own, wave 0 hands them in together.
__device__ __forceinline__ void store_block_tiles256(double (&acc)[17], unsigned char *smem, double *tile_partials,
                                                     uint32_t lb, uint32_t ntiles) {
    const uint32_t wave = threadIdx.x >> 6, tile = lb * 4u + wave;
    __syncthreads();  // every wave of the block is done with the search's use of smem (row tables, candidate tiles)
    if (tile < ntiles) {
        double s, s16;
        wave_reduce17(acc, (double *)(smem + wave * WAVE_LDS_BYTES), s, s16);
        store17(s, s16, tile_partials, ntiles, tile);
    }
    wait_stores_acked();  // before the barrier behind which the ticket is drawn
}
template <class AP>
__device__ __forceinline__ void finish_block256(double (&acc)[17], unsigned char *smem, AP A, uint32_t lb, uint32_t ntiles) {
    store_block_tiles256(acc, smem, A->tile_partials, lb, ntiles);
    uint32_t *tickets = A->tickets;
    if (!tickets) return;  // (fold_partials_kernel follows)
    __syncthreads();
    if ((threadIdx.x >> 6) != 0u) return;
    const uint32_t first = lb * 4u, handed_in = min(4u, ntiles - first), g = first >> 6;
    ticket_redeem(A, ticket_draw(tickets, g, handed_in), g, handed_in, ntiles, (double *)smem);
}

// MULTI: either index has more than one level (clouds of very uneven density); the single-level instance is the
// plain search on level 0, with none of the level loop in its code.
// The reverse search moves its candidates by (R,T) as it reads them, from the source index itself (CPointCloud::transform's
// expression, so every distance is the one the reference gets from its transformed copy, :73-74): no copy is ever
// written, an index holds no per-iteration state -- fits may share it -- and no rank of a sharded fit does work
// proportional to the WHOLE source.
#ifdef PCR_BLOCK_TIMES
// timing build only (never shipped): start/end of every block of the last grid-kernel launch, 100 MHz ticks
__device__ unsigned long long g_block_times[2 * 65536];
__device__ unsigned long long g_fast_times[4 * 65536];  // filtered kernel: start, forward search done, reverse done, end
__device__ uint32_t g_fast_clocks[16 * 65536];          // ... and wave 0's shader clock at 14 stage boundaries
#define FAST_STAMP(i) do { if (threadIdx.x == 0 && lb < 65536u) g_fast_times[4 * lb + (i)] = wall_clock64(); } while (0)
#else
#define FAST_STAMP(i) do { } while (0)
#endif
template <int KC, int MULTI>  // 0: single-level; 1: nested, dense cell tables only; 2: nested, some level has a compact table
__global__ void __launch_bounds__(256, MULTI ? PCR_MULTI_WAVES : PCR_STEP_WAVES) icp_step_grid_kernel(StepArgs A) {
    const StepDyn *__restrict__ dynp = A.dyn;
    if (dynp->done) return;  // uniform over the grid
    const StepDyn D = *dynp;
    const uint32_t lb = MULTI ? xcd_remap_chunked(blockIdx.x, gridDim.x) : xcd_remap(blockIdx.x, gridDim.x);
#ifdef PCR_BLOCK_TIMES
    if (threadIdx.x == 0 && lb < 65536u) g_block_times[2 * lb] = wall_clock64();
#endif
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
            orig = (uint32_t)A.src.all[pos].w;
            slot = orig;
        }
        const Pt4 s = A.src.all[pos];
        xform(D.M, s.x, s.y, s.z, px, py, pz);  // CurSrcPoint = R*s + T (:86)
    }
    // ---- forward: K-th nearest target point of p (:89-92) ----
    const double fmax_d2 = A.exact ? PCR_INF : A.thr_fwd;
    const bool fin = live && isfinite(px) && isfinite(py) && isfinite(pz);
    TopK<KC> top;
    top.init(fmax_d2);
    // per-lane row table of the ring walk (knn_device.h: rows 256 slots apart), later the block reduction's scratch
    __shared__ __align__(16) unsigned char smem[REDUCE17_LDS_BYTES];
    static_assert(sizeof(uint2) * 9 * 256 <= REDUCE17_LDS_BYTES, "row table fits");
    uint2 *my_rows = (uint2 *)smem + threadIdx.x;
    if constexpr (MULTI) {  // (every lane calls: the level search finishes its longest queries wave-cooperatively)
        multi_lane_search<KC, false, true, MULTI == 2>(A.tgt, A.tgt_mem, A.tgt.all, fin, px, py, pz, px, py, pz, fmax_d2, 1.0 - 1.8189894035458565e-12, 0.0, top,
                              my_rows);
    } else if (fin) {
        lane_search<KC>(A.tgt.v[0], A.tgt.all, px, py, pz, px, py, pz, fmax_d2, 1.0 - 1.8189894035458565e-12, 0.0, top,
                        my_rows);
    }
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
        const Pt4 qpt = A.tgt.all[A.tgt.pos_of[j]];
        qx = qpt.x;
        qy = qpt.y;
        qz = qpt.z;
        const double ux = qx - D.M.t[0], uy = qy - D.M.t[1], uz = qz - D.M.t[2];
        sx = D.rinv[0] * ux + D.rinv[3] * uy + D.rinv[6] * uz;
        sy = D.rinv[1] * ux + D.rinv[4] * uy + D.rinv[7] * uz;
        sz = D.rinv[2] * ux + D.rinv[5] * uy + D.rinv[8] * uz;
    }
    TopK<1> rt;
    rt.d[0] = dF;
    rt.idx[0] = orig;
    const bool rev_active = need_rev && isfinite(sx) && isfinite(sy) && isfinite(sz);
    if constexpr (MULTI) {
        multi_lane_search<1, true, true, MULTI == 2>(A.src, A.src_mem, A.src.all, rev_active, qx, qy, qz, sx, sy, sz, PCR_INF, D.rev_bound_scale,
                                   D.rev_extra_margin, rt, my_rows, &D.M, &dynp->M);
    } else if (rev_active) {
        lane_search<1, true>(A.src.v[0], A.src.all, qx, qy, qz, sx, sy, sz, PCR_INF, D.rev_bound_scale, D.rev_extra_margin,
                             rt, my_rows, &D.M);
    }
    uint32_t r = PCR_NO_INDEX;
    bool keep = false;
    if (need_rev) {
        r = rt.idx[0];
        const Pt4 sp = A.src.all[r == orig ? pos : A.src_pos_of[r]];
        const double dR = cand_d2<true>(&D.M, px, py, pz, sp);  // (CurSrcPoint - TmpPoint).squaredNorm()
        keep = passF && dR < A.thr_rev;  // strict < (:98)
    }
    if (live) {
        accumulate(acc, keep, px, py, pz, qx, qy, qz, dF, D);
        if (A.out_fwd) A.out_fwd[slot] = found ? j : PCR_NO_INDEX;
        if (A.out_rev) A.out_rev[slot] = r;
        if (A.out_keep) A.out_keep[slot] = keep ? 1 : 0;
    }
#ifdef PCR_BLOCK_TIMES
    if (threadIdx.x == 0 && lb < 65536u) g_block_times[2 * lb + 1] = wall_clock64();
#endif
    finish_block256(acc, smem, &A, lb, (A.m + 63u) >> 6);
}

// f4, opt-in (pcr_icp_step_normal_fit): CCorrespondenceEstimationClosestPoints::__insertFitPairs
// (CorrespondenceEstimationClosestPoints.cpp:128-175) as the pair decision, in place of the reciprocal gate (:94-104);
// its call is commented out in the reference (:105).  The K forward neighbours are tried nearest first; the first whose
// direction from the sample is (anti)parallel to its own normal to within NVDT -- and, if asked, whose normal agrees
// with the sample's rotated normal to within NDT (the reference hard-wires that test off, :133) -- makes the pair.
// TmpTgtPoint (read unassigned in the reference: :139,:146,:166) is the neighbour's position here; see oracle/icrp.cpp.
__device__ __forceinline__ void normalize3(double &x, double &y, double &z) {  // Eigen: *this /= sqrt(squaredNorm())
    const double n = sqrt(dadd(dadd(dmul(x, x), dmul(y, y)), dmul(z, z)));
    x = x / n;
    y = y / n;
    z = z / n;
}
template <int KC, bool MULTI>
__global__ void __launch_bounds__(256, MULTI ? PCR_MULTI_WAVES : PCR_STEP_WAVES) icp_step_normal_kernel(StepArgs A) {
    const StepDyn *__restrict__ dynp = A.dyn;
    const StepDyn D = *dynp;
    const uint32_t lb = MULTI ? xcd_remap_chunked(blockIdx.x, gridDim.x) : xcd_remap(blockIdx.x, gridDim.x);
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
            pos = A.src_order[A.sample_begin + t];
            orig = (uint32_t)A.src.all[pos].w;
            slot = orig;
        }
        const Pt4 s = A.src.all[pos];
        xform(D.M, s.x, s.y, s.z, px, py, pz);  // (:86)
    }
    const bool fin = live && isfinite(px) && isfinite(py) && isfinite(pz);
    TopK<KC> top;
    top.init(PCR_INF);
    __shared__ __align__(16) unsigned char smem[REDUCE17_LDS_BYTES];
    uint2 *my_rows = (uint2 *)smem + threadIdx.x;
    if constexpr (MULTI) {
        multi_lane_search<KC, false, (KC <= 8)>(A.tgt, A.tgt_mem, A.tgt.all, fin, px, py, pz, px, py, pz, PCR_INF,
                                                1.0 - 1.8189894035458565e-12, 0.0, top, my_rows);
    } else if (fin) {
        lane_search<KC>(A.tgt.v[0], A.tgt.all, px, py, pz, px, py, pz, PCR_INF, 1.0 - 1.8189894035458565e-12, 0.0, top, my_rows);
    }
    double snx = 0, sny = 0, snz = 0;  // CurSrcNormal = R * n (:87)
    if (fin && A.src_normals) {
        const double *n = A.src_normals + (size_t)orig * 3;
        const double a = n[0], b = n[1], c = n[2];
        snx = dadd(dadd(dmul(D.M.r[0], a), dmul(D.M.r[3], b)), dmul(D.M.r[6], c));
        sny = dadd(dadd(dmul(D.M.r[1], a), dmul(D.M.r[4], b)), dmul(D.M.r[7], c));
        snz = dadd(dadd(dmul(D.M.r[2], a), dmul(D.M.r[5], b)), dmul(D.M.r[8], c));
    }
    uint32_t chosen = PCR_NO_INDEX;
    double qx = 0, qy = 0, qz = 0;
#pragma unroll
    for (int c = 0; c < KC; ++c) {  // (:140) nearest first
        const uint32_t j = top.idx[c];
        if (fin && c < A.k && chosen == PCR_NO_INDEX && j != PCR_NO_INDEX) {
            const double *nn = A.tgt_normals + (size_t)j * 3;
            double nx = nn[0], ny = nn[1], nz = nn[2];  // (:143)
            normalize3(nx, ny, nz);                     // (:145)
            const Pt4 q = A.tgt.all[A.tgt.pos_of[j]];
            double vx = q.x - px, vy = q.y - py, vz = q.z - pz;  // (:146)
            normalize3(vx, vy, vz);                              // (:147)
            const double nvd = fabs(dadd(dadd(dmul(vx, vx), dmul(vy, vy)), dmul(vz, vz)) -
                                    fabs(dadd(dadd(dmul(vx, nx), dmul(vy, ny)), dmul(vz, nz))));  // (:149)
            if (nvd < A.nvdt) {                                                                   // (:155)
                double nd = 0.0;
                if (A.src_normals) {
                    normalize3(snx, sny, snz);                                          // (:157) in place, every time
                    nd = fabs(dadd(dadd(dmul(nx, snx), dmul(ny, sny)), dmul(nz, snz)));  // (:158)
                }
                if (!A.compare_normals || nd > A.ndt) {  // (:164)
                    chosen = j;
                    qx = q.x;
                    qy = q.y;
                    qz = q.z;
                }
            }
        }
    }
    const bool keep = chosen != PCR_NO_INDEX;
    if (live) {
        accumulate(acc, keep, px, py, pz, qx, qy, qz, keep ? sqdist3(px, py, pz, qx, qy, qz) : 0.0, D);
        if (A.out_fwd) A.out_fwd[slot] = chosen;
        if (A.out_keep) A.out_keep[slot] = keep ? 1 : 0;
    }
    finish_block256(acc, smem, &A, lb, (A.m + 63u) >> 6);
}

// The same iteration on single-level grids that carry fp32 shadows: both searches go through the filtered walk
// (fast_search.h) -- fp32 keys select, fp64 decides and verifies -- and a lane whose result that walk cannot vouch for
// repeats the search with the all-fp64 walk of the kernel above.  The reverse search needs no transformed copy of the
// source: keys are taken in the SOURCE frame (q mapped back by R^-1; the key error bound is widened by R's deviation
// from an isometry), and only the few candidates that reach the fp64 stage are moved by (R,T), with
// CPointCloud::transform's expression.  Every index, flag and sum is bit-identical to icp_step_grid_kernel's (tested).
// The rare lanes the filtered walk cannot vouch for repeat the search with the all-fp64 walk.  (Taking these out of
// line -- so that the walk's registers are not part of the filtered kernel's own allocation -- removes the hot path's
// spills but gains nothing: the kernel is bound by instruction issue, not by its 40 MB of spill traffic.)
#ifndef PCR_FALLBACK_INLINE
#define PCR_FALLBACK_INLINE __forceinline__  // (measured: out of line 170 us at 4 waves/SIMD, 155 at "5"; inline 153-155)
#endif
template <int KC>
__device__ PCR_FALLBACK_INLINE void fallback_forward(const StepArgs *__restrict__ Ap, double px, double py, double pz, double fmax_d2,
                                                     double fwd_scale, int k, uint2 *my_rows, double *dF, uint32_t *j,
                                                     const FastExact<KC> *fx = nullptr) {
    TopK<KC> top;
    top.init(fmax_d2);
    const GridView G = Ap->tgt.v[0];  // (through the generic pointer)
    // What the filtered walk found in its 3x3x3 cells is exact when it says so (its ranking verified) and merely did not
    // SETTLE the query -- the K-th neighbour may lie farther out: the typical sample of a sparse stretch of a scan.  The
    // fp64 walk then need not visit those cells again: it starts with that list and goes on from ring 2 -- provided both
    // walks are centred on the same cell (the filtered walk picks its cell in fp32).
    bool seeded = false;
    if (fx && fx->trusted && fx->cell[0] == cell_coord(px, G.lo[0], G.inv_h, G.g[0]) &&
        fx->cell[1] == cell_coord(py, G.lo[1], G.inv_h, G.g[1]) && fx->cell[2] == cell_coord(pz, G.lo[2], G.inv_h, G.g[2])) {
        seeded = true;
#pragma unroll
        for (int a = 0; a < KC + 1; ++a)
            if (a < fx->n && fx->i[a] != PCR_NO_INDEX && better(fx->d[a], fx->i[a], top.worst_d(), top.worst_i()))
                top.insert(fx->d[a], fx->i[a]);
    }
    // (one call site for the shells: lanes of a wave that come here by different routes walk them together)
    bool shells;
    if (seeded) {
        // the fp64 geometry often settles what the filtered walk's fp32 bounds, shrunk for safety, could not
        const double qmargin = 9.094947017729282e-13 * (fabs(px) + fabs(py) + fabs(pz));
        shells = !ring1_settles(G, px, py, pz, G.margin + qmargin, fwd_scale, top.worst_d(), fmax_d2);
    } else {
        shells = !ring1_pruned<KC, false, 64>(G, Ap->tgt.all, px, py, pz, px, py, pz, fmax_d2, fwd_scale, 0.0, top, my_rows,
                                          (const Rigid *)nullptr);
    }
    if (shells)
        shells_beyond_ring1<KC, false, false, 0, 64, false>(G, Ap->tgt.all, px, py, pz, px, py, pz, fmax_d2, fwd_scale, 0.0, top, my_rows);
    top.slot(k, *dF, *j);
}
// BATCH2: lane_search's (the batch kernel, whose small problems' latency is the fall-back lanes': 3 250 -> 3 420 fits/s
// on C4 with it; the single-problem kernel without: 120.2 us per 1 M iteration against 122.8)
template <bool BATCH2>
__device__ PCR_FALLBACK_INLINE uint32_t fallback_reverse(const StepArgs *__restrict__ Ap, double qx, double qy, double qz, double sx,
                                                         double sy, double sz, double rbs, double rem_, double dF, uint32_t orig,
                                                         uint2 *my_rows) {
    TopK<1> rt;
    rt.d[0] = dF;
    rt.idx[0] = orig;
    const GridView G = Ap->src.v[0];
    const Rigid M = Ap->dyn->M;
    lane_search<1, true, 64, BATCH2>(G, Ap->src.all, qx, qy, qz, sx, sy, sz, PCR_INF, rbs, rem_, rt, my_rows, &M);
    return rt.idx[0];
}

#ifndef PCR_FAST_PAIR
#define PCR_FAST_PAIR 0  // phase 3's fp64 gathers shared by lane pairs (fast_search.h): measured 129.3 / 130.5 us against
                         // 128.4 / 128.9 without, same box, alternating runs -- off
#endif
#ifndef PCR_FAST_TRIES_FWD
#define PCR_FAST_TRIES_FWD 1
#endif
#ifndef PCR_FAST_U_FWD
#define PCR_FAST_U_FWD 8  // shadow records per trip of the forward / reverse search (fast_search.h: PCR_FAST_U)
#endif
#ifndef PCR_FAST_U_REV
#define PCR_FAST_U_REV 6
#endif
constexpr int FAST_RS = 64;  // the filtered kernel's blocks are single waves: 64 lanes per row of the LDS row table

// One tile -- 64 samples, one lane each -- of one problem.  smem: the wave's WAVE_LDS_BYTES (row table, then the
// reduction's scratch).
#ifdef PCR_DIAG
static __device__ uint8_t g_trips[2][1 << 20];  // (diagnostic build) per sample slot of the last launch: phase-2 trips
#endif
template <int KC, bool BATCHED = false>  // BATCHED: called by the kernel of many small problems (see fallback_reverse)
__device__ __forceinline__ void fast_tile(const StepArgs *__restrict__ Ap, uint32_t tile, uint32_t ntiles,
                                          unsigned char *smem) {
#ifdef PCR_BLOCK_TIMES
    const uint32_t lb = tile;
#endif
    FAST_STAMP(0);
#ifdef PCR_BLOCK_TIMES
    uint32_t ck[6];
    StageClock ckf, ckr;
    for (int i = 0; i < 4; ++i) ckf.t[i] = ckr.t[i] = 0;
    StageClock *pf = &ckf, *pr = &ckr;
    ck[0] = (uint32_t)__builtin_readcyclecounter();
#else
    StageClock *pf = nullptr, *pr = nullptr;
#endif
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t t = tile * 64u + lane;
    uint2 *my_rows = (uint2 *)smem + lane;
#ifdef PCR_DIAG
    uint32_t diag_trips_f = 0, diag_trips_r = 0;  // this sample's phase-2 trips (forward, reverse): g_trips, read by pcr_diag_read_trips
#endif
    uint32_t pos = 0, orig = 0, slot = 0;
    double px = 0, py = 0, pz = 0;
    float dk_prev = __builtin_huge_valf();  // last iteration's K-th squared distance (loaded beside the point: one wait)
    bool live;
    {
        const PCR_CONST_AS StepArgs *A = reload_from(Ap);
        live = t < A->m;
        if (live) {
            const bool listed = A->sample_idx != nullptr;
            if (listed) {
                orig = A->sample_idx[t];
                pos = A->src_pos_of[orig];
            } else {
                pos = A->src_order[A->sample_begin + t];
            }
            if (A->dk_io) dk_prev = A->dk_io[t];
            const Pt4 s = A->src.all[pos];  // one read of the record: the original index rides in .w
            if (!listed) orig = (uint32_t)s.w;
            slot = listed ? t : orig;
            xform(reload_from(A->dyn)->M, s.x, s.y, s.z, px, py, pz);  // CurSrcPoint = R*s + T (:86)
        }
    }
    // ---- forward: K-th nearest target point of p (:89-92)
    const bool fin = live && isfinite(px) && isfinite(py) && isfinite(pz);
    double dF;
    uint32_t j;
    bool found, passF, need_rev, q_known;
    uint32_t qpos = PCR_NO_INDEX;
    {
        const PCR_CONST_AS StepArgs *A = reload_from(Ap);
        const PCR_CONST_AS GridView &GT = A->tgt.v[0];
        const double fmax_d2 = A->exact ? PCR_INF : A->thr_fwd;
        const double fwd_scale = 1.0 - 1.8189894035458565e-12;
        bool fwd_done = !fin;
        dF = fmax_d2;
        j = PCR_NO_INDEX;
        FastExact<KC> fx;  // what the filtered walk knows exactly: the fp64 walk of an unsettled sample starts from it
        fx.trusted = false;
        // what last iteration knew: its K-th distance, and how far a point can have moved since (both rounded up)
        double bound_d2 = PCR_INF;
        float *dk_io = A->dk_io;
        if (dk_io && live) {
            const float a = sqrtf(dk_prev) * 1.0000005f, b = reload_from(A->dyn)->disp_bound;
            const float s2 = (a + b) * 1.0000005f;
            bound_d2 = (double)(s2 * s2) * 1.000001;  // inf stays inf
        }
        // NeighborSet.back() = K-th nearest (:91-92).  Called by the whole wave (lanes without a sample take no part in
        // phases 1-2 and help their partners in phase 3: fast_search.h, PAIR)
        fwd_done = fast_ring1<KC, false, false, PCR_FAST_TRIES_FWD, PCR_FAST_U_FWD, FAST_RS, PCR_FAST_PAIR != 0>(
            GT, GT.pts, px, py, pz, px, py, pz, fmax_d2, bound_d2, fwd_scale, 0.0, GT.key_delta, A->k, 0.0, 0u, dF, j, my_rows,
            (const Rigid *)nullptr, &fx, pf, nullptr, &qpos, fin
#ifdef PCR_DIAG
            , &diag_trips_f
#endif
            );
        if (!fin) {
            fwd_done = true;
            dF = fmax_d2;
            j = PCR_NO_INDEX;
            fx.trusted = false;
        }
        q_known = fin & fwd_done & (qpos != PCR_NO_INDEX);  // the walk hands over q's position with its index
        DIAG_ADD(32, 1);                          // (diagnostic build) waves
        DIAG_ADD(33, __any(!fwd_done) ? 1 : 0);   // ... in which some lane's forward search goes on to the fp64 walk
        if (__any(!fwd_done)) {
            if (!fwd_done) fallback_forward<KC>(Ap, px, py, pz, fmax_d2, fwd_scale, A->k, my_rows, &dF, &j, fin ? &fx : nullptr);
        }
        found = fin && j != PCR_NO_INDEX && dF < fmax_d2;
        passF = found && dF < A->thr_fwd;  // strict < (:97)
        need_rev = found && (passF || A->exact);
        if (dk_io && live) dk_io[t] = found ? __double2float_ru(dF) : __builtin_huge_valf();
    }
    FAST_STAMP(1);
#ifdef PCR_BLOCK_TIMES
    ck[1] = (uint32_t)__builtin_readcyclecounter();
#endif
    // ---- reverse: nearest TRANSFORMED source point of q (:94-95), seeded with the sample's own image (distance dF)
    double qx = 0, qy = 0, qz = 0;
    uint32_t r = PCR_NO_INDEX;
    bool keep = false;
    {
        const PCR_CONST_AS StepArgs *A = reload_from(Ap);
        const PCR_CONST_AS StepDyn *D = reload_from(A->dyn);
        const PCR_CONST_AS GridView &GS = A->src.v[0];
        double sx = 0, sy = 0, sz = 0;
        if (need_rev) {
            if (!q_known) qpos = A->tgt.pos_of[j];  // (the fp64 walk returns the index only)
            const Pt4 qpt = A->tgt.all[qpos];
            qx = qpt.x;
            qy = qpt.y;
            qz = qpt.z;
            const double ux = qx - D->M.t[0], uy = qy - D->M.t[1], uz = qz - D->M.t[2];
            sx = D->rinv[0] * ux + D->rinv[3] * uy + D->rinv[6] * uz;
            sy = D->rinv[1] * ux + D->rinv[4] * uy + D->rinv[7] * uz;
            sz = D->rinv[2] * ux + D->rinv[5] * uy + D->rinv[8] * uz;
        }
        const bool rev_active = need_rev && isfinite(sx) && isfinite(sy) && isfinite(sz);
        const double rbs = D->rev_bound_scale, rem_ = D->rev_extra_margin;
        const uint32_t rev_delta = D->rev_delta;  // (make_dyn)
        bool rev_done = !rev_active;
        uint32_t rv = orig;  // the reverse winner's index
        Pt4 srec = {0.0, 0.0, 0.0, 0ull};
        {
            double dRv_ = dF;
            uint32_t rv_ = orig;
            Pt4 srec_;
            const bool done_ = fast_ring1<1, true, true, 1, PCR_FAST_U_REV, FAST_RS, PCR_FAST_PAIR != 0>(
                GS, GS.pts, qx, qy, qz, sx, sy, sz, PCR_INF, dF, rbs, rem_, rev_delta, 1, dF, orig, dRv_, rv_, my_rows, &D->M,
                nullptr, pr, &srec_, nullptr, rev_active
#ifdef PCR_DIAG
                , &diag_trips_r
#endif
                );
            if (rev_active) {
                rev_done = done_;
                rv = rv_;
                srec = srec_;
            }
        }
        const bool s_known = rev_active & rev_done;
        DIAG_ADD(34, __any(!rev_done) ? 1 : 0);   // ... some lane's reverse search
        if (__any(!rev_done)) {
            if (!rev_done) rv = fallback_reverse<BATCHED>(Ap, qx, qy, qz, sx, sy, sz, rbs, rem_, dF, orig, my_rows);
        }
        if (need_rev) {
            r = rv;
            // (CurSrcPoint - TmpPoint).squaredNorm(): TmpPoint is the sample's own image when r is the sample -- the same
            // expression on the same operands as p, so the difference is exactly zero -- else the walk's winning record
            if (!s_known && r != orig) srec = A->src.all[A->src_pos_of[r]];
            const double dR = r == orig ? 0.0 : cand_d2<true>(&D->M, px, py, pz, srec);
            keep = passF && dR < A->thr_rev;  // strict < (:98)
        }
    }
#ifdef PCR_DIAG
    if (t < (1u << 20)) {
        g_trips[0][t] = (uint8_t)min(diag_trips_f, 255u);
        g_trips[1][t] = (uint8_t)min(diag_trips_r, 255u);
    }
#endif
    FAST_STAMP(2);
#ifdef PCR_BLOCK_TIMES
    ck[2] = (uint32_t)__builtin_readcyclecounter();
#endif
    const PCR_CONST_AS StepArgs *A = reload_from(Ap);
    uint32_t *tickets = A->tickets;
    double acc[17];
#pragma unroll
    for (int i = 0; i < 17; ++i) acc[i] = 0.0;
    if (live) {
        accumulate(acc, keep, px, py, pz, qx, qy, qz, dF, *reload_from(A->dyn));
        if (A->out_fwd) A->out_fwd[slot] = found ? j : PCR_NO_INDEX;
        if (A->out_rev) A->out_rev[slot] = r;
        if (A->out_keep) A->out_keep[slot] = keep ? 1 : 0;
    }
    {
        double s, s16;
        wave_reduce17(acc, (double *)smem, s, s16);
        store17(s, s16, A->tile_partials, ntiles, tile);
    }
    FAST_STAMP(3);
#ifdef PCR_BLOCK_TIMES
    ck[3] = (uint32_t)__builtin_readcyclecounter();
    if (lane == 0 && lb < 65536u) {
        uint32_t *o = g_fast_clocks + 16 * (size_t)lb;
        for (int i = 0; i < 4; ++i) { o[i] = ck[i]; o[4 + i] = ckf.t[i]; o[8 + i] = ckr.t[i]; }
    }
#endif
    if (!tickets) return;  // (fold_partials_kernel follows)
    wait_stores_acked();
    ticket_redeem(A, ticket_draw(tickets, tile >> 6, 1u), tile >> 6, 1u, ntiles, (double *)smem);
}

// The filtered step kernel: one tile per block, a block is ONE wave -- no barrier anywhere, a wave retires (and the
// hardware hands its slot to the next block) when IT is done.  Measured, 1 M x 1 M, K = 4: 136.2 us against 140.6 / 141.7
// with two / four such waves per block (which would keep neighbouring tiles on one CU's L1), and 135.7 for round 2's
// four waves per block with a block-wide reduction and one ticket per block.
template <int KC>
__global__ void __launch_bounds__(64, KC > 4 ? PCR_FAST_WAVES8 : PCR_FAST_WAVES) icp_step_fast_kernel(const StepArgs *__restrict__ Ap) {
    __shared__ __align__(16) unsigned char smem[WAVE_LDS_BYTES];
    const PCR_CONST_AS StepArgs *A = reload_from(Ap);
    if (reload_from(A->dyn)->done) return;  // uniform over the grid
    fast_tile<KC>(Ap, xcd_remap(blockIdx.x, gridDim.x), gridDim.x, smem);
}

// Many small problems, ONE launch (BASELINE config C4: a scan set registered pair by pair -- independent fits).
// Problem p owns the logical blocks (= tiles) [first[p], first[p+1]); each has its own arguments, state, tickets and
// partials, folds and solves for itself in the wave that hands in its last tile, and drops out of later launches once
// its criteria have fired.
template <int KC>
__global__ void __launch_bounds__(64, KC > 4 ? PCR_FAST_WAVES8 : PCR_FAST_WAVES) icp_step_batch_kernel(const StepArgs *const *__restrict__ argv,
                                                                            const uint32_t *__restrict__ first,
                                                                            uint32_t count) {
    __shared__ __align__(16) unsigned char smem[WAVE_LDS_BYTES];
    const uint32_t b = xcd_remap(blockIdx.x, gridDim.x);
    uint32_t lo = 0, hi = count;  // last p with first[p] <= b
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (first[mid] <= b)
            lo = mid;
        else
            hi = mid;
    }
    const StepArgs *Ap = argv[lo];
    if (reload_from(reload_from(Ap)->dyn)->done) return;  // uniform over the problem's blocks
    fast_tile<KC, true>(Ap, b - first[lo], first[lo + 1] - first[lo], smem);
}

// Small clouds: both searches as tiled brute force (LDS-staged tiles, broadcast reads).
template <int KC>
__global__ void __launch_bounds__(256) icp_step_brute_kernel(StepArgs A) {
    const StepDyn *__restrict__ dynp = A.dyn;
    if (dynp->done) return;
    const StepDyn D = *dynp;
    __shared__ __align__(32) unsigned char smem[REDUCE17_LDS_BYTES];  // candidate tiles, then the block reduction's scratch
    static_assert(sizeof(Pt4) * 512 <= REDUCE17_LDS_BYTES, "tile fits");
    Pt4 *tile = (Pt4 *)smem;
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
            orig = (uint32_t)A.src.all[pos].w;
            slot = orig;
        }
        const Pt4 s = A.src.all[pos];
        xform(D.M, s.x, s.y, s.z, px, py, pz);
    }
    const bool fin = live && isfinite(px) && isfinite(py) && isfinite(pz);
    TopK<KC> top;
    top.init();
    for (uint32_t base = 0; base < A.tgt.n; base += 512) {
        const int cnt = (int)min(512u, A.tgt.n - base);
        for (int u = threadIdx.x; u < cnt; u += 256) tile[u] = A.tgt.all[base + u];
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
        const Pt4 qpt = A.tgt.all[A.tgt.pos_of[j]];
        qx = qpt.x;
        qy = qpt.y;
        qz = qpt.z;
    }
    TopK<1> rt;
    rt.d[0] = found ? dF : PCR_INF;
    rt.idx[0] = found ? orig : PCR_NO_INDEX;
    for (uint32_t base = 0; base < A.src.n; base += 512) {
        const int cnt = (int)min(512u, A.src.n - base);
        for (int u = threadIdx.x; u < cnt; u += 256) {  // the transformed source (:73-74), tile by tile as it is staged
            const Pt4 sp = A.src.all[base + u];
            Pt4 tp;
            xform(D.M, sp.x, sp.y, sp.z, tp.x, tp.y, tp.z);
            tp.w = sp.w;
            tile[u] = tp;
        }
        __syncthreads();
        if (found) tile_scan<1>(tile, cnt, qx, qy, qz, rt);
        __syncthreads();
    }
    bool keep = false;
    uint32_t r = PCR_NO_INDEX;
    if (found) {
        r = rt.idx[0];
        const double dR = cand_d2<true>(&D.M, px, py, pz, A.src.all[A.src_pos_of[r]]);
        keep = passF && dR < A.thr_rev;
    }
    if (live) {
        accumulate(acc, keep, px, py, pz, qx, qy, qz, dF, D);
        if (A.out_fwd) A.out_fwd[slot] = found ? j : PCR_NO_INDEX;
        if (A.out_rev) A.out_rev[slot] = r;
        if (A.out_keep) A.out_keep[slot] = keep ? 1 : 0;
    }
    finish_block256(acc, smem, &A, blockIdx.x, (A.m + 63u) >> 6);
}

// Fold tile partials in a fixed order (explicit pair lists only: pcr_reduce_pairs): block j of the grid owns column j
// (one of the 17 sums); its 256 threads stride over the tiles' values (coalesced), then a tree.  Same order every run => the sums are
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
                                                           uint32_t P, StepDyn D, double *tile_partials) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    double acc[17];
#pragma unroll
    for (int j = 0; j < 17; ++j) acc[j] = 0.0;
    if (t < P) {
        const double px = pp[3 * (size_t)t], py = pp[3 * (size_t)t + 1], pz = pp[3 * (size_t)t + 2];
        const double qx = pq[3 * (size_t)t], qy = pq[3 * (size_t)t + 1], qz = pq[3 * (size_t)t + 2];
        accumulate(acc, true, px, py, pz, qx, qy, qz, sqdist3(px, py, pz, qx, qy, qz), D);
    }
    __shared__ __align__(16) unsigned char smem[REDUCE17_LDS_BYTES];
    store_block_tiles256(acc, smem, tile_partials, blockIdx.x, (P + 63u) >> 6);  // (fold_partials_kernel follows)
}

static double clouds_scale(const pcr_cloud *a, const pcr_cloud *b) {
    double scale = 0;
    for (int i = 0; i < 3; ++i) {
        scale = fmax(scale, fmax(fabs(a->lo[i]), fabs(a->hi[i])));
        scale = fmax(scale, fmax(fabs(b->lo[i]), fabs(b->hi[i])));
    }
    return scale;
}

static void cloud_centre(const pcr_cloud *c, double out[3]) {
    for (int a = 0; a < 3; ++a) out[a] = 0.5 * (c->lo[a] + c->hi[a]);
}

// ---- device-resident fit loop: state, solve and convergence on the device ---------------------------------------
struct FitDev {
    StepDyn dyn;  // first member: the step kernels read it through a StepDyn pointer
    double c_src[3], c_tgt[3], cloud_scale;
    double src_rad;  // half the diagonal of the source's bounds: no source point is farther from c_src
    double src_key_sc;       // the source index's key scale and error bound (make_dyn: StepDyn::rev_delta)
    uint32_t src_key_delta;
    double uniq;  // CUniqueData of the target: scales the MSE thresholds
    pcr_convergence_params conv;
    int32_t conv_enabled;
    // CDefaultConvergenceCriteria state (DefaultConvergenceCriterial.h:43-48)
    double pre_mse, cur_mse, pre_rot[9];
    uint32_t similar;
    uint32_t iteration;  // completed iterations
    int32_t state;       // pcr_fit_state
    uint32_t traj_cap;
    double last_n;
    double last_mse;  // MSE of the last completed iteration (what pcr_fit_read reports; cur_mse/pre_mse belong to the
                      // criteria emulation, whose short-circuit order can leave them one iteration behind)
};

__global__ void fit_init_kernel(FitDev value, FitDev *dst) { *dst = value; }

// One thread: what CIterativeFit::_fitStepV does after the correspondences (IterativeFit.cpp:156-174) --
// computeRotationAndTranslation, then the convergence criteria -- followed by the preparation of the next step.
__device__ __noinline__ void fit_solve_body(FitDev *__restrict__ F, const double *__restrict__ partials,
                                            double *__restrict__ traj) {
    // one thread, and every load from device memory costs it a microsecond: fetch the sums and the whole state at once
    // (all loads in flight together, the `done` flag among them) and read the copy from here on; stores go to *F
    double part[17];
    for (int i = 0; i < 17; ++i) part[i] = ld_agent(&partials[i]);  // (written by other lanes of this block: fold17)
    const FitDev f = *F;
    if (f.dyn.done) return;
    const uint32_t it = f.iteration + 1;  // the reference counts from 1 (IterativeFit.cpp:121-123)
    F->iteration = it;
    F->last_n = part[0];
    double R[9], T[3];
    for (int i = 0; i < 9; ++i) R[i] = f.dyn.M.r[i];
    for (int i = 0; i < 3; ++i) T[i] = f.dyn.M.t[i];
    double *row = (traj && it <= f.traj_cap) ? traj + (size_t)(it - 1) * 14 : nullptr;
    if (!(part[0] >= 1.0)) {  // the reference divides by zero here (RegUtilityFunctions.cpp:71)
        F->state = PCR_FIT_NO_CORRESPONDENCES;
        F->dyn.done = 1;
        if (row) {
            for (int i = 0; i < 9; ++i) row[i] = R[i];
            for (int i = 0; i < 3; ++i) row[9 + i] = T[i];
            row[12] = __builtin_nan("");
            row[13] = 0.0;
        }
        return;
    }
    double mse;
    solve_rigid_core(part, f.dyn.pivot_p, f.dyn.pivot_q, R, T, nullptr, nullptr, &mse);
    F->last_mse = mse;
    if (row) {
        for (int i = 0; i < 9; ++i) row[i] = R[i];
        for (int i = 0; i < 3; ++i) row[9 + i] = T[i];
        row[12] = mse;
        row[13] = part[0];
    }
    // ---- CDefaultConvergenceCriteria::_isConvergedV (DefaultConvergenceCriterial.cpp:69-83): short-circuit OR
    int state = PCR_FIT_NOT_CONVERGED;
    bool converged = false;
    double cur_mse = f.cur_mse;  // (stays as it was when an earlier test of the short-circuit chain fires)
    if (f.conv_enabled) {
        const pcr_convergence_params &c = f.conv;
        uint32_t similar = f.similar;
        auto gate = [&](int st) {  // the MIST counter shared by the three tests (:111-123)
            if (similar < c.max_similar) {
                ++similar;
                return false;
            }
            similar = 0;
            state = st;
            return true;
        };
        do {
            if (it >= c.max_iterations) {  // (:86-99)
                state = PCR_FIT_ITERATIONS;
                if (!c.failure_after_max_iterations) {
                    converged = true;
                    break;
                }
            }
            double rot = 0;  // (:103-133); T is the ACCUMULATED translation
            for (int i = 0; i < 9; ++i) rot += (f.pre_rot[i] - R[i]) * (f.pre_rot[i] - R[i]);
            const double tr = T[0] * T[0] + T[1] * T[1] + T[2] * T[2];
            if (rot < c.rotation_threshold && tr < c.translation_threshold && gate(PCR_FIT_TRANSFORM)) {
                converged = true;
                break;
            }
            cur_mse = mse;  // (:137-170)
            if (fabs(cur_mse - f.pre_mse) < c.abs_mse * f.uniq && gate(PCR_FIT_ABS_MSE)) {
                converged = true;
                break;
            }
            if (fabs(cur_mse - f.pre_mse) / f.pre_mse < c.rel_mse * f.uniq && gate(PCR_FIT_REL_MSE)) {  // (:174-207)
                converged = true;
                break;
            }
        } while (false);
        F->similar = similar;
        F->cur_mse = cur_mse;
        F->pre_mse = cur_mse;  // (:77-79)
        for (int i = 0; i < 9; ++i) F->pre_rot[i] = R[i];
    } else {
        F->cur_mse = mse;
    }
    F->state = state;
    // ---- next iteration's constants
    StepDyn next;
    const bool ok = make_dyn(R, T, f.c_src, f.c_tgt, f.cloud_scale, f.src_key_sc, f.src_key_delta, next, nullptr);
    if (!ok) {
        F->state = PCR_FIT_DEGENERATE;
        converged = true;
    }
    next.done = converged ? 1u : 0u;
    {
        // |p_new - p_old| = |dR (s - c) + (dR c + dT)| <= |dR|_F * rad + |dR c + dT| for every source point s
        double fro = 0, v[3];
        for (int i = 0; i < 9; ++i) fro += (R[i] - f.dyn.M.r[i]) * (R[i] - f.dyn.M.r[i]);
        for (int r = 0; r < 3; ++r)
            v[r] = (R[r] - f.dyn.M.r[r]) * f.c_src[0] + (R[3 + r] - f.dyn.M.r[3 + r]) * f.c_src[1] +
                   (R[6 + r] - f.dyn.M.r[6 + r]) * f.c_src[2] + (T[r] - f.dyn.M.t[r]);
        const double disp = sqrt(fro) * f.src_rad + sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        // slack: the roundings of both evaluations of R*s + T, and of this bound itself
        next.disp_bound = __double2float_ru(disp * 1.000001 + 1e-12 * (f.cloud_scale + fabs(T[0]) + fabs(T[1]) + fabs(T[2])));
    }
    F->dyn = next;
}

__global__ void fit_solve_kernel(FitDev *F, const double *__restrict__ partials, double *__restrict__ traj) {
    if (threadIdx.x == 0 && blockIdx.x == 0) fit_solve_body(F, partials, traj);
}

}  // namespace pcr

using namespace pcr;

struct pcr_fit {
    pcr_ctx *ctx;
    const pcr_index *src, *tgt;
    StepArgs args;
    uint32_t nb, kc, traj_cap;  // nb: TILES (64 samples each)
    FitDev *dev_d;        // device state
    FitDev *dev_h;        // pinned host mirror (pcr_fit_read)
    double *partials_d;   // [17]
    double *block_partials_d;  // StepArgs::tile_partials [17][nb]
    double *traj_d;       // [traj_cap][14]
    uint32_t *samples_d;  // owned copy of a host sample list (pcr_fit_set_samples)
    StepArgs *args_d;     // [2] device copies of args (what the filtered kernel reads): [1] also solves in the kernel
    double *group_partials_d;
    uint32_t *tickets_d;
    float *dk_d;          // [m] StepArgs::dk_io
    FitDev init;          // host-side template of the state (reset copies it)
    hipGraphExec_t graph_exec;  // pcr_fit_enqueue_iterations: `graph_count` iterations captured once, replayed
    uint32_t graph_count;
};
static void fit_drop_graph(pcr_fit *f) {
    if (f->graph_exec) hipGraphExecDestroy(f->graph_exec);
    f->graph_exec = nullptr;
    f->graph_count = 0;
}

// (re)allocates the in-kernel fold's scratch for f->nb blocks; tickets start at zero and every launch leaves them so
static hipError_t fit_alloc_fold(pcr_fit *f) {
    if (f->group_partials_d) hipFree(f->group_partials_d);
    if (f->tickets_d) hipFree(f->tickets_d);
    f->group_partials_d = nullptr;
    f->tickets_d = nullptr;
    const size_t ng = (f->nb + 63u) / 64u;
    hipError_t e = hipMalloc(&f->group_partials_d, sizeof(double) * 17 * (ng ? ng : 1));
    if (e == hipSuccess) e = hipMalloc(&f->tickets_d, sizeof(uint32_t) * (1 + ng));
    if (e == hipSuccess) e = hipMemsetAsync(f->tickets_d, 0, sizeof(uint32_t) * (1 + ng), f->ctx->stream);
    f->args.group_partials = f->group_partials_d;
    f->args.tickets = f->tickets_d;
    return e;
}

// the two device copies of the arguments: [0] the step folds, [1] the step folds and solves
static void fit_upload_args(pcr_fit *f) {
    StepArgs A = f->args;
    A.fit_inline = nullptr;
    A.fit_traj = nullptr;
    hipLaunchKernelGGL(set_args_kernel, dim3(1), dim3(256), 0, f->ctx->stream, A, f->args_d);
    A.fit_inline = f->dev_d;
    A.fit_traj = f->traj_d;
    hipLaunchKernelGGL(set_args_kernel, dim3(1), dim3(256), 0, f->ctx->stream, A, f->args_d + 1);
}

// args_d: a device copy of A (set_args_kernel), what the filtered kernel reads its arguments from
static thread_local const char *g_last_step_kernel = "none";

// Returns true when the launch also folded (and, if args_d says so, solved) in the step kernel itself.
// nt: tiles (64 samples each).  The filtered kernel runs one single-wave block per tile, every other step kernel
// ceil(nt / 4) blocks of four waves.
static bool launch_step_kernels(pcr_ctx *ctx, const pcr_index *src, const StepArgs &A, uint32_t nt, uint32_t k,
                                double *partials_d, const StepArgs *args_d) {
    const uint32_t nb = (nt + 3u) / 4u;
    // single-level grids with key-space shadows and K <= 8: the filtered kernel (tune 5: 1 = never, for A/B runs and tests)
    const bool fast = src->kind == PCR_INDEX_GRID && g_tune[5] != 1 && k <= 8 && A.src.levels == 1 && A.tgt.levels == 1 &&
                      A.src.v[0].shadow && A.tgt.v[0].shadow && args_d;
    if (fast) {
        g_last_step_kernel = k == 1 ? "icp_step_fast_kernel<1>" : k <= 4 ? "icp_step_fast_kernel<4>" : "icp_step_fast_kernel<8>";
        {
            ProfScope ps(ctx, PCR_PROF_STEP);
            if (k == 1)
                hipLaunchKernelGGL(icp_step_fast_kernel<1>, dim3(nt), dim3(64), 0, ctx->stream, args_d);
            else if (k <= 4)
                hipLaunchKernelGGL(icp_step_fast_kernel<4>, dim3(nt), dim3(64), 0, ctx->stream, args_d);
            else
                hipLaunchKernelGGL(icp_step_fast_kernel<8>, dim3(nt), dim3(64), 0, ctx->stream, args_d);
        }
        if (A.tickets) return true;  // folded (and, with fit_inline in *args_d, solved) by the step kernel's last wave
        ProfScope ps(ctx, PCR_PROF_FOLD);
        hipLaunchKernelGGL(fold_partials_kernel, dim3(17), dim3(256), 0, ctx->stream, (const double *)A.tile_partials, nt,
                           partials_d);
        return false;
    }
    bool compact = false;  // some level of either index carries a compact cell table
    for (int l = 0; l < A.src.levels; ++l) compact |= A.src.v[l].cbits != nullptr;
    for (int l = 0; l < A.tgt.levels; ++l) compact |= A.tgt.v[l].cbits != nullptr;
    g_last_step_kernel = src->kind != PCR_INDEX_GRID ? "icp_step_brute_kernel"
                         : (A.src.levels > 1 || A.tgt.levels > 1) ? (compact ? "icp_step_grid_kernel<multi-level, compact tables>" : "icp_step_grid_kernel<multi-level>")
                                                                  : "icp_step_grid_kernel";
#define LAUNCH_STEP(KC)                                                                               \
    do {                                                                                              \
        if (src->kind == PCR_INDEX_GRID && (A.src.levels > 1 || A.tgt.levels > 1) && compact)         \
            hipLaunchKernelGGL((icp_step_grid_kernel<KC, 2>), dim3(nb), dim3(256), 0, ctx->stream, A);  \
        else if (src->kind == PCR_INDEX_GRID && (A.src.levels > 1 || A.tgt.levels > 1))               \
            hipLaunchKernelGGL((icp_step_grid_kernel<KC, 1>), dim3(nb), dim3(256), 0, ctx->stream, A);  \
        else if (src->kind == PCR_INDEX_GRID)                                                         \
            hipLaunchKernelGGL((icp_step_grid_kernel<KC, 0>), dim3(nb), dim3(256), 0, ctx->stream, A); \
        else                                                                                          \
            hipLaunchKernelGGL(icp_step_brute_kernel<KC>, dim3(nb), dim3(256), 0, ctx->stream, A);    \
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
    if (!A.tickets) {
        ProfScope ps(ctx, PCR_PROF_FOLD);
        hipLaunchKernelGGL(fold_partials_kernel, dim3(17), dim3(256), 0, ctx->stream, (const double *)A.tile_partials, nt,
                           partials_d);
    }
    return false;
}

// argument checks and the iteration-independent part of StepArgs, shared by the single step and the fit loop
static int prepare_step_args(pcr_ctx *ctx, const pcr_index *src, const pcr_index *tgt, const uint32_t *sample_idx_d,
                             uint32_t sample_begin, uint32_t m, uint32_t k, double thr_fwd_sq, double thr_rev_sq,
                             uint32_t flags, StepArgs &A, const char *who) {
    PCR_REQUIRE(src->ctx->device == ctx->device && tgt->ctx->device == ctx->device, "%s: index lives on another device", who);
    PCR_REQUIRE(src->kind == tgt->kind, "%s: source and target index must be of the same kind", who);
    PCR_REQUIRE(k >= 1 && k <= tgt->n, "%s: k=%u out of range for %u target points", who, k, tgt->n);
    if (k > PCR_MAX_K) {
        set_last_error("%s: k=%u above PCR_MAX_K=%u", who, k, PCR_MAX_K);
        return PCR_E_UNSUPPORTED;
    }
    if (!sample_idx_d)
        PCR_REQUIRE((uint64_t)sample_begin + m <= src->n, "%s: sample range [%u,%u+%u) exceeds %u source points", who,
                    sample_begin, sample_begin, m, src->n);
    memset(&A, 0, sizeof A);
    A.src = src->view;
    A.tgt = tgt->view;
    A.src_mem = src->view_d;
    A.tgt_mem = tgt->view_d;
    A.src_pos_of = src->pos_of_d;
    A.src_order = src->order_d;
    A.sample_idx = sample_idx_d;
    A.sample_begin = sample_idx_d ? 0 : sample_begin;
    A.m = m;
    A.k = (int)k;
    A.thr_fwd = thr_fwd_sq;
    A.thr_rev = thr_rev_sq;
    A.exact = (flags & PCR_STEP_EXACT_INDICES) ? 1u : 0u;
    return PCR_OK;
}

extern "C" {

const char *pcr_last_step_kernel(void) { return g_last_step_kernel; }

int pcr_icp_pivots(const pcr_index *src, const pcr_index *tgt, const double R[9], const double T[3],
                   double pivot_p[3], double pivot_q[3]) {
    PCR_REQUIRE(src && tgt && R && T && pivot_p && pivot_q, "pcr_icp_pivots: NULL argument");
    double c_src[3], c_tgt[3];
    cloud_centre(src->cloud, c_src);
    cloud_centre(tgt->cloud, c_tgt);
    for (int r = 0; r < 3; ++r) {
        pivot_p[r] = R[r] * c_src[0] + R[3 + r] * c_src[1] + R[6 + r] * c_src[2] + T[r];
        pivot_q[r] = c_tgt[r];
    }
    return PCR_OK;
}

// device-side scratch of one step besides the block partials
struct StepScratch {
    StepDyn dyn;
    StepArgs args;
};
// the in-kernel fold's scratch for nb blocks follows a StepScratch: the tickets (padded to 8 bytes), then the group partials
static size_t fold_tickets_bytes(uint32_t nb) { return (sizeof(uint32_t) * (1 + (nb + 63u) / 64u) + 7) & ~(size_t)7; }
static size_t fold_scratch_bytes(uint32_t nb) { return fold_tickets_bytes(nb) + sizeof(double) * 17 * ((nb + 63u) / 64u); }

// What pcr_icp_step_d and pcr_icp_step_noalloc_d share: argument checks, the per-iteration constants, the launches.
// dyn_d and block_partials_d are the step's scratch (device memory beyond inputs and outputs); who owns them is the
// only difference between the two entry points.
struct NormalFit {  // the normal-aware variant's thresholds (pcr_icp_step_normal_fit); NULL: the ICRP step
    double nvdt, ndt;
    int compare_normals;
};
static int step_enqueue(pcr_ctx *ctx, const pcr_index *src, const pcr_index *tgt, const uint32_t *sample_idx_d,
                        uint32_t sample_begin, uint32_t m, const double R[9], const double T[3], uint32_t k,
                        double thr_fwd_sq, double thr_rev_sq, uint32_t flags, double *partials_d, uint32_t *out_fwd_d,
                        uint32_t *out_rev_d, uint8_t *out_keep_d, StepScratch *sc_d, double *group_partials_d,
                        uint32_t *tickets_d, double *block_partials_d, const char *who, const NormalFit *nf = nullptr) {
    StepArgs A;
    int rc = prepare_step_args(ctx, src, tgt, sample_idx_d, sample_begin, m, k, thr_fwd_sq, thr_rev_sq, flags, A, who);
    if (rc != PCR_OK) return rc;
    for (int i = 0; i < 9; ++i) PCR_REQUIRE(isfinite(R[i]), "%s: R is not finite", who);
    for (int i = 0; i < 3; ++i) PCR_REQUIRE(isfinite(T[i]), "%s: T is not finite", who);
    StepDyn D;
    double c_src[3], c_tgt[3], defect = 0;
    cloud_centre(src->cloud, c_src);
    cloud_centre(tgt->cloud, c_tgt);
    PCR_REQUIRE(make_dyn(R, T, c_src, c_tgt, clouds_scale(src->cloud, tgt->cloud), src->view.v[0].key_sc,
                         src->view.v[0].shadow ? src->view.v[0].key_delta : 0u, D, &defect),
                "%s: R must be a rotation to within 1e-3 (orthogonality defect %g)", who, defect);
    A.out_fwd = out_fwd_d;
    A.out_rev = out_rev_d;
    A.out_keep = out_keep_d;
    const uint32_t nb = (m + 63u) / 64u;  // tiles
    if (nb == 0) {
        PCR_HIP(hipMemsetAsync(partials_d, 0, sizeof(double) * 17, ctx->stream));
        return PCR_OK;
    }
    A.tile_partials = block_partials_d;
    A.dyn = &sc_d->dyn;
    A.group_partials = group_partials_d;
    A.tickets = tickets_d;  // zero here, and every launch leaves them so
    A.partials_out = partials_d;
    // the per-iteration values and the arguments travel as kernel arguments of tiny kernels: stream-ordered, no staging
    hipLaunchKernelGGL(set_dyn_kernel, dim3(1), dim3(1), 0, ctx->stream, D, &sc_d->dyn);
    if (nf) {
        A.src_normals = src->cloud->normals_d;
        A.tgt_normals = tgt->cloud->normals_d;
        A.nvdt = nf->nvdt;
        A.ndt = nf->ndt;
        A.compare_normals = nf->compare_normals ? 1u : 0u;
        const bool multi = A.src.levels > 1 || A.tgt.levels > 1;
        g_last_step_kernel = multi ? "icp_step_normal_kernel<multi-level>" : "icp_step_normal_kernel";
#define LAUNCH_NORMAL(KC)                                                                                  \
    do {                                                                                                   \
        if (multi)                                                                                         \
            hipLaunchKernelGGL((icp_step_normal_kernel<KC, true>), dim3((nb + 3u) / 4u), dim3(256), 0, ctx->stream, A);  \
        else                                                                                               \
            hipLaunchKernelGGL((icp_step_normal_kernel<KC, false>), dim3((nb + 3u) / 4u), dim3(256), 0, ctx->stream, A); \
    } while (0)
        ProfScope ps(ctx, PCR_PROF_STEP);
        if (k <= 4)
            LAUNCH_NORMAL(4);
        else if (k <= 8)
            LAUNCH_NORMAL(8);
        else
            LAUNCH_NORMAL(32);
#undef LAUNCH_NORMAL
        PCR_HIP(hipGetLastError());
        return PCR_OK;  // (folded by the kernel's last blocks: the tickets are always there on this path)
    }
    hipLaunchKernelGGL(set_args_kernel, dim3(1), dim3(256), 0, ctx->stream, A, &sc_d->args);
    launch_step_kernels(ctx, src, A, nb, k, partials_d, &sc_d->args);
    PCR_HIP(hipGetLastError());
    return PCR_OK;
}

static int step_d_impl(pcr_ctx *ctx, const pcr_index *src, const pcr_index *tgt, const uint32_t *sample_idx_d,
                       uint32_t sample_begin, uint32_t m, const double R[9], const double T[3], uint32_t k,
                       double thr_fwd_sq, double thr_rev_sq, uint32_t flags, double *partials_d, uint32_t *out_fwd_d,
                       uint32_t *out_rev_d, uint8_t *out_keep_d, const NormalFit *nf) {
    PCR_REQUIRE(ctx && src && tgt && R && T && partials_d, "pcr_icp_step_d: NULL argument");
    PCR_HIP(hipSetDevice(ctx->device));
    const uint32_t nb = (m + 63u) / 64u;  // tiles
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
    StepScratch *dyn_d = (StepScratch *)ctx_scratch(ctx, 5, sizeof(StepScratch) + fold_scratch_bytes(nb));
    if (!dyn_d) return PCR_E_NOMEM;
    // the scratch is laid out per call (the number of groups varies with m): the tickets are cleared per call
    uint32_t *tickets_d = (uint32_t *)(dyn_d + 1);
    double *gp_d = (double *)((char *)tickets_d + fold_tickets_bytes(nb));
    PCR_HIP(hipMemsetAsync(tickets_d, 0, fold_tickets_bytes(nb), ctx->stream));
    return step_enqueue(ctx, src, tgt, sample_idx_d, sample_begin, m, R, T, k, thr_fwd_sq, thr_rev_sq, flags, partials_d,
                        out_fwd_d, out_rev_d, out_keep_d, dyn_d, gp_d, tickets_d, ctx->block_partials_d,
                        nf ? "pcr_icp_step_normal_fit" : "pcr_icp_step_d", nf);
}

int pcr_icp_step_d(pcr_ctx *ctx, const pcr_index *src, const pcr_index *tgt, const uint32_t *sample_idx_d,
                   uint32_t sample_begin, uint32_t m, const double R[9], const double T[3], uint32_t k,
                   double thr_fwd_sq, double thr_rev_sq, uint32_t flags, double *partials_d, uint32_t *out_fwd_d,
                   uint32_t *out_rev_d, uint8_t *out_keep_d) {
    return step_d_impl(ctx, src, tgt, sample_idx_d, sample_begin, m, R, T, k, thr_fwd_sq, thr_rev_sq, flags, partials_d,
                       out_fwd_d, out_rev_d, out_keep_d, nullptr);
}

// ---- the entry point that allocates nothing (include/pcr.h: pcr_icp_step_noalloc_d) ------------------------------
// Its scratch -- the per-iteration constants and the block partials of the step -- lives in a file-scope device array
// of fixed size instead of context-owned allocations.  The array's NAME is a host-side stand-in, never a device
// address: the address is looked up once per device (noalloc_scratch_init, called by pcr_ctx_create with that device
// current) and only that pointer is ever handed to a kernel.  Every GPU the library runs on holds its own copy until
// the process ends.
struct NoAllocScratch {
    StepScratch step;
    double group_partials[17 * (size_t)(PCR_NOALLOC_MAX_SAMPLES / 64u / 64u)];
    uint32_t tickets[2 + PCR_NOALLOC_MAX_SAMPLES / 64u / 64u];  // cleared at the start of every call
    double block_partials[17 * (size_t)(PCR_NOALLOC_MAX_SAMPLES / 64u)];  // [17][tiles]
};
static __device__ __align__(256) unsigned char scratch[sizeof(NoAllocScratch)];
constexpr int NOALLOC_MAX_DEVICES = 64;
static NoAllocScratch *g_noalloc_ptr[NOALLOC_MAX_DEVICES];  // per device: that device's copy of `scratch`
static std::mutex g_noalloc_mutex;

}  // extern "C"
namespace pcr {
int noalloc_scratch_init(int device) {
    if (device < 0 || device >= NOALLOC_MAX_DEVICES) return PCR_OK;  // such a device simply has no no-alloc entry point
    // contexts may be created from several threads at once: one lookup per device, published under the lock
    std::lock_guard<std::mutex> lock(g_noalloc_mutex);
    if (g_noalloc_ptr[device]) return PCR_OK;
    void *p = nullptr;
    PCR_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));  // (that device is current: pcr_ctx_create set it)
    g_noalloc_ptr[device] = (NoAllocScratch *)p;
    return PCR_OK;
}
static NoAllocScratch *noalloc_scratch_of(int device) {
    if (device < 0 || device >= NOALLOC_MAX_DEVICES) return nullptr;
    std::lock_guard<std::mutex> lock(g_noalloc_mutex);
    return g_noalloc_ptr[device];
}
}  // namespace pcr
extern "C" {

int pcr_icp_step_noalloc_d(pcr_ctx *ctx, const pcr_index *src, const pcr_index *tgt, const uint32_t *sample_idx_d,
                           uint32_t sample_begin, uint32_t m, const double R[9], const double T[3], uint32_t k,
                           double thr_fwd_sq, double thr_rev_sq, uint32_t flags, double *partials_d,
                           uint32_t *out_fwd_d, uint32_t *out_rev_d, uint8_t *out_keep_d) {
    PCR_REQUIRE(ctx && src && tgt && R && T && partials_d, "pcr_icp_step_noalloc_d: NULL argument");
    PCR_REQUIRE(m <= PCR_NOALLOC_MAX_SAMPLES,
                "pcr_icp_step_noalloc_d: m=%u exceeds PCR_NOALLOC_MAX_SAMPLES=%u (the fixed scratch of this entry point); "
                "use pcr_icp_step_d, or split the samples into ranges and add the partials",
                m, PCR_NOALLOC_MAX_SAMPLES);
    NoAllocScratch *S = noalloc_scratch_of(ctx->device);
    PCR_REQUIRE(S, "pcr_icp_step_noalloc_d: no scratch on device %d (set up by pcr_ctx_create)", ctx->device);
    PCR_HIP(hipSetDevice(ctx->device));
    // The fold's tickets must start at zero.  Every launch that completes leaves them so -- but the array lives as long
    // as the process, and a launch that died half-way (or two calls that overlapped against the documented rule) would
    // otherwise poison every later call on this GPU: clear them here, stream-ordered, every call -- with a tiny kernel
    // (allocates nothing, and is an ordinary kernel node under stream capture).
    hipLaunchKernelGGL(fill_u32_kernel, dim3(1), dim3(256), 0, ctx->stream, S->tickets, (uint32_t)(sizeof S->tickets / 4), 0u);
    // (the context's event profiler creates its events on first use: not inside this call)
    const bool profiling = ctx->profiling;
    ctx->profiling = false;
    const int rc = step_enqueue(ctx, src, tgt, sample_idx_d, sample_begin, m, R, T, k, thr_fwd_sq, thr_rev_sq, flags,
                                partials_d, out_fwd_d, out_rev_d, out_keep_d, &S->step, S->group_partials, S->tickets,
                                S->block_partials, "pcr_icp_step_noalloc_d");
    ctx->profiling = profiling;
    return rc;
}

// test hook (not in include/pcr.h): leaves this GPU's fixed tickets non-zero, as a launch that died mid-fold would
int pcr_noalloc_debug_poison_tickets(pcr_ctx *ctx) {
    PCR_REQUIRE(ctx, "pcr_noalloc_debug_poison_tickets: NULL argument");
    NoAllocScratch *S = noalloc_scratch_of(ctx->device);
    PCR_REQUIRE(S, "pcr_noalloc_debug_poison_tickets: no scratch on device %d", ctx->device);
    PCR_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(fill_u32_kernel, dim3(1), dim3(256), 0, ctx->stream, S->tickets, (uint32_t)(sizeof S->tickets / 4), 0x01010101u);
    PCR_HIP(hipGetLastError());
    return PCR_OK;
}

static int step_host_impl(pcr_ctx *ctx, const pcr_index *src, const pcr_index *tgt, const uint32_t *sample_idx, uint32_t m,
                          const double R[9], const double T[3], uint32_t k, double thr_fwd_sq, double thr_rev_sq,
                          uint32_t flags, double partials[17], uint32_t *out_fwd, uint32_t *out_rev, uint8_t *out_keep,
                          const NormalFit *nf) {
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
        rc = step_d_impl(ctx, src, tgt, s_d, 0, m, R, T, k, thr_fwd_sq, thr_rev_sq, flags, ctx->result_d, f_d, r_d, k_d, nf);
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

int pcr_icp_step(pcr_ctx *ctx, const pcr_index *src, const pcr_index *tgt, const uint32_t *sample_idx, uint32_t m,
                 const double R[9], const double T[3], uint32_t k, double thr_fwd_sq, double thr_rev_sq,
                 uint32_t flags, double partials[17], uint32_t *out_fwd, uint32_t *out_rev, uint8_t *out_keep) {
    return step_host_impl(ctx, src, tgt, sample_idx, m, R, T, k, thr_fwd_sq, thr_rev_sq, flags, partials, out_fwd, out_rev,
                          out_keep, nullptr);
}

int pcr_icp_step_normal_fit(pcr_ctx *ctx, const pcr_index *src, const pcr_index *tgt, const uint32_t *sample_idx,
                            uint32_t m, const double R[9], const double T[3], uint32_t k, double nvdt, double ndt,
                            int compare_normals, double partials[17], uint32_t *out_tgt, uint8_t *out_keep) {
    PCR_REQUIRE(ctx && src && tgt && partials, "pcr_icp_step_normal_fit: NULL argument");
    PCR_REQUIRE(src->kind == PCR_INDEX_GRID && tgt->kind == PCR_INDEX_GRID, "pcr_icp_step_normal_fit: grid indices only");
    PCR_REQUIRE(tgt->cloud->normals_d, "pcr_icp_step_normal_fit: the target cloud has no normals (pcr_cloud_set_normals)");
    PCR_REQUIRE(!compare_normals || src->cloud->normals_d,
                "pcr_icp_step_normal_fit: comparing normals needs the source cloud's (pcr_cloud_set_normals)");
    PCR_REQUIRE(nvdt == nvdt && ndt == ndt, "pcr_icp_step_normal_fit: a threshold is NaN");
    const NormalFit nf = {nvdt, ndt, compare_normals};
    return step_host_impl(ctx, src, tgt, sample_idx, m, R, T, k, 0.0, 0.0, 0, partials, out_tgt, nullptr, out_keep, &nf);
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
    const uint32_t nb = (P + 63u) / 64u;  // tiles
    double *buf = nullptr;  // [3P | 3P | 17*nb | 17]
    const size_t n3 = (size_t)P * 3;
    if (hipMalloc(&buf, sizeof(double) * (2 * n3 + 17 * (size_t)nb + 17)) != hipSuccess) {
        set_last_error("pcr_reduce_pairs: device allocation for %u pairs failed", P);
        return PCR_E_NOMEM;
    }
    StepDyn D;
    memset(&D, 0, sizeof D);
    memcpy(D.pivot_p, pivot_p, sizeof D.pivot_p);
    memcpy(D.pivot_q, pivot_q, sizeof D.pivot_q);
    double *block_partials = buf + 2 * n3;
    double *res_d = block_partials + 17 * (size_t)nb;
    hipError_t e = hipMemcpyAsync(buf, pair_p, sizeof(double) * n3, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(buf + n3, pair_q, sizeof(double) * n3, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(reduce_pairs_kernel, dim3((nb + 3u) / 4u), dim3(256), 0, ctx->stream, (const double *)buf,
                           (const double *)(buf + n3), P, D, block_partials);
        hipLaunchKernelGGL(fold_partials_kernel, dim3(17), dim3(256), 0, ctx->stream, (const double *)block_partials, nb,
                           res_d);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(partials, res_d, sizeof(double) * 17, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    hipFree(buf);
    PCR_HIP(e);
    return PCR_OK;
}

// ---- the device-resident loop (CIterativeFit::__fitLoop, IterativeFit.cpp:119-175) ------------------------------

int pcr_fit_destroy(pcr_fit *f) {
    if (!f) return PCR_OK;
    hipSetDevice(f->ctx->device);
    hipStreamSynchronize(f->ctx->stream);
    fit_drop_graph(f);
    if (f->dev_d) hipFree(f->dev_d);
    if (f->dev_h) hipHostFree(f->dev_h);
    if (f->partials_d) hipFree(f->partials_d);
    if (f->block_partials_d) hipFree(f->block_partials_d);
    if (f->traj_d) hipFree(f->traj_d);
    if (f->samples_d) hipFree(f->samples_d);
    if (f->args_d) hipFree(f->args_d);
    if (f->group_partials_d) hipFree(f->group_partials_d);
    if (f->tickets_d) hipFree(f->tickets_d);
    if (f->dk_d) hipFree(f->dk_d);
    delete f;
    return PCR_OK;
}

int pcr_fit_create(pcr_ctx *ctx, const pcr_index *src, const pcr_index *tgt, const uint32_t *sample_idx_d,
                   uint32_t sample_begin, uint32_t m, uint32_t k, double thr_fwd_sq, double thr_rev_sq,
                   const pcr_convergence_params *conv, double unique_sq_dist, uint32_t traj_cap, pcr_fit **out) {
    PCR_REQUIRE(ctx && src && tgt && out, "pcr_fit_create: NULL argument");
    *out = nullptr;
    PCR_HIP(hipSetDevice(ctx->device));
    pcr_fit *f = new pcr_fit();
    f->ctx = ctx;
    f->src = src;
    f->tgt = tgt;
    int rc = prepare_step_args(ctx, src, tgt, sample_idx_d, sample_begin, m, k, thr_fwd_sq, thr_rev_sq, 0u, f->args,
                               "pcr_fit_create");
    if (rc != PCR_OK) {
        delete f;
        return rc;
    }
    f->nb = (m + 63u) / 64u;  // tiles
    f->kc = k;
    f->traj_cap = traj_cap;
    FitDev &I = f->init;
    memset(&I, 0, sizeof I);
    cloud_centre(src->cloud, I.c_src);
    cloud_centre(tgt->cloud, I.c_tgt);
    I.cloud_scale = clouds_scale(src->cloud, tgt->cloud);
    I.src_key_sc = src->view.v[0].shadow ? src->view.v[0].key_sc : 0.0;
    I.src_key_delta = src->view.v[0].shadow ? src->view.v[0].key_delta : 0u;
    {
        double d2 = 0;
        for (int a = 0; a < 3; ++a) d2 += (src->cloud->hi[a] - src->cloud->lo[a]) * (src->cloud->hi[a] - src->cloud->lo[a]);
        I.src_rad = 0.5 * sqrt(d2) * (1.0 + 1e-12);
    }
    I.uniq = unique_sq_dist;
    if (conv) {
        I.conv = *conv;
        I.conv_enabled = 1;
    }
    I.traj_cap = traj_cap;
    hipError_t e = hipMalloc(&f->dev_d, sizeof(FitDev));
    if (e == hipSuccess) e = hipHostMalloc(&f->dev_h, sizeof(FitDev));
    if (e == hipSuccess) e = hipMalloc(&f->partials_d, sizeof(double) * 32);
    if (e == hipSuccess) e = hipMalloc(&f->block_partials_d, sizeof(double) * 17 * (size_t)(f->nb ? f->nb : 1));
    if (e == hipSuccess && traj_cap) e = hipMalloc(&f->traj_d, sizeof(double) * 14 * (size_t)traj_cap);
    if (e == hipSuccess) e = hipMalloc(&f->args_d, sizeof(StepArgs) * 2);
    if (e == hipSuccess) e = fit_alloc_fold(f);
    if (e == hipSuccess && m) e = hipMalloc(&f->dk_d, sizeof(float) * (size_t)m);
    if (e != hipSuccess) {
        set_last_error("pcr_fit_create: allocation failed: %s", hipGetErrorString(e));
        pcr_fit_destroy(f);
        return e == hipErrorOutOfMemory ? PCR_E_NOMEM : PCR_E_HIP;
    }
    f->args.dyn = &f->dev_d->dyn;
    f->args.tile_partials = f->block_partials_d;
    f->args.dk_io = f->dk_d;
    f->args.partials_out = f->partials_d;
    fit_upload_args(f);
    const double eye[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, zero[3] = {0, 0, 0};
    rc = pcr_fit_reset(f, eye, zero);
    if (rc != PCR_OK) {
        pcr_fit_destroy(f);
        return rc;
    }
    *out = f;
    return PCR_OK;
}

int pcr_fit_set_samples(pcr_fit *f, const uint32_t *sample_idx, uint32_t m) {
    PCR_REQUIRE(f && (sample_idx || m == 0), "pcr_fit_set_samples: NULL argument");
    for (uint32_t i = 0; i < m; ++i)
        PCR_REQUIRE(sample_idx[i] < f->src->n, "pcr_fit_set_samples: sample_idx[%u]=%u out of range", i, sample_idx[i]);
    PCR_HIP(hipSetDevice(f->ctx->device));
    PCR_HIP(hipStreamSynchronize(f->ctx->stream));  // nothing in flight may still read the old list or partials
    fit_drop_graph(f);                              // (a captured launch holds the old grid size)
    // allocate everything the new sample set needs FIRST; the fit changes only once all of it exists (a failure leaves
    // it exactly as it was, its device-side arguments still pointing at live memory)
    const uint32_t nb = (m + 63u) / 64u;  // tiles
    const size_t ng = (nb + 63u) / 64u;
    uint32_t *samples_new = nullptr, *tickets_new = nullptr;
    double *bp_new = nullptr, *gp_new = nullptr;
    float *dk_new = nullptr;
    hipError_t e = hipSuccess;
    if (nb > f->nb) e = hipMalloc(&bp_new, sizeof(double) * 17 * (size_t)nb);
    if (e == hipSuccess && m) e = hipMalloc(&samples_new, sizeof(uint32_t) * (size_t)m);
    if (e == hipSuccess && m) e = hipMalloc(&dk_new, sizeof(float) * (size_t)m);
    if (e == hipSuccess) e = hipMalloc(&gp_new, sizeof(double) * 17 * (ng ? ng : 1));
    if (e == hipSuccess) e = hipMalloc(&tickets_new, sizeof(uint32_t) * (1 + ng));
    if (e == hipSuccess && m) e = hipMemcpy(samples_new, sample_idx, sizeof(uint32_t) * (size_t)m, hipMemcpyHostToDevice);
    if (e == hipSuccess && m) e = hipMemsetD32Async((hipDeviceptr_t)dk_new, 0x7f800000, m, f->ctx->stream);  // +inf: nothing known
    if (e == hipSuccess) e = hipMemsetAsync(tickets_new, 0, sizeof(uint32_t) * (1 + ng), f->ctx->stream);
    if (e != hipSuccess) {
        if (bp_new) hipFree(bp_new);
        if (samples_new) hipFree(samples_new);
        if (dk_new) hipFree(dk_new);
        if (gp_new) hipFree(gp_new);
        if (tickets_new) hipFree(tickets_new);
        PCR_HIP(e);
    }
    if (bp_new) {
        hipFree(f->block_partials_d);
        f->block_partials_d = bp_new;
        f->args.tile_partials = bp_new;
    }
    if (f->samples_d) hipFree(f->samples_d);
    if (f->dk_d) hipFree(f->dk_d);
    if (f->group_partials_d) hipFree(f->group_partials_d);
    if (f->tickets_d) hipFree(f->tickets_d);
    f->samples_d = samples_new;
    f->dk_d = dk_new;
    f->group_partials_d = gp_new;
    f->tickets_d = tickets_new;
    f->args.group_partials = gp_new;
    f->args.tickets = tickets_new;
    f->args.dk_io = f->dk_d;
    f->args.sample_idx = f->samples_d;
    f->args.sample_begin = 0;
    f->args.m = m;
    f->nb = nb;
    fit_upload_args(f);
    PCR_HIP(hipGetLastError());
    return PCR_OK;
}

int pcr_fit_set_debug_outputs(pcr_fit *f, uint32_t *out_fwd_d, uint32_t *out_rev_d, uint8_t *out_keep_d) {
    PCR_REQUIRE(f, "pcr_fit_set_debug_outputs: NULL argument");
    PCR_HIP(hipSetDevice(f->ctx->device));
    fit_drop_graph(f);  // (a captured launch of a kernel that takes its arguments by value holds the old pointers)
    f->args.out_fwd = out_fwd_d;
    f->args.out_rev = out_rev_d;
    f->args.out_keep = out_keep_d;
    fit_upload_args(f);  // stream-ordered: iterations already enqueued keep what they were enqueued with
    PCR_HIP(hipGetLastError());
    return PCR_OK;
}

int pcr_fit_reset(pcr_fit *f, const double R[9], const double T[3]) {
    PCR_REQUIRE(f && R && T, "pcr_fit_reset: NULL argument");
    for (int i = 0; i < 9; ++i) PCR_REQUIRE(isfinite(R[i]), "pcr_fit_reset: R is not finite");
    for (int i = 0; i < 3; ++i) PCR_REQUIRE(isfinite(T[i]), "pcr_fit_reset: T is not finite");
    PCR_HIP(hipSetDevice(f->ctx->device));
    FitDev I = f->init;
    double defect = 0;
    PCR_REQUIRE(make_dyn(R, T, I.c_src, I.c_tgt, I.cloud_scale, I.src_key_sc, I.src_key_delta, I.dyn, &defect),
                "pcr_fit_reset: R must be a rotation to within 1e-3 (orthogonality defect %g)", defect);
    // CDefaultConvergenceCriteria::__init (DefaultConvergenceCriterial.cpp:21-25); m_PreRot is uninitialised in the
    // reference: NaN makes the first transformation test false, as garbage far from a rotation does
    I.pre_mse = I.cur_mse = 1.7976931348623157e308;
    for (int i = 0; i < 9; ++i) I.pre_rot[i] = __builtin_nan("");
    I.last_mse = __builtin_nan("");
    I.similar = 0;
    I.iteration = 0;
    I.state = PCR_FIT_NOT_CONVERGED;
    if (f->dk_d)  // +inf: the first iteration knows nothing about its neighbour distances
        PCR_HIP(hipMemsetD32Async((hipDeviceptr_t)f->dk_d, 0x7f800000, f->args.m, f->ctx->stream));
    hipLaunchKernelGGL(fit_init_kernel, dim3(1), dim3(1), 0, f->ctx->stream, I, f->dev_d);
    PCR_HIP(hipGetLastError());
    return PCR_OK;
}

int pcr_fit_enqueue_step(pcr_fit *f) {
    PCR_REQUIRE(f, "pcr_fit_enqueue_step: NULL argument");
    PCR_HIP(hipSetDevice(f->ctx->device));
    if (f->nb == 0) {
        PCR_HIP(hipMemsetAsync(f->partials_d, 0, sizeof(double) * 17, f->ctx->stream));
        return PCR_OK;
    }
    launch_step_kernels(f->ctx, f->src, f->args, f->nb, f->kc, f->partials_d, f->args_d);
    PCR_HIP(hipGetLastError());
    return PCR_OK;
}

struct pcr_fit_batch {
    pcr_ctx *ctx;
    uint32_t count, blocks, kc;
    const StepArgs **argv_d;
    uint32_t *first_d;
};

int pcr_fit_batch_destroy(pcr_fit_batch *b) {
    if (!b) return PCR_OK;
    hipSetDevice(b->ctx->device);
    hipStreamSynchronize(b->ctx->stream);
    if (b->argv_d) hipFree(b->argv_d);
    if (b->first_d) hipFree(b->first_d);
    delete b;
    return PCR_OK;
}

int pcr_fit_batch_create(pcr_ctx *ctx, pcr_fit *const *fits, uint32_t count, pcr_fit_batch **out) {
    PCR_REQUIRE(ctx && fits && out && count > 0, "pcr_fit_batch_create: NULL or empty argument");
    *out = nullptr;
    std::vector<const StepArgs *> argv(count);
    std::vector<uint32_t> first(count + 1, 0);
    // the kernel instance a fit's K selects (the same three the single-fit launch picks from)
    auto k_class = [](uint32_t k) { return k == 1 ? 1u : k <= 4 ? 4u : 8u; };
    const uint32_t kc = fits[0] ? k_class(fits[0]->kc) : 0u;
    for (uint32_t i = 0; i < count; ++i) {
        const pcr_fit *f = fits[i];
        PCR_REQUIRE(f, "pcr_fit_batch_create: fits[%u] is NULL", i);
        PCR_REQUIRE(f->ctx == ctx, "pcr_fit_batch_create: fits[%u] was created on another context", i);
        for (uint32_t j = 0; j < i; ++j)
            PCR_REQUIRE(fits[j] != f, "pcr_fit_batch_create: fits[%u] and fits[%u] are the same fit", j, i);
        const bool fast = f->src->kind == PCR_INDEX_GRID && f->kc <= 8 && f->args.src.levels == 1 && f->args.tgt.levels == 1 &&
                          f->args.src.v[0].shadow && f->args.tgt.v[0].shadow;
        if (!fast || k_class(f->kc) != kc) {
            set_last_error("pcr_fit_batch_create: fits[%u] cannot join this batch (needs single-level grid indices, K <= 8, "
                           "and every fit's K in the same class: 1, 2..4 or 5..8)", i);
            return PCR_E_UNSUPPORTED;
        }
        argv[i] = f->args_d + 1;  // the copy whose step also solves
        first[i + 1] = first[i] + f->nb;
    }
    PCR_REQUIRE(first[count] > 0, "pcr_fit_batch_create: no samples in any fit");
    PCR_HIP(hipSetDevice(ctx->device));
    pcr_fit_batch *b = new pcr_fit_batch();
    b->ctx = ctx;
    b->count = count;
    b->blocks = first[count];
    b->kc = kc;
    b->argv_d = nullptr;
    b->first_d = nullptr;
    hipError_t e = hipMalloc(&b->argv_d, sizeof(StepArgs *) * count);
    if (e == hipSuccess) e = hipMalloc(&b->first_d, sizeof(uint32_t) * (count + 1));
    if (e == hipSuccess) e = hipMemcpy(b->argv_d, argv.data(), sizeof(StepArgs *) * count, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(b->first_d, first.data(), sizeof(uint32_t) * (count + 1), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        set_last_error("pcr_fit_batch_create: %s", hipGetErrorString(e));
        pcr_fit_batch_destroy(b);
        return e == hipErrorOutOfMemory ? PCR_E_NOMEM : PCR_E_HIP;
    }
    *out = b;
    return PCR_OK;
}

int pcr_fit_batch_enqueue_iteration(pcr_fit_batch *b) {
    PCR_REQUIRE(b, "pcr_fit_batch_enqueue_iteration: NULL argument");
    PCR_HIP(hipSetDevice(b->ctx->device));
    g_last_step_kernel = b->kc == 1 ? "icp_step_batch_kernel<1>" : b->kc == 4 ? "icp_step_batch_kernel<4>" : "icp_step_batch_kernel<8>";
    ProfScope ps(b->ctx, PCR_PROF_STEP);
    if (b->kc == 1)
        hipLaunchKernelGGL(icp_step_batch_kernel<1>, dim3(b->blocks), dim3(64), 0, b->ctx->stream, b->argv_d,
                           (const uint32_t *)b->first_d, b->count);
    else if (b->kc == 4)
        hipLaunchKernelGGL(icp_step_batch_kernel<4>, dim3(b->blocks), dim3(64), 0, b->ctx->stream, b->argv_d,
                           (const uint32_t *)b->first_d, b->count);
    else
        hipLaunchKernelGGL(icp_step_batch_kernel<8>, dim3(b->blocks), dim3(64), 0, b->ctx->stream, b->argv_d,
                           (const uint32_t *)b->first_d, b->count);
    PCR_HIP(hipGetLastError());
    return PCR_OK;
}

int pcr_fit_enqueue_iteration(pcr_fit *f) {
    PCR_REQUIRE(f, "pcr_fit_enqueue_iteration: NULL argument");
    PCR_HIP(hipSetDevice(f->ctx->device));
    bool solved = false;
    if (f->nb == 0)
        PCR_HIP(hipMemsetAsync(f->partials_d, 0, sizeof(double) * 17, f->ctx->stream));
    else
        solved = launch_step_kernels(f->ctx, f->src, f->args, f->nb, f->kc, f->partials_d, f->args_d + 1);
    if (!solved)
        hipLaunchKernelGGL(fit_solve_kernel, dim3(1), dim3(64), 0, f->ctx->stream, f->dev_d, (const double *)f->partials_d,
                           f->traj_d);
    PCR_HIP(hipGetLastError());
    return PCR_OK;
}

int pcr_fit_enqueue_iterations(pcr_fit *f, uint32_t count) {
    PCR_REQUIRE(f, "pcr_fit_enqueue_iterations: NULL argument");
    if (count == 0) return PCR_OK;
    PCR_HIP(hipSetDevice(f->ctx->device));
    // (event brackets are per launch: no graph while kernels are being timed; the legacy default stream cannot capture)
    if (count == 1 || f->ctx->profiling || f->ctx->stream == nullptr) {
        for (uint32_t i = 0; i < count; ++i) {
            const int rc = pcr_fit_enqueue_iteration(f);
            if (rc != PCR_OK) return rc;
        }
        return PCR_OK;
    }
    if (!f->graph_exec || f->graph_count != count) {
        // capture `count` iterations once: no kernel argument depends on the iteration (the state lives in device
        // memory), so the same graph serves every later call with this count
        fit_drop_graph(f);
        hipGraph_t graph = nullptr;
        PCR_HIP(hipStreamBeginCapture(f->ctx->stream, hipStreamCaptureModeThreadLocal));
        int rc = PCR_OK;
        for (uint32_t i = 0; i < count && rc == PCR_OK; ++i) rc = pcr_fit_enqueue_iteration(f);
        const hipError_t e = hipStreamEndCapture(f->ctx->stream, &graph);
        if (rc != PCR_OK || e != hipSuccess || !graph) {
            if (graph) hipGraphDestroy(graph);
            if (rc != PCR_OK) return rc;
            PCR_HIP(e != hipSuccess ? e : hipErrorUnknown);
        }
        const hipError_t ei = hipGraphInstantiate(&f->graph_exec, graph, nullptr, nullptr, 0);
        hipGraphDestroy(graph);
        if (ei != hipSuccess) {
            f->graph_exec = nullptr;
            PCR_HIP(ei);
        }
        f->graph_count = count;
    }
    PCR_HIP(hipGraphLaunch(f->graph_exec, f->ctx->stream));
    return PCR_OK;
}

double *pcr_fit_partials_d(pcr_fit *f) { return f ? f->partials_d : nullptr; }

int pcr_fit_enqueue_solve(pcr_fit *f) {
    PCR_REQUIRE(f, "pcr_fit_enqueue_solve: NULL argument");
    PCR_HIP(hipSetDevice(f->ctx->device));
    hipLaunchKernelGGL(fit_solve_kernel, dim3(1), dim3(64), 0, f->ctx->stream, f->dev_d, (const double *)f->partials_d,
                       f->traj_d);
    PCR_HIP(hipGetLastError());
    return PCR_OK;
}

int pcr_fit_read(pcr_fit *f, double R[9], double T[3], uint32_t *iterations, int *state, double *mse, double *pairs,
                 double *traj, uint32_t traj_rows) {
    PCR_REQUIRE(f, "pcr_fit_read: NULL argument");
    PCR_HIP(hipSetDevice(f->ctx->device));
    PCR_HIP(hipMemcpyAsync(f->dev_h, f->dev_d, sizeof(FitDev), hipMemcpyDeviceToHost, f->ctx->stream));
    PCR_HIP(hipStreamSynchronize(f->ctx->stream));
    const FitDev &H = *f->dev_h;
    if (R) memcpy(R, H.dyn.M.r, sizeof(double) * 9);
    if (T) memcpy(T, H.dyn.M.t, sizeof(double) * 3);
    if (iterations) *iterations = H.iteration;
    if (state) *state = H.state;
    if (mse) *mse = H.last_mse;
    if (pairs) *pairs = H.last_n;
    if (traj && traj_rows && f->traj_d) {
        uint32_t rows = H.iteration < f->traj_cap ? H.iteration : f->traj_cap;
        if (rows > traj_rows) rows = traj_rows;
        if (rows) PCR_HIP(hipMemcpy(traj, f->traj_d, sizeof(double) * 14 * rows, hipMemcpyDeviceToHost));
    }
    return PCR_OK;
}

}  // extern "C"

#ifdef PCR_BLOCK_TIMES
extern "C" int pcr_diag_solve_times(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(pcr::g_solve_times), sizeof(unsigned long long) * 4) == hipSuccess ? 0 : -2;
}
extern "C" int pcr_diag_fast_clocks(unsigned *out, unsigned n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(pcr::g_fast_clocks), sizeof(unsigned) * 16 * n) == hipSuccess ? 0 : -2;
}
extern "C" int pcr_diag_fast_times(unsigned long long *out, unsigned n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(pcr::g_fast_times), sizeof(unsigned long long) * 4 * n) == hipSuccess ? 0 : -2;
}
extern "C" int pcr_diag_block_times(unsigned long long *out, unsigned n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(pcr::g_block_times), sizeof(unsigned long long) * 2 * n) == hipSuccess ? 0 : -2;
}
#endif
#ifdef PCR_DIAG
extern "C" int pcr_diag_read_trips(unsigned char *out /* [2][1 << 20] */) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(pcr::g_trips), 2u << 20) == hipSuccess ? 0 : -2;
}
extern "C" int pcr_diag_read_icp(unsigned long long out[48], int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(pcr::g_diag), sizeof(unsigned long long) * 48) != hipSuccess) return -2;
    if (reset) {
        unsigned long long z[48] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(pcr::g_diag), z, sizeof z) != hipSuccess) return -2;
    }
    return 0;
}
#endif
