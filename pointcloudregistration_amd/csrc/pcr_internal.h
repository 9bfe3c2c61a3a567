// Internal declarations shared by the translation units of libpcr.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/pcr.h"

#define PCR_INF_H __builtin_huge_val()

namespace pcr {

// tuning knobs (pcr_tune, include/pcr.h)
extern int g_tune[8];

void set_last_error(const char *fmt, ...);

#define PCR_HIP(call)                                                                          \
    do {                                                                                       \
        hipError_t e__ = (call);                                                               \
        if (e__ != hipSuccess) {                                                               \
            ::pcr::set_last_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__),      \
                                  __FILE__, __LINE__);                                         \
            return e__ == hipErrorOutOfMemory ? PCR_E_NOMEM : PCR_E_HIP;                       \
        }                                                                                      \
    } while (0)

#define PCR_REQUIRE(cond, ...)                 \
    do {                                       \
        if (!(cond)) {                         \
            ::pcr::set_last_error(__VA_ARGS__); \
            return PCR_E_INVALID;              \
        }                                      \
    } while (0)

// One point of a cell-sorted cloud: x, y, z and the ORIGINAL index in the low 32 bits of w.
// 32 bytes, so a candidate is two 16-byte loads from one aligned line segment.
struct __attribute__((aligned(32))) Pt4 {
    double x, y, z;
    unsigned long long w;
};

// What the query kernels need to know about an index (passed by value).
struct GridView {
    const Pt4 *pts;       // [n] in cell order (GRID) or in original order (BRUTE)
    const uint32_t *cs;   // [cells+1] exclusive cell starts; NULL for BRUTE and for a level with a COMPACT table:
    // ---- compact cell table (sparse levels of nested indices: a LiDAR sweep's fine levels occupy under 1 % of their
    // cells, and a dense table is gigabytes of zeros' worth of repeated starts).  cbits[w] = {occupancy bits of cells
    // 32w .. 32w+31, number of occupied cells before cell 32w}; cstart[j] = start of the j-th occupied cell, [occupied] = n
    // (three more copies of n behind it).  The dense table's entry c is cstart[rank(c)], rank(c) = the occupied cells
    // before c: one 8-byte load, a popcount, one dependent 4-byte load; an empty range is known after the first.
    const uint2 *cbits;       // [cells / 32 + 2]; NULL: the level has the dense table
    const uint32_t *cstart;   // [occupied + 4]
    const uint32_t *cs8;  // [g[0]*gy8*gz8 + 1] the same for STRIPS of 8x8 (y,z) rows, x at full resolution: the
                          // points of cells x0..x1 of a strip number cs8[b+x1+1] - cs8[b+x0] -- zero lets a search
                          // skip 64 rows with two loads.  NULL for BRUTE
    int gy8;              // strips along y = ceil(g[1] / 8)
    double lo[3], hi[3];  // bounds of the points
    double h, inv_h;      // cell edge
    double margin;        // absolute safety margin on face distances, ~2^-40 * coordinate scale
    int g[3];             // grid dimensions
    uint32_t n;
    // ---- integer "key space" copy of the points for the filtered search (fast_search.h); NULL when the index has none
    // (brute force, grids beyond 2^30 cells).  One key-space unit is h / 21844 (KEY_CELL): a cell's edge is an INTEGER
    // number of units and 1.5 cells fit 15 bits.  shadow[i] = {X, Y16 | Z16 << 16} with (X, Y, Z) = (pts[i] - lo) *
    // key_sc rounded to the nearest integer, X whole, Y and Z modulo 2^16 -- 8 bytes per point.  A candidate's squared
    // distance in these units, from ONE packed 16-bit subtract (wrapping: exact while |dy|, |dz| < 1.5 h), one 16-bit dot
    // product and one 24-bit multiply-add, is a 32-bit integer KEY that orders candidates almost like the exact fp64
    // distance does; key_delta bounds |key - exact| (both roundings to integers, the truncation of the key's low bits)
    // for candidates less than 1.5 h away along y and z -- a farther one wraps to a key that is TOO SMALL, which the
    // walk detects when it evaluates the candidate exactly.  The exact fp64 arithmetic decides every result.
    const uint2 *shadow;
    double key_sc, key_S, key_invS;  // key = key_S * d2, key_S = key_sc^2, key_sc = KEY_CELL / h
    uint32_t key_delta;
    float key_eps;                   // what every fp32 gap of the walk's geometry is shrunk by, in root units
    float key_ext[3];                // (hi - lo) * key_sc, rounded up
};

// An index is a small stack of such grids over DISJOINT subsets of the cloud ("levels"): level 0 spans the whole
// cloud, every further level covers the box where the level before it is much denser than elsewhere, with a cell
// size of its own (a LiDAR sweep's density falls by orders of magnitude with range; one cell size cannot fit
// that).  All levels share one point array (each owns a contiguous segment), one pos_of table and one transformed
// copy.  A uniform cloud has one level.  A search visits the levels whose box can still hold a better point, carrying
// its top-K from one to the next -- exact, because the union of exact searches over a partition is exact.
constexpr int PCR_MAX_LEVELS = 6;
struct MultiView {
    int levels;
    uint32_t n;              // all points
    const Pt4 *all;          // [n] every level's segment, level 0 first
    const uint32_t *pos_of;  // [n] original index -> position in `all`
    double lo[3], hi[3];     // bounds of the whole cloud
    GridView v[PCR_MAX_LEVELS];
};

}  // namespace pcr

// kernel classes timed by the optional profiling hooks (pcr_ctx_set_profiling)
enum { PCR_PROF_TRANSFORM = 0 /* (no longer launched: kept so that the class numbers stay) */, PCR_PROF_STEP = 1, PCR_PROF_FOLD = 2, PCR_PROF_KNN = 3, PCR_PROF_CLASSES = 4 };
constexpr int PCR_PROF_CAP = 1024;  // launches recorded per class between two reads

struct pcr_prof {
    hipEvent_t start[PCR_PROF_CLASSES][PCR_PROF_CAP];
    hipEvent_t stop[PCR_PROF_CLASSES][PCR_PROF_CAP];
    int created[PCR_PROF_CLASSES];
    int used[PCR_PROF_CLASSES];
};

struct pcr_ctx {
    int device;
    bool profiling;
    unsigned prof_mask;  // bit c: launches of kernel class c are bracketed while profiling is on
    unsigned prof_stride;                  // every prof_stride-th launch of a class is bracketed (0, 1: every launch)
    unsigned prof_seq[4];                  // launches seen per class since profiling was switched on
    pcr_prof *prof;
    hipStream_t stream;
    bool own_stream;
    int cu_count;  // compute units of the device (launch shapes)
    // reusable device scratch for block partials of the fused step (grown on demand, never per call
    // once warm) and a pinned host staging area for the 17 results.
    double *block_partials_d;
    size_t block_partials_cap;  // in doubles
    double *result_d;           // [32] device
    double *result_h;           // [32] pinned host
    // grow-only device scratch slots (query ordering, temporaries): never freed before the context dies,
    // so steady-state calls allocate nothing
    void *scratch_d[6];
    size_t scratch_cap[6];
};

namespace pcr {
// RAII bracket: records a start/stop event pair around a launch when profiling is on.
struct ProfScope {
    pcr_ctx *c;
    int cls, slot;
    ProfScope(pcr_ctx *ctx, int k) : c(ctx), cls(k), slot(-1) {
        if (!c->profiling || !c->prof || !((c->prof_mask >> cls) & 1u)) return;
        if (c->prof_stride > 1 && (c->prof_seq[cls]++ % c->prof_stride) != 0) return;
        pcr_prof &p = *c->prof;
        if (p.used[cls] >= PCR_PROF_CAP) return;
        if (p.used[cls] >= p.created[cls]) {
            if (hipEventCreate(&p.start[cls][p.created[cls]]) != hipSuccess) return;
            if (hipEventCreate(&p.stop[cls][p.created[cls]]) != hipSuccess) return;
            ++p.created[cls];
        }
        slot = p.used[cls]++;
        (void)hipEventRecord(p.start[cls][slot], c->stream);
    }
    ~ProfScope() {
        if (slot >= 0) (void)hipEventRecord(c->prof->stop[cls][slot], c->stream);
    }
};
}  // namespace pcr

namespace pcr {
// looks up this device's copy of the fixed scratch of pcr_icp_step_noalloc_d (pcr_icp.hip); the device must be current
int noalloc_scratch_init(int device);
// returns ctx->scratch_d[slot] grown to at least `bytes` (NULL + last error on failure)
void *ctx_scratch(pcr_ctx *ctx, int slot, size_t bytes);
// positions [0,n) ordered by the Morton key of the (>=4x4x4-cell) brick their point falls in; exactly one of
// pts / xyz_aos is non-NULL.  deterministic: order inside a brick made ascending (index build only).
// The positions written are base + i (a level's segment of a multi-level index starts at `base`).
int brick_order(pcr_ctx *ctx, const GridView &g, const Pt4 *pts, const double *xyz_aos, uint32_t n,
                uint32_t *order_d, bool deterministic, uint32_t base = 0);
// cells above this many points are ordered by the segmented sort (pcr_sort.hip), the others by one thread each
#define PCR_HEAVY_CELL 32u
// orders the points of every heavy cell by (x, original index); tmp is scratch of the same size as pts
int sort_heavy_cells(pcr_ctx *ctx, const uint32_t *cs_d, uint64_t cells, uint32_t n, Pt4 *pts, Pt4 *tmp);
// positions [0,nq) of a query batch grouped by the top bits of a Morton key (fast, not deterministic inside a group)
int coarse_query_order(pcr_ctx *ctx, const GridView &g, const double *q_d, uint32_t nq, uint32_t *order_d);
}  // namespace pcr

struct pcr_cloud {
    pcr_ctx *ctx;
    uint32_t n;
    const double *xyz_d;  // AoS [3n]
    bool owned;
    double lo[3], hi[3];
    const double *normals_d;  // AoS [3n], original order; NULL unless attached (pcr_cloud_set_normals: the normal-aware variant)
    bool normals_owned;
};

struct pcr_index {
    pcr_ctx *ctx;
    const pcr_cloud *cloud;
    int kind;  // PCR_INDEX_GRID or PCR_INDEX_BRUTE
    uint32_t n;
    pcr::Pt4 *pts_d;    // [n] cell order (GRID) / original order (BRUTE)
    uint2 *shadow_d;    // [n] same order: integer key-space copy (GridView::shadow)
    uint32_t *pos_of_d; // [n] original index -> position in pts_d
    uint32_t *order_d;  // [n] positions in pts_d ordered by grid brick: the processing order of "all samples"
    pcr::MultiView view;
    pcr::MultiView *view_d;  // device copy of `view` (the filtered K-NN kernel reads the grid's description from there)
    uint32_t *cs_d[pcr::PCR_MAX_LEVELS];    // [cells+1] of every level, positions relative to the level's segment
    uint32_t *cs_raw_d[pcr::PCR_MAX_LEVELS];  // the allocations behind cs_d: cs_d - 1, [cells+3] (a zero entry either side)
    uint64_t cells[pcr::PCR_MAX_LEVELS];
    uint32_t *cs8_d[pcr::PCR_MAX_LEVELS];   // strip tables (GridView::cs8)
    uint2 *cbits_d[pcr::PCR_MAX_LEVELS];    // compact tables (GridView::cbits, cstart); cs_d[l] is NULL where a level has one
    uint32_t *cstart_d[pcr::PCR_MAX_LEVELS];
    uint64_t occupied[pcr::PCR_MAX_LEVELS];  // occupied cells of every level
    pcr_index_info info;
};
