// Wave-staged, lane-scanned exact K-NN on the uniform grid -- gfx950, 64-wide wavefronts.
//
// The 64 lanes of a wave hold 64 spatially coherent queries (callers order queries by the Morton key
// of their grid brick).  The wave takes the bounding box of its queries' cells plus one cell of halo
// and stages it ONCE through LDS with coalesced loads:
//   * the points of the box (each (y,z) row of the box is one contiguous run of the cell-sorted array),
//   * a small cell table (start offset of every cell of the box inside the staged tile).
// Every lane then walks the 3x3x3 cells around ITS OWN query out of LDS (~100-cycle latency, no TA
// serialisation) instead of gathering from global memory (~1-2 k cycles under load).  The nine (y,z)
// rows are visited centre first, and a row -- or the outer cell of a row -- is skipped as soon as its
// slab is provably farther than the current K-th best, so a lane reads only the handful of cells its
// answer can come from.
// Lanes whose K-th best is not provably final after ring 1 (rare when cells hold ~4+ points) continue
// with ring 2.. on the global-memory path (grid_search with r_start = 2: disjoint shells, no duplicates).
// A wave whose box does not fit the tile is split into halves, then quarters (smaller boxes); only a
// quarter that still does not fit (incoherent queries) falls back to the per-lane global search.
#pragma once
#include "knn_device.h"

namespace pcr {

#ifdef PCR_DIAG
// diagnostic build only (libpcr_diag.so): where a wave's cycles go.  Never compiled into libpcr.so.
static __device__ unsigned long long g_diag[16];
#define DIAG_T() ((unsigned long long)__builtin_amdgcn_s_memtime())
#define DIAG_ADD(slot, v)                                                                \
    do {                                                                                 \
        if ((threadIdx.x & 63) == 0) atomicAdd(&g_diag[slot], (unsigned long long)(v)); \
    } while (0)
#else
#define DIAG_T() 0ull
#define DIAG_ADD(slot, v) \
    do {                  \
    } while (0)
#endif

constexpr int TILE_CAP = 512;    // points staged per wave (16 KiB)
constexpr int TILE_ROWS = 128;   // (y,z) rows of the box
constexpr int TILE_CELLS = 1024; // entries of the cell table: rows * (nx + 1)

// one wave's private LDS
struct WaveTile {
    Pt4 pts[TILE_CAP];               // w = original index (low 32 bits)
    uint32_t row_beg[TILE_ROWS];     // position of the row's first point in the global array
    uint16_t row_off[TILE_ROWS + 2]; // offset of the row's first point in pts[]; [nrows] = total
    uint16_t cell[TILE_CELLS];       // offset in pts[] of every cell's first point, row-major [row][nx+1]
};

__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
    return v;
}

// LDS traffic inside one wave: a wave's DS operations are issued and completed in program order, so no
// hardware barrier is needed; the fence only stops the compiler from moving accesses across it.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// What a lane needs to know about its own query for the ring-1 scan.
struct LaneQuery {
    double qx, qy, qz;  // query in the candidates' frame (distances)
    double sx, sy, sz;  // query in the grid's frame (cells, bounds)
    int cx, cy, cz;     // its cell
};

// Stage the box of the lanes in `grp` (a wave-uniform subset of the wave's lanes: all, a half or a
// quarter) and scan ring 1 for them.  Returns false (wave-uniform) if the box does not fit the tile.
template <int KC>
__device__ __forceinline__ bool tile_try(const GridView &g, const Pt4 *__restrict__ cand, bool in_grp,
                                         const LaneQuery &Q, double max_d2, double bound_scale, double margin,
                                         TopK<KC> &top, WaveTile &T, bool &done) {
    const uint32_t *__restrict__ cs = g.cs;
    const int lane = threadIdx.x & 63;
    const int gx = g.g[0], gy = g.g[1], gz = g.g[2];
    const int big = 0x3fffffff;
    const int bx0 = wave_min_i(in_grp ? Q.cx : big), bx1 = wave_max_i(in_grp ? Q.cx : -big);
    const int by0 = wave_min_i(in_grp ? Q.cy : big), by1 = wave_max_i(in_grp ? Q.cy : -big);
    const int bz0 = wave_min_i(in_grp ? Q.cz : big), bz1 = wave_max_i(in_grp ? Q.cz : -big);
    if (bx1 < bx0) return true;  // nobody in this group
    const unsigned long long t_begin = DIAG_T();

    // the box: bounding box of the group's cells + 1 cell of halo, clamped to the grid
    const int x0 = max(bx0 - 1, 0), x1 = min(bx1 + 1, gx - 1);
    const int y0 = max(by0 - 1, 0), y1 = min(by1 + 1, gy - 1);
    const int z0 = max(bz0 - 1, 0), z1 = min(bz1 + 1, gz - 1);
    const int nx1 = x1 - x0 + 2;  // cell-table entries per row (nx + 1)
    const int ny = y1 - y0 + 1;
    const long long nrows_ll = (long long)ny * (z1 - z0 + 1);
    if (nrows_ll > TILE_ROWS || nrows_ll * nx1 > TILE_CELLS) return false;
    const int nrows = (int)nrows_ll;

    // ---- pass 1: one lane per row -> row extents, running offsets, total ----
    uint32_t total = 0;
    for (int r0 = 0; r0 < nrows; r0 += 64) {
        const int row = r0 + lane;
        uint32_t beg = 0, cnt = 0;
        if (row < nrows) {
            const size_t base = ((size_t)(z0 + row / ny) * gy + (y0 + row % ny)) * gx;
            beg = cs[base + x0];
            cnt = cs[base + x1 + 1] - beg;
        }
        uint32_t incl = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t v = __shfl_up(incl, off, 64);
            if (lane >= off) incl += v;
        }
        if (row < nrows) {
            T.row_beg[row] = beg;
            T.row_off[row] = (uint16_t)min(total + incl - cnt, 65535u);
        }
        total += __shfl(incl, 63, 64);
        if (total > TILE_CAP) return false;
    }
    if (lane == 0) T.row_off[nrows] = (uint16_t)total;
    wave_lds_sync();
    const unsigned long long t_rows = DIAG_T();

    // ---- pass 2: cell table (coalesced along x) and points (flat over the lanes).  Each lane first
    // resolves all its slots, then issues all its loads, then stores: independent chains overlap. ----
    const int ncell = nrows * nx1;
    for (int e0 = 0; e0 < ncell; e0 += 256) {
        uint32_t v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + u * 64 + lane;
            v[u] = 0;
            if (e < ncell) {
                const int row = e / nx1, xi = e - row * nx1;
                const size_t base = ((size_t)(z0 + row / ny) * gy + (y0 + row % ny)) * gx;
                v[u] = T.row_off[row] + (cs[base + x0 + xi] - T.row_beg[row]);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + u * 64 + lane;
            if (e < ncell) T.cell[e] = (uint16_t)v[u];
        }
    }
    {
        uint32_t p[TILE_CAP / 64];
#pragma unroll
        for (int u = 0; u < TILE_CAP / 64; ++u) {
            const uint32_t j = (uint32_t)(u * 64 + lane);
            p[u] = 0;
            if (j < total) {
                int lo = 0, hi = nrows - 1;  // last row whose offset is <= j
#pragma unroll
                for (int step = 0; step < 7; ++step) {  // TILE_ROWS = 2^7
                    const int mid = (lo + hi + 1) >> 1;
                    const bool le = T.row_off[mid] <= j;
                    lo = le ? mid : lo;
                    hi = le ? hi : mid - 1;
                }
                p[u] = T.row_beg[lo] + (j - T.row_off[lo]);
            }
        }
        Pt4 c[TILE_CAP / 64];
#pragma unroll
        for (int u = 0; u < TILE_CAP / 64; ++u) c[u] = cand[p[u]];  // slots beyond total read element 0 (unused)
#pragma unroll
        for (int u = 0; u < TILE_CAP / 64; ++u)
            if ((uint32_t)(u * 64 + lane) < total) T.pts[u * 64 + lane] = c[u];
    }
    wave_lds_sync();
    const unsigned long long t_stage = DIAG_T();

    // ---- ring 1 out of LDS: the 3x3x3 cells around this lane's own query, centre row first ----
    if (in_grp) {
        double wd = top.worst_d();
        uint32_t wi = top.worst_i();
        const int cx = Q.cx, cy = Q.cy, cz = Q.cz;
        // squared distance from the query to the slab of cells at offset -1 / 0 / +1 along each axis,
        // shrunk by the safety margin and the bound scale: a slab may be skipped once wd < that value
        double gl, gh;
        gl = Q.sx - (g.lo[0] + (double)cx * g.h) - margin;
        gh = (g.lo[0] + (double)(cx + 1) * g.h) - Q.sx - margin;
        gl = gl > 0.0 ? gl : 0.0;
        gh = gh > 0.0 ? gh : 0.0;
        const double gxl = gl * gl * bound_scale, gxh = gh * gh * bound_scale;
        gl = Q.sy - (g.lo[1] + (double)cy * g.h) - margin;
        gh = (g.lo[1] + (double)(cy + 1) * g.h) - Q.sy - margin;
        gl = gl > 0.0 ? gl : 0.0;
        gh = gh > 0.0 ? gh : 0.0;
        const double gyl = gl * gl * bound_scale, gyh = gh * gh * bound_scale;
        gl = Q.sz - (g.lo[2] + (double)cz * g.h) - margin;
        gh = (g.lo[2] + (double)(cz + 1) * g.h) - Q.sz - margin;
        gl = gl > 0.0 ? gl : 0.0;
        gh = gh > 0.0 ? gh : 0.0;
        const double gzl = gl * gl * bound_scale, gzh = gh * gh * bound_scale;

        // rows in order of increasing slab distance class: centre, 4 faces, 4 edges; (dy+1) | (dz+1) << 2
        // packed 4 bits each
        const unsigned long long order = 0x5ull | (0x4ull << 4) | (0x6ull << 8) | (0x1ull << 12) | (0x9ull << 16) |
                                         (0x0ull << 20) | (0x2ull << 24) | (0x8ull << 28) | (0xaull << 32);
        // One loop, straight-line body: (a) a lane that has run out of candidates opens its next row
        // (at most one per trip; the row may turn out empty or skipped), (b) a lane that has a candidate
        // evaluates it.  Trips per lane = candidates + rows that were empty or skipped.
        int ri = 0;
        uint32_t p = 0, e = 0;
        bool alive = true;
        while (alive) {
            if ((p >= e) & (ri < 9)) {
                const int code = (int)(order >> (4 * ri)) & 15;
                ++ri;
                const int dy = (code & 3) - 1, dz = (code >> 2) - 1;
                const int ry = cy + dy, rz = cz + dz;
                const double gy2 = dy < 0 ? gyl : (dy > 0 ? gyh : 0.0);
                const double gz2 = dz < 0 ? gzl : (dz > 0 ? gzh : 0.0);
                const double gyz = gy2 + gz2;
                const bool open = (ry >= 0) & (ry < gy) & (rz >= 0) & (rz < gz) & !(wd < gyz);
                const int xs = (cx > 0) & !(wd < gyz + gxl) ? cx - 1 : cx;
                const int xe = (cx < gx - 1) & !(wd < gyz + gxh) ? cx + 1 : cx;
                const int trow = open ? ((rz - z0) * ny + (ry - y0)) * nx1 - x0 : 0;
                const uint32_t np = T.cell[open ? trow + xs : 0];
                const uint32_t ne = T.cell[open ? trow + xe + 1 : 0];
                p = open ? np : 0u;
                e = open ? ne : 0u;
            }
            if (p < e) {
                const Pt4 c = T.pts[p];
                ++p;
                const double d = sqdist3(Q.qx, Q.qy, Q.qz, c.x, c.y, c.z);
                const uint32_t ci = (uint32_t)c.w;
                if (better(d, ci, wd, wi)) {
                    top.insert(d, ci);
                    wd = top.worst_d();
                    wi = top.worst_i();
                }
            }
            alive = (p < e) | (ri < 9);
        }
        const int lx0 = max(cx - 1, 0), lx1 = min(cx + 1, gx - 1);
        const int ly0 = max(cy - 1, 0), ly1 = min(cy + 1, gy - 1);
        const int lz0 = max(cz - 1, 0), lz1 = min(cz + 1, gz - 1);
        const bool all = (lx0 == 0) & (lx1 == gx - 1) & (ly0 == 0) & (ly1 == gy - 1) & (lz0 == 0) & (lz1 == gz - 1);
        done = true;
        if (!all) {
            double ox = fmax(fmax(g.lo[0] - Q.sx, Q.sx - g.hi[0]), 0.0) - margin;
            double oy = fmax(fmax(g.lo[1] - Q.sy, Q.sy - g.hi[1]), 0.0) - margin;
            double oz = fmax(fmax(g.lo[2] - Q.sz, Q.sz - g.hi[2]), 0.0) - margin;
            ox = ox > 0.0 ? ox : 0.0;
            oy = oy > 0.0 ? oy : 0.0;
            oz = oz > 0.0 ? oz : 0.0;
            const double b2 = box_bound2(g, Q.sx, Q.sy, Q.sz, margin, ox * ox, oy * oy, oz * oz, lx0, lx1, ly0, ly1,
                                         lz0, lz1) *
                              bound_scale;
            done = (wd < b2) | (max_d2 <= b2);
        }
    }
    wave_lds_sync();  // the tile is reused by the next group / the caller's next search
    DIAG_ADD(0, 1);
    DIAG_ADD(2, nrows);
    DIAG_ADD(4, total);
    DIAG_ADD(5, t_rows - t_begin);
    DIAG_ADD(6, t_stage - t_rows);
    DIAG_ADD(7, DIAG_T() - t_stage);
    return true;
}

// `active` lanes hold a query; the others only help with staging.  Other arguments as grid_search.
template <int KC>
__device__ __forceinline__ void tile_search(const GridView &g, const Pt4 *__restrict__ cand, bool active, double qx,
                                            double qy, double qz, double sx, double sy, double sz, double max_d2,
                                            double bound_scale, double extra_margin, TopK<KC> &top, WaveTile &T,
                                            bool per_lane) {
    const int lane = threadIdx.x & 63;
    LaneQuery Q;
    Q.qx = qx;
    Q.qy = qy;
    Q.qz = qz;
    Q.sx = sx;
    Q.sy = sy;
    Q.sz = sz;
    Q.cx = cell_coord(sx, g.lo[0], g.inv_h, g.g[0]);
    Q.cy = cell_coord(sy, g.lo[1], g.inv_h, g.g[1]);
    Q.cz = cell_coord(sz, g.lo[2], g.inv_h, g.g[2]);
    const double margin = g.margin + extra_margin + 9.094947017729282e-13 * (fabs(sx) + fabs(sy) + fabs(sz));
    if (!__any(active)) return;
    if (per_lane) {  // tuning knob: skip the tile entirely
        if (active) lane_search<KC>(g, cand, qx, qy, qz, sx, sy, sz, max_d2, bound_scale, extra_margin, top);
        return;
    }
    const unsigned long long t_begin = DIAG_T();
    bool done = true;        // ring 1 settled this lane (only meaningful for active lanes)
    bool searched = false;   // this lane went through a tile
    if (!per_lane) {
        if (tile_try<KC>(g, cand, active, Q, max_d2, bound_scale, margin, top, T, done)) {
            searched = true;
        } else {
#pragma nounroll
            for (int h = 0; h < 2; ++h) {
                const bool in_h = active & ((lane >> 5) == h);
                if (tile_try<KC>(g, cand, in_h, Q, max_d2, bound_scale, margin, top, T, done)) {
                    searched |= in_h;
                } else {
#pragma nounroll
                    for (int q4 = 0; q4 < 2; ++q4) {
                        const bool in_q = in_h & (((lane >> 4) & 1) == q4);
                        if (tile_try<KC>(g, cand, in_q, Q, max_d2, bound_scale, margin, top, T, done)) searched |= in_q;
                    }
                }
            }
        }
    }
    const unsigned long long t_tiles = DIAG_T();
    // lanes that never fitted a tile search from scratch; lanes ring 1 could not settle continue at ring 2
    if (active & !searched) {
        lane_search<KC>(g, cand, qx, qy, qz, sx, sy, sz, max_d2, bound_scale, extra_margin, top);
    } else if (active & !done) {
        grid_search<KC>(g, cand, qx, qy, qz, sx, sy, sz, max_d2, bound_scale, extra_margin, top, 2);
    }
    DIAG_ADD(1, __popcll(__ballot(active & !searched)));
    DIAG_ADD(9, __popcll(__ballot(active & searched & !done)));
    DIAG_ADD(8, DIAG_T() - t_begin);
    DIAG_ADD(10, DIAG_T() - t_tiles);
    DIAG_ADD(11, 1);
}

}  // namespace pcr
