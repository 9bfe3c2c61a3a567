// Wave-cooperative exact K-NN on the uniform grid -- gfx950, 64-wide wavefronts.
//
// The 64 lanes of a wave hold 64 spatially coherent queries (the callers order queries by grid
// brick).  Instead of every lane gathering its own 27 cells from global memory, the WAVE takes the
// bounding box of its queries' cells (+1 cell of halo), stages every point of that box ONCE through
// LDS -- each (y,z) row of the box is one contiguous run of the cell-sorted point array -- and all
// lanes then scan the staged tile with broadcast LDS reads (every lane reads the same slot: no bank
// conflict, no divergence; the only per-lane work is the fp64 distance and the top-K insert).
// If a lane's K-th best is not yet provably final, the whole wave grows the box by one cell in
// every direction and stages only the new shell.  Termination per lane is the same conservative
// bound as in grid_search (knn_device.h), taken against the wave's box.
#pragma once
#include "knn_device.h"

namespace pcr {

#ifdef PCR_DIAG
// diagnostic build only (libpcr_diag.so): where a wave's cycles go.  Never compiled into libpcr.so.
static __device__ unsigned long long g_diag[16];
#define DIAG_T() ((unsigned long long)__builtin_amdgcn_s_memtime())
#define DIAG_ADD(slot, v)                                                                      \
    do {                                                                                       \
        if ((threadIdx.x & 63) == 0) atomicAdd(&g_diag[slot], (unsigned long long)(v));       \
    } while (0)
#else
#define DIAG_T() 0ull
#define DIAG_ADD(slot, v) \
    do {                  \
    } while (0)
#endif

constexpr int WAVE_TILE_CAP = 256;        // candidates staged per pass: 8 KiB of LDS per wave
constexpr int WAVE_MAX_ROWS = 1024;       // beyond this the wave's queries are not coherent: per-lane search
constexpr uint32_t WAVE_MAX_CAND = 2048;  // likewise when the first box holds more candidates than this

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

// LDS traffic inside one wave: a wave's DS operations are issued and completed in program order, so
// no hardware barrier is needed; the fence only stops the compiler from moving accesses across it.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// tile: this wave's private LDS region of WAVE_TILE_CAP points; seg_incl/seg_beg: 64 words each, also
// private to the wave.  `active` lanes hold a query; the
// others only help with staging.  Arguments otherwise as grid_search.
template <int KC>
__device__ __forceinline__ void wave_search(const GridView &g, const Pt4 *__restrict__ cand, bool active, double qx,
                                            double qy, double qz, double sx, double sy, double sz, double max_d2,
                                            double bound_scale, double extra_margin, TopK<KC> &top, Pt4 *tile,
                                            uint32_t *seg_incl, uint32_t *seg_beg, int halo0 = 1,
                                            bool per_lane = false) {
    const uint32_t *__restrict__ cs = g.cs;
    const int lane = threadIdx.x & 63;
    const int gx = g.g[0], gy = g.g[1], gz = g.g[2];
    const int cx = cell_coord(sx, g.lo[0], g.inv_h, gx);
    const int cy = cell_coord(sy, g.lo[1], g.inv_h, gy);
    const int cz = cell_coord(sz, g.lo[2], g.inv_h, gz);
    const int big = 0x3fffffff;
    int bx0 = wave_min_i(active ? cx : big), bx1 = wave_max_i(active ? cx : -big);
    int by0 = wave_min_i(active ? cy : big), by1 = wave_max_i(active ? cy : -big);
    int bz0 = wave_min_i(active ? cz : big), bz1 = wave_max_i(active ? cz : -big);
    if (bx1 < bx0) return;  // no active lane in this wave

    const double margin = g.margin + extra_margin + 9.094947017729282e-13 * (fabs(sx) + fabs(sy) + fabs(sz));
    double ox = fmax(fmax(g.lo[0] - sx, sx - g.hi[0]), 0.0) - margin;
    double oy = fmax(fmax(g.lo[1] - sy, sy - g.hi[1]), 0.0) - margin;
    double oz = fmax(fmax(g.lo[2] - sz, sz - g.hi[2]), 0.0) - margin;
    ox = ox > 0.0 ? ox : 0.0;
    oy = oy > 0.0 ? oy : 0.0;
    oz = oz > 0.0 ? oz : 0.0;
    const double ox2 = ox * ox, oy2 = oy * oy, oz2 = oz * oz;

    // not coherent enough for a shared tile (too many rows, or too many candidates in the first box):
    // every lane searches for itself
    bool coherent = !per_lane && (long long)(by1 - by0 + 1 + 2 * halo0) * (bz1 - bz0 + 1 + 2 * halo0) <= WAVE_MAX_ROWS;
    if (coherent) {
        const int x0 = max(bx0 - halo0, 0), x1 = min(bx1 + halo0, gx - 1);
        const int y0 = max(by0 - halo0, 0), y1 = min(by1 + halo0, gy - 1);
        const int z0 = max(bz0 - halo0, 0), z1 = min(bz1 + halo0, gz - 1);
        const int ny = y1 - y0 + 1, nrows = ny * (z1 - z0 + 1);
        uint32_t mine = 0;
        for (int row = lane; row < nrows; row += 64) {
            const size_t base = ((size_t)(z0 + row / ny) * gy + (y0 + row % ny)) * gx;
            mine += cs[base + x1 + 1] - cs[base + x0];
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off, 64);
        coherent = mine <= WAVE_MAX_CAND;
    }
    if (!coherent) {
        DIAG_ADD(1, 1);
        if (active) grid_search<KC>(g, cand, qx, qy, qz, sx, sy, sz, max_d2, bound_scale, extra_margin, top);
        return;
    }
    DIAG_ADD(0, 1);
    const unsigned long long t_begin = DIAG_T();
    unsigned long long t_lookup = 0, t_stage = 0, t_scan = 0;

    double wd = top.worst_d();
    uint32_t wi = top.worst_i();
    int px0 = 1, px1 = 0, py0 = 1, py1 = 0, pz0 = 1, pz1 = 0;  // box already staged (empty)
    const int rmax = max(max(gx, gy), gz);
    for (int grow = halo0; grow <= rmax; ++grow) {
        const int x0 = max(bx0 - grow, 0), x1 = min(bx1 + grow, gx - 1);
        const int y0 = max(by0 - grow, 0), y1 = min(by1 + grow, gy - 1);
        const int z0 = max(bz0 - grow, 0), z1 = min(bz1 + grow, gz - 1);
        const int ny = y1 - y0 + 1;
        const int nrows = ny * (z1 - z0 + 1);
        const bool have_prev = px1 >= px0;
        const int nitems = have_prev ? 2 * nrows : nrows;  // rows inside the old y/z range: two end segments
        DIAG_ADD(2, 1);
        for (int item0 = 0; item0 < nitems; item0 += 64) {
            DIAG_ADD(3, 1);
            const unsigned long long t0 = DIAG_T();
            // ---- each lane looks up one row segment of the new shell ----
            const int item = item0 + lane;
            uint32_t beg = 0, cnt = 0;
            if (item < nitems) {
                const int row = have_prev ? item >> 1 : item;
                const int seg = have_prev ? item & 1 : 0;
                const int ry = y0 + row % ny, rz = z0 + row / ny;
                const bool inner = have_prev & (ry >= py0) & (ry <= py1) & (rz >= pz0) & (rz <= pz1);
                int xs, xe;
                if (!inner) {
                    xs = x0;
                    xe = seg == 0 ? x1 : x0 - 1;  // whole row once
                } else if (seg == 0) {
                    xs = x0;
                    xe = px0 - 1;
                } else {
                    xs = px1 + 1;
                    xe = x1;
                }
                if (xs <= xe) {
                    const size_t base = ((size_t)rz * gy + ry) * gx;
                    beg = cs[base + xs];
                    cnt = cs[base + xe + 1] - beg;
                }
            }
            // inclusive prefix sum of cnt over the wave
            uint32_t incl = cnt;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t v = __shfl_up(incl, off, 64);
                if (lane >= off) incl += v;
            }
            const uint32_t total = __shfl(incl, 63, 64);
            t_lookup += DIAG_T() - t0;
            if (total == 0) continue;
            DIAG_ADD(4, total);
            // publish the segment table of this pass: flat candidate f lies in the first segment whose
            // inclusive prefix exceeds f
            seg_incl[lane] = incl;
            seg_beg[lane] = beg;
            wave_lds_sync();
            // ---- stage and scan, WAVE_TILE_CAP candidates at a time ----
            for (uint32_t base = 0; base < total; base += WAVE_TILE_CAP) {
                const int m = (int)min((uint32_t)WAVE_TILE_CAP, total - base);
                const unsigned long long t1 = DIAG_T();
                // flat, coalesced staging: lane l copies candidates l, l+64, ... of this window; the loads of
                // the (up to 4) iterations are independent
#pragma unroll
                for (int u = 0; u < WAVE_TILE_CAP / 64; ++u) {
                    const int j = u * 64 + lane;
                    if (j < m) {
                        const uint32_t f = base + (uint32_t)j;
                        int lo = 0, hi = 63;
#pragma unroll
                        for (int step = 0; step < 6; ++step) {
                            const int mid = (lo + hi) >> 1;
                            if (seg_incl[mid] > f)
                                hi = mid;
                            else
                                lo = mid + 1;
                        }
                        const uint32_t before = lo ? seg_incl[lo - 1] : 0u;
                        const uint32_t p = seg_beg[lo] + (f - before);
                        Pt4 c = cand[p];
                        c.w = (c.w & 0xffffffffull) | ((unsigned long long)p << 32);  // carry the position
                        tile[j] = c;
                    }
                }
                wave_lds_sync();
                const unsigned long long t2 = DIAG_T();
                t_stage += t2 - t1;
                if (active) {
#pragma unroll 4
                    for (int j = 0; j < m; ++j) {
                        const Pt4 c = tile[j];
                        const double d = sqdist3(qx, qy, qz, c.x, c.y, c.z);
                        const uint32_t ci = (uint32_t)c.w;
                        if (better(d, ci, wd, wi)) {
                            top.insert(d, ci, (uint32_t)(c.w >> 32));
                            wd = top.worst_d();
                            wi = top.worst_i();
                        }
                    }
                }
                wave_lds_sync();
                t_scan += DIAG_T() - t2;
            }
        }
        const bool all = (x0 == 0) & (x1 == gx - 1) & (y0 == 0) & (y1 == gy - 1) & (z0 == 0) & (z1 == gz - 1);
        if (all) break;
        bool done = true;
        if (active) {
            const double b2 = box_bound2(g, sx, sy, sz, margin, ox2, oy2, oz2, x0, x1, y0, y1, z0, z1) * bound_scale;
            done = (wd < b2) | (max_d2 <= b2);
        }
        if (!__any(!done)) break;
        DIAG_ADD(9, __popcll(__ballot(!done)));
        px0 = x0;
        px1 = x1;
        py0 = y0;
        py1 = y1;
        pz0 = z0;
        pz1 = z1;
    }
    DIAG_ADD(5, t_lookup);
    DIAG_ADD(6, t_stage);
    DIAG_ADD(7, t_scan);
    DIAG_ADD(8, DIAG_T() - t_begin);
}

}  // namespace pcr
