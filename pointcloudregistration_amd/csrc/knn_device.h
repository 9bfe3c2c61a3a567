// Device-side exact K-NN on the uniform grid (and the brute-force tile scan) -- gfx950.
//
// Exactness contract: the result is the K smallest candidates under the total order
// (d2, original index), d2 = ((qx-x)^2 + (qy-y)^2) + (qz-z)^2 evaluated in that order
// without FMA contraction (Common/KNNSearch.h:137-143 of the reference).  The ring search
// stops only when a lower bound on the distance of every UNVISITED point, made conservative
// against every rounding in cell assignment and in the bound itself, exceeds the K-th best.
#pragma once
#include "pcr_internal.h"

namespace pcr {

#define PCR_INF __builtin_huge_val()

// fp64 product and sum that the compiler may not fuse (the whole library is also built with
// -ffp-contract=off; these keep the hot expressions safe if a flag is ever lost).
__device__ __forceinline__ double dmul(double a, double b) { return __dmul_rn(a, b); }
__device__ __forceinline__ double dadd(double a, double b) { return __dadd_rn(a, b); }

// Common/KNNSearch.h:137-143: Result = 0; Result += Diff*Diff for k = 0,1,2
__device__ __forceinline__ double sqdist3(double qx, double qy, double qz, double x, double y, double z) {
    const double dx = qx - x, dy = qy - y, dz = qz - z;
    return dadd(dadd(dmul(dx, dx), dmul(dy, dy)), dmul(dz, dz));
}

// PointCloud.cpp:368-370: ((r0*x + r3*y) + r6*z) + t0 on column-major R
struct Rigid {
    double r[9], t[3];
};
__device__ __forceinline__ void xform(const Rigid &M, double x, double y, double z, double &ox, double &oy,
                                      double &oz) {
    ox = dadd(dadd(dadd(dmul(M.r[0], x), dmul(M.r[3], y)), dmul(M.r[6], z)), M.t[0]);
    oy = dadd(dadd(dadd(dmul(M.r[1], x), dmul(M.r[4], y)), dmul(M.r[7], z)), M.t[1]);
    oz = dadd(dadd(dadd(dmul(M.r[2], x), dmul(M.r[5], y)), dmul(M.r[8], z)), M.t[2]);
}

__device__ __forceinline__ bool better(double d, uint32_t i, double bd, uint32_t bi) {
    return d < bd || (d == bd && i < bi);
}

// Sorted top-K in registers.  Every access is statically indexed (all loops over KC unroll), so the
// arrays stay in VGPRs; a search always fills all KC slots -- the K <= KC nearest a caller asked for
// are the first K of them (same total order), so no run-time K reaches the hot loop.
// pos = position of the candidate in the array it was read from (to fetch coordinates later).
template <int KC>
struct TopK {
    double d[KC];
    uint32_t idx[KC];
    uint32_t pos[KC];
    __device__ __forceinline__ void init(double d0 = PCR_INF) {
#pragma unroll
        for (int j = 0; j < KC; ++j) {
            d[j] = d0;
            idx[j] = PCR_NO_INDEX;
            pos[j] = PCR_NO_INDEX;
        }
    }
    __device__ __forceinline__ double worst_d() const { return d[KC - 1]; }
    __device__ __forceinline__ uint32_t worst_i() const { return idx[KC - 1]; }
    // Slot k-1 for a run-time k, WITHOUT a dynamic register index (which would push the arrays to
    // scratch memory): OR together the slots under an all-ones/all-zeros mask.
    __device__ __forceinline__ void slot(int k, double &od, uint32_t &oi, uint32_t &op) const {
        unsigned long long db = 0;
        uint32_t ib = 0, pb = 0;
#pragma unroll
        for (int j = 0; j < KC; ++j) {
            const uint32_t m = (uint32_t)0 - (uint32_t)(j == k - 1);
            const unsigned long long m64 = ((unsigned long long)m << 32) | m;
            db |= (unsigned long long)__double_as_longlong(d[j]) & m64;
            ib |= idx[j] & m;
            pb |= pos[j] & m;
        }
        od = __longlong_as_double((long long)db);
        oi = ib;
        op = pb;
    }
    // insert (dist, i, p) keeping the slots sorted ascending by (d, idx); caller has checked that the
    // candidate beats slot KC-1.
    __device__ __forceinline__ void insert(double dist, uint32_t i, uint32_t p) {
#pragma unroll
        for (int j = KC - 1; j >= 1; --j) {
            const bool shift = better(dist, i, d[j - 1], idx[j - 1]);
            const bool here = !shift && better(dist, i, d[j], idx[j]);
            d[j] = shift ? d[j - 1] : (here ? dist : d[j]);
            idx[j] = shift ? idx[j - 1] : (here ? i : idx[j]);
            pos[j] = shift ? pos[j - 1] : (here ? p : pos[j]);
        }
        if (better(dist, i, d[0], idx[0])) {
            d[0] = dist;
            idx[0] = i;
            pos[0] = p;
        }
    }
};

__device__ __forceinline__ int cell_coord(double v, double lo, double inv_h, int g) {
    double t = (v - lo) * inv_h;
    t = t < 0.0 ? 0.0 : t;  // also catches NaN -> false -> keeps t; handled by the caller's finite check
    const double gm = (double)(g - 1);
    t = t > gm ? gm : t;
    return (int)t;
}

// Squared lower bound on the distance from the grid-frame query s to every point whose cell lies
// OUTSIDE the cell box [x0..x1]x[y0..y1]x[z0..z1] (faces on the grid boundary have nothing beyond
// them).  Conservative against rounding: each face gap is shrunk by `margin`; o?2 are the squared
// per-axis distances of s to the bounds of the points (0 inside), already shrunk the same way.
__device__ __forceinline__ double box_bound2(const GridView &g, double sx, double sy, double sz, double margin,
                                             double ox2, double oy2, double oz2, int x0, int x1, int y0, int y1,
                                             int z0, int z1) {
    double b2 = PCR_INF;
    const double o_yz = oy2 + oz2, o_xz = ox2 + oz2, o_xy = ox2 + oy2;
    if (x0 > 0) {
        double gap = sx - (g.lo[0] + (double)x0 * g.h) - margin;
        gap = gap > 0.0 ? gap : 0.0;
        b2 = fmin(b2, gap * gap + o_yz);
    }
    if (x1 < g.g[0] - 1) {
        double gap = (g.lo[0] + (double)(x1 + 1) * g.h) - sx - margin;
        gap = gap > 0.0 ? gap : 0.0;
        b2 = fmin(b2, gap * gap + o_yz);
    }
    if (y0 > 0) {
        double gap = sy - (g.lo[1] + (double)y0 * g.h) - margin;
        gap = gap > 0.0 ? gap : 0.0;
        b2 = fmin(b2, gap * gap + o_xz);
    }
    if (y1 < g.g[1] - 1) {
        double gap = (g.lo[1] + (double)(y1 + 1) * g.h) - sy - margin;
        gap = gap > 0.0 ? gap : 0.0;
        b2 = fmin(b2, gap * gap + o_xz);
    }
    if (z0 > 0) {
        double gap = sz - (g.lo[2] + (double)z0 * g.h) - margin;
        gap = gap > 0.0 ? gap : 0.0;
        b2 = fmin(b2, gap * gap + o_xy);
    }
    if (z1 < g.g[2] - 1) {
        double gap = (g.lo[2] + (double)(z1 + 1) * g.h) - sz - margin;
        gap = gap > 0.0 ? gap : 0.0;
        b2 = fmin(b2, gap * gap + o_xy);
    }
    return b2;
}

// Exact K-NN of one query on the grid.
//   cand     : candidate coordinates, parallel to g.pts (g.pts itself for a plain search, the
//              transformed copy for the reverse search of the ICP step)
//   (qx,qy,qz): the query in the frame of `cand` -- distances are evaluated here
//   (sx,sy,sz): the query in the frame of the GRID (equal to q for a plain search)
//   bound_scale: factor (<1) applied to the squared lower bound before it is compared with a
//              candidate distance: covers rounding of the bound and, for the reverse search, the
//              deviation of R from an isometry
//   extra_margin: added to g.margin (reverse search: rounding of the inverse map)
//   max_d2   : candidates with d2 >= max_d2 are of no interest; the search may stop as soon as every
//              unvisited point is at least that far
// top must be initialised by the caller (init(), or pre-seeded with a known candidate).
template <int KC>
__device__ __forceinline__ void grid_search(const GridView &g, const Pt4 *__restrict__ cand, double qx, double qy,
                                            double qz, double sx, double sy, double sz, double max_d2,
                                            double bound_scale, double extra_margin, TopK<KC> &top) {
    const uint32_t *__restrict__ cs = g.cs;
    const int gx = g.g[0], gy = g.g[1], gz = g.g[2];
    const int cx = cell_coord(sx, g.lo[0], g.inv_h, gx);
    const int cy = cell_coord(sy, g.lo[1], g.inv_h, gy);
    const int cz = cell_coord(sz, g.lo[2], g.inv_h, gz);
    const double margin = g.margin + extra_margin + 9.094947017729282e-13 * (fabs(sx) + fabs(sy) + fabs(sz));

    // distance of the grid-frame query to the bounds of the points, per axis (0 inside)
    double ox = fmax(fmax(g.lo[0] - sx, sx - g.hi[0]), 0.0) - margin;
    double oy = fmax(fmax(g.lo[1] - sy, sy - g.hi[1]), 0.0) - margin;
    double oz = fmax(fmax(g.lo[2] - sz, sz - g.hi[2]), 0.0) - margin;
    ox = ox > 0.0 ? ox : 0.0;
    oy = oy > 0.0 ? oy : 0.0;
    oz = oz > 0.0 ? oz : 0.0;
    const double ox2 = ox * ox, oy2 = oy * oy, oz2 = oz * oz;

    double wd = top.worst_d();
    uint32_t wi = top.worst_i();

    // previously scanned (clamped) box; empty before the first ring
    int px0 = 1, px1 = 0, py0 = 1, py1 = 0, pz0 = 1, pz1 = 0;
    const int rmax = max(max(gx, gy), gz);
    for (int r = 1; r <= rmax; ++r) {
        const int x0 = max(cx - r, 0), x1 = min(cx + r, gx - 1);
        const int y0 = max(cy - r, 0), y1 = min(cy + r, gy - 1);
        const int z0 = max(cz - r, 0), z1 = min(cz + r, gz - 1);
        const int ny = y1 - y0 + 1;
        const int nrows = ny * (z1 - z0 + 1);
        // ---- flattened scan of the shell: one candidate per trip, rows advanced on demand ----
        int row = 0, ry = y0, rz = z0, seg = 0;
        uint32_t p = 0, e = 0;
        for (;;) {
            while (p >= e) {
                if (row >= nrows) break;
                const bool inner = (ry >= py0) & (ry <= py1) & (rz >= pz0) & (rz <= pz1);
                int xs, xe;
                bool advance;
                if (!inner) {
                    xs = x0;
                    xe = x1;
                    advance = true;
                } else if (seg == 0) {
                    xs = x0;
                    xe = px0 - 1;
                    seg = 1;
                    advance = false;
                } else {
                    xs = px1 + 1;
                    xe = x1;
                    seg = 0;
                    advance = true;
                }
                if (xs <= xe) {
                    const size_t base = ((size_t)rz * gy + ry) * gx;
                    p = cs[base + xs];
                    e = cs[base + xe + 1];
                }
                if (advance) {
                    ++row;
                    if (++ry > y1) {
                        ry = y0;
                        ++rz;
                    }
                }
            }
            if (p >= e) break;
            const Pt4 c = cand[p];
            const double d = sqdist3(qx, qy, qz, c.x, c.y, c.z);
            const uint32_t ci = (uint32_t)c.w;
            if (better(d, ci, wd, wi)) {
                top.insert(d, ci, p);
                wd = top.worst_d();
                wi = top.worst_i();
            }
            ++p;
        }
        // ---- can anything outside the scanned box still matter? ----
        const bool all = (x0 == 0) & (x1 == gx - 1) & (y0 == 0) & (y1 == gy - 1) & (z0 == 0) & (z1 == gz - 1);
        if (all) break;
        double b2 = box_bound2(g, sx, sy, sz, margin, ox2, oy2, oz2, x0, x1, y0, y1, z0, z1);
        b2 *= bound_scale;
        // strict: an unvisited point at exactly the K-th distance could still win on index
        if (wd < b2 || max_d2 <= b2) break;
        px0 = x0;
        px1 = x1;
        py0 = y0;
        py1 = y1;
        pz0 = z0;
        pz1 = z1;
    }
}

// Brute-force scan of one LDS tile (all lanes read the same address: LDS broadcast).
template <int KC>
__device__ __forceinline__ void tile_scan(const Pt4 *tile, int count, uint32_t pos_base, double qx, double qy,
                                          double qz, TopK<KC> &top) {
    double wd = top.worst_d();
    uint32_t wi = top.worst_i();
    for (int j = 0; j < count; ++j) {
        const Pt4 c = tile[j];
        const double d = sqdist3(qx, qy, qz, c.x, c.y, c.z);
        const uint32_t ci = (uint32_t)c.w;
        if (better(d, ci, wd, wi)) {
            top.insert(d, ci, pos_base + j);
            wd = top.worst_d();
            wi = top.worst_i();
        }
    }
}

// XCD-aware block remap: consecutive LOGICAL blocks (spatially adjacent queries) land on the same
// XCD, so each XCD's private L2 holds one contiguous slab of the index.  Bijective for any nb.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t bid, uint32_t nb) {
    const uint32_t q = nb >> 3, r = nb & 7u;
    const uint32_t xcd = bid & 7u, slot = bid >> 3;
    const uint32_t start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + slot;
}

}  // namespace pcr
