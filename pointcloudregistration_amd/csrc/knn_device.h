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
#ifndef PCR_WALK_UNROLL
#define PCR_WALK_UNROLL 3
#endif

#ifdef PCR_DIAG
// diagnostic build only (libpcr_diag.so): counters of the lane search.  Never compiled into libpcr.so.
static __device__ unsigned long long g_diag[48];  // (32..47: the filtered step kernel's own, pcr_icp.hip)
#define DIAG_T() ((unsigned long long)__builtin_amdgcn_s_memtime())
// PCR_DIAG_LIGHT: only the per-wave TIME slots (42..47) are kept -- a handful of atomics per wave, where the event counters'
// per-lane atomics slow a kernel tenfold and distort every share
#ifdef PCR_DIAG_LIGHT
#define DIAG_KEEP(slot) ((slot) >= 42)
#else
#define DIAG_KEEP(slot) true
#endif
#define DIAG_ADD(slot, v)                                                                \
    do {                                                                                 \
        if (DIAG_KEEP(slot) && (int)(threadIdx.x & 63) == __ffsll((unsigned long long)__ballot(1)) - 1) atomicAdd(&g_diag[slot], (unsigned long long)(v)); \
    } while (0)
#else
#define DIAG_T() 0ull
#define DIAG_ADD(slot, v) \
    do {                  \
    } while (0)
#endif
#if defined(PCR_DIAG) && !defined(PCR_DIAG_LIGHT)
#define DIAG_LANE_ADD(slot, v) atomicAdd(&g_diag[slot], (unsigned long long)(v))
#else
#define DIAG_LANE_ADD(slot, v) \
    do {                       \
    } while (0)
#endif

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
// (RG: Rigid in any address space -- the filtered kernel reads it through the constant address space)
template <class RG>
__device__ __forceinline__ void xform(const RG &M, double x, double y, double z, double &ox, double &oy,
                                      double &oz) {
    ox = dadd(dadd(dadd(dmul(M.r[0], x), dmul(M.r[3], y)), dmul(M.r[6], z)), M.t[0]);
    oy = dadd(dadd(dadd(dmul(M.r[1], x), dmul(M.r[4], y)), dmul(M.r[7], z)), M.t[1]);
    oz = dadd(dadd(dadd(dmul(M.r[2], x), dmul(M.r[5], y)), dmul(M.r[8], z)), M.t[2]);
}

// Squared distance from the query to a candidate record.  XF: the record holds the point in the index's own frame
// and is moved by the current (R,T) first -- CPointCloud::transform's expression, so the value is the one the
// reference gets from its transformed copy, without that copy ever being written or read.
template <bool XF, class RG = Rigid>
__device__ __forceinline__ double cand_d2(const RG *xf, double qx, double qy, double qz, const Pt4 &c) {
    if constexpr (XF) {
        double x, y, z;
        xform(*xf, c.x, c.y, c.z, x, y, z);
        return sqdist3(qx, qy, qz, x, y, z);
    } else {
        return sqdist3(qx, qy, qz, c.x, c.y, c.z);
    }
}

// (d, i) before (bd, bi) in the total order (d2, index).  Bitwise operators on purpose: no short-circuit
// branches in the hot loops.
__device__ __forceinline__ bool better(double d, uint32_t i, double bd, uint32_t bi) {
    return (d < bd) | ((d == bd) & (i < bi));
}

// Sorted top-K in registers.  Every access is statically indexed (all loops over KC unroll), so the
// arrays stay in VGPRs; a search always fills all KC slots -- the K <= KC nearest a caller asked for
// are the first K of them (same total order), so no run-time K reaches the hot loop.  Only (d2, original
// index) is kept; a caller that needs coordinates looks the position up through the index's pos_of table.
template <int KC>
struct TopK {
    double d[KC];
    uint32_t idx[KC];
    __device__ __forceinline__ void init(double d0 = PCR_INF) {
#pragma unroll
        for (int j = 0; j < KC; ++j) {
            d[j] = d0;
            idx[j] = PCR_NO_INDEX;
        }
    }
    __device__ __forceinline__ double worst_d() const { return d[KC - 1]; }
    __device__ __forceinline__ uint32_t worst_i() const { return idx[KC - 1]; }
    // Slot k-1 for a run-time k, WITHOUT a dynamic register index (which would push the arrays to
    // scratch memory): OR together the slots under an all-ones/all-zeros mask.
    __device__ __forceinline__ void slot(int k, double &od, uint32_t &oi) const {
        unsigned long long db = 0;
        uint32_t ib = 0;
#pragma unroll
        for (int j = 0; j < KC; ++j) {
            const uint32_t m = (uint32_t)0 - (uint32_t)(j == k - 1);
            const unsigned long long m64 = ((unsigned long long)m << 32) | m;
            db |= (unsigned long long)__double_as_longlong(d[j]) & m64;
            ib |= idx[j] & m;
        }
        od = __longlong_as_double((long long)db);
        oi = ib;
    }
    // Insert (dist, i), keeping the slots sorted ascending by (d, idx); the caller has checked that the
    // candidate beats slot KC-1.  The candidate replaces the worst slot and bubbles up by compare-exchange:
    // one comparison and one conditional swap per slot, no second select level.
    __device__ __forceinline__ void insert(double dist, uint32_t i) {
        d[KC - 1] = dist;
        idx[KC - 1] = i;
#pragma unroll
        for (int j = KC - 1; j >= 1; --j) {
            const bool up = better(d[j], idx[j], d[j - 1], idx[j - 1]);
            // distances are finite or +inf, never NaN: min/max order them exactly (equal values: either way)
            const double dl = fmin(d[j], d[j - 1]), dh = fmax(d[j], d[j - 1]);
            const uint32_t il = up ? idx[j] : idx[j - 1], ih = up ? idx[j - 1] : idx[j];
            d[j - 1] = dl;
            d[j] = dh;
            idx[j - 1] = il;
            idx[j] = ih;
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

// The positions [p, e) of the points of cells a .. b-1 (consecutive along x; b may be one past the last cell).
// CT: the level may carry a compact cell table (GridView::cbits) -- only the searches of nested indices are compiled
// with it; an empty range then costs no second load.
__device__ __forceinline__ uint32_t cell_rank(const uint2 *__restrict__ cbits, size_t c) {
    const uint2 w = cbits[c >> 5];
    return w.y + (uint32_t)__popc(w.x & ((1u << (c & 31u)) - 1u));
}
template <bool CT, class GV>
__device__ __forceinline__ void cell_range(const GV &g, size_t a, size_t b, uint32_t &p, uint32_t &e) {
    if constexpr (CT) {
        if (g.cbits) {
            const uint32_t ra = cell_rank(g.cbits, a), rb = cell_rank(g.cbits, b);
            p = e = 0;
            if (ra != rb) {
                p = g.cstart[ra];
                e = g.cstart[rb];
            }
            return;
        }
    }
    p = g.cs[a];
    e = g.cs[b];
}

// Distance from coordinate s to the slab of cells with index c along one axis (0 inside), shrunk by the margin.
__device__ __forceinline__ double slab_gap(double s, double lo, int c, double h, double margin) {
    const double f0 = lo + (double)c * h, f1 = lo + (double)(c + 1) * h;
    const double gap = fmax(fmax(f0 - s, s - f1), 0.0) - margin;
    return gap > 0.0 ? gap : 0.0;
}

// sqrt(v) for v >= 0 (or +inf), never too small: a float root, inflated past its rounding.  Only ever used to
// decide which cells to open, with a cell of slack on top.
__device__ __forceinline__ double ball_half_width(double v) {
    return (double)sqrtf(__double2float_ru(v)) * 1.000001;
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

// A long candidate range (a row of cells through a dense part of the cloud).  Rows are sorted by grid-frame x
// (pcr_sort.hip), so instead of scanning all of it the lane enters at the query's x by binary search and walks
// outwards in both directions, stopping a direction as soon as the x-gap alone -- shrunk by the safety margin and
// the bound scale -- exceeds the current K-th distance: everything beyond is farther still.  Exact for the same
// reason every other pruning here is: the test is strict, and `wd` only ever shrinks.
//   order: the grid-frame points the range is sorted by (g.pts); cand: what distances are evaluated against
//   (g.pts again, or the transformed copy for the reverse search of the ICP step).
#define PCR_LONG_RANGE 48u
template <int KC, bool XF = false>
__device__ __forceinline__ void windowed_scan(const Pt4 *__restrict__ order, const Pt4 *__restrict__ cand, uint32_t b,
                                              uint32_t e, double qx, double qy, double qz, double sx, double margin,
                                              double bound_scale, TopK<KC> &top, double &wd, uint32_t &wi,
                                              const Rigid *xf = nullptr) {
    uint32_t lo = b, hi = e;
    while (lo < hi) {  // first position with x >= sx
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (order[mid].x < sx)
            lo = mid + 1;
        else
            hi = mid;
    }
    for (uint32_t i = lo; i < e; ++i) {
        const double gap = order[i].x - sx - margin;
        if ((gap > 0.0) & (gap * gap * bound_scale > wd)) break;
        const Pt4 c = cand[i];
        const double d = cand_d2<XF>(xf, qx, qy, qz, c);
        const uint32_t ci = (uint32_t)c.w;
        if (better(d, ci, wd, wi)) {
            top.insert(d, ci);
            wd = top.worst_d();
            wi = top.worst_i();
        }
    }
    for (uint32_t i = lo; i > b;) {
        --i;
        const double gap = sx - order[i].x - margin;
        if ((gap > 0.0) & (gap * gap * bound_scale > wd)) break;
        const Pt4 c = cand[i];
        const double d = cand_d2<XF>(xf, qx, qy, qz, c);
        const uint32_t ci = (uint32_t)c.w;
        if (better(d, ci, wd, wi)) {
            top.insert(d, ci);
            wd = top.worst_d();
            wi = top.worst_i();
        }
    }
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
// r_first (0: r_start): the radius of the first shell, when the caller knows that a wider one is needed anyway.
// LANE_RMAX > 0: give up before a shell of a larger radius than that and return the radius of the cube scanned so far
// (the caller continues from there -- coop_shells below).  Returns 0 when the search is finished.
template <int KC, bool WINDOW = true, bool XF = false, int LANE_RMAX = 0, bool CT = false>
__device__ __forceinline__ int grid_search(const GridView &g, const Pt4 *__restrict__ cand, double qx, double qy,
                                            double qz, double sx, double sy, double sz, double max_d2,
                                            double bound_scale, double extra_margin, TopK<KC> &top,
                                            int r_start = 1, const Rigid *xf = nullptr, int r_first = 0) {
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

    // previously scanned (clamped) box; empty before the first ring.  r_start > 1: the caller has already
    // scanned the cube of radius r_start-1 around the query's cell.
    int px0 = 1, px1 = 0, py0 = 1, py1 = 0, pz0 = 1, pz1 = 0;
    if (r_start > 1) {
        px0 = max(cx - (r_start - 1), 0);
        px1 = min(cx + (r_start - 1), gx - 1);
        py0 = max(cy - (r_start - 1), 0);
        py1 = min(cy + (r_start - 1), gy - 1);
        pz0 = max(cz - (r_start - 1), 0);
        pz1 = min(cz + (r_start - 1), gz - 1);
    }
    const int rmax = max(max(gx, gy), gz);
    const double inv_bs = 1.0 / bound_scale;
    DIAG_LANE_ADD(8 * (KC == 1) + 6, 1);  // shell searches
#ifdef PCR_DIAG
    unsigned long long diag_rows = 0, dg_c = 0, dg_l = 0, dg_r = 0, dg_s = 0, dg_sh = 0;
    const unsigned long long diag_t0 = DIAG_T();
#define DG(x) ++(x)
#else
#define DG(x) \
    do {      \
    } while (0)
#endif
    // (r_first > r_start: the first shell is wider than the next radius -- the caller applied the rule for the next
    // radius, below, to what it knows already)
    for (int r = max(r_start, min(r_first, rmax)); r <= rmax;) {
        DIAG_LANE_ADD(8 * (KC == 1) + 7, 1);  // shells
        DG(dg_sh);
        const int x0 = max(cx - r, 0), x1 = min(cx + r, gx - 1);
        const int y0 = max(cy - r, 0), y1 = min(cy + r, gy - 1);
        const int z0 = max(cz - r, 0), z1 = min(cz + r, gz - 1);
        // ---- scan of the shell, flattened into ONE loop: every trip a lane does one unit of work -- evaluates a
        // candidate, or moves to the next layer, or handles the next row (segment) -- so a wave runs as long as its
        // busiest lane, not as long as all its lanes' empty rows added up (an inner loop "until a row has
        // candidates" would make every lane wait for every other lane's search for one).  Of the box only the part
        // inside the BALL of everything that can still matter is looked at: a layer's rows are limited to the disc
        // the ball cuts out of it, a row's cells to the chord (one cell of slack either side; the radius is a float
        // square root rounded up), and a row is pruned with the K-th distance of the moment.  A query hovering
        // above a dense surface thus opens the few rows below it, not the whole box. ----
        // Order: first the "cross" through the query's own cell -- its row in every layer, then every row of its own
        // layer -- and only then the rest, layer by layer.  The cross is cheap (two lines of rows) and meets every
        // nearby surface close to the point where that surface is nearest, so the K-th distance is tight before
        // the bulk of the box is looked at, most of which is then pruned unopened.
        // (Only worth its extra trips in a large shell; a small one is scanned plainly: phase 3.)
        int rz = z0 - 1, ry = 0, ylast = -1, seg = 0, phase = r >= 8 ? 0 : 3;  // ry > ylast: the layer is used up
        bool layer_start = false, alive = true;
        double dz2 = 0.0;
        uint32_t p = 0, e = 0;
        while (alive) {
            if (p < e) {
                // PCR_WALK_UNROLL candidates per trip, as in the ring walk: their loads are in flight together and
                // the trip's bookkeeping is paid once.  Slots past the end re-read the last candidate and are ignored.
                DG(dg_c);
                Pt4 c[PCR_WALK_UNROLL];
                const uint32_t last = e - 1, left = e - p;
#pragma unroll
                for (int u = 0; u < PCR_WALK_UNROLL; ++u) c[u] = cand[min(p + (uint32_t)u, last)];
                p += min(left, (uint32_t)PCR_WALK_UNROLL);
                DIAG_LANE_ADD(8 * (KC == 1) + 4, min(left, (uint32_t)PCR_WALK_UNROLL));  // candidates of the shell search
#pragma unroll
                for (int u = 0; u < PCR_WALK_UNROLL; ++u) {
                    const double d = cand_d2<XF>(xf, qx, qy, qz, c[u]);
                    const uint32_t ci = (uint32_t)c[u].w;
                    if (((uint32_t)u < left) & better(d, ci, wd, wi)) {
                        top.insert(d, ci);
                        wd = top.worst_d();
                        wi = top.worst_i();
                    }
                }
            } else if (ry > ylast) {  // the next layer, or the next phase
                DG(dg_l);
                ++rz;
                if (phase == 1) {  // the query's own layer is done: now all the others
                    phase = 2;
                    rz = z0;
                }
                if (rz > z1) {
                    if (phase == 0) {  // the query's row has been seen in every layer: now its own layer
                        phase = 1;
                        rz = cz;
                    } else {
                        alive = false;
                    }
                }
                if (alive & !((phase == 2) & (rz == cz))) {
                    const double dz = slab_gap(sz, g.lo[2], rz, g.h, margin);
                    dz2 = dz * dz;
                    const double rem_z = fmin(wd, max_d2) * inv_bs - dz2;
                    if (rem_z >= 0.0) {  // else the whole layer is out of reach: ry > ylast stays, the next trip moves on
                        const double w = ball_half_width(rem_z) + margin;
                        double lo_c = (sy - w - g.lo[1]) * g.inv_h - 1.0, hi_c = (sy + w - g.lo[1]) * g.inv_h + 1.0;
                        lo_c = fmin(fmax(lo_c, (double)y0), (double)y1 + 1.0);
                        hi_c = fmax(fmin(hi_c, (double)y1), (double)y0 - 1.0);
                        ry = (int)lo_c;
                        ylast = (int)hi_c;
                        if (phase == 0) {  // only the query's own row
                            ry = max(ry, cy);
                            ylast = min(ylast, cy);
                        }
                        layer_start = true;
                    }
                }
            } else if (((phase == 1) | (phase == 2)) & (ry == cy)) {  // the query's own row was handled in phase 0
                ++ry;
            } else {  // row ry of layer rz (its next segment)
                bool skip = false;
                if constexpr (WINDOW) {
                    // entering a strip of 8x8 rows (or the layer, in the middle of one): if the strip holds no point
                    // at all in the shell's x-range, its rows of this layer are skipped with two loads
                    if ((seg == 0) & (((ry & 7) == 0) | layer_start) && g.cs8) {
                        const size_t b8 = ((size_t)(rz >> 3) * g.gy8 + (ry >> 3)) * gx;
                        skip = g.cs8[b8 + x1 + 1] == g.cs8[b8 + x0];
                    }
                }
                layer_start = false;
                DG(dg_r);
                if (skip) {
                    DG(dg_s);
                    ry = (ry | 7) + 1;
                } else {
                    const bool inner = (ry >= py0) & (ry <= py1) & (rz >= pz0) & (rz <= pz1);
                    int xs, xe;
                    if (!inner) {
                        xs = x0;
                        xe = x1;
                    } else if (seg == 0) {
                        xs = x0;
                        xe = px0 - 1;
                        seg = 1;
                    } else {
                        xs = px1 + 1;
                        xe = x1;
                        seg = 0;
                    }
                    if (xs <= xe) {
                        const double dy = slab_gap(sy, g.lo[1], ry, g.h, margin);
                        const double rem = fmin(wd, max_d2) * inv_bs - dz2 - dy * dy;  // with the K-th distance of NOW
                        if (rem >= 0.0) {
                            const double w = ball_half_width(rem) + margin;
                            double lo_c = (sx - w - g.lo[0]) * g.inv_h - 1.0, hi_c = (sx + w - g.lo[0]) * g.inv_h + 1.0;
                            lo_c = fmin(fmax(lo_c, (double)xs), (double)xe + 1.0);
                            hi_c = fmax(fmin(hi_c, (double)xe), (double)xs - 1.0);
                            xs = (int)lo_c;
                            xe = (int)hi_c;
                            if (xs <= xe) {
                                DIAG_LANE_ADD(8 * (KC == 1) + 3, 1);  // rows opened by the shell search
                                DG(diag_rows);
                                const size_t base = ((size_t)rz * gy + ry) * gx;
                                cell_range<CT>(g, base + xs, base + xe + 1, p, e);
                                if constexpr (WINDOW) {
                                    if (e - p > PCR_LONG_RANGE && e > p) {  // dense stretch: enter at the query's x instead
                                        windowed_scan<KC, XF>(g.pts, cand, p, e, qx, qy, qz, sx, margin, bound_scale, top, wd,
                                                              wi, xf);
                                        p = e;
                                    }
                                }
                            }
                        }
                    }
                    if (seg == 0) ++ry;  // the row's last segment has been handled
                }
            }
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
        // Next radius: double, but never past what the ball of everything that can still matter needs.  (Doubling
        // even when the K-th distance is known: it may come from another level of the index, or from a far
        // candidate, and say little about where THIS grid's points are; the shells scanned on the way cost a
        // third of the last one.)
        const double reach2 = fmin(wd, max_d2);  // nothing at or beyond this squared distance can matter
        int rn = 2 * r;
        if (reach2 < PCR_INF) {
            const double cells_needed = (sqrt(reach2 * inv_bs) + margin) * g.inv_h + 2.0;
            const int need = cells_needed < (double)rmax ? (int)cells_needed : rmax;
            rn = max(r + 1, min(rn, need));
        }
        if constexpr (LANE_RMAX > 0) {
            if (min(rn, rmax) > LANE_RMAX) return r;
        }
        r = min(rn, rmax);  // at r == rmax the box is the whole grid and the loop ends through `all`
    }
#if defined(PCR_DIAG) && !defined(PCR_DIAG_LIGHT)
    {
        const unsigned long long dt = dg_c + dg_l + dg_r;
        if (atomicMax(&g_diag[24], dt) < dt) {  // the shell search with the most trips so far: what did it do?
            g_diag[25] = dg_c;
            g_diag[26] = dg_l;
            g_diag[27] = dg_r;
            g_diag[28] = dg_s;
            g_diag[29] = dg_sh;
            g_diag[30] = diag_rows;
            g_diag[31] = (unsigned long long)(g.h * 1e6);
        }
    }
#endif
#undef DG
    return 0;
}


// Ring 1 (the 3x3x3 cells around the query's own cell) straight from global memory, one lane per query.
// The nine (y,z) rows are visited centre first, and a row -- or the outer cell of a row -- is skipped once its
// slab is provably farther than the current K-th best.  What a row needs when it is opened (its slab distance
// and the cell-table offset of its centre cell) is computed ONCE in straight-line code and parked in the lane's
// column of a small LDS table; opening a row in the loop is then one LDS read, two compares and two cell-table
// loads instead of ~90 instructions of decode and 64-bit address arithmetic.
// One loop with a straight-line body (open next row | evaluate up to PCR_WALK_UNROLL candidates).
// rowtab: this lane's column, row r at rowtab[r * PCR_ROW_STRIDE] (consecutive lanes, 8 bytes: conflict-free).
// Returns true if the K-th best is provably final (or the grid is exhausted).
// RS: lanes per row of the table the caller handed in -- 256 in the 256-lane kernels, 64 in the filtered step kernel,
// whose blocks are single waves.
#define PCR_ROW_STRIDE 256
template <int KC, bool XF = false, int RS = PCR_ROW_STRIDE, bool CT = false>
__device__ __forceinline__ bool ring1_pruned(const GridView &g, const Pt4 *__restrict__ cand, double qx, double qy,
                                             double qz, double sx, double sy, double sz, double max_d2,
                                             double bound_scale, double extra_margin, TopK<KC> &top, uint2 *rowtab,
                                             const Rigid *xf = nullptr) {
    const int gx = g.g[0], gy = g.g[1], gz = g.g[2];
    const int cx = cell_coord(sx, g.lo[0], g.inv_h, gx);
    const int cy = cell_coord(sy, g.lo[1], g.inv_h, gy);
    const int cz = cell_coord(sz, g.lo[2], g.inv_h, gz);
    const double margin = g.margin + extra_margin + 9.094947017729282e-13 * (fabs(sx) + fabs(sy) + fabs(sz));
    double wd = top.worst_d();
    uint32_t wi = top.worst_i();
    // squared distance from the query to the slab of cells at offset -1 / +1 along each axis, shrunk by the
    // safety margin and the bound scale: a slab may be skipped once wd < that value
    double gl, gh;
    gl = sx - (g.lo[0] + (double)cx * g.h) - margin;
    gh = (g.lo[0] + (double)(cx + 1) * g.h) - sx - margin;
    gl = gl > 0.0 ? gl : 0.0;
    gh = gh > 0.0 ? gh : 0.0;
    const double fxl = gl, fxh = gh;
    const double gxl = gl * gl * bound_scale, gxh = gh * gh * bound_scale;
    gl = sy - (g.lo[1] + (double)cy * g.h) - margin;
    gh = (g.lo[1] + (double)(cy + 1) * g.h) - sy - margin;
    gl = gl > 0.0 ? gl : 0.0;
    gh = gh > 0.0 ? gh : 0.0;
    const double fyl = gl, fyh = gh;
    const double gyl = gl * gl * bound_scale, gyh = gh * gh * bound_scale;
    gl = sz - (g.lo[2] + (double)cz * g.h) - margin;
    gh = (g.lo[2] + (double)(cz + 1) * g.h) - sz - margin;
    gl = gl > 0.0 ? gl : 0.0;
    gh = gh > 0.0 ? gh : 0.0;
    const double fzl = gl, fzh = gh;
    const double gzl = gl * gl * bound_scale, gzh = gh * gh * bound_scale;
    // When will the K-th best be final?  An unvisited point is either in a pruned row/cell (farther than the bound
    // that pruned it, which was >= the final wd) or outside the 3x3x3 box, i.e. beyond a box face: at least one whole
    // cell plus the (margin-shrunk, clamped) gap to the own cell's face away along that axis.  Faces on the grid
    // boundary have nothing beyond them.  Geometry only -- computed here, before the walk, so that one value
    // instead of six gaps stays live across the loop.
    double settle_b2 = PCR_INF;
    if (cx > 1) settle_b2 = fmin(settle_b2, (fxl + g.h) * (fxl + g.h));
    if (cx < gx - 2) settle_b2 = fmin(settle_b2, (fxh + g.h) * (fxh + g.h));
    if (cy > 1) settle_b2 = fmin(settle_b2, (fyl + g.h) * (fyl + g.h));
    if (cy < gy - 2) settle_b2 = fmin(settle_b2, (fyh + g.h) * (fyh + g.h));
    if (cz > 1) settle_b2 = fmin(settle_b2, (fzl + g.h) * (fzl + g.h));
    if (cz < gz - 2) settle_b2 = fmin(settle_b2, (fzh + g.h) * (fzh + g.h));
    settle_b2 *= bound_scale * (1.0 - 4.440892098500626e-16);
    // row table, rows in order of increasing slab distance class: centre, 4 faces, 4 edges.  The slab distance is
    // kept as a float rounded DOWN (a smaller bound only prunes less); rows outside the grid get +inf = never open.
    // The cell table has <= 2^30 entries (pcr_index_params.max_cells), so its offsets fit 32 bits.
    {
        constexpr int DY[9] = {0, -1, 1, 0, 0, -1, 1, -1, 1};
        constexpr int DZ[9] = {0, 0, 0, -1, 1, -1, -1, 1, 1};
        const uint32_t cbase = ((uint32_t)cz * (uint32_t)gy + (uint32_t)cy) * (uint32_t)gx + (uint32_t)cx;
        const int zstep = gx * gy;
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            const int ry = cy + DY[r], rz = cz + DZ[r];
            const double gy2 = DY[r] < 0 ? gyl : (DY[r] > 0 ? gyh : 0.0);
            const double gz2 = DZ[r] < 0 ? gzl : (DZ[r] > 0 ? gzh : 0.0);
            const bool inside = (ry >= 0) & (ry < gy) & (rz >= 0) & (rz < gz);
            const float f = inside ? __double2float_rd(gy2 + gz2) : __builtin_huge_valf();
            rowtab[r * RS] = make_uint2(__float_as_uint(f), cbase + (uint32_t)(DY[r] * gx + DZ[r] * zstep));
        }
    }
    const bool has_xl = cx > 0, has_xh = cx < gx - 1;
    int ri = 0;
    uint32_t p = 0, e = 0;
    bool alive = true;
    while (alive) {
        DIAG_ADD(8 * (KC == 1) + 0, 1);                                        // wave trips
        DIAG_ADD(8 * (KC == 1) + 1, __any((p >= e) & (ri < 9)) ? 1 : 0);       // wave-level open executions
        if ((p >= e) & (ri < 9)) {
            const uint2 row = rowtab[ri * RS];
            ++ri;
            const double gyz = (double)__uint_as_float(row.x);
            // rows outside the grid carry +inf; the explicit test matters while wd itself is still +inf
            if ((row.x != 0x7f800000u) & !(wd < gyz)) {
                const uint32_t xs = (has_xl & !(wd < gyz + gxl)) ? row.y - 1u : row.y;
                const uint32_t xe = (has_xh & !(wd < gyz + gxh)) ? row.y + 2u : row.y + 1u;
                cell_range<CT>(g, xs, xe, p, e);
                if (e - p > PCR_LONG_RANGE && e > p) {  // dense stretch: enter at the query's x instead
                    windowed_scan<KC, XF>(g.pts, cand, p, e, qx, qy, qz, sx, margin, bound_scale, top, wd, wi, xf);
                    p = e;
                }
            }
        }
        if (p < e) {
            // PCR_WALK_UNROLL candidates per trip: their loads are in flight together, and the loop bookkeeping
            // is paid once.  Slots past the end of the range re-read its last candidate and are ignored.
            Pt4 c[PCR_WALK_UNROLL];
            const uint32_t last = e - 1;
#pragma unroll
            for (int u = 0; u < PCR_WALK_UNROLL; ++u) c[u] = cand[min(p + (uint32_t)u, last)];
            const uint32_t left = e - p;
            p += min(left, (uint32_t)PCR_WALK_UNROLL);
            DIAG_LANE_ADD(8 * (KC == 1) + 2, min(left, (uint32_t)PCR_WALK_UNROLL));
#pragma unroll
            for (int u = 0; u < PCR_WALK_UNROLL; ++u) {
                const double d = cand_d2<XF>(xf, qx, qy, qz, c[u]);
                const uint32_t ci = (uint32_t)c[u].w;
                if (((uint32_t)u < left) & better(d, ci, wd, wi)) {
                    top.insert(d, ci);
                    wd = top.worst_d();
                    wi = top.worst_i();
                }
            }
        }
        alive = (p < e) | (ri < 9);
    }
    return (wd < settle_b2) | (max_d2 <= settle_b2);
}

// The shell of radius 2 around the query's cell -- the 5x5x5 cells less the 3x3x3 that ring1_pruned or the filtered walk
// has been through -- for the queries of a nested index that the first walk leaves open (a quarter of a 16 M LiDAR
// sweep's samples: 1 cm of noise against 6 mm cells).  grid_search does it one row per trip of its flat loop, each trip
// behind a dependent pair of cell-table loads: 25 rows, 9 of them opened, 16 candidates -- some 27 dependent trips for a
// lane, and the wave runs as long as its slowest.  Here a LAYER's (up to eight) x-segments are looked up in straight-line
// code, all loads in flight together, parked in the lane's column of the row table, and walked as one stream: a layer
// costs one round trip to the cell table plus its candidates.  Same cells, same pruning rule (a layer or a row is skipped
// when its slab is provably farther than the K-th best of the moment), same total order: the same exact result.  Returns true when the K-th best is final; false: continue with grid_search from radius 3.
template <int KC, bool XF, int RS, bool CT>
__device__ __forceinline__ bool shell2_batched(const GridView &g, const Pt4 *__restrict__ cand, double qx, double qy,
                                               double qz, double sx, double sy, double sz, double max_d2,
                                               double bound_scale, double extra_margin, TopK<KC> &top, uint2 *rowtab,
                                               const Rigid *xf = nullptr) {
    const int gx = g.g[0], gy = g.g[1], gz = g.g[2];
    const int cx = cell_coord(sx, g.lo[0], g.inv_h, gx);
    const int cy = cell_coord(sy, g.lo[1], g.inv_h, gy);
    const int cz = cell_coord(sz, g.lo[2], g.inv_h, gz);
    const double margin = g.margin + extra_margin + 9.094947017729282e-13 * (fabs(sx) + fabs(sy) + fabs(sz));
    const double inv_bs = 1.0 / bound_scale;
    double wd = top.worst_d();
    uint32_t wi = top.worst_i();
    const int x0 = max(cx - 2, 0), x1 = min(cx + 2, gx - 1), y0 = max(cy - 2, 0), y1 = min(cy + 2, gy - 1);
    const int z0 = max(cz - 2, 0), z1 = min(cz + 2, gz - 1);
    const int px0 = max(cx - 1, 0), px1 = min(cx + 1, gx - 1);  // the cube walked already
    DIAG_LANE_ADD(35, 1);  // layer-at-a-time scans of the 5x5x5 cells
    double dy2_[5];  // the rows' slab distances, squared: the same in every layer
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const double dy = slab_gap(sy, g.lo[1], cy + (j == 0 ? 0 : (j & 1 ? -1 : 1) * ((j + 1) >> 1)), g.h, margin);
        dy2_[j] = dy * dy;
    }
#pragma unroll 1
    for (int li = 0; li < 5; ++li) {
        const int rz = cz + (li == 0 ? 0 : (li & 1 ? -1 : 1) * ((li + 1) >> 1));  // 0, -1, +1, -2, +2: nearest layers first
        uint32_t cnt = 0;
        if ((rz >= 0) & (rz < gz)) {
            const double dz = slab_gap(sz, g.lo[2], rz, g.h, margin);
            const double rem_z = fmin(wd, max_d2) * inv_bs - dz * dz;
            if (rem_z >= 0.0) {
                const bool z_in = (rz >= cz - 1) & (rz <= cz + 1);
                uint32_t p_[8], e_[8];
                // slots 0..4: the first (or only) segment of rows cy, cy-1, cy+1, cy-2, cy+2; 5..7: the second segment of
                // the three rows that cross the cube walked already
#pragma unroll
                for (int sl = 0; sl < 8; ++sl) {
                    const int j = sl < 5 ? sl : sl - 5;
                    const int ry = cy + (j == 0 ? 0 : (j & 1 ? -1 : 1) * ((j + 1) >> 1));
                    const bool inner = z_in & (j < 3);
                    bool ok = (ry >= 0) & (ry < gy) & (sl < 5 || inner);
                    int xs = sl < 5 ? x0 : px1 + 1, xe = sl < 5 ? (inner ? px0 - 1 : x1) : x1;
                    const double rem = rem_z - dy2_[j];
                    ok &= (rem >= 0.0) & (xs <= xe);
                    // (no chord here: trimming a row of at most five cells to what the ball cuts out of it costs more fp64
                    // arithmetic than the few candidates it spares -- 16 M LiDAR iteration 11.74 with, 11.61 ms without)
                    ok &= xs <= xe;
                    const size_t base = ((size_t)(ok ? rz : 0) * gy + (ok ? ry : 0)) * gx;
                    cell_range<CT>(g, base + (ok ? xs : 0), base + (ok ? xe + 1 : 0), p_[sl], e_[sl]);
                    if (!ok) e_[sl] = p_[sl];
                }
#pragma unroll
                for (int sl = 0; sl < 8; ++sl)
                    if (e_[sl] > p_[sl]) {
                        rowtab[cnt * RS] = make_uint2(p_[sl], e_[sl]);
                        ++cnt;
                    }
            }
        }
        uint32_t ri = 0, p = 0, e = 0;
        bool alive = cnt != 0u;
        while (alive) {
            if ((p >= e) & (ri < cnt)) {
                const uint2 row = rowtab[ri * RS];
                ++ri;
                p = row.x;
                e = row.y;
                if (e - p > PCR_LONG_RANGE) {  // dense stretch: enter at the query's x instead
                    windowed_scan<KC, XF>(g.pts, cand, p, e, qx, qy, qz, sx, margin, bound_scale, top, wd, wi, xf);
                    p = e;
                }
            }
            if (p < e) {
                Pt4 c[PCR_WALK_UNROLL];
                const uint32_t last = e - 1;
#pragma unroll
                for (int u = 0; u < PCR_WALK_UNROLL; ++u) c[u] = cand[min(p + (uint32_t)u, last)];
                const uint32_t left = e - p;
                p += min(left, (uint32_t)PCR_WALK_UNROLL);
#pragma unroll
                for (int u = 0; u < PCR_WALK_UNROLL; ++u) {
                    const double d = cand_d2<XF>(xf, qx, qy, qz, c[u]);
                    const uint32_t ci = (uint32_t)c[u].w;
                    if (((uint32_t)u < left) & better(d, ci, wd, wi)) {
                        top.insert(d, ci);
                        wd = top.worst_d();
                        wi = top.worst_i();
                    }
                }
            }
            alive = (p < e) | (ri < cnt);
        }
    }
    // ---- can anything outside the 5x5x5 cells still matter? (grid_search's rule)
    if ((x0 == 0) & (x1 == gx - 1) & (y0 == 0) & (y1 == gy - 1) & (z0 == 0) & (z1 == gz - 1)) return true;
    double ox = fmax(fmax(g.lo[0] - sx, sx - g.hi[0]), 0.0) - margin;
    double oy = fmax(fmax(g.lo[1] - sy, sy - g.hi[1]), 0.0) - margin;
    double oz = fmax(fmax(g.lo[2] - sz, sz - g.hi[2]), 0.0) - margin;
    ox = ox > 0.0 ? ox : 0.0;
    oy = oy > 0.0 ? oy : 0.0;
    oz = oz > 0.0 ? oz : 0.0;
    const double b2 = box_bound2(g, sx, sy, sz, margin, ox * ox, oy * oy, oz * oz, x0, x1, y0, y1, z0, z1) * bound_scale;
    return (wd < b2) | (max_d2 <= b2);
}

// Everything beyond the 3x3x3 cells: the 5x5x5 cells a layer at a time, then -- for what that leaves open -- the shell
// search, its first shell at the radius grid_search's own rule for the next radius gives with what is known by now
// (double, but never past what the ball of everything that can still matter needs).  Returns what grid_search returns.
template <int KC, bool WINDOW, bool XF, int LANE_RMAX, int RS, bool CT>
__device__ __forceinline__ int shells_beyond_ring1(const GridView &g, const Pt4 *__restrict__ cand, double qx, double qy,
                                                   double qz, double sx, double sy, double sz, double max_d2,
                                                   double bound_scale, double extra_margin, TopK<KC> &top, uint2 *rowtab,
                                                   const Rigid *xf = nullptr) {
    if (shell2_batched<KC, XF, RS, CT>(g, cand, qx, qy, qz, sx, sy, sz, max_d2, bound_scale, extra_margin, top, rowtab, xf)) return 0;
    int r_first = 4;
    const double reach2 = fmin(top.worst_d(), max_d2);
    if (reach2 < PCR_INF) {
        const double margin = g.margin + extra_margin + 9.094947017729282e-13 * (fabs(sx) + fabs(sy) + fabs(sz));
        const double cells_needed = (sqrt(reach2 / bound_scale) + margin) * g.inv_h + 2.0;
        r_first = cells_needed < 4.0 ? max((int)cells_needed, 3) : 4;
    }
    return grid_search<KC, WINDOW, XF, LANE_RMAX, CT>(g, cand, qx, qy, qz, sx, sy, sz, max_d2, bound_scale, extra_margin, top, 3, xf,
                                                      r_first);
}

// The per-lane search the kernels use: pruned ring 1, then (rarely) the shells beyond it.
// BATCH2 = false: the shells are grid_search's from radius 2 on (the filtered step kernel's reverse fall-back: with the
// layer-at-a-time scan compiled in as well, that kernel spills 78 registers instead of 56 and its 1 M iteration takes
// 122.8 us instead of 120.2)
template <int KC, bool XF = false, int RS = PCR_ROW_STRIDE, bool BATCH2 = true>
__device__ __forceinline__ void lane_search(const GridView &g, const Pt4 *__restrict__ cand, double qx, double qy,
                                            double qz, double sx, double sy, double sz, double max_d2,
                                            double bound_scale, double extra_margin, TopK<KC> &top,
                                            uint2 *rowtab, const Rigid *xf = nullptr) {
    DIAG_LANE_ADD(8 * (KC == 1) + 5, 1);  // lane searches
    const bool fin1 = ring1_pruned<KC, XF, RS>(g, cand, qx, qy, qz, sx, sy, sz, max_d2, bound_scale, extra_margin, top, rowtab, xf);
    if (!fin1) {
        if constexpr (BATCH2)
            shells_beyond_ring1<KC, false, XF, 0, RS, false>(g, cand, qx, qy, qz, sx, sy, sz, max_d2, bound_scale, extra_margin, top, rowtab, xf);
        else
            grid_search<KC, false, XF>(g, cand, qx, qy, qz, sx, sy, sz, max_d2, bound_scale, extra_margin, top, 2, xf);
    }
}

// squared distance from the grid-frame query to a level's box of points, shrunk by the margin (0 inside)
__device__ __forceinline__ double level_box_dist2(const GridView &g, double sx, double sy, double sz, double margin) {
    double ox = fmax(fmax(g.lo[0] - sx, sx - g.hi[0]), 0.0) - margin;
    double oy = fmax(fmax(g.lo[1] - sy, sy - g.hi[1]), 0.0) - margin;
    double oz = fmax(fmax(g.lo[2] - sz, sz - g.hi[2]), 0.0) - margin;
    ox = ox > 0.0 ? ox : 0.0;
    oy = oy > 0.0 ? oy : 0.0;
    oz = oz > 0.0 ? oz : 0.0;
    return ox * ox + oy * oy + oz * oz;
}

// Has the 3x3x3 walk around the query's cell settled a level for the K-th distance wd?  The bound of the end of
// ring1_pruned, from the geometry alone -- asked again later, when other levels have shrunk wd -- and sharper for a
// query outside the level's box: whatever lies beyond a face of the 3x3x3 box is also at least as far away
// along the OTHER axes as the box of the points is.
__device__ __forceinline__ bool ring1_settles(const GridView &g, double sx, double sy, double sz, double margin,
                                              double bound_scale, double wd, double max_d2) {
    const double s[3] = {sx, sy, sz};
    double face2[3][2], o2[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int c = cell_coord(s[a], g.lo[a], g.inv_h, g.g[a]);
        double gl = s[a] - (g.lo[a] + (double)c * g.h) - margin;
        double gh = (g.lo[a] + (double)(c + 1) * g.h) - s[a] - margin;
        gl = gl > 0.0 ? gl : 0.0;
        gh = gh > 0.0 ? gh : 0.0;
        face2[a][0] = c > 1 ? (gl + g.h) * (gl + g.h) : PCR_INF;
        face2[a][1] = c < g.g[a] - 2 ? (gh + g.h) * (gh + g.h) : PCR_INF;
        double o = fmax(fmax(g.lo[a] - s[a], s[a] - g.hi[a]), 0.0) - margin;
        o = o > 0.0 ? o : 0.0;
        o2[a] = o * o;
    }
    double b2 = fmin(face2[0][0], face2[0][1]) + (o2[1] + o2[2]);
    b2 = fmin(b2, fmin(face2[1][0], face2[1][1]) + (o2[0] + o2[2]));
    b2 = fmin(b2, fmin(face2[2][0], face2[2][1]) + (o2[0] + o2[1]));
    b2 *= bound_scale * (1.0 - 4.440892098500626e-16);
    return (wd < b2) | (max_d2 <= b2);
}

// ------------------------------------------------------------------ wave-cooperative continuation of a long search
// One lane's value to every lane (src is wave-uniform).
__device__ __forceinline__ uint32_t wave_bcast(uint32_t v, int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ __forceinline__ double wave_bcast(double v, int src) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)b, src), hi = __builtin_amdgcn_readlane((uint32_t)(b >> 32), src);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

#ifndef PCR_LANE_RMAX
#define PCR_LANE_RMAX 4  // shells up to this radius are walked by the query's own lane (at most 9 x 9 rows); measured 2..10
#endif
static_assert(PCR_LANE_RMAX >= 2 && PCR_LANE_RMAX <= 15, "packed four bits per level");
#ifndef PCR_COOP_INLINE
#define PCR_COOP_INLINE __forceinline__
#endif
// A query that the walk of its own 5x5x5 cells has not settled hovers far (in cells) from the level's data: what is
// left is a ball of hundreds to thousands of rows, nearly all empty, and a lane walking them one dependent cell-table
// load after the other runs for milliseconds while the other 63 lanes of its wave -- and, at the end of a launch, most
// of the GPU -- wait.  Here the WHOLE WAVE finishes that one query: every argument is wave-uniform, the rows of a shell
// are dealt out to the lanes (row i of the shell's box to lane i mod 64), each lane keeps the K best of its rows, and
// after every shell the 64 lists are merged into the query's top-K (*shared, the same in every lane).  The shells, the
// ball, the chords, the pruning and the stopping rule are grid_search's, so the result is the same exact K-NN.
//   r_done: the cube of this radius around the query's cell has been scanned already.
template <int KC, bool XF, bool CT = false>
__device__ PCR_COOP_INLINE void coop_shells(const GridView *gp, const Pt4 *__restrict__ cand, double qx, double qy, double qz,
                                            double sx, double sy, double sz, double max_d2, double bound_scale,
                                            double extra_margin, TopK<KC> *shared, int r_done, const Rigid *xf) {
    const int lane = (int)(threadIdx.x & 63u);
    const Pt4 *__restrict__ order = gp->pts;
    const int gx = gp->g[0], gy = gp->g[1], gz = gp->g[2];
    const double h = gp->h, inv_h = gp->inv_h, lox = gp->lo[0], loy = gp->lo[1], loz = gp->lo[2];
    const int cx = cell_coord(sx, lox, inv_h, gx), cy = cell_coord(sy, loy, inv_h, gy), cz = cell_coord(sz, loz, inv_h, gz);
    const double margin = gp->margin + extra_margin + 9.094947017729282e-13 * (fabs(sx) + fabs(sy) + fabs(sz));
    double ox = fmax(fmax(lox - sx, sx - gp->hi[0]), 0.0) - margin;
    double oy = fmax(fmax(loy - sy, sy - gp->hi[1]), 0.0) - margin;
    double oz = fmax(fmax(loz - sz, sz - gp->hi[2]), 0.0) - margin;
    ox = ox > 0.0 ? ox : 0.0;
    oy = oy > 0.0 ? oy : 0.0;
    oz = oz > 0.0 ? oz : 0.0;
    const double ox2 = ox * ox, oy2 = oy * oy, oz2 = oz * oz;
    TopK<KC> top = *shared;
    int px0 = max(cx - r_done, 0), px1 = min(cx + r_done, gx - 1);
    int py0 = max(cy - r_done, 0), py1 = min(cy + r_done, gy - 1);
    int pz0 = max(cz - r_done, 0), pz1 = min(cz + r_done, gz - 1);
    const int rmax = max(max(gx, gy), gz);
    const double inv_bs = 1.0 / bound_scale;
    int r = r_done;
    for (;;) {
        // next radius (grid_search's rule): double, but never past what the ball of everything that can still matter needs
        {
            const double reach2 = fmin(top.worst_d(), max_d2);
            int rn = 2 * r;
            if (reach2 < PCR_INF) {
                const double cells_needed = (sqrt(reach2 * inv_bs) + margin) * inv_h + 2.0;
                const int need = cells_needed < (double)rmax ? (int)cells_needed : rmax;
                rn = max(r + 1, min(rn, need));
            }
            r = min(rn, rmax);
        }
        const int x0 = max(cx - r, 0), x1 = min(cx + r, gx - 1);
        const int y0 = max(cy - r, 0), y1 = min(cy + r, gy - 1);
        const int z0 = max(cz - r, 0), z1 = min(cz + r, gz - 1);
        const int ny = y1 - y0 + 1;
        const uint32_t nrows = (uint32_t)ny * (uint32_t)(z1 - z0 + 1);
        // the lane's own list starts as K copies of the query's K-th best: whatever enters it beats that
        TopK<KC> mine;
#pragma unroll
        for (int j = 0; j < KC; ++j) {
            mine.d[j] = top.worst_d();
            mine.idx[j] = PCR_NO_INDEX;
        }
        double wd = mine.worst_d();
        uint32_t wi = mine.worst_i();
        for (uint32_t row = (uint32_t)lane; row < nrows; row += 64u) {
            const uint32_t qz_i = row / (uint32_t)ny;  // row = (rz - z0) * ny + (ry - y0)
            const int rz = z0 + (int)qz_i, ry = y0 + (int)(row - qz_i * (uint32_t)ny);
            const double dz = slab_gap(sz, loz, rz, h, margin), dy = slab_gap(sy, loy, ry, h, margin);
            const double rem = fmin(wd, max_d2) * inv_bs - dz * dz - dy * dy;
            if (!(rem >= 0.0)) continue;
            const double w = ball_half_width(rem) + margin;
            const double lo_f = (sx - w - lox) * inv_h - 1.0, hi_f = (sx + w - lox) * inv_h + 1.0;
            const bool inner = (ry >= py0) & (ry <= py1) & (rz >= pz0) & (rz <= pz1);
            const size_t base = ((size_t)rz * gy + ry) * gx;
#pragma unroll 1
            for (int seg = 0; seg < 2; ++seg) {
                int xs = seg == 0 ? x0 : px1 + 1, xe = seg == 0 ? (inner ? px0 - 1 : x1) : x1;
                if ((seg == 1) & !inner) break;
                if (xs > xe) continue;
                const double lo_c = fmin(fmax(lo_f, (double)xs), (double)xe + 1.0);
                const double hi_c = fmax(fmin(hi_f, (double)xe), (double)xs - 1.0);
                xs = (int)lo_c;
                xe = (int)hi_c;
                if (xs > xe) continue;
                uint32_t p, e;
                cell_range<CT>(*gp, base + xs, base + xe + 1, p, e);
                if (e - p > PCR_LONG_RANGE && e > p) {
                    windowed_scan<KC, XF>(order, cand, p, e, qx, qy, qz, sx, margin, bound_scale, mine, wd, wi, xf);
                } else {
                    for (uint32_t i = p; i < e; ++i) {
                        const Pt4 c = cand[i];
                        const double d = cand_d2<XF>(xf, qx, qy, qz, c);
                        const uint32_t ci = (uint32_t)c.w;
                        if (better(d, ci, wd, wi)) {
                            mine.insert(d, ci);
                            wd = mine.worst_d();
                            wi = mine.worst_i();
                        }
                    }
                }
            }
        }
        // ---- merge: K rounds of "the best head of the 64 lists" ----
        if (__any(mine.idx[0] != PCR_NO_INDEX)) {
#pragma unroll 1
            for (int k = 0; k < KC; ++k) {
                double d = mine.d[0];
                uint32_t i = mine.idx[0];
#pragma unroll
                for (int m = 32; m > 0; m >>= 1) {
                    const double od = __shfl_xor(d, m, 64);
                    const uint32_t oi = (uint32_t)__shfl_xor((int)i, m, 64);
                    const bool take = better(od, oi, d, i);
                    d = take ? od : d;
                    i = take ? oi : i;
                }
                if (i == PCR_NO_INDEX) break;  // (uniform) nothing real is left in any list
                if ((mine.idx[0] == i) & (mine.d[0] == d)) {  // the winner's list moves up
#pragma unroll
                    for (int j = 0; j + 1 < KC; ++j) {
                        mine.d[j] = mine.d[j + 1];
                        mine.idx[j] = mine.idx[j + 1];
                    }
                    mine.d[KC - 1] = PCR_INF;
                    mine.idx[KC - 1] = PCR_NO_INDEX;
                }
                if (better(d, i, top.worst_d(), top.worst_i())) top.insert(d, i);
            }
        }
        // ---- can anything outside the scanned box still matter? (grid_search's rule) ----
        const bool all = (x0 == 0) & (x1 == gx - 1) & (y0 == 0) & (y1 == gy - 1) & (z0 == 0) & (z1 == gz - 1);
        if (all) break;
        double b2 = box_bound2(*gp, sx, sy, sz, margin, ox2, oy2, oz2, x0, x1, y0, y1, z0, z1);
        b2 *= bound_scale;
        if (top.worst_d() < b2 || max_d2 <= b2) break;
        px0 = x0;
        px1 = x1;
        py0 = y0;
        py1 = y1;
        pz0 = z0;
        pz1 = z1;
        if (r >= rmax) break;  // (the box was the whole grid: `all` has caught it; belt and braces)
    }
    *shared = top;
}

// (multi_lane_search, the search over all levels of an index, lives in fast_search.h: its first pass uses the filtered walk)

// Brute-force scan of one LDS tile (all lanes read the same address: LDS broadcast).
template <int KC>
__device__ __forceinline__ void tile_scan(const Pt4 *tile, int count, double qx, double qy, double qz,
                                          TopK<KC> &top) {
    double wd = top.worst_d();
    uint32_t wi = top.worst_i();
    for (int j = 0; j < count; ++j) {
        const Pt4 c = tile[j];
        const double d = sqdist3(qx, qy, qz, c.x, c.y, c.z);
        const uint32_t ci = (uint32_t)c.w;
        if (better(d, ci, wd, wi)) {
            top.insert(d, ci);
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

// The same within chunks of `chunk` consecutive logical blocks (about what the GPU runs at once): every XCD works on
// an eighth of EVERY chunk.  For clouds whose cost per block varies along the processing order (nested levels: the
// fine levels' blocks take ten times as long as the coarse ones') -- with the plain remap the XCD that draws the
// expensive end of the order finishes long after the others.
__device__ __forceinline__ uint32_t xcd_remap_chunked(uint32_t bid, uint32_t nb, uint32_t chunk = 1024u) {
    const uint32_t c0 = (bid / chunk) * chunk;
    return c0 + xcd_remap(bid - c0, min(chunk, nb - c0));
}

}  // namespace pcr
