// The filtered 3x3x3 walk: fp32 keys select, fp64 decides -- gfx950.
//
// What it computes is exactly what ring1_pruned (knn_device.h) computes: the K smallest candidates of the 27 cells
// around the query under the total order (d2, original index), d2 in the reference's fp64 expression
// (Common/KNNSearch.h:137-143).  How: every candidate is first looked at through its 16-byte fp32 "shadow"
// (GridView::shadow): six fp32 operations give a squared distance in key units, truncated to a 32-bit integer KEY
// whose low 12 bits carry where the candidate sits (row slot, offset).  The lane keeps the KC+2 smallest keys in
// registers -- a sorted list updated by one v_med3_u32 per slot, no branch, no payload shuffling -- and nothing else
// happens per candidate.  At the end the candidates behind the KC smallest keys are evaluated in fp64 (one 32-byte
// gather each), put in exact order, and the result is VERIFIED: every candidate that is not on the list has a key
// >= the list's next key, hence an exact distance >= that key minus the error bound key_delta; if the exact K-th
// distance is strictly below that, no unlisted candidate can enter or tie, and the exact top-K is final.  If not (two
// candidates within ~1e-4 relative of each other around rank K, a handful per million) one more list entry is
// evaluated, and if that does not settle it either the caller repeats the search with the all-fp64 walk.  Rows and
// outer cells are pruned in key space with the same error bound, so a pruned candidate is provably farther than the
// exact K-th.  Indices are therefore bit-identical to the fp64 walk's; tests compare the two kernels directly.
#pragma once
#include "knn_device.h"

namespace pcr {

constexpr uint32_t KEY_SENTINEL = 0xFFFFFFFFu;
constexpr uint32_t KEY_DIST_MASK = 0xFFFFF000u;  // 20 bits of distance; 12 bits of payload: row slot << 8 | offset in row
#ifndef PCR_FAST_U
#define PCR_FAST_U 4
#endif

__device__ __forceinline__ uint32_t sat_add(uint32_t a, uint32_t b) { return __builtin_elementwise_add_sat(a, b); }
__device__ __forceinline__ uint32_t sat_sub(uint32_t a, uint32_t b) { return __builtin_elementwise_sub_sat(a, b); }
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// float -> u32, truncating and SATURATING (what v_cvt_u32_f32 does; a C cast is undefined beyond the range)
__device__ __forceinline__ uint32_t cvt_u32_sat(float v) {
    uint32_t r;
    asm("v_cvt_u32_f32_e32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}
__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) {
    return max(min(a, b), min(max(a, b), c));  // selected as one v_med3_u32
}

// floor / ceiling in key units of a squared distance in cloud units, conservative and saturating
__device__ __forceinline__ uint32_t key_floor(double v, double S) {
    const double k = fmin(v * S * (1.0 - 9.5367431640625e-07), 4294967295.0);
    return (uint32_t)k;
}
__device__ __forceinline__ uint32_t key_ceil(double v, double S) {
    const double k = fmin(v * S * (1.0 + 9.5367431640625e-07) + 2.0, 4294967295.0);
    return (uint32_t)k;
}

// The KP smallest keys seen, ascending.  insert: t[j] <- med3(t[j-1], t[j], c) from the top down, t[0] <- min.
template <int KP>
struct TopKeys {
    uint32_t t[KP];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int j = 0; j < KP; ++j) t[j] = KEY_SENTINEL;
    }
    __device__ __forceinline__ void insert(uint32_t c) {
#pragma unroll
        for (int j = KP - 1; j >= 1; --j) t[j] = umed3(t[j - 1], t[j], c);
        t[0] = min(t[0], c);
    }
};

// One lane, one query, the 3x3x3 cells around it.
//   g          a single-level grid WITH shadows (g.shadow != NULL)
//   exact_pts  the fp64 records the distances that decide are evaluated against (g.pts)
//   (qx,qy,qz) the query in the frame of those distances; (sx,sy,sz) the query in the frame of the grid
//   XF / xf    the fp64 records are moved by (R,T) before the distance (reverse search of the ICP step); the shadows
//              stay in the grid's frame, and `delta` then also covers R's deviation from an isometry
//   max_d2     as in grid_search: candidates at or beyond are of no interest
//   bound_d2   a squared distance the K-th best is known not to exceed (+inf: nothing known); prunes from the start
//   delta      key-space error bound for this search
//   k          the rank the caller will read (1..KC)
//   top        initialised by the caller (init(max_d2), or seeded); on return true it holds the exact result
// Returns false when this search cannot vouch for the result (not settled within ring 1, a cell row too long for the
// payload, a query far outside the grid, an unresolved near-tie): the caller re-initialises `top` and runs lane_search.
template <int KC, bool XF>
__device__ __forceinline__ bool fast_ring1(const GridView &g, const Pt4 *__restrict__ exact_pts, double qx, double qy,
                                           double qz, double sx, double sy, double sz, double max_d2, double bound_d2,
                                           double bound_scale, double extra_margin, uint32_t delta, int k,
                                           TopK<KC> &top, uint2 *rowtab, const Rigid *xf) {
    constexpr int KP = KC + 2;
    constexpr int U = PCR_FAST_U;
    const uint32_t *__restrict__ cs = g.cs;
    const float4 *__restrict__ shadow = g.shadow;
    const int gx = g.g[0], gy = g.g[1], gz = g.g[2];
    const int cx = cell_coord(sx, g.lo[0], g.inv_h, gx);
    const int cy = cell_coord(sy, g.lo[1], g.inv_h, gy);
    const int cz = cell_coord(sz, g.lo[2], g.inv_h, gz);
    const double margin = g.margin + extra_margin + 9.094947017729282e-13 * (fabs(sx) + fabs(sy) + fabs(sz));
    const double S = g.key_S;
    // ---- geometry in fp64, exactly as ring1_pruned: gaps to the faces of the own cell, the settle bound
    double gl, gh;
    gl = sx - (g.lo[0] + (double)cx * g.h) - margin;
    gh = (g.lo[0] + (double)(cx + 1) * g.h) - sx - margin;
    gl = gl > 0.0 ? gl : 0.0;
    gh = gh > 0.0 ? gh : 0.0;
    const double fxl = gl, fxh = gh;
    const uint32_t gxl_k = key_floor(gl * gl * bound_scale, S), gxh_k = key_floor(gh * gh * bound_scale, S);
    gl = sy - (g.lo[1] + (double)cy * g.h) - margin;
    gh = (g.lo[1] + (double)(cy + 1) * g.h) - sy - margin;
    gl = gl > 0.0 ? gl : 0.0;
    gh = gh > 0.0 ? gh : 0.0;
    const double fyl = gl, fyh = gh;
    const uint32_t gyl_k = key_floor(gl * gl * bound_scale, S), gyh_k = key_floor(gh * gh * bound_scale, S);
    gl = sz - (g.lo[2] + (double)cz * g.h) - margin;
    gh = (g.lo[2] + (double)(cz + 1) * g.h) - sz - margin;
    gl = gl > 0.0 ? gl : 0.0;
    gh = gh > 0.0 ? gh : 0.0;
    const double fzl = gl, fzh = gh;
    const uint32_t gzl_k = key_floor(gl * gl * bound_scale, S), gzh_k = key_floor(gh * gh * bound_scale, S);
    double settle_b2 = PCR_INF;
    if (cx > 1) settle_b2 = fmin(settle_b2, (fxl + g.h) * (fxl + g.h));
    if (cx < gx - 2) settle_b2 = fmin(settle_b2, (fxh + g.h) * (fxh + g.h));
    if (cy > 1) settle_b2 = fmin(settle_b2, (fyl + g.h) * (fyl + g.h));
    if (cy < gy - 2) settle_b2 = fmin(settle_b2, (fyh + g.h) * (fyh + g.h));
    if (cz > 1) settle_b2 = fmin(settle_b2, (fzl + g.h) * (fzl + g.h));
    if (cz < gz - 2) settle_b2 = fmin(settle_b2, (fzh + g.h) * (fzh + g.h));
    settle_b2 *= bound_scale * (1.0 - 4.440892098500626e-16);
    // ---- the query in key space; key_delta holds for queries within 4 cells of the box of the points
    const double pad = 4.0 * g.h;
    bool ok = (sx >= g.lo[0] - pad) & (sx <= g.hi[0] + pad) & (sy >= g.lo[1] - pad) & (sy <= g.hi[1] + pad) &
              (sz >= g.lo[2] - pad) & (sz <= g.hi[2] + pad);
    const float Qx = (float)((sx - g.lo[0]) * g.key_sc), Qy = (float)((sy - g.lo[1]) * g.key_sc),
                Qz = (float)((sz - g.lo[2]) * g.key_sc);
    // what is known about the K-th distance before the walk: the caller's bound, and the reach of interest
    const uint32_t bound_k = key_ceil(fmin(bound_d2, max_d2), S);
    const uint32_t UB0 = sat_add(bound_k, delta);
    // ---- phase 1: the (up to) nine rows, straight-line -- every cell-table load of the search is in flight at once.
    // The four boundaries cs[c-1..c+2] of a row's three cells come with ONE 16-byte buffer load (the table is padded by
    // an entry either side, pcr_core.hip; a row outside the grid may read anything or nothing -- it is masked).
    // A row's entry: x = position of the first point of its cell cx-1, y = the populations of cells cx-1, cx, cx+1 (one
    // byte each) and, in the top byte, the row's slab distance in key units >> 24, rounded down.  Non-empty rows are
    // appended to the lane's column of the LDS table in order of slab-distance class (centre, 4 faces, 4 edges).
    const uint32_t cells = (uint32_t)gx * (uint32_t)gy * (uint32_t)gz;
    const __amdgpu_buffer_rsrc_t rs_cs = __builtin_amdgcn_make_buffer_rsrc((void *)(cs - 1), 0, (int)((cells + 3u) * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_sh = __builtin_amdgcn_make_buffer_rsrc((void *)shadow, 0, (int)(g.n * 16u), 0x00020000);
    uint32_t cnt = 0;
    {
        constexpr int DY[9] = {0, -1, 1, 0, 0, -1, 1, -1, 1};
        constexpr int DZ[9] = {0, 0, 0, -1, 1, -1, -1, 1, 1};
        const uint32_t cb4 = (((uint32_t)cz * (uint32_t)gy + (uint32_t)cy) * (uint32_t)gx + (uint32_t)cx) * 4u;
        const int zstep = gx * gy;
        const bool has_xl = cx > 0, has_xh = cx < gx - 1;
        const bool y_lo = cy > 0, y_hi = cy < gy - 1, z_lo = cz > 0, z_hi = cz < gz - 1;
        u32x4 b[9];
#pragma unroll
        for (int r = 0; r < 9; ++r)
            b[r] = __builtin_amdgcn_raw_buffer_load_b128(rs_cs, cb4 + (uint32_t)(4 * (DY[r] * gx + DZ[r] * zstep)), 0, 0);
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            const uint32_t rowk = sat_add(DY[r] < 0 ? gyl_k : (DY[r] > 0 ? gyh_k : 0u), DZ[r] < 0 ? gzl_k : (DZ[r] > 0 ? gzh_k : 0u));
            const bool inside = (DY[r] < 0 ? y_lo : (DY[r] > 0 ? y_hi : true)) & (DZ[r] < 0 ? z_lo : (DZ[r] > 0 ? z_hi : true));
            const bool open = inside & !(rowk > UB0);
            const uint32_t p0 = has_xl ? b[r].x : b[r].y, p1 = b[r].y, p2 = b[r].z, p3 = has_xh ? b[r].w : b[r].z;
            const uint32_t tot = p3 - p0;
            ok &= !(open & (tot > 255u));  // the payload holds 8 bits of offset: a longer row is the fp64 walk's business
            if (open & (tot != 0u)) {
                rowtab[cnt * PCR_ROW_STRIDE] = make_uint2(p0, (p1 - p0) | ((p2 - p1) << 8) | ((p3 - p2) << 16) | (rowk & 0xFF000000u));
                ++cnt;
            }
        }
    }
    // ---- phase 2: the candidates of all rows as ONE stream, U per trip.  A trip takes what is left of the current
    // row and, when that does not fill it, opens the next row and continues there -- a lane's trips are about
    // (candidates + pruned rows) / U, not one or more per row.  Slot u of a trip reads position (u < rem ? p : p2 - rem)
    // + u: an unused slot may read anything (or, past the end of the array, nothing): its key is the sentinel.
    TopKeys<KP> T;
    T.init();
    uint32_t p = 0, e = 0, pb = 0, ri = 0;
    bool alive = ok & (cnt != 0u);
    DIAG_LANE_ADD(16 + 4 * XF, 1);                    // filtered searches
    DIAG_LANE_ADD(17 + 4 * XF, ok ? 0 : 1);           // ... that cannot start (far query, long row)
    while (alive) {
        DIAG_ADD(24 + XF, 1);  // wave trips
        const uint32_t rem = e - p;
        uint32_t p2 = 0, e2 = 0, pb2 = 0;
        if ((rem < (uint32_t)U) & (ri < cnt)) {
            const uint2 row = rowtab[ri * PCR_ROW_STRIDE];
            const uint32_t n0 = row.y & 255u, n1 = (row.y >> 8) & 255u, n2 = (row.y >> 16) & 255u, rk = row.y & 0xFF000000u;
            const uint32_t UB = sat_add(min(T.t[KC - 1], bound_k) | ~KEY_DIST_MASK, delta);
            const bool trim_lo = sat_add(rk, gxl_k) > UB, trim_hi = sat_add(rk, gxh_k) > UB;
            p2 = row.x + (trim_lo ? n0 : 0u);
            e2 = row.x + n0 + n1 + (trim_hi ? 0u : n2);
            e2 = rk > UB ? p2 : e2;  // the whole row is provably farther than the K-th best
            pb2 = row.x - (ri << 8);
            ++ri;
        }
        const uint32_t rem2 = e2 - p2, tot = rem + rem2;
        const uint32_t base2 = p2 - rem, d1 = p - pb, d2 = base2 - pb2;
        u32x4 c[U];
        uint32_t pay[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool in1 = (uint32_t)u < rem;
            c[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_sh, ((in1 ? p : base2) << 4) + 16u * (uint32_t)u, 0, 0);
            pay[u] = (in1 ? d1 : d2) + (uint32_t)u;
        }
        if (rem >= (uint32_t)U) {
            p += (uint32_t)U;
        } else {
            p = p2 + min((uint32_t)U - rem, rem2);
            e = e2;
            pb = pb2;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float dx = Qx - __uint_as_float(c[u].x), dy = Qy - __uint_as_float(c[u].y), dz = Qz - __uint_as_float(c[u].z);
            float D = dx * dx;
            D = fmaf(dy, dy, D);
            D = fmaf(dz, dz, D);
            T.insert((uint32_t)u < tot ? ((cvt_u32_sat(D) & KEY_DIST_MASK) | pay[u]) : KEY_SENTINEL);
        }
        alive = (p < e) | (ri < cnt);
    }
    // ---- phase 3: fp64 decides.  The candidates behind the KC smallest keys, exactly; then the verification.
    auto evaluate = [&](uint32_t key) {
        if (key != KEY_SENTINEL) {
            const uint32_t pos = rowtab[((key >> 8) & 15u) * PCR_ROW_STRIDE].x + (key & 255u);
            const Pt4 cnd = exact_pts[pos];
            const double d = cand_d2<XF>(xf, qx, qy, qz, cnd);
            const uint32_t ci = (uint32_t)cnd.w;
            if (better(d, ci, top.worst_d(), top.worst_i())) top.insert(d, ci);
        }
    };
    // exact squared distance every candidate with a key >= `key` is at least at
    auto floor_d2 = [&](uint32_t key) {
        return key == KEY_SENTINEL ? PCR_INF
                                   : (double)sat_sub(key & KEY_DIST_MASK, delta) * g.key_invS * (1.0 - 9.313225746154785e-10);
    };
#pragma unroll
    for (int j = 0; j < KC; ++j) evaluate(T.t[j]);
    double ek;
    uint32_t ik;
    top.slot(k, ek, ik);
    double lb = floor_d2(T.t[KC]);
    bool verified = (ek < lb) | (max_d2 <= lb);
    DIAG_LANE_ADD(18 + 4 * XF, (ok & !verified) ? 1 : 0);  // first verification failed
    if (__any(ok & !verified)) {
        DIAG_ADD(26 + XF, 1);  // wave-level second stages
        if (!verified) evaluate(T.t[KC]);
        top.slot(k, ek, ik);
        lb = floor_d2(T.t[KC + 1]);
        verified |= (ek < lb) | (max_d2 <= lb);
    }
    const bool settled = (ek < settle_b2) | (max_d2 <= settle_b2);
    DIAG_LANE_ADD(19 + 4 * XF, (ok & verified & settled) ? 0 : 1);  // handed to the fp64 walk
    DIAG_LANE_ADD(28 + XF, cnt);                                    // rows in the table
    return ok & verified & settled;
}

}  // namespace pcr
