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
#define PCR_FAST_U 6  // shadow records per trip of phase 2.  Measured on the 1 M x 1 M step (us per iteration): 3: 142.1, 4: 136.2,
                      // 5: 133.4, 6: 132.7, 7: 134.4, 8: 136.8 -- fewer dependent trips against more wasted slots (the
                      // optimum was 4 before the searches handed their winners over and the fallback was seeded; the
                      // free-query K-NN kernel, unbounded and with three tries per trip, still does best with 4)
#endif

__device__ __forceinline__ uint32_t sat_add(uint32_t a, uint32_t b) { return __builtin_elementwise_add_sat(a, b); }
__device__ __forceinline__ uint32_t sat_sub(uint32_t a, uint32_t b) { return __builtin_elementwise_sub_sat(a, b); }
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
// lanes 2i and 2i+1 swap a register (DPP quad_perm:[1,0,3,2]); every lane of the wave must be executing
__device__ __forceinline__ uint32_t quad_swap1(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false);
}
// float -> u32, truncating and SATURATING (what v_cvt_u32_f32 does; a C cast is undefined beyond the range)
__device__ __forceinline__ uint32_t cvt_u32_sat(float v) {
    uint32_t r;
    asm("v_cvt_u32_f32_e32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}
__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) {
    return max(min(a, b), min(max(a, b), c));  // selected as one v_med3_u32
}

// A cell's edge in key-space units: key_sc = KEY_CELL / h in every grid (pcr_core.hip).  An integer, and 1.5 cells stay
// below 2^15: y and z differences of everything the 3x3x3 walk can settle fit a signed 16-bit lane.  Keys of interest
// (squared distances up to 2.5 h^2, above everything that walk can settle) lie below KEY_RANGE = 1.19e9; the largest key
// a row can produce, (2 h)^2 + 2 * 32768^2 = 4.06e9, still fits 32 bits and stays below the sentinel.
#define KEY_CELL 21844.0f
constexpr double KEY_RANGE = 2.5 * 21844.0 * 21844.0;
constexpr double KEY_MAX_ROOT = 1.5811388300841898 * 21844.0;  // sqrt(KEY_RANGE)
typedef short s16x2 __attribute__((ext_vector_type(2)));
// key-space integer coordinate of a cloud coordinate (the expression shadow_kernel uses: pcr_core.hip)
__device__ __forceinline__ int key_coord(double v, double lo, double sc) { return __double2int_rn((v - lo) * sc); }
// the squared distance, in key units, between the query (QX; Qyz = Y16 | Z16 << 16) and a shadow record: 4 instructions
// (v_sub_u32, v_pk_sub_i16, v_mul_i32_i24, v_dot2c_i32_i16)
__device__ __forceinline__ uint32_t shadow_key_d2(uint32_t QX, uint32_t Qyz, uint32_t rec_x, uint32_t rec_yz) {
    const int dx = (int)(QX - rec_x);
    const s16x2 dyz = __builtin_bit_cast(s16x2, Qyz) - __builtin_bit_cast(s16x2, rec_yz);  // v_pk_sub_i16, wrapping
    // (dx^2 first, the dot product accumulates onto it: v_mul_i32_i24 + v_dot2c_i32_i16 -- the other way round the
    // accumulating form of the dot product wants a zeroed register first: one instruction more per candidate)
    return (uint32_t)__builtin_amdgcn_sdot2(dyz, dyz, __mul24(dx, dx), false);
}
// One axis: the cell the key-space coordinate Q falls in (clamped to the grid), and the distances from Q to that cell's
// low and high face, each shrunk by eps and clamped at 0.
__device__ __forceinline__ void cell_and_gaps(float Q, int g, float eps, int &c, float &gl, float &gh) {
    float t = Q * (1.0f / KEY_CELL);
    t = fminf(fmaxf(t, 0.0f), (float)(g - 1));  // NaN -> 0
    c = (int)t;
    const float face = (float)c * KEY_CELL;
    gl = fmaxf(Q - face - eps, 0.0f);
    gh = fmaxf(face + KEY_CELL - Q - eps, 0.0f);
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

// Ranks of N (d2, index) pairs under the total order (d2, index): every pair is compared once; of two equal pairs
// (only placeholders are ever equal) the earlier one ranks first -- the ranks are a permutation of 0..N-1.
template <int N>
__device__ __forceinline__ void ranks_of(const double *d, const uint32_t *i, uint32_t *rank) {
#pragma unroll
    for (int a = 0; a < N; ++a) rank[a] = 0;
#pragma unroll
    for (int a = 0; a < N; ++a)
#pragma unroll
        for (int b = a + 1; b < N; ++b) {
            const bool b_first = better(d[b], i[b], d[a], i[a]);
            rank[a] += b_first ? 1u : 0u;
            rank[b] += b_first ? 0u : 1u;
        }
}
// the pair of rank k-1.  Everything statically indexed.
template <int N>
__device__ __forceinline__ void kth_of(const double *d, const uint32_t *i, const uint32_t *rank, int k, double &dk,
                                       uint32_t &ik) {
    dk = d[0];
    ik = i[0];
#pragma unroll
    for (int a = 1; a < N; ++a) {
        const bool sel = rank[a] == (uint32_t)(k - 1);
        dk = sel ? d[a] : dk;
        ik = sel ? i[a] : ik;
    }
}

// What a finished walk knows exactly: n candidates, their squared distances, indices and ranks (0 = nearest).  The ICP
// step reads one rank; the K-NN entry points write all of the first k.
template <int KC>
struct FastExact {
    double d[KC + 2];
    uint32_t i[KC + 2];
    uint32_t rank[KC + 2];
    int n;
    bool trusted;  // the walk could start and its ranking is verified: these ARE the best of the 3x3x3 cells (up to rank k)
    int cell[3];   // the cell the walk was centred on
};

// timing build only (PCR_BLOCK_TIMES): the shader clock at stage boundaries of one search, kept in scalar registers
struct StageClock {
    uint32_t t[4];
};
#ifdef PCR_BLOCK_TIMES
#define STAGE_CLOCK(clk, i) do { if (clk) (clk)->t[i] = (uint32_t)__builtin_readcyclecounter(); } while (0)
#else
#define STAGE_CLOCK(clk, i) do { } while (0)
#endif

// One lane, one query, the 3x3x3 cells around it.
//   g          a single-level grid WITH shadows (g.shadow != NULL)
//   exact_pts  the fp64 records the distances that decide are evaluated against (g.pts)
//   (qx,qy,qz) the query in the frame of those distances; (sx,sy,sz) the query in the frame of the grid
//   XF / xf    the fp64 records are moved by (R,T) before the distance (reverse search of the ICP step); the shadows
//              stay in the grid's frame, and `delta` then also covers R's deviation from an isometry
//   max_d2     as in grid_search: candidates at or beyond are of no interest
//   bound_d2   a squared distance the K-th best is known not to exceed (+inf: nothing known); prunes from the start
//   delta      key-space error bound for this search
//   k          the rank wanted (1..KC)
//   SEEDED     (seed_d, seed_i) is a member of the searched set whose exact distance the caller knows (the reverse
//              search of the ICP step: the sample's own image); it takes part in the ranking
//   out_d/out_i  on return true: the k-th smallest (d2, index) exactly -- (max_d2, PCR_NO_INDEX) if fewer than k
//              candidates lie below max_d2
//   win_pos    (optional) the same as a position in exact_pts (PCR_NO_INDEX: not listed) -- four registers where `win`
//              holds 8 * KC over the ranking; one round trip saved instead of two
//   win        (optional) receives the fp64 record of the point behind out_i when that point was one of the listed
//              candidates (w = its index; zeros otherwise, e.g. when the seed itself wins and was not listed): the
//              caller then need not gather it again -- two dependent round trips saved per search
// Returns false when this search cannot vouch for the result (not settled within ring 1, a cell row too long for the
// payload, a query far outside the grid, an unresolved near-tie): the caller runs lane_search instead.
// TRIES, U: see phase 2 (U: shadow records per trip).  RS: lanes per row of `rowtab` (ring1_pruned).  GV, RG: GridView and Rigid in whatever address space the caller keeps them (the step kernel: the constant one,
// so that every field is a scalar load at its point of use).
//   PAIR       the call is made by ALL 64 lanes of a wave together (`active`: this lane has a query): phase 3's gathers are
//              then shared by lane pairs -- see there
//   CT         the grid may carry a COMPACT cell table (GridView::cbits; levels of nested indices): phase 1 then reads
//              each row's occupancy bits (two 8-byte loads) and four consecutive starts of occupied cells (one 16-byte load)
template <int KC, bool XF, bool SEEDED, int TRIES = 1, int U = PCR_FAST_U, int RS = PCR_ROW_STRIDE, bool PAIR = false, bool CT = false, class GV, class RG>
__device__ __forceinline__ bool fast_ring1(const GV &g, const Pt4 *__restrict__ exact_pts, double qx, double qy,
                                           double qz, double sx, double sy, double sz, double max_d2, double bound_d2,
                                           double bound_scale, double extra_margin, uint32_t delta, int k,
                                           double seed_d, uint32_t seed_i, double &out_d, uint32_t &out_i,
                                           uint2 *rowtab, const RG *xf, FastExact<KC> *all = nullptr,
                                           StageClock *clk = nullptr, Pt4 *win = nullptr, uint32_t *win_pos = nullptr,
                                           bool active = true, uint32_t *diag_trips = nullptr) {
    constexpr int KP = KC + 2;
    const uint32_t *__restrict__ cs = g.cs;
    const uint2 *__restrict__ shadow = g.shadow;
    const int gx = g.g[0], gy = g.g[1], gz = g.g[2];
    const double S = g.key_S;
    // ---- the query in key space, and the geometry of its cell there, in fp32.  One unit ("root unit") is the square
    // root of a key unit; a cell's edge is KEY_CELL root units in every grid.  Everything derived below is a bound:
    // each gap is shrunk by eps, which covers the rounding of Q itself and of every fp32 operation on the way (a few
    // ulps of the largest coordinate; key_eps is 2^-22 of it), so a gap can only come out too small -- also when the
    // rounding of Q puts the query into the cell next to the one exact arithmetic would pick (the gaps are measured from
    // the faces of the cell actually used).  The walk is exact for whatever cell it is centred on.
    float Qx = (float)((sx - g.lo[0]) * g.key_sc), Qy = (float)((sy - g.lo[1]) * g.key_sc),
          Qz = (float)((sz - g.lo[2]) * g.key_sc);
    // (a literal zero margin -- every forward search -- folds away at compile time)
    const float eps = extra_margin == 0.0 ? g.key_eps : g.key_eps + (float)(extra_margin * g.key_sc) * 1.0001f;
    // key_delta holds for queries within 4 cells of the box of the points (NaN fails every comparison)
    bool ok = active & (Qx >= -4.0f * KEY_CELL) & (Qx <= g.key_ext[0] + 4.0f * KEY_CELL) & (Qy >= -4.0f * KEY_CELL) &
              (Qy <= g.key_ext[1] + 4.0f * KEY_CELL) & (Qz >= -4.0f * KEY_CELL) & (Qz <= g.key_ext[2] + 4.0f * KEY_CELL);
    // squared gaps -> keys, rounded down; bound_scale (< 1) as in ring1_pruned
    const float ksc = (float)bound_scale * 0.999999f;
    int cx, cy, cz;
    float glx, ghx, gly, ghy, glz, ghz;
    cell_and_gaps(Qx, gx, eps, cx, glx, ghx);
    cell_and_gaps(Qy, gy, eps, cy, gly, ghy);
    cell_and_gaps(Qz, gz, eps, cz, glz, ghz);
    uint32_t gxl_k = cvt_u32_sat(glx * glx * ksc), gxh_k = cvt_u32_sat(ghx * ghx * ksc);
    const uint32_t gyl_k = cvt_u32_sat(gly * gly * ksc), gyh_k = cvt_u32_sat(ghy * ghy * ksc);
    const uint32_t gzl_k = cvt_u32_sat(glz * glz * ksc), gzh_k = cvt_u32_sat(ghz * ghz * ksc);
    // When will the K-th best be final?  Whatever lies outside the 3x3x3 box is beyond one of its faces: a whole cell
    // (less eps) plus the gap to the own cell's face away along that axis; faces on the grid boundary have nothing
    // beyond them (ring1_pruned, same rule).
    const float hs = KEY_CELL - eps;
    float sb = __builtin_huge_valf();
    if (cx > 1) sb = fminf(sb, glx + hs);
    if (cx < gx - 2) sb = fminf(sb, ghx + hs);
    if (cy > 1) sb = fminf(sb, gly + hs);
    if (cy < gy - 2) sb = fminf(sb, ghy + hs);
    if (cz > 1) sb = fminf(sb, glz + hs);
    if (cz < gz - 2) sb = fminf(sb, ghz + hs);
    const double settle_b2 = (double)(sb * sb * ksc) * g.key_invS;
    // what is known about the K-th distance before the walk: the caller's bound, and the reach of interest
    uint32_t bound_k = key_ceil(fmin(bound_d2, max_d2), S);
    const uint32_t UB0 = sat_add(bound_k, delta);
    // ---- phase 1: the (up to) nine rows, straight-line -- every cell-table load of the search is in flight at once.
    // The four boundaries cs[c-1..c+2] of a row's three cells come with ONE 16-byte buffer load (the table is padded by
    // an entry either side, pcr_core.hip; a row outside the grid may read anything or nothing -- it is masked).
    // A row's entry: x = position of the first point of its cell cx-1, y = the populations of cells cx-1, cx, cx+1 (one
    // byte each) and, in the top byte, the row's slab distance in key units >> 22, rounded down (saturating: a smaller
    // slab distance only prunes less).  Non-empty rows are
    // appended to the lane's column of the LDS table in order of slab-distance class (centre, 4 faces, 4 edges).
    const uint32_t cells = (uint32_t)gx * (uint32_t)gy * (uint32_t)gz;
    const __amdgpu_buffer_rsrc_t rs_cs = __builtin_amdgcn_make_buffer_rsrc((void *)(cs - 1), 0, (int)((cells + 3u) * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_sh = __builtin_amdgcn_make_buffer_rsrc((void *)shadow, 0, (int)(g.n * 8u), 0x00020000);
    // the query in integer key space (what the candidates' shadows are subtracted from)
    const uint32_t QX = (uint32_t)key_coord(sx, g.lo[0], g.key_sc);
    const uint32_t Qyz = ((uint32_t)key_coord(sy, g.lo[1], g.key_sc) & 0xFFFFu) | ((uint32_t)key_coord(sz, g.lo[2], g.key_sc) << 16);
    STAGE_CLOCK(clk, 0);
    uint32_t cnt = 0;
    {
        constexpr int DY[9] = {0, -1, 1, 0, 0, -1, 1, -1, 1};
        constexpr int DZ[9] = {0, 0, 0, -1, 1, -1, -1, 1, 1};
        const uint32_t cb4 = (((uint32_t)cz * (uint32_t)gy + (uint32_t)cy) * (uint32_t)gx + (uint32_t)cx) * 4u;
        const int zstep = gx * gy;
        const bool has_xl = cx > 0, has_xh = cx < gx - 1;
        const bool y_lo = cy > 0, y_hi = cy < gy - 1, z_lo = cz > 0, z_hi = cz < gz - 1;
        uint32_t rowk_[9];
        bool open_[9];
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            rowk_[r] = sat_add(DY[r] < 0 ? gyl_k : (DY[r] > 0 ? gyh_k : 0u), DZ[r] < 0 ? gzl_k : (DZ[r] > 0 ? gzh_k : 0u));
            const bool inside = (DY[r] < 0 ? y_lo : (DY[r] > 0 ? y_hi : true)) & (DZ[r] < 0 ? z_lo : (DZ[r] > 0 ? z_hi : true));
            open_[r] = inside & !(rowk_[r] > UB0);
        }
        uint32_t P0[9], P1[9], P2[9], P3[9];  // the four boundaries of every row's three cells
        bool compact = false;
        if constexpr (CT) compact = g.cbits != nullptr;
        if (compact) {
            // cells a = c - has_xl .. c + has_xh of the row: their occupancy bits from the word(s) that hold them, the rank of
            // cell a, then the starts of the rank-th occupied cell and the three behind it (cstart is padded for that) --
            // cell a+i starts at q[number of occupied cells among a .. a+i-1].  A row outside the grid reads zeros or
            // anything (the buffer check): it is masked, as below.
            const uint32_t words = (cells >> 5) + 2u;
            const __amdgpu_buffer_rsrc_t rs_cb = __builtin_amdgcn_make_buffer_rsrc((void *)g.cbits, 0, (int)(words * 8u), 0x00020000);
            const __amdgpu_buffer_rsrc_t rs_st = __builtin_amdgcn_make_buffer_rsrc((void *)g.cstart, 0, (int)0x7FFFFFFC, 0x00020000);
            const uint32_t cb = cb4 >> 2;
            u32x2 wa[9], wb[9];
            uint32_t a_[9];
#pragma unroll
            for (int r = 0; r < 9; ++r) {
                const uint32_t c = cb + (uint32_t)(DY[r] * gx + DZ[r] * zstep);
                a_[r] = c - (has_xl ? 1u : 0u);
                wa[r] = __builtin_amdgcn_raw_buffer_load_b64(rs_cb, (a_[r] >> 5) * 8u, 0, 0);
                wb[r] = __builtin_amdgcn_raw_buffer_load_b64(rs_cb, ((c + 1u) >> 5) * 8u, 0, 0);
            }
            u32x4 q[9];
            uint32_t win_[9];
#pragma unroll
            for (int r = 0; r < 9; ++r) {
                const uint32_t sh = a_[r] & 31u;
                // (the second word matters only when the three cells straddle a word boundary: sh >= 30)
                const unsigned long long bits = (unsigned long long)wa[r].x | ((unsigned long long)wb[r].x << 32);
                uint32_t win = (uint32_t)(bits >> sh) & 7u;
                win = has_xl ? win : (win << 1) & 7u;     // bit 0: cell c-1, bit 1: cell c, bit 2: cell c+1
                win = has_xh ? win : win & 3u;
                win_[r] = win;
                const uint32_t rank = wa[r].y + (uint32_t)__popc(wa[r].x & ((1u << sh) - 1u));
                q[r] = __builtin_amdgcn_raw_buffer_load_b128(rs_st, open_[r] ? rank * 4u : 0u, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 9; ++r) {
                const uint32_t b0 = win_[r] & 1u, b01 = b0 + ((win_[r] >> 1) & 1u), b012 = b01 + (win_[r] >> 2);
                P0[r] = q[r].x;
                P1[r] = b0 ? q[r].y : q[r].x;
                P2[r] = b01 == 0u ? q[r].x : (b01 == 1u ? q[r].y : q[r].z);
                P3[r] = b012 == 0u ? q[r].x : (b012 == 1u ? q[r].y : (b012 == 2u ? q[r].z : q[r].w));
            }
        } else {
            u32x4 b[9];
#ifdef PCR_FAST_SKIP_CLOSED_ROWS  // (measured: 127.4 against 127.7 us on the 1 M x 1 M step -- within the noise; left off)
            // a row that is closed before it is read (outside the grid, or beyond what the caller's bound leaves open) asks for
            // an offset past the end of the table: the buffer check answers zeros without a request to the cache
#pragma unroll
            for (int r = 0; r < 9; ++r)
                b[r] = __builtin_amdgcn_raw_buffer_load_b128(rs_cs, open_[r] ? cb4 + (uint32_t)(4 * (DY[r] * gx + DZ[r] * zstep)) : 0xFFFFFFF0u, 0, 0);
#else
#pragma unroll
            for (int r = 0; r < 9; ++r)
                b[r] = __builtin_amdgcn_raw_buffer_load_b128(rs_cs, cb4 + (uint32_t)(4 * (DY[r] * gx + DZ[r] * zstep)), 0, 0);
#endif
#pragma unroll
            for (int r = 0; r < 9; ++r) {
                // (the first element goes through an opaque copy: the compiler otherwise turns `has_xl ? b.x : b.y` into a
                // run-time vector index and lowers THAT as a chain of three compare-and-selects over all four elements)
                uint32_t bx = b[r].x;
                asm volatile("" : "+v"(bx));
                P0[r] = has_xl ? bx : b[r].y;
                P1[r] = b[r].y;
                P2[r] = b[r].z;
                P3[r] = has_xh ? b[r].w : b[r].z;
            }
        }
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            const uint32_t rowk = rowk_[r];
            const bool open = open_[r];
            const uint32_t p0 = P0[r], p1 = P1[r], p2 = P2[r], p3 = P3[r];
            const uint32_t tot = p3 - p0;
            ok &= !(open & (tot > 255u));  // the payload holds 8 bits of offset: a longer row is the fp64 walk's business
            if (open & (tot != 0u)) {
                rowtab[cnt * RS] = make_uint2(p0, (p1 - p0) | ((p2 - p1) << 8) | ((p3 - p2) << 16) | (min(rowk >> 22, 255u) << 24));
                ++cnt;
            }
        }
    }
    // ---- phase 2: the candidates of all rows as ONE stream, U per trip.  A trip takes what is left of the current
    // row and, when that does not fill it, opens the next row and continues there -- a lane's trips are about
    // (candidates + pruned rows) / U, not one or more per row.  Slot u of a trip reads position (u < rem ? p : p2 - rem)
    // + u: an unused slot may read anything (or, past the end of the array, nothing): its key is the sentinel.
    STAGE_CLOCK(clk, 1);
    TopKeys<KP> T;
    T.init();
    uint32_t p = 0, e = 0, pb = 0, ri = 0;
    bool alive = ok & (cnt != 0u);
    DIAG_LANE_ADD(16 + 4 * XF, 1);                    // filtered searches
    DIAG_LANE_ADD(17 + 4 * XF, ok ? 0 : 1);           // ... that cannot start (far query, long row)
    while (alive) {
        DIAG_ADD(24 + XF, 1);  // wave trips
        DIAG_LANE_ADD(36 + XF, 1);                                   // ... lane trips (the wave's are the slowest lane's)
#ifdef PCR_DIAG
        if (diag_trips) ++*diag_trips;
#endif
        DIAG_LANE_ADD(38 + XF, (e == p) ? 1 : 0);                    // ... of which with nothing left of the current row
        const uint32_t rem = e - p;
        uint32_t p2 = 0, e2 = 0, pb2 = 0;
        // open the next row when the current one cannot fill the trip; a row that turns out to be pruned as a whole
        // costs another try, not a trip, while tries last (TRIES: 1 where the caller brings a bound -- rows that can be
        // pruned then never enter the table; 3 for an unbounded search -- rows are in order of slab distance, so once one
        // is pruned most of the rest are)
        bool want = (rem < (uint32_t)U) & (ri < cnt);
#pragma unroll
        for (int tries = 0; tries < TRIES; ++tries) {
            if (want) {
                const uint2 row = rowtab[ri * RS];
                const uint32_t n0 = row.y & 255u, n1 = (row.y >> 8) & 255u, n2 = (row.y >> 16) & 255u, rk = (row.y >> 24) << 22;
                const uint32_t UB = sat_add(min(T.t[KC - 1], bound_k) | ~KEY_DIST_MASK, delta);
                const bool trim_lo = sat_add(rk, gxl_k) > UB, trim_hi = sat_add(rk, gxh_k) > UB;
                p2 = row.x + (trim_lo ? n0 : 0u);
                e2 = row.x + n0 + n1 + (trim_hi ? 0u : n2);
                e2 = rk > UB ? p2 : e2;  // the whole row is provably farther than the K-th best
                pb2 = row.x - (ri << 8);
                ++ri;
                want = (e2 == p2) & (ri < cnt);
            }
        }
        // Two records (16 bytes) per load: the current row takes an EVEN number of slots -- when an odd number of its
        // candidates is left, the slot behind the last one is a dummy (it reads whatever follows the row and is masked) --
        // so that every pair of slots (2v, 2v+1) reads two neighbouring positions of one row.  Half the memory
        // instructions, and half the requests the L1 has to look up, per candidate.
        static_assert(U % 2 == 0, "slots are loaded in pairs");
        const uint32_t rem2 = e2 - p2;
        const uint32_t remE = (rem + 1u) & ~1u, totE = remE + rem2;
        const uint32_t dummy = (rem & 1u) ? rem : 0xFFFFFFFFu;
        const uint32_t baseB = p2 - remE, d1 = p - pb, d2 = baseB - pb2;
        uint32_t cx[U], cyz[U];  // (scalars on purpose: this compiler turns a small vector filled element by element from
                                 // the builtin's result into copies of its FIRST dword, and loads only that one)
        uint32_t pay[U];
#pragma unroll
        for (int v = 0; v < U / 2; ++v) {
            const bool inA = (uint32_t)(2 * v) < rem;
            const auto rec = __builtin_amdgcn_raw_buffer_load_b128(rs_sh, ((inA ? p : baseB) << 3) + 16u * (uint32_t)v, 0, 0);
            static_assert(sizeof(rec) == 16, "two shadow records");
            cx[2 * v] = rec[0];
            cyz[2 * v] = rec[1];
            cx[2 * v + 1] = rec[2];
            cyz[2 * v + 1] = rec[3];
            pay[2 * v] = (inA ? d1 : d2) + (uint32_t)(2 * v);
            pay[2 * v + 1] = pay[2 * v] + 1u;
        }
        if (rem >= (uint32_t)U) {
            p += (uint32_t)U;
        } else {
            p = p2 + min((uint32_t)U - remE, rem2);
            e = e2;
            pb = pb2;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            uint32_t key = (shadow_key_d2(QX, Qyz, cx[u], cyz[u]) & KEY_DIST_MASK) | pay[u];
            asm volatile("" : "+v"(key));  // computed in every lane, selected below: no branch per slot
            T.insert((((uint32_t)u < totE) & ((uint32_t)u != dummy)) ? key : KEY_SENTINEL);
        }
        DIAG_LANE_ADD(40 + XF, min(totE, (uint32_t)U));  // slots that held a candidate (or a dummy)
        alive = (p < e) | (ri < cnt);
    }
    STAGE_CLOCK(clk, 2);
    // ---- phase 3: fp64 decides.  The candidates behind the KC smallest keys, exactly; the k-th of them by ranking
    // (every pair compared once); then the verification.
    constexpr int NS = SEEDED ? 1 : 0;
    double ed[KC + 1 + NS];
    uint32_t ei[KC + 1 + NS], rk[KC + 1 + NS];
    Pt4 rec2 = {0.0, 0.0, 0.0, 0ull};  // the second stage's record, if one is read
    uint32_t pos2 = 0u;
    auto evaluate = [&](uint32_t key, double &d, uint32_t &ci) {
        d = max_d2;
        ci = PCR_NO_INDEX;
        if (key != KEY_SENTINEL) {
            const uint32_t pos = rowtab[((key >> 8) & 15u) * RS].x + (key & 255u);
            const Pt4 cnd = exact_pts[pos];
            const double dd = cand_d2<XF, RG>(xf, qx, qy, qz, cnd);
            if (dd < max_d2) {
                d = dd;
                ci = (uint32_t)cnd.w;
                rec2 = cnd;
                pos2 = pos;
            }
        }
    };
    // exact squared distance every candidate with a key >= `key` is at least at
    const double floor_sc = g.key_invS * (1.0 - 9.313225746154785e-10);  // (uniform: one product, not one per call)
    // The bound by DISTANCE, not the worst case over the key range: a key is the squared length of an integer vector that
    // differs from the exact one by at most one unit per axis (the point's rounding and the query's: pcr_core.hip), so
    // |key - exact| <= 2 sqrt(3) * 1.01 * sqrt(exact) + 3.1 -- and a candidate whose exact value is BELOW its (truncated)
    // key has sqrt(exact) < sqrt(key).  `delta` is that bound at the largest distance of interest (plus, for the reverse
    // search, what the caller added for R's deviation from an isometry and the inverse map's rounding: kept as it is);
    // at a typical K-th distance of half a cell the bound by distance is a third of it, and the verification below fails
    // a third as often (a wave runs the second stage when ANY lane's fails: 67 % -> 28 % of the waves of a 1 M x 1 M step).
    const uint32_t delta_extra = delta - min(delta, g.key_delta);
    auto floor_d2 = [&](uint32_t key) {
        const uint32_t km = key & KEY_DIST_MASK;
        const uint32_t by_dist = (uint32_t)(sqrtf((float)km) * 3.4995f) + 72u + delta_extra;  // (3.4988 rounded up past fp32's roundings)
        return key == KEY_SENTINEL ? PCR_INF : (double)sat_sub(km, min(by_dist, delta)) * floor_sc;
    };
    if constexpr (SEEDED) {
        ed[0] = seed_d;
        ei[0] = seed_i;
    }
    Pt4 crec[KC];
    uint32_t cpos[KC];
    {
        // all KC gathers in flight together: positions first (LDS), then the 32-byte records (an empty list slot reads
        // record 0 and is masked), then the arithmetic -- not KC dependent round trips one after the other
#pragma unroll
        for (int j = 0; j < KC; ++j) {
            const uint32_t key = T.t[j];
            const uint32_t base = rowtab[((key >> 8) & 15u) * RS].x;
            cpos[j] = key != KEY_SENTINEL ? base + (key & 255u) : 0u;
        }
        if constexpr (PAIR) {
            // A 32-byte record fetched by one lane is two 16-byte loads, and the 64 lanes of each of them ask for 64
            // different cache lines: the most expensive requests this search makes of the L1.  Lane pairs (2i, 2i+1)
            // share the work instead: the first load fetches the EVEN lane's record -- the even lane its lower half,
            // the odd lane the upper half -- the second the odd lane's: the same two instructions, each touching half
            // as many lines with neighbouring lanes on neighbouring addresses; a swap within the pair (DPP quad_perm,
            // no LDS) hands each lane the half its partner fetched for it.  MEASURED: no gain (the 1 M x 1 M step 129.3 / 130.5
            // us with, 128.4 / 128.9 without) -- what these gathers cost the L1 is not a matter of lines per instruction;
            // the step kernel leaves it off (PCR_FAST_PAIR).
            const uint32_t odd = threadIdx.x & 1u;
            const u32x4 *__restrict__ halves = (const u32x4 *)exact_pts;
            u32x4 ha[KC], hb[KC];
#pragma unroll
            for (int j = 0; j < KC; ++j) {
                const uint32_t theirs = quad_swap1(cpos[j]);
                const uint32_t pos_even = odd ? theirs : cpos[j], pos_odd = odd ? cpos[j] : theirs;
                ha[j] = halves[2u * (size_t)pos_even + odd];
                hb[j] = halves[2u * (size_t)pos_odd + odd];
            }
#pragma unroll
            for (int j = 0; j < KC; ++j) {
                u32x4 got;  // what the partner fetched for this lane: the even lane sends hb, the odd lane ha
#pragma unroll
                for (int c = 0; c < 4; ++c) got[c] = quad_swap1(odd ? ha[j][c] : hb[j][c]);
                const u32x4 lo = odd ? got : ha[j], hi = odd ? hb[j] : got;
                crec[j].x = __builtin_bit_cast(double, ((unsigned long long)lo[1] << 32) | lo[0]);
                crec[j].y = __builtin_bit_cast(double, ((unsigned long long)lo[3] << 32) | lo[2]);
                crec[j].z = __builtin_bit_cast(double, ((unsigned long long)hi[1] << 32) | hi[0]);
                crec[j].w = ((unsigned long long)hi[3] << 32) | hi[2];
            }
        } else {
#pragma unroll
            for (int j = 0; j < KC; ++j) crec[j] = exact_pts[cpos[j]];
        }
#pragma unroll
        for (int j = 0; j < KC; ++j) {
            const double dd = cand_d2<XF, RG>(xf, qx, qy, qz, crec[j]);
            const bool in = (T.t[j] != KEY_SENTINEL) & (dd < max_d2);
            ed[NS + j] = in ? dd : max_d2;
            ei[NS + j] = in ? (uint32_t)crec[j].w : PCR_NO_INDEX;
            // A listed candidate farther than 1.5 h along y or z has WRAPPED to a key far below its distance.  These KC
            // keys are what every pruning bound of the walk came from (UB: T.t[KC - 1]): with all of them honest --
            // key >= exact - delta, checked here against the exact distance -- the walk's proof stands as it is; one that
            // is not sends the lane to the fp64 walk.  (Unlisted candidates need no check: a key that is too small only
            // makes the verification below harder to pass.)
            ok &= !((T.t[j] != KEY_SENTINEL) & (key_floor(dd, S) > sat_add(T.t[j] | ~KEY_DIST_MASK, delta)));
        }
    }
    ei[KC + NS] = PCR_NO_INDEX;
    double ek;
    uint32_t ik;
    ranks_of<KC + NS>(ed, ei, rk);
    kth_of<KC + NS>(ed, ei, rk, k, ek, ik);
    int n_exact = KC + NS;
    double lb = floor_d2(T.t[KC]);
    bool verified = (ek < lb) | (max_d2 <= lb);
    DIAG_LANE_ADD(18 + 4 * XF, (ok & !verified) ? 1 : 0);  // first verification failed
    if (__any(ok & !verified)) {
        DIAG_ADD(26 + XF, 1);  // wave-level second stages
        evaluate(verified ? KEY_SENTINEL : T.t[KC], ed[KC + NS], ei[KC + NS]);
        uint32_t rk2[KC + 1 + NS];
        double ek2;
        uint32_t ik2;
        ranks_of<KC + 1 + NS>(ed, ei, rk2);
        kth_of<KC + 1 + NS>(ed, ei, rk2, k, ek2, ik2);
        const double lb2 = floor_d2(T.t[KC + 1]);
        if (!verified) {
            ek = ek2;
            ik = ik2;
            n_exact = KC + 1 + NS;
#pragma unroll
            for (int j = 0; j < KC + 1 + NS; ++j) rk[j] = rk2[j];
            verified = (ek < lb2) | (max_d2 <= lb2);
        }
    }
    if (all) {
#pragma unroll
        for (int j = 0; j < KC + 1 + NS; ++j) {
            all->d[j] = ed[j];
            all->i[j] = ei[j];
            all->rank[j] = rk[j];
        }
        all->n = n_exact;
        all->trusted = ok & verified;
        all->cell[0] = cx;
        all->cell[1] = cy;
        all->cell[2] = cz;
    }
    if (win_pos) {
        uint32_t wp = PCR_NO_INDEX;
        const bool any = ik != PCR_NO_INDEX;
#pragma unroll
        for (int j = 0; j < KC; ++j) wp = (any & (ei[NS + j] == ik)) ? cpos[j] : wp;
        wp = (any & (ei[KC + NS] == ik)) ? pos2 : wp;
        *win_pos = wp;
    }
    if (win) {  // the record behind the winning index, picked by index (every listed candidate is a different point)
        Pt4 w = {0.0, 0.0, 0.0, 0ull};
        const bool any = ik != PCR_NO_INDEX;
#pragma unroll
        for (int j = 0; j < KC; ++j) {
            const bool sel = any & (ei[NS + j] == ik);
            w.x = sel ? crec[j].x : w.x;
            w.y = sel ? crec[j].y : w.y;
            w.z = sel ? crec[j].z : w.z;
            w.w = sel ? crec[j].w : w.w;
        }
        const bool sel2 = any & (ei[KC + NS] == ik);
        w.x = sel2 ? rec2.x : w.x;
        w.y = sel2 ? rec2.y : w.y;
        w.z = sel2 ? rec2.z : w.z;
        w.w = sel2 ? rec2.w : w.w;
        *win = w;
    }
    STAGE_CLOCK(clk, 3);
    const bool settled = (ek < settle_b2) | (max_d2 <= settle_b2);
    DIAG_LANE_ADD(19 + 4 * XF, (ok & verified & settled) ? 0 : 1);  // handed to the fp64 walk
    out_d = ek;
    out_i = ik;
    return ok & verified & settled;
}

// The search over all levels of an index, in three passes.
//   1. The 3x3x3 walk on every level that can hold a better point.  A level the query lies outside of is entered
//      at its cell nearest to the query.  Cheap, and afterwards the K-th distance is finite for all but the
//      strangest queries.
//   2. The levels that walk did not settle are continued with the shell search up to a radius of PCR_LANE_RMAX cells;
//      its reach is bounded by what the OTHER levels found: a query in the hole a finer level left in a coarse one
//      does not wander through the hole looking for K points of that level.
//   3. The few queries that even that leaves open (hovering many cells away from a fine level's data) are finished
//      one at a time by the whole wave (coop_shells).
// Order in every pass: the levels whose box holds the query, finest first (that is where the nearest points are),
// then the others from coarse to fine -- the boxes are nested, so that is by increasing distance -- each skipped
// once its whole box is provably farther than the current K-th best.  One call site per pass (2 x levels slots).
// Exact: every level is searched exactly against the running top-K, and the levels partition the cloud.
// Called by ALL lanes of a wave (pass 3 shuffles); `active`: this lane has a query.  M_mem, xf_mem: the index's description
// and (R,T) in DEVICE memory -- M and *xf may live in kernel arguments and registers, whose address cannot be handed
// to the cooperative pass (not inlined) without the compiler copying all of it to every lane's scratch memory.
// COOP = false: no pass 3 (pass 2 runs the shell search to its end in the lane; any subset of lanes may call).
// CT = false: the caller vouches that no level carries a compact cell table (GridView::cbits) -- the step kernel keeps an
// instance without that path, which costs the 16 M LiDAR iteration 13.0 -> 14.0 ms when merely compiled in.
template <int KC, bool XF = false, bool COOP = true, bool CT = true>
__device__ __forceinline__ void multi_lane_search(const MultiView &M, const MultiView *M_mem, const Pt4 *__restrict__ cand_all, bool active, double qx,
                                                  double qy, double qz, double sx, double sy, double sz, double max_d2,
                                                  double bound_scale, double extra_margin, TopK<KC> &top, uint2 *rowtab,
                                                  const Rigid *xf = nullptr, const Rigid *xf_mem = nullptr) {
    const int L = M.levels;
    const double qmargin = extra_margin + 9.094947017729282e-13 * (fabs(sx) + fabs(sy) + fabs(sz));
    uint32_t holds = 0, open = 0;
    [[maybe_unused]] const unsigned long long diag_p0 = DIAG_T();  // (diagnostic build: the wave's time in each of the three passes, slots 42..47)
    if (active)
        for (int it = 0; it < 2 * L; ++it) {
            const int l = it < L ? L - 1 - it : it - L;
            const GridView &g = M.v[l];
            bool go;
            if (it < L) {
                const double pad = 2.0 * g.h;  // a query a couple of cells outside is as good as inside
                go = (l == 0) | ((sx >= g.lo[0] - pad) & (sx <= g.hi[0] + pad) & (sy >= g.lo[1] - pad) & (sy <= g.hi[1] + pad) &
                                 (sz >= g.lo[2] - pad) & (sz <= g.hi[2] + pad));
                holds |= go ? 1u << l : 0u;
                // The hole: every point of level l inside the box of level l+1 was handed down to it (assign_level_kernel),
                // so level l holds nothing there.  A query deeper inside that box than the running K-th distance -- the
                // finer levels have been searched by now -- has nothing to find in level l: no walk.  (A query in the
                // densest part of a sweep lies in every level's box; without this it walks all of them.)
                if (go && l + 1 < L) {
                    const GridView &f = M.v[l + 1];
                    const double m = fmin(fmin(fmin(sx - f.lo[0], f.hi[0] - sx), fmin(sy - f.lo[1], f.hi[1] - sy)),
                                          fmin(sz - f.lo[2], f.hi[2] - sz)) -
                                     (g.margin + qmargin);
                    const double h2 = m > 0.0 ? m * m * bound_scale : 0.0;
                    go = !((top.worst_d() < h2) | (max_d2 <= h2));
                }
            } else {
                const double b2 = level_box_dist2(g, sx, sy, sz, g.margin + qmargin) * bound_scale;
                go = !((holds >> l) & 1u) & !((top.worst_d() < b2) | (max_d2 <= b2));
            }
            if (go) {
                DIAG_LANE_ADD(8 * (KC == 1) + 5, 1);  // ring walks
                const Pt4 *lvl_pts = cand_all + (g.pts - M.all);
                bool walked = false, settled = false;
                if constexpr (KC <= 4) {
                    // A level that carries fp32 shadows takes the FILTERED walk (fp32 keys select, fp64 decides and verifies):
                    // its reach is what the levels before it left open -- the running K-th distance, ties included -- and its
                    // exact, verified list (ranks 1..KC) is merged into the running one.  Whether the 3x3x3 cells settle
                    // the level is then asked of the fp64 geometry (ring1_settles), as pass 2 does -- provided both walks
                    // are centred on the same cell (this one picks its cell in fp32).  Anything else -- no shadows, a query
                    // far outside the level's box, an unresolved near-tie -- takes the fp64 walk below.
                    if (g.shadow) {
                        FastExact<KC> fx;
                        fx.trusted = false;
                        // (strictly above the running K-th distance, also when that is zero: a tie at that distance
                        // with a lower index must still be seen)
                        const double reach = fmin(max_d2, top.worst_d() * (1.0 + 4.440892098500626e-16) + 4.9406564584124654e-324);
                        const uint32_t delta =
                            g.key_delta + (uint32_t)fmin((1.0 - bound_scale) * 1.1 * KEY_RANGE +
                                                             2.0 * KEY_MAX_ROOT * 1.01 * extra_margin * g.key_sc + 2.0,
                                                         268435456.0);
                        double dk;
                        uint32_t ik;
                        fast_ring1<KC, XF, false, 1, PCR_FAST_U, PCR_ROW_STRIDE, false, CT>(g, lvl_pts, qx, qy, qz, sx, sy, sz, reach, PCR_INF, bound_scale, extra_margin,
                                                  delta, KC, 0.0, 0u, dk, ik, rowtab, xf, &fx);
                        if (fx.trusted && fx.cell[0] == cell_coord(sx, g.lo[0], g.inv_h, g.g[0]) &&
                            fx.cell[1] == cell_coord(sy, g.lo[1], g.inv_h, g.g[1]) &&
                            fx.cell[2] == cell_coord(sz, g.lo[2], g.inv_h, g.g[2])) {
#pragma unroll
                            for (int a = 0; a < KC + 1; ++a)
                                if (a < fx.n && fx.i[a] != PCR_NO_INDEX && better(fx.d[a], fx.i[a], top.worst_d(), top.worst_i()))
                                    top.insert(fx.d[a], fx.i[a]);
                            walked = true;
                            settled = ring1_settles(g, sx, sy, sz, g.margin + qmargin, bound_scale, top.worst_d(), max_d2);
                        }
                    }
                }
                if (!walked)
                    settled = ring1_pruned<KC, XF, PCR_ROW_STRIDE, CT>(g, lvl_pts, qx, qy, qz, sx, sy, sz, max_d2, bound_scale, extra_margin, top,
                                                   rowtab, xf);
                open |= settled ? 0u : 1u << l;
            }
        }
    DIAG_ADD(42 + 3 * XF, DIAG_T() - diag_p0);
    if (__any(open != 0)) {
        [[maybe_unused]] const unsigned long long diag_p1 = DIAG_T();
        uint32_t left = 0;
        for (int it = 0; it < 2 * L; ++it) {
            const int l = it < L ? L - 1 - it : it - L;
            const GridView &g = M.v[l];
            bool go = ((open >> l) & 1u) & (((holds >> l) & 1u) == (it < L ? 1u : 0u));
            if (go) {
                const double b2 = level_box_dist2(g, sx, sy, sz, g.margin + qmargin) * bound_scale;
                go = !((top.worst_d() < b2) | (max_d2 <= b2)) &&
                     !ring1_settles(g, sx, sy, sz, g.margin + qmargin, bound_scale, top.worst_d(), max_d2);
            }
            if (go) {
                // the 5x5x5 cells first, a layer's rows looked up together; what that leaves open goes on shell by shell
                const int r_done = shells_beyond_ring1<KC, true, XF, COOP ? PCR_LANE_RMAX : 0, PCR_ROW_STRIDE, CT>(
                    g, cand_all + (g.pts - M.all), qx, qy, qz, sx, sy, sz, max_d2, bound_scale, extra_margin, top, rowtab, xf);
                left |= (uint32_t)r_done << (4 * l);  // (PCR_LANE_RMAX <= 15, at most 6 levels)
            }
        }
        DIAG_ADD(43 + 3 * XF, DIAG_T() - diag_p1);
        [[maybe_unused]] const unsigned long long diag_p2 = DIAG_T();
        // ---- pass 3: one unfinished query at a time, by the whole wave ----
        unsigned long long pending = COOP ? __ballot(left != 0) : 0ull;
        while (pending) {
            const int src = __ffsll(pending) - 1;
            pending &= pending - 1ull;
            DIAG_ADD(8 * (KC == 1) + 6, 1);  // cooperative searches
            const uint32_t u_left = wave_bcast(left, src), u_holds = wave_bcast(holds, src);
            const double uqx = wave_bcast(qx, src), uqy = wave_bcast(qy, src), uqz = wave_bcast(qz, src);
            const double usx = XF ? wave_bcast(sx, src) : uqx, usy = XF ? wave_bcast(sy, src) : uqy,
                         usz = XF ? wave_bcast(sz, src) : uqz;
            const double uqm = extra_margin + 9.094947017729282e-13 * (fabs(usx) + fabs(usy) + fabs(usz));
            const double umax_d2 = wave_bcast(max_d2, src);  // (a lane's reach may be its own: temporal bounds)
            TopK<KC> shared;
#pragma unroll
            for (int j = 0; j < KC; ++j) {
                shared.d[j] = wave_bcast(top.d[j], src);
                shared.idx[j] = wave_bcast(top.idx[j], src);
            }
            for (int it = 0; it < 2 * L; ++it) {
                const int l = it < L ? L - 1 - it : it - L;
                const int r_done = (int)((u_left >> (4 * l)) & 15u);
                if (!((r_done != 0) & (((u_holds >> l) & 1u) == (it < L ? 1u : 0u)))) continue;
                const GridView &g = M.v[l];
                // other levels may have shrunk the K-th distance since this one was left open
                const double b2 = level_box_dist2(g, usx, usy, usz, g.margin + uqm) * bound_scale;
                if ((shared.worst_d() < b2) | (umax_d2 <= b2)) continue;
                coop_shells<KC, XF, CT>(&M_mem->v[l], cand_all + (g.pts - M.all), uqx, uqy, uqz, usx, usy, usz, umax_d2, bound_scale,
                                    extra_margin, &shared, r_done, xf_mem);
            }
            if ((int)(threadIdx.x & 63u) == src) top = shared;
        }
        DIAG_ADD(44 + 3 * XF, DIAG_T() - diag_p2);
    }
}


}  // namespace pcr
