// libpcr.so -- CRejectionCentroidDist on the device (gfx950).
//
// Replaces CRejectionCentroidDist::_rejectInvalidPairsV and its helpers
// (PointCloudRegistration/RejectionCentroidDist.cpp:23-208; RegMath.h:11-22,42-55,64-67).
// The class is orphan code in the reference (compiled, never registered: SURVEY.md 0-7); its
// behaviour is reproduced INCLUDING its quirks, because "results identical to the reference" is
// the bar:
//   * "centroid" = (x + sum of neighbours) with ONLY component 0 divided, three times, by
//     (count+1); components 1 and 2 stay undivided sums                        (:103-115)
//   * the source-side query uses the TRANSFORMED source point against the index of the
//     UNTRANSFORMED source cloud                                               (:52,:75)
//   * MSD is compared with a SQUARED distance; the neighbour list is cut at the first entry with
//     d2 >= MSD (std::lower_bound on the ascending list)                       (:165-176,200-208)
// Per pair: two exact K-NN searches with K = MSN (one lane per pair, top-K in registers), the
// statistic d_i = |s_tgt - s_src|; then mean, population sigma (two fixed-order tree reductions)
// and the mask |d_i - mean| <= CDF * sigma.
#include <math.h>
#include <vector>
#include "knn_device.h"
#include "fast_search.h"

namespace pcr {

// MULTI: the index has nested levels; the single-level instance contains none of the level logic (and never indexes
// the levels' descriptions at run time, which would send the whole description to every lane's scratch memory)
template <int KC, bool MULTI>
__device__ __forceinline__ double sq_dist_to_centroid(const MultiView &g, double x, double y, double z, int k,
                                                      double msd, uint2 *rowtab) {
    TopK<KC> top;
    top.init();
    {
        const bool fin = isfinite(x) && isfinite(y) && isfinite(z);
        const double bs = 1.0 - 1.8189894035458565e-12;
        if constexpr (MULTI)  // (lists of up to 32 entries: no wave-cooperative pass, whose merge is sized for a handful)
            multi_lane_search<KC, false, false>(g, nullptr, g.all, fin, x, y, z, x, y, z, PCR_INF, bs, 0.0, top, rowtab);
        else if (fin)
            lane_search<KC>(g.v[0], g.all, x, y, z, x, y, z, PCR_INF, bs, 0.0, top, rowtab);
    }
    double c0 = x, c1 = y, c2 = z;
    int cnt = 0;
    bool open = true;  // prefix of the ascending list with d2 < MSD
#pragma unroll
    for (int j = 0; j < KC; ++j) {
        if (j < k) {
            open = open && top.idx[j] != PCR_NO_INDEX && top.d[j] < msd;
            if (open) {
                const Pt4 p = g.all[g.pos_of[top.idx[j]]];
                c0 += p.x;
                c1 += p.y;
                c2 += p.z;
                ++cnt;
            }
        }
    }
    const double den = (double)(cnt + 1);
    c0 /= den;
    c0 /= den;
    c0 /= den;
    return sqdist3(x, y, z, c0, c1, c2);
}

template <int KC, bool MS, bool MT>
__global__ void __launch_bounds__(256) reject_stat_kernel(MultiView src, MultiView tgt, const double *__restrict__ pp,
                                                          const double *__restrict__ pq, const uint32_t *__restrict__ order,
                                                          uint32_t P, int k, double msd, double *__restrict__ d_out) {
    // order (nullable): the pairs grouped by where their target point lies, so that a wave's 64 pairs search the same
    // neighbourhood (pairs arrive in sample order, which says nothing about space: unsorted, every lane of a wave
    // gathers from a different corner of both indices)
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= P) return;
    const uint32_t i = order ? order[t] : t;
    double s[2];
    __shared__ uint2 rowtab[9][256];  // per-lane row table of the ring walk (knn_device.h), nested indices only
    // target side first, as the reference does (:74-75).  Two call sites on purpose: a run-time choice between the
    // two index arguments would make the compiler copy both (1.4 KB) into every lane's scratch memory.
    uint2 *my_rows = &rowtab[0][threadIdx.x];
    s[0] = sq_dist_to_centroid<KC, MT>(tgt, pq[(size_t)i * 3], pq[(size_t)i * 3 + 1], pq[(size_t)i * 3 + 2], k, msd, my_rows);
    s[1] = sq_dist_to_centroid<KC, MS>(src, pp[(size_t)i * 3], pp[(size_t)i * 3 + 1], pp[(size_t)i * 3 + 2], k, msd, my_rows);
    d_out[i] = fabs(s[0] - s[1]);  // (:77)
}

// ---- single-level grids: ONE WAVE per search ---------------------------------------------------------------------
// K = MSN = 25 does not fit a lane: a sorted 32-slot list in registers is 96 registers and a 32-step insert per accepted
// candidate (the kernel above: 292 VGPRs, one wave per SIMD, 16 ms per 620 k pairs).  Here the 64 lanes of a wave search
// for ONE point together: the K best so far live ACROSS the lanes (lane j holds rank j), the candidates of a shell --
// its rows' ranges, laid end to end through a prefix sum -- are read 64 at a time, one per lane (neighbouring lanes read
// neighbouring records), and a candidate that beats the current K-th is inserted with one compare per lane and a shift
// by one lane.  Same shells, ball chords, pruning and stopping rule as grid_search / coop_shells (knn_device.h), same
// total order (d2, original index): the same exact K nearest.  What the statistic needs of them -- the coordinates of
// those below MSD -- is gathered by the lanes that hold them and added nearest first, as the reference adds them.
// Lane exchanges without address arithmetic: ds_swizzle swaps within 32 lanes by a constant XOR mask, a DPP move shifts
// the whole wave by one lane (HIP's __shfl_* compute a byte address per call and go through ds_bpermute: 7 instructions
// per 32-bit word where these take one).
template <int M>
__device__ __forceinline__ uint32_t xor_lane(uint32_t v) {
    if constexpr (M >= 32)
        return (uint32_t)__shfl_xor((int)v, M, 64);
    else
        return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, (M << 10) | 0x1f);
}
template <int M>
__device__ __forceinline__ double xor_lane(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const uint32_t lo = xor_lane<M>((uint32_t)b), hi = xor_lane<M>((uint32_t)(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ uint32_t lane_below(uint32_t v) {  // lane l receives lane l-1's value (lane 0: 0)
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}
// the same under a lane mask: a lane whose lower neighbour is outside the mask keeps its own value
__device__ __forceinline__ uint32_t lane_below_keep(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}
__device__ __forceinline__ double lane_below_keep(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const uint32_t lo = lane_below_keep((uint32_t)b), hi = lane_below_keep((uint32_t)(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double lane_below(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const uint32_t lo = lane_below((uint32_t)b), hi = lane_below((uint32_t)(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// one compare-exchange step of the bitonic network over (d, ci): partner lane ^ JJ, ascending where (lane & KK) == 0
template <int KK, int JJ>
__device__ __forceinline__ void bitonic_step(double &d, uint32_t &ci, int lane) {
    const double od = xor_lane<JJ>(d);
    const uint32_t oi = xor_lane<JJ>(ci);
    const bool keep_min = ((lane & KK) == 0) == ((lane & JJ) == 0);
    const bool other_first = better(od, oi, d, ci);
    const bool swap = keep_min ? other_first : (!other_first & ((od != d) | (oi != ci)));
    d = swap ? od : d;
    ci = swap ? oi : ci;
}
template <int KK, int JJ>
__device__ __forceinline__ void bitonic_stage(double &d, uint32_t &ci, int lane) {
    bitonic_step<KK, JJ>(d, ci, lane);
    if constexpr (JJ > 1) bitonic_stage<KK, JJ / 2>(d, ci, lane);
}
// inclusive prefix sum over the 64 lanes: four DPP row shifts (zero fill) scan each row of 16, two row broadcasts carry
// the rows' totals on -- six data-parallel moves where six __shfl_up are six trips through ds_bpermute with their
// address arithmetic.  Every lane of the wave must be executing.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_zero(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, true);
}
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    v += dpp_zero<0x111, 0xf>(v);  // row_shr:1
    v += dpp_zero<0x112, 0xf>(v);  // row_shr:2
    v += dpp_zero<0x114, 0xf>(v);  // row_shr:4
    v += dpp_zero<0x118, 0xf>(v);  // row_shr:8
    v += dpp_zero<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
    v += dpp_zero<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
    return v;
}
// (PCR_REJ_DIAG: a diagnostic build's counters -- searches, shells, row groups, chunks, candidates, insertions; measured
// on the kept pairs of a converged 1 M x 1 M fit, per search: 2.4 shells, 7.8 row groups, 5.9 chunks, 259 candidates,
// 48 insertions -- the source side asks for the neighbours of a TRANSFORMED point in the UNtransformed cloud (:52,:75),
// 4.5 cells off the surface at that fit's translation, so its searches reach 5-6 cells out)
#ifndef PCR_REJ_MERGE_MIN
#define PCR_REJ_MERGE_MIN 12  // candidates of a group that beat the k-th best: from this many on the group is sorted and merged
#endif
#ifdef PCR_REJ_DIAG
__device__ unsigned long long g_rej_diag[8];
#define RDIAG(i, v) do { if ((threadIdx.x & 63u) == 0u) atomicAdd(&g_rej_diag[i], (unsigned long long)(v)); } while (0)
#else
#define RDIAG(i, v) do { } while (0)
#endif
// (x, y, z): the point, wave-uniform.  lds: 192 words of the wave's own.  Returns the statistic, wave-uniform.
__device__ __forceinline__ double wave_sq_dist_to_centroid(const GridView &g, const uint32_t *__restrict__ pos_of, double x,
                                                           double y, double z, int k, double msd, uint32_t *lds) {
    const int lane = (int)(threadIdx.x & 63u);
    const Pt4 *__restrict__ pts = g.pts;
    const uint32_t *__restrict__ cs = g.cs;
    RDIAG(0, 1);  // searches
    double Ld = PCR_INF;           // lane j: the (j+1)-th best so far
    uint32_t Li = PCR_NO_INDEX;
    double tau_d = PCR_INF;        // the k-th best so far (lane k-1's), wave-uniform
    uint32_t tau_i = PCR_NO_INDEX;
    bool empty = true;             // nothing has been listed yet (wave-uniform)
    if (isfinite(x) && isfinite(y) && isfinite(z)) {
        const int gx = g.g[0], gy = g.g[1], gz = g.g[2];
        const double h = g.h, inv_h = g.inv_h, lox = g.lo[0], loy = g.lo[1], loz = g.lo[2];
        const int cx = cell_coord(x, lox, inv_h, gx), cy = cell_coord(y, loy, inv_h, gy), cz = cell_coord(z, loz, inv_h, gz);
        const double margin = g.margin + 9.094947017729282e-13 * (fabs(x) + fabs(y) + fabs(z));
        const double bound_scale = 1.0 - 1.8189894035458565e-12, inv_bs = 1.0 / bound_scale;
        double ox = fmax(fmax(lox - x, x - g.hi[0]), 0.0) - margin;
        double oy = fmax(fmax(loy - y, y - g.hi[1]), 0.0) - margin;
        double oz = fmax(fmax(loz - z, z - g.hi[2]), 0.0) - margin;
        ox = ox > 0.0 ? ox : 0.0;
        oy = oy > 0.0 ? oy : 0.0;
        oz = oz > 0.0 ? oz : 0.0;
        const double ox2 = ox * ox, oy2 = oy * oy, oz2 = oz * oz;
        int px0 = 1, px1 = 0, py0 = 1, py1 = 0, pz0 = 1, pz1 = 0;  // the box scanned so far (empty)
        const int rmax = max(max(gx, gy), gz);
        int r = 0;
        for (;;) {
            // next radius (grid_search's rule): double, but never past what the ball of everything that can still matter needs
            if (r == 0) {
                r = 1;
            } else {
                int rn = 2 * r;
                if (tau_d < PCR_INF) {
                    const double cells_needed = (sqrt(tau_d * inv_bs) + margin) * inv_h + 2.0;
                    const int need = cells_needed < (double)rmax ? (int)cells_needed : rmax;
                    rn = max(r + 1, min(rn, need));
                }
                r = min(rn, rmax);
            }
            RDIAG(1, 1);  // shells
            const int x0 = max(cx - r, 0), x1 = min(cx + r, gx - 1);
            const int y0 = max(cy - r, 0), y1 = min(cy + r, gy - 1);
            const int z0 = max(cz - r, 0), z1 = min(cz + r, gz - 1);
            const int ny = y1 - y0 + 1;
            // A row of the shell has ONE new x-segment -- all of it -- unless it crosses the box scanned so far, whose rows
            // have two (left and right of the box).  Entries: the shell's rows, then the box's rows once more for their
            // second segment (round 3: two entries for EVERY row -- 14 groups of 64 for shells 1, 2, 4, 8 where this takes 10).
            const int nyi = py1 >= py0 ? py1 - py0 + 1 : 0, nzi = pz1 >= pz0 ? pz1 - pz0 + 1 : 0;
            const uint32_t nrows = (uint32_t)ny * (uint32_t)(z1 - z0 + 1);
            const uint32_t nent = nrows + (uint32_t)nyi * (uint32_t)nzi;
            const float inv_ny = 1.0f / (float)ny, inv_nyi = 1.0f / (float)(nyi > 0 ? nyi : 1);
            for (uint32_t ebase = 0; ebase < nent; ebase += 64u) {
                // ---- every lane: one (row, segment) of the shell -> its range of positions
                RDIAG(2, 1);  // entry groups
                const uint32_t ent = ebase + (uint32_t)lane;
                uint32_t p = 0, count = 0;
                if (ent < nent) {
                    const uint32_t seg = ent >= nrows ? 1u : 0u;
                    uint32_t row = seg ? ent - nrows : ent;
                    // the first cube's rows nearest first (the point's own row, the four beside it, the four diagonal ones):
                    // the first 64 candidates -- the ones that are sorted into the list -- are then the likeliest, and
                    // few of the later ones beat the k-th of them
                    if (r == 1 && ny == 3 && z1 - z0 == 2) row = (uint32_t)((0x862071534ull >> (4u * row)) & 15ull);
                    // row = qz * w + qy, w the width of whichever box the entry runs over: quotient by a float reciprocal and
                    // one correction either way (a 32-bit integer division is two dozen instructions)
                    const uint32_t wdt = seg ? (uint32_t)nyi : (uint32_t)ny;
                    uint32_t qz_i = (uint32_t)(((float)row + 0.5f) * (seg ? inv_nyi : inv_ny));
                    int qy_i = (int)row - (int)(qz_i * wdt);
                    if (qy_i < 0) {
                        --qz_i;
                        qy_i += (int)wdt;
                    } else if (qy_i >= (int)wdt) {
                        ++qz_i;
                        qy_i -= (int)wdt;
                    }
                    const int rz = (seg ? pz0 : z0) + (int)qz_i, ry = (seg ? py0 : y0) + qy_i;
                    const double dz = slab_gap(z, loz, rz, h, margin), dy = slab_gap(y, loy, ry, h, margin);
                    const double rem = tau_d * inv_bs - dz * dz - dy * dy;  // (+inf while fewer than k are known)
                    const bool inner = (ry >= py0) & (ry <= py1) & (rz >= pz0) & (rz <= pz1);
                    if (rem >= 0.0) {
                        int xs = seg == 0u ? x0 : px1 + 1, xe = seg == 0u ? (inner ? px0 - 1 : x1) : x1;
                        if (xs <= xe) {
                            double lo_c = (double)xs, hi_c = (double)xe;
                            if (rem < PCR_INF) {  // the chord the ball cuts out of the row, a cell of slack either side
                                const double w = ball_half_width(rem) + margin;
                                const double lo_f = (x - w - lox) * inv_h - 1.0, hi_f = (x + w - lox) * inv_h + 1.0;
                                lo_c = fmin(fmax(lo_f, (double)xs), (double)xe + 1.0);
                                hi_c = fmax(fmin(hi_f, (double)xe), (double)xs - 1.0);
                            }
                            xs = (int)lo_c;
                            xe = (int)hi_c;
                            if (xs <= xe) {
                                const size_t base = ((size_t)rz * gy + ry) * gx;
                                p = cs[base + xs];
                                count = cs[base + xe + 1] - p;
                            }
                        }
                    }
                }
                const uint32_t incl = wave_incl_scan(count);
                const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                if (total == 0u) continue;
                __builtin_amdgcn_wave_barrier();
                lds[lane] = incl - count;  // where the lane's range starts in the shell's candidate stream
                lds[64 + lane] = p;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                for (uint32_t cbase = 0; cbase < total; cbase += 64u) {
                    // ---- every lane: one candidate of the stream
                    RDIAG(3, 1);  // chunks
                    RDIAG(4, min(64u, total - cbase));  // candidates
                    const uint32_t n = cbase + (uint32_t)lane;
                    const bool have = n < total;
                    uint32_t j = 0;  // the last range that starts at or before n (empty ranges share their successor's start)
#pragma unroll
                    for (uint32_t step = 32u; step > 0u; step >>= 1)
                        if (lds[j + step] <= n) j += step;
                    double d = PCR_INF;
                    uint32_t ci = PCR_NO_INDEX;
                    if (have) {
                        const Pt4 c = pts[lds[64 + j] + (n - lds[j])];
                        d = sqdist3(x, y, z, c.x, c.y, c.z);
                        ci = (uint32_t)c.w;
                    }
                    // ---- the first 64 candidates (for most points: everything the 3x3x3 cells hold) are SORTED across the
                    // lanes -- a bitonic network, 21 compare-exchange steps -- and the first k of them are the list.  A later
                    // group of 64 in which MANY beat the k-th best so far (the group after the first, typically: the k-th of
                    // 64 is still loose) goes the same way instead of entering one at a time: sorted (what cannot enter as
                    // +inf), reversed, and merged with the list -- min(list[i], sorted[63 - i]) holds the 64 smallest of both as a
                    // bitonic sequence, six more steps sort it.  Same total order (d2, index): the same list.
                    bool pend = have & better(d, ci, tau_d, tau_i);
                    if (empty || __popcll(__ballot(pend)) >= PCR_REJ_MERGE_MIN) {
                        if (!pend) {
                            d = PCR_INF;
                            ci = PCR_NO_INDEX;
                        }
                        bitonic_stage<2, 1>(d, ci, lane);
                        bitonic_stage<4, 2>(d, ci, lane);
                        bitonic_stage<8, 4>(d, ci, lane);
                        bitonic_stage<16, 8>(d, ci, lane);
                        bitonic_stage<32, 16>(d, ci, lane);
                        bitonic_stage<64, 32>(d, ci, lane);
                        if (!empty) {
                            const double rd = xor_lane<31>(xor_lane<32>(d));
                            const uint32_t ri = xor_lane<31>(xor_lane<32>(ci));
                            const bool take = better(rd, ri, Ld, Li);  // (lanes from k on hold +inf: they take the group's)
                            d = take ? rd : Ld;
                            ci = take ? ri : Li;
                            bitonic_stage<64, 32>(d, ci, lane);
                            RDIAG(6, 1);  // merged groups
                        }
                        empty = false;
                        Ld = lane < k ? d : PCR_INF;
                        Li = lane < k ? ci : PCR_NO_INDEX;
                        tau_d = wave_bcast(Ld, k - 1);
                        tau_i = wave_bcast(Li, k - 1);
                        continue;
                    }
                    // ---- later ones that beat the k-th best enter the list, one at a time
                    for (;;) {
                        const unsigned long long m = __ballot(pend);
                        if (!m) break;
                        RDIAG(5, 1);  // insertions
                        const int src = __ffsll((long long)m) - 1;
                        const double nd = wave_bcast(d, src);
                        const uint32_t ni = wave_bcast(ci, src);
                        const bool lt = (lane < k) & better(nd, ni, Ld, Li);  // true from the new one's rank upwards
                        const int first = __ffsll((long long)__ballot(lt)) - 1;  // the new one's rank (it beats the k-th: >= 0)
                        // the lanes from that rank on take their lower neighbour's entry (a DPP shift under the mask: the
                        // lane at the rank itself reads a lane outside the mask and is overwritten next), the lane at the
                        // rank takes the new one
                        if (lt) {
                            Ld = lane_below_keep(Ld);
                            Li = lane_below_keep(Li);
                        }
                        if (lane == first) {
                            Ld = nd;
                            Li = ni;
                        }
                        tau_d = wave_bcast(Ld, k - 1);
                        tau_i = wave_bcast(Li, k - 1);
                        pend = pend & (lane != src) & better(d, ci, tau_d, tau_i);
                    }
                }
                __builtin_amdgcn_wave_barrier();  // (the ranges are overwritten by the next group of rows)
            }
            // ---- can anything outside the scanned box still matter? (grid_search's rule)
            const bool all = (x0 == 0) & (x1 == gx - 1) & (y0 == 0) & (y1 == gy - 1) & (z0 == 0) & (z1 == gz - 1);
            if (all) break;
            double b2 = box_bound2(g, x, y, z, margin, ox2, oy2, oz2, x0, x1, y0, y1, z0, z1);
            b2 *= bound_scale;
            if (tau_d < b2) break;
            px0 = x0;
            px1 = x1;
            py0 = y0;
            py1 = y1;
            pz0 = z0;
            pz1 = z1;
            if (r >= rmax) break;
        }
    }
    // ---- the statistic (:96-118): the neighbours below MSD, a prefix of the ascending list, added NEAREST FIRST as the
    // reference adds them (the sums are not divided -- :113-115 -- so they are large, and what is returned is a difference
    // of two such statistics: another order of the same terms shows at 1e-9 relative on clouds in metres).  The lanes
    // that hold the members park their points in the wave's LDS by rank; lanes 0..2 add a coordinate each, in order.
    const bool member = (lane < k) & (Li != PCR_NO_INDEX) & (Ld < msd);
    const int cnt = __popcll(__ballot(member));
    double *ldsd = (double *)lds;
    __builtin_amdgcn_wave_barrier();
    if (member) {
        const Pt4 q = pts[pos_of[Li]];
        ldsd[lane * 3] = q.x;
        ldsd[lane * 3 + 1] = q.y;
        ldsd[lane * 3 + 2] = q.z;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double acc = lane == 0 ? x : (lane == 1 ? y : z);
    if (lane < 3)
        for (int j = 0; j < cnt; ++j) acc += ldsd[j * 3 + lane];
    __builtin_amdgcn_wave_barrier();
    double c0 = wave_bcast(acc, 0);
    const double c1 = wave_bcast(acc, 1), c2 = wave_bcast(acc, 2);
    const double den = (double)(cnt + 1);
    c0 /= den;
    c0 /= den;
    c0 /= den;
    return sqdist3(x, y, z, c0, c1, c2);
}

// four waves per block, a pair per wave: target side first, as the reference does (:74-75)
__global__ void __launch_bounds__(256) reject_stat_wave_kernel(MultiView src, MultiView tgt, const double *__restrict__ pp,
                                                               const double *__restrict__ pq, const uint32_t *__restrict__ order,
                                                               uint32_t P, int k, double msd, double *__restrict__ d_out) {
    __shared__ __align__(8) uint32_t lds_all[4][192];  // per wave: 128 words of ranges, later 32 x 3 doubles of neighbours
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t t = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4u + wave));
    if (t >= P) return;
    const uint32_t i = order ? (uint32_t)__builtin_amdgcn_readfirstlane((int)order[t]) : t;
    uint32_t *lds = lds_all[wave];
    const double s_tgt = wave_sq_dist_to_centroid(tgt.v[0], tgt.pos_of, pq[(size_t)i * 3], pq[(size_t)i * 3 + 1], pq[(size_t)i * 3 + 2],
                                                  k, msd, lds);
    const double s_src = wave_sq_dist_to_centroid(src.v[0], src.pos_of, pp[(size_t)i * 3], pp[(size_t)i * 3 + 1], pp[(size_t)i * 3 + 2],
                                                  k, msd, lds);
    if ((threadIdx.x & 63u) == 0u) d_out[i] = fabs(s_tgt - s_src);  // (:77)
}

// brute-force flavour for BRUTE indices: one lane per pair, plain global scan (small clouds only)
template <int KC>
__device__ __forceinline__ double sq_dist_to_centroid_brute(const MultiView &g, double x, double y, double z, int k,
                                                            double msd) {
    TopK<KC> top;
    top.init();
    double wd = PCR_INF;
    uint32_t wi = PCR_NO_INDEX;
    for (uint32_t j = 0; j < g.n; ++j) {
        const Pt4 c = g.all[j];
        const double d = sqdist3(x, y, z, c.x, c.y, c.z);
        if (better(d, (uint32_t)c.w, wd, wi)) {
            top.insert(d, (uint32_t)c.w);
            wd = top.worst_d();
            wi = top.worst_i();
        }
    }
    double c0 = x, c1 = y, c2 = z;
    int cnt = 0;
    bool open = true;
#pragma unroll
    for (int j = 0; j < KC; ++j) {
        if (j < k) {
            open = open && top.idx[j] != PCR_NO_INDEX && top.d[j] < msd;
            if (open) {
                const Pt4 p = g.all[g.pos_of[top.idx[j]]];
                c0 += p.x;
                c1 += p.y;
                c2 += p.z;
                ++cnt;
            }
        }
    }
    const double den = (double)(cnt + 1);
    c0 /= den;
    c0 /= den;
    c0 /= den;
    return sqdist3(x, y, z, c0, c1, c2);
}

template <int KC>
__global__ void __launch_bounds__(256) reject_stat_brute_kernel(MultiView src, MultiView tgt,
                                                                const double *__restrict__ pp,
                                                                const double *__restrict__ pq, uint32_t P, int k,
                                                                double msd, double *__restrict__ d_out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= P) return;
    const double st = sq_dist_to_centroid_brute<KC>(tgt, pq[(size_t)i * 3], pq[(size_t)i * 3 + 1], pq[(size_t)i * 3 + 2], k, msd);
    const double ss = sq_dist_to_centroid_brute<KC>(src, pp[(size_t)i * 3], pp[(size_t)i * 3 + 1], pp[(size_t)i * 3 + 2], k, msd);
    d_out[i] = fabs(st - ss);
}

// out[0] = sum_i f(d_i), f = identity (center == NULL) or (d - *center)^2; out[1] = count_slot ? P : untouched;
// out[2] = failed (1.0: this rank could not do its share -- it contributes no pair, and the flag, summed with the rest,
// tells EVERY rank so); out[3..16] = 0.  One block, fixed order.  `out` is a 17-double exchange buffer: in a sharded
// run every rank's is summed (the all-reduce the fit loop uses for its 17 partials) before finish_stat_kernel divides.
__global__ void __launch_bounds__(1024) sum_kernel(const double *__restrict__ d, uint32_t P,
                                                   const double *__restrict__ center, int count_slot, int failed,
                                                   double *__restrict__ out) {
    __shared__ double sm[1024];
    const double c = center ? *center : 0.0;
    double s = 0.0;
    for (uint32_t i = threadIdx.x; i < P; i += 1024) {
        const double v = d[i];
        s += center ? (v - c) * (v - c) : v;
    }
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sm[threadIdx.x] += sm[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sm[0];
    if (threadIdx.x == 1 && count_slot) out[1] = (double)P;
    if (threadIdx.x == 2) out[2] = failed ? 1.0 : 0.0;
    if (threadIdx.x >= 3 && threadIdx.x < 17) out[threadIdx.x] = 0.0;
}

// stat[which] = x[0] / count: the mean (:34-35), then the population variance (:36-37), over ALL ranks' pairs.
// total_out (nullable) keeps the pair count for the second call, whose exchange buffer no longer carries it;
// failed_io accumulates the ranks' failure flags of both exchanges.
__global__ void finish_stat_kernel(const double *__restrict__ x, const double *__restrict__ count, double *__restrict__ stat,
                                   int which, double *__restrict__ total_out, double *__restrict__ failed_io) {
    const double n = *count;
    stat[which] = x[0] / n;  // (no pair on any rank: NaN here, PCR_E_INVALID on the host)
    if (total_out) *total_out = n;
    *failed_io = (which == 0 ? 0.0 : *failed_io) + x[2];
}

__global__ void __launch_bounds__(256) reject_mask_kernel(const double *__restrict__ d, uint32_t P,
                                                          const double *__restrict__ stat /*mean, variance*/,
                                                          double cdf, uint8_t *__restrict__ keep) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= P) return;
    const double tol = cdf * sqrt(stat[1]);            // Tolerance = CDF * sigma (:134)
    keep[i] = fabs(d[i] - stat[0]) <= tol ? 1 : 0;     // (:137)
}

}  // namespace pcr

using namespace pcr;

static int reject_impl(pcr_ctx *ctx, const pcr_index *src, const pcr_index *tgt, const double *pair_p,
                       const double *pair_q, uint32_t P, uint32_t msn, double msd, double cdf, uint8_t *out_keep,
                       double *out_d, double *out_stat, pcr_reduce17_fn reduce, void *reduce_user) {
    PCR_REQUIRE(ctx && src && tgt && (out_keep || P == 0), "pcr_reject_centroid_dist: NULL argument");
    PCR_REQUIRE(src->ctx->device == ctx->device && tgt->ctx->device == ctx->device, "pcr_reject_centroid_dist: index lives on another device");
    PCR_REQUIRE(src->kind == tgt->kind, "pcr_reject_centroid_dist: indices must be of the same kind");
    // (a rank of a sharded run may hold no pair at all: it still takes part in both reductions)
    PCR_REQUIRE((P > 0 && pair_p && pair_q) || (P == 0 && reduce), "pcr_reject_centroid_dist: empty pair set (reference asserts, :27)");
    PCR_REQUIRE(msn >= 1 && msn <= src->n && msn <= tgt->n, "pcr_reject_centroid_dist: MSN=%u out of range", msn);
    if (msn > PCR_MAX_K) {
        set_last_error("pcr_reject_centroid_dist: MSN=%u above PCR_MAX_K=%u", msn, PCR_MAX_K);
        return PCR_E_UNSUPPORTED;
    }
    PCR_HIP(hipSetDevice(ctx->device));
    double *pp_d = nullptr, *pq_d = nullptr, *d_d = nullptr;
    uint8_t *k_d = nullptr;
    const size_t bytes = sizeof(double) * 3 * (size_t)(P ? P : 1);
    hipError_t e = hipMalloc(&pp_d, bytes);
    if (e == hipSuccess) e = hipMalloc(&pq_d, bytes);
    if (e == hipSuccess) e = hipMalloc(&d_d, sizeof(double) * (P ? P : 1));
    if (e == hipSuccess) e = hipMalloc(&k_d, P ? P : 1);
    if (e == hipSuccess && P) e = hipMemcpyAsync(pp_d, pair_p, bytes, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && P) e = hipMemcpyAsync(pq_d, pair_q, bytes, hipMemcpyHostToDevice, ctx->stream);
    // A sharded run promises every rank that the hook is entered EXACTLY TWICE on every rank (the others wait in the
    // collective otherwise).  A rank that cannot do its share -- an allocation or a copy failed -- therefore still goes
    // through both exchanges: it contributes no pair and a failure flag (slot 2), which, summed over the ranks, makes
    // EVERY rank return the same error.  Without a hook a local failure simply returns.
    const bool local_fail = e != hipSuccess;
    const hipError_t local_err = e;
    if (local_fail && !reduce) {
        if (pp_d) hipFree(pp_d);
        if (pq_d) hipFree(pq_d);
        if (d_d) hipFree(d_d);
        if (k_d) hipFree(k_d);
        PCR_HIP(local_err);
    }
    const uint32_t Pw = local_fail ? 0u : P;  // pairs this rank works on
    int reduce_rc = PCR_OK;
    {
        const uint32_t nb = (Pw + 255) / 256;
        const uint32_t *qorder = nullptr;
        if (src->kind == PCR_INDEX_GRID && Pw >= 2048) {
            uint32_t *o = (uint32_t *)ctx_scratch(ctx, 4, sizeof(uint32_t) * (size_t)Pw);
            if (o && coarse_query_order(ctx, tgt->view.v[0], pq_d, Pw, o) == PCR_OK) qorder = o;
        }
#define LAUNCH_STAT(KC)                                                                                              \
    do {                                                                                                             \
        if (src->kind == PCR_INDEX_GRID && src->view.levels > 1 && tgt->view.levels > 1)                            \
            hipLaunchKernelGGL((reject_stat_kernel<KC, true, true>), dim3(nb), dim3(256), 0, ctx->stream, src->view, \
                               tgt->view, (const double *)pp_d, (const double *)pq_d, qorder, Pw, (int)msn, msd, d_d); \
        else if (src->kind == PCR_INDEX_GRID && src->view.levels > 1)                                                \
            hipLaunchKernelGGL((reject_stat_kernel<KC, true, false>), dim3(nb), dim3(256), 0, ctx->stream, src->view, \
                               tgt->view, (const double *)pp_d, (const double *)pq_d, qorder, Pw, (int)msn, msd, d_d); \
        else if (src->kind == PCR_INDEX_GRID && tgt->view.levels > 1)                                                \
            hipLaunchKernelGGL((reject_stat_kernel<KC, false, true>), dim3(nb), dim3(256), 0, ctx->stream, src->view, \
                               tgt->view, (const double *)pp_d, (const double *)pq_d, qorder, Pw, (int)msn, msd, d_d); \
        else if (src->kind == PCR_INDEX_GRID)                                                                        \
            hipLaunchKernelGGL((reject_stat_kernel<KC, false, false>), dim3(nb), dim3(256), 0, ctx->stream, src->view, \
                               tgt->view, (const double *)pp_d, (const double *)pq_d, qorder, Pw, (int)msn, msd, d_d); \
        else                                                                                                         \
            hipLaunchKernelGGL(reject_stat_brute_kernel<KC>, dim3(nb), dim3(256), 0, ctx->stream, src->view,         \
                               tgt->view, (const double *)pp_d, (const double *)pq_d, Pw, (int)msn, msd, d_d);      \
    } while (0)
        const bool wave_kernel = src->kind == PCR_INDEX_GRID && src->view.levels == 1 && tgt->view.levels == 1 && msn > 8 &&
                                 g_tune[7] != 1;
        if (Pw && wave_kernel) {
            ProfScope ps(ctx, PCR_PROF_KNN);
            hipLaunchKernelGGL(reject_stat_wave_kernel, dim3((Pw + 3u) / 4u), dim3(256), 0, ctx->stream, src->view, tgt->view,
                               (const double *)pp_d, (const double *)pq_d, qorder, Pw, (int)msn, msd, d_d);
        } else if (Pw) {
            ProfScope ps(ctx, PCR_PROF_KNN);  // (timed with the K-NN class: two K = MSN searches per pair)
            if (msn <= 4)
                LAUNCH_STAT(4);
            else if (msn <= 8)
                LAUNCH_STAT(8);
            else if (msn <= 16)
                LAUNCH_STAT(16);
            else
                LAUNCH_STAT(32);
        }
#undef LAUNCH_STAT
        double *stat_d = ctx->result_d + 20;  // [mean, variance, total pair count, failed ranks]
        double *xbuf_d = ctx->result_d;       // [17] exchange buffer: {sum, count, failed, 0...}
        hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const double *)d_d, Pw,
                           (const double *)nullptr, 1, local_fail ? 1 : 0, xbuf_d);
        if (reduce) reduce_rc = reduce(reduce_user, xbuf_d);  // sum over ranks of {sum d, P, failed} (enqueued on the stream)
        hipLaunchKernelGGL(finish_stat_kernel, dim3(1), dim3(1), 0, ctx->stream, (const double *)xbuf_d,
                           (const double *)(xbuf_d + 1), stat_d, 0, stat_d + 2, stat_d + 3);
        hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const double *)d_d, Pw,
                           (const double *)stat_d, 0, (local_fail || reduce_rc != PCR_OK) ? 1 : 0, xbuf_d);
        if (reduce) {  // sum over ranks of sum (d - mean)^2 -- entered even if the first exchange failed HERE: the other
                       // ranks are on their way into it
            const int rc2 = reduce(reduce_user, xbuf_d);
            if (reduce_rc == PCR_OK) reduce_rc = rc2;
        }
        hipLaunchKernelGGL(finish_stat_kernel, dim3(1), dim3(1), 0, ctx->stream, (const double *)xbuf_d,
                           (const double *)(stat_d + 2), stat_d, 1, (double *)nullptr, stat_d + 3);
        if (Pw)
            hipLaunchKernelGGL(reject_mask_kernel, dim3(nb), dim3(256), 0, ctx->stream, (const double *)d_d, Pw,
                               (const double *)stat_d, cdf, k_d);
        e = hipGetLastError();
        if (e == hipSuccess && Pw) e = hipMemcpyAsync(out_keep, k_d, Pw, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess && out_d && Pw) e = hipMemcpyAsync(out_d, d_d, sizeof(double) * Pw, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess)
            e = hipMemcpyAsync(ctx->result_h + 20, stat_d, sizeof(double) * 4, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e == hipSuccess && out_stat) {
            out_stat[0] = ctx->result_h[20];
            out_stat[1] = sqrt(ctx->result_h[21]);
        }
    }
    if (pp_d) hipFree(pp_d);
    if (pq_d) hipFree(pq_d);
    if (d_d) hipFree(d_d);
    if (k_d) hipFree(k_d);
    if (local_fail) PCR_HIP(local_err);
    PCR_HIP(e);
    if (reduce_rc != PCR_OK) {
        set_last_error("pcr_reject_centroid_dist_sharded: the reduction hook failed (%d)", reduce_rc);
        return reduce_rc < 0 ? reduce_rc : PCR_E_INVALID;
    }
    if (ctx->result_h[23] > 0.0) {
        set_last_error("pcr_reject_centroid_dist_sharded: %g rank(s) could not do their share (allocation, copy or exchange "
                       "failed there); every rank returns this error", ctx->result_h[23]);
        return PCR_E_HIP;
    }
    if (!(ctx->result_h[22] >= 1.0)) {
        set_last_error("pcr_reject_centroid_dist: no pair on any rank (the reference asserts, RejectionCentroidDist.cpp:27)");
        return PCR_E_INVALID;
    }
    return PCR_OK;
}

extern "C" int pcr_reject_centroid_dist(pcr_ctx *ctx, const pcr_index *src, const pcr_index *tgt, const double *pair_p,
                                        const double *pair_q, uint32_t P, uint32_t msn, double msd, double cdf,
                                        uint8_t *out_keep, double *out_d, double *out_stat) {
    return reject_impl(ctx, src, tgt, pair_p, pair_q, P, msn, msd, cdf, out_keep, out_d, out_stat, nullptr, nullptr);
}

extern "C" int pcr_reject_centroid_dist_sharded(pcr_ctx *ctx, const pcr_index *src, const pcr_index *tgt,
                                                const double *pair_p, const double *pair_q, uint32_t P, uint32_t msn,
                                                double msd, double cdf, uint8_t *out_keep, double *out_d,
                                                double *out_stat, pcr_reduce17_fn reduce, void *reduce_user) {
    PCR_REQUIRE(reduce, "pcr_reject_centroid_dist_sharded: NULL reduction hook");
    return reject_impl(ctx, src, tgt, pair_p, pair_q, P, msn, msd, cdf, out_keep, out_d, out_stat, reduce, reduce_user);
}

#ifdef PCR_REJ_DIAG
extern "C" int pcr_diag_read_reject(unsigned long long out[8], int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(pcr::g_rej_diag), sizeof(unsigned long long) * 8) != hipSuccess) return -2;
    if (reset) {
        unsigned long long z[8] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(pcr::g_rej_diag), z, sizeof z) != hipSuccess) return -2;
    }
    return 0;
}
#endif
