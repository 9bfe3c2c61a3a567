// libpcr.so -- context, clouds, on-device uniform-grid index build (gfx950).
//
// Index build = what hiveCommon::CKNNSearch::__buildSearchIndex (Common/KNNSearch.cpp:79-92)
// does for the KD-tree, re-designed for the GPU: bounds reduction -> cell histogram ->
// exclusive scan -> scatter into cell order (a counting sort keyed by the linear cell id,
// x fastest, so the three x-adjacent cells of a row are ONE contiguous candidate range).
#include <math.h>
#include <stdarg.h>
#include <string.h>
#include <vector>
#include "pcr_internal.h"

namespace pcr {

static thread_local char g_last_error[512] = "";
int g_tune[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // pcr_tune keys, see include/pcr.h (2: brick edge, 3: query order, 4: reverse-search copy)

void set_last_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof g_last_error, fmt, ap);
    va_end(ap);
}

// ------------------------------------------------------------------ bounds
// lvl (nullable): level of every point; only points of level `want` take part (NULL: all of them)
__global__ void __launch_bounds__(256) bounds_kernel(const double *__restrict__ xyz, uint32_t n,
                                                     const uint8_t *__restrict__ lvl, uint8_t want,
                                                     double *__restrict__ out /*[gridDim.x][6]*/) {
    double lo[3] = {PCR_INF_H, PCR_INF_H, PCR_INF_H}, hi[3] = {-PCR_INF_H, -PCR_INF_H, -PCR_INF_H};
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (lvl && lvl[i] != want) continue;
        for (int a = 0; a < 3; ++a) {
            const double v = xyz[(size_t)i * 3 + a];
            lo[a] = fmin(lo[a], v);
            hi[a] = fmax(hi[a], v);
        }
    }
    __shared__ double sm[6][256];
    for (int a = 0; a < 3; ++a) {
        sm[a][threadIdx.x] = lo[a];
        sm[3 + a][threadIdx.x] = hi[a];
    }
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s)
            for (int a = 0; a < 3; ++a) {
                sm[a][threadIdx.x] = fmin(sm[a][threadIdx.x], sm[a][threadIdx.x + s]);
                sm[3 + a][threadIdx.x] = fmax(sm[3 + a][threadIdx.x], sm[3 + a][threadIdx.x + s]);
            }
        __syncthreads();
    }
    if (threadIdx.x < 6) out[(size_t)blockIdx.x * 6 + threadIdx.x] = sm[threadIdx.x][0];
}

// ------------------------------------------------------------------ grid build kernels
struct GridGeom {
    double lo[3];
    double inv_h;
    int g[3];
};

__device__ __forceinline__ uint32_t cell_of(const GridGeom &G, double x, double y, double z) {
    double tx = (x - G.lo[0]) * G.inv_h, ty = (y - G.lo[1]) * G.inv_h, tz = (z - G.lo[2]) * G.inv_h;
    tx = tx < 0.0 ? 0.0 : tx;
    ty = ty < 0.0 ? 0.0 : ty;
    tz = tz < 0.0 ? 0.0 : tz;
    const double mx = (double)(G.g[0] - 1), my = (double)(G.g[1] - 1), mz = (double)(G.g[2] - 1);
    tx = tx > mx ? mx : tx;
    ty = ty > my ? my : ty;
    tz = tz > mz ? mz : tz;
    return (uint32_t)(((size_t)(int)tz * G.g[1] + (int)ty) * G.g[0] + (int)tx);
}

__global__ void __launch_bounds__(256) cell_histogram_kernel(const double *__restrict__ xyz, uint32_t n, GridGeom G,
                                                             const uint8_t *__restrict__ lvl, uint8_t want,
                                                             uint32_t *__restrict__ counts) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (lvl && lvl[i] != want) continue;
        const uint32_t c = cell_of(G, xyz[(size_t)i * 3], xyz[(size_t)i * 3 + 1], xyz[(size_t)i * 3 + 2]);
        atomicAdd(&counts[c], 1u);
    }
}

constexpr int SCAN_TILE = 4096;  // elements per block (256 threads x 16)

// pass 1: per-tile sum, plus occupancy statistics
__global__ void __launch_bounds__(256) scan_tile_sums_kernel(const uint32_t *__restrict__ counts, uint64_t cells,
                                                             uint32_t *__restrict__ tile_sums,
                                                             unsigned long long *__restrict__ stats /*[2]*/) {
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE;
    uint32_t sum = 0, occ = 0, mx = 0;
    for (int j = 0; j < 16; ++j) {
        const uint64_t i = base + (uint64_t)j * 256 + threadIdx.x;
        if (i < cells) {
            const uint32_t c = counts[i];
            sum += c;
            occ += c != 0;
            mx = max(mx, c);
        }
    }
    __shared__ uint32_t s_sum[256], s_occ[256], s_max[256];
    s_sum[threadIdx.x] = sum;
    s_occ[threadIdx.x] = occ;
    s_max[threadIdx.x] = mx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            s_sum[threadIdx.x] += s_sum[threadIdx.x + s];
            s_occ[threadIdx.x] += s_occ[threadIdx.x + s];
            s_max[threadIdx.x] = max(s_max[threadIdx.x], s_max[threadIdx.x + s]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        tile_sums[blockIdx.x] = s_sum[0];
        atomicAdd(&stats[0], (unsigned long long)s_occ[0]);
        atomicMax(&stats[1], (unsigned long long)s_max[0]);
    }
}

// pass 2: exclusive scan of the tile sums by ONE block (tiles <= 2^30/4096 = 262144)
__global__ void __launch_bounds__(1024) scan_tile_offsets_kernel(uint32_t *__restrict__ tile_sums, uint32_t tiles) {
    __shared__ uint32_t sm[1024];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < tiles; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < tiles ? tile_sums[i] : 0;
        sm[threadIdx.x] = v;
        __syncthreads();
        for (int s = 1; s < 1024; s <<= 1) {  // Hillis-Steele inclusive scan
            const uint32_t add = (int)threadIdx.x >= s ? sm[threadIdx.x - s] : 0;
            __syncthreads();
            sm[threadIdx.x] += add;
            __syncthreads();
        }
        const uint32_t incl = sm[threadIdx.x];
        const uint32_t c = carry;
        if (i < tiles) tile_sums[i] = c + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = c + incl;
        __syncthreads();
    }
}

// pass 3: exclusive scan inside each tile, offset by the tile's start; in place (counts -> starts)
__global__ void __launch_bounds__(256) scan_apply_kernel(uint32_t *__restrict__ counts, uint64_t cells,
                                                         const uint32_t *__restrict__ tile_offsets, uint32_t total) {
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * 16;
    uint32_t v[16], sum = 0;
    for (int j = 0; j < 16; ++j) {
        v[j] = base + j < cells ? counts[base + j] : 0;
        sum += v[j];
    }
    __shared__ uint32_t sm[256];
    sm[threadIdx.x] = sum;
    __syncthreads();
    for (int s = 1; s < 256; s <<= 1) {
        const uint32_t add = (int)threadIdx.x >= s ? sm[threadIdx.x - s] : 0;
        __syncthreads();
        sm[threadIdx.x] += add;
        __syncthreads();
    }
    uint32_t run = tile_offsets[blockIdx.x] + sm[threadIdx.x] - sum;
    for (int j = 0; j < 16; ++j) {
        if (base + j < cells) counts[base + j] = run;
        run += v[j];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) counts[cells] = total;
}

__global__ void __launch_bounds__(256) scatter_kernel(const double *__restrict__ xyz, uint32_t n, GridGeom G,
                                                      const uint8_t *__restrict__ lvl, uint8_t want,
                                                      uint32_t *__restrict__ cursor, Pt4 *__restrict__ pts) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (lvl && lvl[i] != want) continue;
        const double x = xyz[(size_t)i * 3], y = xyz[(size_t)i * 3 + 1], z = xyz[(size_t)i * 3 + 2];
        const uint32_t c = cell_of(G, x, y, z);
        const uint32_t p = atomicAdd(&cursor[c], 1u);
        Pt4 o;
        o.x = x;
        o.y = y;
        o.z = z;
        o.w = i;
        pts[p] = o;
    }
}

// The scatter's atomics leave the order inside a cell to chance.  Every cell is put into a fixed order -- by x,
// ties by original index -- for two reasons: runs become reproducible to the byte, and a (y,z) row of cells is
// then sorted by x as a whole, which lets the search window long rows (knn_device.h, windowed_scan).  One thread per
// POINT: it counts the points of its cell that come before it and writes itself to that place of the ordered copy
// (a cell's few records are read by all of its points at once: cache hits; no dependent chain, no thread for the
// empty nine tenths of the table -- one thread per cell doing an insertion sort took 296 us per 1 M points, 28 % of the
// build).  Cells above PCR_HEAVY_CELL points are copied as they are and go through the segmented sort (pcr_sort.hip).
__global__ void __launch_bounds__(256) cell_order_kernel(const uint32_t *__restrict__ cs, GridGeom G,
                                                         const Pt4 *__restrict__ in, uint32_t n, Pt4 *__restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Pt4 v = in[i];
    const uint32_t c = cell_of(G, v.x, v.y, v.z);  // the scatter's own expression on the same doubles
    const uint32_t b = cs[c], e = cs[c + 1];
    uint32_t rank = i - b;
    if (e - b <= PCR_HEAVY_CELL) {
        rank = 0;
        for (uint32_t j = b; j < e; ++j) {
            const Pt4 u = in[j];
            rank += ((u.x < v.x) | ((u.x == v.x) & ((uint32_t)u.w < (uint32_t)v.w))) ? 1u : 0u;
        }
    }
    out[b + rank] = v;
}

__global__ void __launch_bounds__(256) pos_of_kernel(const Pt4 *__restrict__ pts, uint32_t n,
                                                     uint32_t *__restrict__ pos_of) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) pos_of[(uint32_t)pts[i].w] = i;
}

// integer key-space copy of a level's points (GridView::shadow): the same expression the query side uses (key_coord)
__global__ void __launch_bounds__(256) shadow_kernel(const Pt4 *__restrict__ pts, uint32_t n, double lo0, double lo1,
                                                     double lo2, double sc, uint2 *__restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const Pt4 p = pts[i];
        const uint32_t X = (uint32_t)__double2int_rn((p.x - lo0) * sc), Y = (uint32_t)__double2int_rn((p.y - lo1) * sc),
                       Z = (uint32_t)__double2int_rn((p.z - lo2) * sc);
        out[i] = make_uint2(X, (Y & 0xFFFFu) | (Z << 16));
    }
}

__global__ void __launch_bounds__(256) pack_kernel(const double *__restrict__ xyz, uint32_t n, Pt4 *__restrict__ pts,
                                                   uint32_t *__restrict__ pos_of) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        Pt4 o;
        o.x = xyz[(size_t)i * 3];
        o.y = xyz[(size_t)i * 3 + 1];
        o.z = xyz[(size_t)i * 3 + 2];
        o.w = i;
        pts[i] = o;
        pos_of[i] = i;
    }
}

// ------------------------------------------------------------------ strip table
// ---- compact cell table (GridView::cbits / cstart) from the dense one.  One thread per cell; a wave's ballot is the
// occupancy word of 64 cells, i.e. of two table words.
__global__ void __launch_bounds__(256) cell_bits_kernel(const uint32_t *__restrict__ cs, uint64_t cells, uint64_t words,
                                                        uint2 *__restrict__ cbits, uint32_t *__restrict__ word_pop) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const int lane = (int)(threadIdx.x & 63u);
    for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < words * 32u; c += stride) {
        const bool occ = c < cells && cs[c + 1] > cs[c];
        const unsigned long long m = __ballot(occ);
        if ((lane & 31) == 0) {
            const uint32_t bits = (uint32_t)(m >> lane);
            cbits[c >> 5].x = bits;
            word_pop[c >> 5] = (uint32_t)__popc(bits);
        }
    }
}

// word_rank: the exclusive scan of word_pop.  Writes the ranks into the table's words and every occupied cell's start
// to its place in cstart; cstart[occupied .. occupied+3] = n.
__global__ void __launch_bounds__(256) cell_compact_kernel(const uint32_t *__restrict__ cs, uint64_t cells, uint64_t words,
                                                           const uint32_t *__restrict__ word_rank, uint2 *__restrict__ cbits,
                                                           uint32_t *__restrict__ cstart, uint32_t occupied, uint32_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t first = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (first < 4) cstart[occupied + first] = n;
    for (uint64_t c = first; c < words * 32u; c += stride) {
        const uint64_t w = c >> 5;
        const uint32_t bits = cbits[w].x, rank = word_rank[w], b = (uint32_t)(c & 31u);
        if (b == 0) cbits[w].y = rank;
        if (c < cells && ((bits >> b) & 1u)) cstart[rank + (uint32_t)__popc(bits & ((1u << b) - 1u))] = cs[c];
    }
}

// Points per strip cell: a strip is 8x8 (y,z) rows of cells, x stays at full resolution (GridView::cs8).  One thread
// per strip cell; consecutive threads are consecutive x, so every one of the 64 reads is coalesced.
__global__ void __launch_bounds__(256) strip_count_kernel(const uint32_t *__restrict__ cs, int gx, int gy, int gz, int gy8,
                                                          uint64_t strips, uint32_t *__restrict__ out) {
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= strips) return;
    const int x = (int)(t % (uint64_t)gx);
    const uint64_t s = t / (uint64_t)gx;
    const int y8 = (int)(s % (uint64_t)gy8), z8 = (int)(s / (uint64_t)gy8);
    uint32_t c = 0;
    for (int dz = 0; dz < 8; ++dz) {
        const int z = z8 * 8 + dz;
        if (z >= gz) break;
        for (int dy = 0; dy < 8; ++dy) {
            const int y = y8 * 8 + dy;
            if (y >= gy) break;
            const size_t cell = ((size_t)z * gy + y) * gx + x;
            c += cs[cell + 1] - cs[cell];
        }
    }
    out[t] = c;
}

// ------------------------------------------------------------------ nested levels (uneven density)
// Where are the cells that hold too many points?  For two thresholds thr[0] <= thr[1], over the cells with more
// than thr[t] points: their total mass and the box of their cell coordinates.
//   stats[t*8 + 0] mass, [t*8 + 1..3] min cell coordinate per axis, [t*8 + 4..6] max cell coordinate per axis
// (initialised by the host to 0 / 0xffffffff / 0).  One atomic set per wave.
__global__ void __launch_bounds__(256) heavy_stats_kernel(const uint32_t *__restrict__ counts, uint64_t cells, uint32_t gx,
                                                          uint32_t gy, uint32_t thr0, uint32_t thr1,
                                                          uint32_t *__restrict__ stats) {
    uint32_t mass[2] = {0, 0}, mn[2][3], mx[2][3];
    for (int t = 0; t < 2; ++t)
        for (int a = 0; a < 3; ++a) {
            mn[t][a] = 0xffffffffu;
            mx[t][a] = 0u;
        }
    for (uint64_t c = (uint64_t)blockIdx.x * 256 + threadIdx.x; c < cells; c += (uint64_t)gridDim.x * 256) {
        const uint32_t v = counts[c];
        if (v > thr0) {
            const uint32_t co[3] = {(uint32_t)(c % gx), (uint32_t)((c / gx) % gy), (uint32_t)(c / ((uint64_t)gx * gy))};
            for (int t = 0; t < 2; ++t)
                if (v > (t ? thr1 : thr0)) {
                    mass[t] += v;
                    for (int a = 0; a < 3; ++a) {
                        mn[t][a] = min(mn[t][a], co[a]);
                        mx[t][a] = max(mx[t][a], co[a]);
                    }
                }
        }
    }
    for (int t = 0; t < 2; ++t) {
        for (int off = 32; off > 0; off >>= 1) {
            mass[t] += __shfl_down(mass[t], off, 64);
            for (int a = 0; a < 3; ++a) {
                mn[t][a] = min(mn[t][a], (uint32_t)__shfl_down(mn[t][a], off, 64));
                mx[t][a] = max(mx[t][a], (uint32_t)__shfl_down(mx[t][a], off, 64));
            }
        }
        if ((threadIdx.x & 63) == 0 && mass[t]) {
            atomicAdd(&stats[t * 8], mass[t]);
            for (int a = 0; a < 3; ++a) {
                atomicMin(&stats[t * 8 + 1 + a], mn[t][a]);
                atomicMax(&stats[t * 8 + 4 + a], mx[t][a]);
            }
        }
    }
}

struct Box3 {
    double lo[3], hi[3];
};

// points of level `want` inside the box move one level down; *moved counts them
__global__ void __launch_bounds__(256) assign_level_kernel(const double *__restrict__ xyz, uint32_t n, Box3 B,
                                                           uint8_t *__restrict__ lvl, uint8_t want,
                                                           uint32_t *__restrict__ moved) {
    uint32_t c = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (lvl[i] != want) continue;
        const double x = xyz[(size_t)i * 3], y = xyz[(size_t)i * 3 + 1], z = xyz[(size_t)i * 3 + 2];
        if ((x >= B.lo[0]) & (x <= B.hi[0]) & (y >= B.lo[1]) & (y <= B.hi[1]) & (z >= B.lo[2]) & (z <= B.hi[2])) {
            lvl[i] = want + 1;
            ++c;
        }
    }
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(moved, c);
}

// ------------------------------------------------------------------ brick ordering of queries
struct BrickGeom {
    GridGeom G;
    int shift;  // brick edge = 2^shift cells
    int bits;   // Morton bits per axis
};

// spread the low 10 bits of v so that there are two zero bits between consecutive bits
__device__ __forceinline__ uint32_t spread3(uint32_t v) {
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

__device__ __forceinline__ uint32_t brick_of(const BrickGeom &B, double x, double y, double z) {
    double tx = (x - B.G.lo[0]) * B.G.inv_h, ty = (y - B.G.lo[1]) * B.G.inv_h, tz = (z - B.G.lo[2]) * B.G.inv_h;
    // non-finite coordinates land in brick 0 (their lanes are masked later)
    tx = tx > 0.0 ? tx : 0.0;
    ty = ty > 0.0 ? ty : 0.0;
    tz = tz > 0.0 ? tz : 0.0;
    const double mx = (double)(B.G.g[0] - 1), my = (double)(B.G.g[1] - 1), mz = (double)(B.G.g[2] - 1);
    tx = tx > mx ? mx : tx;
    ty = ty > my ? my : ty;
    tz = tz > mz ? mz : tz;
    const int bx = (int)tx >> B.shift, by = (int)ty >> B.shift, bz = (int)tz >> B.shift;
    // Morton (Z-order) key: consecutive keys are neighbours in space, so 64 consecutive queries share a
    // compact box
    return spread3((uint32_t)bx) | (spread3((uint32_t)by) << 1) | (spread3((uint32_t)bz) << 2);
}

__device__ __forceinline__ void load_xyz(const Pt4 *pts, const double *aos, uint32_t i, double &x, double &y,
                                         double &z) {
    if (pts) {
        const Pt4 p = pts[i];
        x = p.x;
        y = p.y;
        z = p.z;
    } else {
        x = aos[(size_t)i * 3];
        y = aos[(size_t)i * 3 + 1];
        z = aos[(size_t)i * 3 + 2];
    }
}

__global__ void __launch_bounds__(256) brick_histogram_kernel(const Pt4 *__restrict__ pts,
                                                              const double *__restrict__ aos, uint32_t n, BrickGeom B,
                                                              uint32_t *__restrict__ counts,
                                                              uint32_t *__restrict__ brick_id) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        double x, y, z;
        load_xyz(pts, aos, i, x, y, z);
        const uint32_t b = brick_of(B, x, y, z);
        brick_id[i] = b;
        atomicAdd(&counts[b], 1u);
    }
}

__global__ void __launch_bounds__(256) brick_scatter_kernel(const uint32_t *__restrict__ brick_id, uint32_t n,
                                                            uint32_t base, uint32_t *__restrict__ cursor,
                                                            uint32_t *__restrict__ order) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        order[atomicAdd(&cursor[brick_id[i]], 1u)] = base + i;
}

// After the scatter cursor[b] == end of brick b; start = end of the previous brick.  The scatter's atomics
// leave the order inside a brick to chance; make it ascending so that the processing order (and with it
// the order of every floating-point sum) is the same on every run.  One WAVE per brick, rank sort: the
// values are distinct, so rank = number of smaller values is a permutation.
constexpr int BRICK_SORT_CAP = 1024;
__global__ void __launch_bounds__(256) brick_sort_kernel(const uint32_t *__restrict__ ends, uint32_t nbricks,
                                                         uint32_t *__restrict__ order) {
    __shared__ uint32_t sm[4][BRICK_SORT_CAP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t nwaves = gridDim.x * 4u;
    uint32_t *buf = sm[wave];
    for (uint32_t b = blockIdx.x * 4u + wave; b < nbricks; b += nwaves) {
        const uint32_t beg = b ? ends[b - 1] : 0u, end = ends[b];
        const uint32_t len = end - beg;
        if (len < 2 || len > BRICK_SORT_CAP) continue;  // wave-uniform
        for (uint32_t i = lane; i < len; i += 64) buf[i] = order[beg + i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (uint32_t i = lane; i < len; i += 64) {
            const uint32_t v = buf[i];
            uint32_t rank = 0;
            for (uint32_t j = 0; j < len; ++j) rank += buf[j] < v;
            order[beg + rank] = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

void *ctx_scratch(pcr_ctx *ctx, int slot, size_t bytes) {
    if (bytes <= ctx->scratch_cap[slot] && ctx->scratch_d[slot]) return ctx->scratch_d[slot];
    if (ctx->scratch_d[slot]) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(ctx->scratch_d[slot]);
        ctx->scratch_d[slot] = nullptr;
        ctx->scratch_cap[slot] = 0;
    }
    const size_t want = bytes + bytes / 4 + 256;
    hipError_t e = hipMalloc(&ctx->scratch_d[slot], want);
    if (e != hipSuccess) {
        set_last_error("scratch allocation of %zu bytes failed: %s", want, hipGetErrorString(e));
        ctx->scratch_d[slot] = nullptr;
        return nullptr;
    }
    ctx->scratch_cap[slot] = want;
    // new scratch starts zeroed (the in-kernel fold's counters rely on it; nothing else minds)
    (void)hipMemsetAsync(ctx->scratch_d[slot], 0, want, ctx->stream);
    return ctx->scratch_d[slot];
}

static int grid_blocks(uint64_t work, int block = 256, int cap = 2048) {
    uint64_t b = (work + block - 1) / block;
    if (b < 1) b = 1;
    if (b > (uint64_t)cap) b = cap;
    return (int)b;
}

static int compute_bounds(pcr_ctx *ctx, const double *xyz_d, uint32_t n, const uint8_t *lvl_d, uint8_t want, double lo[3],
                          double hi[3]) {
    const int nb = grid_blocks(n, 256, 1024);
    double *out_d = nullptr;
    PCR_HIP(hipMalloc(&out_d, sizeof(double) * 6 * nb));
    hipLaunchKernelGGL(bounds_kernel, dim3(nb), dim3(256), 0, ctx->stream, xyz_d, n, lvl_d, want, out_d);
    std::vector<double> h((size_t)6 * nb);
    hipError_t e = hipMemcpyAsync(h.data(), out_d, sizeof(double) * 6 * nb, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    hipFree(out_d);
    PCR_HIP(e);
    for (int a = 0; a < 3; ++a) {
        lo[a] = h[a];
        hi[a] = h[3 + a];
    }
    for (int b = 1; b < nb; ++b)
        for (int a = 0; a < 3; ++a) {
            lo[a] = fmin(lo[a], h[(size_t)b * 6 + a]);
            hi[a] = fmax(hi[a], h[(size_t)b * 6 + 3 + a]);
        }
    return PCR_OK;
}


int brick_order(pcr_ctx *ctx, const GridView &g, const Pt4 *pts, const double *xyz_aos, uint32_t n,
                uint32_t *order_d, bool deterministic, uint32_t base) {
    BrickGeom B;
    for (int a = 0; a < 3; ++a) {
        B.G.lo[a] = g.lo[a];
        B.G.g[a] = g.g[a];
    }
    B.G.inv_h = g.inv_h;
    B.shift = g_tune[2] > 0 ? g_tune[2] : 1;
    uint64_t nbricks;
    for (;;) {  // Morton key space 2^(3*bits) <= 2^21 entries: grow the brick until 7 bits per axis suffice
        int maxnb = 1;
        for (int a = 0; a < 3; ++a) maxnb = maxnb > ((g.g[a] - 1) >> B.shift) + 1 ? maxnb : ((g.g[a] - 1) >> B.shift) + 1;
        B.bits = 1;
        while ((1 << B.bits) < maxnb) ++B.bits;
        if (B.bits <= 7) break;
        ++B.shift;
    }
    nbricks = 1ull << (3 * B.bits);
    const uint32_t tiles = (uint32_t)((nbricks + SCAN_TILE - 1) / SCAN_TILE);
    uint32_t *counts = (uint32_t *)ctx_scratch(ctx, 0, sizeof(uint32_t) * (nbricks + 1));
    uint32_t *tile_sums = (uint32_t *)ctx_scratch(ctx, 1, sizeof(uint32_t) * (tiles + 1));
    uint32_t *brick_id = (uint32_t *)ctx_scratch(ctx, 2, sizeof(uint32_t) * (size_t)n);
    unsigned long long *stats = (unsigned long long *)ctx_scratch(ctx, 3, 64);
    if (!counts || !tile_sums || !brick_id || !stats) return PCR_E_NOMEM;
    PCR_HIP(hipMemsetAsync(counts, 0, sizeof(uint32_t) * (nbricks + 1), ctx->stream));
    PCR_HIP(hipMemsetAsync(stats, 0, 16, ctx->stream));
    hipLaunchKernelGGL(brick_histogram_kernel, dim3(grid_blocks(n)), dim3(256), 0, ctx->stream, pts, xyz_aos, n, B,
                       counts, brick_id);
    hipLaunchKernelGGL(scan_tile_sums_kernel, dim3(tiles), dim3(256), 0, ctx->stream, counts, nbricks, tile_sums,
                       stats);
    hipLaunchKernelGGL(scan_tile_offsets_kernel, dim3(1), dim3(1024), 0, ctx->stream, tile_sums, tiles);
    hipLaunchKernelGGL(scan_apply_kernel, dim3(tiles), dim3(256), 0, ctx->stream, counts, nbricks, tile_sums, n);
    hipLaunchKernelGGL(brick_scatter_kernel, dim3(grid_blocks(n)), dim3(256), 0, ctx->stream, brick_id, n, base,
                       counts, order_d);
    if (deterministic)
        hipLaunchKernelGGL(brick_sort_kernel, dim3(2048), dim3(256), 0, ctx->stream, counts, (uint32_t)nbricks,
                           order_d);
    PCR_HIP(hipGetLastError());
    return PCR_OK;
}

// ------------------------------------------------------------------ coarse ordering of a query batch
// Queries arrive in arbitrary order; the search wants the 64 lanes of a wave to be neighbours in space.  A full
// sort is not needed for that -- a counting sort by the TOP bits of a 30-bit Morton key (a few hundred queries
// per bucket) is enough, and it can be done without a single scattered global atomic: every block histograms its
// tile in LDS, reserves its share of each bucket with one coalesced returning atomic per touched bucket, and a
// second kernel rebuilds the bucket starts in LDS and scatters with LDS atomics.  Not deterministic inside a
// bucket, which no K-NN result depends on (every query's answer goes to its own row).
constexpr int QS_TILE = 8192;        // queries per block (1 M queries, K = 1, e9 queries/s: 2048: 8.3, 4096: 9.05, 8192: 9.17, 16384: 8.8)
constexpr int QS_THREADS = 1024;
constexpr int QS_MAX_BUCKETS = 8192;  // 32 KiB of LDS

struct QSortGeom {
    double lo[3], scale;  // (v - lo) * scale -> [0, 1024)
    int shift;            // key = morton30 >> shift
    uint32_t buckets;
};

__device__ __forceinline__ uint32_t qsort_key(const QSortGeom &Q, const double *__restrict__ q, uint32_t i) {
    double tx = (q[(size_t)i * 3] - Q.lo[0]) * Q.scale, ty = (q[(size_t)i * 3 + 1] - Q.lo[1]) * Q.scale,
           tz = (q[(size_t)i * 3 + 2] - Q.lo[2]) * Q.scale;
    tx = tx > 0.0 ? tx : 0.0;  // NaN -> 0: such queries are masked by the search itself
    ty = ty > 0.0 ? ty : 0.0;
    tz = tz > 0.0 ? tz : 0.0;
    tx = tx > 1023.0 ? 1023.0 : tx;
    ty = ty > 1023.0 ? 1023.0 : ty;
    tz = tz > 1023.0 ? 1023.0 : tz;
    const uint32_t m = spread3((uint32_t)tx) | (spread3((uint32_t)ty) << 1) | (spread3((uint32_t)tz) << 2);
    return m >> Q.shift;
}

__global__ void __launch_bounds__(QS_THREADS) qsort_hist_kernel(const double *__restrict__ q, uint32_t nq, QSortGeom Q,
                                                         uint16_t *__restrict__ keys, uint32_t *__restrict__ totals,
                                                         uint32_t *__restrict__ basemat) {
    __shared__ uint32_t hist[QS_MAX_BUCKETS];
    for (uint32_t b = threadIdx.x; b < Q.buckets; b += QS_THREADS) hist[b] = 0;
    __syncthreads();
    const uint32_t first = blockIdx.x * QS_TILE;
    const uint32_t last = min(first + (uint32_t)QS_TILE, nq);
    constexpr int PER = QS_TILE / QS_THREADS;
    {
        // a thread's PER queries: all coordinate loads in flight together, then the keys
        uint32_t k[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const uint32_t i = first + threadIdx.x + (uint32_t)u * QS_THREADS;
            k[u] = i < last ? qsort_key(Q, q, i) : 0xFFFFFFFFu;
        }
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const uint32_t i = first + threadIdx.x + (uint32_t)u * QS_THREADS;
            if (k[u] != 0xFFFFFFFFu) {
                keys[i] = (uint16_t)k[u];
                atomicAdd(&hist[k[u]], 1u);
            }
        }
    }
    __syncthreads();
    // The block's share of every bucket it touched: one returning atomic per bucket -- consecutive lanes, consecutive
    // buckets -- and ALL of a thread's atomics in flight before the first result is stored: a loop that stores each result
    // before issuing the next atomic is a chain of QS_MAX_BUCKETS / QS_THREADS round trips to the L2 (that chain WAS this
    // kernel's 16 us per 1 M queries).
    uint32_t *row = basemat + (size_t)blockIdx.x * Q.buckets;
    constexpr int NB = QS_MAX_BUCKETS / QS_THREADS;
    uint32_t got[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        const uint32_t b = threadIdx.x + (uint32_t)u * QS_THREADS;
        const uint32_t c = b < Q.buckets ? hist[b] : 0u;
        got[u] = c ? atomicAdd(&totals[b], c) : 0u;
    }
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        const uint32_t b = threadIdx.x + (uint32_t)u * QS_THREADS;
        if (b < Q.buckets) row[b] = got[u];
    }
}

__global__ void __launch_bounds__(QS_THREADS) qsort_scatter_kernel(const uint16_t *__restrict__ keys, uint32_t nq, QSortGeom Q,
                                                            const uint32_t *__restrict__ totals,
                                                            const uint32_t *__restrict__ basemat,
                                                            uint32_t *__restrict__ order) {
    __shared__ uint32_t cur[QS_MAX_BUCKETS];
    __shared__ uint32_t part[QS_THREADS];
    // exclusive scan of the bucket totals: each thread owns a contiguous run of buckets
    const uint32_t per = (Q.buckets + QS_THREADS - 1) / QS_THREADS;
    const uint32_t b0 = threadIdx.x * per, b1 = min(b0 + per, Q.buckets);
    // (fixed trip counts and every load issued before the first use: a run-time bound makes each load wait for the last)
    constexpr int NB = QS_MAX_BUCKETS / QS_THREADS;
    const uint32_t *row = basemat + (size_t)blockIdx.x * Q.buckets;
    uint32_t tot[NB], mine[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        const uint32_t b = b0 + (uint32_t)u;
        tot[u] = b < b1 ? totals[b] : 0u;
        mine[u] = b < b1 ? row[b] : 0u;
    }
    uint32_t sum = 0;
#pragma unroll
    for (int u = 0; u < NB; ++u) sum += tot[u];
    // block-wide exclusive scan of the per-thread sums: shuffles inside a wave, then the 16 wave totals
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    if (lane == 63) part[wave] = incl;
    __syncthreads();
    uint32_t wave_base = 0;
    for (int w = 0; w < wave; ++w) wave_base += part[w];
    __syncthreads();
    part[threadIdx.x] = wave_base + incl - sum;
    __syncthreads();
    uint32_t run = part[threadIdx.x];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        const uint32_t b = b0 + (uint32_t)u;
        if (b < b1) cur[b] = run + mine[u];
        run += tot[u];
    }
    __syncthreads();
    const uint32_t first = blockIdx.x * QS_TILE;
    const uint32_t last = min(first + (uint32_t)QS_TILE, nq);
    constexpr int PER = QS_TILE / QS_THREADS;
    uint32_t k[PER];  // (all key loads in flight together)
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const uint32_t i = first + threadIdx.x + (uint32_t)u * QS_THREADS;
        k[u] = i < last ? (uint32_t)keys[i] : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const uint32_t i = first + threadIdx.x + (uint32_t)u * QS_THREADS;
        if (k[u] != 0xFFFFFFFFu) order[atomicAdd(&cur[k[u]], 1u)] = i;
    }
}

int coarse_query_order(pcr_ctx *ctx, const GridView &g, const double *q_d, uint32_t nq, uint32_t *order_d) {
    QSortGeom Q;
    double ext = 0;
    for (int a = 0; a < 3; ++a) {
        Q.lo[a] = g.lo[a];
        ext = fmax(ext, g.hi[a] - g.lo[a]);
    }
    Q.scale = ext > 0 ? 1024.0 / ext : 0.0;
    // aim at ~192 queries per bucket; points on a surface fill only a fraction of the Morton cube, so be generous
    int bits = 6;
    while (bits < 13 && ((uint64_t)192 << bits) < (uint64_t)nq * 2) ++bits;
    Q.shift = 30 - bits;
    Q.buckets = 1u << bits;
    const uint32_t nblk = (nq + QS_TILE - 1) / QS_TILE;
    uint16_t *keys = (uint16_t *)ctx_scratch(ctx, 0, sizeof(uint16_t) * (size_t)nq);
    uint32_t *totals = (uint32_t *)ctx_scratch(ctx, 1, sizeof(uint32_t) * Q.buckets);
    uint32_t *basemat = (uint32_t *)ctx_scratch(ctx, 2, sizeof(uint32_t) * (size_t)nblk * Q.buckets);
    if (!keys || !totals || !basemat) return PCR_E_NOMEM;
    PCR_HIP(hipMemsetAsync(totals, 0, sizeof(uint32_t) * Q.buckets, ctx->stream));
    hipLaunchKernelGGL(qsort_hist_kernel, dim3(nblk), dim3(QS_THREADS), 0, ctx->stream, q_d, nq, Q, keys, totals, basemat);
    hipLaunchKernelGGL(qsort_scatter_kernel, dim3(nblk), dim3(QS_THREADS), 0, ctx->stream, (const uint16_t *)keys, nq, Q,
                       (const uint32_t *)totals, (const uint32_t *)basemat, order_d);
    PCR_HIP(hipGetLastError());
    return PCR_OK;
}

}  // namespace pcr

using namespace pcr;

extern "C" {

const char *pcr_strerror(int s) {
    switch (s) {
        case PCR_OK: return "ok";
        case PCR_E_INVALID: return "invalid argument";
        case PCR_E_HIP: return "HIP runtime error";
        case PCR_E_NOMEM: return "out of memory";
        case PCR_E_UNSUPPORTED: return "unsupported";
        case PCR_E_EMPTY: return "no correspondence survived the gates";
        case PCR_E_NO_DEVICE: return "no usable gfx950 device";
        default: return "unknown status";
    }
}

const char *pcr_last_error(void) { return g_last_error; }

int pcr_tune(int key, int value) {
    if (key < 0 || key >= 8) return PCR_E_INVALID;
    g_tune[key] = value;
    return PCR_OK;
}

const char *pcr_version(void) { return "libpcr 0.1 (gfx950, fp64, -ffp-contract=off)"; }

int pcr_ctx_create(int device, pcr_ctx **out) {
    PCR_REQUIRE(out, "pcr_ctx_create: out is NULL");
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        set_last_error("pcr_ctx_create: no HIP device visible");
        return PCR_E_NO_DEVICE;
    }
    PCR_REQUIRE(device >= 0 && device < count, "pcr_ctx_create: device %d out of range [0,%d)", device, count);
    PCR_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    PCR_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_last_error("pcr_ctx_create: device %d is %s; this library holds gfx950 code only", device,
                       prop.gcnArchName);
        return PCR_E_NO_DEVICE;
    }
    {
        const int rc = noalloc_scratch_init(device);  // once per device: the fixed scratch of pcr_icp_step_noalloc_d
        if (rc != PCR_OK) return rc;
    }
    pcr_ctx *c = new pcr_ctx();
    c->device = device;
    c->profiling = false;
    c->prof_mask = 0xfu;
    c->prof = nullptr;
    c->own_stream = true;
    c->stream = nullptr;
    c->block_partials_d = nullptr;
    c->block_partials_cap = 0;
    c->result_d = nullptr;
    c->result_h = nullptr;
    for (int i = 0; i < 6; ++i) {
        c->scratch_d[i] = nullptr;
        c->scratch_cap[i] = 0;
    }
    {
        hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            delete c;
            PCR_HIP(e);
        }
    }
    c->cu_count = 0;
    (void)hipDeviceGetAttribute(&c->cu_count, hipDeviceAttributeMultiprocessorCount, device);
    hipError_t e = hipMalloc(&c->result_d, sizeof(double) * 32);
    if (e == hipSuccess) e = hipHostMalloc(&c->result_h, sizeof(double) * 32, hipHostMallocDefault);
    if (e != hipSuccess) {
        pcr_ctx_destroy(c);
        PCR_HIP(e);
    }
    *out = c;
    return PCR_OK;
}

int pcr_ctx_destroy(pcr_ctx *c) {
    if (!c) return PCR_OK;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    if (c->block_partials_d) hipFree(c->block_partials_d);
    if (c->result_d) hipFree(c->result_d);
    if (c->result_h) hipHostFree(c->result_h);
    for (int i = 0; i < 6; ++i)
        if (c->scratch_d[i]) hipFree(c->scratch_d[i]);
    if (c->prof) {
        for (int k = 0; k < PCR_PROF_CLASSES; ++k)
            for (int i = 0; i < c->prof->created[k]; ++i) {
                hipEventDestroy(c->prof->start[k][i]);
                hipEventDestroy(c->prof->stop[k][i]);
            }
        delete c->prof;
    }
    if (c->own_stream && c->stream) hipStreamDestroy(c->stream);
    delete c;
    return PCR_OK;
}

int pcr_ctx_set_profiling(pcr_ctx *c, int on) {
    PCR_REQUIRE(c, "pcr_ctx_set_profiling: ctx is NULL");
    if (on && !c->prof) {
        c->prof = new pcr_prof();
        memset(c->prof, 0, sizeof(pcr_prof));
    }
    c->profiling = on != 0;
    c->prof_mask = on == 1 ? 0xfu : ((unsigned)on >> 1) & 0xfu;
    c->prof_stride = on == 1 ? 1u : ((unsigned)on >> 8) & 0xffu;
    for (int k = 0; k < PCR_PROF_CLASSES; ++k) c->prof_seq[k] = 0;
    return PCR_OK;
}

int pcr_ctx_read_kernel_times(pcr_ctx *c, double sum_ms[4], uint32_t launches[4]) {
    PCR_REQUIRE(c && sum_ms && launches, "pcr_ctx_read_kernel_times: NULL argument");
    for (int k = 0; k < PCR_PROF_CLASSES; ++k) {
        sum_ms[k] = 0;
        launches[k] = 0;
    }
    if (!c->prof) return PCR_OK;
    PCR_HIP(hipSetDevice(c->device));
    PCR_HIP(hipStreamSynchronize(c->stream));
    for (int k = 0; k < PCR_PROF_CLASSES; ++k) {
        for (int i = 0; i < c->prof->used[k]; ++i) {
            float ms = 0;
            PCR_HIP(hipEventElapsedTime(&ms, c->prof->start[k][i], c->prof->stop[k][i]));
            sum_ms[k] += ms;
        }
        launches[k] = (uint32_t)c->prof->used[k];
        c->prof->used[k] = 0;
    }
    return PCR_OK;
}

int pcr_ctx_read_kernel_launches(pcr_ctx *c, int cls, double *ms_out, uint32_t cap, uint32_t *count) {
    PCR_REQUIRE(c && count && (ms_out || cap == 0), "pcr_ctx_read_kernel_launches: NULL argument");
    PCR_REQUIRE(cls >= 0 && cls < PCR_PROF_CLASSES, "pcr_ctx_read_kernel_launches: class %d", cls);
    *count = 0;
    if (!c->prof) return PCR_OK;
    PCR_HIP(hipSetDevice(c->device));
    PCR_HIP(hipStreamSynchronize(c->stream));
    const int n = c->prof->used[cls];
    for (int i = 0; i < n && (uint32_t)i < cap; ++i) {
        float ms = 0;
        PCR_HIP(hipEventElapsedTime(&ms, c->prof->start[cls][i], c->prof->stop[cls][i]));
        ms_out[i] = ms;
    }
    *count = (uint32_t)n;
    return PCR_OK;
}

int pcr_ctx_set_stream(pcr_ctx *c, void *hip_stream) {
    PCR_REQUIRE(c, "pcr_ctx_set_stream: ctx is NULL");
    PCR_HIP(hipSetDevice(c->device));
    PCR_HIP(hipStreamSynchronize(c->stream));  // nothing of ours may still be in flight on the old stream
    if (c->own_stream && c->stream) PCR_HIP(hipStreamDestroy(c->stream));
    c->own_stream = false;
    c->stream = (hipStream_t)hip_stream;
    return PCR_OK;
}

int pcr_ctx_synchronize(pcr_ctx *c) {
    PCR_REQUIRE(c, "pcr_ctx_synchronize: ctx is NULL");
    PCR_HIP(hipSetDevice(c->device));
    PCR_HIP(hipStreamSynchronize(c->stream));
    return PCR_OK;
}

static int cloud_finish(pcr_ctx *ctx, pcr_cloud *c) {
    if (c->n == 0) {
        for (int a = 0; a < 3; ++a) c->lo[a] = c->hi[a] = 0.0;
        return PCR_OK;
    }
    return compute_bounds(ctx, c->xyz_d, c->n, nullptr, 0, c->lo, c->hi);
}

int pcr_cloud_upload(pcr_ctx *ctx, const double *xyz, uint32_t n, pcr_cloud **out) {
    PCR_REQUIRE(ctx && out, "pcr_cloud_upload: NULL argument");
    *out = nullptr;
    PCR_REQUIRE(xyz || n == 0, "pcr_cloud_upload: xyz is NULL with n=%u", n);
    PCR_HIP(hipSetDevice(ctx->device));
    double *d = nullptr;
    if (n) {
        PCR_HIP(hipMalloc(&d, sizeof(double) * 3 * (size_t)n));
        hipError_t e = hipMemcpyAsync(d, xyz, sizeof(double) * 3 * (size_t)n, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            hipFree(d);
            PCR_HIP(e);
        }
    }
    pcr_cloud *c = new pcr_cloud();
    c->ctx = ctx;
    c->n = n;
    c->xyz_d = d;
    c->owned = true;
    int rc = cloud_finish(ctx, c);
    if (rc != PCR_OK) {
        pcr_cloud_free(c);
        return rc;
    }
    *out = c;
    return PCR_OK;
}

int pcr_cloud_wrap_d(pcr_ctx *ctx, const double *xyz_d, uint32_t n, pcr_cloud **out) {
    PCR_REQUIRE(ctx && out, "pcr_cloud_wrap_d: NULL argument");
    *out = nullptr;
    PCR_REQUIRE(xyz_d || n == 0, "pcr_cloud_wrap_d: xyz_d is NULL with n=%u", n);
    PCR_HIP(hipSetDevice(ctx->device));
    pcr_cloud *c = new pcr_cloud();
    c->ctx = ctx;
    c->n = n;
    c->xyz_d = xyz_d;
    c->owned = false;
    int rc = cloud_finish(ctx, c);
    if (rc != PCR_OK) {
        delete c;
        return rc;
    }
    *out = c;
    return PCR_OK;
}

int pcr_cloud_free(pcr_cloud *c) {
    if (!c) return PCR_OK;
    hipSetDevice(c->ctx->device);
    if (c->owned && c->xyz_d) hipFree(const_cast<double *>(c->xyz_d));
    if (c->normals_owned && c->normals_d) hipFree(const_cast<double *>(c->normals_d));
    delete c;
    return PCR_OK;
}

static void drop_normals(pcr_cloud *c) {
    if (c->normals_owned && c->normals_d) hipFree(const_cast<double *>(c->normals_d));
    c->normals_d = nullptr;
    c->normals_owned = false;
}

int pcr_cloud_set_normals(pcr_cloud *c, const double *normals) {
    PCR_REQUIRE(c, "pcr_cloud_set_normals: NULL cloud");
    PCR_HIP(hipSetDevice(c->ctx->device));
    PCR_HIP(hipStreamSynchronize(c->ctx->stream));
    drop_normals(c);
    if (!normals || c->n == 0) return PCR_OK;
    double *d = nullptr;
    PCR_HIP(hipMalloc(&d, sizeof(double) * 3 * (size_t)c->n));
    hipError_t e = hipMemcpyAsync(d, normals, sizeof(double) * 3 * (size_t)c->n, hipMemcpyHostToDevice, c->ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->ctx->stream);
    if (e != hipSuccess) {
        hipFree(d);
        PCR_HIP(e);
    }
    c->normals_d = d;
    c->normals_owned = true;
    return PCR_OK;
}

int pcr_cloud_set_normals_d(pcr_cloud *c, const double *normals_d) {
    PCR_REQUIRE(c, "pcr_cloud_set_normals_d: NULL cloud");
    PCR_HIP(hipSetDevice(c->ctx->device));
    PCR_HIP(hipStreamSynchronize(c->ctx->stream));
    drop_normals(c);
    c->normals_d = normals_d;
    return PCR_OK;
}

uint32_t pcr_cloud_size(const pcr_cloud *c) { return c ? c->n : 0; }

int pcr_cloud_bounds(const pcr_cloud *c, double lo[3], double hi[3]) {
    PCR_REQUIRE(c && lo && hi, "pcr_cloud_bounds: NULL argument");
    for (int a = 0; a < 3; ++a) {
        lo[a] = c->lo[a];
        hi[a] = c->hi[a];
    }
    return PCR_OK;
}

int pcr_index_free(pcr_index *ix) {
    if (!ix) return PCR_OK;
    hipSetDevice(ix->ctx->device);
    if (ix->pts_d) hipFree(ix->pts_d);
    if (ix->shadow_d) hipFree(ix->shadow_d);
    if (ix->view_d) hipFree(ix->view_d);
    for (int l = 0; l < PCR_MAX_LEVELS; ++l) {
        if (ix->cs_raw_d[l]) hipFree(ix->cs_raw_d[l]);
        if (ix->cs8_d[l]) hipFree(ix->cs8_d[l]);
        if (ix->cbits_d[l]) hipFree(ix->cbits_d[l]);
        if (ix->cstart_d[l]) hipFree(ix->cstart_d[l]);
    }
    if (ix->pos_of_d) hipFree(ix->pos_of_d);
    if (ix->order_d) hipFree(ix->order_d);
    delete ix;
    return PCR_OK;
}

namespace {

// Everything the build of ONE level's grid needs and leaves behind.
struct LevelBuild {
    GridGeom G;
    double h;
    uint64_t cells;
    uint32_t *counts_raw;   // the allocation: [cells+3] = one zero entry of padding either side of counts_d, so that the
                            // filtered search can fetch the four row boundaries cs[c-1..c+2] of any cell c with one load
    uint32_t *counts_d;     // = counts_raw + 1, [cells+1]: histogram, then exclusive starts; becomes the level's cell table
    uint32_t *tile_sums_d;  // scan scratch
    size_t counts_cap, tiles_cap;
    unsigned long long stats_h[2];  // occupied cells, largest cell
};

void release(LevelBuild &B) {
    if (B.counts_raw) hipFree(B.counts_raw);
    if (B.tile_sums_d) hipFree(B.tile_sums_d);
    B.counts_raw = B.counts_d = B.tile_sums_d = nullptr;
    B.counts_cap = B.tiles_cap = 0;
}

void make_geom(const double lo[3], const double hi[3], double h, uint64_t max_cells, GridGeom &G, double &h_out,
               uint64_t &cells) {
    for (;;) {
        cells = 1;
        for (int a = 0; a < 3; ++a) {
            const double ext = hi[a] - lo[a];
            double g = floor(ext / h) + 1.0;
            if (!(g >= 1.0)) g = 1.0;
            if (g > 2097152.0) g = 2097152.0;
            G.g[a] = (int)g;
            G.lo[a] = lo[a];
            cells *= (uint64_t)G.g[a];
        }
        if (cells <= max_cells) break;
        h *= 1.26;  // ~ cube root of 2: halves the cell count of a volume per step
    }
    G.inv_h = 1.0 / h;
    h_out = h;
}

// one trial of the histogram at a given cell size: fills counts (device) and the occupancy statistics
int histogram_trial(pcr_ctx *ctx, const pcr_cloud *cloud, const uint8_t *lvl_d, uint8_t level, LevelBuild &B,
                    unsigned long long *stats_d) {
    PCR_HIP(hipMemsetAsync(B.counts_raw, 0, sizeof(uint32_t) * (B.cells + 3), ctx->stream));
    PCR_HIP(hipMemsetAsync(stats_d, 0, sizeof(unsigned long long) * 2, ctx->stream));
    hipLaunchKernelGGL(cell_histogram_kernel, dim3(grid_blocks(cloud->n)), dim3(256), 0, ctx->stream, cloud->xyz_d,
                       cloud->n, B.G, lvl_d, level, B.counts_d);
    const uint32_t tiles = (uint32_t)((B.cells + SCAN_TILE - 1) / SCAN_TILE);
    hipLaunchKernelGGL(scan_tile_sums_kernel, dim3(tiles), dim3(256), 0, ctx->stream, B.counts_d, B.cells,
                       B.tile_sums_d, stats_d);
    PCR_HIP(hipMemcpyAsync(B.stats_h, stats_d, sizeof(unsigned long long) * 2, hipMemcpyDeviceToHost, ctx->stream));
    PCR_HIP(hipStreamSynchronize(ctx->stream));
    return PCR_OK;
}

// Chooses the cell edge for the n_l points of one level inside [lo,hi] -- aiming at target_occupancy points per
// OCCUPIED cell -- and leaves the histogram of the final choice in B.counts_d (tile sums in B.tile_sums_d).
int choose_cells(pcr_ctx *ctx, const pcr_cloud *cloud, const uint8_t *lvl_d, uint8_t level, uint32_t n_l,
                 const double lo[3], const double hi[3], const pcr_index_params &P, double h_min,
                 unsigned long long *stats_d, LevelBuild &B) {
    double vol = 1.0;
    int live = 0;
    for (int a = 0; a < 3; ++a) {
        const double ext = hi[a] - lo[a];
        if (ext > 0) {
            vol *= ext;
            ++live;
        }
    }
    double h;
    if (P.cell_size > 0)
        h = P.cell_size;
    else if (live == 0)
        h = 1.0;
    else
        h = pow(vol * P.target_occupancy / (double)n_l, 1.0 / live);  // volume-uniform first guess
    if (!(h > 0) || !isfinite(h)) h = 1.0;
    if (h < h_min) h = h_min > 0 ? h_min : 1.0;
    for (int trial = 0; trial < 8; ++trial) {
        double h_used;
        make_geom(lo, hi, h, P.max_cells, B.G, h_used, B.cells);
        h = h_used;
        if (B.cells + 1 > B.counts_cap) {
            if (B.counts_raw) hipFree(B.counts_raw);
            B.counts_raw = B.counts_d = nullptr;
            B.counts_cap = 0;
            PCR_HIP(hipMalloc(&B.counts_raw, sizeof(uint32_t) * (B.cells + 3)));
            B.counts_d = B.counts_raw + 1;
            B.counts_cap = B.cells + 1;
        }
        const size_t tiles = (B.cells + SCAN_TILE - 1) / SCAN_TILE + 1;
        if (tiles > B.tiles_cap) {
            if (B.tile_sums_d) hipFree(B.tile_sums_d);
            B.tile_sums_d = nullptr;
            B.tiles_cap = 0;
            PCR_HIP(hipMalloc(&B.tile_sums_d, sizeof(uint32_t) * tiles));
            B.tiles_cap = tiles;
        }
        int rc = histogram_trial(ctx, cloud, lvl_d, level, B, stats_d);
        if (rc != PCR_OK) return rc;
        if (P.cell_size > 0) break;
        const double occ = (double)n_l / (double)(B.stats_h[0] ? B.stats_h[0] : 1);
        const double ratio = P.target_occupancy / occ;
        if (ratio < 1.35 && ratio > 0.74) break;  // close enough
        if (ratio < 1.0 && B.cells == 1) break;
        // points on a surface: occupancy ~ h^2.  (A volume converges too, in a damped oscillation.)
        double nh = h * sqrt(ratio);
        if (nh < h_min) nh = h_min;
        if (nh == h) break;
        // growing h is always possible; shrinking may hit max_cells, in which case make_geom undoes it
        if (trial == 7) break;
        h = nh;
    }
    B.h = h;
    return PCR_OK;
}

// Splitting rule of the nested levels.  A level is split when a noticeable share of its points sits in cells far
// above the occupancy the cell size was chosen for (the cell size follows the MEAN; a LiDAR sweep's density varies
// by four orders of magnitude) and those cells are confined to a part of the level's box.  The box of the cells
// above a lower mark goes down one level, where it gets a cell size of its own.
constexpr uint32_t SPLIT_TRIGGER = 64;     // "far too full"
constexpr uint32_t SPLIT_BOX_MARK = 32;    // the inner box spans the cells above this
constexpr uint32_t SPLIT_MIN_POINTS = 8192;
constexpr double SPLIT_MIN_SHARE = 0.05;   // of the level's points in far-too-full cells
constexpr double SPLIT_MAX_VOLUME = 0.5;   // inner box / level box

// Decides on a split of the level whose histogram is in B; on yes, fills `inner` with the box to hand down.
int find_dense_box(pcr_ctx *ctx, const LevelBuild &B, uint32_t n_l, const double lo[3], const double hi[3],
                   uint32_t *stats16_d, Box3 &inner, bool &split) {
    split = false;
    uint32_t init[16], got[16];
    for (int t = 0; t < 2; ++t) {
        init[t * 8] = 0;
        for (int a = 0; a < 3; ++a) {
            init[t * 8 + 1 + a] = 0xffffffffu;
            init[t * 8 + 4 + a] = 0;
        }
        init[t * 8 + 7] = 0;
    }
    PCR_HIP(hipMemcpyAsync(stats16_d, init, sizeof init, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(heavy_stats_kernel, dim3(grid_blocks(B.cells, 256, 4096)), dim3(256), 0, ctx->stream,
                       (const uint32_t *)B.counts_d, B.cells, (uint32_t)B.G.g[0], (uint32_t)B.G.g[1], SPLIT_BOX_MARK,
                       SPLIT_TRIGGER, stats16_d);
    PCR_HIP(hipMemcpyAsync(got, stats16_d, sizeof got, hipMemcpyDeviceToHost, ctx->stream));
    PCR_HIP(hipStreamSynchronize(ctx->stream));
    const double share = (double)got[8] / (double)n_l;
    if (got[8] < SPLIT_MIN_POINTS || share < SPLIT_MIN_SHARE) return PCR_OK;
    for (int t = 0; t < 2 && !split; ++t) {  // the generous box first, the tight one if that spans too much
        double frac = 1.0;
        for (int a = 0; a < 3; ++a) {
            inner.lo[a] = B.G.lo[a] + (double)got[t * 8 + 1 + a] * B.h;
            inner.hi[a] = B.G.lo[a] + (double)(got[t * 8 + 4 + a] + 1u) * B.h;
            if (got[t * 8 + 1 + a] == 0) inner.lo[a] = lo[a];
            if ((int)got[t * 8 + 4 + a] >= B.G.g[a] - 1 || inner.hi[a] > hi[a]) inner.hi[a] = hi[a];
            const double ext = hi[a] - lo[a];
            if (ext > 0) frac *= (inner.hi[a] - inner.lo[a]) / ext;
        }
        split = frac <= SPLIT_MAX_VOLUME;
    }
    return PCR_OK;
}

}  // namespace

int pcr_index_build(pcr_ctx *ctx, const pcr_cloud *cloud, const pcr_index_params *params, pcr_index **out) {
    PCR_REQUIRE(ctx && cloud && out, "pcr_index_build: NULL argument");
    *out = nullptr;
    PCR_REQUIRE(cloud->n > 0, "pcr_index_build: empty cloud");
    PCR_REQUIRE(cloud->ctx == ctx, "pcr_index_build: cloud belongs to another context");
    for (int a = 0; a < 3; ++a)
        PCR_REQUIRE(isfinite(cloud->lo[a]) && isfinite(cloud->hi[a]), "pcr_index_build: non-finite coordinates");
    PCR_HIP(hipSetDevice(ctx->device));
    pcr_index_params P;
    memset(&P, 0, sizeof P);
    if (params) P = *params;
    if (!(P.target_occupancy > 0)) P.target_occupancy = 4.5;
    if (P.max_cells == 0) P.max_cells = 1ull << 30;
    if (P.max_cells > (1ull << 32) - 8) P.max_cells = (1ull << 32) - 8;
    if (P.brute_max_points == 0) P.brute_max_points = 2048;
    if (P.max_levels == 0 || P.max_levels > (uint32_t)PCR_MAX_LEVELS) P.max_levels = PCR_MAX_LEVELS;
    if (P.cell_size > 0) P.max_levels = 1;  // a fixed cell edge means one uniform grid
    int kind = P.kind;
    PCR_REQUIRE(kind == PCR_INDEX_AUTO || kind == PCR_INDEX_GRID || kind == PCR_INDEX_BRUTE,
                "pcr_index_build: unknown kind %d", kind);
    if (kind == PCR_INDEX_AUTO) kind = cloud->n <= P.brute_max_points ? PCR_INDEX_BRUTE : PCR_INDEX_GRID;

    const uint32_t n = cloud->n;
    pcr_index *ix = new pcr_index();
    memset(ix, 0, sizeof *ix);
    ix->ctx = ctx;
    ix->cloud = cloud;
    ix->kind = kind;
    ix->n = n;
    int rc = PCR_OK;
    LevelBuild B;
    memset(&B, 0, sizeof B);
    uint32_t *cursor_d = nullptr, *stats16_d = nullptr;
    Pt4 *sort_tmp_d = nullptr;
    unsigned long long *stats_d = nullptr;
    uint8_t *lvl_d = nullptr;
#define BUILD_HIP(call)                                                                             \
    do {                                                                                            \
        hipError_t e__ = (call);                                                                    \
        if (e__ != hipSuccess) {                                                                    \
            set_last_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
            rc = e__ == hipErrorOutOfMemory ? PCR_E_NOMEM : PCR_E_HIP;                             \
            goto fail;                                                                              \
        }                                                                                           \
    } while (0)
#define BUILD_RC(call)            \
    do {                          \
        rc = (call);              \
        if (rc != PCR_OK) goto fail; \
    } while (0)

    BUILD_HIP(hipMalloc(&ix->pts_d, sizeof(Pt4) * (size_t)n));
    BUILD_HIP(hipMalloc(&ix->pos_of_d, sizeof(uint32_t) * (size_t)n));
    BUILD_HIP(hipMalloc(&ix->order_d, sizeof(uint32_t) * (size_t)n));

    {
        MultiView &M = ix->view;
        double scale = 0, maxext = 0;
        for (int a = 0; a < 3; ++a) {
            M.lo[a] = cloud->lo[a];
            M.hi[a] = cloud->hi[a];
            scale = fmax(scale, fmax(fabs(cloud->lo[a]), fabs(cloud->hi[a])));
            maxext = fmax(maxext, cloud->hi[a] - cloud->lo[a]);
        }
        const double margin = 9.094947017729282e-13 * (scale + maxext);  // 2^-40 of the coordinate scale
        M.levels = 1;
        M.n = n;
        M.all = ix->pts_d;
        M.pos_of = ix->pos_of_d;

        if (kind == PCR_INDEX_BRUTE) {
            GridView &V = M.v[0];
            for (int a = 0; a < 3; ++a) {
                V.lo[a] = cloud->lo[a];
                V.hi[a] = cloud->hi[a];
            }
            V.margin = margin;
            V.n = n;
            V.pts = ix->pts_d;
            V.cs = nullptr;
            V.cs8 = nullptr;
            V.cbits = nullptr;
            V.cstart = nullptr;
            V.gy8 = 1;
            V.h = V.inv_h = 1.0;
            V.g[0] = V.g[1] = V.g[2] = 1;
            hipLaunchKernelGGL(pack_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, cloud->xyz_d, n,
                               ix->pts_d, ix->pos_of_d);
            BUILD_HIP(hipMemcpyAsync(ix->order_d, ix->pos_of_d, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToDevice,
                                     ctx->stream));
            BUILD_HIP(hipStreamSynchronize(ctx->stream));
            ix->info.kind = kind;
            ix->info.n = n;
            ix->info.dims[0] = ix->info.dims[1] = ix->info.dims[2] = 1;
            ix->info.cells = ix->info.occupied_cells = 1;
            ix->info.cell_size = 0;
            ix->info.mean_occupancy = n;
            ix->info.max_occupancy = n;
            ix->info.levels = 1;
            ix->info.device_bytes = (sizeof(Pt4) + 8) * (uint64_t)n;
            *out = ix;
            return PCR_OK;
        }

        // never finer than 2^-20 of the scale: keeps the safety margin (2^-40) far below a cell
        const double h_min = (scale + maxext) * 9.5367431640625e-07;
        BUILD_HIP(hipMalloc(&stats_d, sizeof(unsigned long long) * 2));
        BUILD_HIP(hipMalloc(&stats16_d, sizeof(uint32_t) * 16));

        double lo[3], hi[3];  // box of the level being built
        for (int a = 0; a < 3; ++a) {
            lo[a] = cloud->lo[a];
            hi[a] = cloud->hi[a];
        }
        uint32_t n_l = n, off = 0;
        uint64_t occupied_total = 0, cells_total = 0, table_bytes = 0;
        uint32_t max_occ = 0;
        for (int l = 0;; ++l) {
            const uint8_t *filter = lvl_d;  // NULL while nothing has been split off: every point is level 0
            BUILD_RC(choose_cells(ctx, cloud, filter, (uint8_t)l, n_l, lo, hi, P, h_min, stats_d, B));
            uint32_t n_inner = 0;
            if (l + 1 < (int)P.max_levels && n_l >= SPLIT_MIN_POINTS && B.stats_h[1] > SPLIT_TRIGGER) {
                Box3 inner;
                bool split = false;
                BUILD_RC(find_dense_box(ctx, B, n_l, lo, hi, stats16_d, inner, split));
                if (split) {
                    if (!lvl_d) {
                        BUILD_HIP(hipMalloc(&lvl_d, n));
                        BUILD_HIP(hipMemsetAsync(lvl_d, 0, n, ctx->stream));
                    }
                    BUILD_HIP(hipMemsetAsync(stats16_d, 0, sizeof(uint32_t), ctx->stream));
                    hipLaunchKernelGGL(assign_level_kernel, dim3(grid_blocks(n)), dim3(256), 0, ctx->stream, cloud->xyz_d, n,
                                       inner, lvl_d, (uint8_t)l, stats16_d);
                    BUILD_HIP(hipMemcpyAsync(&n_inner, stats16_d, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
                    BUILD_HIP(hipStreamSynchronize(ctx->stream));
                    if (n_inner >= n_l) {
                        // cannot happen (the inner box is at most half the level's box, whose faces all touch a
                        // point); undo rather than build an empty level
                        set_last_error("pcr_index_build: level split moved every point");
                        rc = PCR_E_HIP;
                        goto fail;
                    }
                    n_l -= n_inner;
                    // the level keeps the sparse remainder: choose its cells again
                    BUILD_RC(choose_cells(ctx, cloud, lvl_d, (uint8_t)l, n_l, lo, hi, P, h_min, stats_d, B));
                }
            }
            // ---- scan + scatter into the level's segment [off, off + n_l) ----
            {
                const uint32_t tiles = (uint32_t)((B.cells + SCAN_TILE - 1) / SCAN_TILE);
                Pt4 *seg = ix->pts_d + off;
                hipLaunchKernelGGL(scan_tile_offsets_kernel, dim3(1), dim3(1024), 0, ctx->stream, B.tile_sums_d, tiles);
                hipLaunchKernelGGL(scan_apply_kernel, dim3(tiles), dim3(256), 0, ctx->stream, B.counts_d, B.cells,
                                   B.tile_sums_d, n_l);
                BUILD_HIP(hipMalloc(&cursor_d, sizeof(uint32_t) * B.cells));
                BUILD_HIP(hipMemcpyAsync(cursor_d, B.counts_d, sizeof(uint32_t) * B.cells, hipMemcpyDeviceToDevice,
                                         ctx->stream));
                // (one scratch as large as the level: the scatter's target, then the segmented sort's workspace)
                BUILD_HIP(hipMalloc(&sort_tmp_d, sizeof(Pt4) * (size_t)n_l));
                hipLaunchKernelGGL(scatter_kernel, dim3(grid_blocks(n)), dim3(256), 0, ctx->stream, cloud->xyz_d, n, B.G,
                                   (const uint8_t *)lvl_d, (uint8_t)l, cursor_d, sort_tmp_d);
                hipLaunchKernelGGL(cell_order_kernel, dim3((n_l + 255) / 256), dim3(256), 0, ctx->stream,
                                   (const uint32_t *)B.counts_d, B.G, (const Pt4 *)sort_tmp_d, n_l, seg);
                if (B.stats_h[1] > PCR_HEAVY_CELL) BUILD_RC(sort_heavy_cells(ctx, B.counts_d, B.cells, n_l, seg, sort_tmp_d));
                BUILD_HIP(hipStreamSynchronize(ctx->stream));
                BUILD_HIP(hipGetLastError());
                hipFree(cursor_d);
                cursor_d = nullptr;
                if (sort_tmp_d) hipFree(sort_tmp_d);
                sort_tmp_d = nullptr;
            }
            // ---- strip table of the level (lets the shell search skip 8x8 blocks of empty rows) ----
            const int gy8 = (B.G.g[1] + 7) / 8, gz8 = (B.G.g[2] + 7) / 8;
            const uint64_t strips = (uint64_t)B.G.g[0] * gy8 * gz8;
            {
                const uint32_t tiles8 = (uint32_t)((strips + SCAN_TILE - 1) / SCAN_TILE);
                uint32_t *tiles8_d = nullptr;
                BUILD_HIP(hipMalloc(&ix->cs8_d[l], sizeof(uint32_t) * (strips + 1)));
                BUILD_HIP(hipMalloc(&tiles8_d, sizeof(uint32_t) * (tiles8 + 1)));
                hipLaunchKernelGGL(strip_count_kernel, dim3((unsigned)((strips + 255) / 256)), dim3(256), 0, ctx->stream,
                                   (const uint32_t *)B.counts_d, B.G.g[0], B.G.g[1], B.G.g[2], gy8, strips, ix->cs8_d[l]);
                hipLaunchKernelGGL(scan_tile_sums_kernel, dim3(tiles8), dim3(256), 0, ctx->stream,
                                   (const uint32_t *)ix->cs8_d[l], strips, tiles8_d, stats_d);
                hipLaunchKernelGGL(scan_tile_offsets_kernel, dim3(1), dim3(1024), 0, ctx->stream, tiles8_d, tiles8);
                hipLaunchKernelGGL(scan_apply_kernel, dim3(tiles8), dim3(256), 0, ctx->stream, ix->cs8_d[l], strips,
                                   (const uint32_t *)tiles8_d, n_l);
                hipError_t e8 = hipStreamSynchronize(ctx->stream);
                hipFree(tiles8_d);
                BUILD_HIP(e8);
                BUILD_HIP(hipGetLastError());
            }
            GridView &V = M.v[l];
            V.cs8 = ix->cs8_d[l];
            V.gy8 = gy8;
            for (int a = 0; a < 3; ++a) {
                V.lo[a] = lo[a];
                V.hi[a] = hi[a];
                V.g[a] = B.G.g[a];
            }
            V.margin = margin;
            V.n = n_l;
            V.pts = ix->pts_d + off;
            V.cs = B.counts_d;
            V.cbits = nullptr;
            V.cstart = nullptr;
            V.h = B.h;
            V.inv_h = B.G.inv_h;
            ix->cells[l] = B.cells;
            ix->occupied[l] = B.stats_h[0];
            // ---- on request (pcr_index_params.compact_tables, tune 6) a level of a nested index trades its dense table
            // for the compact one (GridView::cbits)
            const bool nested = l > 0 || n_inner > 0;
            const int compact_mode = g_tune[6] == 1 ? 0 : g_tune[6] == 2 ? 2 : g_tune[6] == 3 ? 1 : (int)P.compact_tables;
            if (nested && (compact_mode >= 2 || (compact_mode == 1 && B.cells >= (1u << 16) && B.stats_h[0] * 8 < B.cells))) {
                const uint64_t words = B.cells / 32 + 2;
                const uint32_t occupied = (uint32_t)B.stats_h[0];
                const uint32_t wtiles = (uint32_t)((words + SCAN_TILE - 1) / SCAN_TILE);
                uint32_t *word_rank_d = nullptr, *wtiles_d = nullptr;
                BUILD_HIP(hipMalloc(&ix->cbits_d[l], sizeof(uint2) * words));
                BUILD_HIP(hipMalloc(&ix->cstart_d[l], sizeof(uint32_t) * ((size_t)occupied + 4)));
                hipError_t ec = hipMalloc(&word_rank_d, sizeof(uint32_t) * (words + 1));
                if (ec == hipSuccess) ec = hipMalloc(&wtiles_d, sizeof(uint32_t) * (wtiles + 1));
                if (ec == hipSuccess) {
                    hipLaunchKernelGGL(cell_bits_kernel, dim3(grid_blocks(words * 32, 256, 16384)), dim3(256), 0, ctx->stream,
                                       (const uint32_t *)B.counts_d, B.cells, words, ix->cbits_d[l], word_rank_d);
                    hipLaunchKernelGGL(scan_tile_sums_kernel, dim3(wtiles), dim3(256), 0, ctx->stream, (const uint32_t *)word_rank_d,
                                       words, wtiles_d, stats_d);
                    hipLaunchKernelGGL(scan_tile_offsets_kernel, dim3(1), dim3(1024), 0, ctx->stream, wtiles_d, wtiles);
                    hipLaunchKernelGGL(scan_apply_kernel, dim3(wtiles), dim3(256), 0, ctx->stream, word_rank_d, words,
                                       (const uint32_t *)wtiles_d, occupied);
                    hipLaunchKernelGGL(cell_compact_kernel, dim3(grid_blocks(words * 32, 256, 16384)), dim3(256), 0, ctx->stream,
                                       (const uint32_t *)B.counts_d, B.cells, words, (const uint32_t *)word_rank_d, ix->cbits_d[l],
                                       ix->cstart_d[l], occupied, n_l);
                    ec = hipStreamSynchronize(ctx->stream);
                }
                if (word_rank_d) hipFree(word_rank_d);
                if (wtiles_d) hipFree(wtiles_d);
                BUILD_HIP(ec);
                BUILD_HIP(hipGetLastError());
                V.cs = nullptr;
                V.cbits = ix->cbits_d[l];
                V.cstart = ix->cstart_d[l];
                table_bytes += sizeof(uint2) * words + sizeof(uint32_t) * ((uint64_t)occupied + 4);
                // (the dense table stays with B: the next level's histogram reuses the allocation if it fits)
            } else {
                ix->cs_d[l] = B.counts_d;  // the index owns the table from here on
                ix->cs_raw_d[l] = B.counts_raw;
                B.counts_raw = B.counts_d = nullptr;
                B.counts_cap = 0;
                table_bytes += sizeof(uint32_t) * (B.cells + 3);
            }
            table_bytes += sizeof(uint32_t) * (strips + 1);
            M.levels = l + 1;
            if (getenv("PCR_TRACE_BUILD"))
                fprintf(stderr, "[pcr] level %d: n=%u h=%.6g dims=%dx%dx%d cells=%llu occupied=%llu largest=%llu box=[%.3g %.3g %.3g]-[%.3g %.3g %.3g]\n",
                        l, n_l, B.h, B.G.g[0], B.G.g[1], B.G.g[2], (unsigned long long)B.cells, B.stats_h[0], B.stats_h[1],
                        lo[0], lo[1], lo[2], hi[0], hi[1], hi[2]);
            // processing order of "all samples": by grid brick, so that a wave's 64 queries share a small box
            BUILD_RC(brick_order(ctx, V, V.pts, nullptr, n_l, ix->order_d + off, true, off));
            BUILD_HIP(hipStreamSynchronize(ctx->stream));
            if (l == 0) {
                for (int a = 0; a < 3; ++a) ix->info.dims[a] = (uint32_t)B.G.g[a];
                ix->info.cell_size = B.h;
            }
            cells_total += B.cells;
            occupied_total += B.stats_h[0];
            max_occ = max_occ > (uint32_t)B.stats_h[1] ? max_occ : (uint32_t)B.stats_h[1];
            if (n_inner == 0) break;
            // ---- the next level: the points handed down, in their own (tight) box ----
            off += n_l;
            n_l = n_inner;
            BUILD_RC(compute_bounds(ctx, cloud->xyz_d, n, lvl_d, (uint8_t)(l + 1), lo, hi));
        }
        hipLaunchKernelGGL(pos_of_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ix->pts_d, n, ix->pos_of_d);
        // ---- key-space shadows for the filtered search, level by level (each level has its own cell size, hence its own
        // key scale).  Integer coordinates: the error bound no longer grows with the grid's extent, so every level of every
        // index gets them as long as its coordinates fit 31 bits (grids up to ~98 000 cells along an axis).  A
        // single-level index with shadows takes the filtered step kernel; the levels of a nested one take the filtered
        // walk inside multi_lane_search.
        for (int l = 0; l < M.levels; ++l) {
            GridView &V = M.v[l];
            const uint64_t B_cells = ix->cells[l];
            const double sc = 21844.0 / V.h;  // KEY_CELL (fast_search.h) units per cell
            double ext = 0;
            for (int a = 0; a < 3; ++a) ext = fmax(ext, V.hi[a] - V.lo[a]);
            // per-axis error of a coordinate difference in units: the point and the query each rounded to an integer
            const double e_axis = 1.0 * 1.01;
            // |key - S d2| <= 2 sqrt(D) sqrt(3) e + 3 e^2 + 4096 (truncation of the key's low 12 bits), sqrt(D) <= sqrt(2.5) cells
            const double delta = 2.0 * 1.5811388300841898 * 21844.0 * 1.7320508075688774 * e_axis + 3.0 * e_axis * e_axis + 4096.0 + 64.0;
            // what the walk's fp32 geometry (the query's cell, its gaps to the faces) rounds by: |Q| <= (ext + 4h) * sc
            const double eps = 1.1920928955078125e-07 * (ext + 4.0 * V.h) * sc * 1.01;
            if ((ext + 8.0 * V.h) * sc < 2147483648.0 && B_cells + 3 < (1ull << 30) && (uint64_t)V.n < (1ull << 28)) {  // 32-bit byte offsets
                if (!ix->shadow_d) BUILD_HIP(hipMalloc(&ix->shadow_d, sizeof(uint2) * (size_t)n));
                const size_t off_l = (size_t)(V.pts - ix->pts_d);
                hipLaunchKernelGGL(shadow_kernel, dim3((V.n + 255) / 256), dim3(256), 0, ctx->stream, (const Pt4 *)V.pts, V.n,
                                   V.lo[0], V.lo[1], V.lo[2], sc, ix->shadow_d + off_l);
                V.shadow = ix->shadow_d + off_l;
                V.key_sc = sc;
                V.key_S = sc * sc;
                V.key_invS = 1.0 / V.key_S;
                V.key_delta = (uint32_t)ceil(delta);
                V.key_eps = (float)(4.0 * eps + 1.0);  // a few fp32 roundings of the largest coordinate, and a unit
                for (int a = 0; a < 3; ++a) V.key_ext[a] = (float)((V.hi[a] - V.lo[a]) * sc * 1.000001);
            }
        }
        BUILD_HIP(hipMalloc(&ix->view_d, sizeof(MultiView)));
        BUILD_HIP(hipMemcpyAsync(ix->view_d, &ix->view, sizeof(MultiView), hipMemcpyHostToDevice, ctx->stream));
        BUILD_HIP(hipStreamSynchronize(ctx->stream));
        BUILD_HIP(hipGetLastError());
        ix->info.kind = kind;
        ix->info.n = n;
        ix->info.cells = cells_total;
        ix->info.occupied_cells = occupied_total;
        ix->info.mean_occupancy = (double)n / (double)(occupied_total ? occupied_total : 1);
        ix->info.max_occupancy = max_occ;
        ix->info.levels = (uint32_t)M.levels;
        ix->info.device_bytes = (sizeof(Pt4) + 8 + (ix->shadow_d ? 8 : 0)) * (uint64_t)n + table_bytes;
    }
    release(B);
    if (stats_d) hipFree(stats_d);
    if (stats16_d) hipFree(stats16_d);
    if (lvl_d) hipFree(lvl_d);
    *out = ix;
    return PCR_OK;

fail:
    release(B);
    if (cursor_d) hipFree(cursor_d);
    if (sort_tmp_d) hipFree(sort_tmp_d);
    if (stats_d) hipFree(stats_d);
    if (stats16_d) hipFree(stats16_d);
    if (lvl_d) hipFree(lvl_d);
    pcr_index_free(ix);
    return rc;
#undef BUILD_HIP
#undef BUILD_RC
}

int pcr_index_get_info(const pcr_index *ix, pcr_index_info *info) {
    PCR_REQUIRE(ix && info, "pcr_index_get_info: NULL argument");
    *info = ix->info;
    return PCR_OK;
}

}  // extern "C"
