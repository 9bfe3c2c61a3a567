// ORACLE (test infrastructure only -- see pcr_oracle.h).
//
// CPU restatement of hiveCommon::CKNNSearch -- the KD-tree the reference's ICP loop
// actually uses.  Written from the text of
//   /root/reference/Common/KNNSearch.cpp  and  /root/reference/Common/KNNSearch.h
// and kept step-for-step equivalent where it decides a result: split rule, 3-way
// partition, N/2 clamp, leaf size 10, near-child-first descent, "<=" far-child test,
// "first visited wins" on equal distances.  Storage differs on purpose (flat node
// array, no per-query heap allocation); none of that changes an index.
#include <algorithm>
#include <cfloat>
#include <cstring>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "pcr_oracle.h"

namespace {

struct Box3 {
    double lo[3], hi[3];
};

struct KdNode {
    int first, last;      // leaf: range [first,last) into ind/reorder   (KNNSearch.cpp:217-218)
    int axis;             // inner: split axis                            (:245)
    double divlow, divhigh;  // inner: tightened child faces on axis       (:255-256)
    int left, right;      // -1/-1 for a leaf
};

struct OldTree {
    std::vector<double> pts;  // copy of the caller's points (reference borrows, :26)
    int n = 0;
    unsigned k = 0;
    std::vector<int> ind;        // m_IndexOfData
    std::vector<double> reorder; // m_pReorderData (:84-91)
    std::vector<KdNode> nodes;
    int root = -1;
    Box3 rootbox;

    static constexpr int kLeafMax = 10;  // m_LeafMaxSize, KNNSearch.cpp:5

    double coord(int point, int ax) const { return pts[(size_t)point * 3 + ax]; }

    // __computeMinMax, KNNSearch.cpp:120-131
    void minmax(const int *idx, int cnt, int ax, double &mn, double &mx) const {
        mn = mx = coord(idx[0], ax);
        for (int i = 1; i < cnt; ++i) {
            double v = coord(idx[i], ax);
            if (v < mn) mn = v;
            if (v > mx) mx = v;
        }
    }

    // __splitPlane, KNNSearch.cpp:135-157: two Hoare passes -> [<cv | ==cv | >cv]
    void split_plane(int cnt, int ax, double cv, int *idx, int &lim1, int &lim2) const {
        int l = 0, r = cnt - 1;
        for (;;) {
            while (l <= r && coord(idx[l], ax) < cv) ++l;
            while (l <= r && coord(idx[r], ax) >= cv) --r;
            if (l > r) break;
            std::swap(idx[l], idx[r]);
            ++l;
            --r;
        }
        lim1 = l;
        r = cnt - 1;
        for (;;) {
            while (l <= r && coord(idx[l], ax) <= cv) ++l;
            while (l <= r && coord(idx[r], ax) > cv) --r;
            if (l > r) break;
            std::swap(idx[l], idx[r]);
            ++l;
            --r;
        }
        lim2 = l;
    }

    // __split, KNNSearch.cpp:161-206
    void split(const Box3 &bb, int *idx, int cnt, int &off, int &ax, double &cv) const {
        double span = bb.hi[0] - bb.lo[0];
        ax = 0;
        for (int d = 1; d < 3; ++d) {  // widest bbox side, first max wins (:166-175)
            double s = bb.hi[d] - bb.lo[d];
            if (s > span) {
                span = s;
                ax = d;
            }
        }
        double mn, mx;
        minmax(idx, cnt, ax, mn, mx);  // re-measure on the actual points (:177-180)
        cv = (mn + mx) / 2;
        span = mx - mn;
        const int first_ax = ax;
        for (int d = 0; d < 3; ++d) {  // another axis may really be wider (:182-198)
            if (d == first_ax) continue;
            double s = bb.hi[d] - bb.lo[d];
            if (s > span) {
                minmax(idx, cnt, d, mn, mx);
                s = mx - mn;
                if (s > span) {
                    span = s;
                    ax = d;
                    cv = (mn + mx) / 2;
                }
            }
        }
        int lim1, lim2;
        split_plane(cnt, ax, cv, idx, lim1, lim2);
        if (lim1 > cnt / 2)  // (:203-205)
            off = lim1;
        else if (lim2 < cnt / 2)
            off = lim2;
        else
            off = cnt / 2;
    }

    // __divideTree, KNNSearch.cpp:210-265.  bb is tightened on return.
    int divide(int L, int R, Box3 &bb) {
        int id = (int)nodes.size();
        nodes.push_back(KdNode{0, 0, 0, 0.0, 0.0, -1, -1});
        if (R - L <= kLeafMax) {
            nodes[id].first = L;
            nodes[id].last = R;
            for (int d = 0; d < 3; ++d) bb.lo[d] = bb.hi[d] = coord(ind[L], d);
            for (int j = L + 1; j < R; ++j)
                for (int d = 0; d < 3; ++d) {
                    double v = coord(ind[j], d);
                    if (bb.lo[d] > v) bb.lo[d] = v;
                    if (bb.hi[d] < v) bb.hi[d] = v;
                }
            return id;
        }
        int off, ax;
        double cv;
        split(bb, &ind[L], R - L, off, ax, cv);
        Box3 lb = bb;
        lb.hi[ax] = cv;
        int lc = divide(L, L + off, lb);
        Box3 rb = bb;
        rb.lo[ax] = cv;
        int rc = divide(L + off, R, rb);
        KdNode &nd = nodes[id];
        nd.axis = ax;
        nd.left = lc;
        nd.right = rc;
        nd.divlow = lb.hi[ax];
        nd.divhigh = rb.lo[ax];
        for (int d = 0; d < 3; ++d) {
            bb.lo[d] = std::min(lb.lo[d], rb.lo[d]);
            bb.hi[d] = std::max(lb.hi[d], rb.hi[d]);
        }
        return id;
    }

    void build() {
        ind.resize(n);
        for (int i = 0; i < n; ++i) ind[i] = i;
        // __computeBoundingBox, KNNSearch.cpp:96-116
        for (int d = 0; d < 3; ++d) rootbox.lo[d] = rootbox.hi[d] = coord(0, d);
        for (int i = 1; i < n; ++i)
            for (int d = 0; d < 3; ++d) {
                double v = coord(i, d);
                if (v < rootbox.lo[d]) rootbox.lo[d] = v;
                if (v > rootbox.hi[d]) rootbox.hi[d] = v;
            }
        nodes.reserve((size_t)n / 3 + 16);
        // __divideTree gets m_RootBbox BY REFERENCE and tightens it (:82, :258-262).  For the
        // root that is a no-op in value (the tight box of all points is the root box).
        root = divide(0, n, rootbox);
        reorder.resize((size_t)n * 3);
        for (int i = 0; i < n; ++i)
            for (int d = 0; d < 3; ++d) reorder[(size_t)i * 3 + d] = coord(ind[i], d);
    }
};

// CKNNSimpleResultSet, KNNSearch.h:44-117
struct ResultSet {
    unsigned cap, count;
    double worst;
    double *dist;
    unsigned *index;
    void reset() {
        for (unsigned i = 0; i < cap; ++i) {
            dist[i] = DBL_MAX;
            index[i] = ORC_NO_INDEX;
        }
        worst = DBL_MAX;
        count = 0;
    }
    void add(double d, unsigned idx) {
        if (d >= worst) return;  // equal distance never displaces (KNNSearch.h:75)
        if (count < cap) ++count;
        unsigned i = count - 1;
        for (; i > 0; --i) {
            if (dist[i - 1] > d) {  // strict: stable among equals (KNNSearch.h:80)
                dist[i] = dist[i - 1];
                index[i] = index[i - 1];
            } else
                break;
        }
        dist[i] = d;
        index[i] = idx;
        worst = dist[cap - 1];
    }
};

// __searchLevel, KNNSearch.cpp:311-360
void search(const OldTree &t, int node, const double q[3], double mind, double dists[3], ResultSet &rs) {
    const KdNode &nd = t.nodes[node];
    if (nd.left < 0) {
        const double worst_at_entry = rs.worst;  // (:315) -- add() re-checks the live value
        for (int j = nd.first; j < nd.last; ++j) {
            const double *x = &t.reorder[(size_t)j * 3];
            double d = 0, diff;  // SComputeDistance, KNNSearch.h:137-143
            diff = q[0] - x[0];
            d += diff * diff;
            diff = q[1] - x[1];
            d += diff * diff;
            diff = q[2] - x[2];
            d += diff * diff;
            if (d < worst_at_entry) rs.add(d, (unsigned)t.ind[j]);
        }
        return;
    }
    const int ax = nd.axis;
    const double v = q[ax];
    const double d1 = v - nd.divlow, d2 = v - nd.divhigh;
    int near, far;
    double cut;
    if (d1 + d2 < 0) {
        near = nd.left;
        far = nd.right;
        cut = (v - nd.divhigh) * (v - nd.divhigh);
    } else {
        near = nd.right;
        far = nd.left;
        cut = (v - nd.divlow) * (v - nd.divlow);
    }
    search(t, near, q, mind, dists, rs);
    const double save = dists[ax];
    mind = mind + cut - save;
    dists[ax] = cut;
    if (mind <= rs.worst) search(t, far, q, mind, dists, rs);  // note <= (:355)
    dists[ax] = save;
}

void query_one(const OldTree &t, const double q[3], unsigned *oi, double *od) {
    const unsigned K = t.k;
    double dbuf[64];
    unsigned ibuf[64];
    std::vector<double> dheap;
    std::vector<unsigned> iheap;
    ResultSet rs;
    rs.cap = K;
    if (K <= 64) {
        rs.dist = dbuf;
        rs.index = ibuf;
    } else {
        dheap.resize(K);
        iheap.resize(K);
        rs.dist = dheap.data();
        rs.index = iheap.data();
    }
    rs.reset();
    // __computeInitialDistances, KNNSearch.cpp:289-307
    double dists[3] = {0, 0, 0}, mind = 0;
    for (int d = 0; d < 3; ++d) {
        if (q[d] < t.rootbox.lo[d]) {
            dists[d] = (q[d] - t.rootbox.lo[d]) * (q[d] - t.rootbox.lo[d]);
            mind += dists[d];
        }
        if (q[d] > t.rootbox.hi[d]) {
            dists[d] = (q[d] - t.rootbox.hi[d]) * (q[d] - t.rootbox.hi[d]);
            mind += dists[d];
        }
    }
    search(t, t.root, q, mind, dists, rs);
    // copy(): first min(count,K) entries; the rest of the caller's vector is left as it was
    // in the reference -- the oracle reports ORC_NO_INDEX there.
    for (unsigned i = 0; i < K; ++i) {
        oi[i] = rs.index[i];
        if (od) od[i] = rs.dist[i];
    }
}

}  // namespace

extern "C" {

void *orc_kd_old_build(const double *xyz, uint32_t n, uint32_t k) {
    if (!xyz || n == 0 || k == 0) return nullptr;
    OldTree *t = new OldTree;
    t->pts.assign(xyz, xyz + (size_t)n * 3);
    t->n = (int)n;
    t->k = k;
    t->build();
    return t;
}

void orc_kd_old_query(const void *tree, const double *q, uint32_t nq, uint32_t *out_idx, double *out_d2,
                      int threads) {
    const OldTree &t = *static_cast<const OldTree *>(tree);
    const long K = t.k;
    (void)threads;
#pragma omp parallel for schedule(dynamic, 1024) num_threads(threads > 1 ? threads : 1)
    for (long i = 0; i < (long)nq; ++i)
        query_one(t, q + i * 3, out_idx + i * K, out_d2 ? out_d2 + i * K : nullptr);
}

void orc_kd_old_free(void *tree) { delete static_cast<OldTree *>(tree); }

// internal hooks for the other oracle translation units
const double *orc_kd_old_points(const void *tree) { return static_cast<const OldTree *>(tree)->pts.data(); }
uint32_t orc_kd_old_k(const void *tree) { return static_cast<const OldTree *>(tree)->k; }
void orc_kd_old_query_one(const void *tree, const double *q, uint32_t *oi, double *od) {
    query_one(*static_cast<const OldTree *>(tree), q, oi, od);
}

void orc_knn_brute(const double *xyz, uint32_t n, const double *q, uint32_t nq, uint32_t k, uint32_t *out_idx,
                   double *out_d2) {
#pragma omp parallel for schedule(static)
    for (long qi = 0; qi < (long)nq; ++qi) {
        std::vector<std::pair<double, uint32_t>> best;
        best.reserve(k + 1);
        const double *p = q + qi * 3;
        for (uint32_t j = 0; j < n; ++j) {
            const double *x = xyz + (size_t)j * 3;
            double d = 0, diff;
            diff = p[0] - x[0];
            d += diff * diff;
            diff = p[1] - x[1];
            d += diff * diff;
            diff = p[2] - x[2];
            d += diff * diff;
            std::pair<double, uint32_t> cand(d, j);
            if (best.size() < k || cand < best.back()) {
                best.insert(std::upper_bound(best.begin(), best.end(), cand), cand);
                if (best.size() > k) best.pop_back();
            }
        }
        for (uint32_t i = 0; i < k; ++i) {
            out_idx[qi * k + i] = i < best.size() ? best[i].second : ORC_NO_INDEX;
            if (out_d2) out_d2[qi * k + i] = i < best.size() ? best[i].first : DBL_MAX;
        }
    }
}
}
