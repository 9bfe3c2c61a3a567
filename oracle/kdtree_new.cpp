// ORACLE (test infrastructure only -- see pcr_oracle.h).
//
// CPU restatement of hiveSearch::CKNNSearch behind hiveSearch::CBaseSearch -- the
// batched search plugin.  Written from the text of
//   /root/reference/PointCloudRegistration/NewKNNSearch.{h,cpp}
//   /root/reference/PointCloudRegistration/DistanceQueue.{h,cpp}
//   /root/reference/PointCloudRegistration/BaseSearch.cpp:119-127 (_computeDisV)
// Differences from the old tree that matter: split axis = largest ACTUAL span, one
// Hoare pass, no N/2 clamp (NewKNNSearch.cpp:152-201); max-heap result queue dumped
// farthest first (DistanceQueue.cpp:37-46); any dimension.
#include <algorithm>
#include <queue>
#include <stdexcept>
#include <vector>
#include "pcr_oracle.h"

namespace {

struct HeapItem {
    double d;
    unsigned idx;
    // same ordering as SDistance2Point::operator< (DistanceQueue.h)
    friend bool operator<(const HeapItem &a, const HeapItem &b) { return a.d < b.d; }
};
typedef std::priority_queue<HeapItem, std::vector<HeapItem>, std::less<HeapItem>> MaxHeap;

struct NewNode {
    int axis, first, last;
    double lower, upper;  // SplitLowerBound / SplitUpperBound
    int left, right;
};

struct NewTree {
    std::vector<double> data;
    unsigned n = 0, dim = 0;
    int leaf_max = 10;  // m_MaxNumDataInLeafNode, NewKNNSearch.cpp:10
    std::vector<unsigned> ind;
    std::vector<double> reorder;
    std::vector<NewNode> nodes;
    std::vector<double> root_lo, root_hi;
    int root = -1;
    int depth_guard = 0;

    const double *at(unsigned i) const { return &data[(size_t)i * dim]; }

    // __calculateBoundingBox, NewKNNSearch.cpp:126-148
    void bbox(unsigned s, unsigned e, std::vector<double> &lo, std::vector<double> &hi) const {
        lo.assign(at(ind[s]), at(ind[s]) + dim);
        hi = lo;
        for (unsigned i = s + 1; i < e; ++i)
            for (unsigned d = 0; d < dim; ++d) {
                double v = at(ind[i])[d];
                if (v < lo[d]) lo[d] = v;
                if (v > hi[d]) hi[d] = v;
            }
    }

    // __calculateSplitHyperPlaneInfo + __splitNodeBySwappingIndex, NewKNNSearch.cpp:152-201
    void split(int cnt, unsigned *idx, int &off, int &ax, double &mid) const {
        ax = 0;
        double best = -1;
        for (unsigned d = 0; d < dim; ++d) {
            double mn = at(idx[0])[d], mx = mn;
            for (int i = 1; i < cnt; ++i) {
                double v = at(idx[i])[d];
                if (v < mn) mn = v;
                if (v > mx) mx = v;
            }
            double span = mx - mn;
            if (d == 0 || span > best) {  // strict >, first max wins (:192)
                best = span;
                ax = (int)d;
                mid = (mn + mx) / 2;
            }
        }
        int l = 0, r = cnt - 1;
        for (;;) {
            while (l <= r && at(idx[l])[ax] < mid) ++l;
            while (l <= r && at(idx[r])[ax] >= mid) --r;
            if (l > r) break;
            std::swap(idx[l], idx[r]);
            ++l;
            --r;
        }
        off = l;
    }

    // __splitData2TreeNode, NewKNNSearch.cpp:205-241
    int divide(int L, int R, std::vector<double> &lo, std::vector<double> &hi, int depth) {
        // The reference recurses forever on > leaf_max coincident points (left child empty,
        // SURVEY Appendix B).  The oracle reports that instead of overflowing the stack.
        if (R == L || depth > 4096) throw std::runtime_error("degenerate split (reference would not terminate)");
        int id = (int)nodes.size();
        nodes.push_back(NewNode{0, 0, 0, 0.0, 0.0, -1, -1});
        if (R - L <= leaf_max) {
            nodes[id].first = L;
            nodes[id].last = R;
            bbox(L, R, lo, hi);
            return id;
        }
        int off, ax;
        double mid = 0;
        split(R - L, &ind[L], off, ax, mid);
        std::vector<double> llo = lo, lhi = hi, rlo = lo, rhi = hi;
        lhi[ax] = mid;
        int lc = divide(L, L + off, llo, lhi, depth + 1);
        rlo[ax] = mid;
        int rc = divide(L + off, R, rlo, rhi, depth + 1);
        NewNode &nd = nodes[id];
        nd.axis = ax;
        nd.left = lc;
        nd.right = rc;
        nd.lower = lhi[ax];
        nd.upper = rlo[ax];
        for (unsigned d = 0; d < dim; ++d) {
            lo[d] = std::min(llo[d], rlo[d]);
            hi[d] = std::max(lhi[d], rhi[d]);
        }
        return id;
    }

    void build() {
        ind.resize(n);
        for (unsigned i = 0; i < n; ++i) ind[i] = i;
        bbox(0, n, root_lo, root_hi);
        root = divide(0, (int)n, root_lo, root_hi, 0);
        reorder.resize((size_t)n * dim);
        for (unsigned i = 0; i < n; ++i)
            for (unsigned d = 0; d < dim; ++d) reorder[(size_t)i * dim + d] = at(ind[i])[d];
    }

    // _computeDisV, BaseSearch.cpp:119-127
    double dist(const double *a, const double *b) const {
        double r = 0;
        for (unsigned d = 0; d < dim; ++d) r += (a[d] - b[d]) * (a[d] - b[d]);
        return r;
    }

    // __searchNeighborInTreeNode, NewKNNSearch.cpp:266-304 ; addElement2Queue, DistanceQueue.cpp:15-25
    void search(int node, const double *q, unsigned k, double mind, std::vector<double> &dists, MaxHeap &heap) const {
        const NewNode &nd = nodes[node];
        if (nd.left < 0) {
            for (int j = nd.first; j < nd.last; ++j) {
                double d = dist(q, &reorder[(size_t)j * dim]);
                if (heap.size() < k)
                    heap.push(HeapItem{d, ind[j]});
                else if (d < heap.top().d) {
                    heap.pop();
                    heap.push(HeapItem{d, ind[j]});
                }
            }
            return;
        }
        const int ax = nd.axis;
        int near, far;
        double cut;
        if ((q[ax] - nd.lower) + (q[ax] - nd.upper) < 0) {
            near = nd.left;
            far = nd.right;
            cut = (q[ax] - nd.upper) * (q[ax] - nd.upper);  // std::pow(x,2), NewKNNSearch.h:59
        } else {
            near = nd.right;
            far = nd.left;
            cut = (q[ax] - nd.lower) * (q[ax] - nd.lower);
        }
        search(near, q, k, mind, dists, heap);
        const double save = dists[ax];
        mind = mind + cut - save;
        dists[ax] = cut;
        if (heap.size() < k || mind <= heap.top().d) search(far, q, k, mind, dists, heap);  // (:299)
        dists[ax] = save;
    }
};

}  // namespace

extern "C" {

void *orc_kd_new_build(const double *data, uint32_t n, uint32_t dim, int leaf_max) {
    if (!data || n == 0 || dim == 0) return nullptr;
    NewTree *t = new NewTree;
    t->data.assign(data, data + (size_t)n * dim);
    t->n = n;
    t->dim = dim;
    if (leaf_max > 0) t->leaf_max = leaf_max;
    try {
        t->build();
    } catch (...) {
        delete t;
        return nullptr;  // CBaseSearch::executeSearch would return NULL (BaseSearch.cpp:55-59)
    }
    return t;
}

void orc_kd_new_query(const void *tree, uint32_t k, const double *q, uint32_t nq, uint32_t *out_idx) {
    const NewTree &t = *static_cast<const NewTree *>(tree);
    for (uint32_t i = 0; i < nq; ++i) {
        const double *p = q + (size_t)i * t.dim;
        MaxHeap heap;
        std::vector<double> dists(t.dim, 0.0);
        double mind = 0;  // __calculateInitialDistances, NewKNNSearch.cpp:245-262
        for (unsigned d = 0; d < t.dim; ++d) {
            if (p[d] < t.root_lo[d]) {
                dists[d] = (p[d] - t.root_lo[d]) * (p[d] - t.root_lo[d]);
                mind += dists[d];
            }
            if (p[d] > t.root_hi[d]) {
                dists[d] = (p[d] - t.root_hi[d]) * (p[d] - t.root_hi[d]);
                mind += dists[d];
            }
        }
        t.search(t.root, p, k, mind, dists, heap);
        uint32_t w = 0;  // dumpIndex: top first = farthest first
        while (!heap.empty() && w < k) {
            out_idx[(size_t)i * k + w++] = heap.top().idx;
            heap.pop();
        }
        for (; w < k; ++w) out_idx[(size_t)i * k + w] = ORC_NO_INDEX;
    }
}

void orc_kd_new_free(void *tree) { delete static_cast<NewTree *>(tree); }
}
