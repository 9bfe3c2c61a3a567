// ORACLE (test infrastructure only).
//
// Thin C wrapper around the REFERENCE's own hiveCommon::CKNNSearch, so Python tests can
// drive it.  This file contains no reference code: it includes the reference header from
// where it lies under /root/reference (oracle/Makefile passes -I) and calls its public
// API (initKNNSearch / executeKNN, Common/KNNSearch.h:16-19).  It is compiled only into
// oracle/_ref/libref_knn.so, together with /root/reference/Common/KNNSearch.cpp itself.
#include <cstdint>
#include <vector>
#include "KNNSearch.h"

namespace {
struct RefTree {
    std::vector<double> pts;  // the reference borrows the pointer (KNNSearch.cpp:26): keep it alive
    hiveCommon::CKNNSearch knn;
    unsigned k;
};
}  // namespace

extern "C" {

void *ref_knn_build(const double *xyz, uint32_t n, uint32_t k) {
    RefTree *t = new RefTree;
    t->pts.assign(xyz, xyz + (size_t)n * 3);
    t->k = k;
    t->knn.initKNNSearch(t->pts.data(), n, 3, k);
    return t;
}

void ref_knn_query(void *tree, const double *q, uint32_t nq, uint32_t *out_idx) {
    RefTree *t = static_cast<RefTree *>(tree);
    std::vector<unsigned int> nb;
    for (uint32_t i = 0; i < nq; ++i) {
        nb.assign(t->k, 0xFFFFFFFFu);
        t->knn.executeKNN(q + (size_t)i * 3, nb);
        for (unsigned j = 0; j < t->k; ++j) out_idx[(size_t)i * t->k + j] = nb[j];
    }
}

// The reference destructor frees only the root node (KNNSearch.cpp:12); the leak is the
// reference's, and irrelevant for a test process.
void ref_knn_free(void *tree) { delete static_cast<RefTree *>(tree); }
}
