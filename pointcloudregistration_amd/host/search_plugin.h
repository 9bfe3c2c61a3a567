// hiveSearch::CBaseSearch (PointCloudRegistration/BaseSearch.h:6-47, .cpp:27-127) and its GPU subclass.
//
// Contract kept from the reference:
//   setInputData borrows the array and invalidates the acceleration structure          (.cpp:27-33)
//   executeSearch builds lazily, returns a heap vector [nq*K] OWNED BY THE CALLER, and returns NULL on any
//   failure -- it never throws                                                           (.cpp:37-60)
//   per query the reference subclass (hiveSearch::CKNNSearch) emits neighbours FARTHEST FIRST
//   (DistanceQueue.cpp:37-46); queries are flattened row-major                          (NewKNNSearch.cpp:58-68)
//   _computeDisV is the squared Euclidean distance over `Dimension` coordinates         (.cpp:119-127)
#pragma once
#include <string>
#include <vector>

namespace hiveSearch {

class CBaseSearch {
public:
    CBaseSearch();
    virtual ~CBaseSearch();

    void setInputData(const double *vInputPoint, unsigned int vNumInputPoint, unsigned int vDimension);
    void enableOutputSearchResult() { m_IsOutputResultRequested = true; }
    void disableOutputSearchResult() { m_IsOutputResultRequested = false; }

    std::vector<unsigned int> *executeSearch(unsigned int vNumNearestNeighbor, const double *vTargetPoint,
                                             unsigned int vNumTargetPoint, bool vMultiThreaded = false);

    const double *getInputDataAt(unsigned int vIndex) const { return vIndex < m_NumInput ? m_pData + (size_t)vIndex * m_Dimension : NULL; }
    unsigned int getDimensionOfInputData() const { return m_Dimension; }
    unsigned int getNumInputData() const { return m_NumInput; }

protected:
    virtual bool _buildAccelerationStructureV() { return true; }
    virtual void _executeSearchV(unsigned int, const double *, unsigned int, bool, std::vector<unsigned int> &) const {}
    virtual double _computeDisV(const double *vData1, const double *vData2) const;

private:
    std::string __describeResult(unsigned int vNumNearestNeighbor, const double *vTargetPoint, unsigned int vNumTargetPoint,
                                 const std::vector<unsigned int> &vResult) const;

    const double *m_pData;
    unsigned int m_NumInput, m_Dimension;
    bool m_IsOutputResultRequested;
    bool m_IsAccelerationStructureBuilt;
};

// Exact K-NN on the MI355X grid index behind the CBaseSearch interface.  Replaces hiveSearch::CKNNSearch
// (NewKNNSearch.h:9-60): same results whenever no two candidates are exactly equidistant from a query (ties:
// lowest index here, KD-tree visit order there).  Dimension must be 3; any other dimension, a missing GPU or
// K > number of points makes executeSearch return NULL, the reference's failure convention.
class CGridSearchGPU : public CBaseSearch {
public:
    enum ENeighborOrder { FARTHEST_FIRST = 0 /* reference plugin order */, NEAREST_FIRST = 1 /* hiveCommon::CKNNSearch order */ };

    CGridSearchGPU() : m_Order(FARTHEST_FIRST) {}
    virtual ~CGridSearchGPU();
    void setNeighborOrder(ENeighborOrder vOrder) { m_Order = vOrder; }

protected:
    virtual bool _buildAccelerationStructureV();
    virtual void _executeSearchV(unsigned int vNumNearestNeighbor, const double *vTargetPoint, unsigned int vNumTargetPoint,
                                 bool vMultiThreaded, std::vector<unsigned int> &voResult) const;

private:
    ENeighborOrder m_Order;
};

}  // namespace hiveSearch
