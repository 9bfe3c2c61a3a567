#include "search_plugin.h"
#include <sstream>
#include <stdexcept>
#include "device_session.h"
#include "hive_standin.h"

using namespace hiveSearch;

CBaseSearch::CBaseSearch()
    : m_pData(NULL), m_NumInput(0), m_Dimension(0), m_IsOutputResultRequested(false), m_IsAccelerationStructureBuilt(false) {}

CBaseSearch::~CBaseSearch() {}

void CBaseSearch::setInputData(const double *vInputPoint, unsigned int vNumInputPoint, unsigned int vDimension) {
    m_pData = vInputPoint;
    m_NumInput = vNumInputPoint;
    m_Dimension = vDimension;
    m_IsAccelerationStructureBuilt = false;
}

std::vector<unsigned int> *CBaseSearch::executeSearch(unsigned int vNumNearestNeighbor, const double *vTargetPoint,
                                                      unsigned int vNumTargetPoint, bool vMultiThreaded) {
    // the reference _ASSERTs these (debug only) and would then index NULL; a release build gets NULL back
    if (vNumNearestNeighbor == 0 || vNumTargetPoint == 0 || !vTargetPoint) return NULL;
    std::vector<unsigned int> *pResult = NULL;
    try {
        if (!m_IsAccelerationStructureBuilt) {
            hiveEventLogger::hiveOutputEvent("Start building acceleration data structure...");
            m_IsAccelerationStructureBuilt = _buildAccelerationStructureV();
            if (m_IsAccelerationStructureBuilt)
                hiveEventLogger::hiveOutputEvent("Finish building acceleration data structure.");
            else
                hiveEventLogger::hiveOutputWarning(__EXCEPTION_SITE__, "Fail to build the acceleration data structure for search.");
        }
        if (m_IsAccelerationStructureBuilt) {
            pResult = new std::vector<unsigned int>;
            pResult->reserve((size_t)vNumTargetPoint * vNumNearestNeighbor);
            _executeSearchV(vNumNearestNeighbor, vTargetPoint, vNumTargetPoint, vMultiThreaded, *pResult);
            if (m_IsOutputResultRequested)
                hiveEventLogger::hiveOutputEvent(__describeResult(vNumNearestNeighbor, vTargetPoint, vNumTargetPoint, *pResult));
        }
        return pResult;
    } catch (...) {
        delete pResult;
        return NULL;
    }
}

double CBaseSearch::_computeDisV(const double *vData1, const double *vData2) const {
    double Sum = 0;
    for (unsigned int i = 0; i < m_Dimension; ++i) Sum += (vData1[i] - vData2[i]) * (vData1[i] - vData2[i]);
    return Sum;
}

std::string CBaseSearch::__describeResult(unsigned int vNumNearestNeighbor, const double *vTargetPoint,
                                          unsigned int vNumTargetPoint, const std::vector<unsigned int> &vResult) const {
    std::ostringstream Out;
    Out << "Input of the search:\n  Number of input data: " << m_NumInput << "\n  Dimension of input data: " << m_Dimension
        << "\nOutput of the search:\n";
    for (unsigned int i = 0; i < vNumTargetPoint; ++i) {
        const double *pQuery = vTargetPoint + (size_t)i * m_Dimension;
        Out << "  Target point:";
        for (unsigned int d = 0; d < m_Dimension; ++d) Out << ' ' << pQuery[d];
        Out << '\n';
        for (unsigned int k = 0; k < vNumNearestNeighbor; ++k) {
            const double *pHit = getInputDataAt(vResult.at((size_t)i * vNumNearestNeighbor + k));
            if (!pHit) continue;
            Out << "    " << _computeDisV(pQuery, pHit) << " :";
            for (unsigned int d = 0; d < m_Dimension; ++d) Out << ' ' << pHit[d];
            Out << '\n';
        }
    }
    return Out.str();
}

// ---------------------------------------------------------------------------------------------------------

CGridSearchGPU::~CGridSearchGPU() {
    if (getInputDataAt(0)) pcrhost::CDeviceSession::getInstance()->release(getInputDataAt(0));
}

bool CGridSearchGPU::_buildAccelerationStructureV() {
    if (getDimensionOfInputData() != 3 || getNumInputData() == 0 || !getInputDataAt(0)) return false;
    pcrhost::CDeviceSession *pSession = pcrhost::CDeviceSession::getInstance();
    pSession->release(getInputDataAt(0));  // setInputData invalidated whatever was built on this address
    if (!pSession->acquire(getInputDataAt(0), getNumInputData())) {
        hiveEventLogger::hiveOutputWarning(__EXCEPTION_SITE__, pSession->lastError());
        return false;
    }
    return true;
}

void CGridSearchGPU::_executeSearchV(unsigned int vNumNearestNeighbor, const double *vTargetPoint, unsigned int vNumTargetPoint,
                                     bool /*vMultiThreaded: the device search is parallel either way*/,
                                     std::vector<unsigned int> &voResult) const {
    pcrhost::CDeviceSession *pSession = pcrhost::CDeviceSession::getInstance();
    const pcrhost::SDeviceCloud *pCloud = pSession->acquire(getInputDataAt(0), getNumInputData());
    if (!pCloud) throw std::runtime_error(pSession->lastError());
    voResult.resize((size_t)vNumTargetPoint * vNumNearestNeighbor);
    const int Status = pcr_knn(pSession->context(), pCloud->pIndex, vTargetPoint, vNumTargetPoint, vNumNearestNeighbor,
                               m_Order == FARTHEST_FIRST ? PCR_ORDER_DESCENDING : PCR_ORDER_ASCENDING, -1.0, voResult.data(), NULL);
    if (Status != PCR_OK) {
        hiveEventLogger::hiveOutputWarning(__EXCEPTION_SITE__, std::string(pcr_strerror(Status)) + " -- " + pcr_last_error());
        throw std::runtime_error("pcr_knn failed");  // executeSearch turns this into NULL
    }
}
