#include "device_session.h"

#include <cstring>

using namespace pcrhost;

CDeviceSession *CDeviceSession::getInstance() {
    static CDeviceSession s_Session;
    return &s_Session;
}

pcr_ctx *CDeviceSession::context() {
    if (!m_pContext && !m_ContextTried) {
        m_ContextTried = true;
        const int Status = pcr_ctx_create(m_Device, &m_pContext);
        if (Status != PCR_OK) {
            m_pContext = NULL;
            m_LastError = std::string("pcr_ctx_create: ") + pcr_strerror(Status) + " -- " + pcr_last_error();
        }
    }
    return m_pContext;
}

const SDeviceCloud *CDeviceSession::acquire(const hivePointCloud::CPointCloud *vPointCloud) {
    if (!vPointCloud) return NULL;
    return __acquire(vPointCloud, vPointCloud->getPosPointer(), vPointCloud->getNumPoints());
}

const SDeviceCloud *CDeviceSession::acquire(const double *vPos, unsigned int vNumPoints) { return __acquire(vPos, vPos, vNumPoints); }

const SDeviceCloud *CDeviceSession::__acquire(const void *vKey, const double *vPos, unsigned int vNumPoints) {
    std::map<const void *, SDeviceCloud>::iterator Itr = m_Clouds.find(vKey);
    if (Itr != m_Clouds.end()) {
        if (Itr->second.NumPoints == vNumPoints && Itr->second.pHostPos == vPos) return &Itr->second;
        release(vKey);  // same address, different content: rebuild
    }
    pcr_ctx *pContext = context();
    if (!pContext) return NULL;
    if (!vPos || vNumPoints == 0) {
        m_LastError = "empty point cloud";
        return NULL;
    }
    SDeviceCloud Entry;
    int Status = pcr_cloud_upload(pContext, vPos, vNumPoints, &Entry.pCloud);
    pcr_index_params Params;
    memset(&Params, 0, sizeof Params);  // all defaults ...
    Params.compact_tables = m_CompactTables;  // ... but what the host asked for here
    if (Status == PCR_OK) Status = pcr_index_build(pContext, Entry.pCloud, &Params, &Entry.pIndex);
    if (Status != PCR_OK) {
        m_LastError = std::string(pcr_strerror(Status)) + " -- " + pcr_last_error();
        if (Entry.pCloud) pcr_cloud_free(Entry.pCloud);
        return NULL;
    }
    Entry.NumPoints = vNumPoints;
    Entry.pHostPos = vPos;
    return &(m_Clouds[vKey] = Entry);
}

void CDeviceSession::release(const void *vKey) {
    std::map<const void *, SDeviceCloud>::iterator Itr = m_Clouds.find(vKey);
    if (Itr == m_Clouds.end()) return;
    if (Itr->second.pIndex) pcr_index_free(Itr->second.pIndex);
    if (Itr->second.pCloud) pcr_cloud_free(Itr->second.pCloud);
    m_Clouds.erase(Itr);
}

void CDeviceSession::releaseAll() {
    while (!m_Clouds.empty()) release(m_Clouds.begin()->first);
    if (m_pContext) pcr_ctx_destroy(m_pContext);
    m_pContext = NULL;
    m_ContextTried = false;
}
