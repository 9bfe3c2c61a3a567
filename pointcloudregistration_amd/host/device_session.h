// One GPU context per process (per rank), and one device cloud + grid index per host CPointCloud, shared by
// every plugin that needs it: the correspondence stage, the rejection stage and CUniqueData all query the
// same index where the reference builds a KD-tree each (CorrespondenceEstimationClosestPoints.cpp:123,
// :77-78 per iteration, RejectionCentroidDist.cpp:49-59, UniqueData.cpp:33-34).
// Keyed by (pointer, point count, data pointer): the reference identifies clouds by address only
// (CorrespondenceEstimationClosestPoints.cpp:45), which goes stale when a cloud is refilled in place.
#pragma once
#include <map>
#include <stdexcept>
#include <string>
#include "pcr.h"
#include "point_cloud.h"

namespace pcrhost {

struct SDeviceCloud {
    pcr_cloud *pCloud;
    pcr_index *pIndex;
    unsigned int NumPoints;
    const double *pHostPos;
    mutable bool NormalsAttached;  // the cloud's normals are on the device (the opt-in normal-aware correspondences)
    SDeviceCloud() : pCloud(NULL), pIndex(NULL), NumPoints(0), pHostPos(NULL), NormalsAttached(false) {}
};

class CDeviceSession {
public:
    static CDeviceSession *getInstance();

    // NULL when no usable device/context exists; lastError() says why.  Never throws.
    pcr_ctx *context();
    // upload + index build on first use; NULL on failure
    const SDeviceCloud *acquire(const hivePointCloud::CPointCloud *vPointCloud);
    // raw-array variant for hiveSearch::CBaseSearch::setInputData, which sees no CPointCloud
    const SDeviceCloud *acquire(const double *vPos, unsigned int vNumPoints);
    void release(const void *vKey);
    void releaseAll();
    void setDevice(int vDevice) { m_Device = vDevice; }
    // pcr_index_params.compact_tables of every index built from now on (0: the library's default, dense cell tables)
    void setCompactTables(unsigned int vMode) { m_CompactTables = vMode; }
    const std::string &lastError() const { return m_LastError; }
    // sample sharding for the multi-GPU loop: this process handles samples
    // [begin, begin+count) of every sample set; a hook sums the 17 partials over the ranks.
    typedef int (*AllReduceFn)(double *vioPartials17, void *vUser);
    void setSharding(unsigned int vRank, unsigned int vWorld, AllReduceFn vAllReduce, void *vUser) {
        m_Rank = vRank;
        m_World = vWorld ? vWorld : 1;
        m_AllReduce = vAllReduce;
        m_AllReduceUser = vUser;
    }
    unsigned int rank() const { return m_Rank; }
    unsigned int world() const { return m_World; }
    int allReduce(double *vioPartials17) const { return (m_World > 1 && m_AllReduce) ? m_AllReduce(vioPartials17, m_AllReduceUser) : 0; }

private:
    CDeviceSession() : m_pContext(NULL), m_Device(0), m_CompactTables(0), m_ContextTried(false), m_Rank(0), m_World(1), m_AllReduce(NULL), m_AllReduceUser(NULL) {}
    const SDeviceCloud *__acquire(const void *vKey, const double *vPos, unsigned int vNumPoints);

    pcr_ctx *m_pContext;
    int m_Device;
    unsigned int m_CompactTables;
    bool m_ContextTried;
    std::map<const void *, SDeviceCloud> m_Clouds;
    std::string m_LastError;
    unsigned int m_Rank, m_World;
    AllReduceFn m_AllReduce;
    void *m_AllReduceUser;
};

}  // namespace pcrhost
