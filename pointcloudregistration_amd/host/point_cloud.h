// hivePointCloud::CPointCloud and the point-subset containers, reduced to what the ICP path touches
// (PointCloudRegistration/PointCloud.h:12-85, PointCloudSubset.h:14-127).  OSG, PLY loading, curvature
// arrays and neighbour lists are outside SURVEY.md section 8 and are not restated.
#pragma once
#include <string>
#include <vector>
#include "reg_types.h"

namespace hivePointCloud {

class CPointCloud {
public:
    explicit CPointCloud(unsigned int vDimension = 3) : m_Dimension(vDimension) {}

    unsigned int getNumPoints() const { return (unsigned int)(m_Pos.size() / 3); }
    const double *getPosPointer() const { return m_Pos.empty() ? NULL : m_Pos.data(); }
    const double *getNormalPointer() const { return m_Normal.empty() ? NULL : m_Normal.data(); }

    // PointCloud.h:48,53 -- copy in (AoS x,y,z)
    void setPos(const double *vPos, unsigned int vNumPoint) { m_Pos.assign(vPos, vPos + (size_t)vNumPoint * 3); }
    void setNormal(const double *vNormal, unsigned int vNumPoint) { m_Normal.assign(vNormal, vNormal + (size_t)vNumPoint * 3); }

    // PointCloud.cpp:358-387: in place, x' = ((r0*x + r3*y) + r6*z) + t0 ... ; normals rotated only
    const CPointCloud &transform(const double vRot[9], const double vTra[3]) {
        const size_t N = getNumPoints();
        for (size_t i = 0; i < N; ++i) {
            double *p = &m_Pos[i * 3];
            const double x = p[0], y = p[1], z = p[2];
            p[0] = ((vRot[0] * x + vRot[3] * y) + vRot[6] * z) + vTra[0];
            p[1] = ((vRot[1] * x + vRot[4] * y) + vRot[7] * z) + vTra[1];
            p[2] = ((vRot[2] * x + vRot[5] * y) + vRot[8] * z) + vTra[2];
        }
        for (size_t i = 0; i < m_Normal.size() / 3; ++i) {
            double *n = &m_Normal[i * 3];
            const double x = n[0], y = n[1], z = n[2];
            n[0] = (vRot[0] * x + vRot[3] * y) + vRot[6] * z;
            n[1] = (vRot[1] * x + vRot[4] * y) + vRot[7] * z;
            n[2] = (vRot[2] * x + vRot[5] * y) + vRot[8] * z;
        }
        return *this;
    }

private:
    unsigned int m_Dimension;
    std::vector<double> m_Pos, m_Normal;
};

}  // namespace hivePointCloud

namespace hiveRegistration {

// PointCloudSubset.h:14-77.  One addition over the reference: getPointSet()/getNormalSet() call a virtual
// hook first, so that a subset whose points still live on the GPU (CDevicePairSubset, gpu_stages.h) can
// copy them back only when a host-side stage really reads them.  INTEGRATION.md lists this as the one edit
// to an existing reference header.
class CBasePointCloudSubset : public CICPBaseObject {
public:
    const VecColVector3 &getPointSet() const {
        _materializeV();
        return m_PointSet;
    }
    const VecColVector3 &getNormalSet() const {
        _materializeV();
        return m_NormalSet;
    }
    virtual unsigned int getNumPoint() const {
        _materializeV();
        return (unsigned int)m_PointSet.size();
    }

protected:
    CBasePointCloudSubset() {}
    virtual ~CBasePointCloudSubset() {}
    virtual void _materializeV() const {}

    mutable VecColVector3 m_PointSet;
    mutable VecColVector3 m_NormalSet;
};

// PointCloudSubset.h:81-98
class CPntNormSubset : public CBasePointCloudSubset {
public:
    CPntNormSubset() {}
    void setPointSet(const VecColVector3 &vPointSet) { m_PointSet = vPointSet; }
    void setNormalSet(const VecColVector3 &vNormalSet) { m_NormalSet = vNormalSet; }
    void swapPointSet(VecColVector3 &vPointSet) { m_PointSet.swap(vPointSet); }
    void swapNormalSet(VecColVector3 &vNormalSet) { m_NormalSet.swap(vNormalSet); }

protected:
    virtual ~CPntNormSubset() {}
};

// PointCloudSubset.h:102-127, .cpp:83-113: a list of indices into a cloud; points (and normals when the cloud
// has them) gathered as pos[idx*3 + {0,1,2}].  An empty index list stays empty (the reference's default
// argument), it does NOT mean "all points".
class CIndexSubset : public CBasePointCloudSubset {
public:
    explicit CIndexSubset(const hivePointCloud::CPointCloud &vPointCloud,
                          const std::vector<unsigned int> &vIndexSet = std::vector<unsigned int>(), bool vUpdate = true)
        : m_IndexSet(vIndexSet), m_PointCloud(vPointCloud) {
        if (vUpdate) updatePointAndNormalSet();
    }
    const hivePointCloud::CPointCloud &getPointCloud() const { return m_PointCloud; }
    const std::vector<unsigned int> &getIndexSet() const { return m_IndexSet; }
    void setIndexSet(const std::vector<unsigned int> &vIndexSet, bool vUpdate = true) {
        m_IndexSet = vIndexSet;
        if (vUpdate) updatePointAndNormalSet();
    }
    void swapIndexSet(std::vector<unsigned int> &vIndexSet, bool vUpdate = true) {
        m_IndexSet.swap(vIndexSet);
        if (vUpdate) updatePointAndNormalSet();
    }
    void updatePointAndNormalSet() {
        const double *pPos = m_PointCloud.getPosPointer();
        const double *pNormal = m_PointCloud.getNormalPointer();
        m_PointSet.resize(m_IndexSet.size());
        m_NormalSet.resize(pNormal ? m_IndexSet.size() : 0);
        for (size_t i = 0; i < m_IndexSet.size(); ++i) {
            const size_t k = (size_t)m_IndexSet[i] * 3;
            m_PointSet[i] = ColVector3(pPos[k], pPos[k + 1], pPos[k + 2]);
            if (pNormal) m_NormalSet[i] = ColVector3(pNormal[k], pNormal[k + 1], pNormal[k + 2]);
        }
    }

protected:
    virtual ~CIndexSubset() {}

private:
    std::vector<unsigned int> m_IndexSet;
    const hivePointCloud::CPointCloud &m_PointCloud;
};

}  // namespace hiveRegistration
