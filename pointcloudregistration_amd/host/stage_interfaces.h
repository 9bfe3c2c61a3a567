// The five stage base classes the reference's CIterativeFit drives, with the reference's public and
// protected member names and argument meaning:
//   CBaseSampler                      BaseSampler.h:13-26
//   CBaseCorrespondenceEstimation     BaseCorrespondenceEstimation.h:15-31, .cpp:15-32
//   CBaseCorrespondenceRejection      BaseCorrespondenceRejection.h:12-40
//   CBaseTransformationEstimation     BaseTransformationEstimation.h:7-38
//   CBaseConvergenceCriteria          BaseConvergenceCriteria.h:10-50, .cpp:20-60
// A GPU plugin derives from one of these exactly as a reference plugin does; CIterativeFit does not know
// which kind it is driving.
#pragma once
#include <map>
#include <string>
#include "control_params.h"
#include "hive_standin.h"
#include "point_cloud.h"
#include "reg_types.h"

namespace hiveRegistration {

class CBaseSampler : public hiveCommon::CBaseProduct {
public:
    CorrespondencePointSet samplePointCloud(const hivePointCloud::CPointCloud &vPointCloud) { return _doSampleV(vPointCloud); }

protected:
    virtual ~CBaseSampler() {}
    virtual CorrespondencePointSet _doSampleV(const hivePointCloud::CPointCloud &vPointCloud) = 0;
};

class CBaseCorrespondenceEstimation : public hiveCommon::CBaseProduct {
public:
    CBaseCorrespondenceEstimation() : m_pSourcePointCloud(NULL), m_pTargetPointCloud(NULL) {}

    // BaseCorrespondenceEstimation.cpp:24-32 forwards only when BOTH pointers differ from the stored ones
    // (`&&`), so a second fit that replaces just one cloud keeps stale state.  This layer forwards when EITHER
    // differs: every call the reference forwards is still forwarded, and sequential scan-to-scan registration
    //  works.  The GPU plugin compares pointers itself, so an unchanged cloud costs nothing.
    void setPointCloud(const hivePointCloud::CPointCloud *vSourcePointCloud, const hivePointCloud::CPointCloud *vTargetPointCloud) {
        if (vSourcePointCloud != m_pSourcePointCloud || vTargetPointCloud != m_pTargetPointCloud)
            _setPointCloudV(vSourcePointCloud, vTargetPointCloud);
    }
    void determineCorrespondences(const CorrespondencePointSet &vTgtCorPointSet, const Eigen::Matrix3d &vR, const ColVector3 &vT,
                                  CorrespondencePairSet &vioCorrespondencePairs) {
        _determineCorrespondencesV(vTgtCorPointSet, vR, vT, vioCorrespondencePairs);
    }

protected:
    const hivePointCloud::CPointCloud *m_pSourcePointCloud;
    const hivePointCloud::CPointCloud *m_pTargetPointCloud;

    virtual ~CBaseCorrespondenceEstimation() {}
    virtual void _setPointCloudV(const hivePointCloud::CPointCloud *vSourcePointCloud,
                                 const hivePointCloud::CPointCloud *vTargetPointCloud) = 0;
    virtual void _determineCorrespondencesV(const CorrespondencePointSet &vTgtCorPointSet, const Eigen::Matrix3d &vR,
                                            const ColVector3 &vT, CorrespondencePairSet &vioCorrespondencePairs) = 0;
};

class CBaseCorrespondenceRejection : public hiveCommon::CBaseProduct {
public:
    CBaseCorrespondenceRejection() : m_pSourcePointCloud(NULL), m_pTargetPointCloud(NULL) {}

    void setPointCloud(const hivePointCloud::CPointCloud *vSourcePointCloud, const hivePointCloud::CPointCloud *vTargetPointCloud) {
        if (vSourcePointCloud != m_pSourcePointCloud) _setSrcPointCloudV(vSourcePointCloud);
        if (vTargetPointCloud != m_pTargetPointCloud) _setTgtPointCloudV(vTargetPointCloud);
    }
    void rejectInvalidPairs(const CorrespondencePairSet &vAllCorrespondencePairs, CorrespondencePairSet &voValidCorrespondencePairs) {
        _rejectInvalidPairsV(vAllCorrespondencePairs, voValidCorrespondencePairs);
    }

protected:
    const hivePointCloud::CPointCloud *m_pSourcePointCloud;
    const hivePointCloud::CPointCloud *m_pTargetPointCloud;

    virtual ~CBaseCorrespondenceRejection() {}
    virtual void _rejectInvalidPairsV(const CorrespondencePairSet &vAllCorrespondencePairs,
                                      CorrespondencePairSet &voValidCorrespondencePairs) = 0;
    virtual void _setSrcPointCloudV(const hivePointCloud::CPointCloud *vSrcPointCloud) { m_pSourcePointCloud = vSrcPointCloud; }
    virtual void _setTgtPointCloudV(const hivePointCloud::CPointCloud *vTgtPointCloud) { m_pTargetPointCloud = vTgtPointCloud; }
};

class CBaseTransformationEstimation : public hiveCommon::CBaseProduct {
public:
    // The raw-pointer overload of the reference initialises T from vioR (BaseTransformationEstimation.h:15, a
    // bug; the overload has no caller).  Here it reads vioT.
    void computeRotationAndTranslation(const CorrespondencePairSet &vCorrespondencePairs, double *vioR, double *vioT) {
        Eigen::Matrix3d RMat3;
        ColVector3 TVec3;
        std::memcpy(RMat3.data(), vioR, 9 * sizeof(double));
        std::memcpy(TVec3.data(), vioT, 3 * sizeof(double));
        computeRotationAndTranslation(vCorrespondencePairs, RMat3, TVec3);
        std::memcpy(vioR, RMat3.data(), 9 * sizeof(double));
        std::memcpy(vioT, TVec3.data(), 3 * sizeof(double));
    }
    void computeRotationAndTranslation(const CorrespondencePairSet &vCorrespondencePairs, Eigen::Matrix3d &vioR, ColVector3 &vioT) {
        _computeRotationAndTranslationV(vCorrespondencePairs, vioR, vioT);
    }
    // R <- Rc*R ; T <- Rc*T + Tc   (BaseTransformationEstimation.h:28-32)
    void accumulateTransformation(Eigen::Matrix3d &vioR, ColVector3 &vioT, const Eigen::Matrix3d &vCurR, const ColVector3 &vCurT) {
        vioR = vCurR * vioR;
        vioT = vCurR * vioT + vCurT;
    }

protected:
    virtual ~CBaseTransformationEstimation() {}
    virtual void _computeRotationAndTranslationV(const CorrespondencePairSet &vCorrespondencePairs, Eigen::Matrix3d &vioR,
                                                 ColVector3 &vioT) = 0;
};

class CBaseConvergenceCriteria : public hiveCommon::CBaseProduct {
    PCR_FACTORY_CREATABLE(CBaseConvergenceCriteria)
public:
    CBaseConvergenceCriteria() {
        PTR_CONTROL_PARAMS->setIfNotExist(getKeyMaxIterations(), 1000u);  // BaseConvergenceCriteria.cpp:33
        PTR_CONTROL_PARAMS->setIfNotExist(getKeyConvergenceState(), 0);
        m_MapConvergenceType[0] = "NotConverged";
        m_MapConvergenceType[1] = "Criteria_Iterations";
        _letFactoryReleaseProduct();
    }
    bool operator()(unsigned int vIterationCounter) { return _isConvergedV(vIterationCounter); }
    bool isConverged(unsigned int vIterationCounter) { return _isConvergedV(vIterationCounter); }

    static std::string getClassSig() { return toUpperCopy("BaseConvergenceCriteria"); }
    static std::string getKeyMaxIterations() { return toUpperCopy("BaseConvergenceCriteria.MI"); }
    static std::string getKeyConvergenceState() { return toUpperCopy("BaseConvergenceCriteria.CS"); }

    // The state lives in the config map under getKeyConvergenceState() (the reference stores it as a raw
    // char; here as its integer code).  Unknown codes read as "NotConverged" (BaseConvergenceCriteria.cpp:42-52).
    const std::string &getConvergenceMsg() const {
        int State = 0;
        PTR_CONTROL_PARAMS->dumpValue(getKeyConvergenceState(), State);
        std::map<int, std::string>::const_iterator Itr = m_MapConvergenceType.find(State);
        return Itr == m_MapConvergenceType.end() ? m_MapConvergenceType.find(0)->second : Itr->second;
    }
    void setMapConvergenceType(const char vState, const std::string &vStrConvergenceType) {
        m_MapConvergenceType[(int)vState] = vStrConvergenceType;
    }

protected:
    virtual ~CBaseConvergenceCriteria() {}
    virtual bool _isConvergedV(unsigned int vIterationCounter) { return __isIterationEndedV(vIterationCounter); }
    virtual bool __isIterationEndedV(unsigned int vIterationCounter) const {
        unsigned int MaxIterations = 1000;
        PTR_CONTROL_PARAMS->dumpValue(getKeyMaxIterations(), MaxIterations);
        return vIterationCounter >= MaxIterations;  // BaseConvergenceCriteria.cpp:20-27
    }

private:
    std::map<int, std::string> m_MapConvergenceType;
};

}  // namespace hiveRegistration
