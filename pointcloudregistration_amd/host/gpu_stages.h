// The MI355X plugins for the ICP fine-registration loop.  Each derives from the same stage base class and is
// selected by the same config keys (F_CORRESPONDENCEESTIMATION, F_CORRESPONDENCEREJECTION,
// F_TRANSFORMATIONESTIMATION, F_SAMPLER -- ICPConstGlobleValue.h:8-21) as the reference plugin it replaces:
//
//   CCorrespondenceEstimationClosestPointsGPU  <- CCorrespondenceEstimationClosestPoints
//        (FineRegistrationClosestPoints/CorrespondenceEstimationClosestPoints.cpp:40-124)
//   CCorrespondenceRejectionClosestPoints      <- same name, pass-through (CorrespondenceRejectionClosestPoints.cpp:18-22)
//   CRejectionCentroidDistGPU                  <- CRejectionCentroidDist (RejectionCentroidDist.cpp:23-208)
//   CTransformationEstimationSVD2GPU           <- CTransformationEstimationSVD2 (TransformationEstimationSVD2.cpp:17-117)
//   CUniqueData                                <- CUniqueData (UniqueData.cpp:25-53), on the shared GPU index
//   CSamplerIndexList                          deterministic sampler used by the harness
//
// All device work goes through the C ABI in include/pcr.h; nothing here touches HIP.
#pragma once
#include <string>
#include <vector>
#include "device_session.h"
#include "stage_interfaces.h"

namespace hiveRegistration {

// What one fused ICP step leaves behind: the 17 partial sums (already summed over ranks) and everything needed
// to materialise the pair lists later, if a host-side stage asks for them.
struct SDeviceStepResult {
    double Partials[17], PivotP[3], PivotQ[3];
    Eigen::Matrix3d R;
    ColVector3 T;
    unsigned int K;
    double ThrFwd, ThrRev;
    const hivePointCloud::CPointCloud *pSource, *pTarget;
    std::vector<unsigned int> SampleIdx;  // empty = every source point
    bool Materialised;
    bool NormalFit;  // the opt-in normal-aware pair decision made these pairs (NVDT, NDT, CompareNormals)
    double NVDT, NDT;
    int CompareNormals;
    VecColVector3 SrcPoints, SrcNormals, TgtPoints, TgtNormals;
    SDeviceStepResult()
        : K(0), ThrFwd(0), ThrRev(0), pSource(NULL), pTarget(NULL), Materialised(false), NormalFit(false), NVDT(0), NDT(0),
          CompareNormals(0) {}
    void materialise();  // re-runs the step with index outputs and gathers the kept pairs, in sample order
};

// One side of a CorrespondencePairSet whose points are still on the GPU.  getNumPoint() answers from the
// partials; getPointSet()/getNormalSet() copy the pairs back on first use (CBasePointCloudSubset::_materializeV).
class CDevicePairSubset : public CBasePointCloudSubset {
public:
    CDevicePairSubset(const std::shared_ptr<SDeviceStepResult> &vResult, bool vIsSourceSide)
        : m_pResult(vResult), m_IsSourceSide(vIsSourceSide), m_Copied(false) {}
    const SDeviceStepResult &getStepResult() const { return *m_pResult; }
    virtual unsigned int getNumPoint() const { return (unsigned int)m_pResult->Partials[0]; }

protected:
    virtual ~CDevicePairSubset() {}
    virtual void _materializeV() const;

private:
    std::shared_ptr<SDeviceStepResult> m_pResult;
    bool m_IsSourceSide;
    mutable bool m_Copied;
};

class CUniqueData : public hiveCommon::CBaseProduct {
    PCR_FACTORY_CREATABLE(CUniqueData)
public:
    static std::string getSigSrcUniqueData() { return toUpperCopy("SrcUniqueData"); }
    static std::string getSigTgtUniqueData() { return toUpperCopy("TgtUniqueData"); }
    CUniqueData() : m_UnitSquareDist(0.0) { _letFactoryReleaseProduct(); }
    void setPointCloud(const hivePointCloud::CPointCloud &vPointCloud);
    double getUniqSquareDist() const { return m_UnitSquareDist; }
    // no reference counterpart: lets a harness drive the host-only stages (criteria) without a device
    void setUniqSquareDist(double vValue) { m_UnitSquareDist = vValue; }

private:
    double m_UnitSquareDist;
};

class CSamplerIndexList : public CBaseSampler {
    PCR_FACTORY_CREATABLE(CSamplerIndexList)
public:
    CSamplerIndexList() { _letFactoryReleaseProduct(); }
    static std::string getClassSig() { return toUpperCopy("SamplerIndexList"); }
    // empty list = every point of the cloud, in index order
    void setIndexList(const std::vector<unsigned int> &vIndexList) { m_IndexList = vIndexList; }

protected:
    virtual CorrespondencePointSet _doSampleV(const hivePointCloud::CPointCloud &vPointCloud);

private:
    std::vector<unsigned int> m_IndexList;
};

// CSamplerRandom (Sampler/SamplerRandom.h:9-33, .cpp:25-58) on the device: same keys, same selection rule
// (ratio = SR if in (0,1), else SN/N if 0 < SN < N, else no sample set), a seeded counter-based draw where the
// reference calls an unseeded rand().  SamplerRandom.Seed (no reference counterpart) picks the stream.
class CSamplerRandomGPU : public CBaseSampler {
    PCR_FACTORY_CREATABLE(CSamplerRandomGPU)
public:
    CSamplerRandomGPU() { _letFactoryReleaseProduct(); }
    static std::string getClassSig() { return toUpperCopy("SamplerRandomGPU"); }
    static std::string getKeySampleNum() { return toUpperCopy("SamplerRandom.SN"); }
    static std::string getKeySampleRatio() { return toUpperCopy("SamplerRandom.SR"); }
    static std::string getKeySeed() { return toUpperCopy("SamplerRandom.Seed"); }
    // 1: draw with the host's rand() -- one call per point, in point order, exactly as the reference's loop does
    // (SamplerRandom.cpp:45-48) -- and apply its rule to that stream on the device: with the same C runtime and the same
    // rand() state the sample set is the reference's, index for index.  0 (default): the seeded counter hash.
    static std::string getKeyUseRand() { return toUpperCopy("SamplerRandomGPU.USERAND"); }

protected:
    virtual CorrespondencePointSet _doSampleV(const hivePointCloud::CPointCloud &vPointCloud);
};

// One point per occupied cell of the cloud's GPU grid: the sampler the reference declares and leaves empty
// (Sampler/SamplerUniformGrid.cpp:25-28).
class CSamplerUniformGridGPU : public CBaseSampler {
    PCR_FACTORY_CREATABLE(CSamplerUniformGridGPU)
public:
    CSamplerUniformGridGPU() { _letFactoryReleaseProduct(); }
    static std::string getClassSig() { return toUpperCopy("SamplerUniformGridGPU"); }

protected:
    virtual CorrespondencePointSet _doSampleV(const hivePointCloud::CPointCloud &vPointCloud);
};

class CCorrespondenceEstimationClosestPointsGPU : public CBaseCorrespondenceEstimation {
    PCR_FACTORY_CREATABLE(CCorrespondenceEstimationClosestPointsGPU)
public:
    CCorrespondenceEstimationClosestPointsGPU();
    static std::string getClassSig() { return toUpperCopy("CorrespondenceEstimationClosestPointsGPU"); }
    // the reference plugin's keys, so one RegCfg.txt drives either implementation
    static std::string getKeyNumNeighbors() { return toUpperCopy("CorrespondenceEstimationClosestPoints.NN"); }
    static std::string getKeySquareDistFactorThreshold() { return toUpperCopy("CorrespondenceEstimationClosestPoints.SDFT"); }
    static std::string getKeyNeighbourDistFactorThreshold() { return toUpperCopy("CorrespondenceEstimationClosestPoints.NDFT"); }
    static std::string getKeyNormalDeviationThreshold() { return toUpperCopy("CorrespondenceEstimationClosestPoints.NDT"); }
    static std::string getKeyNormalVectorDistThreshold() { return toUpperCopy("CorrespondenceEstimationClosestPoints.NVDT"); }
    // OPT-IN (no reference counterpart as a switch: the reference comments the call out, :105): 1 = pairs are decided by
    // __insertFitPairs (:128-175) instead of the reciprocal gate; 2 = and normals are compared too (NDT; the reference
    // hard-wires NoNormalCompare).  Both clouds must carry normals.  Runs stage by stage, not in the fused device loop.
    static std::string getKeyNormalFit() { return toUpperCopy("CorrespondenceEstimationClosestPointsGPU.NORMALFIT"); }
    bool usesNormalFit() const { return m_NormalFit != 0; }

protected:
    virtual ~CCorrespondenceEstimationClosestPointsGPU() {}
    virtual void _setPointCloudV(const hivePointCloud::CPointCloud *vSourcePointCloud, const hivePointCloud::CPointCloud *vTargetPointCloud);
    virtual void _determineCorrespondencesV(const CorrespondencePointSet &vSourceCorrespondencePointSet, const Eigen::Matrix3d &vR,
                                            const ColVector3 &vT, CorrespondencePairSet &vioCorrespondencePairs);

private:
    unsigned int m_NumNeighbors;
    double m_SquareDistFactorThreshold, m_NeighbourDistFactorThreshold;
    int m_NormalFit;
    double m_NormalDeviationThreshold, m_NormalVectorDistThreshold;
};

class CCorrespondenceRejectionClosestPoints : public CBaseCorrespondenceRejection {
    PCR_FACTORY_CREATABLE(CCorrespondenceRejectionClosestPoints)
public:
    CCorrespondenceRejectionClosestPoints() { _letFactoryReleaseProduct(); }
    static std::string getClassSig() { return toUpperCopy("CorrespondenceRejectionClosestPoints"); }

protected:
    virtual void _rejectInvalidPairsV(const CorrespondencePairSet &vAll, CorrespondencePairSet &voValid) { voValid = vAll; }
};

class CRejectionCentroidDistGPU : public CBaseCorrespondenceRejection {
    PCR_FACTORY_CREATABLE(CRejectionCentroidDistGPU)
public:
    CRejectionCentroidDistGPU() { _letFactoryReleaseProduct(); }
    static std::string getClassSig() { return toUpperCopy("RejectionCentroidDistGPU"); }
    static std::string getKeyMaxSearchNum() { return toUpperCopy("RejectionCentroidDist.MSN"); }
    static std::string getKeyMaxSearchDist() { return toUpperCopy("RejectionCentroidDist.MSD"); }
    static std::string getKeyCriticalDistFactor() { return toUpperCopy("RejectionCentroidDist.CDF"); }
    // Pairs sharded over ranks (no reference counterpart): mean and sigma are then taken over ALL ranks' pairs, summed
    // through this hook (pcr_reject_centroid_dist_sharded); NULL (the default): every pair is here.
    void setReduction(pcr_reduce17_fn vReduce, void *vUser) { m_Reduce = vReduce; m_ReduceUser = vUser; }

protected:
    virtual void _rejectInvalidPairsV(const CorrespondencePairSet &vAllCorrespondencePairs, CorrespondencePairSet &voValidCorrespondencePairs);

private:
    pcr_reduce17_fn m_Reduce = NULL;
    void *m_ReduceUser = NULL;
};

class CTransformationEstimationSVD2GPU : public CBaseTransformationEstimation {
    PCR_FACTORY_CREATABLE(CTransformationEstimationSVD2GPU)
public:
    CTransformationEstimationSVD2GPU() { _letFactoryReleaseProduct(); }
    static std::string getClassSig() { return toUpperCopy("TransformationEstimationSVD2GPU"); }
    // TransformationEstimationSVD2.cpp:33-49: increment (voR, voT) that maps point set A onto B
    void computeRotationAndTranslation(const VecColVector3 &vPntSetA, const VecColVector3 &vPntSetB, Eigen::Matrix3d &voR, ColVector3 &voT);
    using CBaseTransformationEstimation::computeRotationAndTranslation;

protected:
    virtual void _computeRotationAndTranslationV(const CorrespondencePairSet &vCorrespondencePairs, Eigen::Matrix3d &vioR, ColVector3 &vioT);
};

// Sum |p-q|^2 and pair count of a pair set without walking it when it is still on the device
// (CDefaultConvergenceCriteria::__calculateMSE, DefaultConvergenceCriterial.cpp:46-65).  false if empty.
bool pairSetMSE(const CorrespondencePairSet &vPairs, double &voMSE, unsigned int &voNumPairs);

}  // namespace hiveRegistration
