// CDefaultConvergenceCriteria (DefaultConvergenceCriterial.h:10-66, .cpp:13-207) and CIterativeFit
// (IterativeFit.h:20-67, .cpp:23-206): the host-side loop around the GPU stages.  Same member names, same
// order of operations per iteration, same config keys; the only data it reads per iteration are the 17
// partial sums (pair lists stay on the device unless a host-side stage asks for them).
#pragma once
#include <string>
#include <tuple>
#include "gpu_stages.h"

namespace hiveRegistration {

class CDefaultConvergenceCriteria : public CBaseConvergenceCriteria {
    PCR_FACTORY_CREATABLE(CDefaultConvergenceCriteria)
public:
    CDefaultConvergenceCriteria();

    static std::string getClassSig() { return toUpperCopy("DefaultConvergenceCriteria"); }
    static std::string getKeyMaxIterationSimilarTransforms() { return toUpperCopy("DefaultConvergenceCriteria.MIST"); }
    static std::string getKeyFailureAfterMaxIterations() { return toUpperCopy("DefaultConvergenceCriteria.FAMI"); }
    static std::string getKeyRotationThreshold() { return toUpperCopy("DefaultConvergenceCriteria.RT"); }
    static std::string getKeyTranslationThreshold() { return toUpperCopy("DefaultConvergenceCriteria.TT"); }
    static std::string getKeyRelativeMSE() { return toUpperCopy("DefaultConvergenceCriteria.RMSE"); }
    static std::string getKeyAbsoluteMSE() { return toUpperCopy("DefaultConvergenceCriteria.AMSE"); }

    enum EConvergenceState {
        CONVERGENCE_CRITERIA_NOT_CONVERGED,
        CONVERGENCE_CRITERIA_ITERATIONS,
        CONVERGENCE_CRITERIA_TRANSFORM,
        CONVERGENCE_CRITERIA_ABS_MSE,
        CONVERGENCE_CRITERIA_REL_MSE,
        CONVERGENCE_CRITERIA_NO_CORRESPONDENCES
    };

    // stores a pointer to the caller's object, as the reference does (DefaultConvergenceCriterial.h:26)
    void setCorrespondencePairSet(const CorrespondencePairSet &vCorrespondencePairs) { m_pCorrespondencePairs = &vCorrespondencePairs; }
    bool operator()(const Eigen::Matrix3d &vCurR, const ColVector3 &vCurT, unsigned int vIterationCounter) {
        return _isConvergedV(vCurR, vCurT, vIterationCounter);
    }
    bool isConverged(const Eigen::Matrix3d &vCurR, const ColVector3 &vCurT, unsigned int vIterationCounter) {
        return _isConvergedV(vCurR, vCurT, vIterationCounter);
    }
    // Not in the reference: its criteria object is a factory singleton whose m_PreMSE / m_PreRot /
    // similar-transform counter survive from one fit to the next.  CIterativeFit::fit leaves that
    // as is; a driver that registers many scan pairs calls this between pairs.
    void reset();
    double getCurrentMSE() const { return m_CurMSE; }

protected:
    using CBaseConvergenceCriteria::_isConvergedV;
    virtual bool _isConvergedV(const Eigen::Matrix3d &vCurR, const ColVector3 &vCurT, unsigned int vIterationCounter);

private:
    double m_PreMSE, m_CurMSE;
    Eigen::Matrix3d m_PreRot;
    ColVector3 m_PreTra;
    unsigned int m_IterationsSimilarTransforms;
    const CorrespondencePairSet *m_pCorrespondencePairs;

    bool __isIterationEnded(unsigned int vIterationCounter);
    bool __isTransformationEnded(const Eigen::Matrix3d &vCurR, const ColVector3 &vCurT);
    bool __isMSEAbsoluteEnded();
    bool __isMSERelativeEnded();
    bool __similarTransformGate(EConvergenceState vState);
    double __calculateMSE() const;
};

class CIterativeFit {
public:
    static std::string getKeySamplePointCloud() { return toUpperCopy("IterativeFit.SPC"); }
    // no reference counterpart: 1 (default) lets fit() run the whole loop device-resident (pcr_fit_*) when every
    // stage is the GPU default; 0 forces the stage-by-stage path through the plugin interfaces
    static std::string getKeyFusedLoop() { return toUpperCopy("IterativeFit.FusedLoop"); }

    CIterativeFit()
        : m_pSourcePointCloud(NULL), m_pTargetPointCloud(NULL), m_pPointSampler(NULL), m_pCorrespondenceEstimation(NULL),
          m_pCorrespondenceRejection(NULL), m_pTransformationEsitmation(NULL), m_pConvergenceCriterial(NULL) {}
    virtual ~CIterativeFit() {}

    void setFitClasses(const std::string &vPointSamplerSig, const std::string &vCorrespondenceEsitmationSig,
                       const std::string &vCorrespondenceRejectionSig, const std::string &vTransformationEsitmationSig);
    void setSampler(const std::string &vPointSamplerSig);
    void setCorrespondenceEstimation(const std::string &vCorrespondenceEsitmationSig);
    void setCorrespondenceRejection(const std::string &vCorrespondenceRejectionSig);
    void setTransformationEstimation(const std::string &vTransformationEsitmationSig);
    void setConvergenceCriterial(const std::string &vConvergenceCriterialSig);
    bool isComplete() const {
        return m_pPointSampler && m_pCorrespondenceEstimation && m_pCorrespondenceRejection && m_pTransformationEsitmation &&
               m_pConvergenceCriterial;
    }

    // (R, T, convergence message); boost::tuple in the reference (IterativeFit.h:36-39)
    std::tuple<Eigen::Matrix3d, ColVector3, std::string> fit(const hivePointCloud::CPointCloud &vSourcePointCloud,
                                                              const hivePointCloud::CPointCloud &vTargetPointCloud,
                                                              const Eigen::Matrix3d &vInitR, const ColVector3 &vInitT);

    // observation hook for tests and benchmarks (no reference counterpart): called after every iteration
    typedef void (*IterationObserver)(unsigned int vIteration, const double *vR9, const double *vT3, double vMSE,
                                      unsigned int vNumPairs, void *vUser);
    void setIterationObserver(IterationObserver vObserver, void *vUser) {
        m_Observer = vObserver;
        m_ObserverUser = vUser;
    }
    unsigned int getNumIterations() const { return m_NumIterations; }
    // drop the cached sample subset (the reference caches it across fit calls, IterativeFit.cpp:110-113)
    void clearSampleSubset() { m_SampleSubset.reset(); }

protected:
    enum EFitFlag { FIT_END, FIT_NOT_END };
    virtual void _updateV();
    virtual bool _isCorrespondencePointSetValidV();
    virtual EFitFlag _fitStepV(unsigned int vIterationCounter, Eigen::Matrix3d &vioR, ColVector3 &vioT);

private:
    const hivePointCloud::CPointCloud *m_pSourcePointCloud;
    const hivePointCloud::CPointCloud *m_pTargetPointCloud;
    CorrespondencePointSet m_SampleSubset;
    CBaseSampler *m_pPointSampler;
    CBaseCorrespondenceEstimation *m_pCorrespondenceEstimation;
    CBaseCorrespondenceRejection *m_pCorrespondenceRejection;
    CBaseTransformationEstimation *m_pTransformationEsitmation;
    CDefaultConvergenceCriteria *m_pConvergenceCriterial;
    IterationObserver m_Observer = NULL;
    void *m_ObserverUser = NULL;
    unsigned int m_NumIterations = 0;

    void __setMethodPointCloud();
    void __fitLoop(Eigen::Matrix3d &vioR, ColVector3 &vioT);
    bool __fitLoopFused(Eigen::Matrix3d &vioR, ColVector3 &vioT);
    void __setPointCloud(const hivePointCloud::CPointCloud &vSourcePointCloud, const hivePointCloud::CPointCloud &vTargetPointCloud);
};

}  // namespace hiveRegistration
