#include "iterative_fit.h"
#include <cmath>
#include <limits>

using namespace hiveRegistration;

static hiveCommon::CProductFactory<CDefaultConvergenceCriteria> s_CriteriaCreator(CDefaultConvergenceCriteria::getClassSig());

namespace {
template <typename T>
T *product(const std::string &vSig) {
    return dynamic_cast<T *>(hiveCommon::CProductFactoryData::getInstance()->getOrCreateProduct(vSig));
}
}  // namespace

// ---- CDefaultConvergenceCriteria ---------------------------------------------------------------------------

CDefaultConvergenceCriteria::CDefaultConvergenceCriteria() : m_pCorrespondencePairs(NULL) {
    _letFactoryReleaseProduct();
    CControlParameters *pConfig = PTR_CONTROL_PARAMS;  // defaults of DefaultConvergenceCriterial.cpp:27-34
    pConfig->setIfNotExist(getKeyFailureAfterMaxIterations(), false);
    pConfig->setIfNotExist(getKeyRotationThreshold(), 0.0001);
    pConfig->setIfNotExist(getKeyTranslationThreshold(), 3e-8);
    pConfig->setIfNotExist(getKeyRelativeMSE(), 0.01);
    pConfig->setIfNotExist(getKeyAbsoluteMSE(), 0.01);
    pConfig->setIfNotExist(getKeyMaxIterationSimilarTransforms(), 0u);
    setMapConvergenceType((char)CONVERGENCE_CRITERIA_TRANSFORM, "Criteria_Transform");  // strings of :36-41
    setMapConvergenceType((char)CONVERGENCE_CRITERIA_ABS_MSE, "Criteria_AbsMSE");
    setMapConvergenceType((char)CONVERGENCE_CRITERIA_REL_MSE, "Criteria_RelMSE");
    setMapConvergenceType((char)CONVERGENCE_CRITERIA_NO_CORRESPONDENCES, "Criteria_NoCorrespondeces");
    reset();
}

void CDefaultConvergenceCriteria::reset() {
    m_IterationsSimilarTransforms = 0;
    m_PreMSE = m_CurMSE = std::numeric_limits<double>::max();
    // m_PreRot is uninitialised in the reference (DefaultConvergenceCriterial.h:46): the first transformation test
    // compares against garbage.  NaN makes that first test false, which is what garbage far from a rotation does.
    for (int i = 0; i < 9; ++i) m_PreRot.data()[i] = std::numeric_limits<double>::quiet_NaN();
    m_PreTra.setZero();
}

double CDefaultConvergenceCriteria::__calculateMSE() const {
    double MSE;
    unsigned int NumPairs;
    if (!m_pCorrespondencePairs || !pairSetMSE(*m_pCorrespondencePairs, MSE, NumPairs)) return std::numeric_limits<double>::quiet_NaN();
    return MSE;
}

bool CDefaultConvergenceCriteria::_isConvergedV(const Eigen::Matrix3d &vCurR, const ColVector3 &vCurT, unsigned int vIterationCounter) {
    PTR_CONTROL_PARAMS->set(getKeyConvergenceState(), (int)CONVERGENCE_CRITERIA_NOT_CONVERGED);
    // short-circuit OR, in this order (:76)
    const bool IsConverged = __isIterationEnded(vIterationCounter) || __isTransformationEnded(vCurR, vCurT) || __isMSEAbsoluteEnded() ||
                             __isMSERelativeEnded();
    m_PreMSE = m_CurMSE;
    m_PreRot = vCurR;
    m_PreTra = vCurT;
    return IsConverged;
}

bool CDefaultConvergenceCriteria::__isIterationEnded(unsigned int vIterationCounter) {
    bool FailureAfterMaxIter = false;
    PTR_CONTROL_PARAMS->dumpValue(getKeyFailureAfterMaxIterations(), FailureAfterMaxIter);
    if (!CBaseConvergenceCriteria::__isIterationEndedV(vIterationCounter)) return false;
    PTR_CONTROL_PARAMS->set(getKeyConvergenceState(), (int)CONVERGENCE_CRITERIA_ITERATIONS);
    return !FailureAfterMaxIter;  // (:86-99)
}

bool CDefaultConvergenceCriteria::__similarTransformGate(EConvergenceState vState) {
    unsigned int MaxIterationSimilarTransforms = 0;
    PTR_CONTROL_PARAMS->dumpValue(getKeyMaxIterationSimilarTransforms(), MaxIterationSimilarTransforms);
    if (m_IterationsSimilarTransforms < MaxIterationSimilarTransforms) {
        ++m_IterationsSimilarTransforms;
        return false;
    }
    m_IterationsSimilarTransforms = 0;
    PTR_CONTROL_PARAMS->set(getKeyConvergenceState(), (int)vState);
    return true;
}

bool CDefaultConvergenceCriteria::__isTransformationEnded(const Eigen::Matrix3d &vCurR, const ColVector3 &vCurT) {
    double RotationThreshold = 0, TranslationThreshold = 0;
    PTR_CONTROL_PARAMS->dumpValue(getKeyRotationThreshold(), RotationThreshold);
    PTR_CONTROL_PARAMS->dumpValue(getKeyTranslationThreshold(), TranslationThreshold);
    const double RotDiff = (m_PreRot - vCurR).squaredNorm();
    const double TranslationSqr = vCurT.squaredNorm();  // the ACCUMULATED translation (:106)
    return RotDiff < RotationThreshold && TranslationSqr < TranslationThreshold && __similarTransformGate(CONVERGENCE_CRITERIA_TRANSFORM);
}

static double targetUniqueSquareDist() {
    CUniqueData *pUniqueData = product<CUniqueData>(CUniqueData::getSigTgtUniqueData());
    return pUniqueData ? pUniqueData->getUniqSquareDist() : std::numeric_limits<double>::quiet_NaN();
}

bool CDefaultConvergenceCriteria::__isMSEAbsoluteEnded() {
    double Threshold = 0;
    PTR_CONTROL_PARAMS->dumpValue(getKeyAbsoluteMSE(), Threshold);
    Threshold *= targetUniqueSquareDist();  // (:144-147)
    m_CurMSE = __calculateMSE();
    return std::fabs(m_CurMSE - m_PreMSE) < Threshold && __similarTransformGate(CONVERGENCE_CRITERIA_ABS_MSE);
}

bool CDefaultConvergenceCriteria::__isMSERelativeEnded() {
    double Threshold = 0;
    PTR_CONTROL_PARAMS->dumpValue(getKeyRelativeMSE(), Threshold);
    Threshold *= targetUniqueSquareDist();  // (:181-184)
    m_CurMSE = __calculateMSE();
    return std::fabs(m_CurMSE - m_PreMSE) / m_PreMSE < Threshold && __similarTransformGate(CONVERGENCE_CRITERIA_REL_MSE);
}

// ---- CIterativeFit -------------------------------------------------------------------------------------------

void CIterativeFit::setFitClasses(const std::string &vPointSamplerSig, const std::string &vCorrespondenceEsitmationSig,
                                  const std::string &vCorrespondenceRejectionSig, const std::string &vTransformationEsitmationSig) {
    setSampler(vPointSamplerSig);
    setCorrespondenceEstimation(vCorrespondenceEsitmationSig);
    setCorrespondenceRejection(vCorrespondenceRejectionSig);
    setTransformationEstimation(vTransformationEsitmationSig);
    setConvergenceCriterial(CDefaultConvergenceCriteria::getClassSig());
}
void CIterativeFit::setSampler(const std::string &vSig) { m_pPointSampler = product<CBaseSampler>(toUpperCopy(vSig)); }
void CIterativeFit::setCorrespondenceEstimation(const std::string &vSig) { m_pCorrespondenceEstimation = product<CBaseCorrespondenceEstimation>(toUpperCopy(vSig)); }
void CIterativeFit::setCorrespondenceRejection(const std::string &vSig) { m_pCorrespondenceRejection = product<CBaseCorrespondenceRejection>(toUpperCopy(vSig)); }
void CIterativeFit::setTransformationEstimation(const std::string &vSig) { m_pTransformationEsitmation = product<CBaseTransformationEstimation>(toUpperCopy(vSig)); }
void CIterativeFit::setConvergenceCriterial(const std::string &vSig) { m_pConvergenceCriterial = product<CDefaultConvergenceCriteria>(toUpperCopy(vSig)); }

std::tuple<Eigen::Matrix3d, ColVector3, std::string> CIterativeFit::fit(const hivePointCloud::CPointCloud &vSourcePointCloud,
                                                                         const hivePointCloud::CPointCloud &vTargetPointCloud,
                                                                         const Eigen::Matrix3d &vInitR, const ColVector3 &vInitT) {
    Eigen::Matrix3d Rot = vInitR;
    ColVector3 Tra = vInitT;
    m_NumIterations = 0;
    if (!isComplete()) return std::make_tuple(Rot, Tra, std::string("Fit classes not set!"));  // the reference _ASSERTs (:37-69)
    __setPointCloud(vSourcePointCloud, vTargetPointCloud);
    __setMethodPointCloud();
    if (!_isCorrespondencePointSetValidV()) {
        hiveEventLogger::hiveOutputWarning(__EXCEPTION_SITE__, "control point cannot be less than 5.");
        return std::make_tuple(Rot, Tra, std::string("Point Cloud Not Valid!"));  // (:86)
    }
    __fitLoop(Rot, Tra);
    return std::make_tuple(Rot, Tra, m_pConvergenceCriterial->getConvergenceMsg());
}

void CIterativeFit::_updateV() {
    int SampleTarget = 0;  // RegCfg.txt does not carry IterativeFit.SPC; 0 = sample the source, what ICRP needs
    PTR_CONTROL_PARAMS->dumpValue(getKeySamplePointCloud(), SampleTarget);
    m_SampleSubset = m_pPointSampler->samplePointCloud(*(SampleTarget ? m_pTargetPointCloud : m_pSourcePointCloud));
}

bool CIterativeFit::_isCorrespondencePointSetValidV() {
    // sampled once and cached across fit calls (:110-113) -- and refreshed here when the cached subset was drawn
    // from another cloud, which the reference misses
    const CIndexSubset *pCached = dynamic_cast<const CIndexSubset *>(m_SampleSubset.get());
    if (!m_SampleSubset.get() || m_SampleSubset->getNumPoint() == 0 ||
        (pCached && &pCached->getPointCloud() != m_pSourcePointCloud && &pCached->getPointCloud() != m_pTargetPointCloud))
        _updateV();
    return m_SampleSubset.get() && m_SampleSubset->getNumPoint() >= 5;
}

void CIterativeFit::__fitLoop(Eigen::Matrix3d &vioR, ColVector3 &vioT) {
    unsigned int IterationCounter = 0;  // first step is iteration 1 (:121-123)
    while (FIT_END != _fitStepV(++IterationCounter, vioR, vioT)) {
    }
    m_NumIterations = IterationCounter;
}

CIterativeFit::EFitFlag CIterativeFit::_fitStepV(unsigned int vIterationCounter, Eigen::Matrix3d &vioR, ColVector3 &vioT) {
    CorrespondencePairSet AllCorrespondencePairSet, ValidCorrespondencePairSet;
    m_pCorrespondenceEstimation->determineCorrespondences(m_SampleSubset, vioR, vioT, AllCorrespondencePairSet);
    m_pCorrespondenceRejection->rejectInvalidPairs(AllCorrespondencePairSet, ValidCorrespondencePairSet);

    double MSE = std::numeric_limits<double>::quiet_NaN();
    unsigned int NumPairs = 0;
    if (!pairSetMSE(ValidCorrespondencePairSet, MSE, NumPairs)) {
        // The reference divides by zero here (RegUtilityFunctions.cpp:71) and iterates on a NaN transform until MI.
        // Stop instead, with the state string the reference defines for this case but never sets (:41).
        PTR_CONTROL_PARAMS->set(CBaseConvergenceCriteria::getKeyConvergenceState(),
                                (int)CDefaultConvergenceCriteria::CONVERGENCE_CRITERIA_NO_CORRESPONDENCES);
        if (m_Observer) m_Observer(vIterationCounter, vioR.data(), vioT.data(), MSE, 0, m_ObserverUser);
        return FIT_END;
    }
    m_pTransformationEsitmation->computeRotationAndTranslation(ValidCorrespondencePairSet, vioR, vioT);
    m_pConvergenceCriterial->setCorrespondencePairSet(ValidCorrespondencePairSet);
    const bool IsConverged = m_pConvergenceCriterial->isConverged(vioR, vioT, vIterationCounter);
    if (m_Observer) m_Observer(vIterationCounter, vioR.data(), vioT.data(), MSE, NumPairs, m_ObserverUser);
    return IsConverged ? FIT_END : FIT_NOT_END;
}

void CIterativeFit::__setPointCloud(const hivePointCloud::CPointCloud &vSourcePointCloud, const hivePointCloud::CPointCloud &vTargetPointCloud) {
    // (:183-187) the unit distance of BOTH clouds, every fit; here each reuses the cloud's one GPU index
    CUniqueData *pSrcData = product<CUniqueData>(CUniqueData::getSigSrcUniqueData());
    CUniqueData *pTgtData = product<CUniqueData>(CUniqueData::getSigTgtUniqueData());
    pSrcData->setPointCloud(vSourcePointCloud);
    pTgtData->setPointCloud(vTargetPointCloud);
    m_pSourcePointCloud = &vSourcePointCloud;
    m_pTargetPointCloud = &vTargetPointCloud;
}

void CIterativeFit::__setMethodPointCloud() {
    m_pCorrespondenceEstimation->setPointCloud(m_pSourcePointCloud, m_pTargetPointCloud);
    m_pCorrespondenceRejection->setPointCloud(m_pSourcePointCloud, m_pTargetPointCloud);
}
