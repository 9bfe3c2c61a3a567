#include "iterative_fit.h"
#include <algorithm>
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
    if (!__fitLoopFused(Rot, Tra)) __fitLoop(Rot, Tra);
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

// The same loop with nothing but the final (R,T) leaving the device: possible when every stage is the GPU default
// (closest-points estimation, pass-through rejection, SVD2, default criteria) and the samples are indices into the
// source cloud.  Same iteration count, state and -- bit for bit -- transform as the stage-by-stage path.
// false: not applicable (the caller then runs __fitLoop).
bool CIterativeFit::__fitLoopFused(Eigen::Matrix3d &vioR, ColVector3 &vioT) {
    int Fused = 1;
    PTR_CONTROL_PARAMS->dumpValue(getKeyFusedLoop(), Fused);
    pcrhost::CDeviceSession *pSession = pcrhost::CDeviceSession::getInstance();
    if (!Fused || pSession->world() > 1) return false;  // sharded runs reduce through the host hook, stage by stage
    CCorrespondenceEstimationClosestPointsGPU *pClosest = dynamic_cast<CCorrespondenceEstimationClosestPointsGPU *>(m_pCorrespondenceEstimation);
    if (!pClosest || pClosest->usesNormalFit() ||  // (the opt-in normal-aware pairs run stage by stage)
        !dynamic_cast<CCorrespondenceRejectionClosestPoints *>(m_pCorrespondenceRejection) ||
        !dynamic_cast<CTransformationEstimationSVD2GPU *>(m_pTransformationEsitmation))
        return false;
    const CIndexSubset *pSamples = dynamic_cast<const CIndexSubset *>(m_SampleSubset.get());
    if (!pSamples || &pSamples->getPointCloud() != m_pSourcePointCloud) return false;
    const pcrhost::SDeviceCloud *pSrc = pSession->acquire(m_pSourcePointCloud), *pTgt = pSession->acquire(m_pTargetPointCloud);
    if (!pSrc || !pTgt) return false;  // the stage-by-stage path reports the failure

    CControlParameters *pConfig = PTR_CONTROL_PARAMS;
    unsigned int NumNeighbors = 4;
    double SDFT = 36.0, NDFT = 4.0;
    pConfig->dumpValue(CCorrespondenceEstimationClosestPointsGPU::getKeyNumNeighbors(), NumNeighbors);
    pConfig->dumpValue(CCorrespondenceEstimationClosestPointsGPU::getKeySquareDistFactorThreshold(), SDFT);
    pConfig->dumpValue(CCorrespondenceEstimationClosestPointsGPU::getKeyNeighbourDistFactorThreshold(), NDFT);
    pcr_convergence_params Conv;
    Conv.max_iterations = 1000;
    bool FAMI = false;
    Conv.rotation_threshold = Conv.translation_threshold = Conv.abs_mse = Conv.rel_mse = 0.0;
    Conv.max_similar = 0;
    pConfig->dumpValue(CBaseConvergenceCriteria::getKeyMaxIterations(), Conv.max_iterations);
    pConfig->dumpValue(CDefaultConvergenceCriteria::getKeyFailureAfterMaxIterations(), FAMI);
    pConfig->dumpValue(CDefaultConvergenceCriteria::getKeyRotationThreshold(), Conv.rotation_threshold);
    pConfig->dumpValue(CDefaultConvergenceCriteria::getKeyTranslationThreshold(), Conv.translation_threshold);
    pConfig->dumpValue(CDefaultConvergenceCriteria::getKeyAbsoluteMSE(), Conv.abs_mse);
    pConfig->dumpValue(CDefaultConvergenceCriteria::getKeyRelativeMSE(), Conv.rel_mse);
    pConfig->dumpValue(CDefaultConvergenceCriteria::getKeyMaxIterationSimilarTransforms(), Conv.max_similar);
    Conv.failure_after_max_iterations = FAMI ? 1 : 0;
    const double UniqueSquareDist = product<CUniqueData>(CUniqueData::getSigTgtUniqueData())->getUniqSquareDist();

    const std::vector<unsigned int> &Indices = pSamples->getIndexSet();
    bool Identity = Indices.size() == pSrc->NumPoints;
    for (size_t i = 0; Identity && i < Indices.size(); ++i) Identity = Indices[i] == i;
    // with FAMI the reference iterates past MI until another test fires (possibly for ever): bounded here
    const unsigned int Budget = FAMI ? 4 * Conv.max_iterations + 64 : Conv.max_iterations;
    const unsigned int TrajCap = m_Observer ? Budget : 0;
    pcr_fit *pFit = NULL;
    int Status = pcr_fit_create(pSession->context(), pSrc->pIndex, pTgt->pIndex, NULL, 0, Identity ? pSrc->NumPoints : 0u,
                                NumNeighbors, UniqueSquareDist * SDFT, UniqueSquareDist * NDFT, &Conv, UniqueSquareDist,
                                TrajCap, &pFit);
    if (Status == PCR_OK && !Identity) Status = pcr_fit_set_samples(pFit, Indices.data(), (uint32_t)Indices.size());
    if (Status == PCR_OK) Status = pcr_fit_reset(pFit, vioR.data(), vioT.data());
    uint32_t Iterations = 0;
    int State = PCR_FIT_NOT_CONVERGED;
    unsigned int Enqueued = 0;
    // read into temporaries: a failure half-way must leave the caller's (R,T) untouched, so that the stage-by-stage
    // loop the caller falls back to starts from where this one started, not from a partially advanced transform
    double ReadR[9], ReadT[3];
    while (Status == PCR_OK) {
        // a batch of iterations without the host in between; once the criteria fire the rest of a batch is no-ops
        const unsigned int Batch = std::min(16u, Budget - Enqueued);
        for (unsigned int i = 0; i < Batch && Status == PCR_OK; ++i) Status = pcr_fit_enqueue_iteration(pFit);
        Enqueued += Batch;
        if (Status == PCR_OK) Status = pcr_fit_read(pFit, ReadR, ReadT, &Iterations, &State, NULL, NULL, NULL, 0);
        if (Iterations < Enqueued || Enqueued >= Budget) break;  // stopped by itself, or out of budget
    }
    if (Status == PCR_OK && m_Observer && Iterations) {
        std::vector<double> Traj((size_t)Iterations * 14);
        if (pcr_fit_read(pFit, NULL, NULL, NULL, NULL, NULL, NULL, Traj.data(), Iterations) == PCR_OK)
            for (uint32_t i = 0; i < Iterations && i < TrajCap; ++i)
                m_Observer(i + 1, &Traj[(size_t)i * 14], &Traj[(size_t)i * 14 + 9], Traj[(size_t)i * 14 + 12],
                           (unsigned int)Traj[(size_t)i * 14 + 13], m_ObserverUser);
    }
    if (pFit) pcr_fit_destroy(pFit);
    if (Status != PCR_OK) {
        hiveEventLogger::hiveOutputWarning(__EXCEPTION_SITE__, std::string("device-resident loop: ") + pcr_strerror(Status) + " -- " + pcr_last_error());
        return false;
    }
    std::copy(ReadR, ReadR + 9, vioR.data());
    std::copy(ReadT, ReadT + 3, vioT.data());
    m_NumIterations = Iterations;
    pConfig->set(CBaseConvergenceCriteria::getKeyConvergenceState(), State < 0 ? 0 : State);
    return true;
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
