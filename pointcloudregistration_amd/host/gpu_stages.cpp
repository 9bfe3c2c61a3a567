#include "gpu_stages.h"
#include <cmath>
#include <cstdlib>
#include <limits>

using namespace hiveRegistration;
using pcrhost::CDeviceSession;
using pcrhost::SDeviceCloud;

// factory registration, as every reference plugin does at file scope
static hiveCommon::CProductFactory<CControlParameters> s_ConfigCreator(CControlParameters::getClassSig());
static hiveCommon::CProductFactory<CUniqueData> s_SrcUniqueCreator(CUniqueData::getSigSrcUniqueData());
static hiveCommon::CProductFactory<CUniqueData> s_TgtUniqueCreator(CUniqueData::getSigTgtUniqueData());
static hiveCommon::CProductFactory<CSamplerIndexList> s_SamplerCreator(CSamplerIndexList::getClassSig());
static hiveCommon::CProductFactory<CSamplerRandomGPU> s_RandomSamplerCreator(CSamplerRandomGPU::getClassSig());
static hiveCommon::CProductFactory<CSamplerUniformGridGPU> s_GridSamplerCreator(CSamplerUniformGridGPU::getClassSig());
static hiveCommon::CProductFactory<CCorrespondenceEstimationClosestPointsGPU> s_EstimationCreator(
    CCorrespondenceEstimationClosestPointsGPU::getClassSig());
static hiveCommon::CProductFactory<CCorrespondenceRejectionClosestPoints> s_PassThroughCreator(
    CCorrespondenceRejectionClosestPoints::getClassSig());
static hiveCommon::CProductFactory<CRejectionCentroidDistGPU> s_CentroidRejectionCreator(CRejectionCentroidDistGPU::getClassSig());
static hiveCommon::CProductFactory<CTransformationEstimationSVD2GPU> s_SVD2Creator(CTransformationEstimationSVD2GPU::getClassSig());
static hiveCommon::CProductFactory<CBaseConvergenceCriteria> s_BaseCriteriaCreator(CBaseConvergenceCriteria::getClassSig());

namespace {

void warn(const std::string &vSite, const std::string &vWhat, int vStatus) {
    hiveEventLogger::hiveOutputWarning(vSite, vWhat + ": " + pcr_strerror(vStatus) + " -- " + pcr_last_error());
}

void emptyPairSet(CorrespondencePairSet &voPairs) {
    voPairs.first = makeSharedPtr<CBasePointCloudSubset>(new CPntNormSubset);
    voPairs.second = makeSharedPtr<CBasePointCloudSubset>(new CPntNormSubset);
}

void flatten(const VecColVector3 &vSet, std::vector<double> &voFlat) {
    voFlat.resize(vSet.size() * 3);
    for (size_t i = 0; i < vSet.size(); ++i) {
        voFlat[3 * i] = vSet[i][0];
        voFlat[3 * i + 1] = vSet[i][1];
        voFlat[3 * i + 2] = vSet[i][2];
    }
}

// contiguous split of [0, m) over the ranks, sizes differing by at most one (same rule as dist.shard_range)
void shardRange(unsigned int vM, unsigned int vRank, unsigned int vWorld, unsigned int &voBegin, unsigned int &voCount) {
    const unsigned int Base = vM / vWorld, Extra = vM % vWorld;
    voBegin = vRank * Base + (vRank < Extra ? vRank : Extra);
    voCount = Base + (vRank < Extra ? 1u : 0u);
}

}  // namespace

// ---- CUniqueData -------------------------------------------------------------------------------------------

void CUniqueData::setPointCloud(const hivePointCloud::CPointCloud &vPointCloud) {
    CDeviceSession *pSession = CDeviceSession::getInstance();
    const SDeviceCloud *pCloud = pSession->acquire(&vPointCloud);
    m_UnitSquareDist = std::numeric_limits<double>::quiet_NaN();
    if (!pCloud) {
        hiveEventLogger::hiveOutputWarning(__EXCEPTION_SITE__, "CUniqueData: " + pSession->lastError());
        return;
    }
    const int Status = pcr_unique_sq_dist(pSession->context(), pCloud->pIndex, &m_UnitSquareDist);
    if (Status != PCR_OK) warn(__EXCEPTION_SITE__, "pcr_unique_sq_dist", Status);
}

// ---- CSamplerIndexList ---------------------------------------------------------------------------------------

CorrespondencePointSet CSamplerIndexList::_doSampleV(const hivePointCloud::CPointCloud &vPointCloud) {
    std::vector<unsigned int> Indices = m_IndexList;
    if (Indices.empty()) {
        Indices.resize(vPointCloud.getNumPoints());
        for (unsigned int i = 0; i < Indices.size(); ++i) Indices[i] = i;
    }
    return makeSharedPtr<CBasePointCloudSubset>(new CIndexSubset(vPointCloud, Indices));
}

CorrespondencePointSet CSamplerRandomGPU::_doSampleV(const hivePointCloud::CPointCloud &vPointCloud) {
    int SampleNum = -1;
    double SampleRatio = -1.0;
    unsigned long long Seed = 0;
    PTR_CONTROL_PARAMS->dumpValue(getKeySampleNum(), SampleNum);
    PTR_CONTROL_PARAMS->dumpValue(getKeySampleRatio(), SampleRatio);
    PTR_CONTROL_PARAMS->dumpValue(getKeySeed(), Seed);
    const int NumPnt = (int)vPointCloud.getNumPoints();
    double Ratio;
    if (SampleRatio > 0.0 && SampleRatio < 1.0) Ratio = SampleRatio;  // SamplerRandom.cpp:31-42
    else if (SampleNum > 0 && SampleNum < NumPnt) Ratio = 1.0 * SampleNum / NumPnt;
    else return CorrespondencePointSet();  // the reference returns a null subset here
    CDeviceSession *pSession = CDeviceSession::getInstance();
    if (!pSession->context()) {
        hiveEventLogger::hiveOutputWarning(__EXCEPTION_SITE__, "no device context: " + pSession->lastError());
        return CorrespondencePointSet();
    }
    std::vector<unsigned int> Indices(NumPnt);
    uint32_t Count = 0;
    int UseRand = 0;
    PTR_CONTROL_PARAMS->dumpValue(getKeyUseRand(), UseRand);
    int Status;
    if (UseRand) {
        std::vector<int32_t> Draws(NumPnt);
        for (int i = 0; i < NumPnt; ++i) Draws[i] = rand();  // ICP_RAND's rand(), once per point, in order (ICPMacros.h:4)
        Status = pcr_sample_random_draws(pSession->context(), (uint32_t)NumPnt, Ratio, Draws.data(), (uint32_t)RAND_MAX,
                                         (uint32_t)NumPnt, Indices.data(), &Count);
    } else {
        Status = pcr_sample_random(pSession->context(), (uint32_t)NumPnt, Ratio, Seed, (uint32_t)NumPnt, Indices.data(), &Count);
    }
    if (Status != PCR_OK) {
        warn(__EXCEPTION_SITE__, "pcr_sample_random", Status);
        return CorrespondencePointSet();
    }
    Indices.resize(Count);
    return makeSharedPtr<CBasePointCloudSubset>(new CIndexSubset(vPointCloud, Indices));
}

CorrespondencePointSet CSamplerUniformGridGPU::_doSampleV(const hivePointCloud::CPointCloud &vPointCloud) {
    CDeviceSession *pSession = CDeviceSession::getInstance();
    const SDeviceCloud *pCloud = pSession->acquire(&vPointCloud);
    if (!pCloud) {
        hiveEventLogger::hiveOutputWarning(__EXCEPTION_SITE__, "no device index: " + pSession->lastError());
        return CorrespondencePointSet();
    }
    std::vector<unsigned int> Indices(pCloud->NumPoints);
    uint32_t Count = 0;
    const int Status = pcr_sample_grid(pSession->context(), pCloud->pIndex, pCloud->NumPoints, Indices.data(), &Count);
    if (Status != PCR_OK) {
        warn(__EXCEPTION_SITE__, "pcr_sample_grid", Status);
        return CorrespondencePointSet();
    }
    Indices.resize(Count);
    return makeSharedPtr<CBasePointCloudSubset>(new CIndexSubset(vPointCloud, Indices));
}

// ---- device pair sets ----------------------------------------------------------------------------------------

void SDeviceStepResult::materialise() {
    if (Materialised) return;
    Materialised = true;
    CDeviceSession *pSession = CDeviceSession::getInstance();
    const SDeviceCloud *pSrc = pSession->acquire(pSource), *pTgt = pSession->acquire(pTarget);
    if (!pSrc || !pTgt) {
        hiveEventLogger::hiveOutputWarning(__EXCEPTION_SITE__, "cannot materialise pair set: " + pSession->lastError());
        return;
    }
    const bool All = SampleIdx.empty();
    const unsigned int M = All ? pSrc->NumPoints : (unsigned int)SampleIdx.size();
    std::vector<unsigned int> Fwd(M);
    std::vector<unsigned char> Keep(M);
    double Again[17];
    const int Status =
        NormalFit ? pcr_icp_step_normal_fit(pSession->context(), pSrc->pIndex, pTgt->pIndex, All ? NULL : SampleIdx.data(), M,
                                            R.data(), T.data(), K, NVDT, NDT, CompareNormals, Again, Fwd.data(), Keep.data())
                  : pcr_icp_step(pSession->context(), pSrc->pIndex, pTgt->pIndex, All ? NULL : SampleIdx.data(), M, R.data(),
                                 T.data(), K, ThrFwd, ThrRev, 0u, Again, Fwd.data(), NULL, Keep.data());
    if (Status != PCR_OK) {
        warn(__EXCEPTION_SITE__, "pcr_icp_step (materialise)", Status);
        return;
    }
    const double *pSrcPos = pSource->getPosPointer(), *pSrcNormal = pSource->getNormalPointer();
    const double *pTgtPos = pTarget->getPosPointer(), *pTgtNormal = pTarget->getNormalPointer();
    const size_t Kept = (size_t)Again[0];
    SrcPoints.reserve(Kept);
    TgtPoints.reserve(Kept);
    for (unsigned int i = 0; i < M; ++i) {
        if (!Keep[i]) continue;
        const size_t s = (size_t)(All ? i : SampleIdx[i]) * 3, q = (size_t)Fwd[i] * 3;
        SrcPoints.push_back(R * ColVector3(pSrcPos[s], pSrcPos[s + 1], pSrcPos[s + 2]) + T);  // CurSrcPoint (:86)
        TgtPoints.push_back(ColVector3(pTgtPos[q], pTgtPos[q + 1], pTgtPos[q + 2]));          // TargetPoint (:92)
        if (pSrcNormal) SrcNormals.push_back(R * ColVector3(pSrcNormal[s], pSrcNormal[s + 1], pSrcNormal[s + 2]));  // (:87)
        if (pTgtNormal) TgtNormals.push_back(ColVector3(pTgtNormal[q], pTgtNormal[q + 1], pTgtNormal[q + 2]));      // (:91)
    }
}

void CDevicePairSubset::_materializeV() const {
    if (m_Copied) return;
    m_Copied = true;
    m_pResult->materialise();
    m_PointSet = m_IsSourceSide ? m_pResult->SrcPoints : m_pResult->TgtPoints;
    m_NormalSet = m_IsSourceSide ? m_pResult->SrcNormals : m_pResult->TgtNormals;
}

bool hiveRegistration::pairSetMSE(const CorrespondencePairSet &vPairs, double &voMSE, unsigned int &voNumPairs) {
    voMSE = std::numeric_limits<double>::quiet_NaN();
    voNumPairs = 0;
    if (!vPairs.first.get() || !vPairs.second.get()) return false;
    if (const CDevicePairSubset *pDevice = dynamic_cast<const CDevicePairSubset *>(vPairs.first.get())) {
        const SDeviceStepResult &Step = pDevice->getStepResult();
        voNumPairs = (unsigned int)Step.Partials[0];
        if (voNumPairs == 0) return false;
        voMSE = Step.Partials[16] / Step.Partials[0];
        return true;
    }
    const VecColVector3 &Src = vPairs.first->getPointSet(), &Tgt = vPairs.second->getPointSet();
    voNumPairs = (unsigned int)Src.size();
    if (Src.empty() || Src.size() != Tgt.size()) return false;
    double Sum = 0;
    for (size_t i = 0; i != Src.size(); ++i) Sum += (Src[i] - Tgt[i]).squaredNorm();
    voMSE = Sum / (double)Src.size();
    return true;
}

// ---- CCorrespondenceEstimationClosestPointsGPU -----------------------------------------------------------------

CCorrespondenceEstimationClosestPointsGPU::CCorrespondenceEstimationClosestPointsGPU() {
    CControlParameters *pConfig = PTR_CONTROL_PARAMS;
    pConfig->setIfNotExist(getKeyNumNeighbors(), 4);  // CorrespondenceEstimationClosestPoints.cpp:29
    // the reference asserts that SDFT/NDFT exist (:187-190); RegCfg.txt:74-78 ships 36 and 4
    pConfig->setIfNotExist(getKeySquareDistFactorThreshold(), 36.0);
    pConfig->setIfNotExist(getKeyNeighbourDistFactorThreshold(), 4.0);
    pConfig->dumpValue(getKeyNumNeighbors(), m_NumNeighbors);
    pConfig->dumpValue(getKeySquareDistFactorThreshold(), m_SquareDistFactorThreshold);
    pConfig->dumpValue(getKeyNeighbourDistFactorThreshold(), m_NeighbourDistFactorThreshold);
    pConfig->setIfNotExist(getKeyNormalDeviationThreshold(), 0.85);     // (:30)
    pConfig->setIfNotExist(getKeyNormalVectorDistThreshold(), 0.0125);  // (:31)
    pConfig->setIfNotExist(getKeyNormalFit(), 0);
    pConfig->dumpValue(getKeyNormalDeviationThreshold(), m_NormalDeviationThreshold);
    pConfig->dumpValue(getKeyNormalVectorDistThreshold(), m_NormalVectorDistThreshold);
    pConfig->dumpValue(getKeyNormalFit(), m_NormalFit);
    _letFactoryReleaseProduct();
}

void CCorrespondenceEstimationClosestPointsGPU::_setPointCloudV(const hivePointCloud::CPointCloud *vSourcePointCloud,
                                                                const hivePointCloud::CPointCloud *vTargetPointCloud) {
    m_pSourcePointCloud = vSourcePointCloud;
    m_pTargetPointCloud = vTargetPointCloud;
    CDeviceSession *pSession = CDeviceSession::getInstance();
    // ONE index per cloud for the whole fit: the target's replaces __updateKDTree (:121-124), the source's
    // replaces the KD-tree the reference rebuilds on the transformed copy every iteration (:73-78)
    if (!pSession->acquire(vSourcePointCloud) || !pSession->acquire(vTargetPointCloud))
        hiveEventLogger::hiveOutputWarning(__EXCEPTION_SITE__, "device index build failed: " + pSession->lastError());
}

void CCorrespondenceEstimationClosestPointsGPU::_determineCorrespondencesV(const CorrespondencePointSet &vSamples,
                                                                           const Eigen::Matrix3d &vR, const ColVector3 &vT,
                                                                           CorrespondencePairSet &vioCorrespondencePairs) {
    emptyPairSet(vioCorrespondencePairs);  // what a failure below leaves behind (+ a logged warning)
    CDeviceSession *pSession = CDeviceSession::getInstance();
    const SDeviceCloud *pSrc = pSession->acquire(m_pSourcePointCloud), *pTgt = pSession->acquire(m_pTargetPointCloud);
    if (!pSrc || !pTgt) {
        hiveEventLogger::hiveOutputWarning(__EXCEPTION_SITE__, "no device index: " + pSession->lastError());
        if (pSession->world() > 1) {  // the peers are waiting in this iteration's collective: tell them (see below)
            double Poison[17];
            for (int i = 0; i < 17; ++i) Poison[i] = std::numeric_limits<double>::quiet_NaN();
            pSession->allReduce(Poison);
        }
        return;
    }
    // every sampler of the reference returns a CIndexSubset over the sampled cloud (SamplerRandom.cpp:49); the
    // fused step works on source INDICES (a sample's own image seeds its reverse search)
    const CIndexSubset *pIndexSubset = dynamic_cast<const CIndexSubset *>(vSamples.get());
    if (!pIndexSubset || &pIndexSubset->getPointCloud() != m_pSourcePointCloud) {
        hiveEventLogger::hiveOutputWarning(__EXCEPTION_SITE__, "samples must be a CIndexSubset over the source point cloud");
        return;
    }
    CUniqueData *pUniqueData = dynamic_cast<CUniqueData *>(
        hiveCommon::CProductFactoryData::getInstance()->getOrCreateProduct(CUniqueData::getSigTgtUniqueData()));
    const double UniqueSquareDist = pUniqueData->getUniqSquareDist();  // (:56-57)

    std::shared_ptr<SDeviceStepResult> pStep(new SDeviceStepResult);
    pStep->R = vR;
    pStep->T = vT;
    pStep->K = m_NumNeighbors;
    pStep->ThrFwd = UniqueSquareDist * m_SquareDistFactorThreshold;     // Pi2QiSquaredDistThreshold (:82)
    pStep->ThrRev = UniqueSquareDist * m_NeighbourDistFactorThreshold;  // Pi2NeibSquaredDistThreshold (:81)
    pStep->pSource = m_pSourcePointCloud;
    pStep->pTarget = m_pTargetPointCloud;
    const std::vector<unsigned int> &Indices = pIndexSubset->getIndexSet();
    bool Identity = Indices.size() == pSrc->NumPoints;
    for (size_t i = 0; Identity && i < Indices.size(); ++i) Identity = Indices[i] == i;
    if (!Identity) pStep->SampleIdx = Indices;

    const unsigned int World = pSession->world();
    int Status;
    if (m_NormalFit) {  // opt-in: __insertFitPairs decides the pairs (:128-175)
        pStep->NormalFit = true;
        pStep->NVDT = m_NormalVectorDistThreshold;
        pStep->NDT = m_NormalDeviationThreshold;
        pStep->CompareNormals = m_NormalFit >= 2;
        Status = PCR_OK;
        if (!pTgt->NormalsAttached) {
            Status = pcr_cloud_set_normals(pTgt->pCloud, m_pTargetPointCloud->getNormalPointer());
            pTgt->NormalsAttached = Status == PCR_OK && m_pTargetPointCloud->getNormalPointer();
        }
        if (Status == PCR_OK && !pSrc->NormalsAttached) {
            Status = pcr_cloud_set_normals(pSrc->pCloud, m_pSourcePointCloud->getNormalPointer());
            pSrc->NormalsAttached = Status == PCR_OK && m_pSourcePointCloud->getNormalPointer();
        }
        unsigned int Begin = 0, Count = (unsigned int)Indices.size();
        if (World > 1) shardRange((unsigned int)Indices.size(), pSession->rank(), World, Begin, Count);
        if (Status == PCR_OK)
            Status = pcr_icp_step_normal_fit(pSession->context(), pSrc->pIndex, pTgt->pIndex,
                                             (World == 1 && Identity) ? NULL : Indices.data() + Begin, Count, vR.data(), vT.data(),
                                             m_NumNeighbors, pStep->NVDT, pStep->NDT, pStep->CompareNormals, pStep->Partials,
                                             NULL, NULL);
    } else if (World == 1 && Identity) {
        Status = pcr_icp_step(pSession->context(), pSrc->pIndex, pTgt->pIndex, NULL, pSrc->NumPoints, vR.data(), vT.data(),
                              m_NumNeighbors, pStep->ThrFwd, pStep->ThrRev, 0u, pStep->Partials, NULL, NULL, NULL);
    } else {
        unsigned int Begin, Count;
        shardRange((unsigned int)Indices.size(), pSession->rank(), World, Begin, Count);
        Status = pcr_icp_step(pSession->context(), pSrc->pIndex, pTgt->pIndex, Indices.data() + Begin, Count, vR.data(), vT.data(),
                              m_NumNeighbors, pStep->ThrFwd, pStep->ThrRev, 0u, pStep->Partials, NULL, NULL, NULL);
    }
    if (Status != PCR_OK) {
        warn(__EXCEPTION_SITE__, "pcr_icp_step", Status);
        if (World == 1) return;
        // a sharded run: the other ranks are about to enter the collective -- enter it too, with poisoned partials,
        // so that every rank sees the failure and stops in the same iteration instead of blocking for ever
        for (int i = 0; i < 17; ++i) pStep->Partials[i] = std::numeric_limits<double>::quiet_NaN();
    }
    if (pSession->allReduce(pStep->Partials) != 0) {
        hiveEventLogger::hiveOutputWarning(__EXCEPTION_SITE__, "all-reduce of the step partials failed");
        return;
    }
    if (!(pStep->Partials[0] == pStep->Partials[0])) {  // NaN: this rank or a peer failed its share
        hiveEventLogger::hiveOutputWarning(__EXCEPTION_SITE__, "a rank of the sharded step failed; no correspondences this iteration");
        return;
    }
    pcr_icp_pivots(pSrc->pIndex, pTgt->pIndex, vR.data(), vT.data(), pStep->PivotP, pStep->PivotQ);
    vioCorrespondencePairs.first = makeSharedPtr<CBasePointCloudSubset>(new CDevicePairSubset(pStep, true));
    vioCorrespondencePairs.second = makeSharedPtr<CBasePointCloudSubset>(new CDevicePairSubset(pStep, false));
}

// ---- CRejectionCentroidDistGPU -----------------------------------------------------------------------------------

void CRejectionCentroidDistGPU::_rejectInvalidPairsV(const CorrespondencePairSet &vAll, CorrespondencePairSet &voValid) {
    emptyPairSet(voValid);
    if (!vAll.first.get() || !vAll.second.get()) return;
    const VecColVector3 &AllSrc = vAll.first->getPointSet(), &AllTgt = vAll.second->getPointSet();
    if (AllSrc.size() != AllTgt.size()) return;
    if (AllSrc.empty() && !m_Reduce) return;  // (a rank of a sharded run without pairs still takes part in the reductions)
    CDeviceSession *pSession = CDeviceSession::getInstance();
    const SDeviceCloud *pSrc = pSession->acquire(m_pSourcePointCloud), *pTgt = pSession->acquire(m_pTargetPointCloud);
    if (!pSrc || !pTgt) {
        hiveEventLogger::hiveOutputWarning(__EXCEPTION_SITE__, "no device index: " + pSession->lastError());
        return;
    }
    int NumNeighbors = 25;      // m_DefaultNumNeighbors (RejectionCentroidDist.cpp:14)
    double MaxSearchDist = 8.0; // m_DefaultMaxSearchRadius (:15)
    double CriticalDistFactor = 1.0;  // m_DefaultCriticalDistFactor (:16); the reference reads an uninitialised local if the key is absent (:130-132)
    PTR_CONTROL_PARAMS->dumpValue(getKeyMaxSearchNum(), NumNeighbors);
    PTR_CONTROL_PARAMS->dumpValue(getKeyMaxSearchDist(), MaxSearchDist);
    PTR_CONTROL_PARAMS->dumpValue(getKeyCriticalDistFactor(), CriticalDistFactor);

    std::vector<double> FlatSrc, FlatTgt;
    flatten(AllSrc, FlatSrc);
    flatten(AllTgt, FlatTgt);
    std::vector<unsigned char> Keep(AllSrc.size());
    const int Status =
        m_Reduce ? pcr_reject_centroid_dist_sharded(pSession->context(), pSrc->pIndex, pTgt->pIndex, FlatSrc.data(), FlatTgt.data(),
                                                    (uint32_t)AllSrc.size(), (uint32_t)NumNeighbors, MaxSearchDist,
                                                    CriticalDistFactor, Keep.data(), NULL, NULL, m_Reduce, m_ReduceUser)
                 : pcr_reject_centroid_dist(pSession->context(), pSrc->pIndex, pTgt->pIndex, FlatSrc.data(), FlatTgt.data(),
                                            (uint32_t)AllSrc.size(), (uint32_t)NumNeighbors, MaxSearchDist, CriticalDistFactor,
                                            Keep.data(), NULL, NULL);
    if (Status != PCR_OK) {
        warn(__EXCEPTION_SITE__, "pcr_reject_centroid_dist", Status);
        return;
    }
    VecColVector3 ValidSrc, ValidTgt;
    for (size_t i = 0; i < Keep.size(); ++i)
        if (Keep[i]) {
            ValidSrc.push_back(AllSrc[i]);
            ValidTgt.push_back(AllTgt[i]);
        }
    // points only: the reference drops the normals here (:151-152)
    static_cast<CPntNormSubset *>(voValid.first.get())->swapPointSet(ValidSrc);
    static_cast<CPntNormSubset *>(voValid.second.get())->swapPointSet(ValidTgt);
}

// ---- CTransformationEstimationSVD2GPU ------------------------------------------------------------------------------

namespace {
bool reducePairLists(const VecColVector3 &vA, const VecColVector3 &vB, double voPartials[17], double voPivotP[3], double voPivotQ[3]) {
    CDeviceSession *pSession = CDeviceSession::getInstance();
    if (!pSession->context()) {
        hiveEventLogger::hiveOutputWarning(__EXCEPTION_SITE__, "no device context: " + pSession->lastError());
        return false;
    }
    if (vA.empty() || vA.size() != vB.size()) return false;
    std::vector<double> FlatA, FlatB;
    flatten(vA, FlatA);
    flatten(vB, FlatB);
    for (int c = 0; c < 3; ++c) {
        voPivotP[c] = vA[0][c];
        voPivotQ[c] = vB[0][c];
    }
    const int Status = pcr_reduce_pairs(pSession->context(), FlatA.data(), FlatB.data(), (uint32_t)vA.size(), voPivotP, voPivotQ, voPartials);
    if (Status != PCR_OK) {
        warn(__EXCEPTION_SITE__, "pcr_reduce_pairs", Status);
        return false;
    }
    return true;
}
}  // namespace

void CTransformationEstimationSVD2GPU::computeRotationAndTranslation(const VecColVector3 &vPntSetA, const VecColVector3 &vPntSetB,
                                                                     Eigen::Matrix3d &voR, ColVector3 &voT) {
    double Partials[17], PivotP[3], PivotQ[3];
    voR = Eigen::Matrix3d::Identity();
    voT.setZero();
    if (!reducePairLists(vPntSetA, vPntSetB, Partials, PivotP, PivotQ)) return;
    // starting from the identity, the accumulated transform IS the increment
    const int Status = pcr_solve_rigid(Partials, PivotP, PivotQ, voR.data(), voT.data(), NULL, NULL, NULL);
    if (Status != PCR_OK) warn(__EXCEPTION_SITE__, "pcr_solve_rigid", Status);
}

void CTransformationEstimationSVD2GPU::_computeRotationAndTranslationV(const CorrespondencePairSet &vPairs, Eigen::Matrix3d &vioR,
                                                                       ColVector3 &vioT) {
    if (!vPairs.first.get() || !vPairs.second.get()) return;
    double Partials[17], PivotP[3], PivotQ[3];
    if (const CDevicePairSubset *pDevice = dynamic_cast<const CDevicePairSubset *>(vPairs.first.get())) {
        const SDeviceStepResult &Step = pDevice->getStepResult();
        std::memcpy(Partials, Step.Partials, sizeof Partials);
        std::memcpy(PivotP, Step.PivotP, sizeof PivotP);
        std::memcpy(PivotQ, Step.PivotQ, sizeof PivotQ);
    } else if (!reducePairLists(vPairs.first->getPointSet(), vPairs.second->getPointSet(), Partials, PivotP, PivotQ)) {
        return;
    }
    // increment from the 3x3 SVD, then accumulateTransformation (BaseTransformationEstimation.h:28-32)
    const int Status = pcr_solve_rigid(Partials, PivotP, PivotQ, vioR.data(), vioT.data(), NULL, NULL, NULL);
    if (Status != PCR_OK) warn(__EXCEPTION_SITE__, "pcr_solve_rigid", Status);  // (R,T) left as they were
}
