// C entry points over the C++ plugin layer, for the Python test-suite and for non-C++ callers
// (declared in include/pcr_host.h).  Everything here drives the classes exactly as the reference's own
// callers do: plugins are looked up by signature in the factory, configured through CControlParameters, and
// run by CIterativeFit (Registration.cpp:120-158, unit test/FineRegistration/TestFineRegistration.cpp:26-92).
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "iterative_fit.h"
#include "pcr_host.h"
#include "ply_io.h"
#include "registration.h"
#include "search_plugin.h"

using namespace hiveRegistration;

namespace {

struct STrajectory {
    double *pRows;
    uint32_t Cap, Count;
};

void recordIteration(unsigned int vIteration, const double *vR9, const double *vT3, double vMSE, unsigned int vNumPairs, void *vUser) {
    STrajectory *pTraj = static_cast<STrajectory *>(vUser);
    if (pTraj->pRows && vIteration <= pTraj->Cap) {
        double *pRow = pTraj->pRows + (size_t)(vIteration - 1) * 14;
        std::memcpy(pRow, vR9, 72);
        std::memcpy(pRow + 9, vT3, 24);
        pRow[12] = vMSE;
        pRow[13] = (double)vNumPairs;
    }
    pTraj->Count = vIteration;
}

void fillCloud(hivePointCloud::CPointCloud &voCloud, const double *vPos, const double *vNormal, uint32_t vN) {
    voCloud.setPos(vPos, vN);
    if (vNormal) voCloud.setNormal(vNormal, vN);
}

void copyMsg(const std::string &vMsg, char *voBuf, uint32_t vCap) {
    if (!voBuf || vCap == 0) return;
    std::snprintf(voBuf, vCap, "%s", vMsg.c_str());
}

VecColVector3 unflatten(const double *vFlat, uint32_t vN) {
    VecColVector3 Set(vN);
    for (uint32_t i = 0; i < vN; ++i) Set[i] = ColVector3(vFlat[3 * i], vFlat[3 * i + 1], vFlat[3 * i + 2]);
    return Set;
}

}  // namespace

extern "C" {

void pcrh_reset(void) {
    hiveCommon::CProductFactoryData::getInstance()->releaseAllProducts();
    pcrhost::CDeviceSession::getInstance()->releaseAll();
}

void pcrh_set_device(int device) { pcrhost::CDeviceSession::getInstance()->setDevice(device); }
void pcrh_set_compact_tables(uint32_t mode) { pcrhost::CDeviceSession::getInstance()->setCompactTables(mode); }

void pcrh_set_verbose(int on) { hiveEventLogger::quiet() = !on; }

void pcrh_set_sharding(uint32_t rank, uint32_t world, pcrh_allreduce_fn fn, void *user) {
    pcrhost::CDeviceSession::getInstance()->setSharding(rank, world, fn, user);
}

int pcrh_config_parse_file(const char *path) { return PTR_CONTROL_PARAMS->parseConfig(path ? path : "") ? 0 : -1; }

void pcrh_config_parse_text(const char *text) {
    if (text) PTR_CONTROL_PARAMS->parseText(text);
}

void pcrh_config_set(const char *key, const char *value) { PTR_CONTROL_PARAMS->set(std::string(key), std::string(value)); }

int pcrh_config_get(const char *key, char *buf, uint32_t cap) {
    std::string Value;
    if (!PTR_CONTROL_PARAMS->dumpValue(std::string(key), Value)) return -1;
    copyMsg(Value, buf, cap);
    return 0;
}

int pcrh_is_registered(const char *sig) {
    return hiveCommon::CProductFactoryData::getInstance()->isRegistered(toUpperCopy(sig)) ? 1 : 0;
}

int pcrh_fit(const double *src, const double *src_normals, uint32_t ns, const double *tgt, const double *tgt_normals, uint32_t nt,
             const uint32_t *sample_idx, uint32_t m, const char *sampler_sig, const char *estimation_sig, const char *rejection_sig,
             const char *transformation_sig, double R[9], double T[3], double *traj, uint32_t traj_cap, uint32_t *iterations,
             char *msg, uint32_t msg_cap) {
    if (!src || !tgt || !R || !T) return -1;
    hivePointCloud::CPointCloud Source, Target;
    fillCloud(Source, src, src_normals, ns);
    fillCloud(Target, tgt, tgt_normals, nt);

    CIterativeFit Fit;
    const std::string SamplerSig = sampler_sig ? sampler_sig : CSamplerIndexList::getClassSig();
    Fit.setFitClasses(SamplerSig, estimation_sig ? estimation_sig : CCorrespondenceEstimationClosestPointsGPU::getClassSig(),
                      rejection_sig ? rejection_sig : CCorrespondenceRejectionClosestPoints::getClassSig(),
                      transformation_sig ? transformation_sig : CTransformationEstimationSVD2GPU::getClassSig());
    if (!Fit.isComplete()) {
        copyMsg("unknown plugin signature", msg, msg_cap);
        return -2;
    }
    if (CSamplerIndexList *pSampler = dynamic_cast<CSamplerIndexList *>(
            hiveCommon::CProductFactoryData::getInstance()->getOrCreateProduct(toUpperCopy(SamplerSig))))
        pSampler->setIndexList(sample_idx ? std::vector<unsigned int>(sample_idx, sample_idx + m) : std::vector<unsigned int>());

    STrajectory Traj = {traj, traj_cap, 0};
    Fit.setIterationObserver(recordIteration, &Traj);
    Eigen::Matrix3d InitR;
    ColVector3 InitT;
    std::memcpy(InitR.data(), R, 72);
    std::memcpy(InitT.data(), T, 24);
    std::tuple<Eigen::Matrix3d, ColVector3, std::string> Result = Fit.fit(Source, Target, InitR, InitT);
    std::memcpy(R, std::get<0>(Result).data(), 72);
    std::memcpy(T, std::get<1>(Result).data(), 24);
    if (iterations) *iterations = Traj.Count;
    copyMsg(std::get<2>(Result), msg, msg_cap);
    // the clouds die with this frame: nothing keyed by their addresses may outlive them
    pcrhost::CDeviceSession::getInstance()->release(&Source);
    pcrhost::CDeviceSession::getInstance()->release(&Target);
    hiveCommon::CProductFactoryData::getInstance()->releaseProduct(CCorrespondenceEstimationClosestPointsGPU::getClassSig());
    hiveCommon::CProductFactoryData::getInstance()->releaseProduct(CRejectionCentroidDistGPU::getClassSig());
    hiveCommon::CProductFactoryData::getInstance()->releaseProduct(CCorrespondenceRejectionClosestPoints::getClassSig());
    return 0;
}

int pcrh_correspondences(const double *src, uint32_t ns, const double *tgt, uint32_t nt, const uint32_t *sample_idx, uint32_t m,
                         const double R[9], const double T[3], const char *rejection_sig, double *out_pair_p, double *out_pair_q,
                         uint32_t cap, uint32_t *out_pairs) {
    hivePointCloud::CPointCloud Source, Target;
    fillCloud(Source, src, NULL, ns);
    fillCloud(Target, tgt, NULL, nt);
    hiveCommon::CProductFactoryData *pFactory = hiveCommon::CProductFactoryData::getInstance();
    CUniqueData *pUnique = dynamic_cast<CUniqueData *>(pFactory->getOrCreateProduct(CUniqueData::getSigTgtUniqueData()));
    pUnique->setPointCloud(Target);
    CBaseCorrespondenceEstimation *pEstimation = dynamic_cast<CBaseCorrespondenceEstimation *>(
        pFactory->getOrCreateProduct(CCorrespondenceEstimationClosestPointsGPU::getClassSig()));
    CBaseCorrespondenceRejection *pRejection = dynamic_cast<CBaseCorrespondenceRejection *>(
        pFactory->getOrCreateProduct(toUpperCopy(rejection_sig ? rejection_sig : CCorrespondenceRejectionClosestPoints::getClassSig())));
    if (!pEstimation || !pRejection) return -2;
    pEstimation->setPointCloud(&Source, &Target);
    pRejection->setPointCloud(&Source, &Target);
    std::vector<unsigned int> Indices;
    if (sample_idx) Indices.assign(sample_idx, sample_idx + m);
    else {
        Indices.resize(ns);
        for (uint32_t i = 0; i < ns; ++i) Indices[i] = i;
    }
    CorrespondencePointSet Samples = makeSharedPtr<CBasePointCloudSubset>(new CIndexSubset(Source, Indices));
    Eigen::Matrix3d Rot;
    ColVector3 Tra;
    std::memcpy(Rot.data(), R, 72);
    std::memcpy(Tra.data(), T, 24);
    CorrespondencePairSet All, Valid;
    pEstimation->determineCorrespondences(Samples, Rot, Tra, All);
    pRejection->rejectInvalidPairs(All, Valid);
    const VecColVector3 &PairP = Valid.first->getPointSet(), &PairQ = Valid.second->getPointSet();
    int Status = 0;
    if (out_pairs) *out_pairs = (uint32_t)PairP.size();
    if (PairP.size() > cap || PairP.size() != PairQ.size()) Status = -3;
    else
        for (size_t i = 0; i < PairP.size(); ++i)
            for (int c = 0; c < 3; ++c) {
                if (out_pair_p) out_pair_p[3 * i + c] = PairP[i][c];
                if (out_pair_q) out_pair_q[3 * i + c] = PairQ[i][c];
            }
    All = CorrespondencePairSet();
    Valid = CorrespondencePairSet();
    Samples.reset();
    pcrhost::CDeviceSession::getInstance()->release(&Source);
    pcrhost::CDeviceSession::getInstance()->release(&Target);
    pFactory->releaseProduct(CCorrespondenceEstimationClosestPointsGPU::getClassSig());
    pFactory->releaseProduct(pRejection ? toUpperCopy(rejection_sig ? rejection_sig : CCorrespondenceRejectionClosestPoints::getClassSig()) : "");
    return Status;
}

int pcrh_svd2(const double *pair_a, const double *pair_b, uint32_t P, double R[9], double T[3]) {
    CTransformationEstimationSVD2GPU *pSVD2 = dynamic_cast<CTransformationEstimationSVD2GPU *>(
        hiveCommon::CProductFactoryData::getInstance()->getOrCreateProduct(CTransformationEstimationSVD2GPU::getClassSig()));
    if (!pSVD2) return -2;
    Eigen::Matrix3d Rot;
    ColVector3 Tra;
    pSVD2->computeRotationAndTranslation(unflatten(pair_a, P), unflatten(pair_b, P), Rot, Tra);
    std::memcpy(R, Rot.data(), 72);
    std::memcpy(T, Tra.data(), 24);
    return 0;
}

int pcrh_search(const double *points, uint32_t n, uint32_t dim, const double *queries, uint32_t nq, uint32_t k, int nearest_first,
                int multi_threaded, uint32_t *out_idx) {
    hiveSearch::CGridSearchGPU Search;
    Search.setNeighborOrder(nearest_first ? hiveSearch::CGridSearchGPU::NEAREST_FIRST : hiveSearch::CGridSearchGPU::FARTHEST_FIRST);
    Search.setInputData(points, n, dim);
    std::vector<unsigned int> *pResult = Search.executeSearch(k, queries, nq, multi_threaded != 0);
    if (!pResult) return -1;  // the reference's failure convention: NULL
    const bool SizeOk = pResult->size() == (size_t)nq * k;
    if (SizeOk && out_idx) std::memcpy(out_idx, pResult->data(), sizeof(uint32_t) * pResult->size());
    delete pResult;  // owned by the caller (BaseSearch.cpp:48-53)
    return SizeOk ? 0 : -3;
}

int pcrh_ply_load(const char *path, int with_normals, uint32_t cap, double *out_xyz, double *out_normals, uint32_t *out_n) {
    hivePointCloud::CPointCloud Cloud;
    if (!path || !hivePointCloud::plyLoadFile(path, Cloud, with_normals ? 3 : 0)) return -1;
    const uint32_t N = Cloud.getNumPoints();
    if (out_n) *out_n = N;
    if (N > cap) return -3;
    if (out_xyz) std::memcpy(out_xyz, Cloud.getPosPointer(), sizeof(double) * 3 * N);
    if (out_normals && Cloud.getNormalPointer()) std::memcpy(out_normals, Cloud.getNormalPointer(), sizeof(double) * 3 * N);
    return 0;
}

int pcrh_ply_save(const char *path, const double *xyz, const double *normals, uint32_t n) {
    if (!path || !xyz) return -1;
    hivePointCloud::CPointCloud Cloud;
    fillCloud(Cloud, xyz, normals, n);
    return export2Ply(path, Cloud) ? 0 : -1;
}

int pcrh_register_sequence(const double *const *scans, const uint32_t *counts, uint32_t num_scans, const double *init_RT,
                           double *out_relative_RT, double *out_absolute_RT, uint32_t *out_iterations, char *msgs, uint32_t msg_stride) {
    if (!scans || !counts || num_scans < 2) return -1;
    std::vector<hivePointCloud::CPointCloud> Clouds(num_scans);
    std::vector<const hivePointCloud::CPointCloud *> Ptrs(num_scans);
    for (uint32_t i = 0; i < num_scans; ++i) {
        Clouds[i].setPos(scans[i], counts[i]);
        Ptrs[i] = &Clouds[i];
    }
    std::vector<CRegistartion::SPose> Init, Rel, Abs;
    if (init_RT)
        for (uint32_t i = 0; i + 1 < num_scans; ++i) {
            CRegistartion::SPose P;
            std::memcpy(P.R.data(), init_RT + (size_t)i * 12, 72);
            std::memcpy(P.T.data(), init_RT + (size_t)i * 12 + 9, 24);
            Init.push_back(P);
        }
    if (CSamplerIndexList *pSampler = dynamic_cast<CSamplerIndexList *>(
            hiveCommon::CProductFactoryData::getInstance()->getOrCreateProduct(CSamplerIndexList::getClassSig())))
        pSampler->setIndexList(std::vector<unsigned int>());  // every point of each source scan
    CRegistartion Registration;
    Registration.registerSequence(Ptrs, Init, Rel, Abs);
    for (uint32_t i = 0; i + 1 < num_scans && i < Rel.size(); ++i) {
        if (out_relative_RT) {
            std::memcpy(out_relative_RT + (size_t)i * 12, Rel[i].R.data(), 72);
            std::memcpy(out_relative_RT + (size_t)i * 12 + 9, Rel[i].T.data(), 24);
        }
        if (out_absolute_RT) {
            std::memcpy(out_absolute_RT + (size_t)i * 12, Abs[i + 1].R.data(), 72);
            std::memcpy(out_absolute_RT + (size_t)i * 12 + 9, Abs[i + 1].T.data(), 24);
        }
        if (out_iterations) out_iterations[i] = Rel[i].Iterations;
        if (msgs && msg_stride) copyMsg(Rel[i].Msg, msgs + (size_t)i * msg_stride, msg_stride);
    }
    hiveCommon::CProductFactoryData::getInstance()->releaseProduct(CCorrespondenceEstimationClosestPointsGPU::getClassSig());
    hiveCommon::CProductFactoryData::getInstance()->releaseProduct(CRejectionCentroidDistGPU::getClassSig());
    hiveCommon::CProductFactoryData::getInstance()->releaseProduct(CCorrespondenceRejectionClosestPoints::getClassSig());
    return Rel.size() + 1 == num_scans ? 0 : -2;
}

/* host-only: CDefaultConvergenceCriteria driven with explicit pair lists */
void pcrh_conv_reset(double target_unique_sq_dist) {
    hiveCommon::CProductFactoryData *pFactory = hiveCommon::CProductFactoryData::getInstance();
    dynamic_cast<CUniqueData *>(pFactory->getOrCreateProduct(CUniqueData::getSigTgtUniqueData()))->setUniqSquareDist(target_unique_sq_dist);
    dynamic_cast<CDefaultConvergenceCriteria *>(pFactory->getOrCreateProduct(CDefaultConvergenceCriteria::getClassSig()))->reset();
}

int pcrh_conv_step(const double R[9], const double T[3], uint32_t iteration, const double *pair_p, const double *pair_q, uint32_t P,
                   int *state, double *mse, char *msg, uint32_t msg_cap) {
    CDefaultConvergenceCriteria *pCriteria = dynamic_cast<CDefaultConvergenceCriteria *>(
        hiveCommon::CProductFactoryData::getInstance()->getOrCreateProduct(CDefaultConvergenceCriteria::getClassSig()));
    CorrespondencePairSet Pairs;
    CPntNormSubset *pFirst = new CPntNormSubset, *pSecond = new CPntNormSubset;
    Pairs.first = makeSharedPtr<CBasePointCloudSubset>(pFirst);
    Pairs.second = makeSharedPtr<CBasePointCloudSubset>(pSecond);
    pFirst->setPointSet(unflatten(pair_p, P));
    pSecond->setPointSet(unflatten(pair_q, P));
    Eigen::Matrix3d Rot;
    ColVector3 Tra;
    std::memcpy(Rot.data(), R, 72);
    std::memcpy(Tra.data(), T, 24);
    pCriteria->setCorrespondencePairSet(Pairs);
    const bool Converged = pCriteria->isConverged(Rot, Tra, iteration);
    int State = 0;
    PTR_CONTROL_PARAMS->dumpValue(CBaseConvergenceCriteria::getKeyConvergenceState(), State);
    if (state) *state = State;
    if (mse) *mse = pCriteria->getCurrentMSE();
    copyMsg(pCriteria->getConvergenceMsg(), msg, msg_cap);
    return Converged ? 1 : 0;
}

}  // extern "C"
