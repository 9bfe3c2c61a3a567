#include "registration.h"

using namespace hiveRegistration;

CRegistartion::CRegistartion(const std::string &vConfigFile) : m_ConfigFile(vConfigFile) {
    __parseConfigFile();
    __fillICPClassesSig();
}

void CRegistartion::__parseConfigFile() {
    if (!m_ConfigFile.empty()) PTR_CONTROL_PARAMS->parseConfig(m_ConfigFile);
}

void CRegistartion::__fillICPClassesSig() {
    CControlParameters *pConfig = PTR_CONTROL_PARAMS;
    pConfig->setIfNotExist(g_FineSampler, CSamplerIndexList::getClassSig());
    pConfig->setIfNotExist(g_FineCorrespondenceEstimation, CCorrespondenceEstimationClosestPointsGPU::getClassSig());
    pConfig->setIfNotExist(g_FineCorrespondenceRejection, CCorrespondenceRejectionClosestPoints::getClassSig());
    pConfig->setIfNotExist(g_FineTransformationEstimation, CTransformationEstimationSVD2GPU::getClassSig());
}

void CRegistartion::__setFineFitClasses() {
    std::string Sampler, Estimation, Rejection, Transformation;
    CControlParameters *pConfig = PTR_CONTROL_PARAMS;
    pConfig->dumpValue(g_FineSampler, Sampler);
    pConfig->dumpValue(g_FineCorrespondenceEstimation, Estimation);
    pConfig->dumpValue(g_FineCorrespondenceRejection, Rejection);
    pConfig->dumpValue(g_FineTransformationEstimation, Transformation);
    m_FineIterativeFit.setFitClasses(Sampler, Estimation, Rejection, Transformation);
}

CRegistartion::RotTraFitMsg CRegistartion::coarseFit(const hivePointCloud::CPointCloud &, const hivePointCloud::CPointCloud &,
                                                     const Eigen::Matrix3d &vInitR, ColVector3 &vInitT) {
    return std::make_tuple(vInitR, vInitT, std::string("CoarseNotBuilt"));
}

CRegistartion::RotTraFitMsg CRegistartion::fineFit(const hivePointCloud::CPointCloud &vSourcePointCloud,
                                                   const hivePointCloud::CPointCloud &vTargetPointCloud, const Eigen::Matrix3d &vInitR,
                                                   ColVector3 &vInitT) {
    __parseConfigFile();  // the reference re-reads the file before every fit (Registration.cpp:150)
    __fillICPClassesSig();
    __setFineFitClasses();
    return m_FineIterativeFit.fit(vSourcePointCloud, vTargetPointCloud, vInitR, vInitT);
}

CRegistartion::RotTraCoarseFineFitMsg CRegistartion::fit(const hivePointCloud::CPointCloud &vSourcePointCloud,
                                                         const hivePointCloud::CPointCloud &vTargetPointCloud,
                                                         const Eigen::Matrix3d &vInitR, ColVector3 &vInitT) {
    RotTraFitMsg Coarse = coarseFit(vSourcePointCloud, vTargetPointCloud, vInitR, vInitT);
    RotTraFitMsg Fine = fineFit(vSourcePointCloud, vTargetPointCloud, std::get<0>(Coarse), std::get<1>(Coarse));
    return std::make_tuple(std::get<0>(Fine), std::get<1>(Fine), std::get<2>(Coarse), std::get<2>(Fine));
}

void CRegistartion::__setPrefixed(const std::string &vKey, const std::string &vSig) {
    if (vSig.find(g_FineRegistrationPrefix) == 0) PTR_CONTROL_PARAMS->set(vKey, toUpperCopy(vSig.substr(g_FineRegistrationPrefix.size())));
    // "C_..." would select a coarse-stage plugin: outside the built scope, ignored
}
void CRegistartion::setSampler(const std::string &vSig) { __setPrefixed(g_FineSampler, vSig); }
void CRegistartion::setCorrespondenceEstimation(const std::string &vSig) { __setPrefixed(g_FineCorrespondenceEstimation, vSig); }
void CRegistartion::setCorrespondenceRejection(const std::string &vSig) { __setPrefixed(g_FineCorrespondenceRejection, vSig); }
void CRegistartion::setTransformationEstimation(const std::string &vSig) { __setPrefixed(g_FineTransformationEstimation, vSig); }

void CRegistartion::registerSequence(const std::vector<const hivePointCloud::CPointCloud *> &vScans, const std::vector<SPose> &vInitial,
                                     std::vector<SPose> &voRelative, std::vector<SPose> &voAbsolute) {
    voRelative.clear();
    voAbsolute.clear();
    if (vScans.empty()) return;
    voAbsolute.push_back(SPose());
    hiveCommon::CProductFactoryData *pFactory = hiveCommon::CProductFactoryData::getInstance();
    for (size_t i = 0; i + 1 < vScans.size(); ++i) {
        // fresh per-pair state: the sample subset cached by CIterativeFit and the criteria's carried MSE/rotation
        m_FineIterativeFit.clearSampleSubset();
        if (CDefaultConvergenceCriteria *pCriteria = dynamic_cast<CDefaultConvergenceCriteria *>(
                pFactory->getOrCreateProduct(CDefaultConvergenceCriteria::getClassSig())))
            pCriteria->reset();
        SPose Init = i < vInitial.size() ? vInitial[i] : SPose();
        RotTraFitMsg Fit = fineFit(*vScans[i + 1], *vScans[i], Init.R, Init.T);
        SPose Rel;
        Rel.R = std::get<0>(Fit);
        Rel.T = std::get<1>(Fit);
        Rel.Msg = std::get<2>(Fit);
        Rel.Iterations = m_FineIterativeFit.getNumIterations();
        voRelative.push_back(Rel);
        SPose Abs;  // x_0 = A_i (R x_{i+1} + T)
        Abs.R = voAbsolute[i].R * Rel.R;
        Abs.T = voAbsolute[i].R * Rel.T + voAbsolute[i].T;
        Abs.Msg = Rel.Msg;
        Abs.Iterations = Rel.Iterations;
        voAbsolute.push_back(Abs);
        // the pair's device clouds: scan i is done with, scan i+1 is the next target and stays cached
        pcrhost::CDeviceSession::getInstance()->release(vScans[i]);
    }
    pcrhost::CDeviceSession::getInstance()->release(vScans.back());
}
