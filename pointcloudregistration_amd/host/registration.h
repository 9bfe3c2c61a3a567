// hiveRegistration::CRegistartion (Registration.h:15-57, .cpp:22-158) -- the caller of the loop -- and the
// scan-sequence driver the Stanford multi-scan configurations imply (BASELINE configs C2/C4).
//
// The façade keeps the reference's shape: plugin quartets are chosen by the config keys F_SAMPLER,
// F_CORRESPONDENCEESTIMATION, F_CORRESPONDENCEREJECTION, F_TRANSFORMATIONESTIMATION (ICPConstGlobleValue.h:8-21);
// keys that are absent are filled with defaults (__fillICPClassesSig, Registration.cpp:88-126) -- here the GPU
// plugins of the closest-points path.  Coarse registration (C_* keys: extreme-curvature sampling, LCP rejection)
// is outside SURVEY.md section 8: coarseFit returns the initial transform with the message "CoarseNotBuilt".
#pragma once
#include <string>
#include <tuple>
#include <vector>
#include "iterative_fit.h"

namespace hiveRegistration {

const std::string g_FineRegistrationPrefix = "F_";
const std::string g_FineSampler = "F_SAMPLER";
const std::string g_FineCorrespondenceEstimation = "F_CORRESPONDENCEESTIMATION";
const std::string g_FineCorrespondenceRejection = "F_CORRESPONDENCEREJECTION";
const std::string g_FineTransformationEstimation = "F_TRANSFORMATIONESTIMATION";

class CRegistartion {
public:
    typedef std::tuple<Eigen::Matrix3d, ColVector3, std::string, std::string> RotTraCoarseFineFitMsg;
    typedef std::tuple<Eigen::Matrix3d, ColVector3, std::string> RotTraFitMsg;

    // vConfigFile "" = keep whatever CControlParameters already holds (the reference always parses RegCfg.txt)
    explicit CRegistartion(const std::string &vConfigFile = "");

    RotTraCoarseFineFitMsg fit(const hivePointCloud::CPointCloud &vSourcePointCloud, const hivePointCloud::CPointCloud &vTargetPointCloud,
                               const Eigen::Matrix3d &vInitR, ColVector3 &vInitT);
    RotTraFitMsg coarseFit(const hivePointCloud::CPointCloud &vSourcePointCloud, const hivePointCloud::CPointCloud &vTargetPointCloud,
                           const Eigen::Matrix3d &vInitR, ColVector3 &vInitT);
    RotTraFitMsg fineFit(const hivePointCloud::CPointCloud &vSourcePointCloud, const hivePointCloud::CPointCloud &vTargetPointCloud,
                         const Eigen::Matrix3d &vInitR, ColVector3 &vInitT);

    // "F_<sig>" selects the fine-stage plugin, as the reference's setters do (Registration.cpp:162-212)
    void setSampler(const std::string &vPointSamplerSig);
    void setCorrespondenceEstimation(const std::string &vCorrespondenceEsitmationSig);
    void setCorrespondenceRejection(const std::string &vCorrespondenceRejectionSig);
    void setTransformationEstimation(const std::string &vTransformationEsitmationSig);

    unsigned int getNumIterationsOfLastFineFit() const { return m_FineIterativeFit.getNumIterations(); }

    // Sequential pairwise registration of a scan sequence (unit test/FineRegistration/TestFineRegistration.cpp:26-92
    // registers one pair; the Stanford sets are chains of such pairs): scan i+1 is registered to scan i starting from
    // vInitial[i] (identity if absent); voRelative[i] maps scan i+1 into the frame of scan i, voAbsolute[i] into the
    // frame of scan 0 (voAbsolute[0] = identity).  Criteria and sampler state are reset between pairs, which the
    // reference's singletons would not do by themselves.
    struct SPose {
        Eigen::Matrix3d R;
        ColVector3 T;
        std::string Msg;
        unsigned int Iterations;
        SPose() : R(Eigen::Matrix3d::Identity()), Iterations(0) {}
    };
    void registerSequence(const std::vector<const hivePointCloud::CPointCloud *> &vScans, const std::vector<SPose> &vInitial,
                          std::vector<SPose> &voRelative, std::vector<SPose> &voAbsolute);

private:
    std::string m_ConfigFile;
    CIterativeFit m_FineIterativeFit;

    void __parseConfigFile();
    void __fillICPClassesSig();
    void __setFineFitClasses();
    void __setPrefixed(const std::string &vKey, const std::string &vSig);
};

}  // namespace hiveRegistration
