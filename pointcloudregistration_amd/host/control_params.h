// hiveRegistration::CControlParameters (ControlParameters.h:15-86, .cpp:22-124): the string key/value store
// every plugin reads its thresholds from, loaded from `Key = Value` lines (RegCfg.txt), `/` starting a
// comment (hiveConfig::HIVE_COMMENT_CHAR is not in the repository; RegCfg.txt:1-4 uses `//`).  Keys are
// compared upper-cased, as every getKey*() in the reference upper-cases (CorrespondenceEstimationClosestPoints.h:27-32).
#pragma once
#include <algorithm>
#include <cctype>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include "hive_standin.h"

namespace hiveRegistration {

inline std::string toUpperCopy(std::string vStr) {
    std::transform(vStr.begin(), vStr.end(), vStr.begin(), [](unsigned char c) { return (char)std::toupper(c); });
    return vStr;
}

// hiveLexicalCast (RegUtilityFunctions.h:118-135 wraps boost::lexical_cast and returns false on failure)
template <typename TFrom, typename TTo>
bool hiveLexicalCast(const TFrom &vFrom, TTo &voTo) {
    std::stringstream Stream;
    Stream.precision(17);
    Stream << vFrom;
    Stream >> voTo;
    return !Stream.fail() && (Stream.eof() || Stream.peek() == std::char_traits<char>::eof());
}
template <typename TFrom>
bool hiveLexicalCast(const TFrom &vFrom, std::string &voTo) {
    std::ostringstream Stream;
    Stream.precision(17);
    Stream << vFrom;
    voTo = Stream.str();
    return true;
}
// boost::lexical_cast<bool> accepts only "0"/"1"; chars are stored as their code (the reference stores the
// convergence state as a char, DefaultConvergenceCriterial.cpp:34)
inline bool hiveLexicalCast(const std::string &vFrom, bool &voTo) {
    if (vFrom == "1") voTo = true;
    else if (vFrom == "0") voTo = false;
    else return false;
    return true;
}

class CControlParameters : public hiveCommon::CBaseProduct {
    PCR_FACTORY_CREATABLE(CControlParameters)
public:
    CControlParameters() { _letFactoryReleaseProduct(); }
    virtual ~CControlParameters() {}

    static std::string getClassSig() { return toUpperCopy("RegistrationConfig"); }

    bool parseConfig(const std::string &vConfigFile) {
        if (vConfigFile.empty()) return false;
        std::ifstream Input(vConfigFile.c_str());
        if (!Input.is_open()) {
            hiveEventLogger::hiveOutputWarning(__EXCEPTION_SITE__, "Fail to locate the configuration file [" + vConfigFile + "].");
            return false;
        }
        std::stringstream Buffer;
        Buffer << Input.rdbuf();
        parseText(Buffer.str());
        return true;
    }
    // the same line grammar from memory (tests and the C harness pass config text directly)
    void parseText(const std::string &vText) {
        std::istringstream Input(vText);
        std::string Line;
        while (std::getline(Input, Line)) {
            const size_t Comment = Line.find('/');
            if (Comment != std::string::npos) Line.erase(Comment);
            std::replace(Line.begin(), Line.end(), '\t', ' ');
            const size_t Eq = Line.find('=');
            if (Eq == std::string::npos) continue;
            const std::string Key = trim(Line.substr(0, Eq)), Value = trim(Line.substr(Eq + 1));
            if (!Key.empty()) m_ControlParams[toUpperCopy(Key)] = Value;
        }
    }

    template <typename T>
    void set(const std::string &vKey, const T &vValue) {
        std::string Tmp;
        hiveLexicalCast(vValue, Tmp);
        m_ControlParams[toUpperCopy(vKey)] = Tmp;
    }
    template <typename T>
    void setIfNotExist(const std::string &vKey, const T &vValue) {
        if (m_ControlParams.find(toUpperCopy(vKey)) == m_ControlParams.end()) set(vKey, vValue);
    }
    template <typename T>
    bool dumpValue(const std::string &vKey, T &voResult) const {
        std::map<std::string, std::string>::const_iterator Itr = m_ControlParams.find(toUpperCopy(vKey));
        if (Itr == m_ControlParams.end()) return false;
        return hiveLexicalCast(Itr->second, voResult);
    }
    bool has(const std::string &vKey) const { return m_ControlParams.count(toUpperCopy(vKey)) != 0; }
    void clear() { m_ControlParams.clear(); }

private:
    static std::string trim(const std::string &vStr) {
        const size_t B = vStr.find_first_not_of(" \r\n");
        if (B == std::string::npos) return std::string();
        return vStr.substr(B, vStr.find_last_not_of(" \r\n") - B + 1);
    }
    std::map<std::string, std::string> m_ControlParams;
};

}  // namespace hiveRegistration

// ICPMacros.h:6
#define PTR_CONTROL_PARAMS                                                                                      \
    (dynamic_cast<hiveRegistration::CControlParameters *>(hiveCommon::CProductFactoryData::getInstance()->getOrCreateProduct( \
        hiveRegistration::CControlParameters::getClassSig())))
