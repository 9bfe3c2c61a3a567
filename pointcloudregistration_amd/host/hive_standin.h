// Stand-ins for the pieces of the closed, un-vendored Hive framework that the reference's registration
// plugins are written against (BaseProduct.h, ProductFactory.h, ProductFactoryData.h, EventLoggerInterface.h
// -- none of them is in the reference repository).  They exist so that this host layer builds and is
// testable on its own; in a real integration the reference's own Hive headers take their place and the
// plugin classes in gpu_stages.h register with the real factory unchanged.
//
// Behaviour mirrored from the call sites in the reference:
//   * every concrete plugin defines `static std::string getClassSig()` and a file-scope
//     `hiveCommon::CProductFactory<T> TheCreator(T::getClassSig());`
//     (FineRegistrationClosestPoints/CorrespondenceEstimationClosestPoints.cpp:24)
//   * consumers call `CProductFactoryData::getInstance()->getOrCreateProduct(sig)` and dynamic_cast the
//     result to a stage base class (IterativeFit.cpp:36-68); one object per signature, owned by the factory
//   * constructors call `_letFactoryReleaseProduct()` (CorrespondenceEstimationClosestPoints.cpp:35)
#pragma once
#include <cstdio>
#include <map>
#include <memory>
#include <string>

namespace hiveCommon {

class CBaseProduct {
public:
    virtual ~CBaseProduct() {}

protected:
    CBaseProduct() : m_FactoryOwned(false) {}
    void _letFactoryReleaseProduct() { m_FactoryOwned = true; }

private:
    bool m_FactoryOwned;
};

class CBaseProductFactory {
public:
    virtual ~CBaseProductFactory() {}
    virtual CBaseProduct *createProductV() const = 0;
};

class CProductFactoryData {
public:
    static CProductFactoryData *getInstance() {
        static CProductFactoryData s_Instance;
        return &s_Instance;
    }
    void registerFactory(const std::string &vSig, CBaseProductFactory *vFactory) { m_Creators[vSig] = vFactory; }
    bool isRegistered(const std::string &vSig) const { return m_Creators.count(vSig) != 0; }
    // NULL for an unknown signature (callers _ASSERT on the dynamic_cast result, IterativeFit.cpp:37)
    CBaseProduct *getOrCreateProduct(const std::string &vSig) {
        std::map<std::string, CBaseProduct *>::iterator Itr = m_Products.find(vSig);
        if (Itr != m_Products.end()) return Itr->second;
        std::map<std::string, CBaseProductFactory *>::iterator Creator = m_Creators.find(vSig);
        if (Creator == m_Creators.end()) return NULL;
        CBaseProduct *pProduct = Creator->second->createProductV();
        m_Products[vSig] = pProduct;
        return pProduct;
    }
    // Not in Hive: drops every product so that a test (or the next pair of scans) starts from fresh
    // singletons -- the reference keeps criteria state across fits (checkpoint row).
    void releaseAllProducts() {
        for (std::map<std::string, CBaseProduct *>::iterator Itr = m_Products.begin(); Itr != m_Products.end(); ++Itr)
            delete Itr->second;
        m_Products.clear();
    }
    void releaseProduct(const std::string &vSig) {
        std::map<std::string, CBaseProduct *>::iterator Itr = m_Products.find(vSig);
        if (Itr == m_Products.end()) return;
        delete Itr->second;
        m_Products.erase(Itr);
    }

private:
    std::map<std::string, CBaseProductFactory *> m_Creators;
    std::map<std::string, CBaseProduct *> m_Products;
};

template <typename TProduct>
class CProductFactory : public CBaseProductFactory {
public:
    explicit CProductFactory(const std::string &vSig) { CProductFactoryData::getInstance()->registerFactory(vSig, this); }
    virtual CBaseProduct *createProductV() const { return TProduct::_createSelf(); }
};

}  // namespace hiveCommon

// Products have protected destructors in the reference and are created only by the factory; this macro gives
// the factory its way in without making the constructor/destructor pair public API.
#define PCR_FACTORY_CREATABLE(T) \
public:                          \
    static hiveCommon::CBaseProduct *_createSelf() { return new T; }

namespace hiveEventLogger {
// EventLoggerInterface.h stand-ins: the reference logs through these (BaseSearch.cpp:67-75).
inline bool &quiet() {
    static bool s_Quiet = true;
    return s_Quiet;
}
inline void hiveOutputEvent(const std::string &vMsg) {
    if (!quiet()) std::fprintf(stderr, "[event] %s\n", vMsg.c_str());
}
inline void hiveOutputWarning(const std::string &vSite, const std::string &vMsg) {
    std::fprintf(stderr, "[warning] %s: %s\n", vSite.c_str(), vMsg.c_str());
}
}  // namespace hiveEventLogger

#define __EXCEPTION_SITE__ (std::string(__FILE__) + ":" + std::to_string(__LINE__))
