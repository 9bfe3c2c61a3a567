#include "ply_io.h"
#include <fstream>
#include <iomanip>
#include <vector>

bool hivePointCloud::plyLoadFile(const std::string &vFileName, CPointCloud &voPointCloud, int vNumProp) {
    std::ifstream InFile(vFileName.c_str());
    if (!InFile.is_open()) return false;
    std::string Token;
    while (InFile >> Token && Token != "vertex") {
    }
    unsigned int NumPoints = 0;
    if (!(InFile >> NumPoints) || NumPoints == 0) return false;
    while (InFile >> Token && Token != "end_header") {
    }
    if (Token != "end_header") return false;
    std::vector<double> Pos((size_t)NumPoints * 3), Prop((size_t)NumPoints * (vNumProp > 0 ? vNumProp : 0));
    for (unsigned int i = 0; i < NumPoints; ++i) {
        if (!(InFile >> Pos[(size_t)i * 3] >> Pos[(size_t)i * 3 + 1] >> Pos[(size_t)i * 3 + 2])) return false;
        for (int k = 0; k < vNumProp; ++k)
            if (!(InFile >> Prop[(size_t)i * vNumProp + k])) return false;
    }
    voPointCloud.setPos(Pos.data(), NumPoints);
    if (vNumProp == 3) voPointCloud.setNormal(Prop.data(), NumPoints);
    return true;
}

bool hivePointCloud::plyLoad(const std::string &vFileID, CPointCloud &voPointCloud) {
    return plyLoadFile(vFileID + "Norm.ply", voPointCloud, 3);
}

bool hiveRegistration::export2Ply(const char *vFileName, const hivePointCloud::CPointCloud &vPointCloud) {
    std::ofstream OutFile(vFileName);
    if (!OutFile.is_open()) return false;
    const double *pPos = vPointCloud.getPosPointer(), *pNormal = vPointCloud.getNormalPointer();
    const unsigned int N = vPointCloud.getNumPoints();
    OutFile << "ply\nformat ascii 1.0\ncomment VCGLIB generated\nelement vertex " << N
            << "\nproperty float x\nproperty float y\nproperty float z\n";
    if (pNormal) OutFile << "property float nx\nproperty float ny\nproperty float nz\n";
    OutFile << "element face 0\nproperty list uchar int vertex_indices\nend_header" << std::endl;
    OutFile << std::setprecision(17);
    for (unsigned int i = 0; i < N; ++i) {
        OutFile << pPos[3 * i] << " " << pPos[3 * i + 1] << " " << pPos[3 * i + 2] << " ";
        if (pNormal) OutFile << pNormal[3 * i] << " " << pNormal[3 * i + 1] << " " << pNormal[3 * i + 2];
        OutFile << "\n";
    }
    return OutFile.good();
}
