// The on-disk format either side of the path: ASCII PLY as the reference reads and writes it.
//   reading  CPLYLoad::__readFilePos (PLYLoad.cpp:76-111): scan tokens up to "vertex", read the count, scan up to
//            "end_header", then per vertex `x y z` followed by NumProp values (3 for the `<id>Norm.ply` file:
//            nx ny nz).  Faces, if any, are ignored.  CPointCloud::plyLoad(id) opens `<id>Norm.ply`
//            (PointCloud.cpp:142-153); curvature side files (`<id>Gss.ply` ...) are outside the ICP path.
//   writing  export2Ply (unit test/TestUnitity.cpp:278-339): the VCGLIB-style header, `x y z nx ny nz` per line
//            (or `x y z` when the cloud has no normals), default ostream precision... here 17 significant digits
//            so a round trip is exact; readers of the reference's files see the same grammar.
#pragma once
#include <string>
#include "point_cloud.h"

namespace hivePointCloud {

// false if the file cannot be opened or holds no vertex element.  vNumProp extra properties per vertex are read
// into the normals (3) or skipped (0).
bool plyLoadFile(const std::string &vFileName, CPointCloud &voPointCloud, int vNumProp = 3);
// CPointCloud::plyLoad convention: `<vFileID>Norm.ply`
bool plyLoad(const std::string &vFileID, CPointCloud &voPointCloud);

}  // namespace hivePointCloud

namespace hiveRegistration {
bool export2Ply(const char *vFileName, const hivePointCloud::CPointCloud &vPointCloud);
}
