// Value types that cross every stage boundary of the reference's registration pipeline (ICPType.h:11-20,
// ICPBaseObject.h:16-31), restated without Eigen/Boost (neither is installed here).  Layouts match what the
// reference hands to raw-pointer code: Matrix3d::data() is 9 doubles COLUMN-major (Eigen default), a
// ColVector3 is 3 contiguous doubles.  Only the operations the ICP path uses are provided, each evaluated
// left to right without FMA contraction (this layer is compiled with -ffp-contract=off) so that host-side
// re-evaluation of R*s + T reproduces the device bit for bit (PointCloud.cpp:368-370 order).
#pragma once
#include <cmath>
#include <cstring>
#include <memory>
#include <utility>
#include <vector>

namespace Eigen {  // same spelling as the reference's code uses, so plugin signatures read identically

struct ColVector3d {
    double v[3];
    ColVector3d() { v[0] = v[1] = v[2] = 0.0; }
    ColVector3d(double x, double y, double z) {
        v[0] = x;
        v[1] = y;
        v[2] = z;
    }
    double *data() { return v; }
    const double *data() const { return v; }
    double &operator()(int r, int = 0) { return v[r]; }
    double operator()(int r, int = 0) const { return v[r]; }
    double &operator[](int r) { return v[r]; }
    double operator[](int r) const { return v[r]; }
    ColVector3d operator+(const ColVector3d &o) const { return ColVector3d(v[0] + o.v[0], v[1] + o.v[1], v[2] + o.v[2]); }
    ColVector3d operator-(const ColVector3d &o) const { return ColVector3d(v[0] - o.v[0], v[1] - o.v[1], v[2] - o.v[2]); }
    double squaredNorm() const { return (v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]; }
    ColVector3d &setZero() {
        v[0] = v[1] = v[2] = 0.0;
        return *this;
    }
};

struct Matrix3d {
    double m[9];  // column-major: (r,c) -> m[c*3+r]
    Matrix3d() { std::memset(m, 0, sizeof m); }
    static Matrix3d Identity() {
        Matrix3d I;
        I.m[0] = I.m[4] = I.m[8] = 1.0;
        return I;
    }
    double *data() { return m; }
    const double *data() const { return m; }
    double &operator()(int r, int c) { return m[c * 3 + r]; }
    double operator()(int r, int c) const { return m[c * 3 + r]; }
    ColVector3d operator*(const ColVector3d &x) const {
        return ColVector3d((m[0] * x.v[0] + m[3] * x.v[1]) + m[6] * x.v[2], (m[1] * x.v[0] + m[4] * x.v[1]) + m[7] * x.v[2],
                           (m[2] * x.v[0] + m[5] * x.v[1]) + m[8] * x.v[2]);
    }
    Matrix3d operator*(const Matrix3d &o) const {
        Matrix3d P;
        for (int c = 0; c < 3; ++c)
            for (int r = 0; r < 3; ++r) P(r, c) = ((*this)(r, 0) * o(0, c) + (*this)(r, 1) * o(1, c)) + (*this)(r, 2) * o(2, c);
        return P;
    }
    Matrix3d operator-(const Matrix3d &o) const {
        Matrix3d D;
        for (int i = 0; i < 9; ++i) D.m[i] = m[i] - o.m[i];
        return D;
    }
    double squaredNorm() const {
        double s = 0;
        for (int i = 0; i < 9; ++i) s += m[i] * m[i];
        return s;
    }
};

}  // namespace Eigen

namespace hiveRegistration {

typedef Eigen::ColVector3d ColVector3;            // ICPType.h:11
typedef std::vector<ColVector3> VecColVector3;    // ICPType.h:14

// ICPBaseObject.h:8-31: objects with protected destructors handed around in shared pointers
class CICPBaseObject {
public:
    void deleteFunc() { _deleteFunc(); }

protected:
    virtual ~CICPBaseObject() {}
    virtual void _deleteFunc() { delete this; }
};
struct BaseObjDeleter {
    void operator()(CICPBaseObject *vPtr) const { vPtr->deleteFunc(); }
};
template <typename T>
std::shared_ptr<T> makeSharedPtr(T *vPtr) {
    return std::shared_ptr<T>(vPtr, BaseObjDeleter());
}

class CBasePointCloudSubset;
typedef std::shared_ptr<CBasePointCloudSubset> SamplePointSet;           // ICPType.h:18
typedef std::shared_ptr<CBasePointCloudSubset> CorrespondencePointSet;   // ICPType.h:19
typedef std::pair<CorrespondencePointSet, CorrespondencePointSet> CorrespondencePairSet;  // ICPType.h:20

}  // namespace hiveRegistration
