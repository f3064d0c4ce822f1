// PoseState.h — mirrors ingvio_estimator/src/PoseState.h:30-243 (SE3, SE23) with the invariant
// retractions of PoseState.cpp:79-88 and :174-186.  Rotations are kept as matrices; the reference
// also caches a quaternion (valueLinearAsQuat), re-derived here on demand.
#pragma once
#include "AuxGammaFunc.h"
#include "VecState.h"

namespace ingvio {

struct Quatd { double w, x, y, z; };
Quatd quatFromRot(const Mat3d& R);
Mat3d rotFromQuat(const Quatd& q);

class SE3 : public Type {
public:
    SE3() : Type(6) { _rot = Mat3d::Identity(); _rot_fej = Mat3d::Identity(); }
    void update(const std::vector<double>& dx) override      // PoseState.cpp:79-88
    {
        const Vec3d dth(dx[idx()], dx[idx() + 1], dx[idx() + 2]), dp(dx[idx() + 3], dx[idx() + 4], dx[idx() + 5]);
        const Mat3d Gamma0 = GammaFunc(dth, 0);
        _rot = Gamma0 * _rot;
        _vec = Gamma0 * _vec + GammaFunc(dth, 1) * dp;
    }
    void setIdentity() override { _rot = Mat3d::Identity(); _vec = Vec3d(); }
    const Mat3d& valueLinearAsMat() const { return _rot; }
    Quatd valueLinearAsQuat() const { return quatFromRot(_rot); }
    const Vec3d& valueTrans() const { return _vec; }
    const Mat3d& fejLinearAsMat() const { return _rot_fej; }
    const Vec3d& fejTrans() const { return _vec_fej; }
    void setValueLinearByMat(const Mat3d& R) { _rot = R; }
    void setValueLinearByQuat(const Quatd& q) { _rot = rotFromQuat(q); }
    void setValueTrans(const Vec3d& p) { _vec = p; }
    void setValue(const Mat3d& R, const Vec3d& p) { _rot = R; _vec = p; }          // setValueByIso
    void setFej(const Mat3d& R, const Vec3d& p) { _rot_fej = R; _vec_fej = p; }    // setFejByIso

protected:
    Mat3d _rot, _rot_fej;
    Vec3d _vec, _vec_fej;
};

class SE23 : public Type {
public:
    SE23() : Type(9) { _rot = Mat3d::Identity(); }
    void update(const std::vector<double>& dx) override      // PoseState.cpp:174-186
    {
        const Vec3d dth(dx[idx()], dx[idx() + 1], dx[idx() + 2]);
        const Vec3d d1(dx[idx() + 3], dx[idx() + 4], dx[idx() + 5]), d2(dx[idx() + 6], dx[idx() + 7], dx[idx() + 8]);
        const Mat3d Gamma0 = GammaFunc(dth, 0), Gamma1 = GammaFunc(dth, 1);
        _rot = Gamma0 * _rot;
        _vec1 = Gamma0 * _vec1 + Gamma1 * d1;
        _vec2 = Gamma0 * _vec2 + Gamma1 * d2;
    }
    void setIdentity() override { _rot = Mat3d::Identity(); _vec1 = Vec3d(); _vec2 = Vec3d(); }
    const Mat3d& valueLinearAsMat() const { return _rot; }
    Quatd valueLinearAsQuat() const { return quatFromRot(_rot); }
    const Vec3d& valueTrans1() const { return _vec1; }
    const Vec3d& valueTrans2() const { return _vec2; }
    void setValueLinearByMat(const Mat3d& R) { _rot = R; }
    void setValueLinearByQuat(const Quatd& q) { _rot = rotFromQuat(q); }
    void setValueTrans1(const Vec3d& p) { _vec1 = p; }
    void setValueTrans2(const Vec3d& v) { _vec2 = v; }

protected:
    Mat3d _rot;
    Vec3d _vec1, _vec2;
};

}  // namespace ingvio
