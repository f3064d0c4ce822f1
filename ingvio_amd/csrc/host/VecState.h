// VecState.h — mirrors ingvio_estimator/src/VecState.h:32-134 (Type / Vec3 / Scalar): the
// type-indexed state variables whose (idx, size) address the covariance.  Same names, same
// argument meaning; `dx` is a plain array of curr_cov_size() doubles instead of Eigen::VectorXd.
#pragma once
#include <memory>
#include <vector>

#include "Mat3.h"

namespace ingvio {

class Type {
public:
    explicit Type(int size) { _size = size; }
    virtual ~Type() {}
    virtual void set_cov_idx(int new_cov_idx) { _idx = new_cov_idx; }
    int idx() const { return _idx; }
    int size() const { return _size; }
    virtual void update(const std::vector<double>& dx) = 0;
    virtual void setIdentity() = 0;

protected:
    int _idx = -1;
    int _size = -1;
};

class Vec3 : public Type {
public:
    Vec3() : Type(3) {}
    void update(const std::vector<double>& dx) override      // VecState.cpp:25-29
    {
        for (int i = 0; i < 3; ++i) _vec[i] += dx[idx() + i];
    }
    void setIdentity() override { _vec = Vec3d(); }
    const Vec3d& value() const { return _vec; }
    const Vec3d& fej() const { return _vec_fej; }
    void setValue(const Vec3d& v) { _vec = v; }
    void setFej(const Vec3d& v) { _vec_fej = v; }

protected:
    Vec3d _vec, _vec_fej;
};

class Scalar : public Type {
public:
    Scalar() : Type(1) {}
    void update(const std::vector<double>& dx) override { _scalar += dx[idx()]; }      // VecState.cpp:41-45
    void setIdentity() override { _scalar = 0.0; }
    const double& value() const { return _scalar; }
    void setValue(const double& s) { _scalar = s; }

protected:
    double _scalar = 0.0, _scalar_fej = 0.0;
};

}  // namespace ingvio
