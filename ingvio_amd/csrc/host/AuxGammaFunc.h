// AuxGammaFunc.h — mirrors ingvio_estimator/src/AuxGammaFunc.h:28-38 (same names, same argument meaning).
#pragma once
#include "Mat3.h"

namespace ingvio {
Mat3d skew(const Vec3d& vec);
Vec3d vee(const Mat3d& mat);
Mat3d GammaFunc(const Vec3d& vec, int m = 0);
Mat3d Psi1Func(const Vec3d& tilde_omega, const Vec3d& tilde_acc, double dt);
Mat3d Psi2Func(const Vec3d& tilde_omega, const Vec3d& tilde_acc, double dt);
}  // namespace ingvio
