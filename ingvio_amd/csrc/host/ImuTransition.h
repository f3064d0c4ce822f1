// ImuTransition.h — the closed-form nominal-state step and error-state transition of
// ImuPropagator::stateAndCovTransition (ingvio_estimator/src/ImuPropagator.cpp:98-162 analytic branch, :163-229 RK4 branch),
// as a pure function so that both the ImuPropagator shim and the C facade share it.
#pragma once
#include "AuxGammaFunc.h"

namespace ingvio {

// Phi: 15x15, G: 15x12, both column-major (what Eigen::Matrix<double,15,15>::data() holds).
// R, p, v are advanced in place.
void imuTransitionAnalytic(Mat3d& R, Vec3d& p, Vec3d& v, const Vec3d& bg, const Vec3d& ba,
                           const Vec3d& gyro_raw, const Vec3d& accel_raw, const Vec3d& gravity, double dt,
                           double Phi[225], double G[180]);

// The isAnalytic == false branch (:163-229): RK4 on p, v with mid-point rotations, Phi = third-order Taylor of F.
void imuTransitionRK4(Mat3d& R, Vec3d& p, Vec3d& v, const Vec3d& bg, const Vec3d& ba,
                      const Vec3d& gyro_raw, const Vec3d& accel_raw, const Vec3d& gravity, double dt,
                      double Phi[225], double G[180]);

}  // namespace ingvio
