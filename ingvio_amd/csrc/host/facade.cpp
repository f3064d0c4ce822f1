// facade.cpp — small extern "C" surface of libingvio_host.so for the Python harness (bench.py,
// tests): lets Python drive the C++ host shim without a C++ test runner.  Not part of the HIP ABI.
#include "ImuTransition.h"
#include "Update.h"

extern "C" {

// ImuPropagator::stateAndCovTransition (ImuPropagator.cpp:98-162), analytic branch.
// R row-major 3x3; Phi/G column-major.
void ingvio_host_imu_transition(double* R, double* p, double* v, const double* bg, const double* ba,
                                const double* gyro, const double* acc, const double* gravity, double dt,
                                double* Phi, double* G)
{
    ingvio::Mat3d Rm(R);
    ingvio::Vec3d pv(p), vv(v);
    ingvio::imuTransitionAnalytic(Rm, pv, vv, ingvio::Vec3d(bg), ingvio::Vec3d(ba), ingvio::Vec3d(gyro),
                                  ingvio::Vec3d(acc), ingvio::Vec3d(gravity), dt, Phi, G);
    for (int i = 0; i < 9; ++i) R[i] = Rm.m[i];
    for (int i = 0; i < 3; ++i) { p[i] = pv[i]; v[i] = vv[i]; }
}

void ingvio_host_gamma(const double* vec, int m, double* out)
{
    const ingvio::Mat3d g = ingvio::GammaFunc(ingvio::Vec3d(vec), m);
    for (int i = 0; i < 9; ++i) out[i] = g.m[i];
}

double ingvio_host_chi2_quantile(int dof, double p) { return ingvio::chi2Quantile(dof, p); }

}  // extern "C"
