#include "ImuTransition.h"

namespace ingvio {

namespace {
inline void put(double* M, int ld, int r0, int c0, const Mat3d& B)
{
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) M[(size_t)(c0 + j) * ld + r0 + i] = B(i, j);
}
}  // namespace

void imuTransitionAnalytic(Mat3d& R, Vec3d& p, Vec3d& v, const Vec3d& bg, const Vec3d& ba,
                           const Vec3d& gyro_raw, const Vec3d& accel_raw, const Vec3d& gravity, double dt,
                           double Phi[225], double G[180])
{
    std::memset(Phi, 0, 225 * sizeof(double));
    std::memset(G, 0, 180 * sizeof(double));
    for (int i = 0; i < 15; ++i) Phi[i * 15 + i] = 1.0;                    // Phi.setIdentity(), :105

    const Mat3d R_hat = R;
    const Vec3d p_hat = p, v_hat = v;
    const Mat3d I3 = Mat3d::Identity();

    put(G, 15, 0, 0, R_hat);                                               // :112-117
    put(G, 15, 3, 0, skew(p_hat) * R_hat);
    put(G, 15, 6, 0, skew(v_hat) * R_hat);
    put(G, 15, 6, 3, R_hat);
    put(G, 15, 9, 6, I3);
    put(G, 15, 12, 9, I3);

    const Vec3d gyro_unbiased = gyro_raw - bg;                             // :121-122
    const Vec3d acc_unbiased = accel_raw - ba;

    const Mat3d Gamma0 = GammaFunc(dt * gyro_unbiased, 0);                 // :126-128
    const Mat3d Gamma1 = GammaFunc(dt * gyro_unbiased, 1);
    const Mat3d Gamma2 = GammaFunc(dt * gyro_unbiased, 2);

    const Mat3d R_hat_new = R_hat * Gamma0;                                // :130
    const Mat3d RG1 = R_hat * Gamma1, RG2 = R_hat * Gamma2;
    const Vec3d v_hat_new = v_hat + gravity * dt + (RG1 * acc_unbiased) * dt;                                   // :133
    const Vec3d p_hat_new = p_hat + v_hat * dt + 0.5 * gravity * std::pow(dt, 2) + (RG2 * acc_unbiased) * std::pow(dt, 2);   // :136

    put(Phi, 15, 3, 0, 0.5 * skew(gravity) * std::pow(dt, 2));             // :150
    put(Phi, 15, 3, 6, dt * I3);                                           // :151
    put(Phi, 15, 6, 0, skew(gravity) * dt);                                // :152
    put(Phi, 15, 0, 9, -(RG1 * dt));                                       // :154
    put(Phi, 15, 6, 12, -(RG1 * dt));                                      // :155
    put(Phi, 15, 3, 12, -(RG2 * std::pow(dt, 2)));                         // :157
    put(Phi, 15, 6, 9, -(skew(v_hat_new) * RG1 * dt) + R_hat * Psi1Func(gyro_unbiased, acc_unbiased, dt));   // :159
    put(Phi, 15, 3, 9, -(skew(p_hat_new) * RG1 * dt) + R_hat * Psi2Func(gyro_unbiased, acc_unbiased, dt));   // :161

    R = R_hat_new; p = p_hat_new; v = v_hat_new;
}

}  // namespace ingvio
