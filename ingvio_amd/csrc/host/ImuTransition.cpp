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

// The isAnalytic == false branch (ImuPropagator.cpp:163-229).  The reference rotates through quaternions
// (q * AngleAxis(|d|/2), q * AngleAxis(|d|)); here the same two rotations are R * Exp(d/2) and R * Exp(d) on matrices.
// Phi is the reference's I + F dt + F^2 dt^2/2 + F^3 dt^3/6 written out: with the block rows (theta, p, v, bg, ba)
// F has only F[theta,bg] = -R, F[p,v] = I, F[p,bg] = -[p]x R, F[v,theta] = [g]x, F[v,bg] = -[v]x R, F[v,ba] = -R, so that
// F^2 has row p = row v of F and F^2[v,bg] = -[g]x R, F^3 has only F^3[p,bg] = -[g]x R, and F^4 = 0.
void imuTransitionRK4(Mat3d& R, Vec3d& p, Vec3d& v, const Vec3d& bg, const Vec3d& ba,
                      const Vec3d& gyro_raw, const Vec3d& accel_raw, const Vec3d& gravity, double dt,
                      double Phi[225], double G[180])
{
    std::memset(Phi, 0, 225 * sizeof(double));
    std::memset(G, 0, 180 * sizeof(double));
    for (int i = 0; i < 15; ++i) Phi[i * 15 + i] = 1.0;

    const Mat3d R_hat = R;
    const Vec3d p_hat = p, v_hat = v;
    const Mat3d I3 = Mat3d::Identity();
    const Mat3d pxR = skew(p_hat) * R_hat, vxR = skew(v_hat) * R_hat, gx = skew(gravity);

    put(G, 15, 0, 0, R_hat);                                               // :112-117 (shared by both branches)
    put(G, 15, 3, 0, pxR);
    put(G, 15, 6, 0, vxR);
    put(G, 15, 6, 3, R_hat);
    put(G, 15, 9, 6, I3);
    put(G, 15, 12, 9, I3);

    const Vec3d gyro_unbiased = gyro_raw - bg;                             // :165-166
    const Vec3d acc_unbiased = accel_raw - ba;
    const Vec3d delta_angle = dt * gyro_unbiased;                          // :172
    const Mat3d R_half = R_hat * GammaFunc(0.5 * delta_angle, 0);          // :174
    const Mat3d R_full = R_hat * GammaFunc(delta_angle, 0);                // :176

    const Vec3d k1_v = R_hat * acc_unbiased + gravity, k1_p = v_hat;       // :179-180
    const Vec3d k2_v = R_half * acc_unbiased + gravity, k2_p = v_hat + k1_v * (dt / 2.0);   // :183-184
    const Vec3d k3_v = k2_v, k3_p = v_hat + k2_v * (dt / 2.0);             // :187-188 (same mid-point rotation)
    const Vec3d k4_v = R_full * acc_unbiased + gravity, k4_p = v_hat + k3_v * dt;           // :191-192

    const Vec3d v_hat_new = v_hat + (dt / 6.0) * (k1_v + 2.0 * k2_v + 2.0 * k3_v + k4_v);   // :195
    const Vec3d p_hat_new = p_hat + (dt / 6.0) * (k1_p + 2.0 * k2_p + 2.0 * k3_p + k4_p);   // :196

    const double dt2 = dt * dt / 2.0, dt3 = dt * dt * dt / 6.0;
    const Mat3d gxR = gx * R_hat;
    put(Phi, 15, 0, 9, -(R_hat * dt));
    put(Phi, 15, 3, 0, gx * dt2);
    put(Phi, 15, 3, 6, dt * I3);
    put(Phi, 15, 3, 9, -(pxR * dt) - vxR * dt2 - gxR * dt3);
    put(Phi, 15, 3, 12, -(R_hat * dt2));
    put(Phi, 15, 6, 0, gx * dt);
    put(Phi, 15, 6, 9, -(vxR * dt) - gxR * dt2);
    put(Phi, 15, 6, 12, -(R_hat * dt));

    R = R_full; p = p_hat_new; v = v_hat_new;
}

}  // namespace ingvio
