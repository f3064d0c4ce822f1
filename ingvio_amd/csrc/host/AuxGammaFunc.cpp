// AuxGammaFunc.cpp — host shim restatement of ingvio_estimator/src/AuxGammaFunc.cpp:28-225
// (closed-form Gamma_m and Psi_1/Psi_2 of the invariant IMU error propagation).
#include "AuxGammaFunc.h"

#include <cassert>

namespace ingvio {

Mat3d skew(const Vec3d& v)      // AuxGammaFunc.cpp:28-35
{
    Mat3d r;
    r(0, 1) = -v.z(); r(0, 2) = v.y();
    r(1, 0) = v.z();  r(1, 2) = -v.x();
    r(2, 0) = -v.y(); r(2, 1) = v.x();
    return r;
}

Vec3d vee(const Mat3d& mat) { return Vec3d(mat(2, 1), mat(0, 2), mat(1, 0)); }      // :37-44

Mat3d GammaFunc(const Vec3d& vec, int m)      // :46-113
{
    assert(m >= 0 && m <= 3);
    const double theta = vec.norm();
    if (std::fabs(theta) < 1e-06) {
        const double factor = m == 3 ? 1.0 / 6.0 : (m == 2 ? 0.5 : 1.0);
        return factor * Mat3d::Identity();
    }
    const Mat3d n_cross = skew(vec * (1.0 / theta));
    const Mat3d n_cross2 = n_cross * n_cross;
    const double s = std::sin(theta), c = std::cos(theta);
    double f0, f1, f2;
    switch (m) {
    case 1: f0 = 1.0; f1 = (1.0 - c) / theta; f2 = (theta - s) / theta; break;
    case 2: f0 = 0.5; f1 = (theta - s) / std::pow(theta, 2); f2 = (std::pow(theta, 2) + 2.0 * c - 2.0) / (2.0 * std::pow(theta, 2)); break;
    case 3: {
        const double t3 = std::pow(theta, 3);
        f0 = 1.0 / 6.0; f1 = (std::pow(theta, 2) + 2.0 * c - 2.0) / (2.0 * t3); f2 = (t3 - 6.0 * theta + 6.0 * s) / (6.0 * t3);
        break;
    }
    default: f0 = 1.0; f1 = s; f2 = 1.0 - c; break;
    }
    return f0 * Mat3d::Identity() + f1 * n_cross + f2 * n_cross2;
}

namespace {
struct SkewProducts { Mat3d WA, WAW, WAW2, W2A, W2AW, W2AW2; };
SkewProducts products(const Vec3d& w, const Vec3d& a)      // :123-133 / :177-187
{
    SkewProducts p;
    const Mat3d W = skew(w);
    p.WA = W * skew(a);
    p.WAW = p.WA * W;
    p.WAW2 = p.WAW * W;
    p.W2A = W * p.WA;
    p.W2AW = p.W2A * W;
    p.W2AW2 = p.W2AW * W;
    return p;
}
}  // namespace

Mat3d Psi1Func(const Vec3d& w, const Vec3d& a, double dt)      // :115-166
{
    if ((w * dt).norm() < 1e-08) return Mat3d::Zero();
    const Mat3d M1 = skew(a) * GammaFunc(-(w * dt), 2) * std::pow(dt, 2.0);
    const SkewProducts p = products(w, a);
    const double eta = w.norm(), xi = eta * dt, xi2 = std::pow(xi, 2.0);
    const double sx = std::sin(xi), cx = std::cos(xi), s2 = std::sin(2 * xi), c2 = std::cos(2 * xi);
    const double eta3 = std::pow(eta, 3), eta4 = eta * eta3, eta5 = eta * eta4, eta6 = eta * eta5;
    const double c1 = (sx - xi * cx) / eta3;
    const double cc2 = (c2 - 4 * cx + 3) / (4 * eta4);
    const double c3 = (4 * sx + s2 - 4 * xi * cx - 2 * xi) / (4 * eta5);
    const double c4 = (xi2 - 2 * xi * sx - 2 * cx + 2) / (2 * eta4);
    const double c5 = (6 * xi - 8 * sx + s2) / (4 * eta5);
    const double c6 = (2 * xi2 - 4 * xi * sx - c2 + 1) / (4 * eta6);
    // the reference multiplies M1 by the bracket (:163); kept as written
    return M1 * (c1 * p.WA + cc2 * p.WAW + c3 * p.WAW2 + c4 * p.W2A + c5 * p.W2AW + c6 * p.W2AW2);
}

Mat3d Psi2Func(const Vec3d& w, const Vec3d& a, double dt)      // :168-225
{
    if ((w * dt).norm() < 1e-07) return Mat3d::Zero();
    const Mat3d M1 = skew(a) * GammaFunc(-(w * dt), 3) * std::pow(dt, 3);
    const SkewProducts p = products(w, a);
    const double eta = w.norm(), xi = eta * dt, xi2 = std::pow(xi, 2.0), xi3 = xi * xi2;
    const double sx = std::sin(xi), cx = std::cos(xi), s2 = std::sin(2 * xi), c2 = std::cos(2 * xi);
    const double eta3 = std::pow(eta, 3), eta4 = eta * eta3, eta5 = eta * eta4, eta6 = eta * eta5, eta7 = eta * eta6;
    const double c1 = (xi * sx + 2 * cx - 2) / eta4;
    const double cc2 = (6 * xi - 8 * sx + s2) / (8 * eta5);
    const double c3 = (2 * xi2 + 8 * xi * sx + 16 * cx + c2 - 17) / (8 * eta6);
    const double c4 = (xi3 + 6 * xi - 12 * sx + 6 * xi * cx) / (6 * eta5);
    const double c5 = (6 * xi2 + 16 * cx - c2 - 15) / (8 * eta6);
    const double c6 = (4 * xi3 + 6 * xi - 24 * sx - 3 * s2 + 24 * xi * cx) / (24 * eta7);
    return M1 * (c1 * p.WA + cc2 * p.WAW + c3 * p.WAW2 + c4 * p.W2A + c5 * p.W2AW + c6 * p.W2AW2);
}

}  // namespace ingvio
