// Mat3.h — minimal fixed-size 3-vector / 3x3 matrix types for the ROS-free host shim.
// (The reference uses Eigen::Vector3d / Matrix3d; Eigen is not available in this image, and the
// shim's public signatures use plain double arrays so that Eigen::Map adapters are zero-copy.)
#pragma once
#include <cmath>
#include <cstring>

namespace ingvio {

struct Vec3d {
    double v[3];
    Vec3d() { v[0] = v[1] = v[2] = 0.0; }
    Vec3d(double x, double y, double z) { v[0] = x; v[1] = y; v[2] = z; }
    explicit Vec3d(const double* p) { v[0] = p[0]; v[1] = p[1]; v[2] = p[2]; }
    double& operator[](int i) { return v[i]; }
    double operator[](int i) const { return v[i]; }
    double x() const { return v[0]; }
    double y() const { return v[1]; }
    double z() const { return v[2]; }
    double norm() const { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }
    Vec3d operator+(const Vec3d& o) const { return Vec3d(v[0] + o.v[0], v[1] + o.v[1], v[2] + o.v[2]); }
    Vec3d operator-(const Vec3d& o) const { return Vec3d(v[0] - o.v[0], v[1] - o.v[1], v[2] - o.v[2]); }
    Vec3d operator-() const { return Vec3d(-v[0], -v[1], -v[2]); }
    Vec3d operator*(double s) const { return Vec3d(v[0] * s, v[1] * s, v[2] * s); }
    Vec3d& operator+=(const Vec3d& o) { v[0] += o.v[0]; v[1] += o.v[1]; v[2] += o.v[2]; return *this; }
};
inline Vec3d operator*(double s, const Vec3d& a) { return a * s; }

// row-major 3x3
struct Mat3d {
    double m[9];
    Mat3d() { std::memset(m, 0, sizeof m); }
    explicit Mat3d(const double* p) { std::memcpy(m, p, sizeof m); }
    static Mat3d Identity() { Mat3d r; r.m[0] = r.m[4] = r.m[8] = 1.0; return r; }
    static Mat3d Zero() { return Mat3d(); }
    double& operator()(int i, int j) { return m[3 * i + j]; }
    double operator()(int i, int j) const { return m[3 * i + j]; }
    Mat3d operator*(const Mat3d& o) const
    {
        Mat3d r;
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j)
            r.m[3 * i + j] = m[3 * i] * o.m[j] + m[3 * i + 1] * o.m[3 + j] + m[3 * i + 2] * o.m[6 + j];
        return r;
    }
    Vec3d operator*(const Vec3d& x) const
    {
        return Vec3d(m[0] * x[0] + m[1] * x[1] + m[2] * x[2], m[3] * x[0] + m[4] * x[1] + m[5] * x[2],
                     m[6] * x[0] + m[7] * x[1] + m[8] * x[2]);
    }
    Mat3d operator*(double s) const { Mat3d r; for (int i = 0; i < 9; ++i) r.m[i] = m[i] * s; return r; }
    Mat3d operator+(const Mat3d& o) const { Mat3d r; for (int i = 0; i < 9; ++i) r.m[i] = m[i] + o.m[i]; return r; }
    Mat3d operator-(const Mat3d& o) const { Mat3d r; for (int i = 0; i < 9; ++i) r.m[i] = m[i] - o.m[i]; return r; }
    Mat3d operator-() const { Mat3d r; for (int i = 0; i < 9; ++i) r.m[i] = -m[i]; return r; }
    Mat3d transpose() const { Mat3d r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[3 * i + j] = m[3 * j + i]; return r; }
    double norm() const { double s = 0; for (int i = 0; i < 9; ++i) s += m[i] * m[i]; return std::sqrt(s); }
};
inline Mat3d operator*(double s, const Mat3d& a) { return a * s; }

}  // namespace ingvio
