#include "PoseState.h"

namespace ingvio {

Quatd quatFromRot(const Mat3d& R)      // Eigen::Quaterniond(Matrix3d) (Shepperd), used by PoseState.cpp:27,84,181
{
    Quatd q;
    const double t = R(0, 0) + R(1, 1) + R(2, 2);
    if (t > 0.0) {
        double s = std::sqrt(t + 1.0);
        q.w = 0.5 * s; s = 0.5 / s;
        q.x = (R(2, 1) - R(1, 2)) * s; q.y = (R(0, 2) - R(2, 0)) * s; q.z = (R(1, 0) - R(0, 1)) * s;
    } else {
        int i = 0;
        if (R(1, 1) > R(0, 0)) i = 1;
        if (R(2, 2) > R(i, i)) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        double s = std::sqrt(R(i, i) - R(j, j) - R(k, k) + 1.0);
        double v[3];
        v[i] = 0.5 * s; s = 0.5 / s;
        q.w = (R(k, j) - R(j, k)) * s;
        v[j] = (R(j, i) + R(i, j)) * s; v[k] = (R(k, i) + R(i, k)) * s;
        q.x = v[0]; q.y = v[1]; q.z = v[2];
    }
    return q;
}

Mat3d rotFromQuat(const Quatd& qi)
{
    const double n = std::sqrt(qi.w * qi.w + qi.x * qi.x + qi.y * qi.y + qi.z * qi.z);
    const double w = qi.w / n, x = qi.x / n, y = qi.y / n, z = qi.z / n;
    Mat3d R;
    R(0, 0) = 1 - 2 * (y * y + z * z); R(0, 1) = 2 * (x * y - z * w); R(0, 2) = 2 * (x * z + y * w);
    R(1, 0) = 2 * (x * y + z * w); R(1, 1) = 1 - 2 * (x * x + z * z); R(1, 2) = 2 * (y * z - x * w);
    R(2, 0) = 2 * (x * z - y * w); R(2, 1) = 2 * (y * z + x * w); R(2, 2) = 1 - 2 * (x * x + y * y);
    return R;
}

}  // namespace ingvio
