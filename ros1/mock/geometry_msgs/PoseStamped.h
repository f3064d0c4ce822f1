// stub for a syntax check only (ros1/mock/README.md): geometry_msgs
#pragma once
#include "std_msgs/Header.h"
namespace geometry_msgs {
struct Point { double x = 0, y = 0, z = 0; };
struct Vector3 { double x = 0, y = 0, z = 0; };
struct Quaternion { double x = 0, y = 0, z = 0, w = 1; };
struct Pose { Point position; Quaternion orientation; };
struct PoseWithCovariance { Pose pose; double covariance[36]; };
struct Twist { Vector3 linear, angular; };
struct TwistWithCovariance { Twist twist; double covariance[36]; };
struct PoseStamped { std_msgs::Header header; Pose pose; };
}  // namespace geometry_msgs
