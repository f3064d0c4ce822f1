// stub for a syntax check only (ros1/mock/README.md): the tf calls of IngvioRosNode::visualize
#pragma once
#include <string>
#include "std_msgs/Header.h"
namespace tf {
struct Vector3 { Vector3(double, double, double) {} };
struct Quaternion { Quaternion(double, double, double, double) {} };
struct Transform { void setOrigin(const Vector3&); void setRotation(const Quaternion&); };
struct StampedTransform { StampedTransform(const Transform&, const ros::Time&, const std::string&, const std::string&) {} };
struct TransformBroadcaster { void sendTransform(const StampedTransform&); };
}  // namespace tf
