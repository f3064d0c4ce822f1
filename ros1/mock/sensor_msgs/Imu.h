// stub for a syntax check only (ros1/mock/README.md): sensor_msgs/Imu
#pragma once
#include <memory>
#include "geometry_msgs/PoseStamped.h"
namespace sensor_msgs {
struct Imu {
    std_msgs::Header header;
    geometry_msgs::Quaternion orientation; double orientation_covariance[9];
    geometry_msgs::Vector3 angular_velocity; double angular_velocity_covariance[9];
    geometry_msgs::Vector3 linear_acceleration; double linear_acceleration_covariance[9];
};
typedef std::shared_ptr<const Imu> ImuConstPtr;
}  // namespace sensor_msgs
