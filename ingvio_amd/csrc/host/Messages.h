// Messages.h — plain structs with the field names and types of the ROS messages ingvio_estimator exchanges (SURVEY.md 8f row
// f-4): std_msgs/Header, sensor_msgs/Imu (the fields callbackIMU reads, IngvioFilter.cpp:381-395), feature_tracker/msg/
// {MonoMeas,MonoFrame,StereoMeas,StereoFrame}.msg and the nav_msgs/Odometry fields IngvioFilter::visualize fills (:409-447).
// A ROS1 wrapper node copies field by field; without ROS the same structs are fed from a replay file (Replay.h).
#pragma once
#include <cmath>
#include <cstdint>
#include <string>
#include <vector>

namespace ingvio {
namespace msg {

struct Time {                                   // ros::Time
    uint32_t sec = 0, nsec = 0;
    double toSec() const { return (double)sec + 1e-9 * (double)nsec; }
    static Time fromNSec(uint64_t ns) { Time t; t.sec = (uint32_t)(ns / 1000000000ULL); t.nsec = (uint32_t)(ns % 1000000000ULL); return t; }
    uint64_t toNSec() const { return (uint64_t)sec * 1000000000ULL + nsec; }
};
struct Header { uint32_t seq = 0; Time stamp; std::string frame_id; };      // std_msgs/Header

struct Vector3 { double x = 0, y = 0, z = 0; };
struct Quaternion { double x = 0, y = 0, z = 0, w = 1; };
struct Imu { Header header; Quaternion orientation; Vector3 angular_velocity; Vector3 linear_acceleration; };      // sensor_msgs/Imu (covariances omitted: unused)

struct MonoMeas { uint64_t id = 0; double u0 = 0, v0 = 0; };                                    // feature_tracker/MonoMeas.msg
struct MonoFrame { Header header; std::vector<MonoMeas> mono_features; };                       // feature_tracker/MonoFrame.msg
struct StereoMeas { uint64_t id = 0; double u0 = 0, v0 = 0, u1 = 0, v1 = 0; };                  // feature_tracker/StereoMeas.msg
struct StereoFrame { Header header; std::vector<StereoMeas> stereo_features; };                 // feature_tracker/StereoFrame.msg

struct Odometry {                               // nav_msgs/Odometry as visualize() fills it: header, child_frame_id, pose.pose, twist.twist.linear
    Header header; std::string child_frame_id;
    Vector3 position; Quaternion orientation; Vector3 linear_velocity;
};

}  // namespace msg
}  // namespace ingvio
