// stub for a syntax check only (ros1/mock/README.md): nav_msgs/Odometry
#pragma once
#include <string>
#include "geometry_msgs/PoseStamped.h"
namespace nav_msgs { struct Odometry { std_msgs::Header header; std::string child_frame_id; geometry_msgs::PoseWithCovariance pose; geometry_msgs::TwistWithCovariance twist; }; }
