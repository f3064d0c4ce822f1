// stub for a syntax check only (ros1/mock/README.md): nav_msgs/Path
#pragma once
#include <vector>
#include "geometry_msgs/PoseStamped.h"
namespace nav_msgs { struct Path { std_msgs::Header header; std::vector<geometry_msgs::PoseStamped> poses; }; }
