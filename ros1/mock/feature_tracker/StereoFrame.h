// stub for a syntax check only (ros1/mock/README.md): feature_tracker/StereoFrame (fields of feature_tracker/msg/Stereo{Frame,Meas}.msg)
#pragma once
#include <memory>
#include <vector>
#include "std_msgs/Header.h"
namespace feature_tracker {
struct StereoMeas { uint64_t id = 0; double u0 = 0, v0 = 0, u1 = 0, v1 = 0; };
struct StereoFrame { std_msgs::Header header; std::vector<StereoMeas> stereo_features; };
typedef std::shared_ptr<const StereoFrame> StereoFrameConstPtr;
}  // namespace feature_tracker
