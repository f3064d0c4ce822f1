// stub for a syntax check only (ros1/mock/README.md): feature_tracker/MonoFrame (fields of feature_tracker/msg/Mono{Frame,Meas}.msg)
#pragma once
#include <memory>
#include <vector>
#include "std_msgs/Header.h"
namespace feature_tracker {
struct MonoMeas { uint64_t id = 0; double u0 = 0, v0 = 0; };
struct MonoFrame { std_msgs::Header header; std::vector<MonoMeas> mono_features; };
typedef std::shared_ptr<const MonoFrame> MonoFrameConstPtr;
}  // namespace feature_tracker
