// stub for a syntax check only (ros1/mock/README.md): gnss_comm/StampedFloat64Array
#pragma once
#include <memory>
#include <vector>
#include "std_msgs/Header.h"
namespace gnss_comm {
struct StampedFloat64Array { std_msgs::Header header; std::vector<double> data; };
typedef std::shared_ptr<const StampedFloat64Array> StampedFloat64ArrayConstPtr;
}  // namespace gnss_comm
