// stub for a syntax check only (ros1/mock/README.md): std_msgs/Header
#pragma once
#include <cstdint>
#include <string>
namespace ros {
struct Time {
    uint32_t sec = 0, nsec = 0;
    Time() {}
    explicit Time(double t) : sec((uint32_t)t), nsec((uint32_t)((t - (uint32_t)t) * 1e9)) {}
    double toSec() const { return (double)sec + 1e-9 * (double)nsec; }
    static Time now();
};
}  // namespace ros
namespace std_msgs { struct Header { uint32_t seq = 0; ros::Time stamp; std::string frame_id; }; }
