// stub for a syntax check only (ros1/mock/README.md): the roscpp calls of ingvio_node.cpp
#pragma once
#include <memory>
#include <string>
#include "std_msgs/Header.h"
#define ROSCONSOLE_DEFAULT_NAME "ros"
#define ROS_ERROR(...) ((void)0)
namespace ros {
struct Subscriber {};
struct Publisher { template <class M> void publish(const M&) const; };
class NodeHandle {
public:
    explicit NodeHandle(const std::string& ns = "");
    template <class T> bool param(const std::string& name, T& value, const T& def) const;
    template <class M, class C> Subscriber subscribe(const std::string& topic, uint32_t queue, void (C::*fp)(const std::shared_ptr<const M>&), C* obj);
    template <class M> Publisher advertise(const std::string& topic, uint32_t queue);
};
void init(int& argc, char** argv, const std::string& name);
void spin();
namespace console { namespace levels { enum Level { Debug, Info, Warn, Error, Fatal }; } bool set_logger_level(const std::string& name, levels::Level level); }
}  // namespace ros
