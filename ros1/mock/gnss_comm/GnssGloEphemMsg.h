// stub for a syntax check only (ros1/mock/README.md): gnss_comm/GnssGloEphemMsg (fields of gnss_comm/msg/GnssGloEphemMsg.msg)
#pragma once
#include <memory>
#include "gnss_comm/GnssTimeMsg.h"
namespace gnss_comm {
struct GnssGloEphemMsg {
    uint32_t sat = 0; GnssTimeMsg ttr, toe; int32_t freqo = 0; uint32_t iode = 0, health = 0, age = 0;
    double ura = 0, pos_x = 0, pos_y = 0, pos_z = 0, vel_x = 0, vel_y = 0, vel_z = 0, acc_x = 0, acc_y = 0, acc_z = 0, tau_n = 0, gamma = 0, delta_tau_n = 0;
};
typedef std::shared_ptr<const GnssGloEphemMsg> GnssGloEphemMsgConstPtr;
}  // namespace gnss_comm
