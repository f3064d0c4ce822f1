// stub for a syntax check only (ros1/mock/README.md): gnss_comm/GnssEphemMsg (fields of gnss_comm/msg/GnssEphemMsg.msg)
#pragma once
#include <memory>
#include "gnss_comm/GnssTimeMsg.h"
namespace gnss_comm {
struct GnssEphemMsg {
    uint32_t sat = 0; GnssTimeMsg ttr, toe, toc; double toe_tow = 0; uint32_t week = 0, iode = 0, iodc = 0, health = 0, code = 0;
    double ura = 0, A = 0, e = 0, i0 = 0, omg = 0, OMG0 = 0, M0 = 0, delta_n = 0, OMG_dot = 0, i_dot = 0, cuc = 0, cus = 0, crc = 0, crs = 0, cic = 0, cis = 0,
           af0 = 0, af1 = 0, af2 = 0, tgd0 = 0, tgd1 = 0, A_dot = 0, n_dot = 0;
};
typedef std::shared_ptr<const GnssEphemMsg> GnssEphemMsgConstPtr;
}  // namespace gnss_comm
