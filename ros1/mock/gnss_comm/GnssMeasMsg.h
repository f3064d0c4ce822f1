// stub for a syntax check only (ros1/mock/README.md): gnss_comm/GnssMeasMsg = GnssObsMsg[] meas (gnss_comm/msg/Gnss{Meas,Obs}Msg.msg)
#pragma once
#include <memory>
#include <vector>
#include "gnss_comm/GnssTimeMsg.h"
namespace gnss_comm {
struct GnssObsMsg {
    GnssTimeMsg time; uint32_t sat = 0;
    std::vector<double> freqs, CN0; std::vector<uint8_t> LLI, code; std::vector<double> psr, psr_std, cp, cp_std, dopp, dopp_std; std::vector<uint8_t> status;
};
struct GnssMeasMsg { std::vector<GnssObsMsg> meas; };
typedef std::shared_ptr<const GnssMeasMsg> GnssMeasMsgConstPtr;
}  // namespace gnss_comm
