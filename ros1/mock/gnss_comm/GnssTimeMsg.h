// stub for a syntax check only (ros1/mock/README.md): gnss_comm/GnssTimeMsg
#pragma once
#include <cstdint>
namespace gnss_comm { struct GnssTimeMsg { uint32_t week = 0; double tow = 0; }; }
