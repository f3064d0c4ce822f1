// GnssComm.h — the slice of the vendored gnss_comm library that sits between a GNSS epoch and GnssUpdate::updateTrackedSys
// (first half): pseudo-range and Doppler residuals with their line-of-sight Jacobians from satellite
// states and L1 observations (gnss_comm/src/gnss_spp.cpp:100-146 psr_res, :256-282 dopp_res), elevation / azimuth
// (gnss_utility.cpp:762-772 sat_azel with ecef2geo :347-388 and ecef2enu :733-743).  The satellite states themselves
// (ephemeris -> position / velocity / clock, gnss_utility.cpp:390-731) and the atmosphere models (:774-899) arrive in the
// message: they are the second half of f-3.
#pragma once
#include <vector>

#include "Mat3.h"

namespace ingvio {
namespace gnss {

constexpr double LIGHT_SPEED = 2.99792458e8;          // gnss_constant.hpp:214
constexpr double EARTH_OMG_GPS = 7.2921151467e-5;     // :208
constexpr double EARTH_SEMI_MAJOR = 6378137.0;        // :205
constexpr double EARTH_ECCE_2 = 6.69437999014e-3;     // :203
constexpr double FREQ1 = 1.57542e9, FREQ1_BDS = 1.561098e9;      // :49, :61

// One L1 observation with the state of its satellite at transmit time: gnss_comm::Obs (psr, dopp, stds, frequency at l1_idx),
// gnss_comm::SatState (gnss_constant.hpp:506-516: pos, vel, dt, ddt, tgd) and the ephemeris' ura.
struct SatObs {
    int sys = 0;                        // gnss_comm::sys2idx: GPS 0, GLO 1, GAL 2, BDS 3   (gnss_constant.hpp:264-270)
    double psr = 0, dopp = 0, psr_std = 1, dopp_std = 1, freq = FREQ1;
    Vec3d sv_pos, sv_vel;               // ECEF, m and m/s
    double sv_dt = 0, sv_ddt = 0, tgd = 0, ura = 2;
    double ion_delay = 0, tro_delay = 0;      // calculate_ion_delay / calculate_trop_delay outputs, m
};

Vec3d ecef2geo(const Vec3d& xyz);                                      // (lat deg, lon deg, alt m), gnss_utility.cpp:347-388
Vec3d geo2ecef(const Vec3d& lla);                                      // :335-345
Mat3d geo2rotation(const Vec3d& ref_geo);                              // R_ecef_enu, :745-755
Vec3d ecef2enu(const Vec3d& ref_lla, const Vec3d& v_ecef);             // :733-743
void sat_azel(const Vec3d& rcv_pos, const Vec3d& sat_pos, double azel[2]);   // :762-772

// psr_res (gnss_spp.cpp:100-146): rcv_state = (ecef xyz, clock bias of GPS / GLO / GAL / BDS in m).  res[i] = estimated - measured,
// J[i] = (-unit_rv2sv, one-hot clock column); the atmosphere delays of the observation are used as given.
void psr_res(const double rcv_state[7], const std::vector<SatObs>& obs, std::vector<double>& res, std::vector<Vec3d>& unit_rv2sv,
             std::vector<double>& az, std::vector<double>& el);
// dopp_res (:256-282): rcv_state = (ecef velocity, clock drift m/s)
void dopp_res(const double rcv_state[4], const Vec3d& rcv_ecef, const std::vector<SatObs>& obs, std::vector<double>& res);

}  // namespace gnss
}  // namespace ingvio
