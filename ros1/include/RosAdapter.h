// RosAdapter.h — the field-by-field bridge between the ROS1 messages ingvio_estimator exchanges and the shim's plain structs
// (reference: ingvio_estimator/src/IngvioFilter.cpp:50-122 initIO, :409-498 visualize / visualizeSpp,
// GnssProcessor.cpp:32-220, gnss_comm/src/gnss_ros.cpp:72-221 msg2ephem / msg2glo_ephem / msg2meas).
//
// Every conversion is a TEMPLATE on the message type: it only names the fields the .msg files define (sensor_msgs/Imu,
// feature_tracker/{Mono,Stereo}Frame, gnss_comm/Gnss{Ephem,GloEphem,Meas}Msg, nav_msgs/Odometry, geometry_msgs/PoseStamped), so
// the same code serves the generated ROS classes in ros1/src/ingvio_node.cpp and the mock structs of the unit test
// (tests/cpp/test_ros_adapter.cpp) that is built where ROS is not installed.
#pragma once
#include <cmath>
#include <cstdint>
#include <map>
#include <vector>

#include "GnssUpdate.h"
#include "GvioAligner.h"
#include "Messages.h"
#include "PoseState.h"
#include "ingvio_hip.h"

namespace ingvio {
namespace ros1 {

// ---- camera / IMU side ---------------------------------------------------------------------------------------------------------
template <class HeaderT>
msg::Header headerFromRos(const HeaderT& h)
{
    msg::Header o;
    o.seq = h.seq; o.stamp.sec = h.stamp.sec; o.stamp.nsec = h.stamp.nsec; o.frame_id = h.frame_id;
    return o;
}

template <class ImuT>                              // sensor_msgs/Imu -> callbackIMU (IngvioFilter.cpp:381-395)
msg::Imu imuFromRos(const ImuT& m)
{
    msg::Imu o;
    o.header = headerFromRos(m.header);
    o.angular_velocity.x = m.angular_velocity.x; o.angular_velocity.y = m.angular_velocity.y; o.angular_velocity.z = m.angular_velocity.z;
    o.linear_acceleration.x = m.linear_acceleration.x; o.linear_acceleration.y = m.linear_acceleration.y; o.linear_acceleration.z = m.linear_acceleration.z;
    o.orientation.x = m.orientation.x; o.orientation.y = m.orientation.y; o.orientation.z = m.orientation.z; o.orientation.w = m.orientation.w;
    return o;
}

template <class FrameT>                            // feature_tracker/MonoFrame
msg::MonoFrame monoFrameFromRos(const FrameT& m)
{
    msg::MonoFrame o;
    o.header = headerFromRos(m.header);
    o.mono_features.reserve(m.mono_features.size());
    for (const auto& f : m.mono_features) { msg::MonoMeas q; q.id = f.id; q.u0 = f.u0; q.v0 = f.v0; o.mono_features.push_back(q); }
    return o;
}

template <class FrameT>                            // feature_tracker/StereoFrame
msg::StereoFrame stereoFrameFromRos(const FrameT& m)
{
    msg::StereoFrame o;
    o.header = headerFromRos(m.header);
    o.stereo_features.reserve(m.stereo_features.size());
    for (const auto& f : m.stereo_features) { msg::StereoMeas q; q.id = f.id; q.u0 = f.u0; q.v0 = f.v0; q.u1 = f.u1; q.v1 = f.v1; o.stereo_features.push_back(q); }
    return o;
}

// IngvioFilter::visualize (:409-447): nav_msgs/Odometry "world" -> "uav" and the geometry_msgs/PoseStamped appended to the path
template <class OdomT>
void odometryToRos(const msg::Odometry& s, OdomT& o)
{
    o.header.stamp.sec = s.header.stamp.sec; o.header.stamp.nsec = s.header.stamp.nsec;
    o.header.frame_id = s.header.frame_id; o.child_frame_id = s.child_frame_id;
    o.pose.pose.position.x = s.position.x; o.pose.pose.position.y = s.position.y; o.pose.pose.position.z = s.position.z;
    o.pose.pose.orientation.x = s.orientation.x; o.pose.pose.orientation.y = s.orientation.y; o.pose.pose.orientation.z = s.orientation.z;
    o.pose.pose.orientation.w = s.orientation.w;
    o.twist.twist.linear.x = s.linear_velocity.x; o.twist.twist.linear.y = s.linear_velocity.y; o.twist.twist.linear.z = s.linear_velocity.z;
}
template <class OdomT, class PoseStampedT>
void poseStampedFromOdometry(const OdomT& o, PoseStampedT& p)
{
    p.header.stamp = o.header.stamp; p.header.frame_id = o.header.frame_id;
    p.pose.position = o.pose.pose.position; p.pose.orientation = o.pose.pose.orientation;
}

// ---- GNSS side -----------------------------------------------------------------------------------------------------------------
// gnss_comm's satellite numbering (gnss_constant.hpp:88-112, gnss_utility.cpp:74-93 satsys): 32 GPS, 27 GLONASS, 38 Galileo,
// 63 BeiDou slots in that order; sys2idx (gnss_constant.hpp:264-270): GPS 0, GLO 1, GAL 2, BDS 3.
inline int satSysIdx(uint32_t sat, uint32_t* prn = nullptr)
{
    constexpr uint32_t NG = 32, NR = 27, NE = 38, NC = 63;
    int sys = -1;
    uint32_t p = 0;
    if (sat >= 1 && sat <= NG) { sys = 0; p = sat; }
    else if (sat > NG && sat <= NG + NR) { sys = 1; p = sat - NG; }
    else if (sat > NG + NR && sat <= NG + NR + NE) { sys = 2; p = sat - NG - NR; }
    else if (sat > NG + NR + NE && sat <= NG + NR + NE + NC) { sys = 3; p = sat - NG - NR - NE; }
    if (prn) *prn = p;
    return sys;
}

// gnss_utility.cpp:933-962 L1_freq: the index of the L1 / E1 / B1 / G1 observation of an epoch, -1 if there is none
template <class ObsT>
int l1Index(const ObsT& obs, int sys, double* freq = nullptr)
{
    double lo = -1.0, hi = -1.0;
    if (sys == 0 || sys == 2) lo = hi = 1.57542e9;
    else if (sys == 3) lo = hi = 1.561098e9;
    else if (sys == 1) { lo = 1.602e9 - 7 * 0.5625e6; hi = 1.602e9 + 6 * 0.5625e6; }
    for (size_t i = 0; i < obs.freqs.size(); ++i)
        if (obs.freqs[i] >= lo && obs.freqs[i] <= hi) { if (freq) *freq = obs.freqs[i]; return (int)i; }
    return -1;
}

constexpr double WEEK_SEC = 604800.0;
inline double gpstAbs(uint32_t week, double tow) { return (double)week * WEEK_SEC + tow; }      // seconds since the GPS epoch (gpst2time, up to its origin)

struct EphemRecord { double t_abs = 0.0; double rec[INGVIO_EPH_N]; };                          // t_abs: toe, for the validity search

// gnss_ros.cpp:72-109 msg2ephem -> the flat Kepler record of include/ingvio_hip.h (times as seconds of the GPS week of the epoch)
template <class EphemMsgT>
EphemRecord ephemFromRos(const EphemMsgT& m)
{
    EphemRecord e;
    for (double& x : e.rec) x = 0.0;
    uint32_t prn = 0;
    const int sys = satSysIdx(m.sat, &prn);
    e.t_abs = gpstAbs(m.toe.week, m.toe.tow);
    double* r = e.rec;
    r[0] = sys; r[1] = prn; r[2] = m.toe.tow; r[3] = m.toe_tow; r[4] = m.toc.tow;
    r[5] = m.A; r[6] = m.e; r[7] = m.i0; r[8] = m.omg; r[9] = m.OMG0; r[10] = m.M0; r[11] = m.delta_n; r[12] = m.OMG_dot; r[13] = m.i_dot;
    r[14] = m.cuc; r[15] = m.cus; r[16] = m.crc; r[17] = m.crs; r[18] = m.cic; r[19] = m.cis; r[20] = m.af0; r[21] = m.af1; r[22] = m.af2;
    r[23] = m.tgd0; r[24] = m.ura;
    return e;
}

// gnss_ros.cpp:143-167 msg2glo_ephem -> the flat GLONASS record (pos / vel / acc of the PZ-90 state vector, tau_n, gamma)
template <class GloMsgT>
EphemRecord gloEphemFromRos(const GloMsgT& m)
{
    EphemRecord e;
    for (double& x : e.rec) x = 0.0;
    uint32_t prn = 0;
    satSysIdx(m.sat, &prn);
    e.t_abs = gpstAbs(m.toe.week, m.toe.tow);
    double* r = e.rec;
    r[0] = 1; r[1] = prn; r[2] = m.toe.tow;
    r[5] = m.pos_x; r[6] = m.pos_y; r[7] = m.pos_z; r[8] = m.vel_x; r[9] = m.vel_y; r[10] = m.vel_z; r[11] = m.acc_x; r[12] = m.acc_y; r[13] = m.acc_z;
    r[14] = m.tau_n; r[15] = m.gamma; r[16] = m.delta_tau_n; r[24] = m.ura;
    return e;
}

// GnssData (GnssData.h) + the selection part of GnssProcessor::callbackGnssMeas (GnssProcessor.cpp:133-196): ephemerides per
// satellite, the one closest in time (within EPH_VALID_SECONDS = 7200 s) serves an observation; tracking counters and the
// psr / Doppler std thresholds as written (:173-186, both compare psr_std).
class GnssFrontEnd {
public:
    double psr_std_thres = 2.0, dopp_std_thres = 2.0;      // IngvioParams _gnss_psr_std_thres / _gnss_dopp_std_thres
    int track_num_thres = 20;                              // _gnss_track_num_thres
    std::vector<double> iono;                              // latest_gnss_iono_params (callbackIonoParams, GnssProcessor.cpp:102-117)

    void addEphem(uint32_t sat, const EphemRecord& e)      // callbackEphem / callbackGloEphem (:32-100): keep one record per toe
    {
        auto& lst = _eph[sat];
        for (auto& q : lst) if (q.t_abs == e.t_abs) { q = e; return; }
        lst.push_back(e);
    }
    template <class IonoMsgT>
    void setIono(const IonoMsgT& m) { if (m.data.size() == 8) iono.assign(m.data.begin(), m.data.end()); }

    // gnss_comm/GnssMeasMsg -> the raw epoch (flat records of the valid L1 observations with their ephemerides); doy = day of
    // year of the epoch (gnss_comm::time2doy).  Returns the number of satellites kept.
    template <class MeasMsgT>
    int epochFromRos(const MeasMsgT& m, double doy, RawGnssEpoch& out, double* t_abs_out = nullptr)
    {
        out.eph.clear(); out.obs.clear(); out.doy = doy;
        for (const auto& o : m.meas) {
            const int sys = satSysIdx(o.sat);
            if (sys < 0) continue;
            auto it = _eph.find(o.sat);
            if (it == _eph.end() || o.freqs.empty()) continue;
            double freq = -1.0;
            const int l1 = l1Index(o, sys, &freq);
            if (l1 < 0) continue;
            const double t_obs = gpstAbs(o.time.week, o.time.tow);
            const EphemRecord* best = nullptr;
            double best_dt = 7200.0;                                                    // EPH_VALID_SECONDS
            for (const auto& e : it->second) if (std::fabs(e.t_abs - t_obs) < best_dt) { best_dt = std::fabs(e.t_abs - t_obs); best = &e; }
            if (!best) continue;
            if (o.psr_std[l1] > psr_std_thres || o.psr_std[l1] > dopp_std_thres) { _track[o.sat] = 0; continue; }      // as written (:173-174)
            if (++_track[o.sat] < track_num_thres) continue;
            out.eph.insert(out.eph.end(), best->rec, best->rec + INGVIO_EPH_N);
            const double rec[INGVIO_OBS_N] = { o.time.tow, o.psr[l1], o.dopp[l1], o.psr_std[l1], o.dopp_std[l1], freq };
            out.obs.insert(out.obs.end(), rec, rec + INGVIO_OBS_N);
            if (t_abs_out) *t_abs_out = t_obs;
        }
        return out.n_sat();
    }

private:
    std::map<uint32_t, std::vector<EphemRecord>> _eph;
    std::map<uint32_t, int> _track;
};

// The per-satellite part of GnssMeas the update reads (gnss::SatObs) from the device's evaluation of a raw epoch at the
// receiver's current ECEF state: satellite state + atmosphere delays (record layout: INGVIO_GNSS_SAT_REC of ingvio_hip.h)
inline GnssMeas gnssMeasFromEval(double stamp, const RawGnssEpoch& raw, const std::vector<double>& iono, const double* rec)
{
    GnssMeas g;
    g.stamp = stamp; g.raw_eph = raw.eph; g.raw_obs = raw.obs; g.iono = iono; g.doy = raw.doy;
    for (int i = 0; i < raw.n_sat(); ++i) {
        const double* r = rec + (size_t)i * INGVIO_GNSS_SAT_REC;
        if (r[9] == 0.0) continue;
        const double* e = raw.eph.data() + (size_t)i * INGVIO_EPH_N;
        const double* o = raw.obs.data() + (size_t)i * INGVIO_OBS_N;
        gnss::SatObs s;
        s.sys = (int)e[0]; s.psr = o[1]; s.dopp = o[2]; s.psr_std = o[3]; s.dopp_std = o[4]; s.freq = o[5];
        s.sv_pos = Vec3d(r + 10); s.sv_vel = Vec3d(r + 13); s.sv_dt = r[16]; s.sv_ddt = r[17]; s.tgd = r[18]; s.ura = e[24];
        s.ion_delay = r[7]; s.tro_delay = r[8];
        g.sats.push_back(s);
    }
    return g;
}

}  // namespace ros1
}  // namespace ingvio
