// test_ros_adapter.cpp — the conversion core of the ROS1 node (ros1/include/RosAdapter.h) instantiated with MOCK message structs
// that have the fields of the .msg files (sensor_msgs/Imu, feature_tracker/StereoFrame, gnss_comm/GnssEphemMsg / GnssGloEphemMsg /
// GnssMeasMsg, nav_msgs/Odometry, geometry_msgs/PoseStamped): the templates compile against the generated ROS classes and these
// alike.  No ROS, no GPU.  
#include <cstdio>
#include <string>
#include <vector>

#include "RosAdapter.h"

namespace mock {
struct Time { uint32_t sec = 0, nsec = 0; };
struct Header { uint32_t seq = 0; Time stamp; std::string frame_id; };
struct V3 { double x = 0, y = 0, z = 0; };
struct Q { double x = 0, y = 0, z = 0, w = 1; };
struct Imu { Header header; Q orientation; V3 angular_velocity, linear_acceleration; };
struct StereoMeas { uint64_t id; double u0, v0, u1, v1; };
struct StereoFrame { Header header; std::vector<StereoMeas> stereo_features; };
struct MonoMeas { uint64_t id; double u0, v0; };
struct MonoFrame { Header header; std::vector<MonoMeas> mono_features; };
struct Pose { V3 position; Q orientation; };
struct PoseCov { Pose pose; };
struct Twist { V3 linear, angular; };
struct TwistCov { Twist twist; };
struct Odometry { Header header; std::string child_frame_id; PoseCov pose; TwistCov twist; };
struct PoseStamped { Header header; Pose pose; };
struct GTime { uint32_t week = 0; double tow = 0; };
struct Ephem {
    uint32_t sat; GTime ttr, toe, toc; double toe_tow; uint32_t week, iode, iodc, health, code;
    double ura, A, e, i0, omg, OMG0, M0, delta_n, OMG_dot, i_dot, cuc, cus, crc, crs, cic, cis, af0, af1, af2, tgd0, tgd1, A_dot, n_dot;
};
struct GloEphem { uint32_t sat; GTime ttr, toe; int32_t freqo; uint32_t iode, health, age; double ura, pos_x, pos_y, pos_z, vel_x, vel_y, vel_z, acc_x, acc_y, acc_z, tau_n, gamma, delta_tau_n; };
struct Obs { GTime time; uint32_t sat; std::vector<double> freqs, CN0; std::vector<uint8_t> LLI, code; std::vector<double> psr, psr_std, cp, cp_std, dopp, dopp_std; std::vector<uint8_t> status; };
struct Meas { std::vector<Obs> meas; };
struct Iono { Header header; std::vector<double> data; };
}  // namespace mock

static int fails = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #c); ++fails; } } while (0)

int main()
{
    using namespace ingvio;
    // ---- IMU / frames / odometry ------------------------------------------------------------------------------------------
    mock::Imu im; im.header.stamp.sec = 12; im.header.stamp.nsec = 500000000; im.angular_velocity.y = 0.25; im.linear_acceleration.z = 9.8;
    const msg::Imu a = ros1::imuFromRos(im);
    CHECK(a.header.stamp.toSec() == 12.5 && a.angular_velocity.y == 0.25 && a.linear_acceleration.z == 9.8);
    mock::StereoFrame sf; sf.header.stamp.sec = 3; sf.stereo_features = { { 7, 0.1, 0.2, 0.3, 0.4 }, { 9, -0.1, -0.2, -0.3, -0.4 } };
    const msg::StereoFrame s = ros1::stereoFrameFromRos(sf);
    CHECK(s.stereo_features.size() == 2 && s.stereo_features[1].id == 9 && s.stereo_features[1].v1 == -0.4 && s.header.stamp.sec == 3);
    mock::MonoFrame mf; mf.mono_features = { { 5, 0.5, 0.6 } };
    CHECK(ros1::monoFrameFromRos(mf).mono_features[0].v0 == 0.6);
    msg::Odometry od; od.header.frame_id = "world"; od.child_frame_id = "uav"; od.position.x = 1; od.orientation.w = 0.5; od.linear_velocity.z = -2;
    mock::Odometry ro; ros1::odometryToRos(od, ro);
    CHECK(ro.header.frame_id == "world" && ro.child_frame_id == "uav" && ro.pose.pose.position.x == 1 && ro.pose.pose.orientation.w == 0.5 && ro.twist.twist.linear.z == -2);
    mock::PoseStamped ps; ros1::poseStampedFromOdometry(ro, ps);
    CHECK(ps.pose.position.x == 1 && ps.header.frame_id == "world");
    // ---- satellite numbering / L1 selection (gnss_utility.cpp:74-93, :933-962) ------------------------------------------------
    uint32_t prn = 0;
    CHECK(ros1::satSysIdx(1, &prn) == 0 && prn == 1); CHECK(ros1::satSysIdx(32, &prn) == 0 && prn == 32);
    CHECK(ros1::satSysIdx(33, &prn) == 1 && prn == 1); CHECK(ros1::satSysIdx(59, &prn) == 1 && prn == 27);
    CHECK(ros1::satSysIdx(60, &prn) == 2 && prn == 1); CHECK(ros1::satSysIdx(97, &prn) == 2 && prn == 38);
    CHECK(ros1::satSysIdx(98, &prn) == 3 && prn == 1); CHECK(ros1::satSysIdx(160, &prn) == 3 && prn == 63);
    CHECK(ros1::satSysIdx(0) == -1 && ros1::satSysIdx(161) == -1);
    mock::Obs o; o.freqs = { 1.2276e9, 1.57542e9 };
    double fq = 0; CHECK(ros1::l1Index(o, 0, &fq) == 1 && fq == 1.57542e9); CHECK(ros1::l1Index(o, 3) == -1);
    o.freqs = { 1.602e9 - 3 * 0.5625e6 }; CHECK(ros1::l1Index(o, 1) == 0);
    // ---- ephemerides -> flat records, best-ephemeris choice, tracking counter, raw epoch ---------------------------------------
    ros1::GnssFrontEnd fe; fe.track_num_thres = 2;
    mock::Ephem e{}; e.sat = 5; e.toe.week = 2200; e.toe.tow = 360000; e.toc = e.toe; e.toe_tow = 360000; e.A = 2.656e7; e.e = 0.01; e.ura = 2.0; e.tgd0 = 1e-8; e.af0 = 1e-4;
    fe.addEphem(e.sat, ros1::ephemFromRos(e));
    mock::Ephem e2 = e; e2.toe.tow = 367200; e2.af0 = 2e-4; fe.addEphem(e2.sat, ros1::ephemFromRos(e2));
    mock::GloEphem g{}; g.sat = 33 + 3; g.toe.week = 2200; g.toe.tow = 360900; g.pos_x = 1e7; g.vel_y = 3e3; g.acc_z = 1e-6; g.tau_n = 1e-5; g.gamma = 1e-12; g.ura = 2.0;
    fe.addEphem(g.sat, ros1::gloEphemFromRos(g));
    mock::Iono io; io.data = { 1, 2, 3, 4, 5, 6, 7, 8 }; fe.setIono(io);
    CHECK(fe.iono.size() == 8 && fe.iono[7] == 8);
    mock::Meas mm;
    mock::Obs o1; o1.time.week = 2200; o1.time.tow = 366000; o1.sat = 5; o1.freqs = { 1.57542e9 }; o1.psr = { 2.2e7 }; o1.psr_std = { 0.5 }; o1.dopp = { -800 }; o1.dopp_std = { 0.3 };
    mock::Obs o2 = o1; o2.sat = 36; o2.freqs = { 1.602e9 }; o2.psr = { 2.0e7 };
    mock::Obs o3 = o1; o3.sat = 7;                         // no ephemeris: dropped
    mm.meas = { o1, o2, o3 };
    RawGnssEpoch raw;
    CHECK(fe.epochFromRos(mm, 270.0, raw) == 0);           // first sighting: below the tracking threshold
    CHECK(fe.epochFromRos(mm, 270.0, raw) == 2);
    CHECK(raw.eph[0] == 0 && raw.eph[1] == 5 && raw.eph[20] == 2e-4);                  // the ephemeris closest in time (toe 367200)
    CHECK(raw.eph[INGVIO_EPH_N] == 1 && raw.eph[INGVIO_EPH_N + 1] == 4 && raw.eph[INGVIO_EPH_N + 5] == 1e7 && raw.eph[INGVIO_EPH_N + 9] == 3e3 && raw.eph[INGVIO_EPH_N + 14] == 1e-5);
    CHECK(raw.obs[0] == 366000 && raw.obs[1] == 2.2e7 && raw.obs[5] == 1.57542e9 && raw.obs[INGVIO_OBS_N + 5] == 1.602e9);
    mock::Obs bad = o1; bad.psr_std = { 9.0 }; mock::Meas mb; mb.meas = { bad };
    CHECK(fe.epochFromRos(mb, 270.0, raw) == 0);           // std above the threshold resets the counter ...
    mock::Meas m1; m1.meas = { o1 };
    CHECK(fe.epochFromRos(m1, 270.0, raw) == 0);           // ... so the next good epoch is a first sighting again
    // ---- evaluated records -> GnssMeas ------------------------------------------------------------------------------------------
    CHECK(fe.epochFromRos(mm, 270.0, raw) == 2);
    std::vector<double> rec(2 * INGVIO_GNSS_SAT_REC, 0.0);
    rec[9] = 1; rec[7] = 3.5; rec[8] = 2.5; rec[10] = 1.5e7; rec[14] = -2e3; rec[16] = 1e-4; rec[18] = 1e-8;
    rec[INGVIO_GNSS_SAT_REC + 9] = 0;                      // second satellite unusable
    const GnssMeas gm = ros1::gnssMeasFromEval(100.5, raw, fe.iono, rec.data());
    CHECK(gm.stamp == 100.5 && gm.sats.size() == 1 && gm.sats[0].sys == 0 && gm.sats[0].sv_pos[0] == 1.5e7 && gm.sats[0].sv_vel[1] == -2e3);
    CHECK(gm.sats[0].ion_delay == 3.5 && gm.sats[0].tro_delay == 2.5 && gm.sats[0].tgd == 1e-8 && gm.raw_obs.size() == 2 * INGVIO_OBS_N && gm.iono.size() == 8);
    // ---- GnssSync::storeTimePair (GnssSync.cpp:66-134) as the node feeds it: arrivals on the local clock, header stamps on ANOTHER
    // clock (a bag played without --clock): the offset must map GNSS time onto the header clock, minus the arrival difference ----
    {
        GnssSync sync(0.05);
        const double hdr_off = -1000.0;                   // header clock = local clock - 1000 s
        const double gps_off = 3.0e8;                     // GNSS time = local clock + 3e8 s
        // three IMU headers arrive first: not enough GNSS epochs yet
        for (int i = 0; i < 3; ++i) sync.storeTimePairHeader(50.00 + 0.005 * i, 50.00 + 0.005 * i + hdr_off);
        CHECK(!sync.isSync());
        sync.storeTimePairGnss(50.10, 50.10 + gps_off - 0.02);      // the epoch was measured 20 ms before it arrived
        sync.storeTimePairGnss(50.20, 50.20 + gps_off - 0.02);
        CHECK(!sync.isSync());
        GnssMeas early; early.stamp = 1.0;
        sync.bufferGnssMeas(early);                       // ignored before the sync (GnssSync.cpp:27-31)
        GnssMeas out;
        CHECK(!sync.getGnssMeasAt(1.0, out));
        sync.storeTimePairHeader(50.298, 50.298 + hdr_off);
        sync.storeTimePairGnss(50.30, 50.30 + gps_off - 0.02);      // third epoch: closest pair = (50.30, 50.298), 2 ms apart
        CHECK(sync.isSync());
        // offset = header - gnss - (arrival_h - arrival_g) = (50.298 - 1000) - (50.30 + 3e8 - 0.02) - (50.298 - 50.30)
        const double expect = (50.298 + hdr_off) - (50.30 + gps_off - 0.02) - (50.298 - 50.30);
        CHECK(std::fabs(sync.getUnsyncTime() - expect) < 1e-6);
        // an epoch measured at GNSS time g is found at header time g + offset
        GnssMeas m; m.stamp = (60.0 + gps_off) + sync.getUnsyncTime();
        sync.bufferGnssMeas(m);
        CHECK(sync.getGnssMeasAt(60.0 + hdr_off + 0.02, out) && out.stamp == m.stamp);
        // arrivals further apart than the threshold: no sync, tables cleared, collection starts over
        GnssSync far(0.05);
        for (int i = 0; i < 3; ++i) far.storeTimePairHeader(10.0 + 0.005 * i, 10.0);
        for (int i = 0; i < 3; ++i) far.storeTimePairGnss(10.5 + 0.1 * i, 7.0);
        CHECK(!far.isSync());
        far.storeTimePairGnss(10.9, 7.4);                 // only one epoch in the fresh table: still waiting
        CHECK(!far.isSync());
        // use_fix_time_offset (GnssSync.h:60-74): synchronised from the start with the configured offset
        IngvioParams fp; fp._use_fix_time_offset = 1; fp._gnss_local_offset = -18.0;
        GnssSync fixed(fp);
        CHECK(fixed.isSync() && fixed.getUnsyncTime() == -18.0);
        IngvioParams fq2;
        CHECK(!GnssSync(fq2).isSync());
        IngvioParams fq3; fq3._enable_gnss = 0; fq3._use_fix_time_offset = 1;
        CHECK(!GnssSync(fq3).isSync());                   // GnssSync.h:64-65: returns before the fixed offset is looked at
        // the matching window of getGnssMeasAt / getSppAt is the reference's 0.13 s (GnssSync.h:61, GnssSync.cpp:148-156), ADVICE r04:
        // an epoch 100 ms older than the frame (ordinary receiver latency) is matched, one 131 ms older is dropped, one 130 ms
        // newer is left for a later frame
        GnssSync win(fp);
        GnssMeas a, b, c; a.stamp = 9.869; b.stamp = 9.90; c.stamp = 10.13;
        win.bufferGnssMeas(a); win.bufferGnssMeas(b); win.bufferGnssMeas(c);
        CHECK(win.getGnssMeasAt(10.0, out) && out.stamp == 9.90);      // 9.869 < 10 - 0.13: dropped; 9.90 taken
        CHECK(!win.getGnssMeasAt(10.0, out));                            // 10.13 >= 10 + 0.13: stays in the buffer ...
        CHECK(win.getGnssMeasAt(10.05, out) && out.stamp == 10.13);     // ... for the next frame
        SppMeas s1; s1.stamp = 9.95;
        win.bufferSppMeas(s1);
        SppMeas so;
        CHECK(win.getSppAt(10.0, so) && so.stamp == 9.95);
    }
    std::printf("ros adapter: %d failures\n", fails);
    return fails ? 1 : 0;
}
