// Replay.h — a rosbag-free recording of the topics ingvio subscribes to (IngvioFilter::initIO,
// IngvioFilter.cpp:50-122) and a driver that plays it into the callback surface.
//
// File = magic "INGVIOR1", then records { u8 type, u64 stamp_ns, u32 payload_bytes, payload } (little endian):
//   0 PARAMS        ASCII "key: value" lines with the key names of config/*/ingvio_{mono,stereo}.yaml (+ T_cl2i / T_cr2i as 12
//                   numbers, row-major 3x4: the camera yaml's T_cam_imu) — stands in for the OpenCV-YAML reader
//   1 IMU           f64 x 6: angular_velocity xyz, linear_acceleration xyz                              (sensor_msgs/Imu)
//   2 MONO_FRAME    u32 n, n x { u64 id, f64 u0, f64 v0 }                                                (feature_tracker/MonoFrame)
//   3 STEREO_FRAME  u32 n, n x { u64 id, f64 u0, v0, u1, v1 }                                            (feature_tracker/StereoFrame)
//   4 GNSS_MEAS     u32 n, n x { i32 sys, f64 x 17: psr, dopp, psr_std, dopp_std, freq, sv_pos 3, sv_vel 3, sv_dt, sv_ddt, tgd,
//                   ura, ion_delay, tro_delay }                                     (GnssMeas: observations with satellite states)
//   5 SPP_MEAS      f64 x 11: posSpp 7, velSpp 4
//   6 ALIGNMENT     f64 x 14: aligned, yaw_offset, R_enu2ecef 9, anchor_ecef 3                          (GvioAligner result)
//   7 GROUND_TRUTH  f64 x 7: position 3, orientation xyzw                                               (evaluation only)
//   8 GNSS_RAW      f64 doy, f64 x 8 Klobuchar parameters, u32 n, n x { f64 x 25 ephemeris, f64 x 6 observation } in the layout of
//                   ingvio_gnss_epoch (include/ingvio_hip.h): the raw content of the GNSS_MEAS record that FOLLOWS with the same stamp
//                   (GnssData.h GnssMeas = (obs, ephems) + latest_gnss_iono_params) - what GvioAligner::batchAlign reads; a recording
//                   without an ALIGNMENT record and with these lets the filter align itself (IngvioFilter.cpp:344-345)
#pragma once
#include <cstdint>
#include <cstdio>
#include <functional>
#include <string>
#include <vector>

#include "IngvioFilter.h"
#include "Messages.h"
#include "SynthStream.h"

namespace ingvio {

enum ReplayType : uint8_t { RP_PARAMS = 0, RP_IMU, RP_MONO_FRAME, RP_STEREO_FRAME, RP_GNSS_MEAS, RP_SPP_MEAS, RP_ALIGNMENT, RP_GROUND_TRUTH, RP_GNSS_RAW, RP_COUNT };

struct ReplayRecord { uint8_t type = 0; uint64_t stamp_ns = 0; std::vector<uint8_t> payload; };

class ReplayReader {
public:
    bool open(const std::string& path);
    bool next(ReplayRecord& rec);               // false at end of file or on a malformed record (see error())
    const std::string& error() const { return _err; }
    ~ReplayReader() { if (_f) std::fclose(_f); }
private:
    FILE* _f = nullptr;
    std::string _err;
};

class ReplayWriter {
public:
    bool open(const std::string& path);
    void params(const std::string& text);
    void imu(const msg::Imu& m);
    void mono(const msg::MonoFrame& m);
    void stereo(const msg::StereoFrame& m);
    void gnss(const GnssMeas& m);
    void spp(const SppMeas& m);
    void alignment(const GvioAlignment& a, double stamp);
    void truth(double stamp, const double p[3], const double q_xyzw[4]);
    void gnssRaw(double stamp, double doy, const double iono[8], const std::vector<double>& eph, const std::vector<double>& obs);
    void close() { if (_f) std::fclose(_f); _f = nullptr; }
    ~ReplayWriter() { close(); }
private:
    void put(uint8_t type, uint64_t stamp_ns, const std::vector<uint8_t>& payload);
    FILE* _f = nullptr;
};

// "key: value" lines -> IngvioParams (the keys of IngvioParams.cpp:27-174 that the covariance path reads)
bool applyParamsText(const std::string& text, IngvioParams& p);

// decoders of the message records
bool decodeImu(const ReplayRecord& r, msg::Imu& m);
bool decodeMono(const ReplayRecord& r, msg::MonoFrame& m);
bool decodeStereo(const ReplayRecord& r, msg::StereoFrame& m);
bool decodeGnss(const ReplayRecord& r, GnssMeas& m);
bool decodeSpp(const ReplayRecord& r, SppMeas& m);
bool decodeAlignment(const ReplayRecord& r, GvioAlignment& a);
bool decodeGnssRaw(const ReplayRecord& r, GnssMeas& m);      // fills m.raw_eph / raw_obs / iono / doy

struct ReplayStats { uint64_t counts[RP_COUNT] = { 0 }; uint64_t features = 0; double t_first = 0, t_last = 0; int frames_processed = 0; };

// Plays a file into a filter built from its PARAMS record (overrides applied last); on_odom is called after every camera frame the
// filter processed, with the odometry visualize() would publish.  dump_only: parse and count, no filter (needs no GPU).
bool replayFile(const std::string& path, const std::string& overrides, bool dump_only,
                const std::function<void(const msg::Odometry&, const IngvioFilter&)>& on_odom, ReplayStats& stats, std::string& err);

// ---- synthetic streams (SynthStream.h) ------------------------------------------------------------------------------------
// Writes the stream of `cfg` as an INGVIOR1 file.
bool writeSynthRecording(const SynthConfig& cfg, const std::string& path);

// One processed camera frame of a timed play: wall time of the callback (it ends with the frame's results on the host: the
// update calls synchronise), what the RemoveLost / key-frame (or SwMarg) updates did, and the state afterwards.
struct FrameTiming { int k = 0; double ms = 0; int lost_rows = 0, lost_accepted = 0, select_rows = 0, n = 0, clones = 0; };

// Plays the stream of `cfg` straight into a filter (no file), timing every camera callback.  `truth_err` (may be NULL) receives
// the position error against the ground truth at the last frame.  Needs a GPU.
// `on_frame` (may be empty) is called after every PROCESSED camera frame with the frame index k and the filter (ingvio_replay --trace).
bool playSynth(const SynthConfig& cfg, const std::string& overrides, std::vector<FrameTiming>& timings, double* truth_err, std::string& err,
               const std::function<void(int, IngvioFilter&)>& on_frame = std::function<void(int, IngvioFilter&)>());

}  // namespace ingvio
