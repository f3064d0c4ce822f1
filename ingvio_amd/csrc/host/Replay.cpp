#include "Replay.h"
#include "StateManager.h"

#include <chrono>
#include <cmath>
#include <cstring>
#include <sstream>

namespace ingvio {

namespace {
template <class T> void app(std::vector<uint8_t>& b, const T& v) { const uint8_t* p = reinterpret_cast<const uint8_t*>(&v); b.insert(b.end(), p, p + sizeof(T)); }
struct Cur {
    const std::vector<uint8_t>& b; size_t o = 0; bool ok = true;
    template <class T> T get() { T v{}; if (o + sizeof(T) > b.size()) { ok = false; return v; } std::memcpy(&v, b.data() + o, sizeof(T)); o += sizeof(T); return v; }
};
const char kMagic[8] = { 'I', 'N', 'G', 'V', 'I', 'O', 'R', '1' };
}  // namespace

bool ReplayReader::open(const std::string& path)
{
    _f = std::fopen(path.c_str(), "rb");
    if (!_f) { _err = "cannot open " + path; return false; }
    char m[8];
    if (std::fread(m, 1, 8, _f) != 8 || std::memcmp(m, kMagic, 8) != 0) { _err = "not an INGVIOR1 replay file"; return false; }
    return true;
}

bool ReplayReader::next(ReplayRecord& rec)
{
    uint8_t hdr[13];
    const size_t got = std::fread(hdr, 1, 13, _f);
    if (got == 0) return false;
    if (got != 13) { _err = "truncated record header"; return false; }
    rec.type = hdr[0];
    uint32_t n;
    std::memcpy(&rec.stamp_ns, hdr + 1, 8); std::memcpy(&n, hdr + 9, 4);
    if (rec.type >= RP_COUNT || n > (1u << 26)) { _err = "malformed record"; return false; }
    rec.payload.resize(n);
    if (n && std::fread(rec.payload.data(), 1, n, _f) != n) { _err = "truncated payload"; return false; }
    return true;
}

bool ReplayWriter::open(const std::string& path)
{
    _f = std::fopen(path.c_str(), "wb");
    if (!_f) return false;
    std::fwrite(kMagic, 1, 8, _f);
    return true;
}
void ReplayWriter::put(uint8_t type, uint64_t ns, const std::vector<uint8_t>& p)
{
    const uint32_t n = (uint32_t)p.size();
    std::fwrite(&type, 1, 1, _f); std::fwrite(&ns, 8, 1, _f); std::fwrite(&n, 4, 1, _f);
    if (n) std::fwrite(p.data(), 1, n, _f);
}
void ReplayWriter::params(const std::string& t) { put(RP_PARAMS, 0, std::vector<uint8_t>(t.begin(), t.end())); }
void ReplayWriter::imu(const msg::Imu& m)
{
    std::vector<uint8_t> b;
    for (double v : { m.angular_velocity.x, m.angular_velocity.y, m.angular_velocity.z, m.linear_acceleration.x, m.linear_acceleration.y, m.linear_acceleration.z }) app(b, v);
    put(RP_IMU, m.header.stamp.toNSec(), b);
}
void ReplayWriter::mono(const msg::MonoFrame& m)
{
    std::vector<uint8_t> b;
    app(b, (uint32_t)m.mono_features.size());
    for (const auto& f : m.mono_features) { app(b, f.id); app(b, f.u0); app(b, f.v0); }
    put(RP_MONO_FRAME, m.header.stamp.toNSec(), b);
}
void ReplayWriter::stereo(const msg::StereoFrame& m)
{
    std::vector<uint8_t> b;
    app(b, (uint32_t)m.stereo_features.size());
    for (const auto& f : m.stereo_features) { app(b, f.id); app(b, f.u0); app(b, f.v0); app(b, f.u1); app(b, f.v1); }
    put(RP_STEREO_FRAME, m.header.stamp.toNSec(), b);
}
void ReplayWriter::gnss(const GnssMeas& m)
{
    std::vector<uint8_t> b;
    app(b, (uint32_t)m.sats.size());
    for (const auto& s : m.sats) {
        app(b, (int32_t)s.sys);
        for (double v : { s.psr, s.dopp, s.psr_std, s.dopp_std, s.freq, s.sv_pos[0], s.sv_pos[1], s.sv_pos[2], s.sv_vel[0], s.sv_vel[1], s.sv_vel[2],
                          s.sv_dt, s.sv_ddt, s.tgd, s.ura, s.ion_delay, s.tro_delay }) app(b, v);
    }
    put(RP_GNSS_MEAS, (uint64_t)std::llround(m.stamp * 1e9), b);
}
void ReplayWriter::spp(const SppMeas& m)
{
    std::vector<uint8_t> b;
    for (double v : m.posSpp) app(b, v);
    for (double v : m.velSpp) app(b, v);
    put(RP_SPP_MEAS, (uint64_t)std::llround(m.stamp * 1e9), b);
}
void ReplayWriter::alignment(const GvioAlignment& a, double stamp)
{
    std::vector<uint8_t> b;
    app(b, a.aligned ? 1.0 : 0.0); app(b, a.yaw_offset);
    for (double v : a.R_enu2ecef.m) app(b, v);
    for (int i = 0; i < 3; ++i) app(b, a.anchor_ecef[i]);
    put(RP_ALIGNMENT, (uint64_t)std::llround(stamp * 1e9), b);
}
void ReplayWriter::gnssRaw(double stamp, double doy, const double iono[8], const std::vector<double>& eph, const std::vector<double>& obs)
{
    std::vector<uint8_t> b;
    app(b, doy);
    for (int i = 0; i < 8; ++i) app(b, iono[i]);
    const uint32_t n = (uint32_t)(obs.size() / INGVIO_OBS_N);
    app(b, n);
    for (uint32_t i = 0; i < n; ++i) {
        for (int q = 0; q < INGVIO_EPH_N; ++q) app(b, eph[(size_t)i * INGVIO_EPH_N + q]);
        for (int q = 0; q < INGVIO_OBS_N; ++q) app(b, obs[(size_t)i * INGVIO_OBS_N + q]);
    }
    put(RP_GNSS_RAW, (uint64_t)std::llround(stamp * 1e9), b);
}
void ReplayWriter::truth(double stamp, const double p[3], const double q[4])
{
    std::vector<uint8_t> b;
    for (int i = 0; i < 3; ++i) app(b, p[i]);
    for (int i = 0; i < 4; ++i) app(b, q[i]);
    put(RP_GROUND_TRUTH, (uint64_t)std::llround(stamp * 1e9), b);
}

bool decodeGnssRaw(const ReplayRecord& r, GnssMeas& m)
{
    Cur c{ r.payload };
    m.doy = c.get<double>();
    m.iono.assign(8, 0.0);
    for (int i = 0; i < 8; ++i) m.iono[i] = c.get<double>();
    const uint32_t n = c.get<uint32_t>();
    if (!c.ok || (size_t)n * 8 * (INGVIO_EPH_N + INGVIO_OBS_N) + 76 != r.payload.size() || n > INGVIO_GNSS_MAX_SAT) return false;
    m.raw_eph.assign((size_t)n * INGVIO_EPH_N, 0.0); m.raw_obs.assign((size_t)n * INGVIO_OBS_N, 0.0);
    for (uint32_t i = 0; i < n; ++i) {
        for (int q = 0; q < INGVIO_EPH_N; ++q) m.raw_eph[(size_t)i * INGVIO_EPH_N + q] = c.get<double>();
        for (int q = 0; q < INGVIO_OBS_N; ++q) m.raw_obs[(size_t)i * INGVIO_OBS_N + q] = c.get<double>();
    }
    return c.ok && c.o == r.payload.size();
}
bool decodeImu(const ReplayRecord& r, msg::Imu& m)
{
    Cur c{ r.payload };
    m.header.stamp = msg::Time::fromNSec(r.stamp_ns);
    m.angular_velocity.x = c.get<double>(); m.angular_velocity.y = c.get<double>(); m.angular_velocity.z = c.get<double>();
    m.linear_acceleration.x = c.get<double>(); m.linear_acceleration.y = c.get<double>(); m.linear_acceleration.z = c.get<double>();
    return c.ok && c.o == r.payload.size();
}
bool decodeMono(const ReplayRecord& r, msg::MonoFrame& m)
{
    Cur c{ r.payload };
    m.header.stamp = msg::Time::fromNSec(r.stamp_ns);
    const uint32_t n = c.get<uint32_t>();
    if (!c.ok || (size_t)n * 24 + 4 != r.payload.size()) return false;
    m.mono_features.resize(n);
    for (auto& f : m.mono_features) { f.id = c.get<uint64_t>(); f.u0 = c.get<double>(); f.v0 = c.get<double>(); }
    return c.ok;
}
bool decodeStereo(const ReplayRecord& r, msg::StereoFrame& m)
{
    Cur c{ r.payload };
    m.header.stamp = msg::Time::fromNSec(r.stamp_ns);
    const uint32_t n = c.get<uint32_t>();
    if (!c.ok || (size_t)n * 40 + 4 != r.payload.size()) return false;
    m.stereo_features.resize(n);
    for (auto& f : m.stereo_features) { f.id = c.get<uint64_t>(); f.u0 = c.get<double>(); f.v0 = c.get<double>(); f.u1 = c.get<double>(); f.v1 = c.get<double>(); }
    return c.ok;
}
bool decodeGnss(const ReplayRecord& r, GnssMeas& m)
{
    Cur c{ r.payload };
    m.stamp = 1e-9 * (double)r.stamp_ns;
    const uint32_t n = c.get<uint32_t>();
    if (!c.ok || (size_t)n * (4 + 17 * 8) + 4 != r.payload.size()) return false;
    m.sats.resize(n);
    for (auto& s : m.sats) {
        s.sys = c.get<int32_t>();
        s.psr = c.get<double>(); s.dopp = c.get<double>(); s.psr_std = c.get<double>(); s.dopp_std = c.get<double>(); s.freq = c.get<double>();
        for (int i = 0; i < 3; ++i) s.sv_pos[i] = c.get<double>();
        for (int i = 0; i < 3; ++i) s.sv_vel[i] = c.get<double>();
        s.sv_dt = c.get<double>(); s.sv_ddt = c.get<double>(); s.tgd = c.get<double>(); s.ura = c.get<double>();
        s.ion_delay = c.get<double>(); s.tro_delay = c.get<double>();
    }
    return c.ok;
}
bool decodeSpp(const ReplayRecord& r, SppMeas& m)
{
    Cur c{ r.payload };
    m.stamp = 1e-9 * (double)r.stamp_ns;
    for (double& v : m.posSpp) v = c.get<double>();
    for (double& v : m.velSpp) v = c.get<double>();
    return c.ok && c.o == r.payload.size();
}
bool decodeAlignment(const ReplayRecord& r, GvioAlignment& a)
{
    Cur c{ r.payload };
    a.aligned = c.get<double>() != 0.0; a.yaw_offset = c.get<double>();
    for (double& v : a.R_enu2ecef.m) v = c.get<double>();
    for (int i = 0; i < 3; ++i) a.anchor_ecef[i] = c.get<double>();
    return c.ok && c.o == r.payload.size();
}

bool applyParamsText(const std::string& text, IngvioParams& p)
{
    std::istringstream in(text);
    std::string line;
    bool ok = true;
    while (std::getline(in, line)) {
        const size_t colon = line.find(':');
        if (colon == std::string::npos || line[0] == '#' || line[0] == '%') continue;
        std::string key = line.substr(0, colon), val = line.substr(colon + 1);
        while (!key.empty() && (key.back() == ' ' || key.back() == '\t')) key.pop_back();
        std::istringstream vs(val);
        auto num = [&](double& d) { vs >> d; };
        auto inum = [&](int& d) { double t = d; vs >> t; d = (int)t; };
        auto iso = [&](Iso3& T) { double a[12]; for (double& x : a) vs >> x; for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) T.R(r, c) = a[4 * r + c]; T.t[r] = a[4 * r + 3]; } };
        if (key == "cam_nums") inum(p._cam_nums);
        else if (key == "max_sliding_window_poses") inum(p._max_sw_clones);
        else if (key == "is_key_frame") inum(p._is_key_frame);
        else if (key == "max_landmark_features") inum(p._max_lm_feats);
        else if (key == "gv_align_batch_size") inum(p._gv_align_batch_size);
        else if (key == "gv_align_max_iter") inum(p._gv_align_max_iter);
        else if (key == "gv_align_conv_epsilon") num(p._gv_align_conv_epsilon);
        else if (key == "gv_align_vel_thres") num(p._gv_align_vel_thres);
        else if (key == "enable_gnss") inum(p._enable_gnss);
        else if (key == "noise_gyro") num(p._noise_g);
        else if (key == "noise_accel") num(p._noise_a);
        else if (key == "noise_bias_gyro") num(p._noise_bg);
        else if (key == "noise_bias_accel") num(p._noise_ba);
        else if (key == "noise_rcv_clockbias") num(p._noise_clockbias);
        else if (key == "noise_rcv_clockbias_randomwalk") num(p._noise_cb_rw);
        else if (key == "init_cov_rot") num(p._init_cov_rot);
        else if (key == "init_cov_pos") num(p._init_cov_pos);
        else if (key == "init_cov_vel") num(p._init_cov_vel);
        else if (key == "init_cov_bg") num(p._init_cov_bg);
        else if (key == "init_cov_ba") num(p._init_cov_ba);
        else if (key == "init_cov_ext_rot") num(p._init_cov_ext_rot);
        else if (key == "init_cov_ext_pos") num(p._init_cov_ext_pos);
        else if (key == "init_cov_rcv_clockbias") num(p._init_cov_rcv_clockbias);
        else if (key == "init_cov_rcv_clockbias_randomwalk") num(p._init_cov_rcv_clockbias_randomwalk);
        else if (key == "init_cov_yof") num(p._init_cov_yof);
        else if (key == "gravity_norm") num(p._init_gravity);
        else if (key == "max_imu_buffer_size") inum(p._max_imu_buffer_size);
        else if (key == "init_imu_buffer_sp") inum(p._init_imu_buffer_sp);
        else if (key == "trans_thres") num(p._trans_thres);
        else if (key == "huber_epsilon") num(p._huber_epsilon);
        else if (key == "conv_precision") num(p._conv_precision);
        else if (key == "init_damping") num(p._init_damping);
        else if (key == "outer_loop_max_iter") inum(p._outer_loop_max_iter);
        else if (key == "inner_loop_max_iter") inum(p._inner_loop_max_iter);
        else if (key == "max_depth") num(p._max_depth);
        else if (key == "min_depth") num(p._min_depth);
        else if (key == "chi2_max_dof") inum(p._chi2_max_dof);
        else if (key == "chi2_thres") num(p._chi2_thres);
        else if (key == "visual_noise") num(p._visual_noise);
        else if (key == "frame_select_interval") inum(p._frame_select_interval);
        else if (key == "gnss_chi2_test") inum(p._is_gnss_chi2_test);
        else if (key == "gnss_strong_reject") inum(p._is_gnss_strong_reject);
        else if (key == "psr_noise_amp") num(p._psr_noise_amp);
        else if (key == "dopp_noise_amp") num(p._dopp_noise_amp);
        else if (key == "is_adjust_yof") inum(p._is_adjust_yof);
        else if (key == "use_fix_time_offset") inum(p._use_fix_time_offset);
        else if (key == "gnss_local_offset") num(p._gnss_local_offset);
        else if (key == "T_cl2i") iso(p._T_cl2i);
        else if (key == "T_cr2i") iso(p._T_cr2i);
        else if (key == "hip_n_max") inum(p._hip_n_max);
        else if (key == "hip_f_max") inum(p._hip_f_max);
        else if (key == "hip_device") inum(p._hip_device);
        else if (key == "hip_max_valid_ids") inum(p._hip_max_valid_ids);
        else if (key == "hip_compress_rule") inum(p._hip_compress_rule);
        else if (key == "hip_fuse_triangulation") inum(p._hip_fuse_triangulation);
        else continue;                             // topics, tracker and aligner keys: not read by this path
        if (vs.fail()) ok = false;
    }
    return ok;
}

bool replayFile(const std::string& path, const std::string& overrides, bool dump_only,
                const std::function<void(const msg::Odometry&, const IngvioFilter&)>& on_odom, ReplayStats& st, std::string& err)
{
    ReplayReader rd;
    if (!rd.open(path)) { err = rd.error(); return false; }
    IngvioParams fp;
    std::unique_ptr<IngvioFilter> filter;
    ReplayRecord rec;
    bool first = true;
    GnssMeas raw_pending;                         // a GNSS_RAW record waits for the GNSS_MEAS record of the same stamp
    uint64_t raw_stamp = 0;
    bool have_raw = false;
    auto make_filter = [&]() {
        if (filter || dump_only) return;
        applyParamsText(overrides, fp);
        // device capacity from the window: 21 + 6 GNSS scalars + 6 (C + 2) + landmarks
        if (fp._hip_n_max < 21 + 6 + 6 * (fp._max_sw_clones + 2) + 3 * fp._max_lm_feats) fp._hip_n_max = 21 + 6 + 6 * (fp._max_sw_clones + 2) + 3 * fp._max_lm_feats + 16;
        filter.reset(new IngvioFilter(fp, std::make_shared<Triangulator>(fp)));
        if (fp._enable_gnss) filter->gnssSync()->setSync();          // stamps in the file are local times already (GnssSync::storeTimePair is ROS plumbing)
    };
    while (rd.next(rec)) {
        ++st.counts[rec.type];
        const double t = 1e-9 * (double)rec.stamp_ns;
        if (rec.type != RP_PARAMS) { if (first) { st.t_first = t; first = false; } st.t_last = t; }
        switch (rec.type) {
        case RP_PARAMS:
            if (filter) { err = "PARAMS record after the first message"; return false; }
            if (!applyParamsText(std::string(rec.payload.begin(), rec.payload.end()), fp)) { err = "unreadable value in the PARAMS record"; return false; }
            break;
        case RP_IMU: {
            msg::Imu m;
            if (!decodeImu(rec, m)) { err = "bad IMU record"; return false; }
            make_filter();
            if (filter) filter->callbackIMU(m);
            break;
        }
        case RP_MONO_FRAME: case RP_STEREO_FRAME: {
            msg::MonoFrame mm; msg::StereoFrame sm;
            const bool mono = rec.type == RP_MONO_FRAME;
            if (mono ? !decodeMono(rec, mm) : !decodeStereo(rec, sm)) { err = "bad frame record"; return false; }
            st.features += mono ? mm.mono_features.size() : sm.stereo_features.size();
            make_filter();
            if (!filter) break;
            const int before = filter->framesProcessed();
            if (mono) filter->callbackMonoFrame(mm); else filter->callbackStereoFrame(sm);
            if (filter->framesProcessed() > before) {
                ++st.frames_processed;
                msg::Odometry od;
                if (filter->odometry(mono ? mm.header : sm.header, od) && on_odom) on_odom(od, *filter);
            }
            break;
        }
        case RP_GNSS_RAW: { if (!decodeGnssRaw(rec, raw_pending)) { err = "bad GNSS_RAW record"; return false; } raw_stamp = rec.stamp_ns; have_raw = true; break; }
        case RP_GNSS_MEAS: {
            GnssMeas g;
            if (!decodeGnss(rec, g)) { err = "bad GNSS record"; return false; }
            if (have_raw && raw_stamp == rec.stamp_ns) { g.raw_eph = raw_pending.raw_eph; g.raw_obs = raw_pending.raw_obs; g.iono = raw_pending.iono; g.doy = raw_pending.doy; }
            have_raw = false;
            make_filter();
            if (filter) filter->callbackGnssMeas(g);
            break;
        }
        case RP_SPP_MEAS: { SppMeas s; if (!decodeSpp(rec, s)) { err = "bad SPP record"; return false; } make_filter(); if (filter) filter->callbackSppMeas(s); break; }
        case RP_ALIGNMENT: { GvioAlignment a; if (!decodeAlignment(rec, a)) { err = "bad ALIGNMENT record"; return false; } make_filter(); if (filter) filter->setGnssAlignment(a); break; }
        default: break;
        }
    }
    if (!rd.error().empty()) { err = rd.error(); return false; }
    return true;
}

namespace {

struct FileSink : SynthSink {
    ReplayWriter w;
    void params(const std::string& t) override { w.params(t); }
    void imu(const msg::Imu& m) override { w.imu(m); }
    void stereo(const msg::StereoFrame& m) override { w.stereo(m); }
    void mono(const msg::MonoFrame& m) override { w.mono(m); }
    void truth(double stamp, const double p[3], const double q[4]) override { w.truth(stamp, p, q); }
    void gnss(const GnssMeas& m) override { w.gnss(m); }
    void spp(const SppMeas& m) override { w.spp(m); }
    void alignment(const GvioAlignment& a, double stamp) override { w.alignment(a, stamp); }
};

struct FilterSink : SynthSink {
    std::string overrides;
    std::unique_ptr<IngvioFilter> filter;
    IngvioParams fp;
    std::vector<FrameTiming>* out = nullptr;
    std::function<void(int, IngvioFilter&)> on_frame;
    double last_truth[3] = { 0, 0, 0 }, first_truth[3] = { 0, 0, 0 }, first_p[3] = { 0, 0, 0 };
    bool have_first = false;
    bool ok = true;
    std::string err;
    void params(const std::string& t) override
    {
        if (!applyParamsText(t, fp) || !applyParamsText(overrides, fp)) { ok = false; err = "unreadable value in the parameters"; return; }
        filter.reset(new IngvioFilter(fp, std::make_shared<Triangulator>(fp)));
        if (fp._enable_gnss) filter->gnssSync()->setSync();
    }
    void imu(const msg::Imu& m) override { if (filter) filter->callbackIMU(m); }
    void gnss(const GnssMeas& m) override { if (filter) filter->callbackGnssMeas(m); }
    void spp(const SppMeas& m) override { if (filter) filter->callbackSppMeas(m); }
    void alignment(const GvioAlignment& a, double) override { if (filter) filter->setGnssAlignment(a); }
    template <class F> void frame(const F& m, uint32_t k, bool is_mono)
    {
        if (!filter) return;
        const int before = filter->framesProcessed();
        const auto t0 = std::chrono::steady_clock::now();
        callFrame(m);
        ingvio_sync(StateManager::ctx(filter->state()));      // the covariance calls without results only enqueue: the frame ends when the device is done
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (filter->framesProcessed() > before && !have_first) {        // displacement reference: the first processed frame (its truth record follows)
            const Vec3d p = filter->state()->_extended_pose->valueTrans1();
            for (int i = 0; i < 3; ++i) first_p[i] = p[i];
            have_first = true; want_first_truth = true;
        }
        if (filter->framesProcessed() > before && out) {
            FrameTiming ft;
            ft.k = (int)k; ft.ms = ms;
            ft.lost_rows = filter->removeLostUpdate()->lastRows(); ft.lost_accepted = filter->removeLostUpdate()->lastAccepted();
            ft.select_rows = fp._is_key_frame ? filter->keyframeUpdate()->lastRows() : filter->swMargUpdate()->lastRows();
            ft.n = filter->state()->curr_cov_size(); ft.clones = (int)filter->state()->_sw_camleft_poses.size();
            out->push_back(ft);
        }
        if (filter->framesProcessed() > before && on_frame) on_frame((int)k, *filter);
        (void)is_mono;
    }
    void callFrame(const msg::StereoFrame& m) { filter->callbackStereoFrame(m); }
    void callFrame(const msg::MonoFrame& m) { filter->callbackMonoFrame(m); }
    void stereo(const msg::StereoFrame& m) override { frame(m, m.header.seq, false); }
    void mono(const msg::MonoFrame& m) override { frame(m, m.header.seq, true); }
    bool want_first_truth = false;
    void truth(double, const double p[3], const double*) override
    {
        for (int i = 0; i < 3; ++i) last_truth[i] = p[i];
        if (want_first_truth) { for (int i = 0; i < 3; ++i) first_truth[i] = p[i]; want_first_truth = false; }
    }
};

}  // namespace

bool writeSynthRecording(const SynthConfig& cfg, const std::string& path)
{
    FileSink s;
    if (!s.w.open(path)) return false;
    synthStream(cfg, s);
    s.w.close();
    return true;
}

bool playSynth(const SynthConfig& cfg, const std::string& overrides, std::vector<FrameTiming>& timings, double* truth_err, std::string& err,
               const std::function<void(int, IngvioFilter&)>& on_frame)
{
    FilterSink s;
    s.overrides = overrides; s.out = &timings; s.on_frame = on_frame;
    synthStream(cfg, s);
    if (!s.ok || !s.filter) { err = s.err.empty() ? "no filter was built" : s.err; return false; }
    if (truth_err) {
        // the filter's world frame starts at the origin with the gravity-aligned attitude: compare displacements from the first processed frame
        const Vec3d p = s.filter->state()->_extended_pose->valueTrans1();
        double e2 = 0;
        for (int i = 0; i < 3; ++i) { const double d = (p[i] - s.first_p[i]) - (s.last_truth[i] - s.first_truth[i]); e2 += d * d; }
        *truth_err = std::sqrt(e2);
    }
    return true;
}

}  // namespace ingvio
