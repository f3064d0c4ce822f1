#include "SynthStream.h"

#include <cmath>
#include <cstdio>

namespace ingvio {

double SplitMix64::normal()
{
    // Box-Muller on two fresh draws, cosine branch only: no cached second value, so a consumer that skips a value skips a
    // fixed number of next() calls.  1 - u keeps the logarithm's argument in (0, 1].
    const double u1 = 1.0 - uniform(), u2 = uniform();
    return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586476925286766559 * u2);
}

namespace {

// config/sportsfield/stereo_{left,right}_config.yaml (T_cam_imu), the values ingvio_amd/synth.py carries too
const double R_CL2I[9] = { 0.9999890386957373, -0.0043227774403168, 0.0017989117755288, 0.0043276579084841, 0.9999869417854389,
                           -0.0027180205355500, -0.0017871388870994, 0.0027257758172719, 0.9999946881262878 };
const double T_CL2I[3] = { -0.0759472920952561, -0.0039320527565750, -0.0016395029500217 };
const double R_CR2I[9] = { 0.9999014076382304, -0.0133731297219721, 0.0042818692791948, 0.0133731003056063, 0.9999105754655292,
                           0.0000355022536769, -0.0042819611512717, 0.0000217631139403, 0.9999908321255077 };
const double T_CR2I[3] = { 0.0341738532732442, -0.0032623030537933, -0.0017782029037505 };

const double T_STATIC = 2.0;          // seconds of standstill before the motion starts
const int IMU_PER_FRAME = 10;         // 200 Hz / 20 Hz
const double DT_IMU = 0.005;

struct Truth { double R[9], p[3], v[3], w_body[3], f_body[3]; };

// circle of radius 5 m, angular rate ramping 0 -> 0.4 rad/s over 2 s; body axes: z forward along the tangent, y down
Truth truthAt(double tau)
{
    double th, thd, thdd;
    if (tau <= 0) { th = 0; thd = 0; thdd = 0; }
    else if (tau <= 2.0) { th = 0.1 * tau * tau; thd = 0.2 * tau; thdd = 0.2; }
    else { th = 0.4 + 0.4 * (tau - 2.0); thd = 0.4; thdd = 0.0; }
    const double c = std::cos(th), s = std::sin(th);
    Truth t;
    const double R[9] = { c, 0.0, -s, s, 0.0, c, 0.0, -1.0, 0.0 };
    for (int i = 0; i < 9; ++i) t.R[i] = R[i];
    t.p[0] = 5 * c; t.p[1] = 5 * s; t.p[2] = 1.0;
    t.v[0] = -5 * s * thd; t.v[1] = 5 * c * thd; t.v[2] = 0.0;
    const double a[3] = { -5 * c * thd * thd - 5 * s * thdd, -5 * s * thd * thd + 5 * c * thdd, 9.8 };      // + gravity: specific force
    const double w[3] = { 0.0, 0.0, thd };
    for (int i = 0; i < 3; ++i) {
        t.w_body[i] = R[0 + i] * w[0] + R[3 + i] * w[1] + R[6 + i] * w[2];      // R^T w
        t.f_body[i] = R[0 + i] * a[0] + R[3 + i] * a[1] + R[6 + i] * a[2];
    }
    return t;
}

void matmul3(const double* A, const double* B, double* C)
{
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}

// left-camera pose in the world at frame k
void camPose(int k, double* Rc, double* pc)
{
    const Truth t = truthAt(k * IMU_PER_FRAME * DT_IMU);
    matmul3(t.R, R_CL2I, Rc);
    for (int i = 0; i < 3; ++i) pc[i] = t.p[i] + t.R[3 * i] * T_CL2I[0] + t.R[3 * i + 1] * T_CL2I[1] + t.R[3 * i + 2] * T_CL2I[2];
}

// sub-stream `tag` / item `i` of frame interval k: every item is regenerable on its own
SplitMix64 sub(const SynthConfig& cfg, long long k, int tag, int i)
{
    SplitMix64 base(cfg.seed + (uint64_t)k);
    return SplitMix64(base.next() + ((uint64_t)tag << 40) + (uint64_t)i);
}

struct Track { long long birth; long long gen; };

Track trackOf(const SynthConfig& cfg, int k, int slot)
{
    const int phase = cfg.cohort ? ((cfg.life + 1 - cfg.birth_frame) % cfg.life + cfg.life) % cfg.life : slot % cfg.life;
    const long long gen = (k - 1 + phase) / cfg.life;
    return Track{ 1 - phase + gen * cfg.life, gen };
}

void worldPoint(const SynthConfig& cfg, const Track& t, int slot, double* pw)
{
    SplitMix64 r = sub(cfg, t.birth, 2, slot);
    const double d = r.uniform(3.0, 15.0), x = r.uniform(-0.5, 0.5) * d, y = r.uniform(-0.4, 0.4) * d;
    double Rc[9], pc[3];
    camPose((int)t.birth, Rc, pc);
    const double q[3] = { x, y, d };
    for (int i = 0; i < 3; ++i) pw[i] = pc[i] + Rc[3 * i] * q[0] + Rc[3 * i + 1] * q[1] + Rc[3 * i + 2] * q[2];
}

}  // namespace

namespace {

// ---- GNSS side (enable_gnss): local ENU at (31 N, 121.4 E, 30 m); the filter's world frame is the truth frame shifted to the starting
// point (the gravity initialisation of a level start reproduces the truth attitude up to the IMU noise in its average), rotated
// against ENU by the true yaw offset.  Formulas = what psr_res / dopp_res invert (gnss_spp.cpp:100-146, :256-282). ----
const double YO_TRUE = 0.3;
const int GN_NS = 8;
const int GN_SYS[GN_NS] = { 0, 0, 0, 0, 3, 3, 2, 2 };
const double GN_CB0[4] = { 150.0, 0.0, 165.0, 180.0 }, GN_FS = 5.0;

GvioAlignment synthAlignment()
{
    GvioAlignment al;
    const Vec3d lla(31.0, 121.4, 30.0);
    al.aligned = true; al.yaw_offset = YO_TRUE + 0.01; al.R_enu2ecef = gnss::geo2rotation(lla); al.anchor_ecef = gnss::geo2ecef(lla);
    return al;
}

void synthGnssEpoch(const SynthConfig& cfg, int k, double t, GnssMeas& gm, SppMeas& spp)
{
    const GvioAlignment al = synthAlignment();
    Mat3d Rz = Mat3d::Identity();
    Rz(0, 0) = std::cos(YO_TRUE); Rz(0, 1) = -std::sin(YO_TRUE); Rz(1, 0) = std::sin(YO_TRUE); Rz(1, 1) = std::cos(YO_TRUE);
    const Mat3d Rw2ecef = al.R_enu2ecef * Rz;
    const Truth tr = truthAt(t - T_STATIC), tr0 = truthAt(0.0);
    const Vec3d pw(tr.p[0] - tr0.p[0], tr.p[1] - tr0.p[1], tr.p[2] - tr0.p[2]), vw(tr.v[0], tr.v[1], tr.v[2]);
    const Vec3d rcv = Rw2ecef * pw + al.anchor_ecef, vel = Rw2ecef * vw;
    gm = GnssMeas(); gm.stamp = t;
    for (int i = 0; i < GN_NS; ++i) {
        SplitMix64 r = sub(cfg, k, 4, i);
        const double el = (25 + 50.0 * i / GN_NS) * M_PI / 180, az = 2 * M_PI * (i * 0.37 + 0.1);
        const Vec3d dir(std::cos(el) * std::sin(az), std::cos(el) * std::cos(az), std::sin(el));
        gnss::SatObs o;
        o.sys = GN_SYS[i]; o.freq = o.sys == 3 ? gnss::FREQ1_BDS : gnss::FREQ1; o.psr_std = 1.0; o.dopp_std = 0.5; o.ura = 2.0;
        o.sv_vel = al.R_enu2ecef * Vec3d(2500.0 * std::cos(az), -2500.0 * std::sin(az), 300.0 * (i % 3 - 1));
        o.sv_pos = al.anchor_ecef + al.R_enu2ecef * (dir * 2.2e7) + o.sv_vel * t;
        o.sv_dt = 1e-5 * (i + 1); o.sv_ddt = 1e-11 * i; o.tgd = 2e-9 * i; o.ion_delay = 2.0 + 0.3 * i; o.tro_delay = 2.5 + 0.2 * i;
        const Vec3d d = o.sv_pos - rcv;
        const double range = d.norm();
        const Vec3d u = d * (1.0 / range);
        const double cb = GN_CB0[o.sys] + GN_FS * t;
        const double sag = gnss::EARTH_OMG_GPS * (o.sv_pos[0] * rcv[1] - o.sv_pos[1] * rcv[0]) / gnss::LIGHT_SPEED;
        o.psr = range + sag + cb - o.sv_dt * gnss::LIGHT_SPEED + o.tro_delay + o.ion_delay + o.tgd * gnss::LIGHT_SPEED + 0.8 * r.normal();
        if (i == 5 && k >= 8) o.psr += 80.0;
        const double sagd = gnss::EARTH_OMG_GPS / gnss::LIGHT_SPEED * (o.sv_vel[0] * rcv[1] + o.sv_pos[0] * vel[1] - o.sv_vel[1] * rcv[0] - o.sv_pos[1] * vel[0]);
        const Vec3d dv = o.sv_vel - vel;
        const double est = dv[0] * u[0] + dv[1] * u[1] + dv[2] * u[2] + GN_FS + sagd - o.sv_ddt * gnss::LIGHT_SPEED;
        o.dopp = -(est + 0.05 * r.normal()) * o.freq / gnss::LIGHT_SPEED;
        gm.sats.push_back(o);
    }
    SplitMix64 r = sub(cfg, k, 5, 0);
    spp = SppMeas(); spp.stamp = t;
    for (int c = 0; c < 3; ++c) spp.posSpp[c] = rcv[c] + 2.0 * r.normal();
    for (int s4 = 0; s4 < 4; ++s4) { const double n = 3.0 * r.normal(); spp.posSpp[3 + s4] = GN_CB0[s4] == 0.0 ? 0.0 : GN_CB0[s4] + GN_FS * t + n; }
    for (int c = 0; c < 3; ++c) spp.velSpp[c] = vel[c];
    spp.velSpp[3] = GN_FS + 0.2 * r.normal();
}

}  // namespace

std::string synthParamsText(const SynthConfig& cfg)
{
    char buf[4096];
    const int window_cols = 21 + (cfg.enable_gnss ? 6 : 0) + 6 * (cfg.clones + 2);
    std::snprintf(buf, sizeof buf,
        "%%YAML:1.0\n# synthetic stream (SynthStream.h): config/sportsfield/ingvio_stereo.yaml values, window and tracker sizes of the BASELINE config\n"
        "cam_nums: %d\nmax_sliding_window_poses: %d\nis_key_frame: %d\nmax_landmark_features: 0\nenable_gnss: %d\n"
        "noise_gyro: 0.004\nnoise_accel: 0.08\nnoise_bias_gyro: 0.0002\nnoise_bias_accel: 0.008\n"
        "init_cov_rot: 0.0\ninit_cov_pos: 0.0\ninit_cov_vel: 0.25\ninit_cov_bg: 0.01\ninit_cov_ba: 0.01\ninit_cov_ext_rot: 1.8e-02\ninit_cov_ext_pos: 2e-03\n"
        "gravity_norm: 9.8\nmax_imu_buffer_size: 3000\ninit_imu_buffer_sp: 300\ntrans_thres: 0.25\nhuber_epsilon: 0.01\nconv_precision: 5e-07\n"
        "init_damping: 1e-03\nouter_loop_max_iter: 10\ninner_loop_max_iter: 10\nmax_depth: 40.0\nmin_depth: 0.2\nchi2_max_dof: 150\nchi2_thres: 0.95\n"
        "visual_noise: %.17g\nframe_select_interval: 18\n"
        "T_cl2i: %.16f %.16f %.16f %.16f %.16f %.16f %.16f %.16f %.16f %.16f %.16f %.16f\n"
        "T_cr2i: %.16f %.16f %.16f %.16f %.16f %.16f %.16f %.16f %.16f %.16f %.16f %.16f\n"
        "hip_f_max: %d\nhip_n_max: %d\n",
        cfg.stereo ? 2 : 1, cfg.clones, cfg.is_key_frame, cfg.enable_gnss, cfg.visual_noise,
        R_CL2I[0], R_CL2I[1], R_CL2I[2], T_CL2I[0], R_CL2I[3], R_CL2I[4], R_CL2I[5], T_CL2I[1], R_CL2I[6], R_CL2I[7], R_CL2I[8], T_CL2I[2],
        R_CR2I[0], R_CR2I[1], R_CR2I[2], T_CR2I[0], R_CR2I[3], R_CR2I[4], R_CR2I[5], T_CR2I[1], R_CR2I[6], R_CR2I[7], R_CR2I[8], T_CR2I[2],
        ((2 * cfg.feats + 15) / 16) * 16, ((window_cols + 15) / 16) * 16);
    return std::string(buf) + cfg.extra_params;
}

void synthFrame(const SynthConfig& cfg, int k, msg::StereoFrame& out)
{
    out.stereo_features.clear();
    out.header.seq = (uint32_t)k;
    out.header.stamp = msg::Time::fromNSec((uint64_t)std::llround((T_STATIC + k * IMU_PER_FRAME * DT_IMU) * 1e9));
    out.header.frame_id = "cam0";
    double Rc[9], pc[3];
    camPose(k, Rc, pc);
    // T_cl2cr = T_cr2i^-1 T_cl2i (State.cpp:33)
    double Rlr[9], tlr[3];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) Rlr[3 * i + j] = R_CR2I[i] * R_CL2I[j] + R_CR2I[3 + i] * R_CL2I[3 + j] + R_CR2I[6 + i] * R_CL2I[6 + j];
        tlr[i] = R_CR2I[i] * (T_CL2I[0] - T_CR2I[0]) + R_CR2I[3 + i] * (T_CL2I[1] - T_CR2I[1]) + R_CR2I[6 + i] * (T_CL2I[2] - T_CR2I[2]);
    }
    for (int slot = 0; slot < cfg.feats; ++slot) {
        const Track tr = trackOf(cfg, k, slot);
        double pw[3];
        worldPoint(cfg, tr, slot, pw);
        double q[3], qr[3];
        for (int i = 0; i < 3; ++i) q[i] = Rc[i] * (pw[0] - pc[0]) + Rc[3 + i] * (pw[1] - pc[1]) + Rc[6 + i] * (pw[2] - pc[2]);
        for (int i = 0; i < 3; ++i) qr[i] = Rlr[3 * i] * q[0] + Rlr[3 * i + 1] * q[1] + Rlr[3 * i + 2] * q[2] + tlr[i];
        SplitMix64 r = sub(cfg, k, 3, slot);
        msg::StereoMeas m;
        m.id = (uint64_t)(tr.gen * cfg.feats + slot + 1);
        m.u0 = q[0] / q[2] + cfg.pixel_noise * r.normal();
        m.v0 = q[1] / q[2] + cfg.pixel_noise * r.normal();
        m.u1 = qr[0] / qr[2] + cfg.pixel_noise * r.normal();
        m.v1 = qr[1] / qr[2] + cfg.pixel_noise * r.normal();
        if (cfg.outlier_every > 0 && (m.id - 1) % (uint64_t)cfg.outlier_every == 0 && k - tr.birth == cfg.life / 2) m.u0 += 0.5;
        out.stereo_features.push_back(m);
    }
}

int synthStream(const SynthConfig& cfg, SynthSink& sink)
{
    sink.params(synthParamsText(cfg));
    const int n_static = (int)std::llround(T_STATIC / DT_IMU);
    const int n_imu = n_static + cfg.frames * IMU_PER_FRAME;
    int frames = 0;
    {   // the tracker is up before the filter: IngvioFilter::callbackIMU drops every IMU sample that arrives before the first image
        // (IngvioFilter.cpp:393), and the first image itself only sets the flag (:257-261) - one header-only frame at t = 0
        msg::StereoFrame f0;
        f0.header.seq = 0; f0.header.stamp = msg::Time::fromNSec(0); f0.header.frame_id = "cam0";
        if (cfg.stereo) sink.stereo(f0);
        else { msg::MonoFrame m0; m0.header = f0.header; sink.mono(m0); }
        if (cfg.enable_gnss) sink.alignment(synthAlignment(), 0.0);
    }
    for (int n = 1; n <= n_imu; ++n) {
        const double t = n * DT_IMU;
        const long long k = (n - 1) / IMU_PER_FRAME - n_static / IMU_PER_FRAME + 1;      // frame interval the sample leads to (<= 0: static phase)
        const Truth tr = truthAt(t - T_STATIC);
        SplitMix64 r = sub(cfg, k, 1, (n - 1) % IMU_PER_FRAME);
        msg::Imu m;
        m.header.seq = (uint32_t)n;
        m.header.stamp = msg::Time::fromNSec((uint64_t)std::llround(t * 1e9));
        m.angular_velocity.x = tr.w_body[0] + 0.004 * r.normal();
        m.angular_velocity.y = tr.w_body[1] + 0.004 * r.normal();
        m.angular_velocity.z = tr.w_body[2] + 0.004 * r.normal();
        m.linear_acceleration.x = tr.f_body[0] + 0.08 * r.normal();
        m.linear_acceleration.y = tr.f_body[1] + 0.08 * r.normal();
        m.linear_acceleration.z = tr.f_body[2] + 0.08 * r.normal();
        sink.imu(m);
        if (n > n_static && (n - n_static) % IMU_PER_FRAME == 0) {
            const int kf = (n - n_static) / IMU_PER_FRAME;
            if (cfg.enable_gnss) {                        // the epoch of this frame time arrives before the frame (GnssSync buffers it)
                GnssMeas gm; SppMeas sp;
                synthGnssEpoch(cfg, kf, t, gm, sp);
                sink.gnss(gm); sink.spp(sp);
            }
            msg::StereoFrame f;
            synthFrame(cfg, kf, f);
            if (cfg.stereo) sink.stereo(f);
            else {
                msg::MonoFrame mf;
                mf.header = f.header;
                for (const auto& s : f.stereo_features) { msg::MonoMeas mm; mm.id = s.id; mm.u0 = s.u0; mm.v0 = s.v0; mf.mono_features.push_back(mm); }
                sink.mono(mf);
            }
            // ground truth of the IMU frame: quaternion xyzw of R_i2w
            const double* R = tr.R;
            double q[4];
            const double w = std::sqrt(std::fmax(0.0, 1 + R[0] + R[4] + R[8])) / 2;
            if (w > 1e-6) { q[0] = (R[7] - R[5]) / (4 * w); q[1] = (R[2] - R[6]) / (4 * w); q[2] = (R[3] - R[1]) / (4 * w); q[3] = w; }
            else { const double x = std::sqrt(std::fmax(0.0, 1 + R[0] - R[4] - R[8])) / 2; q[0] = x; q[1] = (R[1] + R[3]) / (4 * x); q[2] = (R[2] + R[6]) / (4 * x); q[3] = (R[7] - R[5]) / (4 * x); }
            sink.truth(t, tr.p, q);
            ++frames;
        }
    }
    return frames;
}

}  // namespace ingvio
