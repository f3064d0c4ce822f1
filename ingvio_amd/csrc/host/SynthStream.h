// SynthStream.h — deterministic synthetic sensor streams for the filter's callback surface ("generator implemented
// identically in C++ ... SplitMix64, seed = 0x1A6F10 + frame_id").  No reference counterpart: the reference is fed from rosbags.
//
// A stream is what /imu0 and /stereo_tracker/stereo_feature would carry for a camera on a circle (r = 5 m, angular rate ramping
// to 0.4 rad/s = 2 m/s, IMU 200 Hz, frames 20 Hz, 2 s static first for the gravity initialisation, IngvioFilter.cpp:396-406):
//   * one header-only camera frame at t = 0 comes first: the filter drops IMU samples until the first image has arrived
//     (IngvioFilter.cpp:393) and that image only raises the flag (:257-261) - without it the static phase would be lost;
//   * every random draw of frame interval k (its 10 IMU samples, the features born at frame k, the pixel noise of frame k) comes
//     from SplitMix64(seed + k) in a fixed order, so any frame can be regenerated on its own, in any language;
//   * feature tracks: `cohort` = all F tracks are born together and lost together after `life` frames (the frame after a cohort
//     dies carries an F-feature RemoveLost update: the heavy frame of BASELINE configs 2 / 5); otherwise the deaths are spread
//     evenly (F / life tracks lost per frame: the steady state of a tracker);
//   * every `outlier_every`-th track gets one gross outlier (+0.5 in u0 at the middle of its life) so the chi^2 gate has work.
// The records go to a sink with the ReplayWriter's methods, i.e. to an INGVIOR1 file or straight into a filter.
#pragma once
#include <cstdint>
#include <string>

#include "GnssUpdate.h"
#include "Messages.h"

namespace ingvio {

struct SplitMix64 {
    uint64_t s;
    explicit SplitMix64(uint64_t seed) : s(seed) {}
    uint64_t next()
    {
        uint64_t z = (s += 0x9E3779B97F4A7C15ULL);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        return z ^ (z >> 31);
    }
    double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }      // [0, 1), 53 bits
    double uniform(double a, double b) { return a + (b - a) * uniform(); }
    double normal();                                                                    // Box-Muller, two draws per value, no cache
};

struct SynthConfig {
    int feats = 150;              // tracks visible per frame (feature_tracker's max_pts_frame)
    int clones = 11;              // max_sliding_window_poses
    int stereo = 1;
    int is_key_frame = 1;
    int life = 10;                // frames a track lives (observations a lost feature carries); <= clones - 1 in key-frame mode
    int cohort = 1;
    int birth_frame = 3;          // cohort mode: the cohorts are born at frames birth_frame + m * life.  In key-frame mode the clone of
                                  // every other frame is marginalised one frame later and takes the never-triangulated tracks anchored
                                  // at it along (KeyframeUpdate.cpp:280-328): tracks born on those frames never reach an update.
                                  // The first processed frame is frame 1 (the image that raises _hasImageCome is the header-only one
                                  // at t = 0), so the window first holds `clones` poses at frame `clones` and marginalises from then on
                                  // every second frame, always the clone of the frame before: births must have the parity of `clones`
                                  // (3 for an 11-pose window, 2 for a 30-pose one)
    int outlier_every = 20;
    int frames = 60;              // camera frames after the static phase
    double pixel_noise = 1e-3;    // normalised image coordinates
    double visual_noise = 0.08;
    uint64_t seed = 0x1A6F10ULL;
    int enable_gnss = 0;          // 1: every camera frame k >= 1 is preceded by a GNSS epoch (8 satellites: GPS x4, BDS x2, GAL x2; pseudo-range and
                                  // Doppler with the satellite states and atmosphere delays already evaluated, as the replay format carries them) and
                                  // its SPP fix; the stream opens with the ENU alignment (yaw offset 0.31 against a true 0.30).  Satellite 5 carries an
                                  // 80 m pseudo-range outlier from frame 8 on (the per-row gate of gnss_chi2_test has work)
    std::string extra_params;     // appended "key: value" lines
};

struct SynthSink {
    virtual ~SynthSink() {}
    virtual void params(const std::string& text) = 0;
    virtual void imu(const msg::Imu& m) = 0;
    virtual void stereo(const msg::StereoFrame& m) = 0;
    virtual void mono(const msg::MonoFrame& m) = 0;
    virtual void truth(double stamp, const double p[3], const double q_xyzw[4]) = 0;
    virtual void gnss(const GnssMeas&) {}
    virtual void spp(const SppMeas&) {}
    virtual void alignment(const GvioAlignment&, double /*stamp*/) {}
};

// The PARAMS text of a config (key names of config/*/ingvio_stereo.yaml; extrinsics of config/sportsfield/stereo_*_config.yaml).
std::string synthParamsText(const SynthConfig& cfg);

// Generates the whole stream into `sink`.  Returns the number of camera frames written.
int synthStream(const SynthConfig& cfg, SynthSink& sink);

// One frame's feature message alone (frame index k >= 1 counted from the end of the static phase), regenerated from the seed:
// what `ingvio_replay --synth-frame k` prints and tests/golden/synth_frame.npz pins.
void synthFrame(const SynthConfig& cfg, int k, msg::StereoFrame& out);

}  // namespace ingvio
