// ingvio_replay — plays an INGVIOR1 recording (ingvio_amd/csrc/host/Replay.h) into the callback surface of the filter and prints
// the odometry IngvioFilter::visualize would publish, one line per processed camera frame:
//     stamp px py pz qx qy qz qw vx vy vz N clones
// usage: ingvio_replay <file> [--dump] [--set "key: value" ...]
//        ingvio_replay --synth "feats=150,clones=11,life=10,cohort=1,frames=60,..." [--write <file> | --frame <k> | --time] [--set ...]
//   --dump   parse and count the records only (no device needed)
//   --synth  a synthetic stream (ingvio_amd/csrc/host/SynthStream.h; keys = the fields of SynthConfig) instead of a file:
//            --write  store it as an INGVIOR1 file (no device needed)
//            --frame  print the feature message of frame k regenerated from the seed alone: "FEAT id u0 v0 u1 v1" (no device needed)
//            --trace  play it into a filter and print, for every processed camera frame, what the policy layer decided and the state
//                     afterwards ("TRACE k stamp" ... "END"): the comparison side of tests/golden/stream_*.npz (tests/test_stream_golden.py)
//            --time   play it into a filter and print the wall time of every camera callback: "FRAME k ms lost_rows lost_accepted
//                     select_rows N clones", then "LATENCY ..." (single-filter latency)
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sstream>
#include <set>
#include <string>

#include "../ingvio_amd/csrc/host/Replay.h"
#include "../ingvio_amd/csrc/host/StateManager.h"

static bool parseSynth(const std::string& spec, ingvio::SynthConfig& c)
{
    std::istringstream in(spec);
    std::string kv;
    while (std::getline(in, kv, ',')) {
        const size_t eq = kv.find('=');
        if (eq == std::string::npos) { if (kv.empty()) continue; return false; }
        const std::string k = kv.substr(0, eq), v = kv.substr(eq + 1);
        if (k == "feats") c.feats = std::atoi(v.c_str());
        else if (k == "clones") c.clones = std::atoi(v.c_str());
        else if (k == "stereo") c.stereo = std::atoi(v.c_str());
        else if (k == "key") c.is_key_frame = std::atoi(v.c_str());
        else if (k == "life") c.life = std::atoi(v.c_str());
        else if (k == "cohort") c.cohort = std::atoi(v.c_str());
        else if (k == "birth_frame") c.birth_frame = std::atoi(v.c_str());
        else if (k == "outlier_every") c.outlier_every = std::atoi(v.c_str());
        else if (k == "frames") c.frames = std::atoi(v.c_str());
        else if (k == "pixel_noise") c.pixel_noise = std::atof(v.c_str());
        else if (k == "visual_noise") c.visual_noise = std::atof(v.c_str());
        else if (k == "seed") c.seed = std::strtoull(v.c_str(), nullptr, 0);
        else if (k == "gnss") c.enable_gnss = std::atoi(v.c_str());
        else return false;
    }
    return c.feats > 0 && c.clones >= 3 && c.life >= 1 && c.frames >= 1;
}

static void printInts(const char* tag, const std::vector<int>& v)
{
    std::printf("%s", tag);
    for (int x : v) std::printf(" %d", x);
    std::printf("\n");
}
static void printDoubles(const char* tag, const std::vector<double>& v)
{
    std::printf("%s", tag);
    for (double x : v) std::printf(" %.17g", x);
    std::printf("\n");
}

// One processed frame as the golden streams record it (oracle/stream_filter.py: Filter.callback_stereo's trace dict).
static void printTrace(int k, ingvio::IngvioFilter& f, bool keyframe)
{
    using namespace ingvio;
    auto state = f.state();
    std::printf("TRACE %d %.17g\n", k, state->_timestamp);
    const UpdateRecord& rl = f.removeLostUpdate()->lastRecord();
    printInts("LOST_IDS", rl.ids); printInts("LOST_ACC", rl.accepted); printInts("LOST_DIRECT", rl.direct);
    std::printf("LOST_ROWS %d\n", rl.rows);
    const UpdateRecord& se = keyframe ? f.keyframeUpdate()->lastRecord() : f.swMargUpdate()->lastRecord();
    const MaintenanceRecord& mt = keyframe ? f.keyframeUpdate()->maintenance() : f.swMargUpdate()->maintenance();
    printDoubles("SEL_STAMPS", se.stamps); printInts("SEL_IDS", se.ids); printInts("SEL_ACC", se.accepted);
    std::printf("SEL_ROWS %d\n", se.rows);
    printDoubles("MARG_STAMPS", mt.marg_stamps); printInts("CLEAN_ERASED", mt.clean_erased); printInts("ANCHOR_ERASED", mt.anchor_erased);
    printInts("ANCHOR_MOVED", mt.anchor_moved); printInts("INVALID_ERASED", f.lastInvalidErased());
    // the GNSS block of the callback (IngvioFilter.cpp:329-362): rows handed to ekfUpdate, which candidate rows passed their gate, variables
    // added so far (delayed initialisation), the GNSS scalars GPS GLO GAL BDS FS YOF (nan: not in the state)
    std::printf("GNSS_ROWS %d\nGNSS_ADDED_TOTAL %d\n", f.lastGnssRows(), f.gnssVarsAdded());
    printInts("GNSS_KEEP", f.gnssUpdate()->lastKeep());
    {
        std::vector<double> gv;
        for (int t = 0; t < 6; ++t) { auto it = state->_gnss.find(t); gv.push_back(it != state->_gnss.end() ? it->second->value() : std::nan("")); }
        printDoubles("GNSS_VALS", gv);
    }
    // in-state SLAM landmarks (max_landmark_features > 0): what the update evaluated and accepted, what the delayed initialisation added, the
    // landmarks in the state after the frame (ascending id) with their world positions, and what left the state during the frame
    {
        static std::set<int> lm_prev;
        const auto& lu = *f.landmarkUpdate();
        printInts("LM_UPD_IDS", lu.lastUpdateIds()); printInts("LM_UPD_ACC", lu.lastUpdateAccepted()); printInts("LM_INIT_IDS", lu.lastInitIds());
        std::vector<int> ids;
        for (const auto& it : state->_anchored_landmarks) ids.push_back(it.first);
        std::sort(ids.begin(), ids.end());
        std::vector<double> vals;
        for (int id : ids) { const Vec3d p = state->_anchored_landmarks.at(id)->valuePosXyz(); vals.insert(vals.end(), p.v, p.v + 3); }
        printInts("LM_IDS", ids); printDoubles("LM_VALS", vals);
        std::set<int> had = lm_prev;
        for (int id : lu.lastInitIds()) had.insert(id);
        std::vector<int> gone;
        for (int id : had) if (!std::binary_search(ids.begin(), ids.end(), id)) gone.push_back(id);
        printInts("LM_MARG_IDS", gone);
        lm_prev = std::set<int>(ids.begin(), ids.end());
    }
    // the world <-> ENU / ECEF alignment the GNSS updates use: given with the recording, or found by GvioAligner::batchAlign on the raw epochs
    {
        const GvioAlignment& al = f.gnssAlignment();
        std::vector<double> a = { al.aligned ? 1.0 : 0.0, al.yaw_offset, al.anchor_ecef[0], al.anchor_ecef[1], al.anchor_ecef[2] };
        printDoubles("ALIGN", a);
    }
    std::vector<int> table;
    for (const auto& v : StateManager::errVariables(state)) { table.push_back(v->idx()); table.push_back(v->size()); }
    printInts("TABLE", table);
    std::vector<double> sw;
    for (const auto& c : state->_sw_camleft_poses) sw.push_back(c.first);
    printDoubles("SW_STAMPS", sw);
    std::vector<int> ids;
    for (const auto& m : *f.mapServer()) ids.push_back(m.first);
    printInts("MAP_IDS", ids);
    std::vector<double> pose;
    const Mat3d R = state->_extended_pose->valueLinearAsMat();
    pose.insert(pose.end(), R.m, R.m + 9);
    for (const Vec3d& v : { state->_extended_pose->valueTrans1(), state->_extended_pose->valueTrans2(), state->_bg->value(), state->_ba->value() })
        pose.insert(pose.end(), v.v, v.v + 3);
    const Mat3d Re = state->_camleft_imu_extrinsics->valueLinearAsMat();
    pose.insert(pose.end(), Re.m, Re.m + 9);
    const Vec3d pe = state->_camleft_imu_extrinsics->valueTrans();
    pose.insert(pose.end(), pe.v, pe.v + 3);
    printDoubles("POSE", pose);
    const MatXd P = StateManager::getFullCov(state);
    std::vector<double> diag;
    double fro = 0;
    for (int j = 0; j < P.cols(); ++j) { diag.push_back(P(j, j)); for (int i = 0; i < P.rows(); ++i) fro += P(i, j) * P(i, j); }
    printDoubles("DIAG", diag);
    std::printf("NORM %.17g\nEND\n", std::sqrt(fro));
}

static double median(std::vector<double> v)
{
    if (v.empty()) return 0.0;
    std::sort(v.begin(), v.end());
    return v.size() % 2 ? v[v.size() / 2] : 0.5 * (v[v.size() / 2 - 1] + v[v.size() / 2]);
}

int main(int argc, char** argv)
{
    if (argc < 2) { std::fprintf(stderr, "usage: ingvio_replay <file> [--dump] [--set \"key: value\" ...] | --synth <spec> [--write f | --frame k | --time]\n"); return 2; }
    bool dump = false, timed = false, trace = false;
    std::string overrides, synth, write_path, file;
    int frame_k = -1;
    for (int i = 1; i < argc; ++i) {
        if (!std::strcmp(argv[i], "--dump")) dump = true;
        else if (!std::strcmp(argv[i], "--time")) timed = true;
        else if (!std::strcmp(argv[i], "--trace")) trace = true;
        else if (!std::strcmp(argv[i], "--set") && i + 1 < argc) { overrides += argv[++i]; overrides += "\n"; }
        else if (!std::strcmp(argv[i], "--synth") && i + 1 < argc) synth = argv[++i];
        else if (!std::strcmp(argv[i], "--write") && i + 1 < argc) write_path = argv[++i];
        else if (!std::strcmp(argv[i], "--frame") && i + 1 < argc) frame_k = std::atoi(argv[++i]);
        else if (argv[i][0] != '-' && file.empty()) file = argv[i];
        else { std::fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
    }
    if (!synth.empty() || timed || frame_k >= 0 || !write_path.empty()) {
        ingvio::SynthConfig cfg;
        if (!parseSynth(synth, cfg)) { std::fprintf(stderr, "bad --synth spec\n"); return 2; }
        if (frame_k >= 0) {
            ingvio::msg::StereoFrame f;
            ingvio::synthFrame(cfg, frame_k, f);
            std::printf("STAMP %llu\n", (unsigned long long)f.header.stamp.toNSec());
            for (const auto& m : f.stereo_features) std::printf("FEAT %llu %.17g %.17g %.17g %.17g\n", (unsigned long long)m.id, m.u0, m.v0, m.u1, m.v1);
            return 0;
        }
        if (!write_path.empty()) {
            if (!ingvio::writeSynthRecording(cfg, write_path)) { std::fprintf(stderr, "cannot write %s\n", write_path.c_str()); return 1; }
            return 0;
        }
        std::vector<ingvio::FrameTiming> tm;
        double terr = 0;
        std::string err;
        std::function<void(int, ingvio::IngvioFilter&)> on_frame;
        const bool keyframe = cfg.is_key_frame != 0;
        if (trace) on_frame = [keyframe](int k, ingvio::IngvioFilter& f) { printTrace(k, f, keyframe); };
        if (!ingvio::playSynth(cfg, overrides, tm, &terr, err, on_frame)) { std::fprintf(stderr, "synthetic play failed: %s\n", err.c_str()); return 1; }
        std::vector<double> heavy, light, all;
        int heavy_acc = 0, heavy_rows = 0;
        for (const auto& t : tm) {
            std::printf("FRAME %d %.4f %d %d %d %d %d\n", t.k, t.ms, t.lost_rows, t.lost_accepted, t.select_rows, t.n, t.clones);
            // warm-up: the first frames allocate workspaces and build the window; heavy = a RemoveLost update over at least half the tracks
            if (t.k <= 2 * cfg.life + 2) continue;
            all.push_back(t.ms);
            if (t.lost_accepted * 2 >= cfg.feats || (t.lost_rows > 0 && cfg.cohort)) { heavy.push_back(t.ms); heavy_acc = std::max(heavy_acc, t.lost_accepted); heavy_rows = std::max(heavy_rows, t.lost_rows); }
            else light.push_back(t.ms);
        }
        std::printf("LATENCY frames=%zu timed=%zu median_ms=%.4f heavy_frames=%zu heavy_median_ms=%.4f heavy_min_ms=%.4f heavy_accepted=%d heavy_rows=%d "
                    "other_median_ms=%.4f final_pos_err_m=%.4f\n", tm.size(), all.size(), median(all), heavy.size(), median(heavy),
                    heavy.empty() ? 0.0 : *std::min_element(heavy.begin(), heavy.end()), heavy_acc, heavy_rows, median(light), terr);
        return 0;
    }
    ingvio::ReplayStats st;
    std::string err;
    int trace_k = 0;
    const bool ok = ingvio::replayFile(file, overrides, dump,
        [&](const ingvio::msg::Odometry& od, const ingvio::IngvioFilter& f) {
            auto& flt = const_cast<ingvio::IngvioFilter&>(f);
            if (trace) { printTrace(++trace_k, flt, flt.params()._is_key_frame != 0); return; }      // ingvio_replay <file> --trace: as --synth ... --trace
            std::printf("ODOM %.9f %.9f %.9f %.9f %.9f %.9f %.9f %.9f %.6f %.6f %.6f %d %zu\n", od.header.stamp.toSec(), od.position.x, od.position.y,
                        od.position.z, od.orientation.x, od.orientation.y, od.orientation.z, od.orientation.w, od.linear_velocity.x,
                        od.linear_velocity.y, od.linear_velocity.z, flt.state()->curr_cov_size(), flt.state()->_sw_camleft_poses.size());
        }, st, err);
    std::printf("RECORDS params=%llu imu=%llu mono=%llu stereo=%llu gnss=%llu spp=%llu align=%llu truth=%llu features=%llu frames_processed=%d span=%.3f\n",
                (unsigned long long)st.counts[0], (unsigned long long)st.counts[1], (unsigned long long)st.counts[2], (unsigned long long)st.counts[3],
                (unsigned long long)st.counts[4], (unsigned long long)st.counts[5], (unsigned long long)st.counts[6], (unsigned long long)st.counts[7],
                (unsigned long long)st.features, st.frames_processed, st.t_last - st.t_first);
    if (!ok) { std::fprintf(stderr, "replay failed: %s\n", err.c_str()); return 1; }
    return 0;
}
