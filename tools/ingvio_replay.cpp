// ingvio_replay — plays an INGVIOR1 recording (ingvio_amd/csrc/host/Replay.h) into the callback surface of the filter and prints
// the odometry IngvioFilter::visualize would publish, one line per processed camera frame:
//     stamp px py pz qx qy qz qw vx vy vz N clones
// usage: ingvio_replay <file> [--dump] [--set "key: value" ...]
//   --dump  parse and count the records only (no device needed)
#include <cstdio>
#include <cstring>
#include <string>

#include "../ingvio_amd/csrc/host/Replay.h"

int main(int argc, char** argv)
{
    if (argc < 2) { std::fprintf(stderr, "usage: ingvio_replay <file> [--dump] [--set \"key: value\" ...]\n"); return 2; }
    bool dump = false;
    std::string overrides;
    for (int i = 2; i < argc; ++i) {
        if (!std::strcmp(argv[i], "--dump")) dump = true;
        else if (!std::strcmp(argv[i], "--set") && i + 1 < argc) { overrides += argv[++i]; overrides += "\n"; }
    }
    ingvio::ReplayStats st;
    std::string err;
    const bool ok = ingvio::replayFile(argv[1], overrides, dump,
        [](const ingvio::msg::Odometry& od, const ingvio::IngvioFilter& f) {
            auto& flt = const_cast<ingvio::IngvioFilter&>(f);
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
