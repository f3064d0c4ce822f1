#include "IngvioFilter.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "StateManager.h"

namespace ingvio {

IngvioFilter::IngvioFilter(const IngvioParams& params, std::shared_ptr<Triangulator> tri) : _filter_params(params)
{
    _state = std::make_shared<State>(_filter_params);                                   // IngvioFilter.cpp:76-98
    _imu_propa = std::make_shared<ImuPropagator>(_filter_params);
    _tri = tri ? tri : std::make_shared<Triangulator>();
    _map_server = std::make_shared<MapServer>();
    _remove_lost_update = std::make_shared<RemoveLostUpdate>(_filter_params);
    _sw_marg_update = std::make_shared<SwMargUpdate>(_filter_params);
    _keyframe_update = std::make_shared<KeyframeUpdate>(_filter_params);
    _landmark_update = std::make_shared<LandmarkUpdate>(_filter_params);                // :90
    _gnss_update = std::make_shared<GnssUpdate>(_filter_params);                        // :92-96
    _gnss_sync = std::make_shared<GnssSync>(_filter_params);                            // GnssSync.h:60-74
    _aligner = std::make_shared<GvioAligner>(StateManager::ctx(_state), _filter_params._gv_align_batch_size, _filter_params._gv_align_max_iter,
                                             _filter_params._gv_align_conv_epsilon, _filter_params._gv_align_vel_thres);      // :95
}

// The GNSS block at the end of both camera callbacks (IngvioFilter.cpp:329-362, :200-233).
void IngvioFilter::gnssBlock(double stamp)
{
    _last_gnss_rows = 0;
    _gnss_update->clearLastKeep();
    if (!_filter_params._enable_gnss) return;
    GnssMeas gnss_meas;
    SppMeas spp_meas;
    const bool flag = _gnss_sync->getSppAt(stamp, spp_meas);                            // :336-340
    if (_gnss_sync->getGnssMeasAt(stamp, gnss_meas)) {
        if (flag && !_gvio_aligner.isAlign() && !gnss_meas.raw_obs.empty()) {          // :344-345
            RawGnssEpoch raw;
            raw.eph = gnss_meas.raw_eph; raw.obs = gnss_meas.raw_obs; raw.doy = gnss_meas.doy;
            _aligner->batchAlign(raw, _state->_extended_pose, gnss_meas.iono);
            if (_aligner->isAlign()) _gvio_aligner = _aligner->alignment();
        }
        if (_gvio_aligner.isAlign()) {
            _gnss_update->checkYofStatus(_state, _gvio_aligner);                        // :349
            _last_gnss_rows = _gnss_update->updateTrackedSys(_state, gnss_meas, _gvio_aligner);      // :353-354
            if (flag) _gnss_vars_added += _gnss_update->addNewTrackedSys(_state, gnss_meas, spp_meas, _gvio_aligner);   // :356-359
        }
    }
}

void IngvioFilter::callbackIMU(const ImuMsg& m)
{
    if (_filter_params._enable_gnss && !_gnss_sync->isSync()) return;                   // :385-391 (storeTimePair itself is the ROS node's)
    if (!_hasImageCome) return;                                                         // :393: IMU samples before the first image are dropped
    _imu_propa->storeImu(ImuCtrl(m.stamp, Vec3d(m.accel), Vec3d(m.gyro)));
    if (!_hasInitState && _imu_propa->isInit()) {                                       // :396-406
        _state->initStateAndCov(m.stamp, _imu_propa->getInitQuat());
        _hasInitState = true;
    }
}

// MapServerManager::collectStereoMeas over FeatureInfoManager::collectStereoMeas (MapServerManager.cpp:203-217 over :146-185)
void IngvioFilter::collectStereoMeas(const StereoFrameMsg& f)
{
    const double timestamp = _state->_timestamp;
    const auto pose_it = _state->_sw_camleft_poses.find(timestamp);
    if (pose_it == _state->_sw_camleft_poses.end()) {
        std::printf("[FeatureInfoManager]: Meas timestamp not in sw!\n");                 // :152-156 (assert(false) there)
        std::exit(EXIT_FAILURE);
    }
    for (const auto& o : f.stereo_meas) {
        auto it = _map_server->find(o.id);
        if (it == _map_server->end()) it = _map_server->insert({ o.id, std::make_shared<FeatureInfo>() }).first;      // :210-211
        const auto& fi = it->second;
        auto sm = std::make_shared<StereoMeas>();
        sm->_id = o.id; sm->_u0 = o.u0; sm->_v0 = o.v0; sm->_u1 = o.u1; sm->_v1 = o.v1;
        if (fi->_stereo_obs.size() == 0) {                                              // :158-168: first observation
            fi->_stereo_obs[timestamp] = sm;
            fi->_id = o.id;
            fi->_ftype = FeatureInfo::MSCKF;
            fi->_isToMarg = false;
            fi->_isTri = false;
            fi->_landmark->resetAnchoredPose(pose_it->second);
        } else {
            if (fi->hasStereoObsAt(timestamp)) {                                        // :171-175
                std::printf("[FeatureInfoManager]: Meas timestamp already in stereo obs, skip adding! \n");
                continue;
            }
            if (fi->_ftype == FeatureInfo::SLAM) fi->_stereo_obs.clear();               // :177-178
            fi->_stereo_obs[timestamp] = sm;
            fi->_isToMarg = false;                                                      // :184
        }
    }
}

// MapServerManager::collectMonoMeas over FeatureInfoManager::collectMonoMeas (MapServerManager.cpp:187-201 over :105-144)
void IngvioFilter::collectMonoMeas(const MonoFrameMsg& f)
{
    const double timestamp = _state->_timestamp;
    const auto pose_it = _state->_sw_camleft_poses.find(timestamp);
    if (pose_it == _state->_sw_camleft_poses.end()) {
        std::printf("[FeatureInfoManager]: Meas timestamp not in sw!\n");
        std::exit(EXIT_FAILURE);
    }
    for (const auto& o : f.mono_meas) {
        auto it = _map_server->find(o.id);
        if (it == _map_server->end()) it = _map_server->insert({ o.id, std::make_shared<FeatureInfo>() }).first;
        const auto& fi = it->second;
        auto mm = std::make_shared<MonoMeas>();
        mm->_id = o.id; mm->_u0 = o.u0; mm->_v0 = o.v0;
        if (fi->_mono_obs.size() == 0) {
            fi->_mono_obs[timestamp] = mm;
            fi->_id = o.id;
            fi->_ftype = FeatureInfo::MSCKF;
            fi->_isToMarg = false;
            fi->_isTri = false;
            fi->_landmark->resetAnchoredPose(pose_it->second);
        } else {
            if (fi->hasMonoObsAt(timestamp)) {
                std::printf("[FeatureInfoManager]: Meas timestamp already in mono obs, skip adding! \n");
                continue;
            }
            if (fi->_ftype == FeatureInfo::SLAM) fi->_mono_obs.clear();
            fi->_mono_obs[timestamp] = mm;
            fi->_isToMarg = false;
        }
    }
}

void IngvioFilter::callbackStereoFrame(const StereoFrameMsg& frame)
{
    if (_filter_params._enable_gnss && !_gnss_sync->isSync()) return;                   // :254-255
    if (!_hasImageCome) { _hasImageCome = true; return; }                               // :257-261
    if (!_hasInitState) return;
    const double target_time = frame.stamp;
    if (_state->_timestamp >= target_time) return;                                      // :267
    // INGVIO_SHIM_TIMING=1: host wall time of the callback's phases on stderr (where the host waits for the device shows up in the
    // phase that fetches a result); debugging aid of the single-filter latency figure
    static const bool timing = std::getenv("INGVIO_SHIM_TIMING") != nullptr;
    using clk = std::chrono::steady_clock;
    clk::time_point t0, t1, t2, t3, t4;
    if (timing) t0 = clk::now();
    _imu_propa->propagateAugmentAtEnd(_state, target_time);
    if (_state->_timestamp < target_time) return;                                       // :273
    if (timing) t1 = clk::now();
    collectStereoMeas(frame);
    if (timing) t2 = clk::now();
    _remove_lost_update->updateStateStereo(_state, _map_server, _tri);
    if (timing) t3 = clk::now();
    struct Report {
        const bool on; const clk::time_point &a, &b, &c, &d; const int frame;
        ~Report() {
            if (!on) return;
            auto us = [](clk::time_point x, clk::time_point y) { return std::chrono::duration<double, std::micro>(y - x).count(); };
            std::fprintf(stderr, "SHIM frame %d us: propagate+clone %.1f collect %.1f remove_lost %.1f rest %.1f\n", frame, us(a, b), us(b, c), us(c, d),
                         us(d, clk::now()));
        }
    } report{ timing, t0, t1, t2, t3, _frames };
    if (_filter_params._is_key_frame) {
        clk::time_point k0, k1, k2, k3;
        if (timing) k0 = clk::now();
        _keyframe_update->updateStateStereo(_state, _map_server, _tri);
        if (timing) k1 = clk::now();
        if (_filter_params._max_lm_feats > 0) {                                         // :283-289
            _landmark_update->updateLandmarkStereo(_state, _map_server);
            _landmark_update->initNewLandmarkStereo(_state, _map_server, _tri, _filter_params._max_sw_clones);
        }
        _keyframe_update->cleanStereoObsAtMargTime(_state, _map_server);
        if (timing) k2 = clk::now();
        _keyframe_update->changeMSCKFAnchor(_state, _map_server);
        if (timing) k3 = clk::now();
        if (_filter_params._max_lm_feats > 0) {                                         // :295-301
            std::vector<double> marg_kfs;
            _keyframe_update->getMargKfs(_state, marg_kfs);
            _landmark_update->changeLandmarkAnchor(_state, _map_server, marg_kfs);
        }
        _keyframe_update->margSwPose(_state);
        if (timing) {
            auto us = [](clk::time_point x, clk::time_point y) { return std::chrono::duration<double, std::micro>(y - x).count(); };
            std::fprintf(stderr, "SHIM key frame us: update %.1f clean %.1f anchor change %.1f marginalise %.1f\n", us(k0, k1), us(k1, k2), us(k2, k3), us(k3, clk::now()));
        }
    } else {
        _sw_marg_update->updateStateStereo(_state, _map_server, _tri);
        if (_filter_params._max_lm_feats > 0) {                                         // :309-315
            _landmark_update->updateLandmarkStereo(_state, _map_server);
            _landmark_update->initNewLandmarkStereo(_state, _map_server, _tri, _filter_params._max_sw_clones);
        }
        _sw_marg_update->cleanStereoObsAtMargTime(_state, _map_server);
        _sw_marg_update->changeMSCKFAnchor(_state, _map_server);
        if (_filter_params._max_lm_feats > 0) _landmark_update->changeLandmarkAnchor(_state, _map_server);      // :321-322
        _sw_marg_update->margSwPose(_state);
    }
    eraseInvalidFeatures(_map_server, _state, &_last_invalid_erased);
    gnssBlock(frame.stamp);
    ++_frames;
}

void IngvioFilter::callbackMonoFrame(const MonoFrameMsg& frame)
{
    if (_filter_params._enable_gnss && !_gnss_sync->isSync()) return;                   // :126-127
    if (!_hasImageCome) { _hasImageCome = true; return; }
    if (!_hasInitState) return;
    const double target_time = frame.stamp;
    if (_state->_timestamp >= target_time) return;
    _imu_propa->propagateAugmentAtEnd(_state, target_time);
    if (_state->_timestamp < target_time) return;
    collectMonoMeas(frame);
    _remove_lost_update->updateStateMono(_state, _map_server, _tri);
    if (_filter_params._is_key_frame) {
        _keyframe_update->updateStateMono(_state, _map_server, _tri);
        if (_filter_params._max_lm_feats > 0) {                                         // :155-161
            _landmark_update->updateLandmarkMono(_state, _map_server);
            _landmark_update->initNewLandmarkMono(_state, _map_server, _tri, _filter_params._max_sw_clones);
        }
        _keyframe_update->cleanMonoObsAtMargTime(_state, _map_server);
        _keyframe_update->changeMSCKFAnchor(_state, _map_server);
        if (_filter_params._max_lm_feats > 0) {                                         // :167-173
            std::vector<double> marg_kfs;
            _keyframe_update->getMargKfs(_state, marg_kfs);
            _landmark_update->changeLandmarkAnchor(_state, _map_server, marg_kfs);
        }
        _keyframe_update->margSwPose(_state);
    } else {
        _sw_marg_update->updateStateMono(_state, _map_server, _tri);
        if (_filter_params._max_lm_feats > 0) {                                         // :181-187
            _landmark_update->updateLandmarkMono(_state, _map_server);
            _landmark_update->initNewLandmarkMono(_state, _map_server, _tri, _filter_params._max_sw_clones);
        }
        _sw_marg_update->cleanMonoObsAtMargTime(_state, _map_server);
        _sw_marg_update->changeMSCKFAnchor(_state, _map_server);
        if (_filter_params._max_lm_feats > 0) _landmark_update->changeLandmarkAnchor(_state, _map_server);      // :193-194
        _sw_marg_update->margSwPose(_state);
    }
    eraseInvalidFeatures(_map_server, _state, &_last_invalid_erased);
    gnssBlock(frame.stamp);
    ++_frames;
}

// ---- the ROS-shaped surface (Messages.h) ------------------------------------------------------------------------------
void IngvioFilter::callbackIMU(const msg::Imu& m)
{
    ImuMsg i; i.stamp = m.header.stamp.toSec();
    i.gyro[0] = m.angular_velocity.x; i.gyro[1] = m.angular_velocity.y; i.gyro[2] = m.angular_velocity.z;
    i.accel[0] = m.linear_acceleration.x; i.accel[1] = m.linear_acceleration.y; i.accel[2] = m.linear_acceleration.z;
    callbackIMU(i);
}

void IngvioFilter::callbackStereoFrame(const msg::StereoFrame& f)
{
    StereoFrameMsg s; s.stamp = f.header.stamp.toSec();
    s.stereo_meas.reserve(f.stereo_features.size());
    for (const auto& o : f.stereo_features) { StereoObsMsg q; q.id = (int)o.id; q.u0 = o.u0; q.v0 = o.v0; q.u1 = o.u1; q.v1 = o.v1; s.stereo_meas.push_back(q); }
    callbackStereoFrame(s);
}

void IngvioFilter::callbackMonoFrame(const msg::MonoFrame& f)
{
    MonoFrameMsg s; s.stamp = f.header.stamp.toSec();
    s.mono_meas.reserve(f.mono_features.size());
    for (const auto& o : f.mono_features) { MonoObsMsg q; q.id = (int)o.id; q.u0 = o.u0; q.v0 = o.v0; s.mono_meas.push_back(q); }
    callbackMonoFrame(s);
}

bool IngvioFilter::odometry(const msg::Header& header, msg::Odometry& od) const
{
    const Mat3d R = _state->_extended_pose->valueLinearAsMat();
    const Vec3d p = _state->_extended_pose->valueTrans1(), v = _state->_extended_pose->valueTrans2();
    for (int i = 0; i < 9; ++i) if (R.m[i] != R.m[i]) return false;                      // hasNaN, :418-419
    for (int i = 0; i < 3; ++i) if (p[i] != p[i] || v[i] != v[i]) return false;
    od.header.stamp = header.stamp; od.header.seq = header.seq; od.header.frame_id = "world"; od.child_frame_id = "uav";      // :425-427
    od.position.x = p[0]; od.position.y = p[1]; od.position.z = p[2];
    const Quatd q = _state->_extended_pose->valueLinearAsQuat();
    od.orientation.x = q.x; od.orientation.y = q.y; od.orientation.z = q.z; od.orientation.w = q.w;
    od.linear_velocity.x = v[0]; od.linear_velocity.y = v[1]; od.linear_velocity.z = v[2];
    return true;
}

}  // namespace ingvio
