// IngvioFilter.h — ROS-free mirror of the callback surface of ingvio_estimator/src/IngvioFilter.{h,cpp}
// (IngvioFilter.cpp:252-407): callbackIMU / callbackStereoFrame / callbackMonoFrame with POD
// messages shaped like sensor_msgs/Imu and feature_tracker/{Mono,Stereo}Frame, same early-return logic
// and the same per-frame orchestration (propagate+clone -> collect -> RemoveLost -> Keyframe|SwMarg ->
// clean / re-anchor / marginalise -> erase invalid -> GNSS block (:329-362: pick the epoch of this frame, checkYofStatus,
// updateTrackedSys, addNewTrackedSys)).  callbackGnssMeas / callbackSppMeas buffer the epochs as GnssProcessor does
// (GnssProcessor.cpp:119-220).  ROS, tf, publishers and the RINEX / ephemeris handling are out of scope.  GvioAligner::batchAlign
// runs when the buffered epochs carry their raw content (GnssMeas::raw_*); otherwise the alignment is set with setGnssAlignment().
#pragma once
#include <memory>
#include <vector>

#include "GnssUpdate.h"
#include "GvioAligner.h"
#include "ImuPropagator.h"
#include "IngvioParams.h"
#include "MapServer.h"
#include "Messages.h"
#include "LandmarkUpdate.h"
#include "MsckfUpdates.h"
#include "State.h"

namespace ingvio {

struct ImuMsg { double stamp; double gyro[3]; double accel[3]; };                       // sensor_msgs/Imu subset
struct StereoObsMsg { int id; double u0, v0, u1, v1; };                                 // feature_tracker/StereoMeas
struct StereoFrameMsg { double stamp; std::vector<StereoObsMsg> stereo_meas; };         // feature_tracker/StereoFrame
struct MonoObsMsg { int id; double u0, v0; };
struct MonoFrameMsg { double stamp; std::vector<MonoObsMsg> mono_meas; };

class IngvioFilter {
public:
    IngvioFilter(const IngvioParams& params, std::shared_ptr<Triangulator> tri = nullptr);
    void callbackIMU(const ImuMsg& imu_msg);                                            // IngvioFilter.cpp:381-407
    void callbackStereoFrame(const StereoFrameMsg& stereo_frame);                       // :252-379
    void callbackMonoFrame(const MonoFrameMsg& mono_frame);                             // :124-250
    // the same callbacks on structs shaped like the ROS messages (Messages.h): what a ROS1 wrapper node or the replay driver feeds
    void callbackIMU(const msg::Imu& imu_msg);
    void callbackStereoFrame(const msg::StereoFrame& stereo_frame);
    void callbackMonoFrame(const msg::MonoFrame& mono_frame);
    // IngvioFilter::visualize (:409-447): the nav_msgs/Odometry it publishes; false when the state holds NaN (:418-419)
    bool odometry(const msg::Header& header, msg::Odometry& out) const;
    void callbackGnssMeas(const GnssMeas& gnss_meas) { _gnss_sync->bufferGnssMeas(gnss_meas); }       // GnssProcessor.cpp:119-220 -> GnssSync
    void callbackSppMeas(const SppMeas& spp_meas) { _gnss_sync->bufferSppMeas(spp_meas); }
    void setGnssAlignment(const GvioAlignment& a) { _gvio_aligner = a; }
    const GvioAlignment& gnssAlignment() const { return _gvio_aligner; }               // given (setGnssAlignment) or found by batchAlign
    const IngvioParams& params() const { return _filter_params; }
    std::shared_ptr<GvioAligner> gvioAligner() { return _aligner; }                     // batchAlign on the raw epochs (IngvioFilter.cpp:344-345)
    std::shared_ptr<GnssSync> gnssSync() { return _gnss_sync; }
    std::shared_ptr<GnssUpdate> gnssUpdate() { return _gnss_update; }
    int lastGnssRows() const { return _last_gnss_rows; }
    int gnssVarsAdded() const { return _gnss_vars_added; }

    std::shared_ptr<State> state() { return _state; }
    std::shared_ptr<MapServer> mapServer() { return _map_server; }
    std::shared_ptr<ImuPropagator> imuPropagator() { return _imu_propa; }
    std::shared_ptr<LandmarkUpdate> landmarkUpdate() { return _landmark_update; }
    int framesProcessed() const { return _frames; }
    const std::vector<int>& lastInvalidErased() const { return _last_invalid_erased; }      // eraseInvalidFeatures of the last frame (trace)
    std::shared_ptr<RemoveLostUpdate> removeLostUpdate() { return _remove_lost_update; }
    std::shared_ptr<KeyframeUpdate> keyframeUpdate() { return _keyframe_update; }
    std::shared_ptr<SwMargUpdate> swMargUpdate() { return _sw_marg_update; }

protected:
    void collectStereoMeas(const StereoFrameMsg& f);                                    // MapServerManager.cpp:147-217
    void collectMonoMeas(const MonoFrameMsg& f);
    void gnssBlock(double stamp);                                                       // IngvioFilter.cpp:329-362 / :200-233
    IngvioParams _filter_params;
    std::shared_ptr<State> _state;
    std::shared_ptr<ImuPropagator> _imu_propa;
    std::shared_ptr<Triangulator> _tri;
    std::shared_ptr<MapServer> _map_server;
    std::shared_ptr<RemoveLostUpdate> _remove_lost_update;
    std::shared_ptr<SwMargUpdate> _sw_marg_update;
    std::shared_ptr<KeyframeUpdate> _keyframe_update;
    std::shared_ptr<LandmarkUpdate> _landmark_update;
    std::shared_ptr<GnssUpdate> _gnss_update;
    std::shared_ptr<GnssSync> _gnss_sync;
    GvioAlignment _gvio_aligner;
    std::shared_ptr<GvioAligner> _aligner;
    int _last_gnss_rows = 0, _gnss_vars_added = 0;
    bool _hasImageCome = false, _hasInitState = false;
    int _frames = 0;
    std::vector<int> _last_invalid_erased;
};

}  // namespace ingvio
