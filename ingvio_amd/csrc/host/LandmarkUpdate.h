// LandmarkUpdate.h — SURVEY.md §8(f) row f-2: mirrors ingvio_estimator/src/LandmarkUpdate.h:35-125 (update of the
// in-state SLAM landmarks, delayed initialisation of new ones, anchor change before a clone is marginalised) and
// FeatureInfoManager::changeAnchoredPose (MapServerManager.cpp:343-385).  The 2x24 / 4x24 measurement rows are built on
// the host as in the reference; every covariance operation (whitenResidual gate, stacked ekfUpdate, addVariableDelayed,
// replaceVarLinear, marginalize) is a call into libingvio_hip.so through StateManager.
#pragma once
#include <map>
#include <memory>
#include <vector>

#include "IngvioParams.h"
#include "MapServer.h"
#include "MsckfUpdates.h"
#include "Update.h"

namespace ingvio {

class State;

class FeatureInfoManager {
public:
    static void changeAnchoredPose(std::shared_ptr<FeatureInfo> feature_info, std::shared_ptr<State> state, double target_sw_timestamp);
    static void changeAnchoredPose(std::shared_ptr<FeatureInfo> feature_info, std::shared_ptr<State> state);
};

class LandmarkUpdate : public UpdateBase {
public:
    LandmarkUpdate(const IngvioParams& filter_params)
        : UpdateBase(filter_params._chi2_max_dof, filter_params._chi2_thres), _noise(filter_params._visual_noise) {}
    virtual ~LandmarkUpdate() {}

    void updateLandmarkMono(std::shared_ptr<State> state, std::shared_ptr<MapServer> map_server);          // LandmarkUpdate.cpp:32-149
    void updateLandmarkMonoSw(std::shared_ptr<State> state, std::shared_ptr<MapServer> map_server);        // :151-271
    void updateLandmarkStereo(std::shared_ptr<State> state, std::shared_ptr<MapServer> map_server);        // :688-806
    void initNewLandmarkMono(std::shared_ptr<State> state, std::shared_ptr<MapServer> map_server, std::shared_ptr<Triangulator> tri,
                             int min_init_poses);                                                          // :363-424
    void initNewLandmarkStereo(std::shared_ptr<State> state, std::shared_ptr<MapServer> map_server, std::shared_ptr<Triangulator> tri,
                               int min_init_poses);                                                        // :896-956
    void changeLandmarkAnchor(std::shared_ptr<State> state, std::shared_ptr<MapServer> map_server);        // :273-316
    void changeLandmarkAnchor(std::shared_ptr<State> state, std::shared_ptr<MapServer> map_server,
                              const std::vector<double>& marg_kfs);                                        // :318-361
    int lastRows() const { return _last_rows; }
    int lastInitialised() const { return _last_init; }
    // per frame, for the stream traces (tools/ingvio_replay.cpp --trace; oracle/stream_filter.py): the landmarks the update evaluated
    // (ascending id) with their chi^2 verdicts, and the ids the delayed initialisation added, in the order it added them
    const std::vector<int>& lastUpdateIds() const { return _last_upd_ids; }
    const std::vector<int>& lastUpdateAccepted() const { return _last_upd_acc; }
    const std::vector<int>& lastInitIds() const { return _last_init_ids; }

    // rows x 24 = [extended pose 9 | extrinsics 6 | anchor 6 | landmark 3]  (:521-572 mono, :619-686 stereo)
    static void landmarkRows(const std::shared_ptr<FeatureInfo> feature_info, const std::shared_ptr<State> state, bool stereo,
                             VecXd& res, MatXd& H);
    // rows x 15 = [current clone 6 | anchor 6 | landmark 3]  (:574-617)
    static void landmarkRowsSw(const std::shared_ptr<FeatureInfo> feature_info, const std::shared_ptr<State> state, VecXd& res, MatXd& H);
    // every observation inside the window, columns = all clones in time order (:426-500 mono, :808-894 stereo)
    static void featAllObsRows(const std::shared_ptr<FeatureInfo> feature_info, const std::shared_ptr<State> state, bool stereo,
                               VecXd& res_block, MatXd& Hx_block, MatXd& Hf_block);

protected:
    void update(std::shared_ptr<State> state, std::shared_ptr<MapServer> map_server, bool stereo);
    void initNew(std::shared_ptr<State> state, std::shared_ptr<MapServer> map_server, std::shared_ptr<Triangulator> tri,
                 int min_init_poses, bool stereo);
    void reanchor(std::shared_ptr<State> state, std::shared_ptr<MapServer> map_server, const std::vector<std::shared_ptr<SE3>>& old_anchors);
    double _noise;
    int _last_rows = 0, _last_init = 0;
    std::vector<int> _last_upd_ids, _last_upd_acc, _last_init_ids;
};

}  // namespace ingvio
