// MsckfUpdates.h — mirrors the three MSCKF update policies of the reference:
//   RemoveLostUpdate  (RemoveLostUpdate.h:30-75, .cpp:40-167,276-405)
//   SwMargUpdate      (SwMargUpdate.h, .cpp:42-189,216-497)
//   KeyframeUpdate    (KeyframeUpdate.h, .cpp:43-129,280-328,438-761)
// "Who/what to update" (selection, anchor changes, observation cleaning, marginalisation order) is
// restated here on the host; the per-feature Jacobians, nullspace, chi^2 gate, stacking, compression
// and the Kalman update are ONE call into libingvio_hip.so (StateManager::msckfUpdate) on a
// flattened copy of the MapServer.
#pragma once
#include <memory>
#include <vector>

#include <map>

#include "ingvio_hip.h"
#include "IngvioParams.h"
#include "MapServer.h"
#include "Update.h"

namespace ingvio {

class State;

// Triangulator (Triangulator.h:30-120) on the device: the LM iteration of triangulateMonoObs / triangulateStereoObs runs
// in ingvio_triangulate (kernels_tri.hip).  triangulate() is FeatureInfoManager::triangulateFeatureInfo{Mono,Stereo}
// (MapServerManager.cpp:274-341): count the attempt, reject points behind their anchor, set value (and FEJ the first time).
class Triangulator {
public:
    Triangulator() {}
    explicit Triangulator(const IngvioParams& filter_params)
        : _trans_thres(filter_params._trans_thres), _huber_epsilon(filter_params._huber_epsilon),
          _conv_precision(filter_params._conv_precision), _init_damping(filter_params._init_damping),
          _outer_loop_max_iter(filter_params._outer_loop_max_iter), _inner_loop_max_iter(filter_params._inner_loop_max_iter),
          _max_depth(filter_params._max_depth), _min_depth(filter_params._min_depth) {}
    virtual ~Triangulator() {}
    bool triangulateMonoObs(const std::shared_ptr<State> state, const std::map<double, std::shared_ptr<MonoMeas>>& mono_obs,
                            const std::map<double, std::shared_ptr<SE3>>& sw_poses, Vec3d& pf) const;
    bool triangulateStereoObs(const std::shared_ptr<State> state, const std::map<double, std::shared_ptr<StereoMeas>>& stereo_obs,
                              const std::map<double, std::shared_ptr<SE3>>& sw_poses, const Iso3& T_cl2cr, Vec3d& pf) const;
    virtual bool triangulate(std::shared_ptr<FeatureInfo> feature_info, const std::shared_ptr<State> state, bool stereo);
    // The same for a list of features in ONE device call (the reference triangulates inside its per-feature loops,
    // RemoveLostUpdate.cpp:283-299 / SwMargUpdate.cpp:236-257: each feature's triangulation reads the window poses only, so
    // the loop order is immaterial): ok[i] = what triangulate(features[i]) returns.  The window is sent once, each feature
    // names its observations by a mask over the window slots.
    virtual void triangulateMany(const std::vector<std::shared_ptr<FeatureInfo>>& features, const std::shared_ptr<State> state, bool stereo,
                                 std::vector<char>& ok);
    // this object's parameters as the device call takes them (RemoveLostUpdate hands them to ingvio_msckf_update_tri)
    void fillOpts(const std::shared_ptr<State>& state, bool stereo, ingvio_tri_opts& o) const;

protected:
    bool accept(std::shared_ptr<FeatureInfo> fi, bool flag, const Vec3d& pf) const;      // MapServerManager.cpp:283-305 / :318-340
    bool run(const std::shared_ptr<State> state, const std::map<double, std::shared_ptr<SE3>>& sw_poses,
             const std::vector<double>& stamps, const std::vector<double>& uv4, bool stereo, const Iso3& T_cl2cr, Vec3d& pf) const;
    double _trans_thres = 0.1, _huber_epsilon = 0.01, _conv_precision = 5e-7, _init_damping = 1e-3;      // Triangulator.h:67-75
    int _outer_loop_max_iter = 10, _inner_loop_max_iter = 10;
    double _max_depth = 60.0, _min_depth = 0.2;
};

// What one update call consumed and decided (read by tools/ingvio_replay --trace for the golden-stream comparison, tests/
// test_stream_golden.py): feature ids in the order they were handed to the device, their accept flags, the ids dropped without an
// update, the clone stamps the update was restricted to.
struct UpdateRecord {
    std::vector<int> ids, accepted, direct;
    std::vector<double> stamps;
    int rows = 0;
    void clear() { ids.clear(); accepted.clear(); direct.clear(); stamps.clear(); rows = 0; }
};
// what the window maintenance after the updates did: ids erased when their observations ran out, ids erased / re-anchored by the
// anchor change, the marginalised clone stamps
struct MaintenanceRecord {
    std::vector<int> clean_erased, anchor_erased, anchor_moved;
    std::vector<double> marg_stamps;
    void clear() { clean_erased.clear(); anchor_erased.clear(); anchor_moved.clear(); marg_stamps.clear(); }
};

class RemoveLostUpdate : public UpdateBase {
public:
    RemoveLostUpdate(const IngvioParams& filter_params);
    void updateStateMono(std::shared_ptr<State> state, std::shared_ptr<MapServer> map_server, std::shared_ptr<Triangulator> tri);
    void updateStateStereo(std::shared_ptr<State> state, std::shared_ptr<MapServer> map_server, std::shared_ptr<Triangulator> tri);
    int lastRows() const { return _last_rows; }
    int lastAccepted() const { return _last_accepted; }
    const UpdateRecord& lastRecord() const { return _rec; }
    void setMaxValidIds(int n) { _max_valid_ids = n; }      // RemoveLostUpdate.h:38 (20)

protected:
    void update(std::shared_ptr<State> state, std::shared_ptr<MapServer> map_server, std::shared_ptr<Triangulator> tri, bool stereo);
    int _max_valid_ids;
    int _compress_rule = 0;
    bool _fuse_tri = true;      // IngvioParams::_hip_fuse_triangulation
    double _noise;
    int _last_rows = 0, _last_accepted = 0;
    UpdateRecord _rec;
};

class SwMargUpdate : public UpdateBase {
public:
    SwMargUpdate(const IngvioParams& filter_params);
    void updateStateMono(std::shared_ptr<State> state, std::shared_ptr<MapServer> map_server, std::shared_ptr<Triangulator> tri);
    void updateStateStereo(std::shared_ptr<State> state, std::shared_ptr<MapServer> map_server, std::shared_ptr<Triangulator> tri);
    void cleanMonoObsAtMargTime(std::shared_ptr<State> state, std::shared_ptr<MapServer> map_server);      // SwMargUpdate.cpp:191-214
    void cleanStereoObsAtMargTime(std::shared_ptr<State> state, std::shared_ptr<MapServer> map_server);    // :421-444
    void changeMSCKFAnchor(std::shared_ptr<State> state, std::shared_ptr<MapServer> map_server);           // :367-410
    void margSwPose(std::shared_ptr<State> state);                                                         // :412-419
    void selectSwTimestamps(const std::map<double, std::shared_ptr<SE3>>& sw_poses, const double& marg_time,
                            std::vector<double>& selected_timestamps);                                     // :475-497
    int lastRows() const { return _last_rows; }
    const UpdateRecord& lastRecord() const { return _rec; }
    MaintenanceRecord& maintenance() { return _maint; }

protected:
    void update(std::shared_ptr<State> state, std::shared_ptr<MapServer> map_server, std::shared_ptr<Triangulator> tri, bool stereo);
    double _noise;
    int _frame_select_interval;
    bool _fuse_tri = true;      // IngvioParams::_hip_fuse_triangulation
    int _last_rows = 0;
    UpdateRecord _rec;
    MaintenanceRecord _maint;
};

class KeyframeUpdate : public UpdateBase {
public:
    KeyframeUpdate(const IngvioParams& filter_params);
    void getMargKfs(const std::shared_ptr<State> state, std::vector<double>& marg_kfs);                    // KeyframeUpdate.cpp:43-116
    void updateStateMono(std::shared_ptr<State> state, std::shared_ptr<MapServer> map_server, std::shared_ptr<Triangulator> tri);
    void updateStateStereo(std::shared_ptr<State> state, std::shared_ptr<MapServer> map_server, std::shared_ptr<Triangulator> tri);
    void cleanMonoObsAtMargTime(std::shared_ptr<State> state, std::shared_ptr<MapServer> map_server);
    void cleanStereoObsAtMargTime(std::shared_ptr<State> state, std::shared_ptr<MapServer> map_server);    // :737-761
    void changeMSCKFAnchor(std::shared_ptr<State> state, std::shared_ptr<MapServer> map_server);           // :280-328
    void margSwPose(std::shared_ptr<State> state);                                                         // :118-129
    int lastRows() const { return _last_rows; }
    const UpdateRecord& lastRecord() const { return _rec; }
    MaintenanceRecord& maintenance() { return _maint; }

protected:
    void update(std::shared_ptr<State> state, std::shared_ptr<MapServer> map_server, std::shared_ptr<Triangulator> tri, bool stereo);
    UpdateRecord _rec;
    MaintenanceRecord _maint;
    double _noise;
    int _max_sw_poses;
    bool _fuse_tri = true;      // IngvioParams::_hip_fuse_triangulation
    // the reference keeps this counter as a process-global static (KeyframeUpdate.cpp:41); per filter here
    int _select_cnt = 0;
    double _timestamp = -1;
    std::vector<double> _kfs;
    int _last_rows = 0;
};

// MapServerManager::markMarg{Mono,Stereo}Features (MapServerManager.cpp:219-273): every feature without an observation at the
// current state time is flagged; flagged SLAM landmarks leave the state and the map at once
void markMargFeatures(std::shared_ptr<MapServer> map_server, std::shared_ptr<State> state, bool stereo);
// MapServerManager::eraseInvalidFeatures (MapServerManager.cpp:454-490), MSCKF part
void eraseInvalidFeatures(std::shared_ptr<MapServer> map_server, std::shared_ptr<State> state, std::vector<int>* erased = nullptr);

}  // namespace ingvio
