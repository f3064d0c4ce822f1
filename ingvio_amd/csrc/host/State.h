// State.h — mirrors ingvio_estimator/src/State.h:36-139.  The public members are the reference's;
// the private covariance `Eigen::MatrixXd _cov` (State.h:133) becomes filter `_b` of an
// `ingvio_ctx` in HBM, still reachable only through `friend class StateManager`.
// Licence note: class, member and accessor names below are those of the reference (InGVIO, (C) 2022 Changwu Liu, GNU GPL v3 or later)
// because the drop-in contract is source compatibility with code written against them; distributed under the same licence.
#pragma once
#include <cmath>
#include <map>
#include <memory>
#include <unordered_map>
#include <vector>

#include "../../../include/ingvio_hip.h"
#include "AnchoredLandmark.h"
#include "IngvioParams.h"
#include "PoseState.h"
#include "VecState.h"

namespace ingvio {

class StateParams {
public:
    StateParams() = default;
    StateParams(const IngvioParams& filter_params);      // State.cpp:25-58 (incl. quirk Q1)

    int _cam_nums = 2;
    int _max_sw_poses = 20;
    int _max_landmarks = 25;
    double _noise_g = 0.005, _noise_a = 0.05, _noise_bg = 0.001, _noise_ba = 0.01;
    double _noise_clockbias = 2.0, _noise_cb_rw = 0.2;
    double _init_cov_rot = 0, _init_cov_pos = 0, _init_cov_vel = 0, _init_cov_bg = 0, _init_cov_ba = 0;
    double _init_cov_ext_rot = 0, _init_cov_ext_pos = 0;
    double _init_cov_rcv_clockbias = 0, _init_cov_rcv_clockbias_randomwalk = 0, _init_cov_yof = 0;
    bool _enable_gnss = true;
    Iso3 _T_cl2cr, _T_cl2i;
};

class State {
public:
    enum GNSSType { GPS = 0, GLO, GAL, BDS, FS, YOF };

    // owns a private 1-filter context
    State(const IngvioParams& filter_params);
    // attaches to filter `b` of a shared (batched) context; the caller keeps the context alive
    State(const IngvioParams& filter_params, ingvio_ctx* ctx, int b);
    ~State();
    State(const State&) = delete;

    double nextMargTime()      // State.h:83-92
    {
        double time = INFINITY;
        if ((int)_sw_camleft_poses.size() > _state_params._max_sw_poses)
            for (const auto& item : _sw_camleft_poses)
                if (item.first < time) time = item.first;
        return time;
    }

    void initStateAndCov(double init_timestamp, const Quatd& init_quat_i2w);
    void initStateAndCov(double init_timestamp, const Quatd& init_quat_i2w, const Vec3d& init_pos);
    void initStateAndCov(double init_timestamp, const Quatd& init_quat_i2w, const Vec3d& init_pos,
                         const Vec3d& init_vel, const Vec3d& init_bg, const Vec3d& init_ba);

    int curr_cov_size();
    int curr_err_variable_size() { return (int)_err_variables.size(); }

    double _timestamp = -1;
    StateParams _state_params;
    std::shared_ptr<SE23> _extended_pose;
    std::shared_ptr<Vec3> _bg;
    std::shared_ptr<Vec3> _ba;
    std::shared_ptr<SE3> _camleft_imu_extrinsics;
    std::unordered_map<int, std::shared_ptr<Scalar>> _gnss;      // keyed by GNSSType
    std::unordered_map<int, std::shared_ptr<AnchoredLandmark>> _anchored_landmarks;
    std::map<double, std::shared_ptr<SE3>, std::less<double>> _sw_camleft_poses;

    // device handle (new): for callers that batch several States on one context
    ingvio_ctx* hipContext() const { return _ctx; }
    int hipFilterIndex() const { return _b; }

private:
    friend class StateManager;
    void construct(const IngvioParams& filter_params);
    ingvio_ctx* _ctx = nullptr;      // replaces Eigen::MatrixXd _cov
    int _b = 0;
    bool _own_ctx = false;
    std::vector<std::shared_ptr<Type>> _err_variables;
};

}  // namespace ingvio
