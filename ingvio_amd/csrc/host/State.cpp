#include "State.h"

#include <cstdlib>
#include <iostream>

namespace ingvio {

StateParams::StateParams(const IngvioParams& filter_params)      // State.cpp:25-58
{
    _cam_nums = filter_params._cam_nums;
    _max_sw_poses = filter_params._max_sw_clones;
    _max_landmarks = filter_params._max_lm_feats;
    _T_cl2cr = filter_params._T_cr2i.inverse() * filter_params._T_cl2i;
    _T_cl2i = filter_params._T_cl2i;
    _enable_gnss = static_cast<bool>(filter_params._enable_gnss);
    _noise_a = filter_params._noise_a;
    _noise_g = filter_params._noise_g;
    _noise_ba = filter_params._noise_ba;
    _noise_bg = filter_params._noise_bg;
    _init_cov_rot = filter_params._init_cov_rot;
    _init_cov_pos = filter_params._init_cov_pos;
    _init_cov_vel = filter_params._init_cov_vel;
    _init_cov_bg = filter_params._init_cov_bg;
    _init_cov_ba = filter_params._init_cov_ba;
    _init_cov_ext_rot = filter_params._init_cov_ext_rot;
    _init_cov_ext_pos = filter_params._init_cov_ext_pos;
    if (_enable_gnss) {
        _noise_clockbias = filter_params._noise_clockbias;
        _noise_clockbias = filter_params._noise_cb_rw;      // quirk Q1, reproduced (State.cpp:51-52): _noise_cb_rw keeps 0.2
        _init_cov_rcv_clockbias = filter_params._init_cov_rcv_clockbias;
        _init_cov_rcv_clockbias_randomwalk = filter_params._init_cov_rcv_clockbias_randomwalk;
        _init_cov_yof = filter_params._init_cov_yof;
    }
}

void State::construct(const IngvioParams& filter_params)      // State.cpp:60-91
{
    // configuration-time check: the batched landmark update (ingvio_landmark_*) carries INGVIO_LM_MAX landmarks per filter; a
    // larger max_landmark_features is refused here, at start-up, not in the middle of a run when landmark #65 is initialised
    if (filter_params._max_lm_feats > INGVIO_LM_MAX) {
        std::cout << "[State]: max_landmark_features = " << filter_params._max_lm_feats << " exceeds this build's limit of "
                  << INGVIO_LM_MAX << " in-state landmarks" << std::endl;
        std::exit(EXIT_FAILURE);
    }
    int idx = 0;
    _extended_pose = std::make_shared<SE23>();
    _extended_pose->set_cov_idx(idx);
    _err_variables.push_back(_extended_pose);
    idx += _extended_pose->size();
    _bg = std::make_shared<Vec3>();
    _bg->set_cov_idx(idx);
    _err_variables.push_back(_bg);
    idx += _bg->size();
    _ba = std::make_shared<Vec3>();
    _ba->set_cov_idx(idx);
    _err_variables.push_back(_ba);
    idx += _ba->size();
    _camleft_imu_extrinsics = std::make_shared<SE3>();
    _camleft_imu_extrinsics->set_cov_idx(idx);
    _err_variables.push_back(_camleft_imu_extrinsics);
    idx += _camleft_imu_extrinsics->size();
    std::vector<double> cov((size_t)idx * idx, 0.0);
    for (int i = 0; i < idx; ++i) cov[(size_t)i * idx + i] = std::pow(1e-03, 2);      // :88
    if (ingvio_cov_set(_ctx, _b, cov.data(), idx, idx) != INGVIO_OK) {
        std::cout << "[State]: cannot initialise the device covariance: " << ingvio_last_error(_ctx) << std::endl;
        std::exit(EXIT_FAILURE);
    }
    _camleft_imu_extrinsics->setValue(filter_params._T_cl2i.R, filter_params._T_cl2i.t);
}

State::State(const IngvioParams& filter_params) : _state_params(filter_params)
{
    ingvio_ctx_desc d;
    d.batch = 1; d.n_max = filter_params._hip_n_max;
    d.c_max = filter_params._max_sw_clones + 1;                     // <= 36: the ABI refuses larger windows (INGVIO_E_CAPACITY)
    d.f_max = filter_params._hip_f_max; d.device = filter_params._hip_device; d.stream = nullptr;
    d.m_max = 64;
    if (filter_params._max_lm_feats > 0) {
        // SLAM landmarks (f-2): the delayed initialisation has 4 C rows.  The stacked landmark update (up to 4 L rows) does not
        // count: it runs through ingvio_landmark_* with S outside LDS (up to 64 landmarks)
        const int C = d.c_max;
        if (4 * C > d.m_max) d.m_max = 4 * C;
    }
    const int rc = ingvio_ctx_create(&d, &_ctx);
    if (rc != INGVIO_OK) {
        if (rc == INGVIO_E_CAPACITY)
            std::cout << "[State]: libingvio_hip: configuration exceeds this build's limits (window " << d.c_max << " clones <= 36, update rows "
                      << d.m_max << " <= 128)" << std::endl;
        else
            std::cout << "[State]: libingvio_hip: no MI355X context, error " << rc << " (" << (_ctx ? ingvio_last_error(_ctx) : "no device")
                      << ")" << std::endl;
        std::exit(EXIT_FAILURE);
    }
    _own_ctx = true;
    _b = 0;
    construct(filter_params);
}

State::State(const IngvioParams& filter_params, ingvio_ctx* ctx, int b) : _state_params(filter_params), _ctx(ctx), _b(b)
{
    construct(filter_params);
}

State::~State()
{
    if (_own_ctx && _ctx) ingvio_ctx_destroy(_ctx);
}

int State::curr_cov_size()
{
    int n = 0;
    ingvio_get_n(_ctx, _b, &n);
    return n;
}

void State::initStateAndCov(double init_timestamp, const Quatd& init_quat_i2w, const Vec3d& init_pos,
                            const Vec3d& init_vel, const Vec3d& init_bg, const Vec3d& init_ba)      // State.cpp:126-167
{
    _timestamp = init_timestamp;
    const int n = curr_cov_size();
    std::vector<double> cov((size_t)n * n);
    ingvio_cov_get(_ctx, _b, cov.data(), n);
    auto setd = [&](int i0, int cnt, double s) { for (int i = i0; i < i0 + cnt; ++i) cov[(size_t)i * n + i] = std::pow(s, 2.0); };
    setd(_extended_pose->idx(), 3, _state_params._init_cov_rot);
    setd(_extended_pose->idx() + 3, 3, _state_params._init_cov_pos);
    setd(_extended_pose->idx() + 6, 3, _state_params._init_cov_vel);
    setd(_bg->idx(), 3, _state_params._init_cov_bg);
    setd(_ba->idx(), 3, _state_params._init_cov_ba);
    setd(_camleft_imu_extrinsics->idx(), 3, _state_params._init_cov_ext_rot);
    setd(_camleft_imu_extrinsics->idx() + 3, 3, _state_params._init_cov_ext_pos);
    ingvio_cov_set(_ctx, _b, cov.data(), n, n);
    _extended_pose->setValueLinearByQuat(init_quat_i2w);
    _extended_pose->setValueTrans1(init_pos);
    _extended_pose->setValueTrans2(init_vel);
    _bg->setValue(init_bg);
    _ba->setValue(init_ba);
    _camleft_imu_extrinsics->setValue(_state_params._T_cl2i.R, _state_params._T_cl2i.t);
}

void State::initStateAndCov(double t, const Quatd& q, const Vec3d& p) { initStateAndCov(t, q, p, Vec3d(), Vec3d(), Vec3d()); }
void State::initStateAndCov(double t, const Quatd& q) { initStateAndCov(t, q, Vec3d()); }

}  // namespace ingvio
