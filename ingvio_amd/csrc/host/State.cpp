#include "State.h"

#include <cstdlib>
#include <iostream>

namespace ingvio {

// Licence note: this file mirrors the MEMBER LIST and the registration ORDER of the reference's State.cpp:25-91 (InGVIO, (C) 2022
// Changwu Liu, GNU GPL v3 or later) because the drop-in contract fixes both (the (idx, size) table every caller indexes into); it is
// distributed under the same licence.  The copying itself is table-driven here.
namespace {
// which IngvioParams member feeds which StateParams member (State.cpp:27-48)
struct ParamRow { double StateParams::*dst; double IngvioParams::*src; bool gnss_only; };
const ParamRow kParamRows[] = {
    { &StateParams::_noise_a, &IngvioParams::_noise_a, false }, { &StateParams::_noise_g, &IngvioParams::_noise_g, false },
    { &StateParams::_noise_ba, &IngvioParams::_noise_ba, false }, { &StateParams::_noise_bg, &IngvioParams::_noise_bg, false },
    { &StateParams::_init_cov_rot, &IngvioParams::_init_cov_rot, false }, { &StateParams::_init_cov_pos, &IngvioParams::_init_cov_pos, false },
    { &StateParams::_init_cov_vel, &IngvioParams::_init_cov_vel, false }, { &StateParams::_init_cov_bg, &IngvioParams::_init_cov_bg, false },
    { &StateParams::_init_cov_ba, &IngvioParams::_init_cov_ba, false }, { &StateParams::_init_cov_ext_rot, &IngvioParams::_init_cov_ext_rot, false },
    { &StateParams::_init_cov_ext_pos, &IngvioParams::_init_cov_ext_pos, false },
    // only with enable_gnss (:49-57).  Quirk Q1 (State.cpp:51-52): BOTH clock noises are written to _noise_clockbias, in this order,
    // so it ends up holding the random-walk value and _noise_cb_rw keeps its default 0.2
    { &StateParams::_noise_clockbias, &IngvioParams::_noise_clockbias, true }, { &StateParams::_noise_clockbias, &IngvioParams::_noise_cb_rw, true },
    { &StateParams::_init_cov_rcv_clockbias, &IngvioParams::_init_cov_rcv_clockbias, true },
    { &StateParams::_init_cov_rcv_clockbias_randomwalk, &IngvioParams::_init_cov_rcv_clockbias_randomwalk, true },
    { &StateParams::_init_cov_yof, &IngvioParams::_init_cov_yof, true },
};
}  // namespace

StateParams::StateParams(const IngvioParams& fp)
    : _cam_nums(fp._cam_nums), _max_sw_poses(fp._max_sw_clones), _max_landmarks(fp._max_lm_feats), _enable_gnss(fp._enable_gnss != 0),
      _T_cl2cr(fp._T_cr2i.inverse() * fp._T_cl2i), _T_cl2i(fp._T_cl2i)
{
    for (const ParamRow& r : kParamRows)
        if (!r.gnss_only || _enable_gnss) this->*r.dst = fp.*r.src;
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
    // the four fixed variables in covariance order: extended pose (9), gyro bias (3), accelerometer bias (3), camera extrinsics (6)
    _extended_pose = std::make_shared<SE23>();
    _bg = std::make_shared<Vec3>();
    _ba = std::make_shared<Vec3>();
    _camleft_imu_extrinsics = std::make_shared<SE3>();
    int idx = 0;
    for (const std::shared_ptr<Type>& v : std::initializer_list<std::shared_ptr<Type>>{ _extended_pose, _bg, _ba, _camleft_imu_extrinsics }) {
        v->set_cov_idx(idx);
        _err_variables.push_back(v);
        idx += v->size();
    }
    std::vector<double> cov((size_t)idx * idx, 0.0);
    for (int i = 0; i < idx; ++i) cov[(size_t)i * idx + i] = 1e-3 * 1e-3;                 // :88: (1e-3)^2 I
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
