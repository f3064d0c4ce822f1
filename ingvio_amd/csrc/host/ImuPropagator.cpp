#include "ImuPropagator.h"

#include <iostream>

#include "StateManager.h"

namespace ingvio {

// Eigen::Quaterniond::FromTwoVectors(a, b)
static Quatd fromTwoVectors(const Vec3d& a, const Vec3d& b)
{
    const double na = a.norm(), nb = b.norm();
    const Vec3d v0 = a * (1.0 / na), v1 = b * (1.0 / nb);
    const double c = v0[0] * v1[0] + v0[1] * v1[1] + v0[2] * v1[2];
    Quatd q;
    if (c < -1.0 + 1e-12) {      // opposite: any orthogonal axis
        Vec3d ax = std::fabs(v0[0]) < 0.9 ? Vec3d(1, 0, 0) : Vec3d(0, 1, 0);
        Vec3d ort(v0[1] * ax[2] - v0[2] * ax[1], v0[2] * ax[0] - v0[0] * ax[2], v0[0] * ax[1] - v0[1] * ax[0]);
        const double n = ort.norm();
        q.w = 0; q.x = ort[0] / n; q.y = ort[1] / n; q.z = ort[2] / n;
        return q;
    }
    const Vec3d axis(v0[1] * v1[2] - v0[2] * v1[1], v0[2] * v1[0] - v0[0] * v1[2], v0[0] * v1[1] - v0[1] * v1[0]);
    const double s = std::sqrt((1.0 + c) * 2.0), invs = 1.0 / s;
    q.w = s * 0.5; q.x = axis[0] * invs; q.y = axis[1] * invs; q.z = axis[2] * invs;
    return q;
}

void ImuPropagator::storeImu(const ImuCtrl& imu_ctrl)
{
    if ((int)_imu_ctrl_buffer.size() > _max_imu_buffer_size) {
        std::cout << "[ImuPropagator]: Exceeding imu max buffer size, throw curr imu ctrl!" << std::endl;
        return;
    } else
        _imu_ctrl_buffer.push_back(imu_ctrl);
    if (!_has_gravity_set && _init_imu_buffer_sp > 0) {
        if ((int)_imu_ctrl_buffer.size() < _init_imu_buffer_sp) return;
        std::cout << "[ImuPropagator]: Start init gravity norm ..." << std::endl;
        Vec3d sum_gravity;
        for (const auto& item : _imu_ctrl_buffer) sum_gravity += item._accel_raw;
        sum_gravity = sum_gravity * (1.0 / _imu_ctrl_buffer.size());
        if (std::fabs(sum_gravity.norm() - _init_gravity) / _init_gravity > 0.02) {
            std::cout << "[ImuPropagator]: Keep Camera STEADY!! Reinit gravity ..." << std::endl;
            _imu_ctrl_buffer.clear();
            _has_gravity_set = false;
        } else {
            _gravity = Vec3d(0.0, 0.0, -sum_gravity.norm());
            _quat_init = fromTwoVectors(-sum_gravity, _gravity);
            _has_gravity_set = true;
        }
    }
}

// attitude from the mean specific force of the newest num_ctrls samples: the rotation that takes -f to straight down
bool ImuPropagator::getAvgQuat(Quatd& quat_avg, int num_ctrls)
{
    const int n = (int)_imu_ctrl_buffer.size();
    if (n == 0 || num_ctrls <= 0) { quat_avg = Quatd{ 1, 0, 0, 0 }; return false; }
    const int first = n - num_ctrls > 0 ? n - num_ctrls : 0;
    Vec3d sum_sf;
    for (int i = n - 1; i >= first; --i) sum_sf += _imu_ctrl_buffer[i]._accel_raw;
    sum_sf = sum_sf * (1.0 / (n - first));
    sum_sf = sum_sf * (1.0 / sum_sf.norm());
    quat_avg = fromTwoVectors(-sum_sf, Vec3d(0, 0, -1));
    return true;
}

void ImuPropagator::stateAndCovTransition(std::shared_ptr<State> state, const ImuCtrl& imu_ctrl, double dt,
                                          double Phi[225], double G[180], bool isAnalytic)
{
    Mat3d R = state->_extended_pose->valueLinearAsMat();
    Vec3d p = state->_extended_pose->valueTrans1(), v = state->_extended_pose->valueTrans2();
    state->_timestamp += dt;                                                                  // :124
    (isAnalytic ? imuTransitionAnalytic : imuTransitionRK4)(R, p, v, state->_bg->value(), state->_ba->value(), imu_ctrl._gyro_raw,
                                                            imu_ctrl._accel_raw, _gravity, dt, Phi, G);   // :119-162 | :163-229
    state->_extended_pose->setValueLinearByMat(R);
    state->_extended_pose->setValueTrans1(p);
    state->_extended_pose->setValueTrans2(v);
    if (state->_state_params._enable_gnss)                                                    // :139-148
        for (int i = 0; i < 4; ++i)
            if (state->_gnss.find(i) != state->_gnss.end() && state->_gnss.find(State::FS) != state->_gnss.end())
                state->_gnss.at(i)->setValue(state->_gnss.at(i)->value() + dt * state->_gnss.at(State::FS)->value());
}

void ImuPropagator::propagateUntil(std::shared_ptr<State> state, double t_end, bool isAnalytic)
{
    if (!_has_gravity_set || t_end <= state->_timestamp) return;
    if (_imu_ctrl_buffer.size() == 0) return;
    if (_imu_ctrl_buffer[0]._timestamp > t_end) return;
    int propa_cnt = 0;
    ImuCtrl last_imu_ctrl = _imu_ctrl_buffer[_imu_ctrl_buffer.size() - 1];
    std::vector<double> Phis, Gs, dts;      // covariance steps, flushed as one fused launch
    auto step = [&](const ImuCtrl& ctrl, double dt) {
        Phis.resize(Phis.size() + 225); Gs.resize(Gs.size() + 180); dts.push_back(dt);
        this->stateAndCovTransition(state, ctrl, dt, &Phis[Phis.size() - 225], &Gs[Gs.size() - 180], isAnalytic);
        if (!_fuse_steps || dts.size() == 64) {
            StateManager::propagateStateCovFused(state, (int)dts.size(), Phis.data(), Gs.data(), dts.data());
            Phis.clear(); Gs.clear(); dts.clear();
        }
    };
    for (size_t i = 0; i < _imu_ctrl_buffer.size(); ++i) {
        const double ctrl_time = _imu_ctrl_buffer[i]._timestamp;
        if (ctrl_time < state->_timestamp) { ++propa_cnt; continue; }
        if (ctrl_time > t_end) break;
        ++propa_cnt;
        const double dt = ctrl_time - state->_timestamp;
        if (dt < 1e-6) continue;
        last_imu_ctrl = _imu_ctrl_buffer[i];
        step(_imu_ctrl_buffer[i], dt);
    }
    if (state->_timestamp < t_end) {
        const double dt_last = t_end - state->_timestamp;
        if (dt_last > 1e-06) step(last_imu_ctrl, dt_last);
        else state->_timestamp = t_end;
    }
    if (!dts.empty()) StateManager::propagateStateCovFused(state, (int)dts.size(), Phis.data(), Gs.data(), dts.data());
    _imu_ctrl_buffer.erase(_imu_ctrl_buffer.begin(), _imu_ctrl_buffer.begin() + propa_cnt);
}

void ImuPropagator::propagateAugmentAtEnd(std::shared_ptr<State> state, double t_end, bool isAnalytic)
{
    if (!_has_gravity_set) return;
    this->propagateUntil(state, t_end, isAnalytic);
    if (state->_timestamp < t_end) {
        std::cout << "[ImuPropagator]: Cannot propa to t_end due to no imu ctrl!" << std::endl;
        return;
    } else if (state->_timestamp > t_end) {
        std::cout << "[IMUPropagator]: Cannot propa because t_end < curr state time!" << std::endl;
        return;
    }
    StateManager::augmentSlidingWindowPose(state);
}

void ImuPropagator::propagateToExpectedPoseAndAugment(std::shared_ptr<State> state, double t_end, const Mat3d& R_i2w, const Vec3d& p_i2w)
{
    if (!_has_gravity_set) return;
    state->_extended_pose->setValueLinearByMat(R_i2w);
    state->_extended_pose->setValueTrans1(p_i2w);
    state->_extended_pose->setValueTrans2(Vec3d());
    state->_bg->setIdentity();
    state->_ba->setIdentity();
    state->_camleft_imu_extrinsics->setIdentity();
    state->_timestamp = t_end;
    StateManager::augmentSlidingWindowPose(state);
}

}  // namespace ingvio
