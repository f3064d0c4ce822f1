// ImuPropagator.h — mirrors ingvio_estimator/src/ImuPropagator.h:36-188 (ROS message ctor replaced
// by a POD ctor).  Nominal-state integration + (Phi, G) stay on the host; the
// covariance part of every step goes to StateManager::propagateStateCov(Fused).
#pragma once
#include <memory>
#include <vector>

#include "ImuTransition.h"
#include "IngvioParams.h"
#include "PoseState.h"

namespace ingvio {

class State;

class ImuCtrl {
public:
    ImuCtrl() : _timestamp(-1) {}
    ImuCtrl(double t, const Vec3d& accel, const Vec3d& gyro) : _timestamp(t), _accel_raw(accel), _gyro_raw(gyro) {}
    double _timestamp;
    Vec3d _accel_raw;
    Vec3d _gyro_raw;
};

class ImuPropagator {
public:
    ImuPropagator() : _has_gravity_set(true), _max_imu_buffer_size(1000), _init_imu_buffer_sp(-1), _init_gravity(9.8),
                      _gravity(0.0, 0.0, -9.8), _fuse_steps(true) {}                          // ImuPropagator.h:89-95
    ImuPropagator(const IngvioParams& filter_params)                                          // :97-105
        : _has_gravity_set(false), _max_imu_buffer_size(filter_params._max_imu_buffer_size),
          _init_imu_buffer_sp(filter_params._init_imu_buffer_sp), _init_gravity(filter_params._init_gravity),
          _gravity(0.0, 0.0, -filter_params._init_gravity), _fuse_steps(true)
    {
        if (_init_imu_buffer_sp < 0) _has_gravity_set = true;
    }

    void storeImu(const ImuCtrl& imu_ctrl);                                                   // ImuPropagator.cpp:29-70
    void stateAndCovTransition(std::shared_ptr<State> state, const ImuCtrl& imu_ctrl, double dt,
                               double Phi[225], double G[180], bool isAnalytic = true);        // :98-230 (both branches)
    void propagateUntil(std::shared_ptr<State> state, double t_end, bool isAnalytic = true);  // :232-292
    void propagateAugmentAtEnd(std::shared_ptr<State> state, double t_end, bool isAnalytic = true);   // :294-314
    void propagateToExpectedPoseAndAugment(std::shared_ptr<State> state, double t_end, const Mat3d& R_i2w, const Vec3d& p_i2w);   // :316-334

    bool isInit() const { return _has_gravity_set; }
    const Vec3d& getGravity() const { return _gravity; }
    const Quatd& getInitQuat() const { return _quat_init; }
    bool getInitQuat(Quatd& quat_init) const                                                  // ImuPropagator.h:133-145
    {
        quat_init = _has_gravity_set ? _quat_init : Quatd{ 1, 0, 0, 0 };
        return _has_gravity_set;
    }
    bool getAvgQuat(Quatd& quat_avg, int num_ctrls = 1);                                      // ImuPropagator.cpp:72-96
    size_t bufferSize() const { return _imu_ctrl_buffer.size(); }
    void setGravityInitialised(const Vec3d& g) { _gravity = g; _has_gravity_set = true; }
    // true (default): the k covariance steps of one propagateUntil call are sent as ONE fused launch;
    // false: one launch per IMU sample exactly like the reference loop.
    void setFuseSteps(bool f) { _fuse_steps = f; }

protected:
    std::vector<ImuCtrl> _imu_ctrl_buffer;
    bool _has_gravity_set;
    int _max_imu_buffer_size, _init_imu_buffer_sp;
    double _init_gravity;
    Vec3d _gravity;
    Quatd _quat_init{ 1, 0, 0, 0 };
    bool _fuse_steps;
};

}  // namespace ingvio
