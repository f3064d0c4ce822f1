// AnchoredLandmark.h — mirrors the part of ingvio_estimator/src/AnchoredLandmark.h:28-108 the MSCKF
// path needs: every MSCKF feature's p_f (world XYZ) + anchor pose pointer, and the update of
// AnchoredLandmark.cpp:227-243.  The INV_DEPTH / BEARING body representations are only used by the
// SLAM-landmark update (max_landmark_features = 0 in every shipped config) and are
// not carried here.
#pragma once
#include <iostream>

#include "PoseState.h"

namespace ingvio {

class AnchoredLandmark : public Type {
public:
    AnchoredLandmark() : Type(3) {}
    void update(const std::vector<double>& dx) override      // AnchoredLandmark.cpp:227-243
    {
        const Vec3d delta_p(dx[idx()], dx[idx() + 1], dx[idx() + 2]);
        if (_anchored_pose != nullptr) {
            const int a = _anchored_pose->idx();
            const Vec3d delta_theta(dx[a], dx[a + 1], dx[a + 2]);
            _pos_xyz = GammaFunc(delta_theta, 0) * _pos_xyz + GammaFunc(delta_theta, 1) * delta_p;
        } else {
            std::cout << "[AnchoredLandmark]: Warning! Update without anchored pose!" << std::endl;
            _pos_xyz = _pos_xyz + delta_p;
        }
    }
    void setIdentity() override { _pos_xyz = Vec3d(); }
    void resetAnchoredPose(std::shared_ptr<SE3> new_anchored_pose = nullptr, bool = false) { _anchored_pose = new_anchored_pose; }
    const std::shared_ptr<SE3> getAnchoredPose() const { return _anchored_pose; }
    const Vec3d& valuePosXyz() const { return _pos_xyz; }
    void setValuePosXyz(const Vec3d& xyz_world) { _pos_xyz = xyz_world; }
    // first-estimate copy (AnchoredLandmark.cpp: setFejPosXyz); stored, the Jacobians use current values (quirk Q7)
    const Vec3d& fejPosXyz() const { return _pos_xyz_fej; }
    void setFejPosXyz(const Vec3d& xyz_world) { _pos_xyz_fej = xyz_world; }

protected:
    Vec3d _pos_xyz, _pos_xyz_fej;
    std::shared_ptr<SE3> _anchored_pose;
};

}  // namespace ingvio
