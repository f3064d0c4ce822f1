// IngvioParams.h — the subset of ingvio_estimator/src/IngvioParams.h:30-146 the covariance hot path
// reads (the OpenCV-YAML reader IngvioParams.cpp:27-174 is out of scope: no OpenCV in this image; the
// defaults below are the shipped config/sportsfield/ingvio_stereo.yaml values).
#pragma once
#include "Mat3.h"

namespace ingvio {

struct Iso3 {            // Eigen::Isometry3d stand-in
    Mat3d R = Mat3d::Identity();
    Vec3d t;
    Iso3 inverse() const { Iso3 r; r.R = R.transpose(); r.t = -(r.R * t); return r; }
    Iso3 operator*(const Iso3& o) const { Iso3 r; r.R = R * o.R; r.t = R * o.t + t; return r; }
    Vec3d operator*(const Vec3d& p) const { return R * p + t; }
};

class IngvioParams {
public:
    int _cam_nums = 2;
    int _max_sw_clones = 27;
    int _is_key_frame = 1;
    int _max_lm_feats = 0;
    int _enable_gnss = 1;
    double _noise_g = 0.004, _noise_a = 0.08, _noise_bg = 0.0002, _noise_ba = 0.008;
    double _noise_clockbias = 2.0, _noise_cb_rw = 0.2;
    double _init_cov_rot = 0.0, _init_cov_pos = 0.0, _init_cov_vel = 0.25, _init_cov_bg = 0.01, _init_cov_ba = 0.01;
    double _init_cov_ext_rot = 1.8e-2, _init_cov_ext_pos = 2e-3;
    double _init_cov_rcv_clockbias = 2.0, _init_cov_rcv_clockbias_randomwalk = 1.0, _init_cov_yof = 0.015;
    double _init_gravity = 9.8;
    int _max_imu_buffer_size = 3000, _init_imu_buffer_sp = 300;
    double _trans_thres = 0.25;
    // triangulation (IngvioParams.cpp:78-85, config/*/ingvio_*.yaml)
    double _huber_epsilon = 0.01, _conv_precision = 5e-07, _init_damping = 1e-03, _max_depth = 40.0, _min_depth = 0.2;
    int _outer_loop_max_iter = 10, _inner_loop_max_iter = 10;
    int _chi2_max_dof = 150;
    double _chi2_thres = 0.95;
    double _visual_noise = 0.18;
    int _frame_select_interval = 18;
    int _is_gnss_chi2_test = 0, _is_gnss_strong_reject = 1, _is_adjust_yof = 0;
    int _use_fix_time_offset = 0;                      // IngvioParams.cpp:113-116 (shipped configs: 1 with gnss_local_offset -18.0)
    double _gnss_local_offset = 0.0;
    double _psr_noise_amp = 1.0, _dopp_noise_amp = 1.0;
    // GvioAligner (IngvioParams.cpp:121-124, config/fw_zed2i_f9p/ingvio_stereo.yaml:70-73)
    int _gv_align_batch_size = 25, _gv_align_max_iter = 10;
    double _gv_align_conv_epsilon = 1e-05, _gv_align_vel_thres = 0.4;
    Iso3 _T_cl2i, _T_cr2i;

    // device-side capacity of the covariance engine behind this filter (new: not in the reference)
    int _hip_n_max = 256, _hip_f_max = 160, _hip_device = 0;
    // quirks Q3 / Q2 as parameters (defaults = the reference as written): RemoveLost's accepted-feature cap
    // (RemoveLostUpdate.h:38: 20; 0 = no cap) and its compression rule (0 = keep all rows after the rotation, RemoveLostUpdate.cpp:390;
    // 1 = keep the top n rows like the other two update classes)
    int _hip_max_valid_ids = 20, _hip_compress_rule = 0;
    int _hip_fuse_triangulation = 1;      // RemoveLost: triangulation + update in one device round trip (ingvio_msckf_update_tri); 0: two calls
};

}  // namespace ingvio
