// GvioAligner.h — ingvio_estimator/src/GvioAligner.{h,cpp}: the one-off batch alignment of the VIO world frame with ENU / ECEF
// (second half).  Same public surface as the reference class (isAlign, getYawOffset, getRenu2ecef,
// getTenu2ecef, ..., batchAlign); the three stages of batchAlign (GvioAligner.cpp:199-383) are restated here:
//   coarseLocalization  gnss_comm::psr_pos on the observations of ALL buffered epochs (one receiver position)      :199-233
//   yawAlignment        Gauss-Newton on (yaw offset, receiver clock drift) from the Doppler residuals               :235-312
//   anchorRefinement    per-epoch psr_pos, anchor = mean(spp_i - R_w2ecef p_i), as written (the loop leaves on the
//                       first step that has NOT converged, :367-368)                                              :314-383
// The satellite geodesy underneath (sat_states, atmosphere, psr_res, dopp_res) runs on the device for all epochs of the buffer
// at once (ingvio_gnss_sat_eval, kernels_gnss.hip); the small normal equations (7 x 7, 2 x 2) are solved here.
#pragma once
#include <memory>
#include <vector>

#include "GnssUpdate.h"
#include "IngvioParams.h"
#include "ingvio_hip.h"

namespace ingvio {

class SE23;

// One raw GNSS epoch as GnssProcessor hands it to the aligner (GnssData.h GnssMeas = (obs, ephems)): flat records in the layout
// of ingvio_gnss_epoch (include/ingvio_hip.h): eph [n][INGVIO_EPH_N], obs [n][INGVIO_OBS_N].
struct RawGnssEpoch {
    std::vector<double> eph, obs;
    double doy = 0.0;
    int n_sat() const { return (int)(obs.size() / INGVIO_OBS_N); }
};

class GvioAligner {
public:
    GvioAligner(ingvio_ctx* ctx, int batch_size = 25, int max_iter = 10, double conv_epsilon = 1e-5, double vel_thres = 0.4)
        : _ctx(ctx), _batch_size(batch_size), _max_iter(max_iter), _conv_epsilon(conv_epsilon), _vel_thres(vel_thres) {}
    bool isAlign() const { return _isAligned; }
    double getYawOffset() const { return _yaw_offset; }
    Mat3d getRenu2ecef() const { return _T_enu2ecef.R; }
    Mat3d getRecef2enu() const { return _T_enu2ecef.R.transpose(); }
    Iso3 getTenu2ecef() const { return _T_enu2ecef; }
    Iso3 getTecef2enu() const { return _T_enu2ecef.inverse(); }
    Mat3d getRw2enu() const;
    Mat3d getRenu2w() const { return getRw2enu().transpose(); }
    Iso3 getTw2ecef() const { Iso3 T; T.R = getRw2enu(); return _T_enu2ecef * T; }
    Iso3 getTecef2w() const { Iso3 T; T.R = getRenu2w(); return T * getTecef2enu(); }
    // what GnssUpdate reads (GnssUpdate.h GvioAlignment)
    GvioAlignment alignment() const;
    // GvioAligner.cpp:88-197.  p_w / v_w = epose->valueTrans1() / valueTrans2()
    void batchAlign(const RawGnssEpoch& gnss_meas, const Vec3d& p_w, const Vec3d& v_w, const std::vector<double>& iono);
    void batchAlign(const RawGnssEpoch& gnss_meas, const std::shared_ptr<SE23> epose, const std::vector<double>& iono);
    // gnss_comm::psr_pos (gnss_spp.cpp:148-254) on one set of epochs sharing ONE receiver state; false = no solution
    bool psrPos(const std::vector<const RawGnssEpoch*>& epochs, double xyzt[7]);
    // gnss_comm::dopp_vel (gnss_spp.cpp:284-380): (ecef velocity, clock drift) of one epoch at the reference position ref_ecef
    bool doppVel(const RawGnssEpoch& epoch, const double ref_ecef[3], double vel_ddt[4]);
    void setIono(const std::vector<double>& iono) { _iono_params = iono; }
    int bufferSize() const { return (int)_align_buffer.size(); }
    // diagnostics of the last completed alignment
    double lastRcvDdt() const { return _last_rcv_ddt; }
    const double* lastRoughAnchor() const { return _rough_anchor; }

protected:
    struct Item { Vec3d p, v; RawGnssEpoch meas; };
    bool coarseLocalization(double rough_anchor_ecef[7]);
    bool yawAlignment(const double rough_anchor_ecef[3], double& yaw_offset, double& rcv_ddt);
    bool anchorRefinement(double yaw_offset, double rcv_ddt, const double rough_anchor_ecef[7], double refined_anchor_ecef[7]);
    bool evalEpochs(const std::vector<ingvio_gnss_epoch>& eps, std::vector<double>& rec);
    void reset();

    ingvio_ctx* _ctx;
    int _batch_size, _max_iter;
    double _conv_epsilon, _vel_thres;
    bool _isAligned = false;
    Iso3 _T_enu2ecef;
    double _yaw_offset = 0.0, _last_rcv_ddt = 0.0, _rough_anchor[7] = { 0 };
    std::vector<Item> _align_buffer;
    std::vector<double> _iono_params;
};

}  // namespace ingvio
