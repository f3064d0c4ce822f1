// GnssUpdate.h — the covariance slice of ingvio_estimator/src/GnssUpdate.cpp:84-293
// (GnssUpdate::updateTrackedSys): H assembly for pseudo-range / Doppler rows, the per-row and block
// chi^2 gates and ekfUpdate with a diagonal R.  Its inputs are the OUTPUTS of gnss_comm's
// psr_res / dopp_res / sat_states (line-of-sight units, residuals, elevation, ura/std), which stay
// out of scope.
#pragma once
#include <memory>
#include <vector>

#include "Mat3.h"
#include "Update.h"

namespace ingvio {

class IngvioParams;
class State;

struct GnssResiduals {
    // one entry per satellite, in gnss_meas order
    std::vector<Vec3d> unit_rv2sv;        // = -J_pos_ecef.row(i).head<3>()  (GnssUpdate.cpp:150)
    std::vector<int> sys;                 // gnss_comm::sys2idx: GPS 0, GLO 1, GAL 2, BDS 3
    std::vector<double> res_pos, res_vel; // psr_res / dopp_res outputs
    std::vector<double> sin_el;           // sin(all_sv_azel[i].y())
    std::vector<double> ura, psr_std;     // ephem ura, obs psr_std[l1]
    std::vector<double> dopp_std_mps;     // dopp_std[l1] * LIGHT_SPEED / L1 frequency
    Mat3d R_w2ecef;                       // getRenu2ecef() * calcRw2enu(state)
};

// candidate rows of updateTrackedSys (all pseudo-range rows, then all Doppler rows; H column-major, ldh >= 2 nsat, 15 columns;
// vidx / vsize hold up to 7 variables); returns the row count
int gnssCandidateRows(const GnssResiduals& g, const Vec3d& p_w, const Vec3d& v_w, int idx_se23, int idx_yof, const int idx_cb[4],
                      int idx_fs, double psr_amp, double dopp_amp, double* H, int ldh, double* res, double* Rd, int* vidx, int* vsize,
                      int* nvar);

class GnssUpdate : public UpdateBase {
public:
    GnssUpdate(const IngvioParams& filter_params);
    // returns rows handed to ekfUpdate (0: nothing done)
    int updateTrackedSys(std::shared_ptr<State> state, const GnssResiduals& g);

protected:
    double _psr_noise_amp, _dopp_noise_amp;
    bool _is_gnss_chi2_test, _is_gnss_strong_reject, _is_adjust_yof;
    bool _warned_yof = false;
};

}  // namespace ingvio
