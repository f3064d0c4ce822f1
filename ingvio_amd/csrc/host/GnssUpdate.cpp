#include "GnssUpdate.h"

#include <cmath>
#include <iostream>

#include "IngvioParams.h"
#include "StateManager.h"

namespace ingvio {

GnssUpdate::GnssUpdate(const IngvioParams& fp)
    : UpdateBase(fp._chi2_max_dof, fp._chi2_thres), _psr_noise_amp(fp._psr_noise_amp), _dopp_noise_amp(fp._dopp_noise_amp),
      _is_gnss_chi2_test(fp._is_gnss_chi2_test), _is_gnss_strong_reject(fp._is_gnss_strong_reject), _is_adjust_yof(fp._is_adjust_yof) {}

// Every row the reference would consider (GnssUpdate.cpp:148-272), in its order: pseudo-range rows, then Doppler rows, over
// var_order = [SE23 (9), YOF (1), clock bias of each constellation in order of first appearance, FS].  The per-row gates
// (:190,:259), the compaction and the block gate (:286) run on the device (ingvio_gnss_update_batch).
int gnssCandidateRows(const GnssResiduals& g, const Vec3d& p_w, const Vec3d& v_w, int idx_se23, int idx_yof, const int idx_cb[4],
                      int idx_fs, double psr_amp, double dopp_amp, double* H, int ldh, double* res, double* Rd, int* vidx, int* vsize,
                      int* nvar)
{
    const int nsat = (int)g.sys.size();
    int nv = 0, rows = 0, col_cnt = 10;
    int cb_col[4] = { -1, -1, -1, -1 };
    vidx[nv] = idx_se23; vsize[nv++] = 9;                                                            // :127-131
    vidx[nv] = idx_yof; vsize[nv++] = 1;
    for (int c = 0; c < 15; ++c) for (int i = 0; i < 2 * nsat; ++i) H[i + (size_t)c * ldh] = 0.0;
    const Mat3d RSp = g.R_w2ecef * skew(p_w), RSv = g.R_w2ecef * skew(v_w);
    auto usable = [&](int i) { return g.sys[i] >= 0 && g.sys[i] <= 3 && idx_cb[g.sys[i]] >= 0; };     // :153
    for (int i = 0; i < nsat; ++i) {                                                                  // :148-211
        if (!usable(i)) continue;
        const Vec3d& u = g.unit_rv2sv[i];
        for (int c = 0; c < 3; ++c) {
            H[rows + (size_t)c * ldh] = u[0] * RSp(0, c) + u[1] * RSp(1, c) + u[2] * RSp(2, c);                           // :161
            H[rows + (size_t)(3 + c) * ldh] = -(u[0] * g.R_w2ecef(0, c) + u[1] * g.R_w2ecef(1, c) + u[2] * g.R_w2ecef(2, c));   // :162
        }
        double sin_el = g.sin_el[i];
        if (std::fabs(sin_el) < 1e-6) sin_el = 1e-6;
        const double psr_noise = psr_amp * std::pow(g.ura[i] * g.psr_std[i] / (sin_el * sin_el), 0.5);   // :187
        res[rows] = -g.res_pos[i]; Rd[rows] = psr_noise * psr_noise;
        const int s = g.sys[i];
        if (cb_col[s] < 0) { cb_col[s] = col_cnt++; vidx[nv] = idx_cb[s]; vsize[nv++] = 1; }          // :200-206
        H[rows + (size_t)cb_col[s] * ldh] = 1.0;
        ++rows;
    }
    const int fs_col = col_cnt++;                                                                      // :213-218
    vidx[nv] = idx_fs; vsize[nv++] = 1;
    for (int i = 0; i < nsat; ++i) {                                                                  // :220-272
        if (!usable(i)) continue;
        const Vec3d& u = g.unit_rv2sv[i];
        for (int c = 0; c < 3; ++c) {
            H[rows + (size_t)c * ldh] = u[0] * RSv(0, c) + u[1] * RSv(1, c) + u[2] * RSv(2, c);                           // :236
            H[rows + (size_t)(6 + c) * ldh] = -(u[0] * g.R_w2ecef(0, c) + u[1] * g.R_w2ecef(1, c) + u[2] * g.R_w2ecef(2, c));   // :237
        }
        double sin_el = g.sin_el[i];
        if (std::fabs(sin_el) < 1e-6) sin_el = 1e-6;
        const double dopp_noise = dopp_amp * std::pow(g.ura[i] * g.dopp_std_mps[i] / (sin_el * sin_el), 0.5);   // :256
        res[rows] = -g.res_vel[i]; Rd[rows] = dopp_noise * dopp_noise;
        H[rows + (size_t)fs_col * ldh] = 1.0;
        ++rows;
    }
    *nvar = nv;
    return rows;
}

int GnssUpdate::updateTrackedSys(std::shared_ptr<State> state, const GnssResiduals& g)
{
    if (!state->_state_params._enable_gnss) return 0;
    const int nsat = (int)g.sys.size();
    if (nsat <= 0) return 0;
    if (state->_gnss.find(State::YOF) == state->_gnss.end() || state->_gnss.find(State::FS) == state->_gnss.end()) return 0;   // checkGnssStates
    if (_is_adjust_yof && !_warned_yof) {
        std::cout << "[GnssUpdate]: is_adjust_yof needs GvioAligner's dotRw2enu, not carried by the shim; YOF column stays 0." << std::endl;
        _warned_yof = true;
    }
    int idx_cb[4] = { -1, -1, -1, -1 };
    for (int s = 0; s < 4; ++s) { auto it = state->_gnss.find(s); if (it != state->_gnss.end()) idx_cb[s] = it->second->idx(); }
    const int ldh = 2 * nsat;
    std::vector<double> H((size_t)ldh * 15), res(ldh), Rd(ldh);
    int vidx[8], vsize[8], nvar = 0;
    const int rows = gnssCandidateRows(g, state->_extended_pose->valueTrans1(), state->_extended_pose->valueTrans2(),
                                       state->_extended_pose->idx(), state->_gnss.at(State::YOF)->idx(), idx_cb,
                                       state->_gnss.at(State::FS)->idx(), _psr_noise_amp, _dopp_noise_amp, H.data(), ldh, res.data(),
                                       Rd.data(), vidx, vsize, &nvar);
    if (rows == 0) return 0;
    // the gates need table[1] (rows, :190/:259) and table[rows] (block, :286): UpdateBase::testChiSquared extends its table on demand
    const std::vector<double> table = chi2TableDense(rows + 1);
    ingvio_update_block blk;
    blk.vidx = vidx; blk.vsize = vsize; blk.k = nvar; blk.H = H.data(); blk.ldh = ldh; blk.m = rows; blk.res = res.data(); blk.R = Rd.data();
    ingvio_gnss_opts o;
    o.gate_rows = _is_gnss_chi2_test ? 1 : 0; o.strong_reject = _is_gnss_strong_reject ? 1 : 0;
    o.chi2_table = table.data(); o.chi2_len = (int)table.size();
    ingvio_ctx* ctx = StateManager::ctx(state);
    const int b = StateManager::filterIndex(state);
    VecXd dx(ingvio_ldp(ctx), 0.0);
    int used = 0, status = 0;
    const int rc = ingvio_gnss_update_batch(ctx, b, 1, &blk, &o, dx.data(), &used, nullptr, &status);      // gates + ekfUpdate, one round trip
    if (rc < 0) {
        std::cout << "[GnssUpdate]: device update failed (" << rc << "): " << ingvio_last_error(ctx) << std::endl;
        std::exit(EXIT_FAILURE);
    }
    if (status == INGVIO_NEG_DIAG)
        std::cout << "[StateManager]: EKF Update and found negative diag cov elements! " << std::endl;   // StateManager.cpp:418
    if (used == 0 || status == INGVIO_REJECTED) return 0;
    dx.resize(state->curr_cov_size());
    StateManager::boxPlus(state, dx);                                                                // :290 -> StateManager.cpp:425
    return used;
}

}  // namespace ingvio
