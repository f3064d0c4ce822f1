#include "GnssUpdate.h"

#include <cmath>
#include <iostream>
#include <map>

#include "IngvioParams.h"
#include "StateManager.h"

namespace ingvio {

GnssUpdate::GnssUpdate(const IngvioParams& fp)
    : UpdateBase(fp._chi2_max_dof, fp._chi2_thres), _psr_noise_amp(fp._psr_noise_amp), _dopp_noise_amp(fp._dopp_noise_amp),
      _is_gnss_chi2_test(fp._is_gnss_chi2_test), _is_gnss_strong_reject(fp._is_gnss_strong_reject), _is_adjust_yof(fp._is_adjust_yof) {}

int GnssUpdate::updateTrackedSys(std::shared_ptr<State> state, const GnssResiduals& g)
{
    if (!state->_state_params._enable_gnss) return 0;
    const int nsat = (int)g.sys.size();
    if (nsat <= 0) return 0;
    if (state->_gnss.find(State::YOF) == state->_gnss.end() || state->_gnss.find(State::FS) == state->_gnss.end()) return 0;   // checkGnssStates
    if (_is_adjust_yof)
        std::cout << "[GnssUpdate]: is_adjust_yof needs GvioAligner's dotRw2enu, not carried by the shim; YOF column stays 0." << std::endl;

    std::vector<std::shared_ptr<Type>> var_order;
    std::map<std::shared_ptr<Type>, int> local_var_index;
    var_order.push_back(state->_extended_pose); local_var_index[state->_extended_pose] = 0;          // :127-131
    var_order.push_back(state->_gnss.at(State::YOF)); local_var_index[state->_gnss.at(State::YOF)] = 9;

    const int max_possible_rows = 2 * nsat, max_possible_cols = state->_extended_pose->size() + 6;
    MatXd H(max_possible_rows, max_possible_cols);
    VecXd res(max_possible_rows, 0.0), Rd(max_possible_rows, 0.0);
    int row_cnt = 0, col_cnt = 10;
    const Mat3d Sp = skew(state->_extended_pose->valueTrans1()), Sv = skew(state->_extended_pose->valueTrans2());
    const Mat3d RSp = g.R_w2ecef * Sp, RSv = g.R_w2ecef * Sv;

    auto rowGate = [&](const double h9[9], double r_i, double noise, std::shared_ptr<Type> third) {
        MatXd H_i(1, 11);
        for (int c = 0; c < 9; ++c) H_i(0, c) = h9[c];
        H_i(0, 10) = 1.0;
        VecXd res_i(1, r_i);
        std::vector<std::shared_ptr<Type>> sub_order = { state->_extended_pose, state->_gnss.at(State::YOF), third };
        return testChiSquared(state, res_i, H_i, sub_order, noise);
    };

    for (int i = 0; i < nsat; ++i) {                                                                  // :148-211
        auto cb_it = state->_gnss.find(g.sys[i]);
        if (g.sys[i] < 0 || g.sys[i] > 3 || cb_it == state->_gnss.end()) continue;
        auto cb_state = cb_it->second;
        const Vec3d& u = g.unit_rv2sv[i];
        double h9[9] = { 0 };
        for (int c = 0; c < 3; ++c) {
            h9[c] = u[0] * RSp(0, c) + u[1] * RSp(1, c) + u[2] * RSp(2, c);                           // :161
            h9[3 + c] = -(u[0] * g.R_w2ecef(0, c) + u[1] * g.R_w2ecef(1, c) + u[2] * g.R_w2ecef(2, c));   // :162
        }
        double sin_el = g.sin_el[i];
        if (std::fabs(sin_el) < 1e-6) sin_el = 1e-6;
        const double psr_noise = _psr_noise_amp * std::pow(g.ura[i] * g.psr_std[i] / (sin_el * sin_el), 0.5);   // :187
        const double r_i = -g.res_pos[i];
        if (_is_gnss_chi2_test && !rowGate(h9, r_i, psr_noise, cb_state)) continue;                   // :190
        res[row_cnt] = r_i; Rd[row_cnt] = psr_noise * psr_noise;
        for (int c = 0; c < 9; ++c) H(row_cnt, c) = h9[c];
        if (local_var_index.find(cb_state) == local_var_index.end()) {
            local_var_index[cb_state] = col_cnt; col_cnt += cb_state->size(); var_order.push_back(cb_state);
        }
        H(row_cnt, local_var_index.at(cb_state)) = 1.0;
        ++row_cnt;
    }
    auto cs_state = state->_gnss.at(State::FS);                                                       // :213-218
    local_var_index[cs_state] = col_cnt; col_cnt += cs_state->size(); var_order.push_back(cs_state);
    for (int i = 0; i < nsat; ++i) {                                                                  // :220-272
        if (g.sys[i] < 0 || g.sys[i] > 3 || state->_gnss.find(g.sys[i]) == state->_gnss.end()) continue;
        const Vec3d& u = g.unit_rv2sv[i];
        double h9[9] = { 0 };
        for (int c = 0; c < 3; ++c) {
            h9[c] = u[0] * RSv(0, c) + u[1] * RSv(1, c) + u[2] * RSv(2, c);                           // :236
            h9[6 + c] = -(u[0] * g.R_w2ecef(0, c) + u[1] * g.R_w2ecef(1, c) + u[2] * g.R_w2ecef(2, c));   // :237
        }
        double sin_el = g.sin_el[i];
        if (std::fabs(sin_el) < 1e-6) sin_el = 1e-6;
        const double dopp_noise = _dopp_noise_amp * std::pow(g.ura[i] * g.dopp_std_mps[i] / (sin_el * sin_el), 0.5);   // :256
        const double r_i = -g.res_vel[i];
        if (_is_gnss_chi2_test && !rowGate(h9, r_i, dopp_noise, cs_state)) continue;                  // :259
        res[row_cnt] = r_i; Rd[row_cnt] = dopp_noise * dopp_noise;
        for (int c = 0; c < 9; ++c) H(row_cnt, c) = h9[c];
        H(row_cnt, local_var_index.at(cs_state)) = 1.0;
        ++row_cnt;
    }
    if (row_cnt == 0) return 0;
    MatXd Hc(row_cnt, col_cnt), R(row_cnt, row_cnt);                                                  // :274-284
    VecXd rc(res.begin(), res.begin() + row_cnt);
    for (int j = 0; j < col_cnt; ++j) for (int i = 0; i < row_cnt; ++i) Hc(i, j) = H(i, j);
    for (int i = 0; i < row_cnt; ++i) R(i, i) = Rd[i];
    if (row_cnt <= 14 && _is_gnss_strong_reject && !testChiSquared(state, rc, Hc, var_order, R, row_cnt)) return 0;   // :286
    StateManager::ekfUpdate(state, var_order, Hc, rc, R);                                             // :290
    return row_cnt;
}

}  // namespace ingvio
