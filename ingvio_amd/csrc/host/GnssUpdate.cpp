#include "GnssUpdate.h"
#include <algorithm>
#include <cstring>

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
                      int* nvar, bool adjust_yof)
{
    const int nsat = (int)g.sys.size();
    int nv = 0, rows = 0, col_cnt = 10;
    int cb_col[4] = { -1, -1, -1, -1 };
    vidx[nv] = idx_se23; vsize[nv++] = 9;                                                            // :127-131
    vidx[nv] = idx_yof; vsize[nv++] = 1;
    for (int c = 0; c < 15; ++c) for (int i = 0; i < 2 * nsat; ++i) H[i + (size_t)c * ldh] = 0.0;
    const Mat3d RSp = g.R_w2ecef * skew(p_w), RSv = g.R_w2ecef * skew(v_w);
    const bool yof = adjust_yof && g.has_yof_jac;
    const Vec3d dp = yof ? g.dRw2ecef_dyof * p_w : Vec3d(), dv = yof ? g.dRw2ecef_dyof * v_w : Vec3d();      // Renu2ecef dotRw2enu p | v
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
        if (yof) H[rows + (size_t)9 * ldh] = -(u[0] * dp[0] + u[1] * dp[1] + u[2] * dp[2]);                          // :164-167
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
        if (yof) H[rows + (size_t)9 * ldh] = -(u[0] * dv[0] + u[1] * dv[1] + u[2] * dv[2]);                          // :239-242
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
    _last_keep.clear();
    if (nsat <= 0) return 0;
    if (state->_gnss.find(State::YOF) == state->_gnss.end() || state->_gnss.find(State::FS) == state->_gnss.end()) return 0;   // checkGnssStates
    int idx_cb[4] = { -1, -1, -1, -1 };
    for (int s = 0; s < 4; ++s) { auto it = state->_gnss.find(s); if (it != state->_gnss.end()) idx_cb[s] = it->second->idx(); }
    const int ldh = 2 * nsat;
    std::vector<double> H((size_t)ldh * 15), res(ldh), Rd(ldh);
    int vidx[8], vsize[8], nvar = 0;
    const int rows = gnssCandidateRows(g, state->_extended_pose->valueTrans1(), state->_extended_pose->valueTrans2(),
                                       state->_extended_pose->idx(), state->_gnss.at(State::YOF)->idx(), idx_cb,
                                       state->_gnss.at(State::FS)->idx(), _psr_noise_amp, _dopp_noise_amp, H.data(), ldh, res.data(),
                                       Rd.data(), vidx, vsize, &nvar, _is_adjust_yof);
    if (rows == 0) return 0;
    // the gates need table[1] (rows, :190/:259) and table[rows] (block, :286): UpdateBase::testChiSquared extends its table on demand
    const std::vector<double> table = chi2TableDense(rows + 1);
    ingvio_update_block blk;
    blk.vidx = vidx; blk.vsize = vsize; blk.k = nvar; blk.H = H.data(); blk.ldh = ldh; blk.m = rows; blk.res = res.data(); blk.R = Rd.data();
    ingvio_gnss_opts o;
    std::memset(&o, 0, sizeof o);
    o.gate_rows = _is_gnss_chi2_test ? 1 : 0; o.strong_reject = _is_gnss_strong_reject ? 1 : 0;
    o.chi2_table = table.data(); o.chi2_len = (int)table.size();
    ingvio_ctx* ctx = StateManager::ctx(state);
    const int b = StateManager::filterIndex(state);
    VecXd dx(ingvio_ldp(ctx), 0.0);
    int used = 0, status = 0;
    std::vector<int> keep((size_t)ingvio_mld(ctx), 0);
    const int rc = ingvio_gnss_update_batch(ctx, b, 1, &blk, &o, dx.data(), &used, keep.data(), &status);      // gates + ekfUpdate, one round trip
    if (rc < 0) {
        std::cout << "[GnssUpdate]: device update failed (" << rc << "): " << ingvio_last_error(ctx) << std::endl;
        std::exit(EXIT_FAILURE);
    }
    // per candidate row: survived its gate (trace).  After the status check and clamped (more rows than the context's row
    // capacity is exactly the case the device call refuses - the range must not be formed before that is known)
    _last_keep.assign(keep.begin(), keep.begin() + std::min<size_t>((size_t)rows, keep.size()));
    if (status == INGVIO_NEG_DIAG)
        std::cout << "[StateManager]: EKF Update and found negative diag cov elements! " << std::endl;   // StateManager.cpp:418
    if (used == 0 || status == INGVIO_REJECTED) return 0;
    dx.resize(state->curr_cov_size());
    StateManager::boxPlus(state, dx);                                                                // :290 -> StateManager.cpp:425
    return used;
}

// ---- the GNSS epoch path ------------------------------------------------------------------------------------------------
Mat3d dotRw2enu(double yo)               // GnssManager.cpp:101-113
{
    Mat3d D;
    D(0, 0) = -std::sin(yo); D(0, 1) = -std::cos(yo); D(0, 2) = 0.0;
    D(1, 0) = std::cos(yo);  D(1, 1) = -std::sin(yo); D(1, 2) = 0.0;
    D(2, 0) = 0.0;           D(2, 1) = 0.0;           D(2, 2) = 0.0;
    return D;
}

static Mat3d rotZ(double yaw)           // GnssManager::calcRw2enu (GnssManager.cpp:57-60): AngleAxisd(yaw, UnitZ)
{
    Mat3d R = Mat3d::Identity();
    R(0, 0) = std::cos(yaw); R(0, 1) = -std::sin(yaw); R(1, 0) = std::sin(yaw); R(1, 1) = std::cos(yaw);
    return R;
}

void GnssUpdate::checkYofStatus(std::shared_ptr<State> state, const GvioAlignment& aligner)
{
    if (!state->_state_params._enable_gnss || !aligner.isAlign()) return;
    if (state->_gnss.find(State::YOF) == state->_gnss.end())
        StateManager::addGNSSVariable(state, State::YOF, aligner.yaw_offset, state->_state_params._init_cov_yof);      // :40-43
}

GnssResiduals GnssUpdate::residualsAt(std::shared_ptr<State> state, const GnssMeas& gm, const GvioAlignment& al, const double* cb_override,
                                      const double* fs_override)
{
    const double yo = state->_gnss.count(State::YOF) ? state->_gnss.at(State::YOF)->value() : 0.0;
    const Mat3d Rw2enu = rotZ(yo);
    const Vec3d rcv = al.R_enu2ecef * (Rw2enu * state->_extended_pose->valueTrans1()) + al.anchor_ecef;              // :102 getTenu2ecef * calcTw2enu * p
    double xyzt[7] = { rcv[0], rcv[1], rcv[2], 0, 0, 0, 0 };
    for (int s = 0; s < 4; ++s) {                                                                                     // getClockbiasVec
        auto it = state->_gnss.find(s);
        if (it != state->_gnss.end()) xyzt[3 + s] = it->second->value();
        if (cb_override && cb_override[s] == cb_override[s]) xyzt[3 + s] = cb_override[s];
    }
    const Vec3d vel = al.R_enu2ecef * (Rw2enu * state->_extended_pose->valueTrans2());                               // :108
    double dopp[4] = { vel[0], vel[1], vel[2], state->_gnss.count(State::FS) ? state->_gnss.at(State::FS)->value() : 0.0 };
    if (fs_override) dopp[3] = *fs_override;
    GnssResiduals g;
    std::vector<double> az, el;
    gnss::psr_res(xyzt, gm.sats, g.res_pos, g.unit_rv2sv, az, el);
    gnss::dopp_res(dopp, rcv, gm.sats, g.res_vel);
    for (size_t i = 0; i < gm.sats.size(); ++i) {
        g.sys.push_back(gm.sats[i].sys);
        g.sin_el.push_back(std::sin(el[i]));
        g.ura.push_back(gm.sats[i].ura); g.psr_std.push_back(gm.sats[i].psr_std);
        g.dopp_std_mps.push_back(gm.sats[i].dopp_std * gnss::LIGHT_SPEED / gm.sats[i].freq);                      // :253
    }
    g.R_w2ecef = al.R_enu2ecef * Rw2enu;                                                                              // :141
    g.R_enu2ecef = al.R_enu2ecef;
    g.dRw2ecef_dyof = al.R_enu2ecef * dotRw2enu(yo);                          // zero when YOF is not in the state (GnssManager.cpp:103-104)
    if (!state->_gnss.count(State::YOF)) g.dRw2ecef_dyof = Mat3d::Zero();
    g.has_yof_jac = true;
    return g;
}

int GnssUpdate::updateTrackedSys(std::shared_ptr<State> state, const GnssMeas& gm, const GvioAlignment& al)
{
    if (!state->_state_params._enable_gnss || !al.isAlign()) return 0;                                              // :89-90
    if (gm.sats.empty()) return 0;
    if (state->_gnss.find(State::YOF) == state->_gnss.end() || state->_gnss.find(State::FS) == state->_gnss.end()) return 0;   // checkGnssStates
    bool any = false;
    for (int s = 0; s < 4; ++s) any = any || state->_gnss.count(s);
    if (!any) return 0;
    return updateTrackedSys(state, residualsAt(state, gm, al));
}

void GnssUpdate::removeUntrackedSys(std::shared_ptr<State> state, const GnssMeas& gm)
{
    if (!state->_state_params._enable_gnss) return;
    std::unordered_set<int> sys;                                                                                     // getSysInGnssMeas :47-63
    sys.insert(State::YOF);
    for (const auto& o : gm.sats) sys.insert(o.sys);
    if (sys.size() > 1) sys.insert(State::FS);
    std::vector<int> to_marg;
    for (const auto& kv : state->_gnss) if (!sys.count(kv.first)) to_marg.push_back(kv.first);
    for (int g : to_marg) StateManager::margGNSSVariable(state, (State::GNSSType)g);
}

// addNewTrackedSys (:319-469): every clock state the SPP fix knows about and the filter does not yet carry is initialised from
// the rows of its constellation by StateManager::addVariableDelayed (Givens QR, chi^2 on the rows below, EKF update).
int GnssUpdate::addNewTrackedSys(std::shared_ptr<State> state, const GnssMeas& gm, const SppMeas& spp, const GvioAlignment& al)
{
    if (!state->_state_params._enable_gnss || !al.isAlign()) return 0;
    if (gm.sats.empty()) return 0;
    if (state->_gnss.find(State::YOF) == state->_gnss.end()) return 0;
    std::vector<int> to_add;                                                                                         // getSysInSppMeas / calcSysToAdd
    if (std::fabs(spp.velSpp[3]) > 1e-03 && !state->_gnss.count(State::FS)) to_add.push_back(State::FS);
    for (int i = 0; i < 4; ++i) if (std::fabs(spp.posSpp[3 + i]) > 1e-03 && !state->_gnss.count(i)) to_add.push_back(i);
    if (to_add.empty()) return 0;
    const double nan = std::nan("");
    double cb_over[4] = { nan, nan, nan, nan };
    bool add_fs = false;
    for (int g : to_add) { if (g == State::FS) add_fs = true; else cb_over[g] = spp.posSpp[3 + g]; }                // :351-356
    const double fs_val = spp.velSpp[3];
    if (!add_fs && !state->_gnss.count(State::FS)) std::cout << "[GnssUpdate]: Fs is either added or in the state!" << std::endl;
    const GnssResiduals g = residualsAt(state, gm, al, cb_over, add_fs ? &fs_val : nullptr);
    std::vector<std::shared_ptr<Type>> x_order = { state->_extended_pose, state->_gnss.at(State::YOF) };
    int added = 0;
    // Order of the additions: FS first, then the clock biases GPS, GLO, GAL, BDS (the reference iterates an unordered_set, :360/:433,
    // i.e. an implementation-defined order).  As in the reference the residuals and R_w2ecef are evaluated once, before the loop
    // (:357-358), while [p]x and [v]x are read from the state INSIDE it (valueTrans1/2 at :397 / :458): every successful
    // addVariableDelayed ends with an EKF update that moves p and v, so the second and later systems see the moved values.
    for (int gtype : to_add) {
        const bool fs = gtype == State::FS;
        const Vec3d p = state->_extended_pose->valueTrans1(), v = state->_extended_pose->valueTrans2();
        const Mat3d RSp = g.R_w2ecef * skew(p), RSv = g.R_w2ecef * skew(v);
        std::vector<int> rows;
        for (size_t i = 0; i < gm.sats.size(); ++i) if (fs ? (gm.sats[i].sys >= 0 && gm.sats[i].sys <= 3) : gm.sats[i].sys == gtype) rows.push_back((int)i);
        if (rows.empty()) continue;
        const int m = (int)rows.size();
        MatXd Hx(m, 10), Hf(m, 1);
        VecXd res(m, 0.0);
        double avg = 0.0;
        for (int r = 0; r < m; ++r) {
            const int i = rows[r];
            const Vec3d& u = g.unit_rv2sv[i];
            const Mat3d& RS = fs ? RSv : RSp;
            for (int c = 0; c < 3; ++c) {
                Hx(r, c) = u[0] * RS(0, c) + u[1] * RS(1, c) + u[2] * RS(2, c);                                     // :389 / :440
                Hx(r, (fs ? 6 : 3) + c) = -(u[0] * g.R_w2ecef(0, c) + u[1] * g.R_w2ecef(1, c) + u[2] * g.R_w2ecef(2, c));
            }
            // is_adjust_yof, AS WRITTEN (:401-404 / :462-465): -u^T getRecef2enu() dotRw2enu(state) x with x = v in the FS branch and
            // p in the clock-bias branches - note Recef2enu here, where updateTrackedSys (:166, :241) has Renu2ecef (quirk Q14)
            Hx(r, 9) = 0.0;
            if (_is_adjust_yof) {
                const double yo = state->_gnss.at(State::YOF)->value();
                const Vec3d w = g.R_enu2ecef.transpose() * (dotRw2enu(yo) * (fs ? v : p));
                Hx(r, 9) = -(u[0] * w[0] + u[1] * w[1] + u[2] * w[2]);
            }
            Hf(r, 0) = 1.0;
            res[r] = -(fs ? g.res_vel[i] : g.res_pos[i]);
            double se = g.sin_el[i];
            if (std::fabs(se) < 1e-6) se = 1e-6;
            avg += g.ura[i] * (fs ? g.dopp_std_mps[i] : g.psr_std[i]) / (se * se);                                   // :411-418 / :507-514
        }
        const double noise = (fs ? _dopp_noise_amp : _psr_noise_amp) * std::sqrt(avg / m);
        auto var = std::make_shared<Scalar>();
        var->setValue(fs ? spp.velSpp[3] : spp.posSpp[3 + gtype]);
        if (!StateManager::addVariableDelayed(state, var, x_order, Hx, Hf, res, noise, 0.95, true)) continue;      // :421 / :463
        state->_gnss[gtype] = var;
        ++added;
    }
    return added;
}

}  // namespace ingvio
