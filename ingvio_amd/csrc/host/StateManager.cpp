#include "StateManager.h"
#include "Update.h"

#include <algorithm>
#include <cstdlib>
#include <iostream>

namespace ingvio {

void StateManager::fatal(const std::shared_ptr<State>& state, const char* what, int rc)
{
    std::cout << "[StateManager]: " << what << " failed on the device (" << rc << "): " << ingvio_last_error(state->_ctx) << std::endl;
    std::exit(EXIT_FAILURE);
}

bool StateManager::checkStateContinuity(const std::shared_ptr<State> state)
{
    int idx = 0;
    for (size_t i = 0; i < state->_err_variables.size(); ++i)
        if (state->_err_variables[i]->idx() == idx) idx += state->_err_variables[i]->size();
        else return false;
    return state->curr_cov_size() == idx;
}

static void gnssIdx(const std::shared_ptr<State>& state, int gi[5])
{
    for (int g = 0; g < 5; ++g) {
        auto it = state->_gnss.find(g);
        gi[g] = it == state->_gnss.end() ? -1 : it->second->idx();
    }
}

void StateManager::propagateStateCov(std::shared_ptr<State> state, const double Phi_imu[225], const double G_imu[180], double dt)
{
    propagateStateCovFused(state, 1, Phi_imu, G_imu, &dt);
}

void StateManager::propagateStateCovFused(std::shared_ptr<State> state, int k, const double* Phi, const double* G, const double* dt)
{
    const StateParams& sp = state->_state_params;
    const double sigma[4] = { sp._noise_g, sp._noise_a, sp._noise_bg, sp._noise_ba };
    int gi[5];
    gnssIdx(state, gi);
    const int rc = ingvio_propagate_fused(state->_ctx, state->_b, 1, k, Phi, G, dt, sigma, sp._enable_gnss ? 1 : 0, gi,
                                          sp._noise_clockbias, sp._noise_cb_rw);
    if (rc != INGVIO_OK) fatal(state, "propagateStateCov", rc);
}

MatXd StateManager::getFullCov(std::shared_ptr<State> state)
{
    const int n = state->curr_cov_size();
    MatXd cov(n, n);
    const int rc = ingvio_cov_get(state->_ctx, state->_b, cov.data(), n);
    if (rc != INGVIO_OK) fatal(state, "getFullCov", rc);
    return cov;
}

static void orderOf(const std::vector<std::shared_ptr<Type>>& vars, std::vector<int>& vidx, std::vector<int>& vsize)
{
    vidx.clear(); vsize.clear();
    for (const auto& v : vars) { vidx.push_back(v->idx()); vsize.push_back(v->size()); }
}

MatXd StateManager::getMarginalCov(std::shared_ptr<State> state, const std::vector<std::shared_ptr<Type>>& small_variables)
{
    std::vector<int> vidx, vsize;
    orderOf(small_variables, vidx, vsize);
    const int ns = calcSubVarSize(small_variables);
    MatXd small_cov(ns, ns);
    const int rc = ingvio_cov_get_marginal(state->_ctx, state->_b, vidx.data(), vsize.data(), (int)vidx.size(), small_cov.data());
    if (rc != INGVIO_OK) fatal(state, "getMarginalCov", rc);
    return small_cov;
}

void StateManager::marginalize(std::shared_ptr<State> state, std::shared_ptr<Type> marg)
{
    if (std::find(state->_err_variables.begin(), state->_err_variables.end(), marg) == state->_err_variables.end()) {
        std::cout << "[StateManager]: Marg is not in the current state!" << std::endl;      // :157-161
        std::exit(EXIT_FAILURE);
    }
    const int marg_size = marg->size(), marg_start = marg->idx();
    const int rc = ingvio_marginalize(state->_ctx, state->_b, 1, &marg_start, marg_size);
    if (rc != INGVIO_OK) fatal(state, "marginalize", rc);
    std::vector<std::shared_ptr<Type>> remaining_variables;                                 // :179-191
    for (size_t i = 0; i < state->_err_variables.size(); ++i)
        if (state->_err_variables[i] != marg) {
            if (state->_err_variables[i]->idx() > marg_start)
                state->_err_variables[i]->set_cov_idx(state->_err_variables[i]->idx() - marg_size);
            remaining_variables.push_back(state->_err_variables[i]);
        }
    marg->set_cov_idx(-1);
    state->_err_variables = remaining_variables;
}

void StateManager::addVariableIndependent(std::shared_ptr<State> state, std::shared_ptr<Type> new_state, const MatXd& blk)
{
    int new_idx = -1;
    const int rc = ingvio_append_independent(state->_ctx, state->_b, 1, new_state->size(), blk.data(), &new_idx);
    if (rc != INGVIO_OK) fatal(state, "addVariableIndependent", rc);
    new_state->set_cov_idx(new_idx);                                                        // :208 (== old_cov_size)
    state->_err_variables.push_back(new_state);
}

void StateManager::addGNSSVariable(std::shared_ptr<State> state, const State::GNSSType& gtype, double value, double cov)
{
    if (state->_gnss.find(gtype) != state->_gnss.end())
        std::cout << "[StateManager]: GNSS variable already in the state, adding operation will rewrite such var!" << std::endl;
    state->_gnss[gtype] = std::make_shared<Scalar>();
    state->_gnss[gtype]->setValue(value);
    MatXd scalar_cov(1, 1);
    scalar_cov(0, 0) = cov;
    addVariableIndependent(state, state->_gnss[gtype], scalar_cov);
}

void StateManager::margGNSSVariable(std::shared_ptr<State> state, const State::GNSSType& gtype)
{
    if (state->_gnss.find(gtype) == state->_gnss.end())
        std::cout << "[StateManager]: GNSS variable not in the state, no need to marg!" << std::endl;
    marginalize(state, state->_gnss.at(gtype));
    state->_gnss.erase(gtype);
}

void StateManager::boxPlus(std::shared_ptr<State> state, const VecXd& dx)
{
    for (size_t i = 0; i < state->_err_variables.size(); ++i) state->_err_variables[i]->update(dx);
}

void StateManager::augmentSlidingWindowPose(std::shared_ptr<State> state)
{
    if (state->_sw_camleft_poses.find(state->_timestamp) != state->_sw_camleft_poses.end()) {
        std::cout << "[StateManager]: Curr pose already in the sw, cannot clone!" << std::endl;
        return;
    }
    std::shared_ptr<SE3> clone(new SE3());
    const Mat3d& R_i2w = state->_extended_pose->valueLinearAsMat();
    const Mat3d Rc = R_i2w * state->_camleft_imu_extrinsics->valueLinearAsMat();           // T_i2w * T_cl2i, :263-272
    const Vec3d pc = R_i2w * state->_camleft_imu_extrinsics->valueTrans() + state->_extended_pose->valueTrans1();
    clone->setValue(Rc, pc);
    clone->setFej(Rc, pc);
    int new_idx = -1;
    const int rc = ingvio_augment_clone(state->_ctx, state->_b, 1, R_i2w.m, &new_idx);
    if (rc != INGVIO_OK) fatal(state, "augmentSlidingWindowPose", rc);
    clone->set_cov_idx(new_idx);                                                            // :274
    state->_sw_camleft_poses[state->_timestamp] = clone;
    state->_err_variables.push_back(clone);
}

void StateManager::addAnchoredLandmarkInState(std::shared_ptr<State> state, std::shared_ptr<AnchoredLandmark> lm, int lm_id, const MatXd& cov)
{
    if (state->_anchored_landmarks.find(lm_id) != state->_anchored_landmarks.end()) {
        std::cout << "[StateManager]: Landmark already in the state, cannot add!" << std::endl;
        return;
    }
    state->_anchored_landmarks[lm_id] = lm;
    addVariableIndependent(state, lm, cov);
}

void StateManager::margSlidingWindowPose(std::shared_ptr<State> state, double marg_time)
{
    if (state->_sw_camleft_poses.find(marg_time) == state->_sw_camleft_poses.end())
        std::cout << "[StateManager]: Marg pose time not exists! Cannot marg!" << std::endl;
    marginalize(state, state->_sw_camleft_poses.at(marg_time));
    state->_sw_camleft_poses.erase(marg_time);
}

void StateManager::margSlidingWindowPose(std::shared_ptr<State> state)
{
    const double marg_time = state->nextMargTime();
    if (marg_time == INFINITY) {
        std::cout << "[StateManager]: Auto marg pose gives inf time! Cannot marg!" << std::endl;
        return;
    }
    margSlidingWindowPose(state, marg_time);
}

void StateManager::margAnchoredLandmarkInState(std::shared_ptr<State> state, int lm_id)
{
    if (state->_anchored_landmarks.find(lm_id) == state->_anchored_landmarks.end()) {
        std::cout << "[StateManager]: Landmark id not exists in state! Cannot marg!" << std::endl;
        return;
    }
    marginalize(state, state->_anchored_landmarks.at(lm_id));
    state->_anchored_landmarks.erase(lm_id);
}

// classifies R as scalar*I / diagonal / full so the device can skip the dense noise block
static int classifyR(const MatXd& R, std::vector<double>& out)
{
    const int m = R.rows();
    bool diag = true, scalar = true;
    for (int j = 0; j < m && diag; ++j)
        for (int i = 0; i < m; ++i)
            if (i != j && R(i, j) != 0.0) { diag = false; break; }
    if (diag) for (int i = 1; i < m; ++i) if (R(i, i) != R(0, 0)) { scalar = false; break; }
    if (diag && scalar && m > 1) { out.assign(1, R(0, 0)); return INGVIO_R_SCALAR; }
    if (diag) { out.resize(m); for (int i = 0; i < m; ++i) out[i] = R(i, i); return INGVIO_R_DIAG; }
    out.assign(R.data(), R.data() + (size_t)m * m);
    return INGVIO_R_FULL;
}

void StateManager::ekfUpdate(std::shared_ptr<State> state, const std::vector<std::shared_ptr<Type>>& var_order,
                             const MatXd& H, const VecXd& res, const MatXd& R)
{
    if (!checkSubOrder(state, var_order)) std::exit(EXIT_FAILURE);      // assert(checkSubOrder), :369
    std::vector<int> vidx, vsize;
    orderOf(var_order, vidx, vsize);
    std::vector<double> Rv;
    const int kind = classifyR(R, Rv);
    VecXd dx(state->curr_cov_size(), 0.0);
    const int rc = ingvio_ekf_update(state->_ctx, state->_b, vidx.data(), vsize.data(), (int)vidx.size(), H.data(), H.rows(),
                                     H.rows(), res.data(), Rv.data(), kind, dx.data());
    if (rc < 0) fatal(state, "ekfUpdate", rc);
    if (rc == INGVIO_NEG_DIAG)
        std::cout << "[StateManager]: EKF Update and found negative diag cov elements! " << std::endl;      // :418
    boxPlus(state, dx);                                                                      // :425
}

int StateManager::landmarkUpdate(std::shared_ptr<State> state, const ingvio_landmark_frame& frame, const ingvio_landmark_opts& opts,
                                 std::vector<int>* accept)
{
    int rc = ingvio_landmark_stage(state->_ctx, state->_b, 1, &frame, &opts);
    if (rc < 0) fatal(state, "landmarkUpdate (stage)", rc);
    rc = ingvio_landmark_run(state->_ctx, state->_b, 1);
    if (rc < 0) fatal(state, "landmarkUpdate (run)", rc);
    const int ldp = ingvio_ldp(state->_ctx);
    std::vector<double> dxl((size_t)ldp, 0.0);
    std::vector<int> acc(INGVIO_LM_MAX, 0);
    int rows = 0, status = 0;
    rc = ingvio_landmark_fetch(state->_ctx, state->_b, 1, dxl.data(), &rows, acc.data(), nullptr, &status);
    if (rc < 0) fatal(state, "landmarkUpdate (fetch)", rc);
    if (status < 0) fatal(state, "landmarkUpdate", status);
    if (accept) accept->assign(acc.begin(), acc.begin() + frame.n_lm);
    if (rows == 0) return 0;
    if (status == INGVIO_NEG_DIAG)
        std::cout << "[StateManager]: EKF Update and found negative diag cov elements! " << std::endl;      // StateManager.cpp:418
    VecXd dx(state->curr_cov_size(), 0.0);
    for (int i = 0; i < state->curr_cov_size(); ++i) dx[i] = dxl[i];
    boxPlus(state, dx);
    return rows;
}

void StateManager::addVariableDelayedInvertible(std::shared_ptr<State> state, std::shared_ptr<Type> var_new,
                                                const std::vector<std::shared_ptr<Type>>& var_old_order, const MatXd& H_old,
                                                const MatXd& H_new, const VecXd& res, double noise_iso_meas)
{
    if (std::find(state->_err_variables.begin(), state->_err_variables.end(), var_new) != state->_err_variables.end()) {
        std::cout << "[StateManager]: New var already in state! Cannot perform add var delayed inv!" << std::endl;
        return;
    }
    if (!checkSubOrder(state, var_old_order)) std::exit(EXIT_FAILURE);                       // assert, :473
    if ((int)res.size() != H_old.rows() || H_new.rows() != H_new.cols() || H_new.cols() != var_new->size() ||
        H_old.cols() != calcSubVarSize(var_old_order) || H_new.rows() != H_old.rows()) {      // asserts :477-481
        std::cout << "[StateManager]: add var delayed inv: inconsistent sizes!" << std::endl;
        std::exit(EXIT_FAILURE);
    }
    std::vector<int> vidx, vsize;
    orderOf(var_old_order, vidx, vsize);
    int new_idx = -1;
    const int rc = ingvio_add_variable_delayed_invertible(state->_ctx, state->_b, vidx.data(), vsize.data(), (int)vidx.size(),
                                                          H_old.data(), H_old.rows(), H_new.data(), H_new.rows(), H_new.rows(),
                                                          noise_iso_meas, &new_idx);
    if (rc != INGVIO_OK) fatal(state, "addVariableDelayedInvertible", rc);
    var_new->set_cov_idx(new_idx);                                                           // :536
    state->_err_variables.push_back(var_new);
}

bool StateManager::addVariableDelayed(std::shared_ptr<State> state, std::shared_ptr<Type> var_new,
                                      const std::vector<std::shared_ptr<Type>>& var_old_order, const MatXd& H_old, const MatXd& H_new,
                                      const VecXd& res, double noise_iso_meas, double chi2_mult_factor, bool do_chi2)
{
    if (std::find(state->_err_variables.begin(), state->_err_variables.end(), var_new) != state->_err_variables.end()) {
        std::cout << "[StateManager]: New var already in state! Cannot perform add var delayed inv!" << std::endl;
        return false;
    }
    if (!checkSubOrder(state, var_old_order)) std::exit(EXIT_FAILURE);                       // assert, :564
    if ((int)res.size() != H_old.rows() || (int)res.size() != H_new.rows() || H_new.cols() != var_new->size() ||
        H_old.cols() != calcSubVarSize(var_old_order)) {                                     // asserts :566-570
        std::cout << "[StateManager]: add var delayed: inconsistent sizes!" << std::endl;
        std::exit(EXIT_FAILURE);
    }
    if (H_new.rows() <= H_new.cols()) {
        std::cout << "[StateManager]: H_new rows should be larger than H_new cols!" << std::endl;      // :571-575
        return false;
    }
    std::vector<int> vidx, vsize;
    orderOf(var_old_order, vidx, vsize);
    const double chi2_check = chi2Quantile((int)res.size(), 0.95);                           // boost quantile, :610-612
    VecXd dx(state->curr_cov_size() + var_new->size(), 0.0);
    int added = 0, new_idx = -1;
    const int rc = ingvio_add_variable_delayed(state->_ctx, state->_b, vidx.data(), vsize.data(), (int)vidx.size(), H_old.data(),
                                               H_old.rows(), H_new.data(), H_new.rows(), H_new.rows(), H_new.cols(), res.data(),
                                               noise_iso_meas, chi2_mult_factor, do_chi2 ? 1 : 0, chi2_check, dx.data(), &added,
                                               &new_idx, nullptr);
    if (rc < 0) fatal(state, "addVariableDelayed", rc);
    if (!added) {
        std::cout << "[StateManager]: Cannot add variable due to chi2 test failure!" << std::endl;     // :616
        return false;
    }
    var_new->set_cov_idx(new_idx);
    state->_err_variables.push_back(var_new);
    if (rc == INGVIO_NEG_DIAG)
        std::cout << "[StateManager]: EKF Update and found negative diag cov elements! " << std::endl;
    boxPlus(state, dx);                                                                      // the ekfUpdate of :623-624 ends with boxPlus
    return true;
}

void StateManager::replaceVarLinear(std::shared_ptr<State> state, const std::shared_ptr<Type> target_var,
                                    const std::vector<std::shared_ptr<Type>>& dependence_order, const MatXd& H)
{
    if (std::find(state->_err_variables.begin(), state->_err_variables.end(), target_var) == state->_err_variables.end()) {
        std::cout << "[StateManager]: Target var not in state, cannot linearly replace!" << std::endl;  // :653-657
        return;
    }
    if (!checkSubOrder(state, dependence_order)) std::exit(EXIT_FAILURE);
    if (target_var->size() != H.rows() || calcSubVarSize(dependence_order) != H.cols()) {
        std::cout << "[StateManager]: replace var linear: inconsistent sizes!" << std::endl;
        std::exit(EXIT_FAILURE);
    }
    std::vector<int> vidx, vsize;
    orderOf(dependence_order, vidx, vsize);
    const int rc = ingvio_replace_var_linear(state->_ctx, state->_b, target_var->idx(), target_var->size(), vidx.data(), vsize.data(),
                                             (int)vidx.size(), H.data(), H.rows());
    if (rc != INGVIO_OK) fatal(state, "replaceVarLinear", rc);
}

bool StateManager::checkSubOrder(std::shared_ptr<State> state, const std::vector<std::shared_ptr<Type>>& sub_order)
{
    for (const auto& item : sub_order)
        if (std::find(state->_err_variables.begin(), state->_err_variables.end(), item) == state->_err_variables.end()) {
            std::cout << "[StateManager]: Existing sub order var not in state! " << std::endl;
            return false;
        }
    return true;
}

int StateManager::calcSubVarSize(const std::vector<std::shared_ptr<Type>>& sub_var)
{
    int total_size = 0;
    for (const auto& item : sub_var) if (item != nullptr) total_size += item->size();
    return total_size;
}

double StateManager::whitenResidual(std::shared_ptr<State> state, const VecXd& res, const MatXd& H,
                                    const std::vector<std::shared_ptr<Type>>& var_order, const MatXd& R)
{
    std::vector<int> vidx, vsize;
    orderOf(var_order, vidx, vsize);
    std::vector<double> Rv;
    const int kind = classifyR(R, Rv);
    double gamma = 0.0;
    const int rc = ingvio_chi2_gamma(state->_ctx, state->_b, vidx.data(), vsize.data(), (int)vidx.size(), H.data(), H.rows(),
                                     H.rows(), res.data(), Rv.data(), kind, &gamma);
    if (rc != INGVIO_OK) fatal(state, "whitenResidual", rc);
    return gamma;
}

double StateManager::whitenResidual(std::shared_ptr<State> state, const VecXd& res, const MatXd& H,
                                    const std::vector<std::shared_ptr<Type>>& var_order, double noise)
{
    std::vector<int> vidx, vsize;
    orderOf(var_order, vidx, vsize);
    const double var = noise * noise;                                                        // Update.cpp:53
    double gamma = 0.0;
    const int rc = ingvio_chi2_gamma(state->_ctx, state->_b, vidx.data(), vsize.data(), (int)vidx.size(), H.data(), H.rows(),
                                     H.rows(), res.data(), &var, INGVIO_R_SCALAR, &gamma);
    if (rc != INGVIO_OK) fatal(state, "whitenResidual", rc);
    return gamma;
}

std::vector<double> StateManager::whitenResidualMulti(std::shared_ptr<State> state, const std::vector<GateBlock>& blocks, double noise)
{
    const int nb = (int)blocks.size();
    std::vector<double> gamma(nb, 0.0);
    if (nb == 0) return gamma;
    std::vector<std::vector<int>> vidx(nb), vsize(nb);
    std::vector<ingvio_gate_block> gb(nb);
    for (int g = 0; g < nb; ++g) {
        orderOf(blocks[g].var_order, vidx[g], vsize[g]);
        gb[g].vidx = vidx[g].data(); gb[g].vsize = vsize[g].data(); gb[g].k = (int)vidx[g].size();
        gb[g].H = blocks[g].H->data(); gb[g].ldh = blocks[g].H->rows(); gb[g].m = blocks[g].H->rows(); gb[g].res = blocks[g].res->data();
    }
    const int rc = ingvio_chi2_gamma_multi(state->_ctx, state->_b, nb, gb.data(), noise * noise, gamma.data());
    if (rc != INGVIO_OK) fatal(state, "whitenResidualMulti", rc);
    return gamma;
}

bool StateManager::triangulateOne(std::shared_ptr<State> state, const ingvio_msckf_frame& frame, const ingvio_tri_opts& opts, Vec3d& pf)
{
    const int fm = std::max(ingvio_f_max(state->_ctx), 1);
    std::vector<double> pfo(3 * (size_t)fm, 0.0);
    std::vector<int> ok((size_t)fm, 0);
    const int rc = ingvio_triangulate(state->_ctx, state->_b, 1, &frame, &opts, pfo.data(), ok.data());
    if (rc < 0) fatal(state, "triangulate", rc);
    pf = Vec3d(pfo[0], pfo[1], pfo[2]);
    return ok[0] != 0;
}

void StateManager::triangulateFrame(std::shared_ptr<State> state, const ingvio_msckf_frame& frame, const ingvio_tri_opts& opts,
                                    std::vector<Vec3d>& pf, std::vector<char>& ok)
{
    const int fm = std::max(ingvio_f_max(state->_ctx), 1);
    std::vector<double> pfo(3 * (size_t)fm, 0.0);
    std::vector<int> oko((size_t)fm, 0);
    const int rc = ingvio_triangulate(state->_ctx, state->_b, 1, &frame, &opts, pfo.data(), oko.data());
    if (rc < 0) fatal(state, "triangulate", rc);
    pf.resize((size_t)frame.n_feat); ok.resize((size_t)frame.n_feat);
    for (int j = 0; j < frame.n_feat; ++j) { pf[j] = Vec3d(pfo[3 * j], pfo[3 * j + 1], pfo[3 * j + 2]); ok[j] = oko[j] != 0; }
}

int StateManager::msckfUpdateTri(std::shared_ptr<State> state, const ingvio_msckf_frame& frame, const ingvio_msckf_opts& opts,
                                 const ingvio_tri_opts& tri, std::vector<int>* accepted, std::vector<int>* tri_ok,
                                 const unsigned long long* tri_mask, std::vector<Vec3d>* pf)
{
    const int ldp = ingvio_ldp(state->_ctx), fm = std::max(ingvio_f_max(state->_ctx), frame.n_feat);
    VecXd dx(ldp, 0.0);
    std::vector<int> acc(fm, 0), tok(fm, 0);
    std::vector<double> pfo(pf ? 3 * (size_t)fm : 0, 0.0);
    int rows = 0;
    const unsigned long long* masks[1] = { tri_mask };
    const int rc = ingvio_msckf_update_tri(state->_ctx, state->_b, 1, &frame, &opts, &tri, tri_mask ? masks : nullptr, dx.data(), acc.data(), nullptr,
                                           &rows, pf ? pfo.data() : nullptr, tok.data());
    if (rc < 0) fatal(state, "msckfUpdateTri", rc);
    if (accepted) accepted->assign(acc.begin(), acc.begin() + frame.n_feat);
    if (tri_ok) tri_ok->assign(tok.begin(), tok.begin() + frame.n_feat);
    if (pf) { pf->resize((size_t)frame.n_feat); for (int j = 0; j < frame.n_feat; ++j) (*pf)[j] = Vec3d(pfo[3 * j], pfo[3 * j + 1], pfo[3 * j + 2]); }
    if (rows > 0) {
        dx.resize(state->curr_cov_size());
        boxPlus(state, dx);
    }
    return rows;
}

int StateManager::msckfUpdate(std::shared_ptr<State> state, const ingvio_msckf_frame& frame, const ingvio_msckf_opts& opts,
                              std::vector<int>* accepted)
{
    const int ldp = ingvio_ldp(state->_ctx);
    VecXd dx(ldp, 0.0);
    std::vector<int> acc(std::max(ingvio_f_max(state->_ctx), frame.n_feat), 0);      // the ABI writes f_max entries per filter
    int rows = 0;
    const int rc = ingvio_msckf_update(state->_ctx, state->_b, 1, &frame, &opts, dx.data(), acc.data(), nullptr, &rows);
    if (rc < 0) fatal(state, "msckfUpdate", rc);
    if (accepted) accepted->assign(acc.begin(), acc.begin() + frame.n_feat);
    if (rows > 0) {
        dx.resize(state->curr_cov_size());
        boxPlus(state, dx);
    }
    return rows;
}

}  // namespace ingvio
