#include "LandmarkUpdate.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <iostream>

#include "StateManager.h"

namespace ingvio {

namespace {

Mat3d skewOf(const Vec3d& v)
{
    Mat3d S;
    S(0, 1) = -v[2]; S(0, 2) = v[1];
    S(1, 0) = v[2]; S(1, 2) = -v[0];
    S(2, 0) = -v[1]; S(2, 1) = v[0];
    return S;
}

// d(x/z, y/z)/dq as a 3x3 with a zero third row, so that it chains with Mat3d products
Mat3d projJac(const Vec3d& q)
{
    Mat3d Hp;
    Hp(0, 0) = 1.0 / q.z(); Hp(0, 2) = -q.x() / std::pow(q.z(), 2);
    Hp(1, 1) = 1.0 / q.z(); Hp(1, 2) = -q.y() / std::pow(q.z(), 2);
    return Hp;
}

// rows (r0, r0+1) of H, columns c0..c0+2  <-  sign * top two rows of M
void put2(MatXd& H, int r0, int c0, const Mat3d& M, double sign = 1.0)
{
    for (int r = 0; r < 2; ++r)
        for (int c = 0; c < 3; ++c) H(r0 + r, c0 + c) = sign * M(r, c);
}

bool hasNaN(const Mat3d& M)
{
    for (int i = 0; i < 9; ++i) if (std::isnan(M.m[i])) return true;
    return false;
}

void checkTracked(const std::shared_ptr<MapServer>& map_server, const std::shared_ptr<State>& state, int id, bool stereo)
{
    // the four consistency exits of LandmarkUpdate.cpp:57-80
    if (map_server->find(id) == map_server->end()) {
        std::cout << "[LandmarkUpdate]: Landmark in state not in map server!" << std::endl;
        std::exit(EXIT_FAILURE);
    }
    const auto& fi = map_server->at(id);
    if (fi->_ftype != FeatureInfo::SLAM) {
        std::cout << "[LandmarkUpdate]: Landmark in state not marked SLAM type in map server!" << std::endl;
        std::exit(EXIT_FAILURE);
    }
    const bool tracked = stereo ? fi->_stereo_obs.count(state->_timestamp) > 0 : fi->_mono_obs.count(state->_timestamp) > 0;
    if (!tracked) {
        std::cout << "[LandmarkUpdate]: Landmark in state not tracked to curr time! Should have been marged before!" << std::endl;
        std::exit(EXIT_FAILURE);
    }
    if (fi->_landmark != state->_anchored_landmarks.at(id)) {
        std::cout << "[LandmarkUpdate]: Landmark ptr in state not the same as that in map server!" << std::endl;
        std::exit(EXIT_FAILURE);
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
void FeatureInfoManager::changeAnchoredPose(std::shared_ptr<FeatureInfo> feature_info, std::shared_ptr<State> state,
                                            double target_sw_timestamp)
{
    if (state->_sw_camleft_poses.size() < 2) return;
    if (state->_sw_camleft_poses.find(target_sw_timestamp) == state->_sw_camleft_poses.end() ||
        state->_anchored_landmarks.find(feature_info->_id) == state->_anchored_landmarks.end())
        return;
    if (feature_info->_ftype != FeatureInfo::SLAM) return;
    bool anchor_in_window = false;
    for (const auto& item : state->_sw_camleft_poses)
        if (item.second == feature_info->anchor()) { anchor_in_window = true; break; }
    if (!anchor_in_window || state->_anchored_landmarks.at(feature_info->_id) != feature_info->_landmark) return;
    const auto target = state->_sw_camleft_poses.at(target_sw_timestamp);
    std::vector<std::shared_ptr<Type>> var_order = { feature_info->anchor(), target, state->_anchored_landmarks.at(feature_info->_id) };
    // the landmark error is tied to its anchor's rotation error: moving the anchor is a linear map of (old, new, lm)
    MatXd H(3, 15);
    const Mat3d S = skewOf(feature_info->_landmark->valuePosXyz());
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) { H(r, c) = -S(r, c); H(r, 6 + c) = S(r, c); H(r, 12 + c) = r == c ? 1.0 : 0.0; }
    StateManager::replaceVarLinear(state, feature_info->_landmark, var_order, H);
    feature_info->_landmark->resetAnchoredPose(target, true);
}

void FeatureInfoManager::changeAnchoredPose(std::shared_ptr<FeatureInfo> feature_info, std::shared_ptr<State> state)
{
    if (state->_sw_camleft_poses.size() < 2) return;
    changeAnchoredPose(feature_info, state, state->_sw_camleft_poses.rbegin()->first);
}

// ---------------------------------------------------------------------------------------------------------------------
void LandmarkUpdate::landmarkRows(const std::shared_ptr<FeatureInfo> fi, const std::shared_ptr<State> state, bool stereo, VecXd& res,
                                  MatXd& H)
{
    const double t = state->_timestamp;
    const bool tracked = stereo ? fi->_stereo_obs.count(t) > 0 : fi->_mono_obs.count(t) > 0;
    if (!tracked || fi->_ftype != FeatureInfo::SLAM) {
        std::cout << "[LandmarkUpdate]: Cannot calc curr slam feature " << (stereo ? "stereo" : "mono") << " res and jacobi!" << std::endl;
        std::exit(EXIT_FAILURE);
    }
    const int rows = stereo ? 4 : 2;
    res.assign(rows, 0.0);
    H.resize(rows, 24);
    const Vec3d pf_w = fi->_landmark->valuePosXyz();
    const Mat3d R_i2w_T = state->_extended_pose->valueLinearAsMat().transpose();
    const Mat3d R_cl2i_T = state->_camleft_imu_extrinsics->valueLinearAsMat().transpose();
    const Vec3d pf_i = R_i2w_T * (pf_w - state->_extended_pose->valueTrans1());
    const Vec3d pf_cl = R_cl2i_T * (pf_i - state->_camleft_imu_extrinsics->valueTrans());
    const Mat3d R_w2cl = R_cl2i_T * R_i2w_T;
    const Mat3d Sw = skewOf(pf_w), Si = skewOf(pf_i);
    const Iso3& T_lr = state->_state_params._T_cl2cr;
    for (int eye = 0; eye < (stereo ? 2 : 1); ++eye) {
        const Vec3d q = eye == 0 ? pf_cl : T_lr * pf_cl;
        const Mat3d HL = eye == 0 ? projJac(q) : projJac(q) * T_lr.R;          // projection chained with the left->right rotation
        const double u = stereo ? (eye == 0 ? fi->_stereo_obs.at(t)->_u0 : fi->_stereo_obs.at(t)->_u1) : fi->_mono_obs.at(t)->_u0;
        const double v = stereo ? (eye == 0 ? fi->_stereo_obs.at(t)->_v0 : fi->_stereo_obs.at(t)->_v1) : fi->_mono_obs.at(t)->_v0;
        res[2 * eye] = u - q.x() / q.z();
        res[2 * eye + 1] = v - q.y() / q.z();
        const Mat3d A = HL * R_w2cl, Cc = HL * R_cl2i_T;
        put2(H, 2 * eye, 0, A * Sw);                                            // extended pose: rotation
        put2(H, 2 * eye, 3, A, -1.0);                                           //                position (velocity block stays 0)
        put2(H, 2 * eye, 9, Cc * Si);                                           // extrinsics: rotation
        put2(H, 2 * eye, 12, Cc, -1.0);                                         //             position
        // anchor rotation: left rows -H_proj R_w2cl [pf]x; the right rows are written WITHOUT R_w2cl (LandmarkUpdate.cpp:682, Q12)
        put2(H, 2 * eye, 15, eye == 0 ? A * Sw : HL * Sw, -1.0);
        put2(H, 2 * eye, 21, A);                                                // landmark
    }
}

void LandmarkUpdate::landmarkRowsSw(const std::shared_ptr<FeatureInfo> fi, const std::shared_ptr<State> state, VecXd& res, MatXd& H)
{
    const double t = state->_timestamp;
    if (fi->_mono_obs.count(t) == 0 || fi->_ftype != FeatureInfo::SLAM) {
        std::cout << "[LandmarkUpdate]: Cannot calc curr slam feature mono res and jacobi!" << std::endl;
        std::exit(EXIT_FAILURE);
    }
    const auto curr = state->_sw_camleft_poses.at(t);
    const Vec3d pf_w = fi->_landmark->valuePosXyz();
    const Mat3d R_T = curr->valueLinearAsMat().transpose();
    const Vec3d q = R_T * (pf_w - curr->valueTrans());
    res.assign(2, 0.0);
    H.resize(2, 15);
    res[0] = fi->_mono_obs.at(t)->_u0 - q.x() / q.z();
    res[1] = fi->_mono_obs.at(t)->_v0 - q.y() / q.z();
    const Mat3d A = projJac(q) * R_T;
    if (curr != fi->_landmark->getAnchoredPose()) {
        put2(H, 0, 0, A * skewOf(pf_w));
        put2(H, 0, 6, A * skewOf(pf_w), -1.0);
    }
    put2(H, 0, 3, A, -1.0);
    put2(H, 0, 12, A);
}

void LandmarkUpdate::featAllObsRows(const std::shared_ptr<FeatureInfo> fi, const std::shared_ptr<State> state, bool stereo,
                                    VecXd& res_block, MatXd& Hx_block, MatXd& Hf_block)
{
    // generateSwVarOrder (:502-519): every clone, time order, 6 columns each
    std::map<std::shared_ptr<SE3>, int> col_of;
    int cnt = 0;
    for (const auto& item : state->_sw_camleft_poses) col_of[item.second] = 6 * cnt++;
    const int per = stereo ? 4 : 2;
    const int num_of_cols = 6 * cnt;
    const int num_of_rows = per * (stereo ? fi->numOfStereoFrames() : fi->numOfMonoFrames());
    // As written (:491-496): the shrink-to-fit test compares against the still-empty output, so the blocks keep their
    // full height and skipped observations leave zero rows at the bottom (they count in res.rows() = the chi2 dof).
    res_block.assign(num_of_rows, 0.0);
    Hx_block.resize(num_of_rows, num_of_cols);
    Hf_block.resize(num_of_rows, 3);
    const Vec3d pf_w = fi->_landmark->valuePosXyz();
    const auto anchor = fi->_landmark->getAnchoredPose();
    const Iso3& T_lr = state->_state_params._T_cl2cr;
    std::vector<double> stamps;
    if (stereo) for (const auto& o : fi->_stereo_obs) stamps.push_back(o.first);
    else for (const auto& o : fi->_mono_obs) stamps.push_back(o.first);
    int row = 0;
    for (const double t : stamps) {
        const auto it = state->_sw_camleft_poses.find(t);
        if (it == state->_sw_camleft_poses.end()) continue;
        const auto pose = it->second;
        const Mat3d R_T = pose->valueLinearAsMat().transpose();
        const Vec3d pf_c = R_T * (pf_w - pose->valueTrans());
        const Mat3d Hp = projJac(pf_c);
        const Mat3d dth = R_T * skewOf(pf_w);
        if (hasNaN(Hp) || hasNaN(dth)) continue;
        const int co = col_of.at(pose), ca = col_of.at(anchor);
        for (int eye = 0; eye < (stereo ? 2 : 1); ++eye) {
            const Vec3d q = eye == 0 ? pf_c : T_lr * pf_c;
            const Mat3d HL = eye == 0 ? Hp : projJac(q) * T_lr.R;
            if (pose != anchor) {
                put2(Hx_block, row + 2 * eye, co, HL * dth);
                put2(Hx_block, row + 2 * eye, ca, HL * dth, -1.0);
            }
            put2(Hx_block, row + 2 * eye, co + 3, HL * R_T, -1.0);
            put2(Hf_block, row + 2 * eye, 0, HL * R_T);
            double u, v;
            if (stereo) { const auto& m = fi->_stereo_obs.at(t); u = eye == 0 ? m->_u0 : m->_u1; v = eye == 0 ? m->_v0 : m->_v1; }
            else { const auto& m = fi->_mono_obs.at(t); u = m->_u0; v = m->_v0; }
            res_block[row + 2 * eye] = u - q.x() / q.z();
            res_block[row + 2 * eye + 1] = v - q.y() / q.z();
        }
        row += per;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
void LandmarkUpdate::updateLandmarkMono(std::shared_ptr<State> s, std::shared_ptr<MapServer> m) { update(s, m, false); }
void LandmarkUpdate::updateLandmarkStereo(std::shared_ptr<State> s, std::shared_ptr<MapServer> m) { update(s, m, true); }

void LandmarkUpdate::update(std::shared_ptr<State> state, std::shared_ptr<MapServer> map_server, bool stereo)
{
    // LandmarkUpdate.cpp:32-149: rows of every in-state landmark, a chi^2 gate per landmark on the prior (:98-99, dof = rows),
    // the accepted rows stacked, one ekfUpdate - all of it in one device call (kernels_lmbatch.hip); the nominal values stay here
    _last_rows = 0;
    _last_upd_ids.clear(); _last_upd_acc.clear();
    const int L = (int)state->_anchored_landmarks.size();
    if (L == 0) return;
    if (L > INGVIO_LM_MAX) {          // unreachable: State::construct refuses max_landmark_features > INGVIO_LM_MAX at start-up
        std::cout << "[LandmarkUpdate]: " << L << " in-state landmarks exceed the device limit of " << INGVIO_LM_MAX
                  << ", landmark update skipped for this frame" << std::endl;
        return;
    }
    const int per = stereo ? 4 : 2;
    const double t = state->_timestamp;
    std::vector<int> lm_idx, anchor_idx;
    std::vector<double> pf, uv;
    std::vector<unsigned char> tracked;
    std::vector<int> order_ids;
    for (const auto& item : state->_anchored_landmarks) {
        order_ids.push_back(item.first);
        checkTracked(map_server, state, item.first, stereo);
        const auto fi = map_server->at(item.first);
        lm_idx.push_back(item.second->idx());
        anchor_idx.push_back(item.second->getAnchoredPose()->idx());
        const Vec3d p = item.second->valuePosXyz();
        pf.insert(pf.end(), { p.x(), p.y(), p.z() });
        if (stereo) { const auto& m = fi->_stereo_obs.at(t); uv.insert(uv.end(), { m->_u0, m->_v0, m->_u1, m->_v1 }); }
        else { const auto& m = fi->_mono_obs.at(t); uv.insert(uv.end(), { m->_u0, m->_v0, 0.0, 0.0 }); }
        tracked.push_back(1);
    }
    ingvio_landmark_frame fr;
    std::memset(&fr, 0, sizeof fr);
    const Mat3d R_i2w = state->_extended_pose->valueLinearAsMat(), R_cl2i = state->_camleft_imu_extrinsics->valueLinearAsMat();
    const Vec3d p_i2w = state->_extended_pose->valueTrans1(), p_c2i = state->_camleft_imu_extrinsics->valueTrans();
    std::memcpy(fr.R_i2w, R_i2w.m, sizeof fr.R_i2w); std::memcpy(fr.R_cl2i, R_cl2i.m, sizeof fr.R_cl2i);
    for (int i = 0; i < 3; ++i) { fr.p_i2w[i] = p_i2w[i]; fr.p_c2i[i] = p_c2i[i]; }
    fr.idx_epose = state->_extended_pose->idx(); fr.idx_ext = state->_camleft_imu_extrinsics->idx();
    fr.n_lm = L; fr.lm_idx = lm_idx.data(); fr.anchor_idx = anchor_idx.data(); fr.pf = pf.data(); fr.uv = uv.data(); fr.tracked = tracked.data();
    ingvio_landmark_opts op;
    std::memset(&op, 0, sizeof op);
    op.stereo = stereo ? 1 : 0; op.noise = _noise; op.chi2_thr = chi2TableDense(per + 1)[per];
    const Iso3& T_lr = state->_state_params._T_cl2cr;
    std::memcpy(op.R_cl2cr, T_lr.R.m, sizeof op.R_cl2cr);
    for (int i = 0; i < 3; ++i) op.t_cl2cr[i] = T_lr.t[i];
    std::vector<int> acc;
    _last_rows = StateManager::landmarkUpdate(state, fr, op, &acc);
    std::vector<std::pair<int, int>> rec;
    for (size_t i = 0; i < order_ids.size(); ++i) rec.emplace_back(order_ids[i], i < acc.size() ? (acc[i] != 0) : 0);
    std::sort(rec.begin(), rec.end());
    for (const auto& r : rec) { _last_upd_ids.push_back(r.first); _last_upd_acc.push_back(r.second); }
}

void LandmarkUpdate::updateLandmarkMonoSw(std::shared_ptr<State> state, std::shared_ptr<MapServer> map_server)
{
    _last_rows = 0;
    if (state->_anchored_landmarks.size() == 0) return;
    if (state->_timestamp != state->_sw_camleft_poses.rbegin()->first) {
        std::cout << "[LandmarkUpdate]: Last sw pose is not at curr time!" << std::endl;
        std::exit(EXIT_FAILURE);
    }
    const std::shared_ptr<SE3> curr = state->_sw_camleft_poses.rbegin()->second;
    std::vector<std::shared_ptr<Type>> var_order = { curr };
    std::map<std::shared_ptr<Type>, int> col_of;
    col_of[curr] = 0;
    int col_cnt = 6;
    struct Block { VecXd res; MatXd H; std::shared_ptr<Type> anchor, lm; };
    std::vector<Block> accepted;
    for (const auto& item : state->_anchored_landmarks) {
        checkTracked(map_server, state, item.first, false);
        Block blk;
        blk.anchor = item.second->getAnchoredPose();
        blk.lm = item.second;
        landmarkRowsSw(map_server->at(item.first), state, blk.res, blk.H);
        if (curr == item.second->getAnchoredPose()) {
            std::cout << "[LandmarkUpdate]: Warning! Current pose is the same as anchored pose!" << std::endl;
            continue;
        }
        if (!testChiSquared(state, blk.res, blk.H, { curr, blk.anchor, blk.lm }, _noise)) continue;
        for (const auto& v : { blk.anchor, blk.lm })
            if (col_of.find(v) == col_of.end()) { col_of[v] = col_cnt; col_cnt += v->size(); var_order.push_back(v); }
        accepted.push_back(std::move(blk));
    }
    if (accepted.empty()) return;
    const int rows = 2 * (int)accepted.size();
    MatXd H_large(rows, col_cnt);
    VecXd res_large(rows, 0.0);
    for (size_t a = 0; a < accepted.size(); ++a) {
        const Block& blk = accepted[a];
        const int ca = col_of.at(blk.anchor), cl = col_of.at(blk.lm);
        for (int r = 0; r < 2; ++r) {
            const int R = 2 * (int)a + r;
            res_large[R] = blk.res[r];
            for (int c = 0; c < 6; ++c) { H_large(R, c) = blk.H(r, c); H_large(R, ca + c) = blk.H(r, 6 + c); }
            for (int c = 0; c < 3; ++c) H_large(R, cl + c) = blk.H(r, 12 + c);
        }
    }
    MatXd Rn = MatXd::Identity(rows);
    for (int i = 0; i < rows; ++i) Rn(i, i) = std::pow(_noise, 2);
    StateManager::ekfUpdate(state, var_order, H_large, res_large, Rn);
    _last_rows = rows;
}

// ---------------------------------------------------------------------------------------------------------------------
void LandmarkUpdate::initNewLandmarkMono(std::shared_ptr<State> s, std::shared_ptr<MapServer> m, std::shared_ptr<Triangulator> t, int n)
{ initNew(s, m, t, n, false); }
void LandmarkUpdate::initNewLandmarkStereo(std::shared_ptr<State> s, std::shared_ptr<MapServer> m, std::shared_ptr<Triangulator> t, int n)
{ initNew(s, m, t, n, true); }

void LandmarkUpdate::initNew(std::shared_ptr<State> state, std::shared_ptr<MapServer> map_server, std::shared_ptr<Triangulator> tri,
                             int min_init_poses, bool stereo)
{
    _last_init = 0;
    _last_init_ids.clear();
    if ((int)state->_sw_camleft_poses.size() < min_init_poses) return;
    const int vac_num_lm = state->_state_params._max_landmarks - (int)state->_anchored_landmarks.size();
    if (vac_num_lm <= 0) return;
    std::vector<int> ids_to_init;
    for (const auto& item : *map_server) {
        if ((int)ids_to_init.size() >= vac_num_lm) break;
        const int nobs = stereo ? (int)item.second->_stereo_obs.size() : (int)item.second->_mono_obs.size();
        if (nobs < min_init_poses || item.second->_ftype == FeatureInfo::SLAM) continue;
        if (!tri->triangulate(item.second, state, stereo)) continue;          // FeatureInfoManager::triangulateFeatureInfo*
        ids_to_init.push_back(item.first);
    }
    if (ids_to_init.empty()) return;
    std::vector<std::shared_ptr<Type>> sw_var_type;
    for (const auto& item : state->_sw_camleft_poses) sw_var_type.push_back(item.second);
    for (const int id : ids_to_init) {
        VecXd res_block;
        MatXd Hx_block, Hf_block;
        featAllObsRows(map_server->at(id), state, stereo, res_block, Hx_block, Hf_block);
        if (!StateManager::addVariableDelayed(state, map_server->at(id)->_landmark, sw_var_type, Hx_block, Hf_block, res_block, _noise,
                                              0.95, true))
            continue;
        if (state->_anchored_landmarks.find(id) != state->_anchored_landmarks.end()) {
            std::cout << "[LandmarkUpdate]: The id intended to add already in state!" << std::endl;
            std::exit(EXIT_FAILURE);
        }
        state->_anchored_landmarks[id] = map_server->at(id)->_landmark;
        map_server->at(id)->_ftype = FeatureInfo::SLAM;
        ++_last_init;
        _last_init_ids.push_back(id);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
void LandmarkUpdate::reanchor(std::shared_ptr<State> state, std::shared_ptr<MapServer> map_server,
                              const std::vector<std::shared_ptr<SE3>>& old_anchors)
{
    const double latest = state->_sw_camleft_poses.rbegin()->first;
    const std::shared_ptr<SE3> new_anchor = state->_sw_camleft_poses.rbegin()->second;
    std::vector<int> ids_to_marg;
    for (auto& item : *map_server) {
        if (item.second->_ftype != FeatureInfo::SLAM) continue;
        if (std::find(old_anchors.begin(), old_anchors.end(), item.second->_landmark->getAnchoredPose()) == old_anchors.end()) continue;
        const Vec3d body = new_anchor->valueLinearAsMat().transpose() * (item.second->_landmark->valuePosXyz() - new_anchor->valueTrans());
        if (body.z() <= 0) { ids_to_marg.push_back(item.first); continue; }          // behind the new anchor: drop the landmark
        FeatureInfoManager::changeAnchoredPose(item.second, state, latest);
        if (item.second->_landmark->getAnchoredPose() != new_anchor) ids_to_marg.push_back(item.first);
    }
    for (const int id : ids_to_marg) {
        StateManager::margAnchoredLandmarkInState(state, id);
        map_server->erase(id);
    }
}

void LandmarkUpdate::changeLandmarkAnchor(std::shared_ptr<State> state, std::shared_ptr<MapServer> map_server)
{
    const double marg_time = state->nextMargTime();
    if (marg_time == INFINITY || state->_sw_camleft_poses.find(marg_time) == state->_sw_camleft_poses.end()) return;
    reanchor(state, map_server, { state->_sw_camleft_poses.at(marg_time) });
}

void LandmarkUpdate::changeLandmarkAnchor(std::shared_ptr<State> state, std::shared_ptr<MapServer> map_server,
                                          const std::vector<double>& marg_kfs)
{
    if (marg_kfs.size() == 0) return;
    std::vector<std::shared_ptr<SE3>> old_anchors;
    for (const double& marg : marg_kfs) old_anchors.push_back(state->_sw_camleft_poses.at(marg));
    reanchor(state, map_server, old_anchors);
}

}  // namespace ingvio
