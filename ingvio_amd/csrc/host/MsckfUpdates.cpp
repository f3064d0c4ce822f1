#include <chrono>
#include <typeinfo>
#include <cstdio>
#include <cstdlib>
#include "MsckfUpdates.h"
#include <cstring>
#include <iostream>

#include <cstdlib>
#include <iostream>
#include <unordered_set>

#include "IngvioParams.h"
#include "StateManager.h"

namespace ingvio {

// common timestamps of the window and the observations, in time order (Triangulator.h filterCommonTimestamp), one
// feature, through the C ABI
bool Triangulator::run(const std::shared_ptr<State> state, const std::map<double, std::shared_ptr<SE3>>& sw_poses,
                       const std::vector<double>& stamps, const std::vector<double>& uv4, bool stereo, const Iso3& T_cl2cr,
                       Vec3d& pf) const
{
    std::vector<double> cR, cp, uv;
    std::vector<int> cidx;
    for (size_t k = 0; k < stamps.size(); ++k) {
        const auto it = sw_poses.find(stamps[k]);
        if (it == sw_poses.end()) continue;
        const Mat3d& R = it->second->valueLinearAsMat();
        const Vec3d& p = it->second->valueTrans();
        cR.insert(cR.end(), R.m, R.m + 9);
        cp.insert(cp.end(), p.v, p.v + 3);
        uv.insert(uv.end(), uv4.begin() + 4 * k, uv4.begin() + 4 * k + 4);
        cidx.push_back(0);
    }
    const int C = (int)cidx.size();
    pf = Vec3d();
    if (C == 0) return false;
    const unsigned long long mask = C >= 64 ? ~0ULL : ((1ULL << C) - 1ULL);
    ingvio_msckf_frame fr;
    std::memset(&fr, 0, sizeof fr);
    fr.n_clones = C; fr.clone_idx = cidx.data(); fr.clone_R = cR.data(); fr.clone_p = cp.data();
    fr.n_feat = 1; fr.obs_mask = &mask; fr.uv = uv.data();
    ingvio_tri_opts o;
    std::memset(&o, 0, sizeof o);
    o.stereo = stereo ? 1 : 0;
    std::memcpy(o.R_cl2cr, T_cl2cr.R.m, sizeof o.R_cl2cr);
    std::memcpy(o.t_cl2cr, T_cl2cr.t.v, sizeof o.t_cl2cr);
    o.trans_thres = _trans_thres; o.huber_epsilon = _huber_epsilon; o.conv_precision = _conv_precision;
    o.init_damping = _init_damping; o.outer_loop_max_iter = _outer_loop_max_iter; o.inner_loop_max_iter = _inner_loop_max_iter;
    o.max_depth = _max_depth; o.min_depth = _min_depth;
    return StateManager::triangulateOne(state, fr, o, pf);
}

bool Triangulator::triangulateMonoObs(const std::shared_ptr<State> state, const std::map<double, std::shared_ptr<MonoMeas>>& mono_obs,
                                      const std::map<double, std::shared_ptr<SE3>>& sw_poses, Vec3d& pf) const
{
    std::vector<double> stamps, uv4;
    for (const auto& item : mono_obs) { stamps.push_back(item.first); uv4.insert(uv4.end(), { item.second->_u0, item.second->_v0, 0.0, 0.0 }); }
    return run(state, sw_poses, stamps, uv4, false, Iso3(), pf);
}

bool Triangulator::triangulateStereoObs(const std::shared_ptr<State> state,
                                        const std::map<double, std::shared_ptr<StereoMeas>>& stereo_obs,
                                        const std::map<double, std::shared_ptr<SE3>>& sw_poses, const Iso3& T_cl2cr, Vec3d& pf) const
{
    std::vector<double> stamps, uv4;
    for (const auto& item : stereo_obs) {
        stamps.push_back(item.first);
        uv4.insert(uv4.end(), { item.second->_u0, item.second->_v0, item.second->_u1, item.second->_v1 });
    }
    return run(state, sw_poses, stamps, uv4, true, T_cl2cr, pf);
}

// FeatureInfoManager::triangulateFeatureInfoMono / Stereo (MapServerManager.cpp:274-341)
bool Triangulator::accept(std::shared_ptr<FeatureInfo> fi, bool flag, const Vec3d& pf) const
{
    if (!flag || pf[0] != pf[0] || pf[1] != pf[1] || pf[2] != pf[2]) return false;
    ++fi->_numOfTri;
    const auto anchor = fi->_landmark->getAnchoredPose();
    if (anchor == nullptr) return false;
    const Vec3d body = anchor->valueLinearAsMat().transpose() * (pf - anchor->valueTrans());
    if (body.z() <= 0) return false;                                                          // :290, :325
    if (!fi->_isTri) {
        fi->_landmark->setFejPosXyz(pf);
        fi->_landmark->setValuePosXyz(pf);
        fi->_isTri = true;
        return true;
    }
    fi->_landmark->setValuePosXyz(pf);
    return true;
}

bool Triangulator::triangulate(std::shared_ptr<FeatureInfo> fi, const std::shared_ptr<State> state, bool stereo)
{
    Vec3d pf;
    const bool flag = stereo ? triangulateStereoObs(state, fi->_stereo_obs, state->_sw_camleft_poses, state->_state_params._T_cl2cr, pf)
                             : triangulateMonoObs(state, fi->_mono_obs, state->_sw_camleft_poses, pf);
    return accept(fi, flag, pf);
}

void Triangulator::fillOpts(const std::shared_ptr<State>& state, bool stereo, ingvio_tri_opts& o) const
{
    std::memset(&o, 0, sizeof o);
    o.stereo = stereo ? 1 : 0;
    std::memcpy(o.R_cl2cr, state->_state_params._T_cl2cr.R.m, sizeof o.R_cl2cr);
    std::memcpy(o.t_cl2cr, state->_state_params._T_cl2cr.t.v, sizeof o.t_cl2cr);
    o.trans_thres = _trans_thres; o.huber_epsilon = _huber_epsilon; o.conv_precision = _conv_precision;
    o.init_damping = _init_damping; o.outer_loop_max_iter = _outer_loop_max_iter; o.inner_loop_max_iter = _inner_loop_max_iter;
    o.max_depth = _max_depth; o.min_depth = _min_depth;
}

void Triangulator::triangulateMany(const std::vector<std::shared_ptr<FeatureInfo>>& feats, const std::shared_ptr<State> state, bool stereo,
                                   std::vector<char>& ok)
{
    ok.assign(feats.size(), 0);
    const int fcap = ingvio_f_max(StateManager::ctx(state));
    const auto& sw = state->_sw_camleft_poses;
    const int C = (int)sw.size();
    // no window, or one the batched call cannot hold (more slots than the context's clone capacity or than the 64-bit mask): feature by
    // feature - that path sends a feature's COMMON stamps only (the batched call used to refuse such windows for every feature)
    if (feats.empty() || C == 0 || C > 64 || C > ingvio_c_max(StateManager::ctx(state))) {
        for (size_t i = 0; i < feats.size(); ++i) ok[i] = triangulate(feats[i], state, stereo) ? 1 : 0;
        return;
    }
    std::vector<double> times, cR, cp;
    std::vector<int> cidx((size_t)C, 0);
    for (const auto& item : sw) {                                   // ascending timestamp = the order filterCommonTimestamp yields
        times.push_back(item.first);
        const Mat3d& R = item.second->valueLinearAsMat();
        const Vec3d& p = item.second->valueTrans();
        cR.insert(cR.end(), R.m, R.m + 9);
        cp.insert(cp.end(), p.v, p.v + 3);
    }
    ingvio_tri_opts o;
    fillOpts(state, stereo, o);
    for (size_t f0 = 0; f0 < feats.size(); f0 += (size_t)fcap) {    // the device frame holds f_max features
        const int nf = (int)std::min((size_t)fcap, feats.size() - f0);
        std::vector<unsigned long long> mask((size_t)nf, 0ULL);
        std::vector<double> uv((size_t)nf * C * 4, 0.0);
        for (int j = 0; j < nf; ++j) {
            const auto& fi = feats[f0 + j];
            for (int s = 0; s < C; ++s) {
                double* q = &uv[((size_t)j * C + s) * 4];
                if (stereo) {
                    const auto it = fi->_stereo_obs.find(times[s]);
                    if (it == fi->_stereo_obs.end()) continue;
                    q[0] = it->second->_u0; q[1] = it->second->_v0; q[2] = it->second->_u1; q[3] = it->second->_v1;
                } else {
                    const auto it = fi->_mono_obs.find(times[s]);
                    if (it == fi->_mono_obs.end()) continue;
                    q[0] = it->second->_u0; q[1] = it->second->_v0;
                }
                mask[j] |= 1ULL << s;
            }
        }
        ingvio_msckf_frame fr;
        std::memset(&fr, 0, sizeof fr);
        fr.n_clones = C; fr.clone_idx = cidx.data(); fr.clone_R = cR.data(); fr.clone_p = cp.data();
        fr.n_feat = nf; fr.obs_mask = mask.data(); fr.uv = uv.data();
        std::vector<Vec3d> pf;
        std::vector<char> flag;
        StateManager::triangulateFrame(state, fr, o, pf, flag);
        for (int j = 0; j < nf; ++j) ok[f0 + j] = accept(feats[f0 + j], mask[j] != 0ULL && flag[j], pf[j]) ? 1 : 0;
    }
}

namespace {

// The sliding window + a list of features flattened into the SoA of the C ABI.
struct FlatFrame {
    std::vector<double> times;
    std::vector<std::shared_ptr<SE3>> poses;
    std::vector<int> clone_idx, anchor, dof;
    std::vector<double> clone_R, clone_p, pf, uv;
    std::vector<unsigned long long> mask;
    int C = 0, F = 0;

    explicit FlatFrame(const std::shared_ptr<State>& state)
    {
        for (const auto& item : state->_sw_camleft_poses) {          // std::map: ascending timestamp
            times.push_back(item.first);
            poses.push_back(item.second);
            clone_idx.push_back(item.second->idx());
            const Mat3d& R = item.second->valueLinearAsMat();      // current values, not FEJ (Q7)
            clone_R.insert(clone_R.end(), R.m, R.m + 9);
            const Vec3d& p = item.second->valueTrans();
            clone_p.insert(clone_p.end(), p.v, p.v + 3);
        }
        C = (int)times.size();
    }
    int slotOfPose(const std::shared_ptr<SE3>& p) const
    {
        for (int s = 0; s < C; ++s) if (poses[s] == p) return s;
        return -1;
    }
    std::vector<unsigned long long> all_mask;      // with_all: every observation inside the window (what the triangulation uses)
    // `sel`: nullptr = every observation inside the window (RemoveLost), else only these stamps.  with_all: the measurements of EVERY
    // observed slot are kept (and named in all_mask) while `mask` still names the selected ones - the one-call selected update
    // triangulates from all of them on the device
    bool add(const std::shared_ptr<FeatureInfo>& fi, bool stereo, const std::vector<double>* sel, int dof_value, bool with_all = false)
    {
        const int a = slotOfPose(fi->_landmark->getAnchoredPose());
        if (a < 0) return false;
        unsigned long long m = 0ULL, ma = 0ULL;
        std::vector<double> row((size_t)C * 4, 0.0);
        for (int s = 0; s < C; ++s) {
            bool in = true;
            if (sel) { in = false; for (double t : *sel) if (t == times[s]) in = true; if (!in && !with_all) continue; }
            if (stereo) {
                auto it = fi->_stereo_obs.find(times[s]);
                if (it == fi->_stereo_obs.end()) continue;
                row[4 * s] = it->second->_u0; row[4 * s + 1] = it->second->_v0; row[4 * s + 2] = it->second->_u1; row[4 * s + 3] = it->second->_v1;
            } else {
                auto it = fi->_mono_obs.find(times[s]);
                if (it == fi->_mono_obs.end()) continue;
                row[4 * s] = it->second->_u0; row[4 * s + 1] = it->second->_v0;
            }
            ma |= 1ULL << s;
            if (in) m |= 1ULL << s;
        }
        if (!m) return false;
        if (with_all) all_mask.push_back(ma);
        const Vec3d& p = fi->_landmark->valuePosXyz();
        pf.insert(pf.end(), p.v, p.v + 3);
        uv.insert(uv.end(), row.begin(), row.end());
        anchor.push_back(a); mask.push_back(m); dof.push_back(dof_value);
        ++F;
        return true;
    }
    ingvio_msckf_frame view() const
    {
        ingvio_msckf_frame f;
        f.n_clones = C; f.clone_idx = clone_idx.data(); f.clone_R = clone_R.data(); f.clone_p = clone_p.data();
        f.n_feat = F; f.pf = pf.data(); f.anchor = anchor.data(); f.obs_mask = mask.data(); f.uv = uv.data(); f.dof = dof.data();
        return f;
    }
};

ingvio_msckf_opts makeOpts(const std::shared_ptr<State>& state, bool stereo, double noise, const std::vector<double>& table,
                           int max_accept, int compress_rule, int selected_variant)
{
    ingvio_msckf_opts o;
    o.stereo = stereo ? 1 : 0;
    for (int i = 0; i < 9; ++i) o.R_cl2cr[i] = state->_state_params._T_cl2cr.R.m[i];
    for (int i = 0; i < 3; ++i) o.t_cl2cr[i] = state->_state_params._T_cl2cr.t[i];
    o.noise = noise; o.chi2_table = table.data(); o.chi2_len = (int)table.size();
    o.max_accept = max_accept; o.compress_rule = compress_rule; o.selected_variant = selected_variant;
    return o;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
RemoveLostUpdate::RemoveLostUpdate(const IngvioParams& fp)
    : UpdateBase(fp._chi2_max_dof, fp._chi2_thres), _max_valid_ids(fp._hip_max_valid_ids), _compress_rule(fp._hip_compress_rule), _fuse_tri(fp._hip_fuse_triangulation != 0),
      _noise(fp._visual_noise) {}

void RemoveLostUpdate::updateStateMono(std::shared_ptr<State> s, std::shared_ptr<MapServer> m, std::shared_ptr<Triangulator> t) { update(s, m, t, false); }
void RemoveLostUpdate::updateStateStereo(std::shared_ptr<State> s, std::shared_ptr<MapServer> m, std::shared_ptr<Triangulator> t) { update(s, m, t, true); }

void RemoveLostUpdate::update(std::shared_ptr<State> state, std::shared_ptr<MapServer> map_server, std::shared_ptr<Triangulator> tri, bool stereo)
{
    _last_rows = 0; _last_accepted = 0;
    _rec.clear();
    static const bool timing = std::getenv("INGVIO_SHIM_TIMING") != nullptr;      // host wall time of the update's phases on stderr (debugging aid)
    using clk = std::chrono::steady_clock;
    auto us = [](clk::time_point x, clk::time_point y) { return std::chrono::duration<double, std::micro>(y - x).count(); };
    clk::time_point qm, q0, q1, q2, q3;
    if (timing) qm = clk::now();
    markMargFeatures(map_server, state, stereo);                                       // RemoveLostUpdate.cpp:43 / :279
    std::vector<int> update_ids, direct_marg_ids, cand_ids;
    std::vector<std::shared_ptr<FeatureInfo>> cand;
    for (auto& item : *map_server)
        if (item.second->_ftype == FeatureInfo::MSCKF && item.second->_isToMarg) { cand_ids.push_back(item.first); cand.push_back(item.second); }
    if (timing) q0 = clk::now();
    // Triangulation and update in ONE device round trip (ingvio_msckf_update_tri, round 5): every candidate with enough observations is
    // staged, the device triangulates, drops what fails (MapServerManager.cpp:283-305 / :318-340 incl. the depth in the anchor camera)
    // and updates with the rest - the same features, points and order the two calls below produce, minus a synchronisation, a download,
    // an upload and a second packing of the observations (0.11 of the 0.5 ms a lost cohort of 150 tracks costs a single filter).  Every
    // candidate is erased below either way, so nothing of the triangulation has to return to the map server.  Only for the stock
    // Triangulator (a subclass may triangulate differently) and windows / feature counts one device frame holds.
    const int n_sw = (int)state->_sw_camleft_poses.size();
    if (_fuse_tri && typeid(*tri) == typeid(Triangulator) && !cand.empty() && n_sw > 0 && n_sw <= 64 && n_sw <= ingvio_c_max(StateManager::ctx(state)) &&
        (int)cand.size() <= ingvio_f_max(StateManager::ctx(state))) {
        FlatFrame ff(state);
        std::vector<int> slot(cand.size(), -1);                // -1: not enough observations (dropped), -2: updated outside the frame, >= 0: slot in the frame
        int max_dof = 1;
        for (size_t i = 0; i < cand.size(); ++i) {
            const bool enough = stereo ? cand[i]->numOfStereoFrames() >= 3 : cand[i]->numOfMonoFrames() >= 4;           // :287 / :51
            if (!enough) continue;
            const int dof = (stereo ? (int)cand[i]->_stereo_obs.size() : (int)cand[i]->_mono_obs.size()) - 1;           // :332-333 (Q4)
            if (ff.add(cand[i], stereo, nullptr, dof)) { slot[i] = ff.F - 1; if (dof > max_dof) max_dof = dof; }
            else if (tri->triangulate(cand[i], state, stereo)) slot[i] = -2;      // anchor outside the window / nothing observed in it: as before
        }
        if (timing) q1 = q2 = clk::now();
        std::vector<int> acc, tok;
        if (ff.F > 0) {
            const std::vector<double> table = chi2TableDense(max_dof + 1);
            const ingvio_msckf_frame fr = ff.view();
            const ingvio_msckf_opts op = makeOpts(state, stereo, _noise, table, _max_valid_ids, _compress_rule /* 0 = as written, Q2 */, 0);
            ingvio_tri_opts to;
            tri->fillOpts(state, stereo, to);
            _last_rows = StateManager::msckfUpdateTri(state, fr, op, to, &acc, &tok);
        }
        if (timing) q3 = clk::now();
        for (size_t i = 0; i < cand.size(); ++i) {
            if (slot[i] >= 0 && tok[slot[i]] == 1) { _rec.ids.push_back(cand_ids[i]); _rec.accepted.push_back(acc[slot[i]]); _last_accepted += acc[slot[i]]; }
            else if (slot[i] != -2) _rec.direct.push_back(cand_ids[i]);
        }
        _rec.rows = _last_rows;
        for (int id : cand_ids) map_server->erase(id);                                                // :301-302, :402-403
        if (timing) std::fprintf(stderr, "SHIM remove_lost us (one call): mark + candidates %.1f, frame of %d features %.1f, triangulate + msckfUpdate %.1f, erase %.1f\n",
                                 us(qm, q0), ff.F, us(q0, q1), us(q2, q3), us(q3, clk::now()));
        return;
    }
    std::vector<char> tri_ok;
    tri->triangulateMany(cand, state, stereo, tri_ok);                 // one device call for the frame's lost features
    if (timing) q1 = clk::now();
    for (size_t i = 0; i < cand.size(); ++i) {
        const bool enough = stereo ? cand[i]->numOfStereoFrames() >= 3 : cand[i]->numOfMonoFrames() >= 4;               // :287 / :51
        if (tri_ok[i] && enough) update_ids.push_back(cand_ids[i]);
        else direct_marg_ids.push_back(cand_ids[i]);
    }
    for (const auto& id : direct_marg_ids) map_server->erase(id);
    _rec.direct = direct_marg_ids;
    if (update_ids.size() == 0) return;
    FlatFrame ff(state);
    int max_dof = 1;
    for (int id : update_ids) {
        const auto& fi = map_server->at(id);
        const int dof = (stereo ? (int)fi->_stereo_obs.size() : (int)fi->_mono_obs.size()) - 1;       // :332-333 (Q4)
        if (ff.add(fi, stereo, nullptr, dof)) _rec.ids.push_back(id);
        if (dof > max_dof) max_dof = dof;
    }
    if (timing) q2 = clk::now();
    if (ff.F > 0) {
        const std::vector<double> table = chi2TableDense(max_dof + 1);
        const ingvio_msckf_frame fr = ff.view();
        const ingvio_msckf_opts op = makeOpts(state, stereo, _noise, table, _max_valid_ids, _compress_rule /* 0 = as written, Q2 */, 0);
        std::vector<int> acc;
        _last_rows = StateManager::msckfUpdate(state, fr, op, &acc);
        for (int a : acc) _last_accepted += a;
        _rec.accepted = acc; _rec.rows = _last_rows;
    }
    if (timing) q3 = clk::now();
    for (const auto& id : update_ids) map_server->erase(id);                                          // :402-403
    if (timing) std::fprintf(stderr, "SHIM remove_lost us: mark + candidates %.1f, triangulate %zu features %.1f, frame of %d features %.1f, msckfUpdate %.1f, erase %.1f\n",
                             us(qm, q0), cand.size(), us(q0, q1), ff.F, us(q1, q2), us(q2, q3), us(q3, clk::now()));
}

// ---------------------------------------------------------------------------------------------
SwMargUpdate::SwMargUpdate(const IngvioParams& fp)
    : UpdateBase(fp._chi2_max_dof, fp._chi2_thres), _noise(fp._visual_noise), _frame_select_interval(fp._frame_select_interval),
      _fuse_tri(fp._hip_fuse_triangulation != 0) {}

void SwMargUpdate::selectSwTimestamps(const std::map<double, std::shared_ptr<SE3>>& sw_poses, const double& marg_time,
                                      std::vector<double>& selected_timestamps)
{
    selected_timestamps.clear();
    if (marg_time == INFINITY || sw_poses.find(marg_time) == sw_poses.end()) return;
    int cnt = 1;
    selected_timestamps.push_back(marg_time);
    for (const auto& item : sw_poses) {
        if (item.first <= marg_time) continue;
        if (cnt % _frame_select_interval == 0) selected_timestamps.push_back(item.first);
        ++cnt;
    }
}

void SwMargUpdate::updateStateMono(std::shared_ptr<State> s, std::shared_ptr<MapServer> m, std::shared_ptr<Triangulator> t) { update(s, m, t, false); }
void SwMargUpdate::updateStateStereo(std::shared_ptr<State> s, std::shared_ptr<MapServer> m, std::shared_ptr<Triangulator> t) { update(s, m, t, true); }

static int selectedUpdate(UpdateBase& base, std::shared_ptr<State> state, std::shared_ptr<MapServer> map_server,
                          std::shared_ptr<Triangulator> tri, bool stereo, const std::vector<double>& sel, int dof, double noise, UpdateRecord& rec,
                          bool fuse_tri)
{
    rec.clear();
    rec.stamps = sel;
    // features observed at every selected stamp (SwMargUpdate.cpp:236-257 / KeyframeUpdate.cpp:607-628)
    FlatFrame ff(state);
    std::vector<std::shared_ptr<FeatureInfo>> cand;
    for (const auto& item : *map_server) {
        const auto& fi = item.second;
        if (fi->_ftype != FeatureInfo::MSCKF) continue;
        bool miss = false;
        for (double ts : sel)
            if (stereo ? fi->_stereo_obs.find(ts) == fi->_stereo_obs.end() : fi->_mono_obs.find(ts) == fi->_mono_obs.end()) { miss = true; break; }
        if (!miss) cand.push_back(fi);
    }
    const int n_sw = (int)state->_sw_camleft_poses.size();
    if (fuse_tri && typeid(*tri) == typeid(Triangulator) && !cand.empty() && n_sw > 0 && n_sw <= 64 && n_sw <= ingvio_c_max(StateManager::ctx(state)) &&
        (int)cand.size() <= ingvio_f_max(StateManager::ctx(state))) {
        // triangulation (from EVERY observation in the window) and update (at the selected stamps) in one device round trip, see
        // RemoveLostUpdate::update.  These features live on: value / FEJ position and the attempt counter are set from the call's results
        // exactly as Triangulator::accept sets them.
        std::vector<int> slot(cand.size(), -1);
        for (size_t i = 0; i < cand.size(); ++i) {
            if (ff.add(cand[i], stereo, &sel, dof, true)) slot[i] = ff.F - 1;
            else tri->triangulate(cand[i], state, stereo);         // not stageable (anchor outside the window): triangulated as before, never updated
        }
        if (ff.F == 0) return 0;
        const std::vector<double> table = base.chi2TableDense(dof + 1);
        const ingvio_msckf_frame fr = ff.view();
        const ingvio_msckf_opts op = makeOpts(state, stereo, noise, table, 0, 1 /* top_n, SwMargUpdate.cpp:350-351 */, 1 /* Q10 */);
        ingvio_tri_opts to;
        tri->fillOpts(state, stereo, to);
        std::vector<int> acc, tok;
        std::vector<Vec3d> pf;
        rec.rows = StateManager::msckfUpdateTri(state, fr, op, to, &acc, &tok, ff.all_mask.data(), &pf);
        for (size_t i = 0; i < cand.size(); ++i) {
            if (slot[i] < 0) continue;
            const auto& fi = cand[i];
            if (tok[slot[i]] == 0) continue;                       // MapServerManager.cpp:283-286: nothing counted, nothing set
            ++fi->_numOfTri;
            if (tok[slot[i]] != 1) continue;                       // behind its anchor camera (:290 / :325)
            if (!fi->_isTri) { fi->_landmark->setFejPosXyz(pf[slot[i]]); fi->_isTri = true; }
            fi->_landmark->setValuePosXyz(pf[slot[i]]);
            rec.ids.push_back(fi->_id); rec.accepted.push_back(acc[slot[i]]);
        }
        return rec.rows;
    }
    std::vector<char> tri_ok;
    tri->triangulateMany(cand, state, stereo, tri_ok);
    for (size_t i = 0; i < cand.size(); ++i)
        if (tri_ok[i] && ff.add(cand[i], stereo, &sel, dof)) rec.ids.push_back(cand[i]->_id);
    if (ff.F == 0) return 0;
    const std::vector<double> table = base.chi2TableDense(dof + 1);
    const ingvio_msckf_frame fr = ff.view();
    const ingvio_msckf_opts op = makeOpts(state, stereo, noise, table, 0, 1 /* top_n, SwMargUpdate.cpp:350-351 */, 1 /* Q10 */);
    rec.rows = StateManager::msckfUpdate(state, fr, op, &rec.accepted);
    return rec.rows;
}

void SwMargUpdate::update(std::shared_ptr<State> state, std::shared_ptr<MapServer> map_server, std::shared_ptr<Triangulator> tri, bool stereo)
{
    _last_rows = 0;
    _rec.clear(); _maint.clear();
    const double marg_time = state->nextMargTime();
    if (marg_time == INFINITY) return;
    std::vector<double> selected_timestamps;
    this->selectSwTimestamps(state->_sw_camleft_poses, marg_time, selected_timestamps);
    for (double ts : selected_timestamps)
        if (state->_sw_camleft_poses.find(ts) == state->_sw_camleft_poses.end()) {
            std::cout << "[SwMargUpdate]: selected timestamp not in sw!" << std::endl;                // :462-466
            std::exit(EXIT_FAILURE);
        }
    _last_rows = selectedUpdate(*this, state, map_server, tri, stereo, selected_timestamps, (int)selected_timestamps.size() - 1, _noise, _rec, _fuse_tri);
}

template <bool STEREO>
static void cleanObsAt(std::shared_ptr<MapServer> map_server, const std::vector<double>& stamps, MaintenanceRecord& rec)
{
    std::vector<int>& ids_to_clean = rec.clean_erased;
    ids_to_clean.clear();
    for (auto& item : *map_server)
        for (double t : stamps) {
            if (STEREO) { item.second->_stereo_obs.erase(t); if (item.second->_stereo_obs.empty()) ids_to_clean.push_back(item.first); }
            else { item.second->_mono_obs.erase(t); if (item.second->_mono_obs.empty()) ids_to_clean.push_back(item.first); }
        }
    for (int id : ids_to_clean) map_server->erase(id);
}

void SwMargUpdate::cleanMonoObsAtMargTime(std::shared_ptr<State> state, std::shared_ptr<MapServer> map_server)
{
    const double marg_time = state->nextMargTime();
    if (marg_time == INFINITY) return;
    cleanObsAt<false>(map_server, { marg_time }, _maint);
}
void SwMargUpdate::cleanStereoObsAtMargTime(std::shared_ptr<State> state, std::shared_ptr<MapServer> map_server)
{
    const double marg_time = state->nextMargTime();
    if (marg_time == INFINITY) return;
    cleanObsAt<true>(map_server, { marg_time }, _maint);
}

static void changeAnchor(std::shared_ptr<State> state, std::shared_ptr<MapServer> map_server,
                         const std::unordered_set<std::shared_ptr<SE3>>& old_anchor_set, double min_depth, MaintenanceRecord& rec)
{
    const std::shared_ptr<SE3> new_anchor = state->_sw_camleft_poses.rbegin()->second;
    std::vector<int>& ids_to_marg = rec.anchor_erased;
    ids_to_marg.clear(); rec.anchor_moved.clear();
    for (auto& item : *map_server) {
        if (item.second->_ftype != FeatureInfo::MSCKF) continue;
        if (old_anchor_set.find(item.second->_landmark->getAnchoredPose()) != old_anchor_set.end()) {
            if (item.second->_isTri) {
                const Vec3d pf = item.second->_landmark->valuePosXyz();
                const Vec3d body = new_anchor->valueLinearAsMat().transpose() * (pf - new_anchor->valueTrans());
                if (body.z() <= min_depth) { ids_to_marg.push_back(item.first); continue; }
                item.second->_landmark->resetAnchoredPose(new_anchor, true);
                rec.anchor_moved.push_back(item.first);
            } else
                ids_to_marg.push_back(item.first);
        }
    }
    for (const int& id : ids_to_marg) map_server->erase(id);
}

void SwMargUpdate::changeMSCKFAnchor(std::shared_ptr<State> state, std::shared_ptr<MapServer> map_server)
{
    const double marg_time = state->nextMargTime();
    if (marg_time == INFINITY || state->_sw_camleft_poses.find(marg_time) == state->_sw_camleft_poses.end()) return;
    changeAnchor(state, map_server, { state->_sw_camleft_poses.at(marg_time) }, 0.0, _maint);       // :395 (body.z() <= 0)
}

void SwMargUpdate::margSwPose(std::shared_ptr<State> state)
{
    const double marg_time = state->nextMargTime();
    if (marg_time == INFINITY) return;
    StateManager::margSlidingWindowPose(state, marg_time);
    _maint.marg_stamps.assign(1, marg_time);
}

// ---------------------------------------------------------------------------------------------
KeyframeUpdate::KeyframeUpdate(const IngvioParams& fp)
    : UpdateBase(fp._chi2_max_dof, fp._chi2_thres), _noise(fp._visual_noise), _max_sw_poses(fp._max_sw_clones),
      _fuse_tri(fp._hip_fuse_triangulation != 0) {}

void KeyframeUpdate::getMargKfs(const std::shared_ptr<State> state, std::vector<double>& marg_kfs)
{
    if ((int)state->_sw_camleft_poses.size() < _max_sw_poses || _max_sw_poses < 3) { marg_kfs.clear(); return; }
    if (state->_timestamp == _timestamp && _kfs.size() > 0) { marg_kfs = _kfs; return; }
    if ((int)state->_sw_camleft_poses.size() > _max_sw_poses) {
        std::cout << "[KeyframeUpdate]: Current sw poses larger than max size!" << std::endl;        // :60-64
        std::exit(EXIT_FAILURE);
    }
    _timestamp = state->_timestamp;
    _kfs.clear();
    const int rem = _max_sw_poses - 2;
    const int idx1 = 2 + _select_cnt;
    ++_select_cnt;
    _select_cnt = _select_cnt % rem;
    auto item1 = state->_sw_camleft_poses.rbegin();
    for (int i = 0; i < idx1; ++i) ++item1;
    auto item2 = state->_sw_camleft_poses.rbegin();
    ++item2;
    _kfs.push_back(item1->first);
    _kfs.push_back(item2->first);
    marg_kfs = _kfs;
}

void KeyframeUpdate::updateStateMono(std::shared_ptr<State> s, std::shared_ptr<MapServer> m, std::shared_ptr<Triangulator> t) { update(s, m, t, false); }
void KeyframeUpdate::updateStateStereo(std::shared_ptr<State> s, std::shared_ptr<MapServer> m, std::shared_ptr<Triangulator> t) { update(s, m, t, true); }

void KeyframeUpdate::update(std::shared_ptr<State> state, std::shared_ptr<MapServer> map_server, std::shared_ptr<Triangulator> tri, bool stereo)
{
    _last_rows = 0;
    _rec.clear(); _maint.clear();
    std::vector<double> selected_timestamps;
    this->getMargKfs(state, selected_timestamps);
    if (selected_timestamps.size() == 0) return;
    _last_rows = selectedUpdate(*this, state, map_server, tri, stereo, selected_timestamps, 2 /* KeyframeUpdate.cpp:675-676 */, _noise, _rec, _fuse_tri);
}

void KeyframeUpdate::cleanMonoObsAtMargTime(std::shared_ptr<State> state, std::shared_ptr<MapServer> map_server)
{
    std::vector<double> marg_kfs;
    this->getMargKfs(state, marg_kfs);
    cleanObsAt<false>(map_server, marg_kfs, _maint);
}
void KeyframeUpdate::cleanStereoObsAtMargTime(std::shared_ptr<State> state, std::shared_ptr<MapServer> map_server)
{
    std::vector<double> marg_kfs;
    this->getMargKfs(state, marg_kfs);
    cleanObsAt<true>(map_server, marg_kfs, _maint);
}

void KeyframeUpdate::changeMSCKFAnchor(std::shared_ptr<State> state, std::shared_ptr<MapServer> map_server)
{
    std::vector<double> marg_kfs;
    this->getMargKfs(state, marg_kfs);
    if (marg_kfs.size() == 0) return;
    std::unordered_set<std::shared_ptr<SE3>> old_anchor_set;
    for (const double& marg_time : marg_kfs) old_anchor_set.insert(state->_sw_camleft_poses.at(marg_time));
    changeAnchor(state, map_server, old_anchor_set, 0.3, _maint);                                    // :311
}

void KeyframeUpdate::margSwPose(std::shared_ptr<State> state)
{
    std::vector<double> marg_kfs;
    this->getMargKfs(state, marg_kfs);
    if (marg_kfs.size() == 0) return;
    for (const double& marg_time : marg_kfs) StateManager::margSlidingWindowPose(state, marg_time);
    _maint.marg_stamps = marg_kfs;
}

void markMargFeatures(std::shared_ptr<MapServer> map_server, std::shared_ptr<State> state, bool stereo)
{
    const double curr_timestamp = state->_timestamp;
    std::vector<int> marg_ids;
    for (auto& item : *map_server) {
        const bool has = stereo ? item.second->hasStereoObsAt(curr_timestamp) : item.second->hasMonoObsAt(curr_timestamp);
        if (!has) {
            item.second->_isToMarg = true;
            if (item.second->_ftype == FeatureInfo::SLAM) marg_ids.push_back(item.second->_id);
        }
    }
    for (const int id : marg_ids) {
        StateManager::margAnchoredLandmarkInState(state, id);
        map_server->erase(id);
    }
}

void eraseInvalidFeatures(std::shared_ptr<MapServer> map_server, std::shared_ptr<State> state, std::vector<int>* erased)
{
    std::vector<int> ids_to_remove;
    for (const auto& item : *map_server) {
        const auto& fi = item.second;
        if (!fi->_isTri) continue;
        if (fi->_landmark->getAnchoredPose() == nullptr) { ids_to_remove.push_back(item.first); continue; }
        const auto anchor_ptr = fi->_landmark->getAnchoredPose();
        const Vec3d body = anchor_ptr->valueLinearAsMat().transpose() * (fi->_landmark->valuePosXyz() - anchor_ptr->valueTrans());
        if (body.z() <= 0.2) ids_to_remove.push_back(item.first);
    }
    if (erased) *erased = ids_to_remove;
    for (const int& id : ids_to_remove) {
        if (map_server->at(id)->_ftype == FeatureInfo::SLAM) StateManager::margAnchoredLandmarkInState(state, id);      // :485-486
        map_server->erase(id);
    }
}

}  // namespace ingvio
