// StateManager.h — mirrors ingvio_estimator/src/StateManager.h:38-127: the sole friend of the
// state's covariance.  Every static function keeps the reference's name and argument meaning and
// forwards the covariance arithmetic to libingvio_hip.so; the Type::idx() bookkeeping is done here
// exactly as StateManager.cpp does (append at the end, shift later indices on marginalise).
#pragma once
#include <memory>
#include <vector>

#include "MatX.h"
#include "State.h"

namespace ingvio {

class StateManager {
public:
    StateManager() = delete;

    static bool checkStateContinuity(const std::shared_ptr<State> state);                    // StateManager.cpp:27-40
    static void propagateStateCov(std::shared_ptr<State> state, const double Phi_imu[225],   // :42-119
                                  const double G_imu[180], double dt);
    // k steps of propagateStateCov in one launch (the loop of ImuPropagator.cpp:246-289)
    static void propagateStateCovFused(std::shared_ptr<State> state, int k, const double* Phi, const double* G, const double* dt);
    static MatXd getFullCov(std::shared_ptr<State> state);                                   // :121-126
    static MatXd getMarginalCov(std::shared_ptr<State> state,                                // :128-153
                                const std::vector<std::shared_ptr<Type>>& small_variables);
    static void marginalize(std::shared_ptr<State> state, std::shared_ptr<Type> marg);       // :155-192
    static void addVariableIndependent(std::shared_ptr<State> state, std::shared_ptr<Type> new_state,
                                       const MatXd& new_state_cov_block);                    // :194-214
    static void addGNSSVariable(std::shared_ptr<State> state, const State::GNSSType& gtype, double value, double cov);   // :216-231
    static void margGNSSVariable(std::shared_ptr<State> state, const State::GNSSType& gtype);                            // :233-242
    static void boxPlus(std::shared_ptr<State> state, const VecXd& dx);                      // :244-251
    static void augmentSlidingWindowPose(std::shared_ptr<State> state);                      // :253-296
    static void addAnchoredLandmarkInState(std::shared_ptr<State> state, std::shared_ptr<AnchoredLandmark> anchored_landmark,
                                           int lm_id, const MatXd& cov);                     // :298-314
    static void margSlidingWindowPose(std::shared_ptr<State> state, double marg_time);       // :316-326
    static void margSlidingWindowPose(std::shared_ptr<State> state);                         // :328-338
    static void margAnchoredLandmarkInState(std::shared_ptr<State> state, int lm_id);        // :340-353
    // R: m x m (the reference's signature).  Diagonal / scalar*I matrices are detected and sent as such.
    static void ekfUpdate(std::shared_ptr<State> state, const std::vector<std::shared_ptr<Type>>& var_order,
                          const MatXd& H, const VecXd& res, const MatXd& R);                 // :359-426
    // SLAM-landmark path.  H_old / H_new / res are left untouched: the Givens rotations of
    // :577-589 run on the device copy (the reference rotates its arguments in place and never reads them again).
    static void addVariableDelayedInvertible(std::shared_ptr<State> state, std::shared_ptr<Type> var_new,
                                             const std::vector<std::shared_ptr<Type>>& var_old_order, const MatXd& H_old,
                                             const MatXd& H_new, const VecXd& res, double noise_iso_meas);       // :461-543
    static bool addVariableDelayed(std::shared_ptr<State> state, std::shared_ptr<Type> var_new,
                                   const std::vector<std::shared_ptr<Type>>& var_old_order, const MatXd& H_old, const MatXd& H_new,
                                   const VecXd& res, double noise_iso_meas, double chi2_mult_factor, bool do_chi2 = true);   // :549-637
    static void replaceVarLinear(std::shared_ptr<State> state, const std::shared_ptr<Type> target_var,
                                 const std::vector<std::shared_ptr<Type>>& dependence_order, const MatXd& H);       // :639-693
    static bool checkSubOrder(std::shared_ptr<State> state, const std::vector<std::shared_ptr<Type>>& sub_order);   // :428-445
    static int calcSubVarSize(const std::vector<std::shared_ptr<Type>>& sub_var);            // :447-456

    // whitenResidual's arithmetic (Update.cpp:36-79) lives with the covariance on the device
    static double whitenResidual(std::shared_ptr<State> state, const VecXd& res, const MatXd& H,
                                 const std::vector<std::shared_ptr<Type>>& var_order, const MatXd& R);
    static double whitenResidual(std::shared_ptr<State> state, const VecXd& res, const MatXd& H,
                                 const std::vector<std::shared_ptr<Type>>& var_order, double noise);

    // several gates against the same prior in one device call (ingvio_chi2_gamma_multi); R = noise^2 I for all of them
    struct GateBlock { std::vector<std::shared_ptr<Type>> var_order; const MatXd* H; const VecXd* res; };
    static std::vector<double> whitenResidualMulti(std::shared_ptr<State> state, const std::vector<GateBlock>& blocks, double noise);

    // all in-state landmarks of the current frame in one device call (ingvio_landmark_stage / _run / _fetch): rows, per-landmark
    // gates, stacking, ekfUpdate + boxPlus; returns the rows of the update, accept [n_lm] optional
    static int landmarkUpdate(std::shared_ptr<State> state, const ingvio_landmark_frame& frame, const ingvio_landmark_opts& opts,
                              std::vector<int>* accept = nullptr);
    // MSCKF update on flattened MapServer data + boxPlus; returns rows handed to the Kalman update.
    // f-1: Triangulator::triangulate{Mono,Stereo}Obs of ONE feature on the device (ingvio_triangulate)
    static bool triangulateOne(std::shared_ptr<State> state, const ingvio_msckf_frame& frame, const ingvio_tri_opts& opts, Vec3d& pf);
    // ... and of every feature of `frame` in one call: pf[j], ok[j] for j < frame.n_feat (n_feat <= ingvio_f_max)
    static void triangulateFrame(std::shared_ptr<State> state, const ingvio_msckf_frame& frame, const ingvio_tri_opts& opts,
                                 std::vector<Vec3d>& pf, std::vector<char>& ok);
    static int msckfUpdate(std::shared_ptr<State> state, const ingvio_msckf_frame& frame, const ingvio_msckf_opts& opts,
                           std::vector<int>* accepted = nullptr);
    // the same with the frame's points triangulated on the device first (ingvio_msckf_update_tri): tri_ok[j] for j < frame.n_feat
    // (1 triangulated, 0 failed, 2 behind its anchor camera); tri_mask: the observations the triangulation uses when they differ from
    // frame.obs_mask (nullptr: the same); pf: the triangulated points
    static int msckfUpdateTri(std::shared_ptr<State> state, const ingvio_msckf_frame& frame, const ingvio_msckf_opts& opts,
                              const ingvio_tri_opts& tri, std::vector<int>* accepted, std::vector<int>* tri_ok,
                              const unsigned long long* tri_mask = nullptr, std::vector<Vec3d>* pf = nullptr);

    static ingvio_ctx* ctx(const std::shared_ptr<State>& state) { return state->_ctx; }
    static int filterIndex(const std::shared_ptr<State>& state) { return state->_b; }
    // the error-state variables in covariance order: the (idx, size) table (read-only view for traces and tests)
    static const std::vector<std::shared_ptr<Type>>& errVariables(const std::shared_ptr<State>& state) { return state->_err_variables; }

private:
    static void fatal(const std::shared_ptr<State>& state, const char* what, int rc);
};

}  // namespace ingvio
