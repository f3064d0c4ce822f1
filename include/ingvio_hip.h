/*
 * ingvio_hip.h — C ABI of libingvio_hip.so: the MI355X (gfx950) covariance engine that replaces
 * the Eigen + SuiteSparse path behind InGVIO's StateManager / UpdateBase seam.
 *
 * The reference has no FFI today; its seam is the static API of `StateManager`
 * (ingvio_estimator/src/StateManager.h:38-127, sole friend of State::_cov, State.h:129-135) plus
 * the per-feature helpers of the Update classes.  Each export below names the reference
 * function(s) it replaces.  INTEGRATION.md shows the C++ shim a maintainer adds on the reference
 * side (ingvio_amd/csrc/host/ is that shim, ROS-free).
 *
 * Data model
 *   - A context owns `batch` independent filters on one GPU.  Filter b's covariance lives in HBM
 *     as an FP64 column-major n_b x n_b matrix inside an ldp x ldp buffer (ldp = n_max rounded up
 *     to 16), i.e. exactly Eigen::MatrixXd's layout (State.h:133) so `Eigen::Map` + `cov_set/get`
 *     is a plain strided copy.  The host keeps the typed nominal values and the Type::idx()/size()
 *     table (VecState.h:32-54); only integers and small parameter blocks cross the boundary.
 *   - All pointer arguments are HOST pointers unless the name ends in `_dev`.  Inputs are read
 *     during the call only; outputs are written before the call returns (calls that return
 *     results synchronise the context's stream; pure state mutations are asynchronous).
 *   - Every call returns 0 on success, <0 for the reference's fatal conditions / API misuse
 *     (the host shim turns these back into the reference's std::exit(EXIT_FAILURE) paths),
 *     >0 for soft conditions (e.g. INGVIO_NO_ROWS: nothing accepted, state untouched).
 *   - Thread-compatible per context, no globals (the reference is single-threaded,
 *     IngvioNode.cpp:36).
 *   - 3x3 rotations and T_cl2cr are row-major double[9] (+ double[3]); Phi (15x15), G (15x12),
 *     H and covariance blocks are column-major with explicit leading dimensions.
 */
#ifndef INGVIO_HIP_H
#define INGVIO_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define INGVIO_OK 0
#define INGVIO_NO_ROWS 1             /* soft: no measurement rows survived gating              */
#define INGVIO_NEG_DIAG 2            /* soft: negative diagonal after update (StateManager.cpp:413-421, assert only) */
#define INGVIO_REJECTED 3            /* soft (per-filter status): the block chi^2 gate refused the update, state untouched (GnssUpdate.cpp:286) */
#define INGVIO_E_ARG (-1)            /* bad argument / out of range                            */
#define INGVIO_E_CAPACITY (-2)       /* n_max / c_max / f_max / m_max exceeded                 */
#define INGVIO_E_HIP (-3)            /* HIP runtime error, see ingvio_last_error               */
#define INGVIO_E_NOT_IN_STATE (-4)   /* StateManager.cpp:157-161 "Marg is not in the current state" */
#define INGVIO_E_UNSUPPORTED (-5)
#define INGVIO_E_NOT_PD (-6)         /* the innovation covariance S = H P H^T + R of a generic / landmark update is not positive definite:
                                        the filter's state is left untouched and its dx is zero */

typedef struct ingvio_ctx ingvio_ctx;

typedef struct {
    int batch;      /* independent filters held by this context (>= 1)                          */
    int n_max;      /* max state dimension N (21 + gnss + 6C + 3L)                               */
    int c_max;      /* max clones in the sliding window: <= 36; windows above 16 take the large-window kernels (factored path
                     * only: no dense MSCKF method), see DESIGN.md                */
    int f_max;      /* max features per MSCKF update                                             */
    int m_max;      /* max rows of a generic ekf_update (<= 128 in this build)                   */
    int device;     /* HIP device ordinal                                                        */
    void* stream;   /* hipStream_t to run on, or NULL to create a private non-blocking stream    */
} ingvio_ctx_desc;

/* State ctor (State.cpp:60-91): allocates P (batch x ldp^2 FP64, two ping-pong buffers) + workspaces. */
int ingvio_ctx_create(const ingvio_ctx_desc* desc, ingvio_ctx** out);
int ingvio_ctx_destroy(ingvio_ctx* ctx);
int ingvio_sync(ingvio_ctx* ctx);                     /* waits for the context's stream.  ingvio_propagate(_fused), ingvio_augment_clone,
                                                       * ingvio_marginalize and ingvio_append_independent only ENQUEUE (their inputs are
                                                       * copied into pinned staging before they return; argument / capacity errors are
                                                       * reported at once, a device fault by the next call that synchronises: this one,
                                                       * every cov_get / fetch / update call) */
void* ingvio_ctx_stream(ingvio_ctx* ctx);             /* the hipStream_t all kernels are launched on */
const char* ingvio_last_error(ingvio_ctx* ctx);
int ingvio_ldp(ingvio_ctx* ctx);                      /* leading dimension of the device P buffers  */
int ingvio_f_max(ingvio_ctx* ctx);                    /* feature capacity: length of the accepted[] arrays */
int ingvio_c_max(ingvio_ctx* ctx);                    /* window capacity: the most clones a staged frame may name */
/* Identity of the device code this library was built from (no reference counterpart): a JSON string
 * {"tu": {translation unit: hash of its text + headers + flags}, "kernels": {kernel: translation unit}} written by
 * ingvio_amd/build.py.  Profile counters are stored with it; bench.py prices a kernel only with counters of the same build. */
const char* ingvio_build_id(void);

/* initStateAndCov / getFullCov (State.cpp:126-167, StateManager.cpp:121-126). cov_get synchronises. */
int ingvio_cov_set(ingvio_ctx* ctx, int b, const double* P, int ld, int n);
int ingvio_cov_get(ingvio_ctx* ctx, int b, double* P, int ld);
int ingvio_get_n(ingvio_ctx* ctx, int b, int* n);
/* getMarginalCov (StateManager.cpp:128-153): out is ns x ns column-major, ns = sum(vsize). */
int ingvio_cov_get_marginal(ingvio_ctx* ctx, int b, const int* vidx, const int* vsize, int k, double* out);
/* device-to-device snapshot / restore of all filters' covariances and dims (benchmark hygiene). */
int ingvio_cov_snapshot(ingvio_ctx* ctx);
int ingvio_cov_restore(ingvio_ctx* ctx);

/* propagateStateCov (StateManager.cpp:42-119) for filters [b0, b0+nb).  Phi [nb][225], G [nb][180]
 * (column-major 15x15 / 15x12), dt [nb]; sigma = {noise_g, noise_a, noise_bg, noise_ba};
 * gnss_idx [nb][5] = state idx of {GPS, GLO, GAL, BDS, FS} or -1 (NULL: none);
 * sigma_cb / sigma_rw = StateParams::_noise_clockbias / _noise_cb_rw after quirk Q1. */
int ingvio_propagate(ingvio_ctx* ctx, int b0, int nb, const double* Phi, const double* G, const double* dt,
                     const double sigma[4], int enable_gnss, const int* gnss_idx,
                     double sigma_cb, double sigma_rw);
/* The k-step loop of ImuPropagator::propagateUntil (ImuPropagator.cpp:246-289) in ONE launch:
 * Phi [nb][k][225], G [nb][k][180], dt [nb][k].  The kernel composes the k transitions in LDS and
 * touches the covariance strip once (same result up to FP64 rounding). */
int ingvio_propagate_fused(ingvio_ctx* ctx, int b0, int nb, int k, const double* Phi, const double* G,
                           const double* dt, const double sigma[4], int enable_gnss, const int* gnss_idx,
                           double sigma_cb, double sigma_rw);

/* augmentSlidingWindowPose, covariance part (StateManager.cpp:279-293). R_i2w [nb][9] row-major.
 * new_idx [nb] receives the clone's idx == old N (bit-exact, StateManager.cpp:274). */
int ingvio_augment_clone(ingvio_ctx* ctx, int b0, int nb, const double* R_i2w, int* new_idx);
/* marginalize, covariance part (StateManager.cpp:163-177). idx [nb]. */
int ingvio_marginalize(ingvio_ctx* ctx, int b0, int nb, const int* idx, int size);
/* addVariableIndependent (StateManager.cpp:194-214). blk [nb][size*size] column-major. */
int ingvio_append_independent(ingvio_ctx* ctx, int b0, int nb, int size, const double* blk, int* new_idx);

/* noise kinds for the generic update */
#define INGVIO_R_SCALAR 0   /* R = (*R) * I                                                     */
#define INGVIO_R_DIAG 1     /* R = diag(R[0..m))     (GNSS, GnssUpdate.cpp:195,264)             */
#define INGVIO_R_FULL 2     /* R m x m column-major                                             */

/* ekfUpdate minus boxPlus (StateManager.cpp:359-423) on filter b: var_order as (vidx, vsize)[k],
 * H m x sum(vsize) column-major (ldh), res [m].  dx_out [N] = K res; the host applies boxPlus. */
int ingvio_ekf_update(ingvio_ctx* ctx, int b, const int* vidx, const int* vsize, int k,
                      const double* H, int ldh, int m, const double* res,
                      const double* R, int r_kind, double* dx_out);
/* ekfUpdate for filters [b0, b0+nb) in ONE launch and one synchronisation (e.g. the GNSS update of a whole batch): block i
 * belongs to filter b0+i.  R per block: one double (INGVIO_R_SCALAR) or m doubles (INGVIO_R_DIAG).  dx_out [nb][ldp]
 * (ingvio_ldp), status_out [nb] = INGVIO_OK / INGVIO_NEG_DIAG per filter (may be NULL). */
typedef struct {
    const int* vidx;          /* var_order as (idx, size)[k]                                        */
    const int* vsize;
    int k;
    const double* H;          /* m x sum(vsize), column-major, leading dimension ldh                   */
    int ldh, m;
    const double* res;        /* [m]                                                                */
    const double* R;          /* noise, see r_kind                                                  */
} ingvio_update_block;
int ingvio_ekf_update_batch(ingvio_ctx* ctx, int b0, int nb, const ingvio_update_block* blocks, int r_kind,
                            double* dx_out, int* status_out);
/* ---- GnssUpdate::updateTrackedSys, covariance part, for filters [b0, b0+nb) (GnssUpdate.cpp:148-290) --------------------
 * Block i carries ALL candidate rows of filter b0+i as the host assembled them (pseudo-range rows, then Doppler rows; R = the m
 * row variances) over var_order = [SE23, YOF, clock biases..., FS] (<= 32 columns, <= 256 rows; m = 0: no measurement).  On the
 * device, in one launch sequence and without host round trips:
 *   gate_rows     (_is_gnss_chi2_test, :190,:259)  every row is tested alone on the PRIOR: r^2 / (h Pvv h^T + R_i) < chi2_table[1];
 *                 a row only touches the columns of its own sub_order, so the stacked row gives the reference's 11-column product;
 *                 rejected rows are removed, the rest move up (order kept).  A clock column whose rows were all rejected stays
 *                 in var_order as a zero column: the same posterior as the reference's shorter var_order.
 *   strong_reject (_is_gnss_strong_reject, :286)   if <= 14 rows survive, the block is gated as a whole against chi2_table[rows];
 *                 on failure the state is untouched and the filter's status is INGVIO_REJECTED.
 *   ekfUpdate     (:290) with the diagonal R; dx_out [nb][ldp] = K res (zero when nothing was applied).
 * rows_out [nb]: rows handed to ekfUpdate; keep_out [nb][ingvio_mld()]: 1/0 per candidate row; status_out [nb].
 * stage / run / fetch are the same operation split for device-resident throughput runs (run only enqueues kernels and may be
 * repeated: the staged rows are read-only, each run compacts them into working buffers). */
typedef struct {
    int gate_rows;
    int strong_reject;
    const double* chi2_table;      /* UpdateBase::_chi_squared_table, chi2_table[dof] */
    int chi2_len;
    int in_frame;                  /* ingvio_gnss_stage only: != 0 - the staged update belongs to the frame staged with ingvio_frame_stage
                                    * and is applied BY ingvio_frame_run right after the frame's MSCKF update (IngvioFilter.cpp:329-362
                                    * follows :276-327 in the same callback), in the same sweep over P: the update only reads the <= 16
                                    * columns var_order of the posterior, which are formed first; its rank-<=16 downdate then rides on
                                    * the MSCKF write-back (one read + one write of P for both).  Windows up to 16 clones, at most 16
                                    * candidate rows and 16 columns, no in-frame landmark stage - otherwise ingvio_frame_run applies
                                    * it as a separate pass.  Results: ingvio_gnss_fetch (dx in the state's index space AFTER the
                                    * frame's marginalisation).  ingvio_gnss_run must not be called for an in-frame stage. */
} ingvio_gnss_opts;
int ingvio_gnss_update_batch(ingvio_ctx* ctx, int b0, int nb, const ingvio_update_block* blocks, const ingvio_gnss_opts* opts,
                             double* dx_out, int* rows_out, int* keep_out, int* status_out);
int ingvio_gnss_stage(ingvio_ctx* ctx, int b0, int nb, const ingvio_update_block* blocks, const ingvio_gnss_opts* opts);
int ingvio_gnss_run(ingvio_ctx* ctx, int b0, int nb);
int ingvio_gnss_fetch(ingvio_ctx* ctx, int b0, int nb, double* dx_out, int* rows_out, int* keep_out, double* gamma_out,
                      int* status_out);
int ingvio_mld(ingvio_ctx* ctx);                      /* row stride of keep_out / gamma_out */

/* ---- the gnss_comm front of the GNSS update on the device ----------------------------------------
 * What GnssUpdate::updateTrackedSys obtains from gnss_comm before it builds its rows (GnssUpdate.cpp:98-122): satellite states
 * from the broadcast ephemerides at transmit time (gnss_comm/src/gnss_spp.cpp:50-98, gnss_utility.cpp:390-640), elevation,
 * Saastamoinen/Niell and Klobuchar delays (:762-899), pseudo-range and Doppler residuals (gnss_spp.cpp:100-146, :256-282).
 * ingvio_gnss_front_stage evaluates all of it for every filter of the range (one lane per satellite) and writes the candidate
 * rows where ingvio_gnss_stage would have put them: follow with ingvio_gnss_run / ingvio_gnss_fetch.  GLONASS satellites take
 * geph2svdt / geph2pos / geph2vel (gnss_utility.cpp:642-731: Runge-Kutta integration of the broadcast PZ-90 state vector).
 * Times are seconds of the GPS week.
 * Flat records of doubles:
 *   ephemeris [INGVIO_EPH_N]: sys (gnss_comm::sys2idx: GPS 0, GLO 1, GAL 2, BDS 3), prn, toe, toe in the constellation's own
 *     week (BDS: BDT of toe - 14 s, gnss_utility.cpp:489; else = toe), toc, A, e, i0, omg, OMG0, M0, delta_n, OMG_dot, i_dot,
 *     cuc, cus, crc, crs, cic, cis, af0, af1, af2, tgd[0], ura
 *     GLONASS (sys == 1, GloEphem): sys, prn, toe (GPS week seconds), 0, 0, pos[3], vel[3], acc[3] (m, m/s, m/s^2, PZ-90 ECEF),
 *     tau_n, gamma, zeros up to ura at [24]; the observation's frequency is the satellite's FDMA channel
 *   observation [INGVIO_OBS_N]: receive time, L1 pseudo-range (m), L1 Doppler (Hz), psr_std, dopp_std, L1 frequency (Hz, < 0: no L1) */
#define INGVIO_EPH_N 25
#define INGVIO_OBS_N 6
#define INGVIO_GNSS_MAX_SAT 64
typedef struct {
    int n_sat;                                /* <= INGVIO_GNSS_MAX_SAT                                                */
    const double* eph;                        /* [n_sat][INGVIO_EPH_N]                                                 */
    const double* obs;                        /* [n_sat][INGVIO_OBS_N]                                                 */
    const double* ion;                        /* [8] Klobuchar parameters (latest_gnss_iono_params) or NULL            */
    double doy;                               /* gnss_comm::time2doy of the epoch                                      */
    double p_w[3], v_w[3];                    /* State::_extended_pose valueTrans1() / valueTrans2()                   */
    double cb[4], fs;                         /* GnssManager::getClockbiasVec (m), FS value (m/s)                      */
    double yaw_offset;                        /* YOF value                                                             */
    double R_enu2ecef[9], anchor_ecef[3];     /* GvioAligner::getRenu2ecef (row-major), translation of getTenu2ecef    */
    int idx_se23, idx_yof, idx_fs, idx_cb[4]; /* Type::idx() of the variables (idx_cb[s] = -1: clock s not in the state) */
    double psr_noise_amp, dopp_noise_amp;     /* GnssUpdate::_psr_noise_amp / _dopp_noise_amp                          */
} ingvio_gnss_epoch;
int ingvio_gnss_front_stage(ingvio_ctx* ctx, int b0, int nb, const ingvio_gnss_epoch* epochs, const ingvio_gnss_opts* opts);
/* per-satellite results of the last front_stage: out [nb][INGVIO_GNSS_MAX_SAT][INGVIO_GNSS_SAT_REC] = res_pos, res_vel, unit
 * receiver->satellite (3), azimuth, elevation, ionosphere delay, troposphere delay, usable (1/0), then the gnss_comm::SatState
 * (gnss_constant.hpp:506-516): pos (3), vel (3), dt, ddt, tgd, transmit time */
#define INGVIO_GNSS_SAT_REC 20
int ingvio_gnss_front_fetch(ingvio_ctx* ctx, int b0, int nb, double* out);
/* The same evaluation for any list of epochs, detached from the filters and from the staged rows (the idx_* / noise fields of
 * the epochs are ignored): the residual evaluator of gnss_comm::psr_pos (gnss_spp.cpp:148-254) and GvioAligner::batchAlign
 * (GvioAligner.cpp:88-383), whose iterations live in the host shim (host/GvioAligner.cpp).  To evaluate at a receiver given in
 * ECEF set R_enu2ecef = I, yaw_offset = 0, p_w = 0, anchor_ecef = position, v_w = ECEF velocity.
 * out [n_epochs][INGVIO_GNSS_MAX_SAT][INGVIO_GNSS_SAT_REC]. */
int ingvio_gnss_sat_eval(ingvio_ctx* ctx, int n_epochs, const ingvio_gnss_epoch* epochs, double* out);

/* ---- feature-sharded single filter (optional mode): the per-feature work (K3-K5 gate, K6/K7 in information form)
 * is independent given the prior, so G replicas of ONE filter can each take F/G of the frame's features:
 *   ingvio_frame_run_phase(ctx, restore, 1)   propagate + clone + gate + Gram of the staged (local) features
 *   ingvio_debug_msckf_info(ctx, b, A, &ncol) the local [A | b] = [sum H_j^T H_j | sum H_j^T r_j]  (35 KB at 11 clones, 260 KB at 30)
 *   -- the ONE exchange step: sum over the replicas (all-reduce; ingvio_amd/parallel.py::sharded_frame_update) --
 *   ingvio_info_set(ctx, b, A_sum, ncol, n_accepted_total)
 *   ingvio_frame_run_phase(ctx, 0, 2)         solve + apply + marginalise: identical posterior on every replica
 * phase 0 = ingvio_frame_run.  Factored method only; the accepted-feature cap (max_accept) is a global order and is not
 * supported across shards (INGVIO_E_ARG when the staged options carry max_accept > 0 and phase != 0).
 * Protocol: phase 1 leaves the filters half-stepped (cloned, n + 6, not yet updated); until phase 2 has run, every entry point
 * that changes or snapshots the covariance — ingvio_frame_run, a second phase 1, ingvio_frame_stage, ingvio_ekf_update, ... —
 * returns INGVIO_E_ARG; reading (ingvio_cov_get, ingvio_debug_msckf_info, ingvio_frame_fetch) and ingvio_info_set are allowed.
 * Phase 2 without a pending phase 1 (or twice) is INGVIO_E_ARG.  ingvio_cov_restore abandons a pending split step. */
int ingvio_frame_run_phase(ingvio_ctx* ctx, int restore_prior, int phase);
int ingvio_info_set(ingvio_ctx* ctx, int b, const double* A, int ncol, int n_accepted);
/* The same exchange WITHOUT a host hop: ingvio_info_reduce sums filter b's chunk partials on the device into one contiguous buffer
 * [A | b | n_accepted] (row-major ncol x (ncol + 1), then the local accepted count as a double; *count = ncol (ncol + 1) + 1) and
 * returns its DEVICE pointer (valid until the context is destroyed; complete when the call returns).  The caller all-reduces (sum)
 * the buffer in place over the replicas - ncclAllReduce on the pointer, or torch.distributed on a zero-copy view
 * (__cuda_array_interface__, ingvio_amd/parallel.py::sharded_frame_update) - and must have that collective finished (stream
 * synchronised) before ingvio_info_commit, which installs the buffer as the filter's information for phase 2. */
int ingvio_info_reduce(ingvio_ctx* ctx, int b, double** dev_ptr, int* count, int* ncol_out);
int ingvio_info_commit(ingvio_ctx* ctx, int b);

/* ---- batched SLAM-landmark update (LandmarkUpdate::updateLandmark{Mono,Stereo}, LandmarkUpdate.cpp:32-149) ----------------
 * For every in-state landmark observed in the current frame: rows against [extended pose | extrinsics | anchor clone | landmark]
 * (calcResJacobianSingleLandmark*, :521-572 / :619-686, as written), the per-landmark chi^2 gate on the prior (:98-99), the
 * accepted rows stacked, one ekfUpdate (:146) - rows, gates, stacking and the update on the device for a range of filters, with
 * S = H P H^T + s^2 I factorised outside LDS (up to INGVIO_LM_MAX landmarks = 256 stereo rows per filter).
 * The nominal values are the host's: it hands over the CURRENT pose / extrinsics / landmark positions and applies dx itself. */
#define INGVIO_LM_MAX 64
typedef struct {
    double R_i2w[9], p_i2w[3];    /* extended pose (row-major rotation, position)                          */
    double R_cl2i[9], p_c2i[3];   /* left camera -> IMU extrinsics                                          */
    int idx_epose, idx_ext;       /* state indices (9 and 6 columns)                                        */
    int n_lm;                     /* <= INGVIO_LM_MAX                                                       */
    const int* lm_idx;            /* [n_lm] state index of each landmark (3 columns)                        */
    const int* anchor_idx;        /* [n_lm] state index of its anchor clone (6 columns)                     */
    const double* pf;             /* [n_lm][3] world position (AnchoredLandmark::valuePosXyz)               */
    const double* uv;             /* [n_lm][4] current observation u0 v0 u1 v1 (mono: first two)            */
    const unsigned char* tracked; /* [n_lm] 1 = observed in this frame (0: the landmark is skipped)         */
} ingvio_landmark_frame;
typedef struct {
    int stereo;
    double noise;                 /* visual noise (sigma)                                                   */
    double chi2_thr;              /* quantile(chi_squared(rows per landmark), 0.95) (Update.cpp:98-100)     */
    double R_cl2cr[9], t_cl2cr[3];
    int in_frame;                 /* 1: ingvio_frame_run performs the update between the MSCKF update and the marginalisation
                                     (IngvioFilter.cpp:296-322 order) for every staged filter               */
} ingvio_landmark_opts;
int ingvio_landmark_stage(ingvio_ctx* ctx, int b0, int nb, const ingvio_landmark_frame* frames, const ingvio_landmark_opts* opts);
int ingvio_landmark_run(ingvio_ctx* ctx, int b0, int nb);
/* dx [nb][ldp] (may be NULL), rows [nb] accepted rows, accept [nb][INGVIO_LM_MAX] 1/0, gamma [nb][INGVIO_LM_MAX] (-1: not
 * tracked), status [nb] (INGVIO_NEG_DIAG etc. as ingvio_ekf_update); any pointer may be NULL */
int ingvio_landmark_fetch(ingvio_ctx* ctx, int b0, int nb, double* dx, int* rows, int* accept, double* gamma, int* status);

/* whitenResidual (Update.cpp:36-79): gamma = res^T (H Pcc H^T + R)^-1 res. */
int ingvio_chi2_gamma(ingvio_ctx* ctx, int b, const int* vidx, const int* vsize, int k,
                      const double* H, int ldh, int m, const double* res,
                      const double* R, int r_kind, double* gamma);

/* Many independent whitenResidual gates against the same prior in one launch and one synchronisation (all SLAM landmarks of
 * a frame, LandmarkUpdate.cpp:98-99; the per-row GNSS gates, GnssUpdate.cpp:190,259).  R = noise_var * I. */
typedef struct {
    const int* vidx;          /* var_order of this block as (idx, size)[k]                           */
    const int* vsize;
    int k;
    const double* H;          /* m x sum(vsize), column-major, leading dimension ldh                   */
    int ldh, m;
    const double* res;        /* [m]                                                                */
} ingvio_gate_block;
int ingvio_chi2_gamma_multi(ingvio_ctx* ctx, int b, int nblk, const ingvio_gate_block* blocks, double noise_var,
                            double* gamma_out);

/* ---- SLAM-landmark covariance operations -----------------------------
 * addVariableDelayedInvertible (StateManager.cpp:461-543): H_old s x sum(vsize) (ldh), H_new s x s (ldn),
 * s <= 6; appends s rows/columns, new_idx receives the new variable's idx (== old N). */
int ingvio_add_variable_delayed_invertible(ingvio_ctx* ctx, int b, const int* vidx, const int* vsize, int k,
                                           const double* H_old, int ldh, const double* H_new, int ldn, int s,
                                           double noise, int* new_idx);
/* addVariableDelayed (StateManager.cpp:549-637): H_old m x sum(vsize), H_new m x s, res [m] (inputs are not
 * modified; the Givens rotations run on the device copy).  chi2_check = quantile(chi_squared(m), 0.95)
 * (boost in the reference, :610-612; the caller supplies it).  *added = 0 when m <= s (:571-575) or when the
 * chi2 test fails (:614-618; state untouched).  On success the variable is appended (new_idx), the EKF update
 * with the remaining m-s rows is applied to the covariance and dx_out [N+s] = K res; the host applies boxPlus. */
int ingvio_add_variable_delayed(ingvio_ctx* ctx, int b, const int* vidx, const int* vsize, int k,
                                const double* H_old, int ldh, const double* H_new, int ldn, int m, int s,
                                const double* res, double noise, double chi2_mult, int do_chi2, double chi2_check,
                                double* dx_out, int* added, int* new_idx, double* chi2_out);
/* replaceVarLinear (StateManager.cpp:639-693): the target variable (tidx, tsize <= 6) becomes H * [dependence
 * variables]: its rows/columns <- P H^T, its diagonal block <- H Pcc H^T.  H tsize x sum(vsize) (ldh). */
int ingvio_replace_var_linear(ingvio_ctx* ctx, int b, int tidx, int tsize, const int* vidx, const int* vsize, int k,
                              const double* H, int ldh);

/* ---- MSCKF visual update: K3-K11 ----------------------------------------------------------
 * One frame of flattened MapServer data for one filter (FeatureInfo/_stereo_obs/_landmark,
 * MapServer.h:69-134, flattened by the host shim).  Clones in ascending timestamp. */
typedef struct {
    int n_clones;                        /* C: clones in the window (State::_sw_camleft_poses)    */
    const int* clone_idx;                /* [C] Type::idx() of each clone                         */
    const double* clone_R;               /* [C][9] valueLinearAsMat(), row-major                  */
    const double* clone_p;               /* [C][3] valueTrans()                                   */
    int n_feat;                          /* F                                                     */
    const double* pf;                    /* [F][3] _landmark->valuePosXyz()                       */
    const int* anchor;                   /* [F] window slot of getAnchoredPose()                  */
    const unsigned long long* obs_mask;  /* [F] bit s: observation at slot s takes part           */
    const double* uv;                    /* [F][C][4] StereoMeas::asVec() (mono: [0..1])          */
    const int* dof;                      /* [F] dof passed to testChiSquared (Q4)                 */
} ingvio_msckf_frame;

typedef struct {
    int stereo;               /* 1: calcResJacobian...StereoObs, 0: ...MonoObs                      */
    double R_cl2cr[9];        /* StateParams::_T_cl2cr linear part, row-major                       */
    double t_cl2cr[3];
    double noise;             /* _visual_noise (sigma)                                              */
    const double* chi2_table; /* chi2_table[d], d = 0..chi2_len-1 (UpdateBase::_chi_squared_table)  */
    int chi2_len;
    int max_accept;           /* RemoveLostUpdate::_max_valid_ids (20); <= 0: no cap                */
    int compress_rule;        /* 0 as_written (RemoveLost keeps all rows, Q2), 1 top_n.  The GPU    */
                              /* always compresses to n rows: the two are the same posterior.       */
    int selected_variant;     /* 0 RemoveLost form, 1 SwMarg/Keyframe form (anchor block assigned,  */
                              /* quirk Q10, SwMargUpdate.cpp:302)                                   */
} ingvio_msckf_opts;

/* RemoveLostUpdate::updateState{Mono,Stereo} (RemoveLostUpdate.cpp:40-167,276-405),
 * SwMargUpdate::updateState* (SwMargUpdate.cpp:42-189,216-365), KeyframeUpdate::updateState*
 * (KeyframeUpdate.cpp:438-735) after triangulation, for filters [b0, b0+nb):
 * Jacobians + nullspace (K3,K4), chi2 gate on the prior (K5), accepted-feature cap, TSQR
 * compression (K6,K7) and the Kalman update (K8-K11).
 * dx_out [nb][ldp]; accepted [nb][f_max] (1/0); gamma [nb][f_max] (may be NULL);
 * rows_out [nb] = rows handed to the Kalman update (0: none accepted, state untouched). */
int ingvio_msckf_update(ingvio_ctx* ctx, int b0, int nb, const ingvio_msckf_frame* frames,
                        const ingvio_msckf_opts* opts, double* dx_out, int* accepted, double* gamma,
                        int* rows_out);

/* Selects how ingvio_msckf_update / ingvio_frame_run evaluate K4-K11 (same posterior to FP64 rounding):
 *   0 dense    — literal: projected blocks H_j, dense S_j, Householder TSQR of the stacked H,
 *                Cholesky Kalman update (kernels_msckf.hip, kernels_ekf.hip)
 *   1 factored — default: exploits H_x = Gblk*D (per-observation 4x3 factors times a sparse
 *                selection), the projector identity for gamma, R^T R accumulated as 6x6 clone-pair
 *                blocks and the information-form update (kernels_factored.hip); see DESIGN.md.
 * Also settable at context creation through the environment: INGVIO_MSCKF_METHOD=dense|factored. */
int ingvio_set_msckf_method(ingvio_ctx* ctx, int method);

/* Stacked-QR compression on its own (the SPQR call sites RemoveLostUpdate.cpp:376-397,
 * SwMargUpdate.cpp:336-357, KeyframeUpdate.cpp:707-728): H m x n (ldh) column-major, res [m] ->
 * H_thin n x n upper triangular (ldt) and r_thin [n] with H_thin^T H_thin = H^T H,
 * H_thin^T r_thin = H^T res.  Any m, n <= 4096.  Two methods (ingvio_set_qr_method): blocked Householder QR (kernels_qr.hip; the
 * oracle's reflector convention, rows above 6144 in chunks) and, by default for tall stacks (n >= 128, m >= 6 n: the 35100 x 180
 * stack of 300 features x 30 clones, the 6000 x 800 stress shape), Cholesky-QR: H_thin = chol([H|res]^T [H|res]) on the matrix
 * cores (kernels_chol.hip), diagonal positive, identical H_thin^T H_thin / H_thin^T r_thin - which is all ekfUpdate reads.
 * Touches no filter state; device buffers and the launch graph are cached per shape. */
/* method: 0 automatic (default), 1 Householder always, 2 Cholesky-QR always */
int ingvio_set_qr_method(ingvio_ctx* ctx, int method);
int ingvio_qr_compress(ingvio_ctx* ctx, const double* H, int ldh, int m, int n, const double* res,
                       double* H_thin, int ldt, double* r_thin);

/* ---- feature triangulation ----------------------------------------------
 * Triangulator::triangulateMonoObs / triangulateStereoObs (Triangulator.cpp:173-318, 320-359) for every feature of
 * frames[0..nb): Levenberg-Marquardt on (x/z, y/z, 1/z) in the frame of the last observation, Huber weights, depth /
 * parallax / convergence gates.  Of a frame only n_clones, clone_R, clone_p, n_feat, obs_mask and uv are used
 * (pf, anchor, dof may be NULL; clone_idx must be valid state indices or 0).  Outputs per filter stride f_max:
 * pf_out [nb][f_max][3] world points (0 where it failed), ok_out [nb][f_max].
 * frames == NULL: triangulate the frames already staged by ingvio_frame_stage / ingvio_msckf_update; the results stay
 * in the staged device frame (with mask_failed != 0 failed features lose their observation mask and drop out), so a
 * following ingvio_frame_run uses them without a host round trip. */
typedef struct {
    int stereo;
    double R_cl2cr[9];          /* row-major */
    double t_cl2cr[3];
    double trans_thres, huber_epsilon, conv_precision, init_damping;   /* Triangulator.h:67-70: 0.1, 0.01, 5e-7, 1e-3 */
    int outer_loop_max_iter, inner_loop_max_iter;                      /* :71-72: 10, 10 */
    double max_depth, min_depth;                                       /* :74-75: 60, 0.2 */
    int mask_failed;
} ingvio_tri_opts;
int ingvio_triangulate(ingvio_ctx* ctx, int b0, int nb, const ingvio_msckf_frame* frames, const ingvio_tri_opts* opts,
                       double* pf_out, int* ok_out);
/* RemoveLostUpdate::updateStateStereo / Mono in ONE device round trip (RemoveLostUpdate.cpp:276-405, :41-170): the reference
 * triangulates every lost feature (MapServerManager.cpp:274-341), drops the ones that fail and updates with the rest.  Here the
 * staged features' points are triangulated on the device from the frame's own observations (frames[].pf is ignored), a feature whose
 * triangulation fails - including a point behind its anchor camera, MapServerManager.cpp:290,325 - drops out of the update on the
 * device (accepted[i] = 0), and the rest is ingvio_msckf_update.  tri_ok[i]: 1 triangulated, 0 failed, 2 triangulated but behind its
 * anchor camera (the reference counts that attempt, MapServerManager.cpp:287 / :322).  pf_out [nb][f_max][3], tri_ok [nb][f_max]
 * (may be NULL).  One upload, one download, one synchronisation: ingvio_triangulate + ingvio_msckf_update cost a single real-time
 * filter two of each and the host a second packing of the same observations (round 5).
 * tri_masks (NULL, or nb pointers of which any may be NULL): the observations the TRIANGULATION of filter i's features uses, one
 * mask per feature over the window slots, when they differ from the update's obs_mask - the selected-stamp updates
 * (SwMargUpdate.cpp:236-257, KeyframeUpdate.cpp:607-628) triangulate from every observation and update with the selected stamps
 * only; frames[i].uv must then carry the measurements of every slot of the triangulation mask. */
int ingvio_msckf_update_tri(ingvio_ctx* ctx, int b0, int nb, const ingvio_msckf_frame* frames, const ingvio_msckf_opts* opts,
                            const ingvio_tri_opts* tri, const unsigned long long* const* tri_masks, double* dx_out, int* accepted,
                            double* gamma, int* rows_out, double* pf_out, int* tri_ok);

/* ---- one benchmark "update" for the whole batch ------------------------------
 * k-step propagation + clone + MSCKF update + marginalise one clone, all filters, no host
 * synchronisation between the stages.  stage() uploads inputs (outside any timed region),
 * run() only enqueues kernels, fetch() synchronises and downloads. */
typedef struct {
    int k;                    /* IMU steps                                                          */
    const double* Phi;        /* [k][225]                                                           */
    const double* G;          /* [k][180]                                                           */
    const double* dt;         /* [k]                                                                */
    int gnss_idx[5];
    double R_i2w[9];          /* IMU rotation at clone time                                         */
    int marg_idx;             /* idx of the clone to marginalise afterwards, -1: none               */
} ingvio_frame_step;

int ingvio_frame_stage(ingvio_ctx* ctx, int b0, int nb, const ingvio_frame_step* steps,
                       const ingvio_msckf_frame* frames, const ingvio_msckf_opts* opts,
                       const double sigma[4], int enable_gnss, double sigma_cb, double sigma_rw);
/* Same as ingvio_frame_stage for the WHOLE batch (b0 = 0, nb = batch), but the inputs travel on a separate copy
 * stream into a second set of device input buffers, so the call may be issued while the previous ingvio_frame_run is
 * still executing: PCIe and host packing of frame i+1 overlap the kernels of frame i.  Pattern per frame:
 * run(i); stage_async(i+1); fetch(i).  The next ingvio_frame_run / ingvio_triangulate(staged) waits for the copy. */
int ingvio_frame_stage_async(ingvio_ctx* ctx, int b0, int nb, const ingvio_frame_step* steps,
                             const ingvio_msckf_frame* frames, const ingvio_msckf_opts* opts,
                             const double sigma[4], int enable_gnss, double sigma_cb, double sigma_rw);
/* ---- device-resident track store: the frame hand-over as a DELTA (round 6) -------------------------------------------------
 * The reference's MapServer gains ONE observation per live feature per camera frame (MapServerManager::collectStereoMeas,
 * MapServerManager.cpp:146-217) and forgets the observations of the clones that leave the window (KeyframeUpdate::
 * cleanStereoObsAtMargTime, KeyframeUpdate.cpp:737-761; SwMargUpdate.cpp:425-446); ingvio_frame_stage re-sends every filter's whole
 * [F][C][4] measurement array and its k transition matrices with every frame (61 KB per update at 150 features x 11 clones, k = 10).
 * With a track store the observations of every live track stay on the device (uv [t_max][c_max][4], a 64-bit observation mask and
 * the triangulated point per track and filter), and a frame travels as
 *   - the delta on the store, applied in this order: window slots that leave (the tracks' rows close up), tracks that were erased,
 *     the new clone's column (one measurement per observed track), points that changed;
 *   - the frame the update uses: the clone table and, per feature, its track, anchor slot, dof and - for the Selected-timestamp
 *     updates (SwMargUpdate.cpp:236-257) - a mask ANDed onto the stored one;
 *   - the raw IMU samples of the frame and the nominal state at its start: ImuPropagator::stateAndCovTransition
 *     (ImuPropagator.cpp:98-162, analytic branch) runs on the device (Phi, G, dt and the clone rotation never cross the bus);
 * about 7.6 KB per update for the same frame.  ingvio_frame_stage_tracks leaves the context exactly as ingvio_frame_stage(_async) does:
 * ingvio_frame_run / ingvio_frame_fetch follow.  Slot and track numbers are the CALLER's bookkeeping (the shim's MapServer); tracks
 * are numbered 0 .. t_max - 1 (t_max <= 65536), anchor slots and dof fit a byte. */
typedef struct {
    int n_drop; const int* drop_slots;            /* window slots leaving the window, ascending (before the append)              */
    int n_free; const int* free_tracks;           /* erased features: the track's observations are forgotten                     */
    int append_slot;                              /* window slot of the frame's new clone, -1: no new column                     */
    int n_obs; const int* obs_track; const double* obs_uv;      /* [n_obs], [n_obs][4] (mono: first two)                            */
    int n_pf; const int* pf_track; const double* pf;            /* [n_pf], [n_pf][3] points that changed                            */
    int n_clones; const int* clone_idx; const double* clone_R; const double* clone_p;      /* as ingvio_msckf_frame                */
    int n_feat; const int* feat_track; const int* feat_anchor; const int* feat_dof;
    const unsigned long long* feat_sel;           /* [n_feat] or NULL: every stored observation of the track takes part          */
} ingvio_track_frame;
typedef struct {
    int k;                        /* IMU steps (the same for every filter of a call)                                              */
    const double* imu;            /* [k][7] gyro (3), accel (3), dt of each step, as ImuPropagator::propagateUntil forms them     */
    double R[9], p[3], v[3], bg[3], ba[3], gravity[3];      /* State::_extended_pose (row-major R), biases, gravity at the frame's start */
    int gnss_idx[5];
    int marg_idx;                 /* idx of the clone to marginalise afterwards, -1: none                                        */
} ingvio_frame_step_raw;
int ingvio_tracks_create(ingvio_ctx* ctx, int t_max);             /* allocates (or clears) the store: every track empty            */
int ingvio_frame_stage_tracks(ingvio_ctx* ctx, int b0, int nb, const ingvio_frame_step_raw* steps, const ingvio_track_frame* frames,
                              const ingvio_msckf_opts* opts, const double sigma[4], int enable_gnss, double sigma_cb, double sigma_rw,
                              int async /* != 0: whole batch, on the copy stream into the second input set, as ingvio_frame_stage_async */);
int ingvio_frame_run(ingvio_ctx* ctx, int restore_prior);
/* Throughput batches (round 6, an experiment kept selectable): ingvio_frame_run deals the batch to `parts` slices of filters, each
 * on its own HIP stream; the slices' throughput-bound segments (gate + Gram, apply) are chained by events so that only ONE runs at a
 * time while the other slices' latency-bound kernels (restore, propagate, solve) execute beside it.  Consecutive ingvio_frame_run
 * calls pipeline (the slices are not joined at the end of the call); every other entry point, ingvio_sync and ingvio_ctx_stream
 * included, first makes the context's stream wait for them.  Per filter the same kernels with the same arguments: results are
 * bit-identical to parts = 1 (tests/test_gpu_alternatives.py).  parts: -1 / 0 / 1 off (the default: on MI355X the kernels are each
 * sized to fill a CU's LDS and registers alone, co-resident kernels halve each other's occupancy and the split loses),
 * 2..4 slices.  Windows up to 16 clones, factored method, no in-frame landmark stage.  The reference has no counterpart (one filter,
 * one thread: IngvioNode.cpp:36).  Environment override at context creation: INGVIO_FRAME_PARTS. */
int ingvio_set_frame_parts(ingvio_ctx* ctx, int parts);
int ingvio_frame_fetch(ingvio_ctx* ctx, int b0, int nb, double* dx_out, int* accepted, int* rows_out);
/* The fetch in two halves (round 6): _begin enqueues the result copies (into pinned memory) behind the frame's kernels and returns;
 * _end waits for them and fills the caller's arrays.  In between the host may stage and LAUNCH the next frame - its kernels queue up
 * behind the copies, the device does not idle while the host unpacks:   run(i); fetch_begin(i); stage_async(i+1); run(i+1); fetch_end(i). */
int ingvio_frame_fetch_begin(ingvio_ctx* ctx, int b0, int nb);
int ingvio_frame_fetch_end(ingvio_ctx* ctx, double* dx_out, int* accepted, int* rows_out);

/* parity hook (tests): the stacked measurement information of filter b's LAST MSCKF update, A_out [ncol][ncol+1] row-major =
 * [sum_j H_j^T H_j | sum_j H_j^T r_j] over the used features in window-slot column order, ncol = 6 * n_clones (caller provides
 * (6 c_max) * (6 c_max + 1) doubles).  Factored method: the gram kernel's partial sums; dense method: R^T R from the TSQR factor. */
int ingvio_debug_msckf_info(ingvio_ctx* ctx, int b, double* A_out, int* ncol_out);

/* parity hook (tests): [M | t] of filter b's last factored update as the solve kernel left it (MP x MP row-major + MP, MP = 6 * window class
 * rounded up to 4); count doubles are copied. */
int ingvio_debug_info_solution(ingvio_ctx* ctx, int b, double* out, int count);

/* debug: shader-clock stamps written by block (0,0) of the instrumented kernels (see dev_common.h) */
int ingvio_debug_read(ingvio_ctx* ctx, long long* out, int n);

/* per-kernel device time, measured with hipEvents on the context's stream.  enable=1 brackets
 * every kernel launch with events (adds launch-side overhead, off by default). */
int ingvio_profile_enable(ingvio_ctx* ctx, int enable);
/* restrict the event pairs to one kernel (name as returned by ingvio_profile_get; NULL or "" = every kernel): the
 * timed region of bench.py brackets only the dominant kernel, an event pair around every launch costs ~6 % */
int ingvio_profile_select(ingvio_ctx* ctx, const char* kernel_name);
int ingvio_profile_reset(ingvio_ctx* ctx);
/* names: array of `cap` char* receiving static strings; ms/calls: [cap]; returns number of entries */
int ingvio_profile_get(ingvio_ctx* ctx, const char** names, double* ms, int* calls, int cap);

#ifdef __cplusplus
}
#endif
#endif
