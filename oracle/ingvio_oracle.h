/*
 * ingvio_oracle.h — CPU restatement of InGVIO's ingvio_estimator covariance hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (ingvio_amd/, libingvio_hip.so,
 * libingvio_host.so) may include, link or call this.  Allowed users: tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg — as the checker / timed CPU baseline, never as the thing shipped.
 *
 * Every function cites the reference file:line (relative to /root/reference/) it restates.
 * The reference itself (C++14 on Eigen 3.3.7 + SuiteSparse SPQR + Boost.Math + ROS1) cannot be
 * built in this image (no Eigen/SuiteSparse/Boost/ROS, no network), so this oracle is pinned by
 *   (1) the algebraic identities of the reference's own gtests (ingvio_estimator/test/
 *       TestStateManager.cpp:31-51,95-137,195-255,396-455,478-557; TestPropagator.cpp:254-258),
 *       re-implemented in tests/test_oracle_identities.py, and
 *   (2) golden vectors produced by an independent numpy/scipy transcription that uses LAPACK
 *       SVD / QR / inverse and scipy.stats.chi2 in place of Eigen::JacobiSVD / SPQR / Boost
 *       (oracle/gen_golden.py -> tests/golden/ npz files).
 * PARITY UNPINNED (no reference test touches them, third-party code absent): the JacobiSVD
 * left-nullspace basis, the SPQR factor and Boost's chi-squared quantile.  The posterior is
 * invariant to the choice of orthonormal nullspace basis / QR sign convention; that invariance is
 * itself tested (tests/test_oracle_golden.py).
 *
 * Conventions: all matrices that cross this API are column-major with an explicit leading
 * dimension (so Eigen::Map is zero-copy), except 3x3 rotations and T_cl2cr which are row-major
 * double[9] (+ translation double[3]).  The covariance P is n x n inside an ld x ld buffer.
 */
#ifndef INGVIO_ORACLE_H
#define INGVIO_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* ---- AuxGammaFunc.cpp:28-225 ------------------------------------------------------------- */
void orc_skew(const double v[3], double M[9]);                       /* :28-35  row-major */
void orc_gamma(const double v[3], int m, double out[9]);             /* :46-113 */
void orc_psi1(const double w[3], const double a[3], double dt, double out[9]); /* :115-166 */
void orc_psi2(const double w[3], const double a[3], double dt, double out[9]); /* :168-225 */

/* ---- retractions: PoseState.cpp:79-88 (SE3), :174-186 (SE23); VecState.cpp:25-45 ---------- */
void orc_se3_update(double R[9], double p[3], const double dx[6]);
void orc_se23_update(double R[9], double p[3], double v[3], const double dx[9]);

/* ---- ImuPropagator.cpp:98-162 (analytic branch).  Phi 15x15, G 15x12 column-major. --------
 * Advances the nominal (R,p,v) in place; clock states are advanced by the caller
 * (cb += dt*fs, :139-148). */
void orc_imu_transition(double R[9], double p[3], double v[3],
                        const double bg[3], const double ba[3],
                        const double gyro[3], const double acc[3],
                        const double gravity[3], double dt,
                        double Phi[225], double G[180]);

/* ---- StateManager.cpp:42-119.  gnss_idx = state idx of {GPS,GLO,GAL,BDS,FS} or -1. --------
 * sigma = {noise_g, noise_a, noise_bg, noise_ba}; sigma_cb/sigma_rw = _noise_clockbias/_noise_cb_rw
 * AFTER quirk Q1 (State.cpp:51-52) has been applied by the caller. */
void orc_propagate_cov(double* P, int n, int ld, const double* Phi, const double* G, double dt,
                       const double sigma[4], int enable_gnss, const int gnss_idx[5],
                       double sigma_cb, double sigma_rw);

/* StateManager.cpp:253-296 (covariance part): appends 6 rows/cols, new idx == n. R_i2w row-major. */
void orc_augment_clone(double* P, int n, int ld, const double R_i2w[9]);
/* StateManager.cpp:155-177 (covariance part). */
void orc_marginalize(double* P, int n, int ld, int idx, int size);
/* StateManager.cpp:194-214. blk is size x size column-major. */
void orc_append_independent(double* P, int n, int ld, int size, const double* blk);
/* StateManager.cpp:128-153. out is ns x ns column-major (ns = sum vsize). */
void orc_marginal_cov(const double* P, int ld, const int* vidx, const int* vsize, int k, double* out);

/* r_kind: 0 = scalar variance (R points to one double, R = *R * I), 1 = diagonal (m doubles),
 * 2 = full m x m column-major. */
/* Update.cpp:36-79: gamma = res^T (H Pcc H^T + R)^-1 res (LDLT). */
double orc_whiten_residual(const double* P, int ld, const int* vidx, const int* vsize, int k,
                           const double* H, int ldh, int m, const double* res,
                           const double* R, int r_kind);
/* StateManager.cpp:359-423 (everything except boxPlus): P <- sym(P - K (P H^T)^T), dx = K res.
 * S is inverted by partial-pivot LU as Eigen's MatrixXd::inverse() does.  Returns 0, or 1 if a
 * negative diagonal appeared (the reference only asserts, :413-421). */
int orc_ekf_update(double* P, int n, int ld, const int* vidx, const int* vsize, int k,
                   const double* H, int ldh, int m, const double* res,
                   const double* R, int r_kind, double* dx);

/* ---- MSCKF visual update -------------------------------------------------------------------
 * One call = RemoveLostUpdate::updateState{Mono,Stereo} (RemoveLostUpdate.cpp:40-167,276-405)
 * or the selected-timestamp twins (SwMargUpdate.cpp:42-189,216-365; KeyframeUpdate.cpp:438-735),
 * after triangulation, on flattened inputs.  Window clones are listed in ascending timestamp. */
typedef struct {
    int n_clones;                         /* C: clones in the sliding window                 */
    const int* clone_idx;                 /* [C] state idx of each clone (Type::idx())       */
    const double* clone_R;                /* [C][9] R_c2w row-major (valueLinearAsMat)       */
    const double* clone_p;                /* [C][3] p_c in world (valueTrans)                */
    int n_feat;                           /* F                                               */
    const double* pf;                     /* [F][3] triangulated p_f in world                */
    const int* anchor;                    /* [F] window slot of the anchor clone             */
    const unsigned long long* obs_mask;   /* [F] bit s set: feature observed at slot s AND the
                                             slot takes part in this update (all obs for
                                             RemoveLost; the selected stamps for SwMarg/Kf)  */
    const double* uv;                     /* [F][C][4] (u0,v0,u1,v1); mono uses [0..1]       */
    const int* dof;                       /* [F] chi2 dof handed to testChiSquared (Q4)      */
    int stereo;                           /* 1 stereo, 0 mono                                */
    double R_cl2cr[9];                    /* row-major                                       */
    double t_cl2cr[3];
    double noise;                         /* _visual_noise (sigma, not variance)             */
    const double* chi2_table;             /* chi2_table[d] = quantile(d), d = 0..chi2_len-1  */
    int chi2_len;
    int max_accept;                       /* RemoveLostUpdate.h:38 = 20; <=0 means no cap    */
    int compress_rule;                    /* 0 as_written: keep all m rows after Q^T (Q2);
                                             1 top_n: keep n rows (SwMargUpdate.cpp:350-351) */
    int selected_variant;                 /* 0 RemoveLost form (anchor theta written inside
                                             H_pf2x); 1 SwMarg/Keyframe form (anchor block
                                             ASSIGNED afterwards -> quirk Q10)               */
} orc_msckf_in;

/* dx has n entries (zero if no rows).  accepted[F] gets 1/0 (accepted by chi2 AND within the
 * max_accept cap), gamma[F] the Mahalanobis value (NaN if not evaluated because of the cap `break`).
 * Returns number of stacked rows m' passed to ekfUpdate (0 => no update). */
int orc_msckf_update(double* P, int n, int ld, const orc_msckf_in* in,
                     double* dx, int* accepted, double* gamma);

/* Per-feature block only (RemoveLostUpdate.cpp:407-523 / :169-273 / SwMargUpdate.cpp:499-700):
 * Hj (rho x 6C, global slot columns, column-major ld=rho_max) and rj; returns rho. For tests. */
int orc_msckf_feature_block(const orc_msckf_in* in, int j, double* Hj, int ldh, double* rj);

/* ---- GNSS rows (GnssUpdate.cpp:148-272), inputs are the OUTPUTS of gnss_comm psr_res/dopp_res.
 * Builds H (rows x ncols col-major, ld = 2*nsat), res, Rdiag and the var_order
 * [SE23, YOF, clocks in first-seen order, FS] as (vidx,vsize); returns rows.  Per-row chi2 gating
 * (:190,:259) uses P when chi2_test != 0. */
typedef struct {
    int nsat;
    const double* los;        /* [nsat][3] unit_rv2sv (ECEF)                                  */
    const int* sys;           /* [nsat] 0 GPS,1 GLO,2 GAL,3 BDS                               */
    const double* res_pos;    /* [nsat] psr_res output                                        */
    const double* res_vel;    /* [nsat] dopp_res output                                       */
    const double* sin_el;     /* [nsat]                                                       */
    const double* ura;        /* [nsat]                                                       */
    const double* psr_std;    /* [nsat]                                                       */
    const double* dopp_std_mps; /* [nsat] dopp_std*LIGHT_SPEED/f_L1                           */
    double R_w2ecef[9];       /* row-major                                                    */
    double p_w[3], v_w[3];    /* SE23 trans1 / trans2                                         */
    int idx_se23, idx_yof, idx_fs;
    int idx_cb[4];            /* -1 if that constellation's clock is not in the state         */
    double psr_amp, dopp_amp;
    int chi2_test;
    const double* chi2_table; int chi2_len;
} orc_gnss_in;
int orc_gnss_rows(const double* P, int ld, const orc_gnss_in* in,
                  double* H, int ldh, double* res, double* Rdiag,
                  int* vidx, int* vsize, int* nvar);

/* ---- one benchmark "update" (SURVEY.md 8d): k propagate steps + clone + MSCKF + marginalise --*/
typedef struct {
    int k;                    /* IMU steps                                                    */
    const double* Phi;        /* [k][225]                                                     */
    const double* G;          /* [k][180]                                                     */
    const double* dt;         /* [k]                                                          */
    double sigma[4];
    int enable_gnss; int gnss_idx[5]; double sigma_cb, sigma_rw;
    double R_i2w[9];          /* IMU rotation at clone time                                   */
    int marg_idx;             /* idx of the clone marginalised afterwards (-1: none)          */
} orc_frame_in;
int orc_frame_update(double* P, int* n, int ld, const orc_frame_in* fr, const orc_msckf_in* ms,
                     double* dx, int* accepted, double* gamma);

/* Same for B independent filters, `threads` OpenMP threads (one filter per thread).
 * P is [B][ld*ld]; n_io[B]; frames/ms are arrays of B structs; dx [B][ld]; accepted [B][Fmax]. */
void orc_frame_update_batch(int B, int threads, double* P, int* n_io, int ld,
                            const orc_frame_in* fr, const orc_msckf_in* ms,
                            double* dx, int* accepted, int fmax);

/* dense helpers exposed for tests / the QR micro-benchmark */
/* Householder QR of A (m x n, col-major, lda) in place: A <- Q^T A (upper trapezoid + zeros),
 * b <- Q^T b. */
void orc_qr_compress(double* A, int m, int n, int lda, double* b);

/* ---- "next" row f-1: feature triangulation (Triangulator.cpp) ------------------------------------ */
typedef struct {
    int C;                       /* window slots                                                          */
    const double* clone_R;       /* [C][9] row-major R_cam-left -> world                                  */
    const double* clone_p;       /* [C][3]                                                                */
    unsigned long long mask;     /* slots with an observation, ascending slot = ascending timestamp       */
    const double* uv;            /* [C][4]: u0 v0 (u1 v1)                                                 */
    int stereo;
    double R_lr[9], t_lr[3];     /* T_cl2cr                                                               */
    double trans_thres, huber_epsilon, conv_precision, init_damping;   /* Triangulator.h:67-70           */
    int outer_loop_max_iter, inner_loop_max_iter;                      /* :71-72                         */
    double max_depth, min_depth;                                       /* :74-75                         */
} orc_tri_in;
/* Triangulator::triangulateMonoObs (:173-318) / triangulateStereoObs (:320-359): LM on (x/z, y/z, 1/z) in the frame
 * of the LAST mono-equivalent observation.  Returns 1 and the world point, or 0 (pf = 0; the reference leaves pf
 * untouched on its early "translation too small" exit, callers ignore it on failure). */
int orc_triangulate(const orc_tri_in* in, double pf[3]);

/* ---- SLAM-landmark path ---------------------------------------------
 * StateManager::addVariableDelayedInvertible (StateManager.cpp:461-543): appends s rows/cols to P (n -> n+s),
 * H_old s x sum(vsize) (ldh), H_new s x s (ldn). */
void orc_add_variable_delayed_invertible(double* P, int n, int ld, const int* vidx, const int* vsize, int k,
                                         const double* H_old, int ldh, const double* H_new, int ldn, int s, double noise);
/* StateManager::addVariableDelayed (:549-637): Givens QR of H_new (m x s) applied to H_old / res IN PLACE, chi2 of the
 * lower m-s rows against chi2_mult * chi2_check (chi2_check = quantile(chi_squared(m), 0.95), supplied by the caller),
 * invertible add with the top s rows, EKF update with the rest (dx [n+s], boxPlus is the caller's).
 * Returns 1 (added, *n_io += s) or 0 (rejected / m <= s). */
int orc_add_variable_delayed(double* P, int* n_io, int ld, const int* vidx, const int* vsize, int k,
                             double* H_old, int ldh, double* H_new, int ldn, int m, int s, double* res,
                             double noise, double chi2_mult, int do_chi2, double chi2_check, double* dx, double* chi2_out);
/* StateManager::replaceVarLinear (:639-693): rows/cols of the target variable <- P H^T, diagonal block <- H Pcc H^T. */
void orc_replace_var_linear(double* P, int n, int ld, int tidx, int tsize, const int* vidx, const int* vsize, int k,
                            const double* H, int ldh);
/* LandmarkUpdate::calcResJacobianSingleLandmark{Mono,Stereo} (LandmarkUpdate.cpp:521-572, 619-686):
 * H rows x 24 column-major (ld 4) = [extended pose 9 | extrinsics 6 | anchor 6 | landmark 3]; returns rows. */
int orc_landmark_rows_epose(const double R_i2w[9], const double p_i2w[3], const double R_cl2i[9], const double p_c2i[3],
                            const double pf[3], const double* uv, int stereo, const double R_lr[9], const double t_lr[3],
                            double* H, double* res);
/* the sliding-window-pose form (:574-617): H rows x 15 (ld 4) = [current clone 6 | anchor 6 | landmark 3]. */
int orc_landmark_rows_sw(const double R_cm[9], const double p_cm[3], const double pf[3], const double* uv, int stereo,
                         const double R_lr[9], const double t_lr[3], int curr_is_anchor, double* H, double* res);

/* ---- the gnss_comm front of the GNSS update (oracle/gnss_front_oracle.c) ----------------------
 * Flat records (doubles): a broadcast Kepler ephemeris, one L1 observation, a satellite state. */
enum { ORC_EPH_SYS = 0, ORC_EPH_PRN, ORC_EPH_TOE, ORC_EPH_TOE_SYS, ORC_EPH_TOC, ORC_EPH_A, ORC_EPH_E, ORC_EPH_I0, ORC_EPH_OMG,
       ORC_EPH_OMG0, ORC_EPH_M0, ORC_EPH_DELTA_N, ORC_EPH_OMG_DOT, ORC_EPH_I_DOT, ORC_EPH_CUC, ORC_EPH_CUS, ORC_EPH_CRC,
       ORC_EPH_CRS, ORC_EPH_CIC, ORC_EPH_CIS, ORC_EPH_AF0, ORC_EPH_AF1, ORC_EPH_AF2, ORC_EPH_TGD, ORC_EPH_URA, ORC_EPH_N };
enum { ORC_OBS_TOW = 0, ORC_OBS_PSR, ORC_OBS_DOPP, ORC_OBS_PSR_STD, ORC_OBS_DOPP_STD, ORC_OBS_FREQ, ORC_OBS_N };
enum { ORC_SAT_N = 10 };      /* pos 3, vel 3, dt, ddt, tgd, ttx */
int orc_gnss_sat_state(const double* eph, const double* obs, double* sat);
void orc_gnss_ecef2geo(const double xyz[3], double lla[3]);
void orc_gnss_azel(const double rcv[3], const double sat[3], double azel[2]);
double orc_gnss_trop(double doy, const double lla[3], const double azel[2]);
double orc_gnss_iono(double tow, const double ion[8], const double lla[3], const double azel[2]);
void orc_gnss_residuals(int ns, const double* eph, const double* obs, const double ion[8], int have_ion, double doy,
                        const double rcv_xyzt[7], const double rcv_vel[4], double* res_pos, double* res_vel, double* los,
                        double* azel_out, double* atmos, double* sat_out, int* usable);

#ifdef __cplusplus
}
#endif

#endif
