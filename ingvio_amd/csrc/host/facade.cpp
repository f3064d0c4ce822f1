// facade.cpp — small extern "C" surface of libingvio_host.so for the Python harness (bench.py,
// tests): lets Python drive the C++ host shim without a C++ test runner.  Not part of the HIP ABI.
#include "GnssUpdate.h"
#include "GvioAligner.h"
#include "ImuTransition.h"
#include "Update.h"

extern "C" {

// ImuPropagator::stateAndCovTransition (ImuPropagator.cpp:98-162), analytic branch.
// R row-major 3x3; Phi/G column-major.
void ingvio_host_imu_transition(double* R, double* p, double* v, const double* bg, const double* ba,
                                const double* gyro, const double* acc, const double* gravity, double dt,
                                double* Phi, double* G)
{
    ingvio::Mat3d Rm(R);
    ingvio::Vec3d pv(p), vv(v);
    ingvio::imuTransitionAnalytic(Rm, pv, vv, ingvio::Vec3d(bg), ingvio::Vec3d(ba), ingvio::Vec3d(gyro),
                                  ingvio::Vec3d(acc), ingvio::Vec3d(gravity), dt, Phi, G);
    for (int i = 0; i < 9; ++i) R[i] = Rm.m[i];
    for (int i = 0; i < 3; ++i) { p[i] = pv[i]; v[i] = vv[i]; }
}

// The same for the isAnalytic == false branch (ImuPropagator.cpp:163-229).
void ingvio_host_imu_transition_rk4(double* R, double* p, double* v, const double* bg, const double* ba,
                                    const double* gyro, const double* acc, const double* gravity, double dt,
                                    double* Phi, double* G)
{
    ingvio::Mat3d Rm(R);
    ingvio::Vec3d pv(p), vv(v);
    ingvio::imuTransitionRK4(Rm, pv, vv, ingvio::Vec3d(bg), ingvio::Vec3d(ba), ingvio::Vec3d(gyro),
                             ingvio::Vec3d(acc), ingvio::Vec3d(gravity), dt, Phi, G);
    for (int i = 0; i < 9; ++i) R[i] = Rm.m[i];
    for (int i = 0; i < 3; ++i) { p[i] = pv[i]; v[i] = vv[i]; }
}

void ingvio_host_gamma(const double* vec, int m, double* out)
{
    const ingvio::Mat3d g = ingvio::GammaFunc(ingvio::Vec3d(vec), m);
    for (int i = 0; i < 9; ++i) out[i] = g.m[i];
}

// GnssUpdate::updateTrackedSys, row assembly (GnssUpdate.cpp:148-272) without the gates: los [nsat][3] (= -J_pos_ecef rows),
// sys [nsat] (gnss_comm::sys2idx), idx_cb [4] (-1: constellation not in the state).  H column-major [ldh][15].
int ingvio_host_gnss_rows(int nsat, const double* los, const int* sys, const double* res_pos, const double* res_vel, const double* sin_el,
                          const double* ura, const double* psr_std, const double* dopp_std_mps, const double* R_w2ecef,
                          const double* p_w, const double* v_w, int idx_se23, int idx_yof, const int* idx_cb, int idx_fs,
                          double psr_amp, double dopp_amp, double* H, int ldh, double* res, double* Rd, int* vidx, int* vsize, int* nvar)
{
    ingvio::GnssResiduals g;
    for (int i = 0; i < nsat; ++i) {
        g.unit_rv2sv.push_back(ingvio::Vec3d(los + 3 * i)); g.sys.push_back(sys[i]); g.res_pos.push_back(res_pos[i]);
        g.res_vel.push_back(res_vel[i]); g.sin_el.push_back(sin_el[i]); g.ura.push_back(ura[i]); g.psr_std.push_back(psr_std[i]);
        g.dopp_std_mps.push_back(dopp_std_mps[i]);
    }
    g.R_w2ecef = ingvio::Mat3d(R_w2ecef);
    return ingvio::gnssCandidateRows(g, ingvio::Vec3d(p_w), ingvio::Vec3d(v_w), idx_se23, idx_yof, idx_cb, idx_fs, psr_amp, dopp_amp,
                                     H, ldh, res, Rd, vidx, vsize, nvar);
}

// The same with is_adjust_yof = 1 (GnssUpdate.cpp:164-167, 239-242): the yaw-offset column -u^T Renu2ecef dotRw2enu(yo) p | v,
// GnssManager::dotRw2enu (GnssManager.cpp:101-113) evaluated at `yaw_offset`; R_w2ecef must be Renu2ecef * calcRw2enu(yaw_offset).
int ingvio_host_gnss_rows_yof(int nsat, const double* los, const int* sys, const double* res_pos, const double* res_vel, const double* sin_el,
                              const double* ura, const double* psr_std, const double* dopp_std_mps, const double* R_w2ecef,
                              const double* R_enu2ecef, double yaw_offset, const double* p_w, const double* v_w, int idx_se23, int idx_yof,
                              const int* idx_cb, int idx_fs, double psr_amp, double dopp_amp, double* H, int ldh, double* res, double* Rd,
                              int* vidx, int* vsize, int* nvar)
{
    ingvio::GnssResiduals g;
    for (int i = 0; i < nsat; ++i) {
        g.unit_rv2sv.push_back(ingvio::Vec3d(los + 3 * i)); g.sys.push_back(sys[i]); g.res_pos.push_back(res_pos[i]);
        g.res_vel.push_back(res_vel[i]); g.sin_el.push_back(sin_el[i]); g.ura.push_back(ura[i]); g.psr_std.push_back(psr_std[i]);
        g.dopp_std_mps.push_back(dopp_std_mps[i]);
    }
    g.R_w2ecef = ingvio::Mat3d(R_w2ecef);
    g.R_enu2ecef = ingvio::Mat3d(R_enu2ecef);
    g.dRw2ecef_dyof = g.R_enu2ecef * ingvio::dotRw2enu(yaw_offset);
    g.has_yof_jac = true;
    return ingvio::gnssCandidateRows(g, ingvio::Vec3d(p_w), ingvio::Vec3d(v_w), idx_se23, idx_yof, idx_cb, idx_fs, psr_amp, dopp_amp,
                                     H, ldh, res, Rd, vidx, vsize, nvar, true);
}

// GvioAligner::batchAlign (GvioAligner.cpp:88-197) driven from Python: n_epochs raw epochs (flat records as ingvio_gnss_epoch wants
// them, at most smax satellites each) with the VIO position / velocity of each; the call that finds the buffer full
// (n_epochs = batch_size + 1) runs the alignment.  out[22] = aligned, yaw offset, refined anchor (3), R_enu2ecef row-major (9),
// receiver clock drift of the yaw stage, rough anchor xyzt (7).
int ingvio_host_aligner_run(void* ctx, int n_epochs, int smax, const double* eph, const double* obs, const int* nsat, const double* doy,
                            const double* p_w, const double* v_w, const double* iono, int batch_size, int max_iter, double conv_epsilon,
                            double vel_thres, double* out)
{
    ingvio::GvioAligner al((ingvio_ctx*)ctx, batch_size, max_iter, conv_epsilon, vel_thres);
    std::vector<double> ion;
    if (iono) ion.assign(iono, iono + 8);
    for (int i = 0; i < n_epochs; ++i) {
        ingvio::RawGnssEpoch m;
        m.eph.assign(eph + (size_t)i * smax * INGVIO_EPH_N, eph + ((size_t)i * smax + nsat[i]) * INGVIO_EPH_N);
        m.obs.assign(obs + (size_t)i * smax * INGVIO_OBS_N, obs + ((size_t)i * smax + nsat[i]) * INGVIO_OBS_N);
        m.doy = doy[i];
        al.batchAlign(m, ingvio::Vec3d(p_w + 3 * i), ingvio::Vec3d(v_w + 3 * i), ion);
    }
    const ingvio::GvioAlignment a = al.alignment();
    out[0] = a.aligned ? 1.0 : 0.0; out[1] = a.yaw_offset;
    for (int c = 0; c < 3; ++c) out[2 + c] = a.anchor_ecef[c];
    for (int c = 0; c < 9; ++c) out[5 + c] = a.R_enu2ecef.m[c];
    out[14] = al.lastRcvDdt();
    for (int c = 0; c < 7; ++c) out[15 + c] = al.lastRoughAnchor()[c];
    return a.aligned ? 1 : 0;
}

// gnss_comm::psr_pos + dopp_vel of ONE epoch through the shim (what GnssProcessor::callbackGnssMeas buffers as the SPP fix,
// GnssProcessor.cpp:196-217): out[11] = xyzt (7), velocity + drift (4); returns 1 when both solved.
int ingvio_host_spp(void* ctx, int nsat, const double* eph, const double* obs, double doy, const double* iono, double* out)
{
    ingvio::GvioAligner al((ingvio_ctx*)ctx);
    if (iono) al.setIono(std::vector<double>(iono, iono + 8));
    ingvio::RawGnssEpoch m;
    m.eph.assign(eph, eph + (size_t)nsat * INGVIO_EPH_N); m.obs.assign(obs, obs + (size_t)nsat * INGVIO_OBS_N); m.doy = doy;
    std::vector<const ingvio::RawGnssEpoch*> one{ &m };
    if (!al.psrPos(one, out)) return 0;
    return al.doppVel(m, out, out + 7) ? 1 : 0;
}

double ingvio_host_chi2_quantile(int dof, double p) { return ingvio::chi2Quantile(dof, p); }

}  // extern "C"
