#include "GvioAligner.h"

#include <cmath>
#include <cstring>
#include <iostream>

#include "GnssComm.h"
#include "PoseState.h"

namespace ingvio {

namespace {

// dense symmetric solve  x = -(G^T W G)^-1 G^T W b  through Gauss-Jordan with partial pivoting (the reference's .inverse(), n <= 7)
bool solveNormal(int n, std::vector<double>& N, std::vector<double>& g, double* x)
{
    std::vector<double> a((size_t)n * (n + 1));
    for (int i = 0; i < n; ++i) { for (int j = 0; j < n; ++j) a[i * (n + 1) + j] = N[i * n + j]; a[i * (n + 1) + n] = -g[i]; }
    for (int k = 0; k < n; ++k) {
        int p = k;
        for (int i = k + 1; i < n; ++i) if (std::fabs(a[i * (n + 1) + k]) > std::fabs(a[p * (n + 1) + k])) p = i;
        if (!(std::fabs(a[p * (n + 1) + k]) > 0.0)) return false;
        if (p != k) for (int j = 0; j <= n; ++j) std::swap(a[k * (n + 1) + j], a[p * (n + 1) + j]);
        const double inv = 1.0 / a[k * (n + 1) + k];
        for (int j = k; j <= n; ++j) a[k * (n + 1) + j] *= inv;
        for (int i = 0; i < n; ++i) {
            if (i == k) continue;
            const double f = a[i * (n + 1) + k];
            if (f != 0.0) for (int j = k; j <= n; ++j) a[i * (n + 1) + j] -= f * a[k * (n + 1) + j];
        }
    }
    for (int i = 0; i < n; ++i) x[i] = a[i * (n + 1) + n];
    return true;
}

Mat3d rotZ(double yaw)
{
    Mat3d R = Mat3d::Identity();
    R(0, 0) = std::cos(yaw); R(0, 1) = -std::sin(yaw); R(1, 0) = std::sin(yaw); R(1, 1) = std::cos(yaw);
    return R;
}

void fillEpoch(ingvio_gnss_epoch& e, const RawGnssEpoch& m, const double* ion)
{
    std::memset(&e, 0, sizeof e);
    e.n_sat = m.n_sat(); e.eph = m.eph.data(); e.obs = m.obs.data(); e.ion = ion; e.doy = m.doy;
    const Mat3d I = Mat3d::Identity();
    std::memcpy(e.R_enu2ecef, I.m, sizeof I.m);
    for (int s = 0; s < 4; ++s) e.idx_cb[s] = -1;
    e.psr_noise_amp = e.dopp_noise_amp = 1.0;
}

constexpr int REC = INGVIO_GNSS_SAT_REC, SMAX = INGVIO_GNSS_MAX_SAT;
constexpr double CUT_OFF_DEGREE = 15.0, EPSILON_PVT = 1e-8;      // gnss_spp.cpp:25, gnss_constant.hpp:220-221
constexpr int MAX_ITER_PVT = 30;

}  // namespace

Mat3d GvioAligner::getRw2enu() const { return rotZ(_yaw_offset); }

GvioAlignment GvioAligner::alignment() const
{
    GvioAlignment a;
    a.aligned = _isAligned; a.yaw_offset = _yaw_offset; a.R_enu2ecef = _T_enu2ecef.R; a.anchor_ecef = _T_enu2ecef.t;
    return a;
}

void GvioAligner::reset()
{
    _align_buffer.clear();
    _isAligned = false;
}

bool GvioAligner::evalEpochs(const std::vector<ingvio_gnss_epoch>& eps, std::vector<double>& rec)
{
    rec.assign((size_t)eps.size() * SMAX * REC, 0.0);
    const int rc = ingvio_gnss_sat_eval(_ctx, (int)eps.size(), eps.data(), rec.data());
    if (rc != INGVIO_OK) {
        std::cout << "[GvioAligner]: device evaluation failed (" << rc << "): " << ingvio_last_error(_ctx) << std::endl;
        return false;
    }
    return true;
}

// gnss_comm::psr_pos (gnss_spp.cpp:148-254): weighted Gauss-Newton on (ecef xyz, four clock biases) from xyzt = 0, elevation
// mask 15 degrees, weight sin^2(el) / (psr_std / 0.16) / (ura - 1 | ura - 2 | 4), a pseudo-measurement of weight 1000 for every
// constellation without an observation.  The epochs share the receiver state (coarseLocalization stacks all of them).
bool GvioAligner::psrPos(const std::vector<const RawGnssEpoch*>& epochs, double result[7])
{
    for (int i = 0; i < 7; ++i) result[i] = 0.0;
    int n_valid = 0, sys_mask[4] = { 0, 0, 0, 0 };
    for (const RawGnssEpoch* m : epochs)
        for (int i = 0; i < m->n_sat(); ++i) {
            if (m->obs[(size_t)i * INGVIO_OBS_N + 5] < 0) continue;                       // filter_L1
            const int sys = (int)m->eph[(size_t)i * INGVIO_EPH_N];
            if (sys < 0 || sys > 3) continue;
            ++n_valid; sys_mask[sys] = 1;
        }
    if (n_valid < 4) return false;
    double xyzt[7] = { 0, 0, 0, 0, 0, 0, 0 }, dx_norm = 1.0;
    int num_iter = 0;
    std::vector<ingvio_gnss_epoch> eps(epochs.size());
    std::vector<double> rec;
    while (num_iter < MAX_ITER_PVT && dx_norm > EPSILON_PVT) {
        for (size_t q = 0; q < epochs.size(); ++q) {
            fillEpoch(eps[q], *epochs[q], _iono_params.size() == 8 ? _iono_params.data() : nullptr);
            std::memcpy(eps[q].anchor_ecef, xyzt, 24);
            std::memcpy(eps[q].cb, xyzt + 3, 32);
        }
        if (!evalEpochs(eps, rec)) return false;
        std::vector<double> N(49, 0.0), g(7, 0.0);
        for (size_t q = 0; q < epochs.size(); ++q)
            for (int i = 0; i < epochs[q]->n_sat(); ++i) {
                const double* r = rec.data() + ((size_t)q * SMAX + i) * REC;
                if (r[9] == 0.0) continue;                                                  // res not computed
                if (!(r[6] > CUT_OFF_DEGREE / 180.0 * M_PI)) continue;
                const int sys = (int)epochs[q]->eph[(size_t)i * INGVIO_EPH_N];
                const double ura = epochs[q]->eph[(size_t)i * INGVIO_EPH_N + 24], pstd = epochs[q]->obs[(size_t)i * INGVIO_OBS_N + 3];
                const double sin_el = std::sin(r[6]);
                double w = sin_el * sin_el;
                if (pstd > 0) w /= (pstd / 0.16);
                if (sys == 0 || sys == 3) w /= ura - 1; else if (sys == 2) w /= ura - 2; else if (sys == 1) w /= 4;
                double G[7] = { -r[2], -r[3], -r[4], 0, 0, 0, 0 };
                G[3 + sys] = 1.0;
                for (int a = 0; a < 7; ++a) { g[a] += G[a] * w * r[0]; for (int c = 0; c < 7; ++c) N[a * 7 + c] += G[a] * w * G[c]; }
            }
        for (int k = 0; k < 4; ++k) if (!sys_mask[k]) N[(3 + k) * 7 + 3 + k] += 1000.0;   // extra clock constraint, b = 0
        double dx[7];
        if (!solveNormal(7, N, g, dx)) return false;
        dx_norm = 0.0;
        for (int a = 0; a < 7; ++a) { xyzt[a] += dx[a]; dx_norm += dx[a] * dx[a]; }
        dx_norm = std::sqrt(dx_norm);
        ++num_iter;
    }
    if (num_iter == MAX_ITER_PVT) return false;
    for (int i = 0; i < 7; ++i) result[i] = xyzt[i];
    return true;
}

// gnss_comm::dopp_vel (gnss_spp.cpp:284-380): weighted Gauss-Newton on (ecef velocity, clock drift), elevation from the reference
// position, weight sin^2(el) / (dopp_std / 0.256) / (ura - 1 | ura - 2 | 2).  (The residual is linear in the unknowns: the second
// iteration only confirms convergence.)
bool GvioAligner::doppVel(const RawGnssEpoch& m, const double ref_ecef[3], double out[4])
{
    for (int i = 0; i < 4; ++i) out[i] = 0.0;
    int n_valid = 0;
    for (int i = 0; i < m.n_sat(); ++i) if (m.obs[(size_t)i * INGVIO_OBS_N + 5] >= 0) ++n_valid;
    if (n_valid < 4) return false;
    double x[4] = { 0, 0, 0, 0 }, dx_norm = 1.0;
    int num_iter = 0;
    std::vector<ingvio_gnss_epoch> eps(1);
    std::vector<double> rec;
    while (num_iter < MAX_ITER_PVT && dx_norm > EPSILON_PVT) {
        fillEpoch(eps[0], m, nullptr);
        std::memcpy(eps[0].anchor_ecef, ref_ecef, 24);
        std::memcpy(eps[0].v_w, x, 24);
        eps[0].fs = x[3];
        if (!evalEpochs(eps, rec)) return false;
        std::vector<double> N(16, 0.0), g(4, 0.0);
        for (int i = 0; i < m.n_sat(); ++i) {
            const double* r = rec.data() + (size_t)i * REC;
            if (r[9] == 0.0 || !(r[6] > CUT_OFF_DEGREE / 180.0 * M_PI)) continue;
            const int sys = (int)m.eph[(size_t)i * INGVIO_EPH_N];
            const double ura = m.eph[(size_t)i * INGVIO_EPH_N + 24], dstd = m.obs[(size_t)i * INGVIO_OBS_N + 4];
            const double sin_el = std::sin(r[6]);
            double w = sin_el * sin_el;
            if (dstd > 0) w /= (dstd / 0.256);
            if (sys == 0 || sys == 3) w /= ura - 1; else if (sys == 2) w /= ura - 2; else if (sys == 1) w /= 2;
            const double G[4] = { -r[2], -r[3], -r[4], 1.0 };
            for (int a = 0; a < 4; ++a) { g[a] += G[a] * w * r[1]; for (int c = 0; c < 4; ++c) N[a * 4 + c] += G[a] * w * G[c]; }
        }
        double dx[4];
        if (!solveNormal(4, N, g, dx)) return false;
        dx_norm = 0.0;
        for (int a = 0; a < 4; ++a) { x[a] += dx[a]; dx_norm += dx[a] * dx[a]; }
        dx_norm = std::sqrt(dx_norm);
        ++num_iter;
    }
    for (int i = 0; i < 4; ++i) out[i] = x[i];
    return true;
}

void GvioAligner::batchAlign(const RawGnssEpoch& gnss_meas, const std::shared_ptr<SE23> epose, const std::vector<double>& iono)
{
    batchAlign(gnss_meas, epose->valueTrans1(), epose->valueTrans2(), iono);
}

void GvioAligner::batchAlign(const RawGnssEpoch& gnss_meas, const Vec3d& p_w, const Vec3d& v_w, const std::vector<double>& iono)
{
    if (_isAligned) return;
    _iono_params = iono;
    if ((int)_align_buffer.size() < _batch_size) {                                   // :94-98
        _align_buffer.push_back(Item{ p_w, v_w, gnss_meas });
        return;
    }
    double hv[2] = { 0.0, 0.0 };                                                      // :101-110 horizontal excitation
    for (const Item& it : _align_buffer) { hv[0] += std::fabs(it.v[0]); hv[1] += std::fabs(it.v[1]); }
    hv[0] /= _align_buffer.size(); hv[1] /= _align_buffer.size();
    if (std::sqrt(hv[0] * hv[0] + hv[1] * hv[1]) <= _vel_thres) {
        std::cout << "[GvioAligner]: Horizontal velocity excitation not enough, waiting and restart ..." << std::endl;
        reset();
        return;
    }
    std::cout << "[GvioAligner]: Start batch alignment ..." << std::endl;
    double rough[7], refined[7], yaw = 0.0, ddt = 0.0;
    if (!coarseLocalization(rough) || !yawAlignment(rough, yaw, ddt) || !anchorRefinement(yaw, ddt, rough, refined)) { reset(); return; }
    _align_buffer.clear();
    _isAligned = true;
    _T_enu2ecef.R = gnss::geo2rotation(gnss::ecef2geo(Vec3d(refined)));              // ecef2rotation, gnss_utility.cpp:757-760
    _T_enu2ecef.t = Vec3d(refined);
    _yaw_offset = yaw;
    _last_rcv_ddt = ddt;
    std::memcpy(_rough_anchor, rough, sizeof rough);
    std::cout << "[GvioAligner]: Yaw offset from north = " << yaw * 180.0 / M_PI << " (deg)" << std::endl;
    std::cout << "[GvioAligner]: Refined anchor in ECEF = " << refined[0] << " " << refined[1] << " " << refined[2] << " (m)" << std::endl;
}

bool GvioAligner::coarseLocalization(double rough[7])                                // :199-233
{
    std::vector<const RawGnssEpoch*> all;
    for (const Item& it : _align_buffer) all.push_back(&it.meas);
    double xyzt[7];
    const bool ok = psrPos(all, xyzt);
    const double nrm = std::sqrt(xyzt[0] * xyzt[0] + xyzt[1] * xyzt[1] + xyzt[2] * xyzt[2]);
    if (!ok || !(nrm >= 1e-06)) {
        std::cout << "[GvioAligner]: Coarse anchor localization failed!" << std::endl;
        return false;
    }
    for (int i = 0; i < 4; ++i) if (std::fabs(xyzt[3 + i]) < 1.0) xyzt[3 + i] = 0.0;
    std::memcpy(rough, xyzt, sizeof xyzt);
    return true;
}

bool GvioAligner::yawAlignment(const double rough[3], double& yaw_offset, double& rcv_ddt)      // :235-312
{
    yaw_offset = 0.0; rcv_ddt = 0.0;
    double estYaw = 0.0, estRcvDdt = 0.0, delta_norm = 1.0;
    const Mat3d R = gnss::geo2rotation(gnss::ecef2geo(Vec3d(rough)));
    int iter = 0;
    std::vector<ingvio_gnss_epoch> eps(_align_buffer.size());
    std::vector<double> rec;
    while (iter <= _max_iter && delta_norm > _conv_epsilon) {
        Mat3d dotC3;                                                                  // d Rz / d yaw
        dotC3(0, 0) = -std::sin(estYaw); dotC3(0, 1) = -std::cos(estYaw); dotC3(1, 0) = std::cos(estYaw); dotC3(1, 1) = -std::sin(estYaw);
        const Mat3d Rw2enu = rotZ(estYaw);
        for (size_t q = 0; q < eps.size(); ++q) {
            fillEpoch(eps[q], _align_buffer[q].meas, nullptr);
            std::memcpy(eps[q].anchor_ecef, rough, 24);                              // dopp_res at the rough anchor for every epoch (:271-273)
            const Vec3d ve = R * (Rw2enu * _align_buffer[q].v);
            for (int c = 0; c < 3; ++c) eps[q].v_w[c] = ve[c];
            eps[q].fs = estRcvDdt;
        }
        if (!evalEpochs(eps, rec)) return false;
        std::vector<double> N(4, 0.0), g(2, 0.0);
        for (size_t q = 0; q < eps.size(); ++q) {
            const Vec3d dv = R * (dotC3 * _align_buffer[q].v);
            for (int i = 0; i < _align_buffer[q].meas.n_sat(); ++i) {
                const double* r = rec.data() + ((size_t)q * SMAX + i) * REC;
                // rows of satellites without a state stay zero in the reference (A row = [0 1], b = 0): they still enter A^T A
                const double a0 = r[9] != 0.0 ? -(r[2] * dv[0] + r[3] * dv[1] + r[4] * dv[2]) : 0.0, b = r[9] != 0.0 ? r[1] : 0.0;
                N[0] += a0 * a0; N[1] += a0; N[2] += a0; N[3] += 1.0; g[0] += a0 * b; g[1] += b;
            }
        }
        double d[2];
        if (!solveNormal(2, N, g, d)) return false;
        estYaw += d[0]; estRcvDdt += d[1];
        delta_norm = std::sqrt(d[0] * d[0] + d[1] * d[1]);
        ++iter;
    }
    if (iter > _max_iter) {
        std::cout << "[GvioAligner]: Yaw alignment reaches max iter, failed!" << std::endl;
        return false;
    }
    yaw_offset = estYaw;
    if (yaw_offset > M_PI) yaw_offset -= std::floor(estYaw / (2.0 * M_PI) + 0.5) * (2.0 * M_PI);
    else if (yaw_offset < -M_PI) yaw_offset -= std::ceil(estYaw / (2.0 * M_PI) - 0.5) * (2.0 * M_PI);
    rcv_ddt = estRcvDdt;
    return true;
}

bool GvioAligner::anchorRefinement(double yaw_offset, double, const double rough[7], double refined[7])      // :314-383
{
    std::memcpy(refined, rough, 56);
    const Mat3d Rw2enu = rotZ(yaw_offset);
    std::vector<std::vector<double>> spp(_align_buffer.size(), std::vector<double>(7, 0.0));
    for (size_t i = 0; i < _align_buffer.size(); ++i) {
        std::vector<const RawGnssEpoch*> one{ &_align_buffer[i].meas };
        const bool ok = psrPos(one, spp[i].data());
        const double nrm = std::sqrt(spp[i][0] * spp[i][0] + spp[i][1] * spp[i][1] + spp[i][2] * spp[i][2]);
        if (!ok || !(nrm >= 1e-03)) {
            std::cout << "[GvioAligner]: Anchor refinement failure due to unable to conduct SPP!" << std::endl;
            return false;
        }
    }
    int iter_refine = 0;
    while (iter_refine <= _max_iter) {
        const Mat3d Rw2ecef = gnss::geo2rotation(gnss::ecef2geo(Vec3d(refined))) * Rw2enu;
        Vec3d anchor;
        for (size_t i = 0; i < _align_buffer.size(); ++i) {
            const Vec3d q = Rw2ecef * _align_buffer[i].p;
            for (int c = 0; c < 3; ++c) anchor[c] += spp[i][c] - q[c];
        }
        double dxn = 0.0;
        for (int c = 0; c < 3; ++c) { anchor[c] /= (double)_align_buffer.size(); const double d = anchor[c] - refined[c]; dxn += d * d; refined[c] = anchor[c]; }
        if (std::sqrt(dxn) > _conv_epsilon) break;                                  // as written (:367-368): leaves on the first NON-converged step
        ++iter_refine;
    }
    if (iter_refine > _max_iter) {
        std::cout << "[GvioAligner]: Anchor refinement failure reaching max iter!" << std::endl;
        return false;
    }
    for (int k = 3; k < 7; ++k) refined[k] = spp.back()[k];
    return true;
}

}  // namespace ingvio
