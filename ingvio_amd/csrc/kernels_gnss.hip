// kernels_gnss.hip — SURVEY.md 8(f) row f-3: the gnss_comm front of GnssUpdate::updateTrackedSys as one batched kernel.
// One lane per (filter, satellite): satellite position / velocity / clock from the broadcast Kepler ephemeris at transmit
// time (gnss_comm/src/gnss_spp.cpp:50-98 sat_states, gnss_utility.cpp:390-640 Kepler / eph2svdt / eph2pos / eph2vel),
// azimuth / elevation (:347-388, :733-772), Saastamoinen + Niell troposphere (:774-863), Klobuchar ionosphere (:865-899),
// pseudo-range and Doppler residuals (gnss_spp.cpp:100-146, :256-282) — then the candidate rows of the update
// (ingvio_estimator/src/GnssUpdate.cpp:148-272) written straight into the staged-row buffers ingvio_gnss_run gates and
// applies: a GNSS epoch goes from raw observations to the posterior without leaving the device.
// Scalar geodesy with transcendental functions: bound by the latency of a few hundred dependent FP64 operations per lane, the
// point is that it runs for every filter of the batch at once and needs no host assembly.  GLONASS satellites (round 3) carry a
// PZ-90 state vector instead of Kepler elements: geph2svdt / deq / glo_orbit / geph2pos / geph2vel (gnss_utility.cpp:642-731),
// classical Runge-Kutta in 60 s steps from toe to the transmit time, on the satellite's lane.  gfx950 only.
#include "launch_gnss.h"

#include <math.h>

namespace {

constexpr double kC = 2.99792458e8, kMuGps = 3.9860050000e14, kMu = 3.9860044180e14;
constexpr double kOmgGps = 7.2921151467e-5, kOmgBds = 7.2921150000e-5, kWeek = 604800.0;
constexpr double kSinN5 = -0.0871557427476582, kCosN5 = 0.9961946980917456, kD2R = M_PI / 180.0;

__device__ __forceinline__ double wrap_week(double t) { return t > kWeek / 2 ? t - kWeek : (t < -kWeek / 2 ? t + kWeek : t); }

__device__ double kepler(double mk, double es)           // gnss_utility.cpp:390-405
{
    double e = mk, ek = 1e6;
    for (int it = 0; it < 30 && fabs(e - ek) > 1e-14; ++it) {
        ek = e;
        e -= (e - es * sin(e) - mk) / (1.0 - es * cos(e));
    }
    return ek;
}

__device__ double eph2svdt(double t, const double* __restrict__ ep)      // :437-446
{
    double dt = wrap_week(t - ep[GE_TOC]);
    for (int i = 0; i < 2; ++i) dt -= ep[GE_AF0] + ep[GE_AF1] * dt + ep[GE_AF2] * dt * dt;
    return ep[GE_AF0] + ep[GE_AF1] * dt + ep[GE_AF2] * dt * dt;
}

// eph2pos (:448-531) + eph2vel (:533-640)
__device__ void eph2posvel(double t, const double* __restrict__ ep, double pos[3], double vel[3], double& svdt, double& svddt)
{
    const int sys = (int)ep[GE_SYS], prn = (int)ep[GE_PRN];
    const double mu = sys == 0 ? kMuGps : kMu, om = sys == 3 ? kOmgBds : kOmgGps;
    const double A = ep[GE_A], e = ep[GE_E];
    const double tk = wrap_week(t - ep[GE_TOE]);
    const double n = sqrt(mu / (A * A * A)) + ep[GE_DELTA_N];
    const double Ek = kepler(ep[GE_M0] + n * tk, e);
    const double sE = sin(Ek), cE = cos(Ek);
    const double Ed = n / (1 - e * cE), q = sqrt(1 - e * e);
    const double vd = q * Ed / (1 - e * cE);
    const double phi = atan2(q * sE, cE - e) + ep[GE_OMG];
    const double c2 = cos(2 * phi), s2 = sin(2 * phi);
    const double uk = phi + ep[GE_CUS] * s2 + ep[GE_CUC] * c2;
    const double rk = A * (1 - e * cE) + ep[GE_CRS] * s2 + ep[GE_CRC] * c2;
    const double ik = ep[GE_I0] + ep[GE_I_DOT] * tk + ep[GE_CIS] * s2 + ep[GE_CIC] * c2;
    const double ud = vd + 2 * vd * (ep[GE_CUS] * c2 - ep[GE_CUC] * s2);
    const double rd = A * e * Ed * sE + 2 * vd * (ep[GE_CRS] * c2 - ep[GE_CRC] * s2);
    const double idot = ep[GE_I_DOT] + 2 * vd * (ep[GE_CIS] * c2 - ep[GE_CIC] * s2);
    const double si = sin(ik), ci = cos(ik), su = sin(uk), cu = cos(uk);
    const double xk = rk * cu, yk = rk * su, xd = rd * cu - rk * ud * su, yd = rd * su + rk * ud * cu;
    if (sys == 3 && prn <= 5) {                                          // BeiDou GEO: inertial frame, then two rotations
        const double Ok = ep[GE_OMG0] + ep[GE_OMG_DOT] * tk - om * ep[GE_TOE_SYS], sO = sin(Ok), cO = cos(Ok), Od = ep[GE_OMG_DOT];
        const double t1 = xd - yk * Od * ci, t2 = xk * Od + yd * ci - yk * idot * si;
        const double xg = xk * cO - yk * ci * sO, yg = xk * sO + yk * ci * cO, zg = yk * si;
        const double xgd = t1 * cO - t2 * sO, ygd = t1 * sO + t2 * cO, zgd = yd * si + yd * idot * ci;      // z: as written (:617)
        const double so = sin(om * tk), co = cos(om * tk), sod = om * co, cod = -om * so;
        pos[0] = xg * co + yg * so * kCosN5 + zg * so * kSinN5;
        pos[1] = -xg * so + yg * co * kCosN5 + zg * co * kSinN5;
        pos[2] = -yg * kSinN5 + zg * kCosN5;
        vel[0] = xgd * co + xg * cod + ygd * so * kCosN5 + yg * sod * kCosN5 + zgd * so * kSinN5 + zg * sod * kSinN5;
        vel[1] = -xgd * so - xg * sod + ygd * co * kCosN5 + yg * cod * kCosN5 + zgd * co * kSinN5 + zg * cod * kSinN5;
        vel[2] = -ygd * kSinN5 + zgd * kCosN5;
    } else {
        const double Ok = ep[GE_OMG0] + (ep[GE_OMG_DOT] - om) * tk - om * ep[GE_TOE_SYS], sO = sin(Ok), cO = cos(Ok), Od = ep[GE_OMG_DOT] - om;
        const double t1 = xd - yk * Od * ci, t2 = xk * Od + yd * ci - yk * idot * si;
        pos[0] = xk * cO - yk * ci * sO; pos[1] = xk * sO + yk * ci * cO; pos[2] = yk * si;
        vel[0] = t1 * cO - t2 * sO; vel[1] = t1 * sO + t2 * cO; vel[2] = yd * si + yd * idot * ci;           // z: as written (:632)
    }
    const double dt = wrap_week(t - ep[GE_TOC]);
    svdt = ep[GE_AF0] + ep[GE_AF1] * dt + ep[GE_AF2] * dt * dt - 2.0 * sqrt(mu * A) * e * sE / kC / kC;
    svddt = ep[GE_AF1] + 2.0 * ep[GE_AF2] * dt - 2.0 * sqrt(mu * A) * e * cE * Ed / kC / kC;
}

// ---- GLONASS (record layout: launch_gnss.h GE_GLO_*) ---------------------------------------------------------------------------
constexpr double kOmgGlo = 7.2921150000e-5, kReGlo = 6378136.0, kJ2Glo = 1.0826257E-3, kTstep = 60.0;

__device__ double geph2svdt(double t, const double* __restrict__ ep)      // :679-690
{
    double dt = wrap_week(t - ep[GE_TOE]);
    for (int i = 0; i < 2; ++i) dt -= -ep[GE_GLO_TAUN] + ep[GE_GLO_GAMMA] * dt;
    return -ep[GE_GLO_TAUN] + ep[GE_GLO_GAMMA] * dt;
}

__device__ __forceinline__ void glo_deq(const double x[6], const double acc[3], double xd[6])      // :642-660
{
    const double r2 = x[0] * x[0] + x[1] * x[1] + x[2] * x[2], r3 = r2 * sqrt(r2), omg2 = kOmgGlo * kOmgGlo;
    if (r2 <= 0.0) { for (int i = 0; i < 6; ++i) xd[i] = 0.0; return; }
    const double a = 1.5 * kJ2Glo * kMu * kReGlo * kReGlo / r2 / r3;       // 3/2 J2 mu Ae^2 / r^5
    const double b = 5.0 * x[2] * x[2] / r2;
    const double c = -kMu / r3 - a * (1.0 - b);
    xd[0] = x[3]; xd[1] = x[4]; xd[2] = x[5];
    xd[3] = (c + omg2) * x[0] + 2.0 * kOmgGlo * x[4] + acc[0];
    xd[4] = (c + omg2) * x[1] - 2.0 * kOmgGlo * x[3] + acc[1];
    xd[5] = (c - 2.0 * a) * x[2] + acc[2];
}

// geph2pos (:692-708) + geph2vel (:710-726): one integration serves both
__device__ void geph2posvel(double t, const double* __restrict__ ep, double pos[3], double vel[3], double& svdt, double& svddt)
{
    double x[6], acc[3];
    for (int i = 0; i < 3; ++i) { x[i] = ep[GE_GLO_POS + i]; x[3 + i] = ep[GE_GLO_VEL + i]; acc[i] = ep[GE_GLO_ACC + i]; }
    double dt = wrap_week(t - ep[GE_TOE]);
    svdt = -ep[GE_GLO_TAUN] + ep[GE_GLO_GAMMA] * dt;
    svddt = ep[GE_GLO_GAMMA];
    for (double tt = dt < 0.0 ? -kTstep : kTstep; fabs(dt) > 1e-9; dt -= tt) {
        if (fabs(dt) < kTstep) tt = dt;
        double k1[6], k2[6], k3[6], k4[6], w[6];                          // glo_orbit :662-677
        glo_deq(x, acc, k1);
        for (int i = 0; i < 6; ++i) w[i] = x[i] + 0.5 * k1[i] * tt;
        glo_deq(w, acc, k2);
        for (int i = 0; i < 6; ++i) w[i] = x[i] + 0.5 * k2[i] * tt;
        glo_deq(w, acc, k3);
        for (int i = 0; i < 6; ++i) w[i] = x[i] + k3[i] * tt;
        glo_deq(w, acc, k4);
        for (int i = 0; i < 6; ++i) x[i] += (k1[i] + 2.0 * k2[i] + 2.0 * k3[i] + k4[i]) * tt / 6.0;
    }
    for (int i = 0; i < 3; ++i) { pos[i] = x[i]; vel[i] = x[3 + i]; }
}

__device__ void ecef2geo(const double x[3], double lla[3])               // :347-388
{
    lla[0] = lla[1] = lla[2] = 0.0;
    if (x[0] == 0 && x[1] == 0) return;
    const double e2 = 6.69437999014e-3, a = 6378137.0, a2 = a * a, b2 = a2 * (1 - e2), b = sqrt(b2), ep2 = (a2 - b2) / b2;
    const double p = sqrt(x[0] * x[0] + x[1] * x[1]);
    double s1 = x[2] * a, s2 = p * b, h = sqrt(s1 * s1 + s2 * s2);
    const double st = s1 / h, ct = s2 / h;
    s1 = x[2] + ep2 * b * st * st * st;
    s2 = p - a * e2 * ct * ct * ct;
    h = sqrt(s1 * s1 + s2 * s2);
    const double sl = s1 / h, cl = s2 / h;
    const double N = a2 / sqrt(a2 * cl * cl + b2 * sl * sl);
    lla[0] = atan(s1 / s2) / kD2R; lla[1] = atan2(x[1], x[0]) / kD2R; lla[2] = p / cl - N;
}

__device__ __forceinline__ double interpc(const double* coef, double lat)      // :774-779
{
    const int i = (int)(lat / 15.0);
    if (i < 1) return coef[0];
    if (i > 4) return coef[4];
    return coef[i - 1] * (1.0 - lat / 15.0 + i) + coef[i] * (lat / 15.0 - i);
}
__device__ __forceinline__ double mapf(double el, double a, double b, double c)      // :782-786
{
    const double s = sin(el);
    return (1.0 + a / (1.0 + b / (1.0 + c))) / (s + (a / (s + b / (s + c))));
}

__constant__ double kNmf[9][5] = {
    { 1.2769934E-3, 1.2683230E-3, 1.2465397E-3, 1.2196049E-3, 1.2045996E-3 }, { 2.9153695E-3, 2.9152299E-3, 2.9288445E-3, 2.9022565E-3, 2.9024912E-3 },
    { 62.610505E-3, 62.837393E-3, 63.721774E-3, 63.824265E-3, 64.258455E-3 }, { 0.0, 1.2709626E-5, 2.6523662E-5, 3.4000452E-5, 4.1202191E-5 },
    { 0.0, 2.1414979E-5, 3.0160779E-5, 7.2562722E-5, 11.723375E-5 }, { 0.0, 9.0128400E-5, 4.3497037E-5, 84.795348E-5, 170.37206E-5 },
    { 5.8021897E-4, 5.6794847E-4, 5.8118019E-4, 5.9727542E-4, 6.1641693E-4 }, { 1.4275268E-3, 1.5138625E-3, 1.4572752E-3, 1.5007428E-3, 1.7599082E-3 },
    { 4.3472961E-2, 4.6729510E-2, 4.3908931E-2, 4.4626982E-2, 5.4736038E-2 } };

__device__ double trop_delay(double doy, const double lla[3], double el)      // calculate_trop_delay :841-863 + nmf :797-839
{
    if (lla[2] < -100.0 || 1E4 < lla[2] || el <= 0) return 0.0;
    const double hgt = lla[2] < 0.0 ? 0.0 : lla[2];
    const double pres = 1013.25 * pow(1.0 - 2.2557E-5 * hgt, 5.2568);
    const double temp = 15.0 - 6.5E-3 * hgt + 273.16;
    const double e = 6.108 * 0.7 * exp((17.15 * temp - 4684.0) / (temp - 38.45));
    const double zhd = 0.0022768 * pres / (1.0 - 0.00266 * cos(2.0 * lla[0] * kD2R) - 0.00028 * hgt / 1E3);
    const double zwd = 0.002277 * (1255.0 / temp + 0.05) * e;
    double lat = lla[0];
    const double y = (doy - 28.0) / 365.25 + (lat < 0.0 ? 0.5 : 0.0), cosy = cos(2.0 * M_PI * y);
    lat = fabs(lat);
    double ah[3], aw[3];
    for (int i = 0; i < 3; ++i) { ah[i] = interpc(kNmf[i], lat) - interpc(kNmf[i + 3], lat) * cosy; aw[i] = interpc(kNmf[i + 6], lat); }
    const double dm = (1.0 / sin(el) - mapf(el, 2.53E-5, 5.49E-3, 1.14E-3)) * lla[2] / 1E3;
    return (mapf(el, ah[0], ah[1], ah[2]) + dm) * zhd + mapf(el, aw[0], aw[1], aw[2]) * zwd;
}

__device__ double iono_delay(double tow, const double* __restrict__ ion, const double lla[3], double az, double el)      // :865-899
{
    if (lla[2] < -1E3 || el <= 0) return 0.0;
    const double psi = 0.0137 / (el / M_PI + 0.11) - 0.022;
    double phi = lla[0] / 180.0 + psi * cos(az);
    phi = phi > 0.416 ? 0.416 : (phi < -0.416 ? -0.416 : phi);
    const double lam = lla[1] / 180.0 + psi * sin(az) / cos(phi * M_PI);
    phi += 0.064 * cos((lam - 1.617) * M_PI);
    double tt = 43200.0 * lam + tow;
    tt -= floor(tt / 86400.0) * 86400.0;
    const double f = 1.0 + 16.0 * pow(0.53 - el / M_PI, 3.0);
    double amp = ion[0] + phi * (ion[1] + phi * (ion[2] + phi * ion[3]));
    double per = ion[4] + phi * (ion[5] + phi * (ion[6] + phi * ion[7]));
    amp = amp < 0.0 ? 0.0 : amp;
    per = per < 72000.0 ? 72000.0 : per;
    const double x = 2.0 * M_PI * (tt - 50400.0) / per;
    return kC * f * (fabs(x) < 1.57 ? 5E-9 + amp * (1.0 + x * x * (-0.5 + x * x / 24.0)) : 5E-9);
}

// grid = nb, 64 threads: lane = satellite.  Writes the per-satellite results (front [B][64][GF_N]) and the candidate rows.
__global__ __launch_bounds__(64) void k_gnss_front(GnssFrontLaunch L)
{
    const int bl = blockIdx.x, i = threadIdx.x;
    const double* rc = L.rcv + (size_t)bl * GR_N;
    const int ns = (int)rc[GR_NSAT];
    __shared__ int sCbCol[4];
    __shared__ int sIdx[4];
    double* H = L.H ? L.H + (size_t)bl * L.hstride : nullptr;
    if (H) { for (int e = i; e < L.mld * GNSS_FRONT_NCW; e += 64) H[e] = 0.0; }
    // receiver in ECEF: rcv = R_enu2ecef Rz(yaw) p_w + anchor (GnssUpdate.cpp:102), same rotation for the velocity (:108)
    const double cy = cos(rc[GR_YAW]), sy = sin(rc[GR_YAW]);
    double Rw[9];                                                     // R_w2ecef = R_enu2ecef Rz(yaw), row-major (:141)
    for (int r = 0; r < 3; ++r) {
        Rw[3 * r] = rc[GR_RENU + 3 * r] * cy + rc[GR_RENU + 3 * r + 1] * sy;
        Rw[3 * r + 1] = -rc[GR_RENU + 3 * r] * sy + rc[GR_RENU + 3 * r + 1] * cy;
        Rw[3 * r + 2] = rc[GR_RENU + 3 * r + 2];
    }
    double xyz[3], vel[3];
    for (int r = 0; r < 3; ++r) {
        xyz[r] = Rw[3 * r] * rc[GR_PW] + Rw[3 * r + 1] * rc[GR_PW + 1] + Rw[3 * r + 2] * rc[GR_PW + 2] + rc[GR_ANCHOR + r];
        vel[r] = Rw[3 * r] * rc[GR_VW] + Rw[3 * r + 1] * rc[GR_VW + 1] + Rw[3 * r + 2] * rc[GR_VW + 2];
    }
    double lla[3];
    ecef2geo(xyz, lla);
    const double rn = sqrt(xyz[0] * xyz[0] + xyz[1] * xyz[1] + xyz[2] * xyz[2]);
    bool usable = false;
    int sys = -1;
    double res_pos = 0, res_vel = 0, u[3] = { 0, 0, 0 }, az = 0, el = M_PI / 2, ion_d = 0, tro_d = 0, npsr = 0, ndop = 0;
    double st[10] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };                  // SatState (gnss_constant.hpp:506-516): pos, vel, dt, ddt, tgd, ttx
    if (i < ns) {
        const double* ep = L.eph + ((size_t)bl * L.smax + i) * GE_N;
        const double* ob = L.obs + ((size_t)bl * L.smax + i) * GO_N;
        sys = (int)ep[GE_SYS];
        if (sys >= 0 && sys <= 3 && ob[GO_FREQ] >= 0) {
            // sat_states (gnss_spp.cpp:50-98)
            double ttx = ob[GO_TOW] - ob[GO_PSR] / kC;
            double sp[3], sv[3], dts, ddts, tgd;
            if (sys == 1) {                                           // GLONASS (:72-79): SatState::tgd keeps its default 0
                ttx -= geph2svdt(ttx, ep);
                geph2posvel(ttx, ep, sp, sv, dts, ddts);
                tgd = 0.0;
            } else {
                ttx -= eph2svdt(ttx, ep);
                eph2posvel(ttx, ep, sp, sv, dts, ddts);
                tgd = ep[GE_TGD];
            }
            for (int c = 0; c < 3; ++c) { st[c] = sp[c]; st[3 + c] = sv[c]; }
            st[6] = dts; st[7] = ddts; st[8] = tgd; st[9] = ttx;
            const double d[3] = { sp[0] - xyz[0], sp[1] - xyz[1], sp[2] - xyz[2] };
            const double range = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            for (int c = 0; c < 3; ++c) u[c] = d[c] / range;
            if (rn > 0) {                                             // sat_azel (gnss_utility.cpp:762-772), delays (gnss_spp.cpp:121-128)
                const double lat = lla[0] * kD2R, lon = lla[1] * kD2R, sl = sin(lat), cl = cos(lat), so = sin(lon), co = cos(lon);
                const double e_ = -so * u[0] + co * u[1], n_ = -sl * co * u[0] - sl * so * u[1] + cl * u[2], up = cl * co * u[0] + cl * so * u[1] + sl * u[2];
                az = sqrt(u[0] * u[0] + u[1] * u[1]) < 1e-12 ? 0.0 : atan2(e_, n_);
                if (az < 0) az += 2 * M_PI;
                el = asin(up);
                tro_d = trop_delay(rc[GR_DOY], lla, el);
                ion_d = rc[GR_HAVE_ION] != 0.0 ? iono_delay(ttx, rc + GR_ION, lla, az, el) : 0.0;
            }
            const double sag = kOmgGps * (sp[0] * xyz[1] - sp[1] * xyz[0]) / kC;
            res_pos = range + sag + rc[GR_CB + sys] - dts * kC + tro_d + ion_d + tgd * kC - ob[GO_PSR];      // :132-139
            const double sagd = kOmgGps / kC * (sv[0] * xyz[1] + sp[0] * vel[1] - sv[1] * xyz[0] - sp[1] * vel[0]);
            res_vel = (sv[0] - vel[0]) * u[0] + (sv[1] - vel[1]) * u[1] + (sv[2] - vel[2]) * u[2] + rc[GR_FS] + sagd - ddts * kC
                      + ob[GO_DOPP] * (kC / ob[GO_FREQ]);                                                                  // :267-277
            double se = sin(el);
            if (fabs(se) < 1e-6) se = 1e-6;
            npsr = rc[GR_PSR_AMP] * sqrt(ep[GE_URA] * ob[GO_PSR_STD] / (se * se));                                        // GnssUpdate.cpp:180-187
            ndop = rc[GR_DOPP_AMP] * sqrt(ep[GE_URA] * (ob[GO_DOPP_STD] * kC / ob[GO_FREQ]) / (se * se));                  // :249-256
            usable = true;
        }
    }
    double* fr = L.front + ((size_t)bl * 64 + i) * GF_N;
    fr[0] = res_pos; fr[1] = res_vel; fr[2] = u[0]; fr[3] = u[1]; fr[4] = u[2]; fr[5] = az; fr[6] = el; fr[7] = ion_d; fr[8] = tro_d;
    fr[9] = usable ? 1.0 : 0.0;
    for (int c = 0; c < 10; ++c) fr[10 + c] = st[c];
    if (!H) return;                                                   // satellite evaluation only (ingvio_gnss_sat_eval)
    // ---- candidate rows of updateTrackedSys (GnssUpdate.cpp:148-272): only constellations whose clock is in the state ----
    if (i < 4) sIdx[i] = (int)rc[GR_IDX_CB + i];
    __syncthreads();
    const bool in_state = usable && sIdx[(sys < 0 || sys > 3) ? 0 : sys] >= 0;
    const unsigned long long mk = __ballot(in_state);
    const int nrow = __popcll(mk), rank = __popcll(mk & ((1ULL << i) - 1ULL));
    if (i == 0) {
        int col = 10;
        int* cm = L.colmap + (size_t)bl * GNSS_FRONT_NCW;
        const int i0 = (int)rc[GR_IDX_SE23];
        for (int c = 0; c < 9; ++c) cm[c] = i0 + c;                   // var_order: SE23, YOF, clock biases in order of first appearance, FS
        cm[9] = (int)rc[GR_IDX_YOF];
        for (int s = 0; s < 4; ++s) sCbCol[s] = -1;
        for (int j = 0; j < ns; ++j) {
            if (!((mk >> j) & 1ULL)) continue;
            const int s = (int)L.eph[((size_t)bl * L.smax + j) * GE_N + GE_SYS];
            if (sCbCol[s] < 0) { sCbCol[s] = col; cm[col] = sIdx[s]; ++col; }
        }
        cm[col] = (int)rc[GR_IDX_FS];
        L.nc[bl] = nrow ? col + 1 : 0;
        L.m[bl] = 2 * nrow;
    }
    __syncthreads();
    if (in_state) {
        int fs_col = 10;
        for (int s = 0; s < 4; ++s) if (sCbCol[s] >= 0) ++fs_col;
        const double p[3] = { rc[GR_PW], rc[GR_PW + 1], rc[GR_PW + 2] }, v[3] = { rc[GR_VW], rc[GR_VW + 1], rc[GR_VW + 2] };
        double uR[3];                                                  // u^T R_w2ecef
        for (int c = 0; c < 3; ++c) uR[c] = u[0] * Rw[c] + u[1] * Rw[3 + c] + u[2] * Rw[6 + c];
        // u^T R [x]_x = (uR x x)^T:  row vector w^T [x]_x = (x cross ... ) -> (w^T [x]x)_c = sum_r w_r skew(x)[r][c]
        const double hp[3] = { uR[1] * p[2] - uR[2] * p[1], uR[2] * p[0] - uR[0] * p[2], uR[0] * p[1] - uR[1] * p[0] };      // (:161) = -(uR x p)... see note
        const double hv[3] = { uR[1] * v[2] - uR[2] * v[1], uR[2] * v[0] - uR[0] * v[2], uR[0] * v[1] - uR[1] * v[0] };
        const int r0 = rank, r1 = nrow + rank, mld = L.mld;
        // w^T skew(x): skew(x) = [[0,-x2,x1],[x2,0,-x0],[-x1,x0,0]]  ->  (w1 x2 - w2 x1, w2 x0 - w0 x2, w0 x1 - w1 x0)
        for (int c = 0; c < 3; ++c) {
            H[r0 + (size_t)c * mld] = hp[c];
            H[r0 + (size_t)(3 + c) * mld] = -uR[c];
            H[r1 + (size_t)c * mld] = hv[c];
            H[r1 + (size_t)(6 + c) * mld] = -uR[c];
        }
        H[r0 + (size_t)sCbCol[sys] * mld] = 1.0;
        H[r1 + (size_t)fs_col * mld] = 1.0;
        double* res = L.res + (size_t)bl * mld; double* nz = L.noise + (size_t)bl * mld;
        res[r0] = -res_pos; nz[r0] = npsr * npsr;                       // :170, :195
        res[r1] = -res_vel; nz[r1] = ndop * ndop;                       // :246, :264
    }
}

}  // namespace

void launch_gnss_front(const GnssFrontLaunch& L, int nb, hipStream_t st)
{
    hipLaunchKernelGGL(k_gnss_front, dim3(nb), dim3(64), 0, st, L);
}
