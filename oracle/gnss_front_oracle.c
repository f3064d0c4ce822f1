/* gnss_front_oracle.c — CPU restatement of the gnss_comm slice in front of GnssUpdate::updateTrackedSys:
 * satellite position / velocity / clock from a broadcast (Kepler) ephemeris at signal transmit time, Saastamoinen + Niell
 * troposphere, Klobuchar ionosphere, pseudo-range and Doppler residuals with line of sight and elevation.
 * TEST INFRASTRUCTURE ONLY (see ingvio_oracle.h): linked into liboracle.so, never into the product.
 * gnss_comm is vendored in the reference (gnss_comm/src/gnss_spp.cpp, gnss_utility.cpp) but needs Eigen + glog: not buildable
 * here, and the reference holds NO test for it — parity unpinned by reference tests; pinned by an independent numpy
 * transcription (oracle/gen_gnss_golden.py) and by physical identities (tests/test_gnss_front.py).
 * Times are seconds of the GPS week (the reference's gtime_t differences are the same numbers, up to the wrap at the week
 * boundary).  GLONASS (round 3): geph2svdt / deq / glo_orbit / geph2pos / geph2vel, gnss_utility.cpp:642-731 - the broadcast PZ-90
 * state vector integrated by classical Runge-Kutta in 60 s steps with the J2 + frame-rotation force model; record layout below. */
#define _USE_MATH_DEFINES
#define _GNU_SOURCE
#include <math.h>
#include <string.h>
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#include "ingvio_oracle.h"

#define LIGHT_SPEED 2.99792458e8
#define MU_GPS 3.9860050000e14
#define MU_GAL 3.9860044180e14
#define OMG_GPS 7.2921151467e-5
#define OMG_BDS 7.2921150000e-5
#define WEEK_SECONDS 604800.0
#define SIN_N5 (-0.0871557427476582)
#define COS_N5 0.9961946980917456
#define D2R (M_PI / 180.0)

/* GLONASS ephemeris record (sys == 1), same ORC_EPH_N doubles: sys, prn, toe, -, -, pos[3], vel[3], acc[3], tau_n, gamma; ura */
#define GLO_POS 5
#define GLO_VEL 8
#define GLO_ACC 11
#define GLO_TAUN 14
#define GLO_GAMMA 15
#define OMG_GLO 7.2921150000e-5
#define RE_GLO 6378136.0
#define J2_GLO 1.0826257E-3
#define TSTEP 60.0

static double wrap_week(double t)                        /* gnss_utility.cpp:451-456 */
{
    if (t > WEEK_SECONDS / 2) t -= WEEK_SECONDS;
    else if (t < -WEEK_SECONDS / 2) t += WEEK_SECONDS;
    return t;
}

static double kepler(double mk, double es)               /* :390-405 */
{
    double e = mk, ek = 1e6;
    int it = 0;
    while (it < 30 && fabs(e - ek) > 1e-14) {
        ek = e;
        e -= (e - es * sin(e) - mk) / (1.0 - es * cos(e));
        ++it;
    }
    return ek;
}

static double eph2svdt(double t, const double* eph)      /* :437-446 */
{
    double dt = wrap_week(t - eph[ORC_EPH_TOC]);
    for (int i = 0; i < 2; ++i) dt -= eph[ORC_EPH_AF0] + eph[ORC_EPH_AF1] * dt + eph[ORC_EPH_AF2] * dt * dt;
    return eph[ORC_EPH_AF0] + eph[ORC_EPH_AF1] * dt + eph[ORC_EPH_AF2] * dt * dt;
}

/* eph2pos (:448-531) and eph2vel (:533-640) share everything up to the orbital-plane quantities */
static void eph2posvel(double t, const double* eph, double pos[3], double vel[3], double* svdt, double* svddt)
{
    const int sys = (int)eph[ORC_EPH_SYS], prn = (int)eph[ORC_EPH_PRN];
    const double mu = sys == 0 ? MU_GPS : MU_GAL, earth_omg = sys == 3 ? OMG_BDS : OMG_GPS;
    const double A = eph[ORC_EPH_A], e = eph[ORC_EPH_E];
    const double tk = wrap_week(t - eph[ORC_EPH_TOE]);
    const double n = sqrt(mu / pow(A, 3)) + eph[ORC_EPH_DELTA_N];
    const double Mk = eph[ORC_EPH_M0] + n * tk;
    const double Ek = kepler(Mk, e);
    const double sin_Ek = sin(Ek), cos_Ek = cos(Ek);
    const double Ek_dot = n / (1 - e * cos_Ek);
    const double vk_dot = sqrt(1 - e * e) * Ek_dot / (1 - e * cos_Ek);
    const double vk = atan2(sqrt(1 - e * e) * sin_Ek, cos_Ek - e);
    const double phi = vk + eph[ORC_EPH_OMG];
    const double c2 = cos(2 * phi), s2 = sin(2 * phi);
    const double duk = eph[ORC_EPH_CUS] * s2 + eph[ORC_EPH_CUC] * c2;
    const double drk = eph[ORC_EPH_CRS] * s2 + eph[ORC_EPH_CRC] * c2;
    const double dik = eph[ORC_EPH_CIS] * s2 + eph[ORC_EPH_CIC] * c2;
    const double uk = phi + duk, rk = A * (1 - e * cos_Ek) + drk, ik = eph[ORC_EPH_I0] + eph[ORC_EPH_I_DOT] * tk + dik;
    const double sin_ik = sin(ik), cos_ik = cos(ik), sin_uk = sin(uk), cos_uk = cos(uk);
    const double xk = rk * cos_uk, yk = rk * sin_uk;
    const double duk_dot = 2 * vk_dot * (eph[ORC_EPH_CUS] * c2 - eph[ORC_EPH_CUC] * s2);
    const double drk_dot = 2 * vk_dot * (eph[ORC_EPH_CRS] * c2 - eph[ORC_EPH_CRC] * s2);
    const double dik_dot = 2 * vk_dot * (eph[ORC_EPH_CIS] * c2 - eph[ORC_EPH_CIC] * s2);
    const double uk_dot = vk_dot + duk_dot, rk_dot = A * e * Ek_dot * sin_Ek + drk_dot, ik_dot = eph[ORC_EPH_I_DOT] + dik_dot;
    const double xk_dot = rk_dot * cos_uk - rk * uk_dot * sin_uk, yk_dot = rk_dot * sin_uk + rk * uk_dot * cos_uk;
    const double toe_tow = eph[ORC_EPH_TOE_SYS];
    if (sys == 3 && prn <= 5) {                                            /* BDS GEO */
        const double OMG_k = eph[ORC_EPH_OMG0] + eph[ORC_EPH_OMG_DOT] * tk - earth_omg * toe_tow;
        const double so = sin(OMG_k), co = cos(OMG_k), OMGk_dot = eph[ORC_EPH_OMG_DOT];
        const double term1 = xk_dot - yk * OMGk_dot * cos_ik, term2 = xk * OMGk_dot + yk_dot * cos_ik - yk * ik_dot * sin_ik;
        const double xg = xk * co - yk * cos_ik * so, yg = xk * so + yk * cos_ik * co, zg = yk * sin_ik;
        const double xg_dot = term1 * co - term2 * so, yg_dot = term1 * so + term2 * co;
        const double zg_dot = yk_dot * sin_ik + yk_dot * ik_dot * cos_ik;              /* as written (:617) */
        const double sin_o = sin(earth_omg * tk), cos_o = cos(earth_omg * tk);
        const double sin_o_dot = earth_omg * cos_o, cos_o_dot = -earth_omg * sin_o;
        pos[0] = xg * cos_o + yg * sin_o * COS_N5 + zg * sin_o * SIN_N5;
        pos[1] = -xg * sin_o + yg * cos_o * COS_N5 + zg * cos_o * SIN_N5;
        pos[2] = -yg * SIN_N5 + zg * COS_N5;
        vel[0] = xg_dot * cos_o + xg * cos_o_dot + yg_dot * sin_o * COS_N5 + yg * sin_o_dot * COS_N5 + zg_dot * sin_o * SIN_N5 + zg * sin_o_dot * SIN_N5;
        vel[1] = -xg_dot * sin_o - xg * sin_o_dot + yg_dot * cos_o * COS_N5 + yg * cos_o_dot * COS_N5 + zg_dot * cos_o * SIN_N5 + zg * cos_o_dot * SIN_N5;
        vel[2] = -yg_dot * SIN_N5 + zg_dot * COS_N5;
    } else {
        const double OMG_k = eph[ORC_EPH_OMG0] + (eph[ORC_EPH_OMG_DOT] - earth_omg) * tk - earth_omg * toe_tow;
        const double so = sin(OMG_k), co = cos(OMG_k), OMGk_dot = eph[ORC_EPH_OMG_DOT] - earth_omg;
        const double term1 = xk_dot - yk * OMGk_dot * cos_ik, term2 = xk * OMGk_dot + yk_dot * cos_ik - yk * ik_dot * sin_ik;
        pos[0] = xk * co - yk * cos_ik * so;
        pos[1] = xk * so + yk * cos_ik * co;
        pos[2] = yk * sin_ik;
        vel[0] = term1 * co - term2 * so;
        vel[1] = term1 * so + term2 * co;
        vel[2] = yk_dot * sin_ik + yk_dot * ik_dot * cos_ik;                             /* as written (:632): y'_k, not y_k */
    }
    const double dt = wrap_week(t - eph[ORC_EPH_TOC]);
    *svdt = eph[ORC_EPH_AF0] + eph[ORC_EPH_AF1] * dt + eph[ORC_EPH_AF2] * dt * dt - 2.0 * sqrt(mu * A) * e * sin_Ek / LIGHT_SPEED / LIGHT_SPEED;
    *svddt = eph[ORC_EPH_AF1] + 2.0 * eph[ORC_EPH_AF2] * dt - 2.0 * sqrt(mu * A) * e * cos_Ek * Ek_dot / LIGHT_SPEED / LIGHT_SPEED;
}

static double geph2svdt(double t, const double* eph)     /* :679-690 */
{
    double dt = wrap_week(t - eph[ORC_EPH_TOE]);
    for (int i = 0; i < 2; ++i) dt -= -eph[GLO_TAUN] + eph[GLO_GAMMA] * dt;
    return -eph[GLO_TAUN] + eph[GLO_GAMMA] * dt;
}

static void glo_deq(const double x[6], const double acc[3], double xdot[6])      /* :642-660 */
{
    const double r2 = x[0] * x[0] + x[1] * x[1] + x[2] * x[2], r3 = r2 * sqrt(r2), omg2 = OMG_GLO * OMG_GLO;
    if (r2 <= 0.0) { for (int i = 0; i < 6; ++i) xdot[i] = 0.0; return; }
    const double a = 1.5 * J2_GLO * MU_GAL * RE_GLO * RE_GLO / r2 / r3;           /* 3/2 J2 mu Ae^2 / r^5 (MU: GAL, BDS, GLO share it) */
    const double b = 5.0 * x[2] * x[2] / r2;
    const double c = -MU_GAL / r3 - a * (1.0 - b);
    xdot[0] = x[3]; xdot[1] = x[4]; xdot[2] = x[5];
    xdot[3] = (c + omg2) * x[0] + 2.0 * OMG_GLO * x[4] + acc[0];
    xdot[4] = (c + omg2) * x[1] - 2.0 * OMG_GLO * x[3] + acc[1];
    xdot[5] = (c - 2.0 * a) * x[2] + acc[2];
}

static void glo_orbit(double dt, double x[6], const double acc[3])                /* :662-677 */
{
    double k1[6], k2[6], k3[6], k4[6], w[6];
    glo_deq(x, acc, k1);
    for (int i = 0; i < 6; ++i) w[i] = x[i] + 0.5 * k1[i] * dt;
    glo_deq(w, acc, k2);
    for (int i = 0; i < 6; ++i) w[i] = x[i] + 0.5 * k2[i] * dt;
    glo_deq(w, acc, k3);
    for (int i = 0; i < 6; ++i) w[i] = x[i] + k3[i] * dt;
    glo_deq(w, acc, k4);
    for (int i = 0; i < 6; ++i) x[i] += (k1[i] + 2.0 * k2[i] + 2.0 * k3[i] + k4[i]) * dt / 6.0;
}

/* geph2pos (:692-708) and geph2vel (:710-726) run the same integration; dts = -tau_n + gamma dt, ddts = gamma */
static void geph2posvel(double t, const double* eph, double pos[3], double vel[3], double* svdt, double* svddt)
{
    double x[6];
    for (int i = 0; i < 3; ++i) { x[i] = eph[GLO_POS + i]; x[3 + i] = eph[GLO_VEL + i]; }
    double dt = wrap_week(t - eph[ORC_EPH_TOE]);
    *svdt = -eph[GLO_TAUN] + eph[GLO_GAMMA] * dt;
    *svddt = eph[GLO_GAMMA];
    for (double tt = dt < 0.0 ? -TSTEP : TSTEP; fabs(dt) > 1e-9; dt -= tt) {
        if (fabs(dt) < TSTEP) tt = dt;
        glo_orbit(tt, x, eph + GLO_ACC);
    }
    for (int i = 0; i < 3; ++i) { pos[i] = x[i]; vel[i] = x[3 + i]; }
}

/* sat_states (gnss_spp.cpp:50-98).  sat = {pos 3, vel 3, dt, ddt, tgd, ttx}; returns 0 without an L1 observation. */
int orc_gnss_sat_state(const double* eph, const double* obs, double* sat)
{
    memset(sat, 0, sizeof(double) * ORC_SAT_N);
    const int sys = (int)eph[ORC_EPH_SYS];
    if (sys < 0 || sys > 3 || obs[ORC_OBS_FREQ] < 0) return 0;
    const double tof = obs[ORC_OBS_PSR] / LIGHT_SPEED;
    double ttx = obs[ORC_OBS_TOW] - tof;
    if (sys == 1) {                                                                /* :72-79 */
        double svdt = geph2svdt(ttx, eph), svddt = 0.0;
        ttx -= svdt;
        geph2posvel(ttx, eph, sat, sat + 3, &svdt, &svddt);
        sat[6] = svdt; sat[7] = svddt; sat[8] = 0.0; sat[9] = ttx;                /* SatState::tgd keeps its default 0 */
        return 1;
    }
    double svdt = eph2svdt(ttx, eph), svddt = 0.0;
    ttx -= svdt;
    eph2posvel(ttx, eph, sat, sat + 3, &svdt, &svddt);
    sat[6] = svdt; sat[7] = svddt; sat[8] = eph[ORC_EPH_TGD]; sat[9] = ttx;
    return 1;
}

/* ecef2geo (gnss_utility.cpp:347-388): (lat deg, lon deg, alt m) */
void orc_gnss_ecef2geo(const double xyz[3], double lla[3])
{
    lla[0] = lla[1] = lla[2] = 0.0;
    if (xyz[0] == 0 && xyz[1] == 0) return;
    const double e2 = 6.69437999014e-3, a = 6378137.0, a2 = a * a, b2 = a2 * (1 - e2), b = sqrt(b2), ep2 = (a2 - b2) / b2;
    const double p = sqrt(xyz[0] * xyz[0] + xyz[1] * xyz[1]);
    double s1 = xyz[2] * a, s2 = p * b, h = sqrt(s1 * s1 + s2 * s2);
    const double st = s1 / h, ct = s2 / h;
    s1 = xyz[2] + ep2 * b * pow(st, 3);
    s2 = p - a * e2 * pow(ct, 3);
    h = sqrt(s1 * s1 + s2 * s2);
    const double sin_lat = s1 / h, cos_lat = s2 / h;
    const double N = a2 * pow(a2 * cos_lat * cos_lat + b2 * sin_lat * sin_lat, -0.5);
    lla[0] = atan(s1 / s2) / D2R; lla[1] = atan2(xyz[1], xyz[0]) / D2R; lla[2] = p / cos_lat - N;
}

/* sat_azel (:762-772) */
void orc_gnss_azel(const double rcv[3], const double sat[3], double azel[2])
{
    double lla[3];
    orc_gnss_ecef2geo(rcv, lla);
    double d[3] = { sat[0] - rcv[0], sat[1] - rcv[1], sat[2] - rcv[2] };
    const double nrm = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    for (int i = 0; i < 3; ++i) d[i] /= nrm;
    const double lat = lla[0] * D2R, lon = lla[1] * D2R, sl = sin(lat), cl = cos(lat), so = sin(lon), co = cos(lon);
    const double e = -so * d[0] + co * d[1];                                             /* ecef2enu, :733-743 */
    const double nn = -sl * co * d[0] - sl * so * d[1] + cl * d[2];
    const double u = cl * co * d[0] + cl * so * d[1] + sl * d[2];
    azel[0] = sqrt(d[0] * d[0] + d[1] * d[1]) < 1e-12 ? 0.0 : atan2(e, nn);
    if (azel[0] < 0) azel[0] += 2 * M_PI;
    azel[1] = asin(u);
}

static double interpc(const double coef[], double lat)      /* :774-779 */
{
    int i = (int)(lat / 15.0);
    if (i < 1) return coef[0]; else if (i > 4) return coef[4];
    return coef[i - 1] * (1.0 - lat / 15.0 + i) + coef[i] * (lat / 15.0 - i);
}
static double mapf(double el, double a, double b, double c) /* :782-786 */
{
    const double sinel = sin(el);
    return (1.0 + a / (1.0 + b / (1.0 + c))) / (sinel + (a / (sinel + b / (sinel + c))));
}

/* calculate_trop_delay (:841-863) with the Niell mapping functions (:797-839); doy = time2doy(transmit time) */
double orc_gnss_trop(double doy, const double lla[3], const double azel[2])
{
    static const double coef[][5] = {
        { 1.2769934E-3, 1.2683230E-3, 1.2465397E-3, 1.2196049E-3, 1.2045996E-3 },
        { 2.9153695E-3, 2.9152299E-3, 2.9288445E-3, 2.9022565E-3, 2.9024912E-3 },
        { 62.610505E-3, 62.837393E-3, 63.721774E-3, 63.824265E-3, 64.258455E-3 },
        { 0.0000000E-0, 1.2709626E-5, 2.6523662E-5, 3.4000452E-5, 4.1202191E-5 },
        { 0.0000000E-0, 2.1414979E-5, 3.0160779E-5, 7.2562722E-5, 11.723375E-5 },
        { 0.0000000E-0, 9.0128400E-5, 4.3497037E-5, 84.795348E-5, 170.37206E-5 },
        { 5.8021897E-4, 5.6794847E-4, 5.8118019E-4, 5.9727542E-4, 6.1641693E-4 },
        { 1.4275268E-3, 1.5138625E-3, 1.4572752E-3, 1.5007428E-3, 1.7599082E-3 },
        { 4.3472961E-2, 4.6729510E-2, 4.3908931E-2, 4.4626982E-2, 5.4736038E-2 } };
    const double aht[] = { 2.53E-5, 5.49E-3, 1.14E-3 };
    if (lla[2] < -100.0 || 1E4 < lla[2] || azel[1] <= 0) return 0.0;
    const double hgt = lla[2] < 0.0 ? 0.0 : lla[2];
    const double pres = 1013.25 * pow(1.0 - 2.2557E-5 * hgt, 5.2568);
    const double temp = 15.0 - 6.5E-3 * hgt + 273.16;
    const double e = 6.108 * 0.7 * exp((17.15 * temp - 4684.0) / (temp - 38.45));
    const double zhd = 0.0022768 * pres / (1.0 - 0.00266 * cos(2.0 * lla[0] * D2R) - 0.00028 * hgt / 1E3);
    const double zwd = 0.002277 * (1255.0 / temp + 0.05) * e;
    /* nmf: note it uses the ellipsoidal height lla[2] itself (not the clamped hgt) */
    const double el = azel[1];
    double lat = lla[0];
    const double y = (doy - 28.0) / 365.25 + (lat < 0.0 ? 0.5 : 0.0);
    const double cosy = cos(2.0 * M_PI * y);
    lat = fabs(lat);
    double ah[3], aw[3];
    for (int i = 0; i < 3; ++i) { ah[i] = interpc(coef[i], lat) - interpc(coef[i + 3], lat) * cosy; aw[i] = interpc(coef[i + 6], lat); }
    const double dm = (1.0 / sin(el) - mapf(el, aht[0], aht[1], aht[2])) * lla[2] / 1E3;
    const double mapfw = mapf(el, aw[0], aw[1], aw[2]), mapfh = mapf(el, ah[0], ah[1], ah[2]) + dm;
    return mapfh * zhd + mapfw * zwd;
}

/* calculate_ion_delay (:865-899): Klobuchar with the 8 broadcast parameters; tow = time2gpst(transmit time) */
double orc_gnss_iono(double tow, const double ion[8], const double lla[3], const double azel[2])
{
    if (lla[2] < -1E3 || azel[1] <= 0) return 0.0;
    const double psi = 0.0137 / (azel[1] / M_PI + 0.11) - 0.022;
    double phi = lla[0] / 180.0 + psi * cos(azel[0]);
    if (phi > 0.416) phi = 0.416; else if (phi < -0.416) phi = -0.416;
    const double lam = lla[1] / 180.0 + psi * sin(azel[0]) / cos(phi * M_PI);
    phi += 0.064 * cos((lam - 1.617) * M_PI);
    double tt = 43200.0 * lam + tow;
    tt -= floor(tt / 86400.0) * 86400.0;
    const double f = 1.0 + 16.0 * pow(0.53 - azel[1] / M_PI, 3.0);
    double amp = ion[0] + phi * (ion[1] + phi * (ion[2] + phi * ion[3]));
    double per = ion[4] + phi * (ion[5] + phi * (ion[6] + phi * ion[7]));
    amp = amp < 0.0 ? 0.0 : amp;
    per = per < 72000.0 ? 72000.0 : per;
    const double x = 2.0 * M_PI * (tt - 50400.0) / per;
    return LIGHT_SPEED * f * (fabs(x) < 1.57 ? 5E-9 + amp * (1.0 + x * x * (-0.5 + x * x / 24.0)) : 5E-9);
}

/* sat_states + psr_res (gnss_spp.cpp:100-146) + dopp_res (:256-282) for one epoch of ns satellites.
 * rcv_xyzt = (ecef xyz, clock bias of GPS / GLO / GAL / BDS, m); rcv_vel = (ecef velocity, clock drift m/s).
 * Outputs per satellite: res_pos, res_vel, los [3] (unit receiver -> satellite), azel [2], atmos [2] (ion, tro), sat [ORC_SAT_N];
 * usable[i] = 0 for satellites without a state (no L1 observation): their outputs are zero. */
void orc_gnss_residuals(int ns, const double* eph, const double* obs, const double ion[8], int have_ion, double doy,
                        const double rcv_xyzt[7], const double rcv_vel[4], double* res_pos, double* res_vel, double* los,
                        double* azel_out, double* atmos, double* sat_out, int* usable)
{
    double lla[3];
    orc_gnss_ecef2geo(rcv_xyzt, lla);
    const double rn = sqrt(rcv_xyzt[0] * rcv_xyzt[0] + rcv_xyzt[1] * rcv_xyzt[1] + rcv_xyzt[2] * rcv_xyzt[2]);
    for (int i = 0; i < ns; ++i) {
        const double* e = eph + (size_t)i * ORC_EPH_N;
        const double* o = obs + (size_t)i * ORC_OBS_N;
        double* sat = sat_out + (size_t)i * ORC_SAT_N;
        res_pos[i] = res_vel[i] = 0.0; los[3 * i] = los[3 * i + 1] = los[3 * i + 2] = 0.0;
        azel_out[2 * i] = 0.0; azel_out[2 * i + 1] = M_PI / 2.0; atmos[2 * i] = atmos[2 * i + 1] = 0.0;
        usable[i] = orc_gnss_sat_state(e, o, sat);
        if (!usable[i]) continue;
        const int sys = (int)e[ORC_EPH_SYS];
        double azel[2] = { 0, M_PI / 2.0 }, ion_d = 0, tro_d = 0;
        if (rn > 0) {
            orc_gnss_azel(rcv_xyzt, sat, azel);
            tro_d = orc_gnss_trop(doy, lla, azel);
            ion_d = have_ion ? orc_gnss_iono(sat[9], ion, lla, azel) : 0.0;
        }
        const double d[3] = { sat[0] - rcv_xyzt[0], sat[1] - rcv_xyzt[1], sat[2] - rcv_xyzt[2] };
        const double range = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        const double sag = OMG_GPS * (sat[0] * rcv_xyzt[1] - sat[1] * rcv_xyzt[0]) / LIGHT_SPEED;
        const double est = range + sag + rcv_xyzt[3 + sys] - sat[6] * LIGHT_SPEED + tro_d + ion_d + sat[8] * LIGHT_SPEED;
        res_pos[i] = est - o[ORC_OBS_PSR];
        for (int c = 0; c < 3; ++c) los[3 * i + c] = d[c] / range;
        azel_out[2 * i] = azel[0]; azel_out[2 * i + 1] = azel[1]; atmos[2 * i] = ion_d; atmos[2 * i + 1] = tro_d;
        const double sagd = OMG_GPS / LIGHT_SPEED * (sat[3] * rcv_xyzt[1] + sat[0] * rcv_vel[1] - sat[4] * rcv_xyzt[0] - sat[1] * rcv_vel[0]);
        const double estd = (sat[3] - rcv_vel[0]) * los[3 * i] + (sat[4] - rcv_vel[1]) * los[3 * i + 1] + (sat[5] - rcv_vel[2]) * los[3 * i + 2]
                            + rcv_vel[3] + sagd - sat[7] * LIGHT_SPEED;
        res_vel[i] = estd + o[ORC_OBS_DOPP] * (LIGHT_SPEED / o[ORC_OBS_FREQ]);
    }
}
