#include "GnssComm.h"

#include <cmath>

namespace ingvio {
namespace gnss {

static const double D2R = M_PI / 180.0, R2D = 180.0 / M_PI;

Vec3d geo2ecef(const Vec3d& lla)
{
    const double cos_lat = std::cos(lla[0] * D2R), sin_lat = std::sin(lla[0] * D2R);
    const double N = EARTH_SEMI_MAJOR / std::sqrt(1 - EARTH_ECCE_2 * sin_lat * sin_lat);
    return Vec3d((N + lla[2]) * cos_lat * std::cos(lla[1] * D2R), (N + lla[2]) * cos_lat * std::sin(lla[1] * D2R),
                 (N * (1 - EARTH_ECCE_2) + lla[2]) * sin_lat);
}

Vec3d ecef2geo(const Vec3d& xyz)
{
    if (xyz[0] == 0 && xyz[1] == 0) return Vec3d();                    // "LLA coordinate is not defined if x = 0 and y = 0"
    const double e2 = EARTH_ECCE_2, a = EARTH_SEMI_MAJOR, a2 = a * a, b2 = a2 * (1 - e2), b = std::sqrt(b2), ep2 = (a2 - b2) / b2;
    const double p = std::sqrt(xyz[0] * xyz[0] + xyz[1] * xyz[1]);
    double s1 = xyz[2] * a, s2 = p * b, h = std::sqrt(s1 * s1 + s2 * s2);
    const double sin_theta = s1 / h, cos_theta = s2 / h;
    s1 = xyz[2] + ep2 * b * std::pow(sin_theta, 3);
    s2 = p - a * e2 * std::pow(cos_theta, 3);
    h = std::sqrt(s1 * s1 + s2 * s2);
    const double tan_lat = s1 / s2, sin_lat = s1 / h, cos_lat = s2 / h;
    const double lat = std::atan(tan_lat);
    const double N = a2 * std::pow(a2 * cos_lat * cos_lat + b2 * sin_lat * sin_lat, -0.5);
    return Vec3d(lat * R2D, std::atan2(xyz[1], xyz[0]) * R2D, p / cos_lat - N);
}

Mat3d geo2rotation(const Vec3d& g)
{
    const double lat = g[0] * D2R, lon = g[1] * D2R, sl = std::sin(lat), cl = std::cos(lat), so = std::sin(lon), co = std::cos(lon);
    Mat3d R;
    R(0, 0) = -so; R(0, 1) = -sl * co; R(0, 2) = cl * co;
    R(1, 0) = co;  R(1, 1) = -sl * so; R(1, 2) = cl * so;
    R(2, 0) = 0;   R(2, 1) = cl;       R(2, 2) = sl;
    return R;
}

Vec3d ecef2enu(const Vec3d& ref_lla, const Vec3d& v) { return geo2rotation(ref_lla).transpose() * v; }

void sat_azel(const Vec3d& rcv, const Vec3d& sat, double azel[2])
{
    const Vec3d lla = ecef2geo(rcv);
    Vec3d d = sat - rcv;
    d = d * (1.0 / d.norm());
    const Vec3d e = ecef2enu(lla, d);
    azel[0] = std::sqrt(d[0] * d[0] + d[1] * d[1]) < 1e-12 ? 0.0 : std::atan2(e[0], e[1]);
    azel[0] += (azel[0] < 0 ? 2 * M_PI : 0);
    azel[1] = std::asin(e[2]);
}

void psr_res(const double rcv[7], const std::vector<SatObs>& obs, std::vector<double>& res, std::vector<Vec3d>& los,
             std::vector<double>& az, std::vector<double>& el)
{
    const size_t n = obs.size();
    res.assign(n, 0.0); los.assign(n, Vec3d()); az.assign(n, 0.0); el.assign(n, M_PI / 2.0);
    const Vec3d r(rcv[0], rcv[1], rcv[2]);
    for (size_t i = 0; i < n; ++i) {
        const SatObs& o = obs[i];
        if (o.sys < 0 || o.sys > 3) continue;
        double azel[2] = { 0, M_PI / 2.0 };
        if (r.norm() > 0) sat_azel(r, o.sv_pos, azel);                                          // :121-128
        const Vec3d rv2sv = o.sv_pos - r;
        const double range = rv2sv.norm();
        const double sagnac = EARTH_OMG_GPS * (o.sv_pos[0] * rcv[1] - o.sv_pos[1] * rcv[0]) / LIGHT_SPEED;       // :132-133
        const double est = range + sagnac + rcv[3 + o.sys] - o.sv_dt * LIGHT_SPEED + o.tro_delay + o.ion_delay + o.tgd * LIGHT_SPEED;
        los[i] = rv2sv * (1.0 / range);
        res[i] = est - o.psr;                                                                    // :139
        az[i] = azel[0]; el[i] = azel[1];
    }
}

void dopp_res(const double rcv[4], const Vec3d& rcv_ecef, const std::vector<SatObs>& obs, std::vector<double>& res)
{
    const size_t n = obs.size();
    res.assign(n, 0.0);
    for (size_t i = 0; i < n; ++i) {
        const SatObs& o = obs[i];
        Vec3d u = o.sv_pos - rcv_ecef;
        u = u * (1.0 / u.norm());
        const double sagnac = EARTH_OMG_GPS / LIGHT_SPEED *
            (o.sv_vel[0] * rcv_ecef[1] + o.sv_pos[0] * rcv[1] - o.sv_vel[1] * rcv_ecef[0] - o.sv_pos[1] * rcv[0]);      // :267-269
        const Vec3d dv(o.sv_vel[0] - rcv[0], o.sv_vel[1] - rcv[1], o.sv_vel[2] - rcv[2]);
        const double est = dv[0] * u[0] + dv[1] * u[1] + dv[2] * u[2] + rcv[3] + sagnac - o.sv_ddt * LIGHT_SPEED;
        if (o.freq < 0) continue;
        res[i] = est + o.dopp * (LIGHT_SPEED / o.freq);                                          // :276-277
    }
}

}  // namespace gnss
}  // namespace ingvio
