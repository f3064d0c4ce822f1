"""Fixtures for SURVEY.md 8(f) row f-3 (the gnss_comm front of the GNSS update): an independent numpy transcription of
gnss_comm/src/gnss_spp.cpp:50-146,256-282 and gnss_utility.cpp:347-388,390-640,733-772,774-899, evaluated on a synthetic but
physically sensible constellation (GPS / Galileo / BeiDou MEO + one BeiDou GEO + two GLONASS satellites, whose ephemeris is a PZ-90
state vector integrated by Runge-Kutta, gnss_utility.cpp:642-731 + one entry without an L1 observation that must be skipped).
Runs in the build container only; writes tests/golden/gnss_front.npz.  The C oracle (oracle/gnss_front_oracle.c) and the HIP
kernel are checked against these numbers; nothing of the reference travels.

    python oracle/gen_gnss_golden.py
"""
import math
import os

import numpy as np

C_LIGHT = 2.99792458e8
MU_GPS, MU = 3.9860050000e14, 3.9860044180e14
OMG_GPS, OMG_BDS = 7.2921151467e-5, 7.2921150000e-5
WEEK = 604800.0
SIN_N5, COS_N5 = -0.0871557427476582, 0.9961946980917456
D2R = math.pi / 180.0
(SYS, PRN, TOE, TOE_SYS, TOC, A, E, I0, OMG, OMG0, M0, DELTA_N, OMG_DOT, I_DOT, CUC, CUS, CRC, CRS, CIC, CIS, AF0, AF1, AF2, TGD,
 URA) = range(25)
TOW, PSR, DOPP, PSR_STD, DOPP_STD, FREQ = range(6)
# GLONASS record (sys == 1), same 25 doubles: sys, prn, toe (GPS week seconds), -, -, pos[3], vel[3], acc[3], tau_n, gamma; ura at 24
GLO_POS, GLO_VEL, GLO_ACC, GLO_TAUN, GLO_GAMMA = 5, 8, 11, 14, 15
OMG_GLO, RE_GLO, J2_GLO, TSTEP = 7.2921150000e-5, 6378136.0, 1.0826257E-3, 60.0


def wrap(t):
    return t - WEEK if t > WEEK / 2 else (t + WEEK if t < -WEEK / 2 else t)


def kepler(mk, es):
    e, ek, it = mk, 1e6, 0
    while it < 30 and abs(e - ek) > 1e-14:
        ek = e
        e -= (e - es * math.sin(e) - mk) / (1.0 - es * math.cos(e))
        it += 1
    return ek


def eph2svdt(t, ep):
    dt = wrap(t - ep[TOC])
    for _ in range(2):
        dt -= ep[AF0] + ep[AF1] * dt + ep[AF2] * dt * dt
    return ep[AF0] + ep[AF1] * dt + ep[AF2] * dt * dt


def _orbit(t, ep):
    sys = int(ep[SYS])
    mu = MU_GPS if sys == 0 else MU
    om = OMG_BDS if sys == 3 else OMG_GPS
    tk = wrap(t - ep[TOE])
    n = math.sqrt(mu / ep[A] ** 3) + ep[DELTA_N]
    Ek = kepler(ep[M0] + n * tk, ep[E])
    return sys, mu, om, tk, n, Ek


def eph2pos(t, ep):
    sys, mu, om, tk, n, Ek = _orbit(t, ep)
    sE, cE = math.sin(Ek), math.cos(Ek)
    vk = math.atan2(math.sqrt(1 - ep[E] ** 2) * sE, cE - ep[E])
    phi = vk + ep[OMG]
    c2, s2 = math.cos(2 * phi), math.sin(2 * phi)
    uk = phi + ep[CUS] * s2 + ep[CUC] * c2
    rk = ep[A] * (1 - ep[E] * cE) + ep[CRS] * s2 + ep[CRC] * c2
    ik = ep[I0] + ep[I_DOT] * tk + ep[CIS] * s2 + ep[CIC] * c2
    xk, yk = rk * math.cos(uk), rk * math.sin(uk)
    if sys == 3 and int(ep[PRN]) <= 5:
        Ok = ep[OMG0] + ep[OMG_DOT] * tk - om * ep[TOE_SYS]
        xg = xk * math.cos(Ok) - yk * math.cos(ik) * math.sin(Ok)
        yg = xk * math.sin(Ok) + yk * math.cos(ik) * math.cos(Ok)
        zg = yk * math.sin(ik)
        so, co = math.sin(om * tk), math.cos(om * tk)
        pos = np.array([xg * co + yg * so * COS_N5 + zg * so * SIN_N5, -xg * so + yg * co * COS_N5 + zg * co * SIN_N5,
                        -yg * SIN_N5 + zg * COS_N5])
    else:
        Ok = ep[OMG0] + (ep[OMG_DOT] - om) * tk - om * ep[TOE_SYS]
        pos = np.array([xk * math.cos(Ok) - yk * math.cos(ik) * math.sin(Ok), xk * math.sin(Ok) + yk * math.cos(ik) * math.cos(Ok),
                        yk * math.sin(ik)])
    dt = wrap(t - ep[TOC])
    dts = ep[AF0] + ep[AF1] * dt + ep[AF2] * dt * dt - 2.0 * math.sqrt(mu * ep[A]) * ep[E] * sE / C_LIGHT / C_LIGHT
    return pos, dts


def eph2vel(t, ep):
    sys, mu, om, tk, n, Ek = _orbit(t, ep)
    e = ep[E]
    sE, cE = math.sin(Ek), math.cos(Ek)
    Ed = n / (1 - e * cE)
    vd = math.sqrt(1 - e * e) * Ed / (1 - e * cE)
    vk = math.atan2(math.sqrt(1 - e * e) * sE, cE - e)
    phi = vk + ep[OMG]
    c2, s2 = math.cos(2 * phi), math.sin(2 * phi)
    ud = vd + 2 * vd * (ep[CUS] * c2 - ep[CUC] * s2)
    rd = ep[A] * e * Ed * sE + 2 * vd * (ep[CRS] * c2 - ep[CRC] * s2)
    idot = ep[I_DOT] + 2 * vd * (ep[CIS] * c2 - ep[CIC] * s2)
    uk = phi + ep[CUS] * s2 + ep[CUC] * c2
    rk = ep[A] * (1 - e * cE) + ep[CRS] * s2 + ep[CRC] * c2
    ik = ep[I0] + ep[I_DOT] * tk + ep[CIS] * s2 + ep[CIC] * c2
    su, cu, si, ci = math.sin(uk), math.cos(uk), math.sin(ik), math.cos(ik)
    xk, yk = rk * cu, rk * su
    xd, yd = rd * cu - rk * ud * su, rd * su + rk * ud * cu
    if sys == 3 and int(ep[PRN]) <= 5:
        Ok = ep[OMG0] + ep[OMG_DOT] * tk - om * ep[TOE_SYS]
        sO, cO, Od = math.sin(Ok), math.cos(Ok), ep[OMG_DOT]
        t1 = xd - yk * Od * ci
        t2 = xk * Od + yd * ci - yk * idot * si
        xg, yg, zg = xk * cO - yk * ci * sO, xk * sO + yk * ci * cO, yk * si
        xgd, ygd = t1 * cO - t2 * sO, t1 * sO + t2 * cO
        zgd = yd * si + yd * idot * ci                      # as written (gnss_utility.cpp:617)
        so, co = math.sin(om * tk), math.cos(om * tk)
        sod, cod = om * co, -om * so
        vel = np.array([xgd * co + xg * cod + ygd * so * COS_N5 + yg * sod * COS_N5 + zgd * so * SIN_N5 + zg * sod * SIN_N5,
                        -xgd * so - xg * sod + ygd * co * COS_N5 + yg * cod * COS_N5 + zgd * co * SIN_N5 + zg * cod * SIN_N5,
                        -ygd * SIN_N5 + zgd * COS_N5])
    else:
        Ok = ep[OMG0] + (ep[OMG_DOT] - om) * tk - om * ep[TOE_SYS]
        sO, cO, Od = math.sin(Ok), math.cos(Ok), ep[OMG_DOT] - om
        t1 = xd - yk * Od * ci
        t2 = xk * Od + yd * ci - yk * idot * si
        vel = np.array([t1 * cO - t2 * sO, t1 * sO + t2 * cO, yd * si + yd * idot * ci])      # z: as written (:632)
    dt = wrap(t - ep[TOC])
    ddts = ep[AF1] + 2.0 * ep[AF2] * dt - 2.0 * math.sqrt(mu * ep[A]) * e * cE * Ed / C_LIGHT / C_LIGHT
    return vel, ddts


def geph2svdt(t, ep):                                          # gnss_utility.cpp:679-690
    dt = wrap(t - ep[TOE])
    for _ in range(2):
        dt -= -ep[GLO_TAUN] + ep[GLO_GAMMA] * dt
    return -ep[GLO_TAUN] + ep[GLO_GAMMA] * dt


def glo_deq(x, acc):                                           # :642-660 (PZ-90 force model: central term, J2, frame rotation)
    pos, vel = x[:3], x[3:]
    r2 = float(pos @ pos)
    if r2 <= 0.0:
        return np.zeros(6)
    r3 = r2 * math.sqrt(r2)
    omg2 = OMG_GLO * OMG_GLO
    a = 1.5 * J2_GLO * MU * RE_GLO * RE_GLO / r2 / r3
    b = 5.0 * pos[2] * pos[2] / r2
    c = -MU / r3 - a * (1.0 - b)
    return np.array([vel[0], vel[1], vel[2],
                     (c + omg2) * pos[0] + 2.0 * OMG_GLO * vel[1] + acc[0],
                     (c + omg2) * pos[1] - 2.0 * OMG_GLO * vel[0] + acc[1],
                     (c - 2.0 * a) * pos[2] + acc[2]])


def glo_orbit(dt, x, acc):                                     # :662-677, classical RK4
    k1 = glo_deq(x, acc)
    k2 = glo_deq(x + 0.5 * dt * k1, acc)
    k3 = glo_deq(x + 0.5 * dt * k2, acc)
    k4 = glo_deq(x + dt * k3, acc)
    return x + (k1 + 2.0 * k2 + 2.0 * k3 + k4) * dt / 6.0


def geph2posvel(t, ep):                                        # geph2pos :692-708, geph2vel :710-726 (the same integration twice)
    x = np.r_[ep[GLO_POS:GLO_POS + 3], ep[GLO_VEL:GLO_VEL + 3]].astype(float)
    acc = ep[GLO_ACC:GLO_ACC + 3]
    dt = wrap(t - ep[TOE])                                     # time_diff of absolute times: the wrap of week seconds
    dts = -ep[GLO_TAUN] + ep[GLO_GAMMA] * dt
    tt = -TSTEP if dt < 0.0 else TSTEP
    while abs(dt) > 1e-9:
        if abs(dt) < TSTEP:
            tt = dt
        x = glo_orbit(tt, x, acc)
        dt -= tt
    return x[:3], x[3:], dts, ep[GLO_GAMMA]


def ecef2geo(x):
    e2, a = 6.69437999014e-3, 6378137.0
    a2 = a * a; b2 = a2 * (1 - e2); b = math.sqrt(b2); ep2 = (a2 - b2) / b2
    p = math.hypot(x[0], x[1])
    s1, s2 = x[2] * a, p * b
    h = math.hypot(s1, s2)
    st, ct = s1 / h, s2 / h
    s1 = x[2] + ep2 * b * st ** 3
    s2 = p - a * e2 * ct ** 3
    h = math.hypot(s1, s2)
    sl, cl = s1 / h, s2 / h
    N = a2 * (a2 * cl * cl + b2 * sl * sl) ** -0.5
    return np.array([math.atan(s1 / s2) / D2R, math.atan2(x[1], x[0]) / D2R, p / cl - N])


def geo2ecef(lla):
    e2, a = 6.69437999014e-3, 6378137.0
    sl, cl = math.sin(lla[0] * D2R), math.cos(lla[0] * D2R)
    N = a / math.sqrt(1 - e2 * sl * sl)
    return np.array([(N + lla[2]) * cl * math.cos(lla[1] * D2R), (N + lla[2]) * cl * math.sin(lla[1] * D2R), (N * (1 - e2) + lla[2]) * sl])


def sat_azel(rcv, sat):
    lla = ecef2geo(rcv)
    d = (sat - rcv) / np.linalg.norm(sat - rcv)
    lat, lon = lla[0] * D2R, lla[1] * D2R
    R = np.array([[-math.sin(lon), math.cos(lon), 0], [-math.sin(lat) * math.cos(lon), -math.sin(lat) * math.sin(lon), math.cos(lat)],
                  [math.cos(lat) * math.cos(lon), math.cos(lat) * math.sin(lon), math.sin(lat)]])
    enu = R @ d
    az = 0.0 if math.hypot(d[0], d[1]) < 1e-12 else math.atan2(enu[0], enu[1])
    return np.array([az + (2 * math.pi if az < 0 else 0), math.asin(enu[2])])


NMF = np.array([[1.2769934E-3, 1.2683230E-3, 1.2465397E-3, 1.2196049E-3, 1.2045996E-3],
                [2.9153695E-3, 2.9152299E-3, 2.9288445E-3, 2.9022565E-3, 2.9024912E-3],
                [62.610505E-3, 62.837393E-3, 63.721774E-3, 63.824265E-3, 64.258455E-3],
                [0.0, 1.2709626E-5, 2.6523662E-5, 3.4000452E-5, 4.1202191E-5],
                [0.0, 2.1414979E-5, 3.0160779E-5, 7.2562722E-5, 11.723375E-5],
                [0.0, 9.0128400E-5, 4.3497037E-5, 84.795348E-5, 170.37206E-5],
                [5.8021897E-4, 5.6794847E-4, 5.8118019E-4, 5.9727542E-4, 6.1641693E-4],
                [1.4275268E-3, 1.5138625E-3, 1.4572752E-3, 1.5007428E-3, 1.7599082E-3],
                [4.3472961E-2, 4.6729510E-2, 4.3908931E-2, 4.4626982E-2, 5.4736038E-2]])


def interpc(coef, lat):
    i = int(lat / 15.0)
    if i < 1:
        return coef[0]
    if i > 4:
        return coef[4]
    return coef[i - 1] * (1.0 - lat / 15.0 + i) + coef[i] * (lat / 15.0 - i)


def mapf(el, a, b, c):
    s = math.sin(el)
    return (1.0 + a / (1.0 + b / (1.0 + c))) / (s + (a / (s + b / (s + c))))


def trop(doy, lla, azel):
    if lla[2] < -100.0 or 1E4 < lla[2] or azel[1] <= 0:
        return 0.0
    hgt = max(lla[2], 0.0)
    pres = 1013.25 * (1.0 - 2.2557E-5 * hgt) ** 5.2568
    temp = 15.0 - 6.5E-3 * hgt + 273.16
    e = 6.108 * 0.7 * math.exp((17.15 * temp - 4684.0) / (temp - 38.45))
    zhd = 0.0022768 * pres / (1.0 - 0.00266 * math.cos(2.0 * lla[0] * D2R) - 0.00028 * hgt / 1E3)
    zwd = 0.002277 * (1255.0 / temp + 0.05) * e
    el, lat = azel[1], lla[0]
    y = (doy - 28.0) / 365.25 + (0.5 if lat < 0 else 0.0)
    cosy = math.cos(2.0 * math.pi * y)
    lat = abs(lat)
    ah = [interpc(NMF[i], lat) - interpc(NMF[i + 3], lat) * cosy for i in range(3)]
    aw = [interpc(NMF[i + 6], lat) for i in range(3)]
    dm = (1.0 / math.sin(el) - mapf(el, 2.53E-5, 5.49E-3, 1.14E-3)) * lla[2] / 1E3
    return (mapf(el, *ah) + dm) * zhd + mapf(el, *aw) * zwd


def iono(tow, ion, lla, azel):
    if lla[2] < -1E3 or azel[1] <= 0:
        return 0.0
    psi = 0.0137 / (azel[1] / math.pi + 0.11) - 0.022
    phi = min(max(lla[0] / 180.0 + psi * math.cos(azel[0]), -0.416), 0.416)
    lam = lla[1] / 180.0 + psi * math.sin(azel[0]) / math.cos(phi * math.pi)
    phi += 0.064 * math.cos((lam - 1.617) * math.pi)
    tt = 43200.0 * lam + tow
    tt -= math.floor(tt / 86400.0) * 86400.0
    f = 1.0 + 16.0 * (0.53 - azel[1] / math.pi) ** 3
    amp = max(ion[0] + phi * (ion[1] + phi * (ion[2] + phi * ion[3])), 0.0)
    per = max(ion[4] + phi * (ion[5] + phi * (ion[6] + phi * ion[7])), 72000.0)
    x = 2.0 * math.pi * (tt - 50400.0) / per
    return C_LIGHT * f * (5E-9 + amp * (1.0 + x * x * (-0.5 + x * x / 24.0)) if abs(x) < 1.57 else 5E-9)


def sat_state(ep, ob):
    if ob[FREQ] < 0:
        return None
    ttx = ob[TOW] - ob[PSR] / C_LIGHT
    if int(ep[SYS]) == 1:                                      # gnss_spp.cpp:72-79
        ttx -= geph2svdt(ttx, ep)
        pos, vel, dt, ddt = geph2posvel(ttx, ep)
        return pos, vel, dt, ddt, 0.0, ttx
    ttx -= eph2svdt(ttx, ep)
    pos, dt = eph2pos(ttx, ep)
    vel, ddt = eph2vel(ttx, ep)
    return pos, vel, dt, ddt, ep[TGD], ttx


def residuals(eph, obs, ion, doy, xyzt, velt):
    ns = len(eph)
    out = dict(res_pos=np.zeros(ns), res_vel=np.zeros(ns), los=np.zeros((ns, 3)), azel=np.zeros((ns, 2)), atmos=np.zeros((ns, 2)),
               sat=np.zeros((ns, 10)), usable=np.zeros(ns, dtype=np.int32))
    lla = ecef2geo(xyzt[:3])
    for i in range(ns):
        st = sat_state(eph[i], obs[i])
        out["azel"][i] = [0.0, math.pi / 2]
        if st is None:
            continue
        pos, vel, dt, ddt, tgd, ttx = st
        out["usable"][i] = 1
        out["sat"][i] = np.r_[pos, vel, dt, ddt, tgd, ttx]
        azel = sat_azel(xyzt[:3], pos)
        tro, io = trop(doy, lla, azel), (iono(ttx, ion, lla, azel) if ion is not None else 0.0)
        d = pos - xyzt[:3]
        rng = np.linalg.norm(d)
        sag = OMG_GPS * (pos[0] * xyzt[1] - pos[1] * xyzt[0]) / C_LIGHT
        est = rng + sag + xyzt[3 + int(eph[i][SYS])] - dt * C_LIGHT + tro + io + tgd * C_LIGHT
        out["res_pos"][i] = est - obs[i][PSR]
        u = d / rng
        out["los"][i] = u; out["azel"][i] = azel; out["atmos"][i] = [io, tro]
        sagd = OMG_GPS / C_LIGHT * (vel[0] * xyzt[1] + pos[0] * velt[1] - vel[1] * xyzt[0] - pos[1] * velt[0])
        estd = (vel - velt[:3]) @ u + velt[3] + sagd - ddt * C_LIGHT
        out["res_vel"][i] = estd + obs[i][DOPP] * (C_LIGHT / obs[i][FREQ])
    return out


def make_constellation(rng, rcv, t_rx, want=(("gps", 0, 4), ("bds", 3, 2), ("gal", 2, 2))):
    """Random Kepler ephemerides per constellation; keeps those above 15 degrees at the receiver.  Returns [ns, 25]."""
    semi = dict(gps=26560e3, gal=29600e3, bds=27906e3)
    ephs = []
    for name, sysid, cnt in want:
        got, prn = 0, 6
        while got < cnt:
            prn += 1
            ep = np.zeros(25)
            ep[SYS], ep[PRN] = sysid, prn
            ep[TOE] = t_rx - 1800.0 + rng.uniform(-600, 600); ep[TOC] = ep[TOE]
            ep[TOE_SYS] = ep[TOE] - (14.0 if sysid == 3 else 0.0)
            ep[A] = semi[name] * (1 + rng.uniform(-1e-3, 1e-3)); ep[E] = rng.uniform(0.001, 0.02)
            ep[I0] = 0.96 + rng.uniform(-0.03, 0.03); ep[OMG] = rng.uniform(-math.pi, math.pi)
            ep[OMG0] = rng.uniform(-math.pi, math.pi); ep[M0] = rng.uniform(-math.pi, math.pi)
            ep[DELTA_N] = rng.uniform(3e-9, 6e-9); ep[OMG_DOT] = rng.uniform(-9e-9, -7e-9); ep[I_DOT] = rng.uniform(-5e-10, 5e-10)
            ep[CUC:CIS + 1] = rng.uniform(-1, 1, 6) * np.array([5e-6, 5e-6, 300.0, 100.0, 2e-7, 2e-7])
            ep[AF0], ep[AF1], ep[AF2] = rng.uniform(-5e-4, 5e-4), rng.uniform(-1e-11, 1e-11), 0.0
            ep[TGD], ep[URA] = rng.uniform(-1e-8, 1e-8), 2.0
            pos, _ = eph2pos(t_rx - 0.075, ep)
            if sat_azel(rcv, pos)[1] > 15 * D2R:
                ephs.append(ep); got += 1
    return np.array(ephs)


def make_glonass(rng, rcv, t_rx, cnt=2):
    """GLONASS broadcast records: a circular orbit (r = 25510 km, i = 64.8 deg) sampled at toe, expressed in the rotating PZ-90
    frame (v_ecef = v_inertial - omega x r), lunisolar acceleration of the broadcast order of magnitude; keeps satellites above
    20 degrees.  toe within the +-15 min validity of a GLONASS ephemeris."""
    out, prn = [], 0
    while len(out) < cnt:
        prn += 1
        r, inc = 25510e3, 64.8 * D2R
        raan, u = rng.uniform(-math.pi, math.pi), rng.uniform(-math.pi, math.pi)
        vcirc = math.sqrt(MU / r)
        Rz = np.array([[math.cos(raan), -math.sin(raan), 0], [math.sin(raan), math.cos(raan), 0], [0, 0, 1.0]])
        Rx = np.array([[1, 0, 0], [0, math.cos(inc), -math.sin(inc)], [0, math.sin(inc), math.cos(inc)]])
        pos = Rz @ Rx @ np.array([r * math.cos(u), r * math.sin(u), 0.0])
        vin = Rz @ Rx @ np.array([-vcirc * math.sin(u), vcirc * math.cos(u), 0.0])
        vel = vin - np.cross([0, 0, OMG_GLO], pos)
        ep = np.zeros(25)
        ep[SYS], ep[PRN] = 1, prn
        ep[TOE] = t_rx - rng.uniform(120.0, 800.0) * (1 if len(out) == 0 else -1)      # one integrates forwards, one backwards
        ep[GLO_POS:GLO_POS + 3], ep[GLO_VEL:GLO_VEL + 3] = pos, vel
        ep[GLO_ACC:GLO_ACC + 3] = rng.uniform(-2e-6, 2e-6, 3)
        ep[GLO_TAUN], ep[GLO_GAMMA] = rng.uniform(-2e-4, 2e-4), rng.uniform(-2e-12, 2e-12)
        ep[URA] = 2.0
        p_now = geph2posvel(t_rx - 0.075, ep)[0]
        if sat_azel(rcv, p_now)[1] > 20 * D2R:
            out.append(ep)
    return np.array(out)


def make_obs(rng, eph, rcv, vel, cb, fs, ion, doy, t_rx, noise=True):
    """L1 observations consistent with receiver (rcv, vel), clock biases cb[4] (m) and drift fs (m/s)."""
    obs = np.zeros((len(eph), 6))
    lla = ecef2geo(rcv)
    for i, ep in enumerate(eph):
        obs[i, TOW] = t_rx
        glo = int(ep[SYS]) == 1
        obs[i, FREQ] = 1.561098e9 if int(ep[SYS]) == 3 else (1.602e9 + (int(ep[PRN]) - 3) * 0.5625e6 if glo else 1.57542e9)      # GLONASS FDMA channel
        obs[i, PSR_STD], obs[i, DOPP_STD] = 1.0, 0.5
        ttx = t_rx - 0.075
        for _ in range(6):                                      # light-time iteration
            pos, dts = (geph2posvel(ttx, ep)[0], geph2posvel(ttx, ep)[2]) if glo else eph2pos(ttx, ep)
            rng_ = np.linalg.norm(pos - rcv)
            ttx = t_rx - rng_ / C_LIGHT
        if glo:
            _, velv, _, ddts = geph2posvel(ttx, ep)
        else:
            velv, ddts = eph2vel(ttx, ep)
        azel = sat_azel(rcv, pos)
        sag = OMG_GPS * (pos[0] * rcv[1] - pos[1] * rcv[0]) / C_LIGHT
        obs[i, PSR] = rng_ + sag + cb[int(ep[SYS])] - dts * C_LIGHT + trop(doy, lla, azel) + iono(ttx, ion, lla, azel) + (0.0 if glo else ep[TGD]) * C_LIGHT \
            + (rng.normal(0, 0.8) if noise else 0.0)
        u = (pos - rcv) / rng_
        sagd = OMG_GPS / C_LIGHT * (velv[0] * rcv[1] + pos[0] * vel[1] - velv[1] * rcv[0] - pos[1] * vel[0])
        est = (velv - vel) @ u + fs + sagd - ddts * C_LIGHT
        obs[i, DOPP] = -(est + (rng.normal(0, 0.05) if noise else 0.0)) * obs[i, FREQ] / C_LIGHT
    return obs


def main():
    rng = np.random.default_rng(20260927)
    t_rx, doy = 360300.0, 270.4
    rcv = geo2ecef(np.array([31.0, 121.4, 30.0]))
    vel = np.array([1.2, -0.7, 0.3])
    cb = np.array([150.0, 140.0, 165.0, 180.0]); fs = 5.0
    ion = np.array([0.1118e-07, 0.2235e-07, -0.1192e-06, -0.1192e-06, 0.1167e+06, 0.1802e+06, -0.1311e+06, -0.4588e+06])
    eph = make_constellation(rng, rcv, t_rx)
    geo = eph[4].copy()                                         # a BeiDou GEO entry (prn <= 5 takes the rotated-frame branch)
    geo[PRN], geo[A], geo[E], geo[I0] = 3, 42164e3, 3e-4, 0.08
    for _ in range(4000):
        geo[OMG0], geo[M0] = rng.uniform(-math.pi, math.pi), rng.uniform(-math.pi, math.pi)
        if sat_azel(rcv, eph2pos(t_rx - 0.12, geo)[0])[1] > 20 * D2R:
            break
    glo = make_glonass(rng, rcv, t_rx)                          # two GLONASS satellites (Runge-Kutta orbit)
    nol1 = eph[1].copy()                                        # an entry whose observation has no L1 signal: must come back unusable
    eph = np.vstack([eph, geo, glo, nol1])
    obs = make_obs(rng, eph, rcv, vel, cb, fs, ion, doy, t_rx)
    obs[-1, FREQ] = -1.0
    # evaluate at a perturbed receiver state (what the filter would hold)
    xyzt = np.r_[rcv + np.array([3.0, -2.0, 1.5]), cb + np.array([2.0, -1.5, -1.0, 1.5])]
    velt = np.r_[vel + np.array([0.05, -0.02, 0.01]), fs + 0.1]
    out = residuals(eph, obs, ion, doy, xyzt, velt)
    out_noion = residuals(eph, obs, None, doy, xyzt, velt)
    here = os.path.dirname(os.path.abspath(__file__))
    path = os.path.join(here, "..", "tests", "golden", "gnss_front.npz")
    np.savez_compressed(path, eph=eph, obs=obs, ion=ion, doy=doy, xyzt=xyzt, velt=velt, rcv_true=rcv, vel_true=vel, cb_true=cb,
                        fs_true=fs, **{k: v for k, v in out.items()}, res_pos_noion=out_noion["res_pos"])
    print("wrote", os.path.normpath(path), "sats", len(eph), "usable", int(out["usable"].sum()))
    print("res_pos", np.round(out["res_pos"], 3)); print("res_vel", np.round(out["res_vel"], 4))
    print("el deg", np.round(out["azel"][:, 1] / D2R, 1)); print("atmos", np.round(out["atmos"], 3))


if __name__ == "__main__":
    main()
