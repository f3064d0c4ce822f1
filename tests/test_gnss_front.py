"""the gnss_comm front of the GNSS update (ephemeris -> satellite state, atmosphere models, pseudo-range /
Doppler residuals).  The reference holds no test for gnss_comm, so the oracle (oracle/gnss_front_oracle.c) is pinned by
(1) an independent numpy transcription (oracle/gen_gnss_golden.py -> tests/golden/gnss_front.npz) and (2) identities the
physics imposes; the HIP kernel (kernels_gnss.hip) is then checked against the oracle satellite by satellite and through the
whole update."""
import math

import numpy as np
import pytest

from conftest import load_golden, rel_err

C_LIGHT = 2.99792458e8


def _z():
    return load_golden("gnss_front")


def test_oracle_matches_numpy_transcription(orc):
    z = _z()
    o = orc.gnss_residuals(z["eph"], z["obs"], z["ion"], float(z["doy"]), z["xyzt"], z["velt"])
    assert np.array_equal(o["usable"], z["usable"]) and o["usable"].sum() == 11 and o["usable"][-1] == 0      # the entry without L1 is skipped
    assert (z["eph"][:, 0] == 1).sum() == 2                                                                     # two GLONASS satellites
    assert np.abs(o["sat"][:, :3] - z["sat"][:, :3]).max() < 1e-6          # satellite position, m (range 2.6e7: 4e-14 relative)
    assert np.abs(o["sat"][:, 3:6] - z["sat"][:, 3:6]).max() < 1e-9 and np.abs(o["sat"][:, 6:9] - z["sat"][:, 6:9]).max() < 1e-15
    assert np.abs(o["res_pos"] - z["res_pos"]).max() < 1e-6 and np.abs(o["res_vel"] - z["res_vel"]).max() < 1e-9
    assert np.abs(o["los"] - z["los"]).max() < 1e-13 and np.abs(o["azel"] - z["azel"]).max() < 1e-12
    assert np.abs(o["atmos"] - z["atmos"]).max() < 1e-10
    o2 = orc.gnss_residuals(z["eph"], z["obs"], None, float(z["doy"]), z["xyzt"], z["velt"])
    assert np.abs(o2["res_pos"] - z["res_pos_noion"]).max() < 1e-6 and not o2["atmos"][:, 0].any()


def test_satellite_velocity_is_the_time_derivative_of_position(orc):
    """eph2vel against central differences of eph2pos (transmit time shifted through the receive time of the observation)."""
    z = _z()
    eph, obs = z["eph"], z["obs"]
    h = 0.5
    for i in range(len(eph) - 1):
        sat = {}
        for s in (-1, 0, 1):
            ob = obs[i].copy(); ob[0] += s * h
            sat[s] = orc.gnss_residuals(eph[i:i + 1], ob[None], None, float(z["doy"]), z["xyzt"], z["velt"])["sat"][0]
        fd = (sat[1][:3] - sat[-1][:3]) / (sat[1][9] - sat[-1][9])
        v = sat[0][3:6]
        geo = int(eph[i][0]) == 3 and int(eph[i][1]) <= 5             # the geostationary entry barely moves in ECEF
        glo = int(eph[i][0]) == 1
        assert (np.linalg.norm(v) < 200.0 if geo else 2300.0 < np.linalg.norm(v) < 4200.0) and 2.0e7 < np.linalg.norm(sat[0][:3]) < 4.3e7
        if glo:      # geph2vel integrates the very ODE whose position part is d pos / dt = vel: exact up to the step error of RK4
            assert np.abs(fd - v).max() < 2e-5, (i, fd, v)          # central differences over +-0.5 s: truncation h^2 |jerk| / 6 ~ 5e-6
            assert abs(abs(sat[0][7]) - abs(eph[i][15])) < 1e-18          # clock drift = gamma
            continue
        # (for the GEO entry the as-written z term below leaks into x / y through the 5-degree frame rotation: 6e-4 m/s)
        assert np.abs(fd[:2] - v[:2]).max() < (1e-3 if geo else 2e-4), (i, fd, v)
        # z component: the reference's last term reads y'_k i_dot cos(i) where the derivative has y_k i_dot cos(i)
        # (gnss_utility.cpp:632 / :617) - reproduced as written; the two differ by < 1 cm/s
        assert abs(fd[2] - v[2]) < 1.5e-2, (i, fd[2], v[2])
        # clock drift = d/dt of the clock correction
        assert abs((sat[1][6] - sat[-1][6]) / (sat[1][9] - sat[-1][9]) - sat[0][7]) < 1e-13


def test_residuals_vanish_at_the_true_receiver_state(orc):
    """The observations of the fixture were generated from (rcv_true, vel_true, cb_true, fs_true) with 0.8 m / 5 cm/s noise."""
    z = _z()
    o = orc.gnss_residuals(z["eph"], z["obs"], z["ion"], float(z["doy"]), np.r_[z["rcv_true"], z["cb_true"]], np.r_[z["vel_true"], z["fs_true"]])
    u = o["usable"] == 1
    assert np.abs(o["res_pos"][u]).max() < 4.0 and np.abs(o["res_vel"][u]).max() < 0.25
    assert abs(o["res_pos"][u].mean()) < 1.5


def test_jacobian_of_the_pseudorange_is_minus_the_line_of_sight(orc):
    """psr_res returns J = [-unit_rv2sv | clock one-hot] (gnss_spp.cpp:137-138): finite differences of res_pos in the receiver position."""
    z = _z()
    base = orc.gnss_residuals(z["eph"], z["obs"], z["ion"], float(z["doy"]), z["xyzt"], z["velt"])
    h = 10.0
    for k in range(3):
        d = np.zeros(7); d[k] = h
        up = orc.gnss_residuals(z["eph"], z["obs"], z["ion"], float(z["doy"]), z["xyzt"] + d, z["velt"])["res_pos"]
        dn = orc.gnss_residuals(z["eph"], z["obs"], z["ion"], float(z["doy"]), z["xyzt"] - d, z["velt"])["res_pos"]
        u = base["usable"] == 1
        # beside the geometry only the troposphere moves noticeably: ~3e-4 m per metre of receiver HEIGHT at low elevation
        assert np.abs((up - dn)[u] / (2 * h) + base["los"][u, k]).max() < 1e-3
    d = np.zeros(7); d[3] = 1.0                                  # GPS clock bias: rows of GPS satellites move 1:1, the others not at all
    up = orc.gnss_residuals(z["eph"], z["obs"], z["ion"], float(z["doy"]), z["xyzt"] + d, z["velt"])["res_pos"]
    sysv = z["eph"][:, 0].astype(int)
    assert np.allclose((up - base["res_pos"])[(sysv == 0) & (base["usable"] == 1)], 1.0, atol=1e-6) and np.allclose((up - base["res_pos"])[sysv != 0], 0.0, atol=1e-9)


def test_atmosphere_models_known_values(orc):
    import ctypes as C
    L = orc.lib()
    L.orc_gnss_trop.restype = C.c_double; L.orc_gnss_iono.restype = C.c_double
    lla = (C.c_double * 3)(31.0, 121.4, 0.0)
    zen = (C.c_double * 2)(0.0, math.pi / 2)
    low = (C.c_double * 2)(1.0, math.radians(10.0))
    tz = L.orc_gnss_trop(C.c_double(180.0), lla, zen)
    assert 2.3 < tz < 2.6                                          # Saastamoinen zenith delay at sea level, standard atmosphere, 70 % humidity
    assert 5.0 < L.orc_gnss_trop(C.c_double(180.0), lla, low) / tz < 6.0      # Niell mapping at 10 degrees ~ 5.6
    ion = (C.c_double * 8)(0.1118e-07, 0.2235e-07, -0.1192e-06, -0.1192e-06, 0.1167e+06, 0.1802e+06, -0.1311e+06, -0.4588e+06)
    night = L.orc_gnss_iono(C.c_double(3600.0 * 19.0), ion, lla, zen)       # 19 h GPS time at 121 E = 03 h local: the 5 ns floor
    assert abs(night - 5e-9 * C_LIGHT * (1.0 + 16.0 * 0.03 ** 3)) < 1e-9
    day = L.orc_gnss_iono(C.c_double(3600.0 * 6.0), ion, lla, zen)          # 14 h local: the cosine bump
    assert day > 1.5 * night
    assert L.orc_gnss_trop(C.c_double(180.0), lla, (C.c_double * 2)(0.0, -0.1)) == 0.0 and L.orc_gnss_iono(C.c_double(0.0), ion, lla, (C.c_double * 2)(0.0, 0.0)) == 0.0


# ---------------------------------------------------------------------------------------------------------------------------
def _epochs(z, flts, nb, rng):
    """one raw epoch per filter: the fixture's constellation seen from a slightly different receiver state each"""
    from ingvio_amd import synth
    lat, lon = np.deg2rad(31.0), np.deg2rad(121.4)
    Renu = np.array([[-np.sin(lon), -np.sin(lat) * np.cos(lon), np.cos(lat) * np.cos(lon)],
                     [np.cos(lon), -np.sin(lat) * np.sin(lon), np.cos(lat) * np.sin(lon)], [0.0, np.cos(lat), np.sin(lat)]])
    out = []
    for b in range(nb):
        flt = flts[b]
        yaw = 0.3 + 0.01 * b
        Rz = np.array([[np.cos(yaw), -np.sin(yaw), 0], [np.sin(yaw), np.cos(yaw), 0], [0, 0, 1.0]])
        # anchor such that the filter's position maps onto the fixture's evaluation point (+ a few metres per filter)
        anchor = z["xyzt"][:3] + rng.normal(0, 2.0, 3) - Renu @ Rz @ flt.p
        out.append(dict(eph=z["eph"], obs=z["obs"], ion=z["ion"], doy=float(z["doy"]), p_w=flt.p, v_w=flt.v, cb=z["xyzt"][3:] + rng.normal(0, 1.0, 4),
                        fs=float(z["velt"][3]), yaw_offset=yaw, R_enu2ecef=Renu, anchor_ecef=anchor, idx_se23=0, idx_yof=flt.idx_yof,
                        idx_fs=flt.gnss_idx[4], idx_cb=flt.gnss_idx[:4], psr_amp=1.0, dopp_amp=1.0))
    return out


@pytest.mark.gpu
def test_gnss_front_kernel_vs_oracle_and_through_the_update(orc):
    """k_gnss_front per satellite against the oracle, then the whole epoch (front -> per-row gates -> update) against the same
    update fed with rows the host assembled from the ORACLE's residuals."""
    from ingvio_amd import capi, host, synth
    z = _z()
    nb = 5
    ctx = capi.Context(batch=nb, n_max=256, c_max=11, f_max=8, m_max=64)
    flts = []
    for b in range(nb):
        flt = synth.Filter(lambda P, b=b: capi.DeviceCov(ctx, b, P), host.imu_transition, t0=0.3 * b, n_gnss=6, n_landmarks=2)
        rngb = np.random.default_rng(b)
        for _ in range(3):
            flt.propagate_cov(flt.imu_steps(rngb)); flt.clone()
        flts.append(flt)
    rng = np.random.default_rng(77)
    eps = _epochs(z, flts, nb, rng)
    table = synth.chi2_table()
    ctx.snapshot()
    ctx.gnss_front_stage(0, eps, table, gate_rows=True)
    front = ctx.gnss_front_fetch()
    ctx.gnss_run()
    dx1, used1, keep1, gam1, st1 = ctx.gnss_fetch()
    P1 = [ctx.cov_get(b) for b in range(nb)]
    blocks = []
    for b, e in enumerate(eps):
        yaw = e["yaw_offset"]
        Rz = np.array([[np.cos(yaw), -np.sin(yaw), 0], [np.sin(yaw), np.cos(yaw), 0], [0, 0, 1.0]])
        Rw = e["R_enu2ecef"] @ Rz
        xyzt = np.r_[Rw @ e["p_w"] + e["anchor_ecef"], e["cb"]]
        velt = np.r_[Rw @ e["v_w"], e["fs"]]
        o = orc.gnss_residuals(e["eph"], e["obs"], e["ion"], e["doy"], xyzt, velt)
        ns = len(e["eph"])
        f = front[b, :ns]
        assert np.array_equal(f[:, 9].astype(int), o["usable"])
        assert np.abs(f[:, 0] - o["res_pos"]).max() < 1e-6 and np.abs(f[:, 1] - o["res_vel"]).max() < 1e-9
        assert np.abs(f[:, 2:5] - o["los"]).max() < 1e-12 and np.abs(f[:, 5:7] - o["azel"]).max() < 1e-11 and np.abs(f[:, 7:9] - o["atmos"]).max() < 1e-9
        u = o["usable"] == 1
        g = dict(los=o["los"][u], sys=e["eph"][u, 0].astype(int), res_pos=o["res_pos"][u], res_vel=o["res_vel"][u], sin_el=np.sin(o["azel"][u, 1]),
                 ura=e["eph"][u, 24], psr_std=e["obs"][u, 3], dopp_std_mps=e["obs"][u, 4] * C_LIGHT / e["obs"][u, 5], R_w2ecef=Rw, p_w=e["p_w"],
                 v_w=e["v_w"], idx_se23=0, idx_yof=e["idx_yof"], idx_cb=e["idx_cb"], idx_fs=e["idx_fs"])
        blocks.append(host.gnss_rows(g))
    ctx.restore()
    dx2, used2, keep2, gam2, st2 = ctx.gnss_update_batch(0, blocks, table, gate_rows=True)
    assert np.array_equal(used1, used2) and np.array_equal(keep1, keep2) and (used1 > 10).all() and (st1 == 0).all()
    # the two GLONASS satellites (Runge-Kutta orbit on the device) are part of the epoch: their pseudo-range and Doppler rows pass the
    # per-row gates like everybody else's (candidate rows: the usable satellites in order, pseudo-ranges first)
    sysu = z["eph"][z["usable"] == 1, 0].astype(int)
    glo_rows = np.flatnonzero(sysu == 1)
    assert len(glo_rows) == 2
    assert keep1[:, glo_rows].sum() >= nb and keep1[:, len(sysu) + glo_rows].sum() >= nb      # (a gate may refuse a row; most pass)
    for b in range(nb):
        assert rel_err(P1[b], ctx.cov_get(b)) < 1e-11 and rel_err(dx1[b], dx2[b]) < 1e-7
    ctx.close()
