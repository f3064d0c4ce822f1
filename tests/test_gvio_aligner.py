"""SURVEY.md 8(f) row f-3, second half: GvioAligner::batchAlign (ingvio_estimator/src/GvioAligner.cpp:88-383) — coarse SPP
anchor, Gauss-Newton yaw alignment on the Doppler residuals, anchor refinement — with gnss_comm::psr_pos underneath
(gnss_comm/src/gnss_spp.cpp:148-254).  The reference holds no test for it.  Two independent restatements are compared:
  * the product: host/GvioAligner.cpp (normal equations on the host) over ingvio_gnss_sat_eval (satellite geodesy of all
    buffered epochs in one launch of k_gnss_front), driven through libingvio_host.so's facade;
  * the checker below: a numpy transcription of the same lines over the C ORACLE's residuals (oracle/gnss_front_oracle.c);
and both must recover the yaw offset and the ENU anchor a synthetic receiver track was generated with."""
import math

import numpy as np
import pytest

C_LIGHT = 2.99792458e8


# ---- numpy transcription (checker) -----------------------------------------------------------------------------------------
def ecef2geo(x):
    from oracle import gen_gnss_golden as gg
    return gg.ecef2geo(np.asarray(x, dtype=float))


def geo2rotation(lla):                                       # gnss_utility.cpp:745-755
    lat, lon = math.radians(lla[0]), math.radians(lla[1])
    sl, cl, so, co = math.sin(lat), math.cos(lat), math.sin(lon), math.cos(lon)
    return np.array([[-so, -sl * co, cl * co], [co, -sl * so, cl * so], [0.0, cl, sl]])


def rotz(a):
    return np.array([[math.cos(a), -math.sin(a), 0.0], [math.sin(a), math.cos(a), 0.0], [0.0, 0.0, 1.0]])


def psr_pos(orc, epochs, ion):
    """gnss_spp.cpp:148-254 on epochs that share one receiver state"""
    xyzt = np.zeros(7)
    sys_mask = np.zeros(4, dtype=int)
    n_valid = 0
    for e in epochs:
        ok = e["obs"][:, 5] >= 0
        n_valid += int(ok.sum())
        for s in e["eph"][ok, 0].astype(int):
            sys_mask[s] = 1
    if n_valid < 4:
        return None
    dx_norm, it = 1.0, 0
    while it < 30 and dx_norm > 1e-8:
        N = np.zeros((7, 7)); g = np.zeros(7)
        for e in epochs:
            o = orc.gnss_residuals(e["eph"], e["obs"], ion, e["doy"], xyzt, np.zeros(4))
            for i in range(len(e["obs"])):
                if not o["usable"][i] or not (o["azel"][i, 1] > math.radians(15.0)):
                    continue
                sys = int(e["eph"][i, 0])
                w = math.sin(o["azel"][i, 1]) ** 2
                if e["obs"][i, 3] > 0:
                    w /= e["obs"][i, 3] / 0.16
                w /= (e["eph"][i, 24] - 1) if sys in (0, 3) else ((e["eph"][i, 24] - 2) if sys == 2 else 4)
                G = np.r_[-o["los"][i], np.zeros(4)]; G[3 + sys] = 1.0
                N += w * np.outer(G, G); g += w * G * o["res_pos"][i]
        for k in range(4):
            if not sys_mask[k]:
                N[3 + k, 3 + k] += 1000.0
        dx = -np.linalg.solve(N, g)
        xyzt += dx; dx_norm = np.linalg.norm(dx); it += 1
    return None if it == 30 else xyzt


def batch_align(orc, epochs, p_w, v_w, ion, max_iter=10, eps=1e-5):
    """GvioAligner.cpp:199-383 on a full buffer; returns (yaw, refined anchor xyzt, rough anchor xyzt, rcv_ddt)"""
    rough = psr_pos(orc, epochs, ion)                                         # coarseLocalization
    assert rough is not None and np.linalg.norm(rough[:3]) > 1e-6
    rough[3:][np.abs(rough[3:]) < 1.0] = 0.0
    R = geo2rotation(ecef2geo(rough[:3]))
    yaw, ddt, dn, it = 0.0, 0.0, 1.0, 0                                       # yawAlignment
    while it <= max_iter and dn > eps:
        dot = np.array([[-math.sin(yaw), -math.cos(yaw), 0.0], [math.cos(yaw), -math.sin(yaw), 0.0], [0.0, 0.0, 0.0]])
        A, b = [], []
        for e, v in zip(epochs, v_w):
            o = orc.gnss_residuals(e["eph"], e["obs"], None, e["doy"], np.r_[rough[:3], np.zeros(4)], np.r_[R @ rotz(yaw) @ v, ddt])
            dv = R @ dot @ v
            for i in range(len(e["obs"])):
                u = bool(o["usable"][i])
                A.append([-(o["los"][i] @ dv) if u else 0.0, 1.0]); b.append(o["res_vel"][i] if u else 0.0)
        A, b = np.array(A), np.array(b)
        d = -np.linalg.solve(A.T @ A, A.T @ b)
        yaw += d[0]; ddt += d[1]; dn = np.linalg.norm(d); it += 1
    assert it <= max_iter
    if yaw > math.pi:
        yaw -= math.floor(yaw / (2 * math.pi) + 0.5) * 2 * math.pi
    elif yaw < -math.pi:
        yaw -= math.ceil(yaw / (2 * math.pi) - 0.5) * 2 * math.pi
    spp = [psr_pos(orc, [e], ion) for e in epochs]                            # anchorRefinement
    assert all(s is not None for s in spp)
    refined = rough.copy()
    it = 0
    while it <= max_iter:
        Rw = geo2rotation(ecef2geo(refined[:3])) @ rotz(yaw)
        anchor = np.mean([s[:3] - Rw @ p for s, p in zip(spp, p_w)], axis=0)
        dx = anchor - refined[:3]
        refined[:3] = anchor
        if np.linalg.norm(dx) > eps:                                          # as written (:367-368)
            break
        it += 1
    refined[3:] = spp[-1][3:]
    return yaw, refined, rough, ddt


# ---- a synthetic receiver track ----------------------------------------------------------------------------------------------
def make_track(n_epochs=26, yaw_true=0.6, noise=True, seed=4):
    from oracle import gen_gnss_golden as gg
    rng = np.random.default_rng(seed)
    doy = 270.4
    anchor = gg.geo2ecef(np.array([31.0, 121.4, 30.0]))
    R = geo2rotation(ecef2geo(anchor))
    ion = np.array([0.1118e-07, 0.2235e-07, -0.1192e-06, -0.1192e-06, 0.1167e+06, 0.1802e+06, -0.1311e+06, -0.4588e+06])
    t0 = 360300.0
    eph = np.vstack([gg.make_constellation(rng, anchor, t0 + 13.0), gg.make_glonass(rng, anchor, t0 + 13.0)])
    eph[:, 24] = 3.5                                                           # psr_pos divides the weights by ura - 1 (GPS, BDS) / ura - 2 (GAL)
    cb = np.array([150.0, 140.0, 165.0, 180.0]); fs = 5.0
    epochs, p_w, v_w = [], [], []
    for k in range(n_epochs):
        t = float(k)                                                           # 1 Hz GNSS, a circle of radius 20 m at 3 m/s in the VIO world frame
        w = 3.0 / 20.0
        p = np.array([20.0 * math.cos(w * t), 20.0 * math.sin(w * t), 0.3 * math.sin(0.2 * t)])
        v = np.array([-3.0 * math.sin(w * t), 3.0 * math.cos(w * t), 0.06 * math.cos(0.2 * t)])
        rcv = anchor + R @ rotz(yaw_true) @ p
        vel = R @ rotz(yaw_true) @ v
        obs = gg.make_obs(rng, eph, rcv, vel, cb + fs * t, fs, ion, doy, t0 + t, noise=noise)
        epochs.append(dict(eph=eph, obs=obs, doy=doy))
        p_w.append(p); v_w.append(v)
    return dict(epochs=epochs, p_w=np.array(p_w), v_w=np.array(v_w), ion=ion, anchor=anchor, yaw=yaw_true, fs=fs)


def test_checker_recovers_yaw_and_anchor(orc):
    """the numpy transcription over the oracle's residuals, noise-free observations: anchor to a centimetre, yaw to 1e-3 rad (the
    method evaluates every epoch's Doppler residuals at the ROUGH anchor, GvioAligner.cpp:271-273: the 10 - 20 m between it and
    the true receiver positions tilt the lines of sight by 1e-6 rad, 3 mm/s on a 3 km/s satellite, 3e-4 rad at 3 m/s)"""
    tr = make_track(noise=False)
    yaw, refined, rough, ddt = batch_align(orc, tr["epochs"][:25], tr["p_w"][:25], tr["v_w"][:25], tr["ion"])
    assert abs(yaw - tr["yaw"]) < 1e-3 and abs(ddt - tr["fs"]) < 1e-3
    # the rough anchor is the SPP of all epochs pooled: somewhere inside the 20 m circle; the refinement maps every epoch back
    assert np.linalg.norm(rough[:3] - tr["anchor"]) < 25.0
    assert np.linalg.norm(refined[:3] - tr["anchor"]) < 0.05


@pytest.mark.gpu
@pytest.mark.parametrize("noise", [False, True])
def test_shim_aligner_on_device_matches_checker_and_truth(orc, noise):
    from ingvio_amd import capi, host
    tr = make_track(noise=noise)
    ctx = capi.Context(batch=1, n_max=32, c_max=2, f_max=4, m_max=16)
    # 25 epochs fill the buffer, the 26th call runs the alignment (GvioAligner.cpp:94-99)
    got = host.aligner_run(ctx, tr["epochs"], tr["p_w"], tr["v_w"], iono=tr["ion"], batch_size=25, max_iter=10, conv_epsilon=1e-5, vel_thres=0.4)
    assert got["aligned"]
    yaw, refined, rough, ddt = batch_align(orc, tr["epochs"][:25], tr["p_w"][:25], tr["v_w"][:25], tr["ion"])
    assert abs(got["yaw_offset"] - yaw) < 1e-9 and abs(got["rcv_ddt"] - ddt) < 1e-7
    assert np.linalg.norm(got["rough_anchor"][:3] - rough[:3]) < 1e-4 and np.linalg.norm(got["anchor_ecef"] - refined[:3]) < 1e-4
    assert np.abs(got["R_enu2ecef"] - geo2rotation(ecef2geo(refined[:3]))).max() < 1e-12
    # against the generating values
    assert abs(got["yaw_offset"] - tr["yaw"]) < (2e-2 if noise else 1e-3)
    assert np.linalg.norm(got["anchor_ecef"] - tr["anchor"]) < (3.0 if noise else 0.05)
    # the per-epoch SPP fix GnssProcessor buffers (psr_pos + dopp_vel, GnssProcessor.cpp:196-217) on the last epoch
    R = geo2rotation(ecef2geo(tr["anchor"]))
    k = len(tr["epochs"]) - 1
    fix = host.spp(ctx, tr["epochs"][k], iono=tr["ion"])
    assert fix is not None
    rcv_true = tr["anchor"] + R @ rotz(tr["yaw"]) @ tr["p_w"][k]
    vel_true = R @ rotz(tr["yaw"]) @ tr["v_w"][k]
    assert np.linalg.norm(fix[0][:3] - rcv_true) < (8.0 if noise else 1e-3)
    assert np.linalg.norm(fix[1][:3] - vel_true) < (0.3 if noise else 1e-4) and abs(fix[1][3] - tr["fs"]) < (0.3 if noise else 1e-4)
    # not enough horizontal excitation: the buffer is dropped, nothing is aligned (:101-124)
    slow = host.aligner_run(ctx, tr["epochs"], tr["p_w"], 0.05 * tr["v_w"], iono=tr["ion"])
    assert not slow["aligned"]
    ctx.close()


@pytest.mark.gpu
def test_sat_eval_matches_oracle(orc):
    """ingvio_gnss_sat_eval (free-standing epochs, receiver given in ECEF) against the oracle, including the SatState block"""
    from ingvio_amd import capi
    from conftest import load_golden
    z = load_golden("gnss_front")
    ctx = capi.Context(batch=1, n_max=32, c_max=2, f_max=4, m_max=16)
    eps = []
    for k in range(3):
        xyzt = z["xyzt"] + np.r_[10.0 * k, -5.0 * k, 2.0 * k, 0, 0, 0, 0]
        eps.append(dict(eph=z["eph"], obs=z["obs"], ion=z["ion"] if k != 1 else None, doy=float(z["doy"]), p_w=np.zeros(3), v_w=z["velt"][:3],
                        cb=xyzt[3:], fs=float(z["velt"][3]), yaw_offset=0.0, R_enu2ecef=np.eye(3), anchor_ecef=xyzt[:3]))
    rec = ctx.gnss_sat_eval(eps)
    for k, e in enumerate(eps):
        o = orc.gnss_residuals(e["eph"], e["obs"], e["ion"], e["doy"], np.r_[e["anchor_ecef"], e["cb"]], np.r_[e["v_w"], e["fs"]])
        ns = len(e["obs"]); f = rec[k, :ns]
        assert np.array_equal(f[:, 9].astype(int), o["usable"])
        assert np.abs(f[:, 0] - o["res_pos"]).max() < 1e-6 and np.abs(f[:, 1] - o["res_vel"]).max() < 1e-9
        assert np.abs(f[:, 10:13] - o["sat"][:, :3]).max() < 1e-6 and np.abs(f[:, 13:16] - o["sat"][:, 3:6]).max() < 1e-9
        assert np.abs(f[:, 16:19] - o["sat"][:, 6:9]).max() < 1e-15 and np.abs(f[:, 19] - o["sat"][:, 9]).max() < 1e-9
    ctx.close()
