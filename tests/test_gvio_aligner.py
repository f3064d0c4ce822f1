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


# ---- numpy transcription (checker): oracle/gvio_align.py (shared with the golden stream that runs the aligner inside the filter) ----
from oracle.gvio_align import batch_align, ecef2geo, geo2rotation, psr_pos, rotz      # noqa: E402,F401


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
