"""GvioAligner::batchAlign restated in numpy over the C oracle's GNSS residuals (TEST INFRASTRUCTURE, as everything under oracle/).

ingvio_estimator/src/GvioAligner.cpp:85-383: the buffer of (VIO position, VIO velocity, raw GNSS epoch) items (:88-124), the coarse SPP
anchor over all buffered epochs (coarseLocalization :199-233 over gnss_comm::psr_pos, gnss_spp.cpp:148-254), the Gauss-Newton yaw
alignment on the Doppler residuals (:235-312) and the anchor refinement (:314-383).  The reference holds no test for it; this
transcription is the checker of tests/test_gvio_aligner.py (per call) and - since round 6 - of the golden stream that lets the filter
align ITSELF (oracle/stream_filter.py, stream `sw11_gnss_align`).  The satellite geodesy underneath is oracle/gnss_front_oracle.c."""
import math

import numpy as np

C_LIGHT = 2.99792458e8


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



class Aligner:
    """The state machine of GvioAligner::batchAlign (GvioAligner.cpp:88-197): items are buffered until `batch_size` of them are there;
    the NEXT call (its own epoch is not used) checks the horizontal velocity excitation of the buffered VIO velocities and runs the three
    stages; every failure clears the buffer and waits for a new batch."""

    def __init__(self, orc, batch_size=25, max_iter=10, conv_epsilon=1e-5, vel_thres=0.4):
        self.orc, self.batch_size, self.max_iter, self.eps, self.vel_thres = orc, batch_size, max_iter, conv_epsilon, vel_thres
        self.buf = []
        self.aligned = False
        self.yaw_offset = 0.0
        self.anchor_ecef = np.zeros(3)
        self.R_enu2ecef = np.eye(3)

    def batch_align(self, epoch, p_w, v_w, ion):
        if self.aligned:
            return
        if len(self.buf) < self.batch_size:                              # :97-101
            self.buf.append((np.array(p_w, dtype=float), np.array(v_w, dtype=float), epoch))
            return
        hor = np.mean([np.abs(v[:2]) for _, v, _ in self.buf], axis=0)    # :104-113
        if np.linalg.norm(hor) <= self.vel_thres:
            self.buf = []
            return
        try:
            yaw, refined, rough, ddt = batch_align(self.orc, [e for _, _, e in self.buf], [p for p, _, _ in self.buf], [v for _, v, _ in self.buf],
                                                   ion, self.max_iter, self.eps)
        except AssertionError:                                           # a stage failed (:139-173): buffer dropped, not aligned
            self.buf = []
            return
        self.buf = []
        self.aligned = True
        self.yaw_offset = float(yaw)
        self.anchor_ecef = refined[:3].copy()
        self.R_enu2ecef = geo2rotation(ecef2geo(refined[:3]))            # gnss_comm::ecef2rotation (:183)
