"""ctypes binding of libingvio_host.so — the C++14 host shim (ingvio_amd/csrc/host/) that mirrors
the reference's State / StateManager / ImuPropagator / Update classes above the HIP C ABI."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libingvio_host.so")
c_dp = C.POINTER(C.c_double)
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libingvio_host.so not built; run __graft_entry__.build()")
        _lib = C.CDLL(LIB_PATH)
        _lib.ingvio_host_chi2_quantile.restype = C.c_double
    return _lib


def _d(a):
    return a.ctypes.data_as(c_dp)


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def gamma(v, m=0):
    out = np.zeros(9)
    lib().ingvio_host_gamma(_d(_f(v)), C.c_int(m), _d(out))
    return out.reshape(3, 3)


def imu_transition(R, p, v, bg, ba, gyro, acc, gravity, dt, analytic=True):
    """ImuPropagator::stateAndCovTransition (analytic branch, or the RK4 one with analytic=False).
    Returns (R', p', v', Phi[15,15], G[15,12])."""
    R = _f(R).copy(); p = _f(p).copy(); v = _f(v).copy()
    Phi = np.zeros(225); G = np.zeros(180)
    fn = lib().ingvio_host_imu_transition if analytic else lib().ingvio_host_imu_transition_rk4
    fn(_d(R), _d(p), _d(v), _d(_f(bg)), _d(_f(ba)), _d(_f(gyro)), _d(_f(acc)),
       _d(_f(gravity)), C.c_double(dt), _d(Phi), _d(G))
    return R, p, v, Phi.reshape(15, 15, order="F"), G.reshape(15, 12, order="F")


def chi2_quantile(dof, p=0.95):
    """UpdateBase's table entry: boost::math::quantile(chi_squared(dof), p) without Boost."""
    return lib().ingvio_host_chi2_quantile(C.c_int(dof), C.c_double(p))


def gnss_rows(g):
    """Candidate rows of GnssUpdate::updateTrackedSys (GnssUpdate.cpp:148-272; gates NOT applied — they run on the device,
    ingvio_gnss_update_batch).  g: dict with los[ns,3], sys[ns], res_pos, res_vel, sin_el, ura, psr_std, dopp_std_mps,
    R_w2ecef[3,3], p_w, v_w, idx_se23, idx_yof, idx_cb[4], idx_fs (psr_amp, dopp_amp optional).  With g["adjust_yof"] set (the
    reference's is_adjust_yof = 1) the yaw-offset column is filled from g["R_enu2ecef"] and g["yaw_offset"] (:164-167, :239-242).
    Returns (vidx, vsize, H[rows, ncols], res[rows], Rdiag[rows])."""
    ns = len(g["sys"])
    c_ip = C.POINTER(C.c_int)
    sysv = np.ascontiguousarray(g["sys"], dtype=np.int32); icb = np.ascontiguousarray(g["idx_cb"], dtype=np.int32)
    ldh = max(2 * ns, 1)
    H = np.zeros((ldh, 15), order="F"); res = np.zeros(ldh); Rd = np.zeros(ldh)
    vidx = np.zeros(8, dtype=np.int32); vsize = np.zeros(8, dtype=np.int32); nv = C.c_int(0)
    yof = bool(g.get("adjust_yof", False))
    fn = lib().ingvio_host_gnss_rows_yof if yof else lib().ingvio_host_gnss_rows
    extra = (_d(_f(np.asarray(g["R_enu2ecef"]).reshape(9))), C.c_double(float(g["yaw_offset"]))) if yof else ()
    rows = fn(
        C.c_int(ns), _d(_f(g["los"])), sysv.ctypes.data_as(c_ip), _d(_f(g["res_pos"])), _d(_f(g["res_vel"])), _d(_f(g["sin_el"])),
        _d(_f(g["ura"])), _d(_f(g["psr_std"])), _d(_f(g["dopp_std_mps"])), _d(_f(np.asarray(g["R_w2ecef"]).reshape(9))), *extra,
        _d(_f(g["p_w"])), _d(_f(g["v_w"])), C.c_int(int(g["idx_se23"])), C.c_int(int(g["idx_yof"])), icb.ctypes.data_as(c_ip),
        C.c_int(int(g["idx_fs"])), C.c_double(float(g.get("psr_amp", 1.0))), C.c_double(float(g.get("dopp_amp", 1.0))),
        _d(H), C.c_int(ldh), _d(res), _d(Rd), vidx.ctypes.data_as(c_ip), vsize.ctypes.data_as(c_ip), C.byref(nv))
    k = nv.value
    nc = int(vsize[:k].sum())
    return vidx[:k].copy(), vsize[:k].copy(), H[:rows, :nc].copy(), res[:rows].copy(), Rd[:rows].copy()


def aligner_run(ctx, epochs, p_w, v_w, iono=None, batch_size=25, max_iter=10, conv_epsilon=1e-5, vel_thres=0.4):
    """The shim's GvioAligner::batchAlign (host/GvioAligner.cpp; satellite geodesy on the device through ingvio_gnss_sat_eval) on
    a list of raw epochs: dicts with eph [ns, 25], obs [ns, 6], doy.  `ctx` is a capi.Context.  Returns dict(aligned, yaw_offset,
    anchor_ecef, R_enu2ecef, rcv_ddt, rough_anchor)."""
    n = len(epochs)
    smax = max(len(e["obs"]) for e in epochs)
    eph = np.zeros((n, smax, 25)); obs = np.zeros((n, smax, 6)); ns = np.zeros(n, dtype=np.int32); doy = np.zeros(n)
    for i, e in enumerate(epochs):
        k = len(e["obs"]); eph[i, :k] = e["eph"]; obs[i, :k] = e["obs"]; ns[i] = k; doy[i] = e["doy"]
    out = np.zeros(22)
    L = lib()
    ion = _f(iono) if iono is not None else None
    L.ingvio_host_aligner_run(ctx.h, C.c_int(n), C.c_int(smax), _d(eph), _d(obs), ns.ctypes.data_as(C.POINTER(C.c_int)), _d(doy),
                              _d(_f(p_w)), _d(_f(v_w)), _d(ion) if ion is not None else None, C.c_int(batch_size), C.c_int(max_iter),
                              C.c_double(conv_epsilon), C.c_double(vel_thres), _d(out))
    return dict(aligned=bool(out[0]), yaw_offset=out[1], anchor_ecef=out[2:5].copy(), R_enu2ecef=out[5:14].reshape(3, 3).copy(),
                rcv_ddt=out[14], rough_anchor=out[15:22].copy())


def spp(ctx, epoch, iono=None):
    """gnss_comm::psr_pos + dopp_vel of one raw epoch (dict eph, obs, doy) through the shim: (xyzt [7], vel_ddt [4]) or None."""
    eph, obs = _f(epoch["eph"]), _f(epoch["obs"])
    out = np.zeros(11)
    ion = _f(iono) if iono is not None else None
    ok = lib().ingvio_host_spp(ctx.h, C.c_int(len(obs)), _d(eph), _d(obs), C.c_double(float(epoch["doy"])), _d(ion) if ion is not None else None, _d(out))
    return (out[:7].copy(), out[7:].copy()) if ok else None
