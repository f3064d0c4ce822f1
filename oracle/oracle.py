"""ctypes front-end of the CPU oracle (oracle/ingvio_oracle.c).

TEST INFRASTRUCTURE ONLY — see oracle/ingvio_oracle.h.  Importable from tests/, from
``__graft_entry__.smoke()`` and from ``bench.py``'s cpu_baseline leg; never from ``ingvio_amd``.
All matrices are numpy float64, column-major (``order='F'``) where they cross the C API.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "liboracle.so")

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int)
c_up = C.POINTER(C.c_ulonglong)


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("ingvio_oracle.c", "gnss_front_oracle.c", "ingvio_oracle.h")]
    if not force and os.path.exists(_LIB) and os.path.getmtime(_LIB) >= max(os.path.getmtime(f) for f in srcs):
        return _LIB
    subprocess.check_call(["make", "-C", _HERE, "-s"], stdout=subprocess.DEVNULL)
    return _LIB


class MsckfIn(C.Structure):
    _fields_ = [
        ("n_clones", C.c_int), ("clone_idx", c_ip), ("clone_R", c_dp), ("clone_p", c_dp),
        ("n_feat", C.c_int), ("pf", c_dp), ("anchor", c_ip), ("obs_mask", c_up), ("uv", c_dp),
        ("dof", c_ip), ("stereo", C.c_int), ("R_cl2cr", C.c_double * 9), ("t_cl2cr", C.c_double * 3),
        ("noise", C.c_double), ("chi2_table", c_dp), ("chi2_len", C.c_int),
        ("max_accept", C.c_int), ("compress_rule", C.c_int), ("selected_variant", C.c_int),
    ]


class GnssIn(C.Structure):
    _fields_ = [
        ("nsat", C.c_int), ("los", c_dp), ("sys", c_ip), ("res_pos", c_dp), ("res_vel", c_dp),
        ("sin_el", c_dp), ("ura", c_dp), ("psr_std", c_dp), ("dopp_std_mps", c_dp),
        ("R_w2ecef", C.c_double * 9), ("p_w", C.c_double * 3), ("v_w", C.c_double * 3),
        ("idx_se23", C.c_int), ("idx_yof", C.c_int), ("idx_fs", C.c_int), ("idx_cb", C.c_int * 4),
        ("psr_amp", C.c_double), ("dopp_amp", C.c_double), ("chi2_test", C.c_int),
        ("chi2_table", c_dp), ("chi2_len", C.c_int),
    ]


class FrameIn(C.Structure):
    _fields_ = [
        ("k", C.c_int), ("Phi", c_dp), ("G", c_dp), ("dt", c_dp), ("sigma", C.c_double * 4),
        ("enable_gnss", C.c_int), ("gnss_idx", C.c_int * 5), ("sigma_cb", C.c_double),
        ("sigma_rw", C.c_double), ("R_i2w", C.c_double * 9), ("marg_idx", C.c_int),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB)
        _lib.orc_whiten_residual.restype = C.c_double
        _lib.orc_ekf_update.restype = C.c_int
        _lib.orc_msckf_update.restype = C.c_int
        _lib.orc_msckf_feature_block.restype = C.c_int
        _lib.orc_gnss_rows.restype = C.c_int
        _lib.orc_frame_update.restype = C.c_int
    return _lib


def _d(a):
    return a.ctypes.data_as(c_dp)


def _i(a):
    return a.ctypes.data_as(c_ip)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


# ---- small functions ---------------------------------------------------------------------
def gamma(v, m=0):
    out = np.zeros(9)
    lib().orc_gamma(_d(f64(v)), C.c_int(m), _d(out))
    return out.reshape(3, 3)


def psi1(w, a, dt):
    out = np.zeros(9)
    lib().orc_psi1(_d(f64(w)), _d(f64(a)), C.c_double(dt), _d(out))
    return out.reshape(3, 3)


def psi2(w, a, dt):
    out = np.zeros(9)
    lib().orc_psi2(_d(f64(w)), _d(f64(a)), C.c_double(dt), _d(out))
    return out.reshape(3, 3)


def se3_update(R, p, dx):
    R = f64(R).copy(); p = f64(p).copy()
    lib().orc_se3_update(_d(R), _d(p), _d(f64(dx)))
    return R, p


def se23_update(R, p, v, dx):
    R = f64(R).copy(); p = f64(p).copy(); v = f64(v).copy()
    lib().orc_se23_update(_d(R), _d(p), _d(v), _d(f64(dx)))
    return R, p, v


def imu_transition(R, p, v, bg, ba, gyro, acc, gravity, dt):
    """Returns (R', p', v', Phi(15x15), G(15x12))."""
    R = f64(R).copy(); p = f64(p).copy(); v = f64(v).copy()
    Phi = np.zeros(225); G = np.zeros(180)
    lib().orc_imu_transition(_d(R), _d(p), _d(v), _d(f64(bg)), _d(f64(ba)), _d(f64(gyro)),
                             _d(f64(acc)), _d(f64(gravity)), C.c_double(dt), _d(Phi), _d(G))
    return R, p, v, Phi.reshape(15, 15, order="F"), G.reshape(15, 12, order="F")


# ---- covariance algebra on a padded column-major buffer -------------------------------------
class Cov:
    """n x n covariance inside an ld x ld column-major buffer (mirrors State::_cov)."""

    def __init__(self, P, ld=None):
        P = np.asarray(P, dtype=np.float64)
        self.n = P.shape[0]
        self.ld = ld or max(self.n + 64, 64)
        self.buf = np.zeros((self.ld, self.ld), order="F")
        self.buf[:self.n, :self.n] = P

    @property
    def P(self):
        return self.buf[:self.n, :self.n].copy()

    def propagate(self, Phi, G, dt, sigma, enable_gnss=0, gnss_idx=(-1,) * 5, sigma_cb=0.0, sigma_rw=0.0):
        Phi = np.asfortranarray(Phi, dtype=np.float64); G = np.asfortranarray(G, dtype=np.float64)
        lib().orc_propagate_cov(_d(self.buf), C.c_int(self.n), C.c_int(self.ld), _d(Phi), _d(G),
                                C.c_double(dt), _d(f64(sigma)), C.c_int(enable_gnss), _i(i32(gnss_idx)),
                                C.c_double(sigma_cb), C.c_double(sigma_rw))

    def augment(self, R_i2w):
        assert self.n + 6 <= self.ld
        lib().orc_augment_clone(_d(self.buf), C.c_int(self.n), C.c_int(self.ld), _d(f64(R_i2w)))
        self.n += 6
        return self.n - 6

    def marginalize(self, idx, size):
        lib().orc_marginalize(_d(self.buf), C.c_int(self.n), C.c_int(self.ld), C.c_int(idx), C.c_int(size))
        self.n -= size

    def append_independent(self, blk):
        blk = np.asfortranarray(np.atleast_2d(blk), dtype=np.float64)
        s = blk.shape[0]
        assert self.n + s <= self.ld
        lib().orc_append_independent(_d(self.buf), C.c_int(self.n), C.c_int(self.ld), C.c_int(s), _d(blk))
        self.n += s
        return self.n - s

    def marginal(self, vidx, vsize):
        ns = int(np.sum(vsize))
        out = np.zeros((ns, ns), order="F")
        lib().orc_marginal_cov(_d(self.buf), C.c_int(self.ld), _i(i32(vidx)), _i(i32(vsize)),
                               C.c_int(len(vidx)), _d(out))
        return out

    @staticmethod
    def _R(R, m):
        R = np.asarray(R, dtype=np.float64)
        if R.ndim == 0 or R.size == 1 and m != 1:
            return f64(R.reshape(1)), 0
        if R.ndim == 1:
            return f64(R), 1
        return np.asfortranarray(R), 2

    def whiten(self, vidx, vsize, H, res, R):
        H = np.asfortranarray(np.atleast_2d(H), dtype=np.float64)
        m = H.shape[0]
        Rb, kind = self._R(R, m)
        return lib().orc_whiten_residual(_d(self.buf), C.c_int(self.ld), _i(i32(vidx)), _i(i32(vsize)),
                                         C.c_int(len(vidx)), _d(H), C.c_int(m), C.c_int(m), _d(f64(res)),
                                         _d(Rb), C.c_int(kind))

    def ekf_update(self, vidx, vsize, H, res, R):
        H = np.asfortranarray(np.atleast_2d(H), dtype=np.float64)
        m = H.shape[0]
        Rb, kind = self._R(R, m)
        dx = np.zeros(self.n)
        rc = lib().orc_ekf_update(_d(self.buf), C.c_int(self.n), C.c_int(self.ld), _i(i32(vidx)),
                                  _i(i32(vsize)), C.c_int(len(vidx)), _d(H), C.c_int(m), C.c_int(m),
                                  _d(f64(res)), _d(Rb), C.c_int(kind), _d(dx))
        return dx, rc

    # ---- SURVEY 8(f) f-2 -------------------------------------------------------------------------
    def add_variable_delayed_invertible(self, vidx, vsize, H_old, H_new, noise):
        H_old = np.asfortranarray(np.atleast_2d(H_old), dtype=np.float64)
        H_new = np.asfortranarray(np.atleast_2d(H_new), dtype=np.float64)
        s = H_new.shape[0]
        assert self.n + s <= self.ld
        lib().orc_add_variable_delayed_invertible(_d(self.buf), C.c_int(self.n), C.c_int(self.ld), _i(i32(vidx)), _i(i32(vsize)),
                                                  C.c_int(len(vidx)), _d(H_old), C.c_int(s), _d(H_new), C.c_int(s), C.c_int(s),
                                                  C.c_double(noise))
        self.n += s
        return self.n - s

    def add_variable_delayed(self, vidx, vsize, H_old, H_new, res, noise, chi2_mult=1.0, do_chi2=True, chi2_check=None):
        """Returns (added, dx[n+s], chi2).  chi2_check defaults to quantile(chi2(m), 0.95) (scipy)."""
        H_old = np.array(np.atleast_2d(H_old), dtype=np.float64, order="F")
        H_new = np.array(np.atleast_2d(H_new), dtype=np.float64, order="F")
        res = f64(res).copy()
        m, s = H_new.shape
        if chi2_check is None:
            from scipy.stats import chi2 as _chi2
            chi2_check = float(_chi2.ppf(0.95, m))
        dx = np.zeros(self.n + s); n_io = C.c_int(self.n); chi2 = C.c_double(0.0)
        assert self.n + s <= self.ld
        added = lib().orc_add_variable_delayed(_d(self.buf), C.byref(n_io), C.c_int(self.ld), _i(i32(vidx)), _i(i32(vsize)),
                                               C.c_int(len(vidx)), _d(H_old), C.c_int(m), _d(H_new), C.c_int(m), C.c_int(m),
                                               C.c_int(s), _d(res), C.c_double(noise), C.c_double(chi2_mult),
                                               C.c_int(1 if do_chi2 else 0), C.c_double(chi2_check), _d(dx), C.byref(chi2))
        self.n = n_io.value
        return bool(added), dx, chi2.value

    def replace_var_linear(self, tidx, tsize, vidx, vsize, H):
        H = np.asfortranarray(np.atleast_2d(H), dtype=np.float64)
        lib().orc_replace_var_linear(_d(self.buf), C.c_int(self.n), C.c_int(self.ld), C.c_int(tidx), C.c_int(tsize),
                                     _i(i32(vidx)), _i(i32(vsize)), C.c_int(len(vidx)), _d(H), C.c_int(H.shape[0]))

    def msckf_update(self, frame, **kw):
        ms, keep = make_msckf_in(frame, **kw)
        F = ms.n_feat
        dx = np.zeros(self.n); acc = np.zeros(max(F, 1), dtype=np.int32); gam = np.zeros(max(F, 1))
        m = lib().orc_msckf_update(_d(self.buf), C.c_int(self.n), C.c_int(self.ld), C.byref(ms),
                                   _d(dx), _i(acc), _d(gam))
        return dx, acc[:F].copy(), gam[:F].copy(), m


def make_msckf_in(fr, max_accept=0, compress_rule=1, selected_variant=0):
    """fr: dict with clone_idx[C], clone_R[C,3,3], clone_p[C,3], pf[F,3], anchor[F], obs_mask[F],
    uv[F,C,4], dof[F], stereo, R_cl2cr[3,3], t_cl2cr[3], noise, chi2_table."""
    keep = dict(
        clone_idx=i32(fr["clone_idx"]), clone_R=f64(fr["clone_R"]), clone_p=f64(fr["clone_p"]),
        pf=f64(fr["pf"]), anchor=i32(fr["anchor"]),
        obs_mask=np.ascontiguousarray(fr["obs_mask"], dtype=np.uint64), uv=f64(fr["uv"]),
        dof=i32(fr["dof"]), chi2=f64(fr["chi2_table"]))
    ms = MsckfIn()
    ms.n_clones = len(keep["clone_idx"]); ms.clone_idx = _i(keep["clone_idx"])
    ms.clone_R = _d(keep["clone_R"]); ms.clone_p = _d(keep["clone_p"])
    ms.n_feat = keep["pf"].shape[0] if keep["pf"].size else 0
    ms.pf = _d(keep["pf"]); ms.anchor = _i(keep["anchor"])
    ms.obs_mask = keep["obs_mask"].ctypes.data_as(c_up); ms.uv = _d(keep["uv"]); ms.dof = _i(keep["dof"])
    ms.stereo = int(fr.get("stereo", 1))
    ms.R_cl2cr = (C.c_double * 9)(*np.asarray(fr["R_cl2cr"], dtype=np.float64).reshape(9))
    ms.t_cl2cr = (C.c_double * 3)(*np.asarray(fr["t_cl2cr"], dtype=np.float64).reshape(3))
    ms.noise = float(fr["noise"]); ms.chi2_table = _d(keep["chi2"]); ms.chi2_len = len(keep["chi2"])
    ms.max_accept = int(max_accept); ms.compress_rule = int(compress_rule)
    ms.selected_variant = int(selected_variant)
    ms._keep = keep
    return ms, keep


def feature_block(fr, j, selected_variant=0):
    ms, keep = make_msckf_in(fr, selected_variant=selected_variant)
    Cn = ms.n_clones
    ldh = 4 * Cn
    Hj = np.zeros((ldh, 6 * Cn), order="F"); rj = np.zeros(ldh)
    rho = lib().orc_msckf_feature_block(C.byref(ms), C.c_int(j), _d(Hj), C.c_int(ldh), _d(rj))
    return Hj[:rho].copy(), rj[:rho].copy()


def make_frame_in(step):
    """step: dict with Phi[k,15,15], G[k,15,12], dt[k], sigma[4], enable_gnss, gnss_idx[5], sigma_cb,
    sigma_rw, R_i2w[3,3], marg_idx."""
    k = len(step["dt"])
    keep = dict(
        Phi=f64(np.stack([np.asarray(p).reshape(15, 15).T for p in step["Phi"]])),  # col-major each
        G=f64(np.stack([np.asarray(g).reshape(15, 12).T for g in step["G"]])),
        dt=f64(step["dt"]))
    fr = FrameIn()
    fr.k = k; fr.Phi = _d(keep["Phi"]); fr.G = _d(keep["G"]); fr.dt = _d(keep["dt"])
    fr.sigma = (C.c_double * 4)(*step["sigma"])
    fr.enable_gnss = int(step.get("enable_gnss", 0))
    fr.gnss_idx = (C.c_int * 5)(*[int(x) for x in step.get("gnss_idx", (-1,) * 5)])
    fr.sigma_cb = float(step.get("sigma_cb", 0.0)); fr.sigma_rw = float(step.get("sigma_rw", 0.0))
    fr.R_i2w = (C.c_double * 9)(*np.asarray(step["R_i2w"], dtype=np.float64).reshape(9))
    fr.marg_idx = int(step.get("marg_idx", -1))
    fr._keep = keep
    return fr, keep


def frame_update(cov, step, frame, **kw):
    fr, k1 = make_frame_in(step)
    ms, k2 = make_msckf_in(frame, **kw)
    F = ms.n_feat
    dx = np.zeros(cov.ld); acc = np.zeros(max(F, 1), dtype=np.int32); gam = np.zeros(max(F, 1))
    n = C.c_int(cov.n)
    m = lib().orc_frame_update(_d(cov.buf), C.byref(n), C.c_int(cov.ld), C.byref(fr), C.byref(ms),
                               _d(dx), _i(acc), _d(gam))
    n_upd = cov.n + 6
    cov.n = n.value
    return dx[:n_upd].copy(), acc[:F].copy(), gam[:F].copy(), m


class PreparedBatch:
    """C structs for orc_frame_update_batch built once, so a timed loop measures the C code only."""

    def __init__(self, steps, frames, **kw):
        self.B = len(steps)
        self.frs = (FrameIn * self.B)(); self.mss = (MsckfIn * self.B)()
        self.keeps = []
        self.fmax = 1
        for b in range(self.B):
            fr, k1 = make_frame_in(steps[b]); ms, k2 = make_msckf_in(frames[b], **kw)
            self.frs[b] = fr; self.mss[b] = ms; self.keeps.append((k1, k2, fr, ms))
            self.fmax = max(self.fmax, ms.n_feat)

    def run(self, P, n, ld, threads=0):
        """P [B, ld, ld] float64 C-contiguous (modified in place), n [B] int32 (modified in place)."""
        dx = np.zeros((self.B, ld)); acc = np.zeros((self.B, self.fmax), dtype=np.int32)
        lib().orc_frame_update_batch(C.c_int(self.B), C.c_int(threads), _d(P), _i(n), C.c_int(ld), self.frs, self.mss,
                                     _d(dx), _i(acc), C.c_int(self.fmax))
        return dx, acc


def frame_update_batch(P, n, ld, steps, frames, threads=0, **kw):
    """P: [B, ld, ld] (each column-major => pass array with P[b] = buf.T contiguous, i.e. shape
    [B, ld, ld] C-order holding the TRANSPOSE; symmetric so equivalent).  Returns (dx[B,ld], acc)."""
    B = len(steps)
    frs = (FrameIn * B)(); mss = (MsckfIn * B)()
    keeps = []
    fmax = 1
    for b in range(B):
        fr, k1 = make_frame_in(steps[b]); ms, k2 = make_msckf_in(frames[b], **kw)
        frs[b] = fr; mss[b] = ms; keeps.append((k1, k2, fr, ms))
        fmax = max(fmax, ms.n_feat)
    P = f64(P); n = i32(n).copy()
    dx = np.zeros((B, ld)); acc = np.zeros((B, fmax), dtype=np.int32)
    lib().orc_frame_update_batch(C.c_int(B), C.c_int(threads), _d(P), _i(n), C.c_int(ld), frs, mss,
                                 _d(dx), _i(acc), C.c_int(fmax))
    return P, n, dx, acc


class TriIn(C.Structure):
    _fields_ = [("C", C.c_int), ("clone_R", C.POINTER(C.c_double)), ("clone_p", C.POINTER(C.c_double)),
                ("mask", C.c_ulonglong), ("uv", C.POINTER(C.c_double)), ("stereo", C.c_int),
                ("R_lr", C.c_double * 9), ("t_lr", C.c_double * 3),
                ("trans_thres", C.c_double), ("huber_epsilon", C.c_double), ("conv_precision", C.c_double),
                ("init_damping", C.c_double), ("outer_loop_max_iter", C.c_int), ("inner_loop_max_iter", C.c_int),
                ("max_depth", C.c_double), ("min_depth", C.c_double)]


TRI_DEFAULTS = dict(trans_thres=0.1, huber_epsilon=0.01, conv_precision=5e-7, init_damping=1e-3,
                    outer_loop_max_iter=10, inner_loop_max_iter=10, max_depth=60.0, min_depth=0.2)   # Triangulator.h:67-75


def triangulate(clone_R, clone_p, mask, uv, stereo, R_lr=None, t_lr=None, **params):
    """One feature: clone_R [C,3,3], clone_p [C,3], mask (int), uv [C,4].  Returns (ok, pf[3])."""
    pr = dict(TRI_DEFAULTS); pr.update(params)
    R = f64(clone_R); p = f64(clone_p); z = f64(uv)
    t = TriIn()
    t.C = int(R.shape[0]); t.clone_R = _d(R); t.clone_p = _d(p); t.mask = int(mask); t.uv = _d(z); t.stereo = int(bool(stereo))
    Rl = np.eye(3) if R_lr is None else f64(R_lr); tl = np.zeros(3) if t_lr is None else f64(t_lr)
    for i in range(9):
        t.R_lr[i] = float(Rl.reshape(-1)[i])
    for i in range(3):
        t.t_lr[i] = float(tl[i])
    for k, v in pr.items():
        setattr(t, k, v)
    pf = np.zeros(3)
    ok = lib().orc_triangulate(C.byref(t), _d(pf))
    return bool(ok), pf


def landmark_rows_epose(R_i2w, p_i2w, R_cl2i, p_c2i, pf, uv, stereo, R_lr=None, t_lr=None):
    """LandmarkUpdate::calcResJacobianSingleLandmark{Mono,Stereo}: (H rows x 24, res)."""
    H = np.zeros((4, 24), order="F"); res = np.zeros(4)
    R_lr = np.eye(3) if R_lr is None else R_lr; t_lr = np.zeros(3) if t_lr is None else t_lr
    rows = lib().orc_landmark_rows_epose(_d(f64(R_i2w)), _d(f64(p_i2w)), _d(f64(R_cl2i)), _d(f64(p_c2i)), _d(f64(pf)), _d(f64(uv)),
                                         C.c_int(1 if stereo else 0), _d(f64(R_lr)), _d(f64(t_lr)), _d(H), _d(res))
    return H[:rows].copy(), res[:rows].copy()


def landmark_rows_sw(R_cm, p_cm, pf, uv, stereo, curr_is_anchor, R_lr=None, t_lr=None):
    H = np.zeros((4, 15), order="F"); res = np.zeros(4)
    R_lr = np.eye(3) if R_lr is None else R_lr; t_lr = np.zeros(3) if t_lr is None else t_lr
    rows = lib().orc_landmark_rows_sw(_d(f64(R_cm)), _d(f64(p_cm)), _d(f64(pf)), _d(f64(uv)), C.c_int(1 if stereo else 0),
                                      _d(f64(R_lr)), _d(f64(t_lr)), C.c_int(1 if curr_is_anchor else 0), _d(H), _d(res))
    return H[:rows].copy(), res[:rows].copy()


def qr_compress(A, b):
    A = np.asfortranarray(A, dtype=np.float64).copy(order="F"); b = f64(b).copy()
    m, n = A.shape
    lib().orc_qr_compress(_d(A), C.c_int(m), C.c_int(n), C.c_int(m), _d(b))
    return A, b


def gnss_rows(cov, g):
    keep = {k: f64(g[k]) for k in ("los", "res_pos", "res_vel", "sin_el", "ura", "psr_std", "dopp_std_mps")}
    keep["sys"] = i32(g["sys"]); keep["chi2"] = f64(g["chi2_table"])
    gi = GnssIn()
    ns = len(keep["sys"])
    gi.nsat = ns
    for k in ("los", "res_pos", "res_vel", "sin_el", "ura", "psr_std", "dopp_std_mps"):
        setattr(gi, k, _d(keep[k]))
    gi.sys = _i(keep["sys"])
    gi.R_w2ecef = (C.c_double * 9)(*np.asarray(g["R_w2ecef"], dtype=np.float64).reshape(9))
    gi.p_w = (C.c_double * 3)(*g["p_w"]); gi.v_w = (C.c_double * 3)(*g["v_w"])
    gi.idx_se23 = int(g["idx_se23"]); gi.idx_yof = int(g["idx_yof"]); gi.idx_fs = int(g["idx_fs"])
    gi.idx_cb = (C.c_int * 4)(*[int(x) for x in g["idx_cb"]])
    gi.psr_amp = float(g.get("psr_amp", 1.0)); gi.dopp_amp = float(g.get("dopp_amp", 1.0))
    gi.chi2_test = int(g.get("chi2_test", 0))
    gi.chi2_table = _d(keep["chi2"]); gi.chi2_len = len(keep["chi2"])
    ldh = 2 * ns
    H = np.zeros((ldh, 15), order="F"); res = np.zeros(ldh); Rd = np.zeros(ldh)
    vidx = np.zeros(8, dtype=np.int32); vsize = np.zeros(8, dtype=np.int32); nv = C.c_int(0)
    rows = lib().orc_gnss_rows(_d(cov.buf), C.c_int(cov.ld), C.byref(gi), _d(H), C.c_int(ldh), _d(res),
                               _d(Rd), _i(vidx), _i(vsize), C.byref(nv))
    k = nv.value
    ncols = int(vsize[:k].sum())
    return H[:rows, :ncols].copy(), res[:rows].copy(), Rd[:rows].copy(), vidx[:k].copy(), vsize[:k].copy()


# ---- the gnss_comm front (oracle/gnss_front_oracle.c) --------------------------------------------
EPH_N, OBS_N, SAT_N = 25, 6, 10


def gnss_residuals(eph, obs, ion, doy, rcv_xyzt, rcv_vel):
    """eph [ns, 25], obs [ns, 6] (flat records, see ingvio_oracle.h), ion [8] or None.
    Returns dict(res_pos, res_vel, los [ns,3], azel [ns,2], atmos [ns,2], sat [ns,10], usable [ns])."""
    eph = f64(eph); obs = f64(obs); ns = eph.shape[0]
    ionv = f64(ion if ion is not None else np.zeros(8))
    out = dict(res_pos=np.zeros(ns), res_vel=np.zeros(ns), los=np.zeros((ns, 3)), azel=np.zeros((ns, 2)),
               atmos=np.zeros((ns, 2)), sat=np.zeros((ns, SAT_N)), usable=np.zeros(ns, dtype=np.int32))
    lib().orc_gnss_residuals(C.c_int(ns), _d(eph), _d(obs), _d(ionv), C.c_int(0 if ion is None else 1), C.c_double(doy),
                             _d(f64(rcv_xyzt)), _d(f64(rcv_vel)), _d(out["res_pos"]), _d(out["res_vel"]), _d(out["los"]),
                             _d(out["azel"]), _d(out["atmos"]), _d(out["sat"]), _i(out["usable"]))
    return out
