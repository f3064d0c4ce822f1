"""ctypes binding of the C ABI (include/ingvio_hip.h) — thin plumbing, no arithmetic.

There is no CPU fallback: importing works anywhere (so the symbol table can be checked without a
GPU), but creating a context raises if libingvio_hip.so is missing or no MI355X is visible.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("INGVIO_HIP_LIB") or os.path.join(_HERE, "lib", "libingvio_hip.so")     # override: build experiments only

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int)
c_up = C.POINTER(C.c_ulonglong)

OK, NO_ROWS, NEG_DIAG, REJECTED = 0, 1, 2, 3
E_ARG, E_CAPACITY, E_HIP, E_NOT_IN_STATE, E_UNSUPPORTED, E_NOT_PD = -1, -2, -3, -4, -5, -6
R_SCALAR, R_DIAG, R_FULL = 0, 1, 2

EXPORTS = [
    "ingvio_ctx_create", "ingvio_ctx_destroy", "ingvio_sync", "ingvio_ctx_stream", "ingvio_f_max", "ingvio_c_max", "ingvio_last_error", "ingvio_ldp",
    "ingvio_build_id",
    "ingvio_cov_set", "ingvio_cov_get", "ingvio_get_n", "ingvio_cov_get_marginal", "ingvio_cov_snapshot",
    "ingvio_cov_restore", "ingvio_propagate", "ingvio_propagate_fused", "ingvio_augment_clone", "ingvio_marginalize",
    "ingvio_append_independent", "ingvio_ekf_update", "ingvio_chi2_gamma", "ingvio_msckf_update", "ingvio_msckf_update_tri", "ingvio_qr_compress",
    "ingvio_frame_stage", "ingvio_frame_stage_async", "ingvio_frame_fetch_begin", "ingvio_frame_fetch_end", "ingvio_tracks_create", "ingvio_frame_stage_tracks", "ingvio_frame_run", "ingvio_set_frame_parts", "ingvio_frame_fetch", "ingvio_profile_enable", "ingvio_profile_select",
    "ingvio_profile_reset",
    "ingvio_profile_get", "ingvio_set_msckf_method", "ingvio_set_qr_method", "ingvio_landmark_stage", "ingvio_landmark_run",
    "ingvio_landmark_fetch", "ingvio_frame_run_phase", "ingvio_info_set", "ingvio_debug_read", "ingvio_triangulate",
    "ingvio_gnss_front_stage", "ingvio_gnss_front_fetch", "ingvio_gnss_update_batch", "ingvio_gnss_stage", "ingvio_gnss_run", "ingvio_gnss_fetch", "ingvio_mld", "ingvio_debug_msckf_info", "ingvio_debug_info_solution",
    "ingvio_info_reduce", "ingvio_info_commit", "ingvio_gnss_sat_eval",
    "ingvio_chi2_gamma_multi", "ingvio_ekf_update_batch", "ingvio_add_variable_delayed_invertible", "ingvio_add_variable_delayed", "ingvio_replace_var_linear",
]


class GateBlock(C.Structure):
    _fields_ = [("vidx", C.POINTER(C.c_int)), ("vsize", C.POINTER(C.c_int)), ("k", C.c_int), ("H", C.POINTER(C.c_double)),
                ("ldh", C.c_int), ("m", C.c_int), ("res", C.POINTER(C.c_double))]


class UpdateBlock(C.Structure):
    _fields_ = [("vidx", C.POINTER(C.c_int)), ("vsize", C.POINTER(C.c_int)), ("k", C.c_int), ("H", C.POINTER(C.c_double)),
                ("ldh", C.c_int), ("m", C.c_int), ("res", C.POINTER(C.c_double)), ("R", C.POINTER(C.c_double))]


class GnssOpts(C.Structure):
    _fields_ = [("gate_rows", C.c_int), ("strong_reject", C.c_int), ("chi2_table", C.POINTER(C.c_double)), ("chi2_len", C.c_int),
                ("in_frame", C.c_int)]


class GnssEpoch(C.Structure):
    _fields_ = [("n_sat", C.c_int), ("eph", C.POINTER(C.c_double)), ("obs", C.POINTER(C.c_double)), ("ion", C.POINTER(C.c_double)),
                ("doy", C.c_double), ("p_w", C.c_double * 3), ("v_w", C.c_double * 3), ("cb", C.c_double * 4), ("fs", C.c_double),
                ("yaw_offset", C.c_double), ("R_enu2ecef", C.c_double * 9), ("anchor_ecef", C.c_double * 3), ("idx_se23", C.c_int),
                ("idx_yof", C.c_int), ("idx_fs", C.c_int), ("idx_cb", C.c_int * 4), ("psr_noise_amp", C.c_double),
                ("dopp_noise_amp", C.c_double)]


class LandmarkFrame(C.Structure):
    _fields_ = [("R_i2w", C.c_double * 9), ("p_i2w", C.c_double * 3), ("R_cl2i", C.c_double * 9), ("p_c2i", C.c_double * 3),
                ("idx_epose", C.c_int), ("idx_ext", C.c_int), ("n_lm", C.c_int), ("lm_idx", C.POINTER(C.c_int)),
                ("anchor_idx", C.POINTER(C.c_int)), ("pf", C.POINTER(C.c_double)), ("uv", C.POINTER(C.c_double)),
                ("tracked", C.POINTER(C.c_ubyte))]


class LandmarkOpts(C.Structure):
    _fields_ = [("stereo", C.c_int), ("noise", C.c_double), ("chi2_thr", C.c_double), ("R_cl2cr", C.c_double * 9),
                ("t_cl2cr", C.c_double * 3), ("in_frame", C.c_int)]


LM_MAX = 64


class CtxDesc(C.Structure):
    _fields_ = [("batch", C.c_int), ("n_max", C.c_int), ("c_max", C.c_int), ("f_max", C.c_int), ("m_max", C.c_int),
                ("device", C.c_int), ("stream", C.c_void_p)]


class MsckfFrame(C.Structure):
    _fields_ = [("n_clones", C.c_int), ("clone_idx", c_ip), ("clone_R", c_dp), ("clone_p", c_dp), ("n_feat", C.c_int),
                ("pf", c_dp), ("anchor", c_ip), ("obs_mask", c_up), ("uv", c_dp), ("dof", c_ip)]


class MsckfOpts(C.Structure):
    _fields_ = [("stereo", C.c_int), ("R_cl2cr", C.c_double * 9), ("t_cl2cr", C.c_double * 3), ("noise", C.c_double),
                ("chi2_table", c_dp), ("chi2_len", C.c_int), ("max_accept", C.c_int), ("compress_rule", C.c_int),
                ("selected_variant", C.c_int)]


class FrameStep(C.Structure):
    _fields_ = [("k", C.c_int), ("Phi", c_dp), ("G", c_dp), ("dt", c_dp), ("gnss_idx", C.c_int * 5),
                ("R_i2w", C.c_double * 9), ("marg_idx", C.c_int)]


class TrackFrame(C.Structure):
    _fields_ = [("n_drop", C.c_int), ("drop_slots", c_ip), ("n_free", C.c_int), ("free_tracks", c_ip), ("append_slot", C.c_int),
                ("n_obs", C.c_int), ("obs_track", c_ip), ("obs_uv", c_dp), ("n_pf", C.c_int), ("pf_track", c_ip), ("pf", c_dp),
                ("n_clones", C.c_int), ("clone_idx", c_ip), ("clone_R", c_dp), ("clone_p", c_dp),
                ("n_feat", C.c_int), ("feat_track", c_ip), ("feat_anchor", c_ip), ("feat_dof", c_ip), ("feat_sel", c_up)]


class FrameStepRaw(C.Structure):
    _fields_ = [("k", C.c_int), ("imu", c_dp), ("R", C.c_double * 9), ("p", C.c_double * 3), ("v", C.c_double * 3), ("bg", C.c_double * 3),
                ("ba", C.c_double * 3), ("gravity", C.c_double * 3), ("gnss_idx", C.c_int * 5), ("marg_idx", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libingvio_hip.so not built (%s); run `python -c 'import __graft_entry__ as g; g.build()'`"
                               % LIB_PATH)
        _lib = C.CDLL(LIB_PATH)
        _lib.ingvio_last_error.restype = C.c_char_p
        _lib.ingvio_ctx_stream.restype = C.c_void_p
        _lib.ingvio_build_id.restype = C.c_char_p
    return _lib


def build_id():
    """{"tu": {...}, "kernels": {...}} of the LOADED library (ingvio_build_id, written by build.py at build time)."""
    import json
    return json.loads(lib().ingvio_build_id().decode())


def _d(a):
    return a.ctypes.data_as(c_dp)


def _i(a):
    return a.ctypes.data_as(c_ip)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


class IngvioError(RuntimeError):
    def __init__(self, code, msg=""):
        super().__init__("ingvio_hip error %d %s" % (code, msg))
        self.code = code


def make_frame(fr):
    keep = dict(clone_idx=i32(fr["clone_idx"]), clone_R=f64(fr["clone_R"]), clone_p=f64(fr["clone_p"]), pf=f64(fr["pf"]),
                anchor=i32(fr["anchor"]), obs_mask=np.ascontiguousarray(fr["obs_mask"], dtype=np.uint64),
                uv=f64(fr["uv"]), dof=i32(fr["dof"]))
    f = MsckfFrame()
    f.n_clones = len(keep["clone_idx"]); f.clone_idx = _i(keep["clone_idx"]); f.clone_R = _d(keep["clone_R"])
    f.clone_p = _d(keep["clone_p"]); f.n_feat = keep["pf"].shape[0] if keep["pf"].size else 0; f.pf = _d(keep["pf"])
    f.anchor = _i(keep["anchor"]); f.obs_mask = keep["obs_mask"].ctypes.data_as(c_up); f.uv = _d(keep["uv"])
    f.dof = _i(keep["dof"])
    return f, keep


def make_opts(fr, max_accept=0, compress_rule=1, selected_variant=0):
    chi2 = f64(fr["chi2_table"])
    o = MsckfOpts()
    o.stereo = int(fr.get("stereo", 1))
    o.R_cl2cr = (C.c_double * 9)(*np.asarray(fr["R_cl2cr"], dtype=np.float64).reshape(9))
    o.t_cl2cr = (C.c_double * 3)(*np.asarray(fr["t_cl2cr"], dtype=np.float64).reshape(3))
    o.noise = float(fr["noise"]); o.chi2_table = _d(chi2); o.chi2_len = len(chi2)
    o.max_accept = int(max_accept); o.compress_rule = int(compress_rule); o.selected_variant = int(selected_variant)
    return o, chi2


def make_step(step):
    k = len(step["dt"])
    keep = dict(Phi=f64(np.stack([np.asarray(p).reshape(15, 15).T for p in step["Phi"]])),
                G=f64(np.stack([np.asarray(g).reshape(15, 12).T for g in step["G"]])), dt=f64(step["dt"]))
    s = FrameStep()
    s.k = k; s.Phi = _d(keep["Phi"]); s.G = _d(keep["G"]); s.dt = _d(keep["dt"])
    s.gnss_idx = (C.c_int * 5)(*[int(x) for x in step.get("gnss_idx", (-1,) * 5)])
    s.R_i2w = (C.c_double * 9)(*np.asarray(step["R_i2w"], dtype=np.float64).reshape(9))
    s.marg_idx = int(step.get("marg_idx", -1))
    return s, keep


def make_track_frame(d):
    """d: dict(drop=[...], free=[...], append=slot or -1, obs_track, obs_uv [n,4], pf_track, pf [n,3], clone_idx, clone_R, clone_p,
    feat_track, feat_anchor, feat_dof, feat_sel (optional))"""
    keep = dict(drop=i32(d.get("drop", [])), free=i32(d.get("free", [])), obs_track=i32(d.get("obs_track", [])), obs_uv=f64(d.get("obs_uv", np.zeros((0, 4)))),
                pf_track=i32(d.get("pf_track", [])), pf=f64(d.get("pf", np.zeros((0, 3)))), clone_idx=i32(d["clone_idx"]), clone_R=f64(d["clone_R"]),
                clone_p=f64(d["clone_p"]), feat_track=i32(d["feat_track"]), feat_anchor=i32(d["feat_anchor"]), feat_dof=i32(d["feat_dof"]))
    f = TrackFrame()
    f.n_drop = len(keep["drop"]); f.drop_slots = _i(keep["drop"]); f.n_free = len(keep["free"]); f.free_tracks = _i(keep["free"])
    f.append_slot = int(d.get("append", -1)); f.n_obs = len(keep["obs_track"]); f.obs_track = _i(keep["obs_track"]); f.obs_uv = _d(keep["obs_uv"])
    f.n_pf = len(keep["pf_track"]); f.pf_track = _i(keep["pf_track"]); f.pf = _d(keep["pf"])
    f.n_clones = len(keep["clone_idx"]); f.clone_idx = _i(keep["clone_idx"]); f.clone_R = _d(keep["clone_R"]); f.clone_p = _d(keep["clone_p"])
    f.n_feat = len(keep["feat_track"]); f.feat_track = _i(keep["feat_track"]); f.feat_anchor = _i(keep["feat_anchor"]); f.feat_dof = _i(keep["feat_dof"])
    if d.get("feat_sel") is not None:
        keep["feat_sel"] = np.ascontiguousarray(d["feat_sel"], dtype=np.uint64)
        f.feat_sel = keep["feat_sel"].ctypes.data_as(c_up)
    return f, keep


def make_step_raw(step):
    raw = step["raw"]
    keep = dict(imu=f64(raw["imu"]))
    s = FrameStepRaw()
    s.k = keep["imu"].shape[0]; s.imu = _d(keep["imu"])
    s.R = (C.c_double * 9)(*np.asarray(raw["R"], dtype=np.float64).reshape(9))
    for name in ("p", "v", "bg", "ba", "gravity"):
        setattr(s, name, (C.c_double * 3)(*np.asarray(raw[name], dtype=np.float64).reshape(3)))
    s.gnss_idx = (C.c_int * 5)(*[int(x) for x in step.get("gnss_idx", (-1,) * 5)])
    s.marg_idx = int(step.get("marg_idx", -1))
    return s, keep


class Context:
    """A batch of independent filters on one GPU (ingvio_ctx)."""

    def __init__(self, batch=1, n_max=256, c_max=11, f_max=160, m_max=64, device=0, stream=None):
        self.L = lib()
        d = CtxDesc(batch, n_max, c_max, f_max, m_max, device, stream)
        self.h = C.c_void_p()
        rc = self.L.ingvio_ctx_create(C.byref(d), C.byref(self.h))
        if rc != 0:
            msg = self.L.ingvio_last_error(self.h).decode() if self.h else "no HIP device / library"
            raise IngvioError(rc, msg)
        self.batch, self.n_max, self.c_max, self.f_max, self.device = batch, n_max, c_max, f_max, device
        self.ldp = self.L.ingvio_ldp(self.h)

    def close(self):
        if self.h:
            self.L.ingvio_ctx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, soft_ok=True):
        if rc < 0 or (rc > 0 and not soft_ok):
            raise IngvioError(rc, self.L.ingvio_last_error(self.h).decode())
        return rc

    def sync(self):
        self._chk(self.L.ingvio_sync(self.h))

    def set_method(self, method):
        """'dense' (literal TSQR path) or 'factored' (default, structure-exploiting information form)."""
        self._chk(self.L.ingvio_set_msckf_method(self.h, {"dense": 0, "factored": 1}[method]))

    def set_qr_method(self, method):
        """qr_compress: "auto" (default), "householder", "cholesky" (ingvio_set_qr_method)"""
        self._chk(self.L.ingvio_set_qr_method(self.h, {"auto": 0, "householder": 1, "cholesky": 2}[method]))

    def debug_read(self, n=32):
        out = (C.c_longlong * n)()
        self._chk(self.L.ingvio_debug_read(self.h, out, n))
        return list(out)

    def stream(self):
        return self.L.ingvio_ctx_stream(self.h)

    # ---- covariance I/O -----------------------------------------------------------------
    def cov_set(self, b, P):
        P = np.asfortranarray(P, dtype=np.float64)
        n = P.shape[0]
        self._chk(self.L.ingvio_cov_set(self.h, b, _d(P), max(n, 1), n))

    def n(self, b=0):
        v = C.c_int(0)
        self._chk(self.L.ingvio_get_n(self.h, b, C.byref(v)))
        return v.value

    def cov_get(self, b=0):
        n = self.n(b)
        P = np.zeros((n, n), order="F")
        self._chk(self.L.ingvio_cov_get(self.h, b, _d(P), max(n, 1)))
        return P

    def marginal(self, b, vidx, vsize):
        ns = int(np.sum(vsize))
        out = np.zeros((ns, ns), order="F")
        self._chk(self.L.ingvio_cov_get_marginal(self.h, b, _i(i32(vidx)), _i(i32(vsize)), len(vidx), _d(out)))
        return out

    def snapshot(self):
        self._chk(self.L.ingvio_cov_snapshot(self.h))

    def restore(self):
        self._chk(self.L.ingvio_cov_restore(self.h))

    # ---- structure + propagation (batched over [b0, b0+nb)) ------------------------------
    def propagate(self, b0, Phi, G, dt, sigma, enable_gnss=0, gnss_idx=None, sigma_cb=0.0, sigma_rw=0.0, fused=False):
        """Phi [nb,k,15,15] (row-major numpy), G [nb,k,15,12], dt [nb,k]."""
        Phi = np.asarray(Phi, dtype=np.float64); G = np.asarray(G, dtype=np.float64); dt = np.asarray(dt, dtype=np.float64)
        if Phi.ndim == 2:
            Phi = Phi[None, None]; G = G[None, None]; dt = dt.reshape(1, 1)
        elif Phi.ndim == 3:
            Phi = Phi[None]; G = G[None]; dt = dt.reshape(1, -1)
        nb, k = Phi.shape[0], Phi.shape[1]
        PhiC = f64(np.swapaxes(Phi, -1, -2)); GC = f64(np.swapaxes(G, -1, -2)); dt = f64(dt)
        gi = None if gnss_idx is None else i32(np.asarray(gnss_idx).reshape(nb, 5))
        sig = f64(sigma)
        if fused or k > 1:
            rc = self.L.ingvio_propagate_fused(self.h, b0, nb, k, _d(PhiC), _d(GC), _d(dt), _d(sig), int(enable_gnss),
                                               _i(gi) if gi is not None else None, C.c_double(sigma_cb), C.c_double(sigma_rw))
        else:
            rc = self.L.ingvio_propagate(self.h, b0, nb, _d(PhiC), _d(GC), _d(dt), _d(sig), int(enable_gnss),
                                         _i(gi) if gi is not None else None, C.c_double(sigma_cb), C.c_double(sigma_rw))
        self._chk(rc)

    def augment(self, b0, R_i2w):
        R = f64(np.asarray(R_i2w).reshape(-1, 9))
        nb = R.shape[0]
        idx = np.zeros(nb, dtype=np.int32)
        self._chk(self.L.ingvio_augment_clone(self.h, b0, nb, _d(R), _i(idx)))
        return idx

    def marginalize(self, b0, idx, size):
        idx = i32(np.atleast_1d(idx))
        self._chk(self.L.ingvio_marginalize(self.h, b0, len(idx), _i(idx), int(size)))

    def append_independent(self, b0, blk):
        blk = np.asarray(blk, dtype=np.float64)
        if blk.ndim == 2:
            blk = blk[None]
        nb, s = blk.shape[0], blk.shape[1]
        b = f64(np.swapaxes(blk, -1, -2))
        idx = np.zeros(nb, dtype=np.int32)
        self._chk(self.L.ingvio_append_independent(self.h, b0, nb, s, _d(b), _i(idx)))
        return idx

    # ---- updates ----------------------------------------------------------------------------
    @staticmethod
    def _R(R, m):
        R = np.asarray(R, dtype=np.float64)
        if R.ndim == 0 or (R.size == 1 and m != 1):
            return f64(R.reshape(1)), R_SCALAR
        if R.ndim == 1:
            return f64(R), R_DIAG
        return np.asfortranarray(R), R_FULL

    def ekf_update(self, b, vidx, vsize, H, res, R):
        H = np.asfortranarray(np.atleast_2d(H), dtype=np.float64)
        m = H.shape[0]
        Rb, kind = self._R(R, m)
        dx = np.zeros(self.n(b))
        rc = self.L.ingvio_ekf_update(self.h, b, _i(i32(vidx)), _i(i32(vsize)), len(vidx), _d(H), m, m, _d(f64(res)),
                                      _d(Rb), kind, _d(dx))
        return dx, self._chk(rc)

    def chi2_gamma(self, b, vidx, vsize, H, res, R):
        H = np.asfortranarray(np.atleast_2d(H), dtype=np.float64)
        m = H.shape[0]
        Rb, kind = self._R(R, m)
        g = C.c_double(0.0)
        self._chk(self.L.ingvio_chi2_gamma(self.h, b, _i(i32(vidx)), _i(i32(vsize)), len(vidx), _d(H), m, m, _d(f64(res)),
                                           _d(Rb), kind, C.byref(g)))
        return g.value

    def ekf_update_batch(self, b0, blocks, diag=True):
        """blocks: list of (vidx, vsize, H, res, R) for filters b0, b0+1, ...; R scalar or 1-D per block (all of one kind).
        Returns (dx[nb, ldp], status[nb])."""
        nb = len(blocks)
        arr = (UpdateBlock * nb)(); keep = []
        for g, (vidx, vsize, H, res, R) in enumerate(blocks):
            H = np.asfortranarray(np.atleast_2d(H), dtype=np.float64)
            vi, vs, r, Rv = i32(vidx), i32(vsize), f64(res), f64(np.atleast_1d(R))
            keep.append((H, vi, vs, r, Rv))
            arr[g].vidx = _i(vi); arr[g].vsize = _i(vs); arr[g].k = len(vi); arr[g].H = _d(H); arr[g].ldh = H.shape[0]
            arr[g].m = H.shape[0]; arr[g].res = _d(r); arr[g].R = _d(Rv)
        dx = np.zeros((nb, self.ldp)); st = np.zeros(nb, dtype=np.int32)
        self._chk(self.L.ingvio_ekf_update_batch(self.h, b0, nb, arr, 1 if diag else 0, _d(dx), _i(st)))
        return dx, st

    # ---- GnssUpdate::updateTrackedSys for a batch: per-row gates + compaction + block gate + ekfUpdate on the device ----
    @staticmethod
    def _update_blocks(blocks):
        nb = len(blocks)
        arr = (UpdateBlock * nb)(); keep = []
        for g, blk in enumerate(blocks):
            if blk is None:                               # no measurement for this filter
                arr[g].m = 0; arr[g].k = 0
                continue
            vidx, vsize, H, res, R = blk
            H = np.asfortranarray(np.atleast_2d(H), dtype=np.float64)
            vi, vs, r, Rv = i32(vidx), i32(vsize), f64(res), f64(np.atleast_1d(R))
            keep.append((H, vi, vs, r, Rv))
            arr[g].vidx = _i(vi); arr[g].vsize = _i(vs); arr[g].k = len(vi); arr[g].H = _d(H); arr[g].ldh = H.shape[0]
            arr[g].m = H.shape[0]; arr[g].res = _d(r); arr[g].R = _d(Rv)
        return arr, keep

    def gnss_stage(self, b0, blocks, chi2_table, gate_rows=True, strong_reject=False, in_frame=False):
        """blocks: per filter (vidx, vsize, H [m, nc] candidate rows, res [m], Rdiag [m]) or None.  in_frame: applied by frame_run
        right after the frame's MSCKF update, in the same sweep over P (no gnss_run)."""
        arr, keep = self._update_blocks(blocks)
        tab = f64(chi2_table)
        o = GnssOpts(); o.gate_rows = int(gate_rows); o.strong_reject = int(strong_reject); o.chi2_table = _d(tab); o.chi2_len = len(tab)
        o.in_frame = int(in_frame)
        self._chk(self.L.ingvio_gnss_stage(self.h, b0, len(blocks), arr, C.byref(o)))
        self._gnss_range = (b0, len(blocks))

    @staticmethod
    def _gnss_epochs(epochs):
        nb = len(epochs)
        arr = (GnssEpoch * nb)(); keep = []
        for i, e in enumerate(epochs):
            eph, obs = f64(e["eph"]), f64(e["obs"])
            ion = f64(e["ion"]) if e.get("ion") is not None else None
            keep.append((eph, obs, ion))
            a = arr[i]
            a.n_sat = eph.shape[0]; a.eph = _d(eph); a.obs = _d(obs); a.ion = _d(ion) if ion is not None else None
            a.doy = float(e["doy"]); a.p_w = (C.c_double * 3)(*e["p_w"]); a.v_w = (C.c_double * 3)(*e["v_w"])
            a.cb = (C.c_double * 4)(*e["cb"]); a.fs = float(e["fs"]); a.yaw_offset = float(e["yaw_offset"])
            a.R_enu2ecef = (C.c_double * 9)(*np.asarray(e["R_enu2ecef"], dtype=np.float64).reshape(9))
            a.anchor_ecef = (C.c_double * 3)(*e["anchor_ecef"])
            a.idx_se23 = int(e["idx_se23"]); a.idx_yof = int(e["idx_yof"]); a.idx_fs = int(e["idx_fs"])
            a.idx_cb = (C.c_int * 4)(*[int(x) for x in e["idx_cb"]])
            a.psr_noise_amp = float(e.get("psr_amp", 1.0)); a.dopp_noise_amp = float(e.get("dopp_amp", 1.0))
        return arr, keep

    def gnss_sat_eval(self, epochs):
        """ingvio_gnss_sat_eval: the front's per-satellite records [n_epochs, 64, 20] for free-standing epochs (dicts as for
        gnss_front_stage; the idx_* entries may be omitted)."""
        eps = [dict(dict(idx_se23=0, idx_yof=0, idx_fs=0, idx_cb=[-1] * 4), **e) for e in epochs]
        arr, keep = self._gnss_epochs(eps)
        out = np.zeros((len(eps), 64, 20))
        self._chk(self.L.ingvio_gnss_sat_eval(self.h, len(eps), arr, _d(out)))
        return out

    def gnss_front_stage(self, b0, epochs, chi2_table, gate_rows=True, strong_reject=False):
        """epochs: per filter a dict(eph [ns,25], obs [ns,6], ion [8] or None, doy, p_w, v_w, cb [4], fs, yaw_offset, R_enu2ecef [3,3],
        anchor_ecef, idx_se23, idx_yof, idx_fs, idx_cb [4], psr_amp, dopp_amp): raw GNSS epochs -> candidate rows on the device."""
        nb = len(epochs)
        arr, keep = self._gnss_epochs(epochs)
        tab = f64(chi2_table)
        o = GnssOpts(); o.gate_rows = int(gate_rows); o.strong_reject = int(strong_reject); o.chi2_table = _d(tab); o.chi2_len = len(tab)
        self._chk(self.L.ingvio_gnss_front_stage(self.h, b0, nb, arr, C.byref(o)))
        self._gnss_range = (b0, nb)

    def gnss_front_fetch(self, b0=None, nb=None):
        """-> [nb, 64, 20]: res_pos, res_vel, los (3), az, el, ion, tro, usable, then the SatState: pos (3), vel (3), dt, ddt, tgd, ttx"""
        b0, nb = (self._gnss_range if b0 is None else (b0, nb))
        out = np.zeros((nb, 64, 20))
        self._chk(self.L.ingvio_gnss_front_fetch(self.h, b0, nb, _d(out)))
        return out

    def frame_run_phase(self, phase, restore_prior=False):
        """1: propagate + clone + gate + Gram of the staged features; 2: solve + apply + marginalise (ingvio_frame_run_phase)"""
        self._chk(self.L.ingvio_frame_run_phase(self.h, int(restore_prior), int(phase)))

    def info_reduce(self, b):
        """[A | b | n_accepted] of filter b summed over its chunk partials ON THE DEVICE: returns (device pointer, count of
        doubles, ncol) - the buffer the feature-sharded filter all-reduces in place (ingvio_info_reduce)."""
        ptr = C.POINTER(C.c_double)(); cnt = C.c_int(0); nc = C.c_int(0)
        self._chk(self.L.ingvio_info_reduce(self.h, b, C.byref(ptr), C.byref(cnt), C.byref(nc)))
        return C.cast(ptr, C.c_void_p).value, cnt.value, nc.value

    def info_commit(self, b):
        self._chk(self.L.ingvio_info_commit(self.h, b))

    def info_set(self, b, A, n_accepted):
        A = f64(A)
        self._chk(self.L.ingvio_info_set(self.h, b, _d(A), A.shape[0], int(n_accepted)))

    def landmark_stage(self, b0, frames, stereo, noise, chi2_thr, R_cl2cr=None, t_cl2cr=None, in_frame=False):
        """frames: per filter a dict(R_i2w, p_i2w, R_cl2i, p_c2i, idx_epose, idx_ext, lm_idx [L], anchor_idx [L], pf [L,3], uv [L,4],
        tracked [L]): the in-state landmarks seen in the current frame (ingvio_landmark_stage)."""
        nb = len(frames)
        arr = (LandmarkFrame * nb)(); keep = []
        for i, f in enumerate(frames):
            li = np.ascontiguousarray(f["lm_idx"], dtype=np.int32); ai = np.ascontiguousarray(f["anchor_idx"], dtype=np.int32)
            pf = f64(f["pf"]).reshape(-1, 3); uv = f64(f["uv"]).reshape(-1, 4)
            tr = np.ascontiguousarray(f.get("tracked", np.ones(len(li))), dtype=np.uint8)
            keep.append((li, ai, pf, uv, tr))
            a = arr[i]
            a.R_i2w = (C.c_double * 9)(*f64(f["R_i2w"]).reshape(9)); a.p_i2w = (C.c_double * 3)(*f64(f["p_i2w"]).reshape(3))
            a.R_cl2i = (C.c_double * 9)(*f64(f["R_cl2i"]).reshape(9)); a.p_c2i = (C.c_double * 3)(*f64(f["p_c2i"]).reshape(3))
            a.idx_epose = int(f["idx_epose"]); a.idx_ext = int(f["idx_ext"]); a.n_lm = len(li)
            a.lm_idx = _i(li); a.anchor_idx = _i(ai); a.pf = _d(pf); a.uv = _d(uv); a.tracked = tr.ctypes.data_as(C.POINTER(C.c_ubyte))
        o = LandmarkOpts(); o.stereo = int(bool(stereo)); o.noise = float(noise); o.chi2_thr = float(chi2_thr)
        o.R_cl2cr = (C.c_double * 9)(*f64(np.eye(3) if R_cl2cr is None else R_cl2cr).reshape(9))
        o.t_cl2cr = (C.c_double * 3)(*f64(np.zeros(3) if t_cl2cr is None else t_cl2cr).reshape(3))
        o.in_frame = int(bool(in_frame))
        self._chk(self.L.ingvio_landmark_stage(self.h, b0, nb, arr, C.byref(o)))
        self._lm_range = (b0, nb)

    def landmark_run(self, b0=None, nb=None):
        b0, nb = (self._lm_range if b0 is None else (b0, nb))
        self._chk(self.L.ingvio_landmark_run(self.h, b0, nb))

    def landmark_fetch(self, b0=None, nb=None):
        """-> (dx [nb, ldp], rows [nb], accept [nb, 64], gamma [nb, 64], status [nb])"""
        b0, nb = (self._lm_range if b0 is None else (b0, nb))
        dx = np.zeros((nb, self.ldp)); rows = np.zeros(nb, dtype=np.int32); acc = np.zeros((nb, LM_MAX), dtype=np.int32)
        gam = np.zeros((nb, LM_MAX)); st = np.zeros(nb, dtype=np.int32)
        self._chk(self.L.ingvio_landmark_fetch(self.h, b0, nb, _d(dx), _i(rows), _i(acc), _d(gam), _i(st)))
        return dx, rows, acc, gam, st

    def gnss_run(self, b0=None, nb=None):
        b0, nb = (self._gnss_range if b0 is None else (b0, nb))
        self._chk(self.L.ingvio_gnss_run(self.h, b0, nb))

    def gnss_fetch(self, b0=None, nb=None):
        """-> (dx [nb, ldp], rows [nb], keep [nb, mld], gamma [nb, mld], status [nb])"""
        b0, nb = (self._gnss_range if b0 is None else (b0, nb))
        mld = self.L.ingvio_mld(self.h)
        dx = np.zeros((nb, self.ldp)); rows = np.zeros(nb, dtype=np.int32); keep = np.zeros((nb, mld), dtype=np.int32)
        gam = np.zeros((nb, mld)); st = np.zeros(nb, dtype=np.int32)
        self._chk(self.L.ingvio_gnss_fetch(self.h, b0, nb, _d(dx), _i(rows), _i(keep), _d(gam), _i(st)))
        return dx, rows, keep, gam, st

    def gnss_update_batch(self, b0, blocks, chi2_table, gate_rows=True, strong_reject=False):
        self.gnss_stage(b0, blocks, chi2_table, gate_rows, strong_reject)
        self.gnss_run()
        return self.gnss_fetch()

    def debug_msckf_info(self, b):
        """[A | b] of filter b's last MSCKF update: (A [ncol, ncol], b [ncol])."""
        cap = 6 * self.c_max
        out = np.zeros(cap * (cap + 1)); nc = C.c_int(0)
        self._chk(self.L.ingvio_debug_msckf_info(self.h, b, _d(out), C.byref(nc)))
        n = nc.value
        M = out[:n * (n + 1)].reshape(n, n + 1)
        return M[:, :n].copy(), M[:, n].copy()

    def debug_info_solution(self, b):
        """(M [nc, nc], t [nc]) of filter b's last factored update, nc = 6 * (window class of c_max)."""
        cls = 6 if self.c_max <= 6 else (11 if self.c_max <= 11 else 16)
        nc = 6 * cls; mp = (nc + 3) & ~3
        out = np.zeros(mp * mp + mp)
        self._chk(self.L.ingvio_debug_info_solution(self.h, b, _d(out), out.size))
        return out[:mp * mp].reshape(mp, mp)[:nc, :nc].copy(), out[mp * mp:mp * mp + nc].copy()

    def chi2_gamma_multi(self, b, blocks, noise_var):
        """blocks: list of (vidx, vsize, H, res); returns gamma[len(blocks)] (one launch, one sync)."""
        nb = len(blocks)
        arr = (GateBlock * nb)(); keep = []
        for g, (vidx, vsize, H, res) in enumerate(blocks):
            H = np.asfortranarray(np.atleast_2d(H), dtype=np.float64)
            vi, vs, r = i32(vidx), i32(vsize), f64(res)
            keep.append((H, vi, vs, r))
            arr[g].vidx = _i(vi); arr[g].vsize = _i(vs); arr[g].k = len(vi); arr[g].H = _d(H); arr[g].ldh = H.shape[0]
            arr[g].m = H.shape[0]; arr[g].res = _d(r)
        out = np.zeros(max(nb, 1))
        self._chk(self.L.ingvio_chi2_gamma_multi(self.h, b, nb, arr, C.c_double(noise_var), _d(out)))
        return out[:nb]

    # ---- SLAM-landmark path (f-2) --------------------------------------------------------------
    def add_variable_delayed_invertible(self, b, vidx, vsize, H_old, H_new, noise):
        H_old = np.asfortranarray(np.atleast_2d(H_old), dtype=np.float64)
        H_new = np.asfortranarray(np.atleast_2d(H_new), dtype=np.float64)
        s = H_new.shape[0]
        idx = C.c_int(-1)
        self._chk(self.L.ingvio_add_variable_delayed_invertible(self.h, b, _i(i32(vidx)), _i(i32(vsize)), len(vidx), _d(H_old), s,
                                                                _d(H_new), s, s, C.c_double(noise), C.byref(idx)))
        return idx.value

    def add_variable_delayed(self, b, vidx, vsize, H_old, H_new, res, noise, chi2_mult=1.0, do_chi2=True, chi2_check=None):
        """Returns (added, dx[n+s] or None, chi2, new_idx)."""
        H_old = np.asfortranarray(np.atleast_2d(H_old), dtype=np.float64)
        H_new = np.asfortranarray(np.atleast_2d(H_new), dtype=np.float64)
        m, s = H_new.shape
        if chi2_check is None:
            from scipy.stats import chi2 as _chi2
            chi2_check = float(_chi2.ppf(0.95, m))
        dx = np.zeros(self.n(b) + s); added = C.c_int(0); idx = C.c_int(-1); chi2 = C.c_double(0.0)
        self._chk(self.L.ingvio_add_variable_delayed(self.h, b, _i(i32(vidx)), _i(i32(vsize)), len(vidx), _d(H_old), m, _d(H_new), m,
                                                     m, s, _d(f64(res)), C.c_double(noise), C.c_double(chi2_mult),
                                                     1 if do_chi2 else 0, C.c_double(chi2_check), _d(dx), C.byref(added),
                                                     C.byref(idx), C.byref(chi2)))
        return bool(added.value), (dx if added.value else None), chi2.value, idx.value

    def replace_var_linear(self, b, tidx, tsize, vidx, vsize, H):
        H = np.asfortranarray(np.atleast_2d(H), dtype=np.float64)
        self._chk(self.L.ingvio_replace_var_linear(self.h, b, int(tidx), int(tsize), _i(i32(vidx)), _i(i32(vsize)), len(vidx),
                                                   _d(H), H.shape[0]))

    def msckf_update(self, b0, frames, max_accept=0, compress_rule=1, selected_variant=0):
        """frames: list of frame dicts (one per filter).  Returns (dx[nb,n], accepted[nb,F], gamma[nb,F], rows[nb])."""
        if isinstance(frames, dict):
            frames = [frames]
        nb = len(frames)
        arr = (MsckfFrame * nb)()
        keeps = []
        for i, fr in enumerate(frames):
            f, k = make_frame(fr)
            arr[i] = f; keeps.append(k)
        o, chi2 = make_opts(frames[0], max_accept, compress_rule, selected_variant)
        dx = np.zeros((nb, self.ldp)); acc = np.zeros((nb, self.f_max), dtype=np.int32)
        gam = np.zeros((nb, self.f_max)); rows = np.zeros(nb, dtype=np.int32)
        rc = self.L.ingvio_msckf_update(self.h, b0, nb, arr, C.byref(o), _d(dx), _i(acc), _d(gam), _i(rows))
        self._chk(rc)
        return dx, acc, gam, rows

    def msckf_update_tri(self, b0, frames, max_accept=0, compress_rule=1, selected_variant=0, stereo=True, tri_masks=None, **params):
        """ingvio_msckf_update_tri: the frames' points are triangulated on the device from their own observations (pf of the frame
        dicts is ignored), failed features drop out.  Returns (dx, accepted, gamma, rows, pf[nb, f_max, 3], tri_ok[nb, f_max])."""
        if isinstance(frames, dict):
            frames = [frames]
        nb = len(frames)
        arr = (MsckfFrame * nb)()
        keeps = []
        for i, fr in enumerate(frames):
            f, k = make_frame(fr)
            arr[i] = f; keeps.append(k)
        o, chi2 = make_opts(frames[0], max_accept, compress_rule, selected_variant)
        pr = dict(trans_thres=0.1, huber_epsilon=0.01, conv_precision=5e-7, init_damping=1e-3, outer_loop_max_iter=10,
                  inner_loop_max_iter=10, max_depth=60.0, min_depth=0.2)
        pr.update(params)
        t = TriOpts()
        t.stereo = 1 if stereo else 0
        Rl = f64(frames[0]["R_cl2cr"]).reshape(-1); tl = f64(frames[0]["t_cl2cr"])
        for i in range(9):
            t.R_cl2cr[i] = float(Rl[i])
        for i in range(3):
            t.t_cl2cr[i] = float(tl[i])
        for k, v in pr.items():
            setattr(t, k, v)
        dx = np.zeros((nb, self.ldp)); acc = np.zeros((nb, self.f_max), dtype=np.int32)
        gam = np.zeros((nb, self.f_max)); rows = np.zeros(nb, dtype=np.int32)
        pf = np.zeros((nb, self.f_max, 3)); ok = np.zeros((nb, self.f_max), dtype=np.int32)
        tm = None
        if tri_masks is not None:                              # one mask array per filter (None: the frame's obs_mask)
            tm_keep = [None if m is None else np.ascontiguousarray(m, dtype=np.uint64) for m in tri_masks]
            tm = (c_up * nb)(*[C.cast(None, c_up) if m is None else m.ctypes.data_as(c_up) for m in tm_keep])
        self._chk(self.L.ingvio_msckf_update_tri(self.h, b0, nb, arr, C.byref(o), C.byref(t), tm, _d(dx), _i(acc), _d(gam), _i(rows), _d(pf), _i(ok)))
        return dx, acc, gam, rows, pf, ok

    def qr_compress(self, H, res):
        H = np.asfortranarray(H, dtype=np.float64)
        m, n = H.shape
        Ht = np.zeros((n, n), order="F"); rt = np.zeros(n)
        self._chk(self.L.ingvio_qr_compress(self.h, _d(H), m, m, n, _d(f64(res)), _d(Ht), n, _d(rt)))
        return Ht, rt

    # ---- whole-batch frame (bench) ------------------------------------------------------------
    def frame_stage(self, b0, steps, frames, sigma, enable_gnss=0, sigma_cb=0.0, sigma_rw=0.0, max_accept=0,
                    compress_rule=1, selected_variant=0):
        nb = len(steps)
        sa = (FrameStep * nb)(); fa = (MsckfFrame * nb)()
        keeps = []
        for i in range(nb):
            s, k1 = make_step(steps[i]); f, k2 = make_frame(frames[i])
            sa[i] = s; fa[i] = f; keeps.append((k1, k2))
        o, chi2 = make_opts(frames[0], max_accept, compress_rule, selected_variant)
        self._chk(self.L.ingvio_frame_stage(self.h, b0, nb, sa, fa, C.byref(o), _d(f64(sigma)), int(enable_gnss),
                                            C.c_double(sigma_cb), C.c_double(sigma_rw)))

    def tracks_create(self, t_max):
        """allocates / clears the device-resident track store (ingvio_tracks_create)"""
        self._chk(self.L.ingvio_tracks_create(self.h, int(t_max)))

    def frame_stage_tracks_prepare(self, b0, steps, track_frames, opts_frame, sigma, enable_gnss=0, sigma_cb=0.0, sigma_rw=0.0, max_accept=0,
                                   compress_rule=1, selected_variant=0, use_async=False):
        """Builds the C argument arrays once and returns a callable that issues ingvio_frame_stage_tracks on them: the frame hand-over
        as a delta on the device-resident track store + raw IMU samples.  steps: step dicts with a "raw" entry (synth.Filter.step_dict);
        track_frames: dicts as make_track_frame takes them; opts_frame: a dict with stereo / R_cl2cr / t_cl2cr / noise / chi2_table."""
        nb = len(steps)
        sa = (FrameStepRaw * nb)(); fa = (TrackFrame * nb)()
        keeps = []
        for i in range(nb):
            s, k1 = make_step_raw(steps[i]); f, k2 = make_track_frame(track_frames[i])
            sa[i] = s; fa[i] = f; keeps.append((k1, k2))
        o, chi2 = make_opts(opts_frame, max_accept, compress_rule, selected_variant)
        sg = f64(sigma)

        def call():
            _keep = (keeps, chi2, sg)
            self._chk(self.L.ingvio_frame_stage_tracks(self.h, b0, nb, sa, fa, C.byref(o), _d(sg), int(enable_gnss), C.c_double(sigma_cb),
                                                       C.c_double(sigma_rw), 1 if use_async else 0))
        return call

    def frame_stage_prepare(self, b0, steps, frames, sigma, enable_gnss=0, sigma_cb=0.0, sigma_rw=0.0, max_accept=0,
                            compress_rule=1, selected_variant=0, use_async=False):
        """Builds the C argument arrays once and returns a callable that re-issues ingvio_frame_stage on them (for timing
        the host hand-over without the Python marshalling)."""
        nb = len(steps)
        sa = (FrameStep * nb)(); fa = (MsckfFrame * nb)()
        keeps = []
        for i in range(nb):
            s, k1 = make_step(steps[i]); f, k2 = make_frame(frames[i])
            sa[i] = s; fa[i] = f; keeps.append((k1, k2))
        o, chi2 = make_opts(frames[0], max_accept, compress_rule, selected_variant)
        sg = f64(sigma)

        fn = self.L.ingvio_frame_stage_async if use_async else self.L.ingvio_frame_stage

        def call(_keep=(keeps, chi2, sa, fa, o, sg)):
            self._chk(fn(self.h, b0, nb, sa, fa, C.byref(o), _d(sg), int(enable_gnss), C.c_double(sigma_cb), C.c_double(sigma_rw)))
        return call

    def frame_run(self, restore_prior=False):
        self._chk(self.L.ingvio_frame_run(self.h, 1 if restore_prior else 0))

    def set_frame_parts(self, parts):
        """-1 automatic, 1 off, 2..4: slices of the batch that ingvio_frame_run runs on their own streams (ingvio_set_frame_parts)"""
        self._chk(self.L.ingvio_set_frame_parts(self.h, int(parts)))

    def frame_fetch(self, b0=0, nb=None):
        nb = self.batch if nb is None else nb
        dx = np.zeros((nb, self.ldp)); acc = np.zeros((nb, self.f_max), dtype=np.int32); rows = np.zeros(nb, dtype=np.int32)
        self._chk(self.L.ingvio_frame_fetch(self.h, b0, nb, _d(dx), _i(acc), _i(rows)))
        return dx, acc, rows

    def frame_fetch_begin(self, b0=0, nb=None):
        """enqueues the result copies behind the frame's kernels (ingvio_frame_fetch_begin); frame_fetch_end() returns them"""
        nb = self.batch if nb is None else nb
        self._fetch_nb = nb
        self._chk(self.L.ingvio_frame_fetch_begin(self.h, b0, nb))

    def frame_fetch_end(self):
        nb = self._fetch_nb
        dx = np.empty((nb, self.ldp)); acc = np.empty((nb, self.f_max), dtype=np.int32); rows = np.empty(nb, dtype=np.int32)
        self._chk(self.L.ingvio_frame_fetch_end(self.h, _d(dx), _i(acc), _i(rows)))
        return dx, acc, rows

    # ---- triangulation (f-1) ------------------------------------------------------------------
    def triangulate(self, b0, frames, stereo=True, mask_failed=False, **params):
        """frames: list of frame dicts (clone_R, clone_p, obs_mask, uv, R_cl2cr, t_cl2cr; pf/anchor/dof unused), or None
        for the staged frames.  Returns (pf[nb, f_max, 3], ok[nb, f_max])."""
        pr = dict(trans_thres=0.1, huber_epsilon=0.01, conv_precision=5e-7, init_damping=1e-3, outer_loop_max_iter=10,
                  inner_loop_max_iter=10, max_depth=60.0, min_depth=0.2)
        pr.update(params)
        o = TriOpts()
        o.stereo = 1 if stereo else 0
        ref = frames[0] if frames else None
        Rl = f64(ref["R_cl2cr"]).reshape(-1) if ref is not None else np.eye(3).reshape(-1)
        tl = f64(ref["t_cl2cr"]) if ref is not None else np.zeros(3)
        if "R_cl2cr" in params:
            Rl = f64(pr.pop("R_cl2cr")).reshape(-1)
        if "t_cl2cr" in params:
            tl = f64(pr.pop("t_cl2cr"))
        for i in range(9):
            o.R_cl2cr[i] = float(Rl[i])
        for i in range(3):
            o.t_cl2cr[i] = float(tl[i])
        for k, v in pr.items():
            setattr(o, k, v)
        o.mask_failed = 1 if mask_failed else 0
        if frames is None:
            nb = self.batch - b0
            fa = None
            keeps = []
        else:
            nb = len(frames)
            fa = (MsckfFrame * nb)(); keeps = []
            for i in range(nb):
                f, k = make_frame(frames[i]); fa[i] = f; keeps.append(k)
        pf = np.zeros((nb, self.f_max, 3)); ok = np.zeros((nb, self.f_max), dtype=np.int32)
        self._chk(self.L.ingvio_triangulate(self.h, b0, nb, fa, C.byref(o), _d(pf), _i(ok)))
        return pf, ok

    # ---- profiling ---------------------------------------------------------------------------
    def profile_enable(self, on=True):
        self._chk(self.L.ingvio_profile_enable(self.h, 1 if on else 0))

    def profile_select(self, name=None):
        self._chk(self.L.ingvio_profile_select(self.h, name.encode() if name else None))

    def profile_reset(self):
        self._chk(self.L.ingvio_profile_reset(self.h))

    def profile_get(self):
        cap = 32
        names = (C.c_char_p * cap)(); ms = (C.c_double * cap)(); calls = (C.c_int * cap)()
        k = self.L.ingvio_profile_get(self.h, names, ms, calls, cap)
        return {names[i].decode(): (ms[i], calls[i]) for i in range(k)}


class TriOpts(C.Structure):
    _fields_ = [("stereo", C.c_int), ("R_cl2cr", C.c_double * 9), ("t_cl2cr", C.c_double * 3),
                ("trans_thres", C.c_double), ("huber_epsilon", C.c_double), ("conv_precision", C.c_double),
                ("init_damping", C.c_double), ("outer_loop_max_iter", C.c_int), ("inner_loop_max_iter", C.c_int),
                ("max_depth", C.c_double), ("min_depth", C.c_double), ("mask_failed", C.c_int)]


class DeviceCov:
    """Single-filter covariance engine with the same surface as oracle.Cov, backed by filter `b`
    of a Context (so ingvio_amd.synth can build scenarios on the GPU)."""

    def __init__(self, ctx, b, P):
        self.ctx, self.b = ctx, b
        ctx.cov_set(b, P)

    @property
    def n(self):
        return self.ctx.n(self.b)

    @property
    def P(self):
        return np.array(self.ctx.cov_get(self.b))

    def propagate(self, Phi, G, dt, sigma, enable_gnss=0, gnss_idx=(-1,) * 5, sigma_cb=0.0, sigma_rw=0.0):
        self.ctx.propagate(self.b, Phi, G, dt, sigma, enable_gnss, gnss_idx if enable_gnss else None, sigma_cb, sigma_rw)

    def augment(self, R_i2w):
        return int(self.ctx.augment(self.b, R_i2w)[0])

    def marginalize(self, idx, size):
        self.ctx.marginalize(self.b, [idx], size)

    def append_independent(self, blk):
        return int(self.ctx.append_independent(self.b, np.atleast_2d(blk))[0])
