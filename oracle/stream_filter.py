"""The POLICY layer of InGVIO's camera callback, restated in Python on top of the CPU oracle's covariance arithmetic.

TEST INFRASTRUCTURE ONLY (same rule as oracle/ingvio_oracle.h): imported by oracle/gen_stream_golden.py and tests/; never by
ingvio_amd/.  Written from the reference sources alone (cited as path:line under /root/reference/ingvio_estimator/src/), NOT
from the C++ shim in ingvio_amd/csrc/host/ — it is the independent second opinion on which tracks go to which update, which
clones are selected, when anchors change and what is erased:

    IngvioFilter::callbackIMU                      IngvioFilter.cpp:381-407
    IngvioFilter::callbackStereoFrame              IngvioFilter.cpp:252-379
    ImuPropagator::storeImu / propagateUntil / propagateAugmentAtEnd   ImuPropagator.cpp:27-68, 232-314
    State::State / initStateAndCov / nextMargTime  State.cpp:61-160, State.h:82-91
    StateManager::augmentSlidingWindowPose / marginalize / boxPlus / margSlidingWindowPose   StateManager.cpp:155-192, 245-296, 318-328
    MapServerManager::collectStereoMeas / markMargStereoFeatures / eraseInvalidFeatures      MapServerManager.cpp:146-217, 245-273, 456-491
    FeatureInfoManager::triangulateFeatureInfoStereo                                          MapServerManager.cpp:309-341
    RemoveLostUpdate::updateStateStereo            RemoveLostUpdate.cpp:276-405
    SwMargUpdate::updateStateStereo / selectSwTimestamps / changeMSCKFAnchor / cleanStereoObsAtMargTime / margSwPose
                                                   SwMargUpdate.cpp:216-365, 475-497, 367-413, 425-446, 415-423
    KeyframeUpdate::getMargKfs / updateStateStereo / changeMSCKFAnchor / cleanStereoObsAtMargTime / margSwPose
                                                   KeyframeUpdate.cpp:43-129, 587-735, 280-328, 737-760, 119-129

    GNSS block of the callback                     IngvioFilter.cpp:329-362; GnssSync.cpp:27-68, 136-194 (buffers, 0.13 s window)
    GnssUpdate::checkYofStatus / updateTrackedSys / addNewTrackedSys / getResJacobianOfSys    GnssUpdate.cpp:33-43, 84-293, 295-523
    GnssManager helpers, gnss_comm psr_res / dopp_res / sat_azel / ecef2geo     GnssManager.cpp:63-133; gnss_spp.cpp:99-146, 256-282;
                                                                                gnss_utility.cpp:347-388, 733-772

The numerical kernels behind it are the oracle's (oracle/ingvio_oracle.c): orc_imu_transition, orc_propagate_cov,
orc_augment_clone, orc_triangulate, orc_msckf_update (per-feature Jacobian, nullspace, chi^2 gate, stacking, compression, EKF
update), orc_marginalize, the retractions; for GNSS orc_gnss_rows (the rows of updateTrackedSys with their per-row gates),
orc_whiten_residual, orc_ekf_update, orc_add_variable_delayed.  Scope: mono and stereo; MSCKF features and - round 6 - in-state SLAM
landmarks (max_landmark_features > 0: LandmarkUpdate.cpp:32-149 / 688-801 update, :363-424 / :892-956 delayed initialisation, :273-361
anchor change, MapServerManager.cpp:225-273 / 343-379 / 456-491; rows orc_landmark_rows_epose, orc_add_variable_delayed,
orc_replace_var_linear); GNSS epochs with the satellite states and atmosphere delays already evaluated (what the INGVIOR1 format carries), the alignment
given (no batchAlign), is_adjust_yof = 0.  Every processed camera frame yields a trace record (see `Trace`)."""
import math

import numpy as np
from scipy.stats import chi2 as _chi2

from . import oracle as orc

INF = float("inf")


def to_sec(ns):
    """ros::Time::toSec() of a stamp given in nanoseconds: sec + 1e-9 * nsec, in this order (bit-exact with the C++ side)."""
    ns = int(ns)
    return float(ns // 1000000000) + 1e-9 * float(ns % 1000000000)


def parse_params(text):
    """The "key: value" lines of an INGVIOR1 PARAMS record -> dict of strings."""
    out = {}
    for line in text.splitlines():
        if not line or line[0] in "#%" or ":" not in line:
            continue
        k, v = line.split(":", 1)
        out[k.strip()] = v.strip()
    return out


def _iso(v):
    a = np.array([float(x) for x in v.split()]).reshape(3, 4)
    return a[:, :3].copy(), a[:, 3].copy()


def quat_from_two_vectors(a, b):
    """Eigen::Quaterniond::FromTwoVectors(a, b) (w, x, y, z), the generic branch (the vectors of the gravity initialisation are
    never opposite)."""
    v0 = a / np.linalg.norm(a)
    v1 = b / np.linalg.norm(b)
    c = float(v1 @ v0)
    assert c > -1.0 + 1e-12
    axis = np.cross(v0, v1)
    s = math.sqrt((1.0 + c) * 2.0)
    invs = 1.0 / s
    return np.array([s * 0.5, axis[0] * invs, axis[1] * invs, axis[2] * invs])


def quat_to_rot(q):
    """Eigen::Quaterniond::toRotationMatrix() of the normalised quaternion (setValueLinearByQuat, PoseState.cpp:225-229)."""
    q = q / np.linalg.norm(q)
    w, x, y, z = q
    tx, ty, tz = 2 * x, 2 * y, 2 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    return np.array([[1 - (tyy + tzz), txy - twz, txz + twy],
                     [txy + twz, 1 - (txx + tzz), tyz - twx],
                     [txz - twy, tyz + twx, 1 - (txx + tyy)]])


# ---- gnss_comm pieces (gnss_constant.hpp:203-214; gnss_utility.cpp; gnss_spp.cpp) ------------------------------------------------------
LIGHT_SPEED = 2.99792458e8
EARTH_OMG_GPS = 7.2921151467e-5
EARTH_SEMI_MAJOR = 6378137.0
EARTH_ECCE_2 = 6.69437999014e-3
D2R = math.pi / 180.0
R2D = 180.0 / math.pi


def ecef2geo(xyz):                                                       # gnss_utility.cpp:347-388
    if xyz[0] == 0 and xyz[1] == 0:
        return np.zeros(3)
    e2, a = EARTH_ECCE_2, EARTH_SEMI_MAJOR
    a2 = a * a
    b2 = a2 * (1 - e2)
    b = math.sqrt(b2)
    ep2 = (a2 - b2) / b2
    p = math.sqrt(xyz[0] * xyz[0] + xyz[1] * xyz[1])
    s1, s2 = xyz[2] * a, p * b
    h = math.sqrt(s1 * s1 + s2 * s2)
    sin_theta, cos_theta = s1 / h, s2 / h
    s1 = xyz[2] + ep2 * b * sin_theta ** 3
    s2 = p - a * e2 * cos_theta ** 3
    h = math.sqrt(s1 * s1 + s2 * s2)
    tan_lat, sin_lat, cos_lat = s1 / s2, s1 / h, s2 / h
    lat = math.atan(tan_lat)
    N = a2 * (a2 * cos_lat * cos_lat + b2 * sin_lat * sin_lat) ** -0.5
    return np.array([lat * R2D, math.atan2(xyz[1], xyz[0]) * R2D, p / cos_lat - N])


def ecef2enu(ref_lla, v):                                                # :733-743
    lat, lon = ref_lla[0] * D2R, ref_lla[1] * D2R
    sl, cl, so, co = math.sin(lat), math.cos(lat), math.sin(lon), math.cos(lon)
    R = np.array([[-so, co, 0.0], [-sl * co, -sl * so, cl], [cl * co, cl * so, sl]])
    return R @ v


def sat_azel(rcv, sat):                                                  # :762-772
    lla = ecef2geo(rcv)
    d = sat - rcv
    d = d / np.linalg.norm(d)
    e = ecef2enu(lla, d)
    az = 0.0 if math.hypot(d[0], d[1]) < 1e-12 else math.atan2(e[0], e[1])
    if az < 0:
        az += 2 * math.pi
    return az, math.asin(e[2])


def psr_res(rcv7, sats):
    """gnss_spp.cpp:99-146 with the observation's own atmosphere delays (the INGVIOR1 record carries calculate_ion_delay /
    calculate_trop_delay's outputs).  -> res [n], unit_rv2sv [n, 3] (= -J.row.head<3>), elevation [n]."""
    n = len(sats)
    res = np.zeros(n); los = np.zeros((n, 3)); el = np.full(n, math.pi / 2.0)
    r = np.asarray(rcv7[:3], dtype=float)
    for i, o in enumerate(sats):
        if np.linalg.norm(r) > 0:
            _, el[i] = sat_azel(r, o["sv_pos"])
        rv2sv = o["sv_pos"] - r
        rng = float(np.linalg.norm(rv2sv))
        sagnac = EARTH_OMG_GPS * (o["sv_pos"][0] * rcv7[1] - o["sv_pos"][1] * rcv7[0]) / LIGHT_SPEED
        est = rng + sagnac + rcv7[3 + o["sys"]] - o["sv_dt"] * LIGHT_SPEED + o["tro"] + o["ion"] + o["tgd"] * LIGHT_SPEED
        los[i] = rv2sv / rng
        res[i] = est - o["psr"]
    return res, los, el


def dopp_res(rcv4, rcv_ecef, sats):                                      # gnss_spp.cpp:256-282
    res = np.zeros(len(sats))
    for i, o in enumerate(sats):
        u = o["sv_pos"] - rcv_ecef
        u = u / np.linalg.norm(u)
        sagnac = EARTH_OMG_GPS / LIGHT_SPEED * (o["sv_vel"][0] * rcv_ecef[1] + o["sv_pos"][0] * rcv4[1]
                                                - o["sv_vel"][1] * rcv_ecef[0] - o["sv_pos"][1] * rcv4[0])
        est = float((o["sv_vel"] - np.asarray(rcv4[:3])) @ u) + rcv4[3] + sagnac - o["sv_ddt"] * LIGHT_SPEED
        if o["freq"] < 0:
            continue
        res[i] = est + o["dopp"] * (LIGHT_SPEED / o["freq"])
    return res


def rot_z(yaw):                                                          # GnssManager::calcRw2enu (GnssManager.cpp:63-66): AngleAxis(yo, UnitZ)
    c, s = math.cos(yaw), math.sin(yaw)
    return np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])


GPS, GLO, GAL, BDS, FS, YOF = range(6)                                   # State::GNSSType (State.h:75)


class Var:
    """One error-state variable (Type: idx, size) with its nominal value."""

    def __init__(self, kind, size):
        self.kind = kind            # 'se23' | 'vec3' | 'se3' | 'scalar'
        self.size = size
        self.idx = -1
        self.R = np.eye(3)
        self.p = np.zeros(3)        # SE23: trans1 (position); SE3: translation; Vec3: the value
        self.v = np.zeros(3)        # SE23: trans2 (velocity)
        self.s = 0.0                # Scalar: the value
        self.anchor = None          # 'lm' (AnchoredLandmark in the state): the anchoring clone Var; .p is the WORLD position (valuePosXyz)
        self.feature = None         # 'lm': the Feature it belongs to (its pf is kept equal to .p)

    def update(self, dx):
        i = self.idx
        if self.kind == "se23":                                          # PoseState.cpp:174-186
            self.R, self.p, self.v = orc.se23_update(self.R, self.p, self.v, dx[i:i + 9])
        elif self.kind == "se3":                                         # PoseState.cpp:79-88
            self.R, self.p = orc.se3_update(self.R, self.p, dx[i:i + 6])
        elif self.kind == "scalar":                                      # VecState.cpp:40-44
            self.s = self.s + dx[i]
        elif self.kind == "lm":                                          # AnchoredLandmark::update (AnchoredLandmark.cpp:227-243): the anchor's d_theta from dx
            a = self.anchor.idx
            dth = dx[a:a + 3]
            self.p = orc.gamma(dth, 0).reshape(3, 3) @ self.p + orc.gamma(dth, 1).reshape(3, 3) @ dx[i:i + 3]
            self.feature.pf = self.p.copy()
        else:                                                            # VecState.cpp:25-29
            self.p = self.p + dx[i:i + 3]


class Feature:
    """FeatureInfo (MapServer.h:69-134)."""

    def __init__(self):
        self.id = -1
        self.slam = False           # _ftype == SLAM: the landmark is a state variable (self.lm)
        self.lm = None
        self.is_to_marg = False
        self.is_tri = False
        self.num_tri = 0
        self.anchor = None          # the clone Var (AnchoredLandmark::_anchored_pose)
        self.pf = np.zeros(3)       # AnchoredLandmark::valuePosXyz (world)
        self.obs = {}               # stamp -> (u0, v0, u1, v1)


class Filter:
    def __init__(self, params_text, overrides=""):
        p = parse_params(params_text)
        p.update(parse_params(overrides))
        g = lambda k, d: float(p.get(k, d))
        gi = lambda k, d: int(float(p.get(k, d)))
        self.max_lm = gi("max_landmark_features", 0)                     # State.cpp: _max_landmarks
        self.landmarks = {}          # id -> 'lm' Var (State::_anchored_landmarks)
        self.enable_gnss = gi("enable_gnss", 1)
        if self.enable_gnss:
            assert gi("is_adjust_yof", 0) == 0, "scope: is_adjust_yof = 0"
            # State.cpp:49-57 incl. quirk Q1: BOTH clock noises are assigned to _noise_clockbias, _noise_cb_rw keeps its default 0.2
            self.sigma_cb = g("noise_rcv_clockbias_randomwalk", 0.2)
            self.sigma_rw = 0.2
            self.init_cov_cb = g("init_cov_rcv_clockbias", 2.0)
            self.init_cov_yof = g("init_cov_yof", 0.015)
            self.gnss_chi2_test = gi("gnss_chi2_test", 0)
            self.gnss_strong_reject = gi("gnss_strong_reject", 1)
            self.psr_amp = g("psr_noise_amp", 1.0)
            self.dopp_amp = g("dopp_noise_amp", 1.0)
        self.gnss = {}               # GNSSType -> scalar Var (State::_gnss)
        self.gnss_buf, self.spp_buf = [], []                              # GnssSync's queues (the replay marks the synchronisation as done)
        self.unsync_thres = 0.13                                          # GnssSync.h:61
        self.align = None            # dict(yaw_offset, R_enu2ecef, anchor_ecef) once aligned: given with the recording, or found by batchAlign
        self.aligner = None          # GvioAligner (oracle/gvio_align.py), created with the first raw epoch
        self.gv_batch, self.gv_max_iter = gi("gv_align_batch_size", 25), gi("gv_align_max_iter", 10)
        self.gv_eps, self.gv_vel_thres = g("gv_align_conv_epsilon", 1e-5), g("gv_align_vel_thres", 0.4)
        self.stereo = gi("cam_nums", 2) == 2                             # IngvioFilter.cpp:100-112: cam_nums 1 -> the mono callback
        self.max_sw = gi("max_sliding_window_poses", 27)
        self.is_key_frame = gi("is_key_frame", 1)
        self.sigma = [g("noise_gyro", 0.004), g("noise_accel", 0.08), g("noise_bias_gyro", 0.0002), g("noise_bias_accel", 0.008)]
        self.init_cov = dict(rot=g("init_cov_rot", 0.0), pos=g("init_cov_pos", 0.0), vel=g("init_cov_vel", 0.25), bg=g("init_cov_bg", 0.01),
                             ba=g("init_cov_ba", 0.01), ext_rot=g("init_cov_ext_rot", 1.8e-2), ext_pos=g("init_cov_ext_pos", 2e-3))
        self.init_gravity = g("gravity_norm", 9.8)
        self.max_imu_buffer = gi("max_imu_buffer_size", 3000)
        self.init_imu_sp = gi("init_imu_buffer_sp", 300)
        self.tri = dict(trans_thres=g("trans_thres", 0.25), huber_epsilon=g("huber_epsilon", 0.01), conv_precision=g("conv_precision", 5e-7),
                        init_damping=g("init_damping", 1e-3), outer_loop_max_iter=gi("outer_loop_max_iter", 10),
                        inner_loop_max_iter=gi("inner_loop_max_iter", 10), max_depth=g("max_depth", 40.0), min_depth=g("min_depth", 0.2))
        self.chi2_max_dof = gi("chi2_max_dof", 150)
        self.chi2_thres = g("chi2_thres", 0.95)
        self.noise = g("visual_noise", 0.18)
        self.frame_select_interval = gi("frame_select_interval", 18)
        # the two quirks the shim exposes as parameters; defaults = the reference as written
        self.max_valid_ids = gi("hip_max_valid_ids", 20)                 # RemoveLostUpdate.h:38
        self.compress_rule = gi("hip_compress_rule", 0)                  # RemoveLostUpdate.cpp:390 keeps row_cnt rows
        R_cl2i, t_cl2i = _iso(p["T_cl2i"])
        R_cr2i, t_cr2i = _iso(p["T_cr2i"])
        # State.cpp:33: _T_cl2cr = T_cr2i^-1 * T_cl2i
        self.R_cl2cr = R_cr2i.T @ R_cl2i
        self.t_cl2cr = R_cr2i.T @ (t_cl2i - t_cr2i)
        # State::State (State.cpp:61-92): SE23 | bg | ba | extrinsics, cov = 1e-6 I
        self.ext_pose = Var("se23", 9)
        self.bg = Var("vec3", 3)
        self.ba = Var("vec3", 3)
        self.extr = Var("se3", 6)
        self.extr.R, self.extr.p = R_cl2i.copy(), t_cl2i.copy()
        self.T_cl2i = (R_cl2i, t_cl2i)
        self.err_vars = [self.ext_pose, self.bg, self.ba, self.extr]
        idx = 0
        for v in self.err_vars:
            v.idx = idx
            idx += v.size
        self.cov = orc.Cov(1e-3 ** 2 * np.eye(idx), ld=((idx + 6 + 6 * (self.max_sw + 3) + 3 * self.max_lm + 15) // 16) * 16)
        self.timestamp = -1.0
        self.sw = {}                 # stamp -> clone Var
        self.map = {}                # id -> Feature
        # ImuPropagator (ImuPropagator.h:98-110)
        self.imu_buf = []            # (stamp, accel, gyro)
        self.has_gravity = self.init_imu_sp < 0
        self.gravity = np.array([0.0, 0.0, -self.init_gravity])
        self.quat_init = np.array([1.0, 0.0, 0.0, 0.0])
        self.has_image_come = False
        self.has_init_state = False
        self.select_cnt = 0          # KeyframeUpdate::_select_cnt (static)
        self.kf_timestamp = -1.0     # KeyframeUpdate::_timestamp
        self.kfs = []
        self.chi2_table = {}
        self.frames = 0

    # ---- helpers --------------------------------------------------------------------------------------------------------
    def chi2(self, dof):
        if dof not in self.chi2_table:
            self.chi2_table[dof] = float(_chi2.ppf(self.chi2_thres, dof))
        return self.chi2_table[dof]

    def sw_sorted(self):
        return sorted(self.sw.items())

    def next_marg_time(self):                                            # State.h:82-91
        if len(self.sw) > self.max_sw:
            return min(self.sw)
        return INF

    def box_plus(self, dx):                                              # StateManager.cpp:245-251
        for v in self.err_vars:
            v.update(dx)

    def marginalize(self, var):                                          # StateManager.cpp:155-192
        assert var in self.err_vars
        self.cov.marginalize(var.idx, var.size)
        rest = []
        for v in self.err_vars:
            if v is not var:
                if v.idx > var.idx:
                    v.idx -= var.size
                rest.append(v)
        var.idx = -1
        self.err_vars = rest

    def marg_sw_pose(self, t):                                           # StateManager.cpp:318-328
        self.marginalize(self.sw[t])
        del self.sw[t]

    # ---- IMU ------------------------------------------------------------------------------------------------------------
    def callback_imu(self, stamp, gyro, accel):                          # IngvioFilter.cpp:381-407
        if not self.has_image_come:
            return
        self.store_imu(stamp, np.asarray(gyro, dtype=float), np.asarray(accel, dtype=float))
        if self.has_gravity and not self.has_init_state:
            # State::initStateAndCov(stamp, quat) (State.cpp:126-160): zero position / velocity / biases
            self.timestamp = stamp
            ic = self.init_cov
            d = np.concatenate([[ic["rot"] ** 2] * 3, [ic["pos"] ** 2] * 3, [ic["vel"] ** 2] * 3, [ic["bg"] ** 2] * 3, [ic["ba"] ** 2] * 3,
                                [ic["ext_rot"] ** 2] * 3, [ic["ext_pos"] ** 2] * 3])
            for i in range(21):
                self.cov.buf[i, i] = d[i]
            self.ext_pose.R = quat_to_rot(self.quat_init)
            self.ext_pose.p = np.zeros(3)
            self.ext_pose.v = np.zeros(3)
            self.bg.p = np.zeros(3)
            self.ba.p = np.zeros(3)
            self.extr.R, self.extr.p = self.T_cl2i[0].copy(), self.T_cl2i[1].copy()
            self.has_init_state = True

    def store_imu(self, stamp, gyro, accel):                             # ImuPropagator.cpp:27-68
        if len(self.imu_buf) > self.max_imu_buffer:
            return
        self.imu_buf.append((stamp, accel, gyro))
        if not self.has_gravity and self.init_imu_sp > 0:
            if len(self.imu_buf) < self.init_imu_sp:
                return
            s = np.zeros(3)
            for _, a, _ in self.imu_buf:
                s = s + a
            s = s / len(self.imu_buf)
            n = float(np.linalg.norm(s))
            if abs(n - self.init_gravity) / self.init_gravity > 0.02:
                self.imu_buf = []
                self.has_gravity = False
            else:
                self.gravity = np.array([0.0, 0.0, -n])
                self.quat_init = quat_from_two_vectors(-s, self.gravity)
                self.has_gravity = True

    def transition(self, ctrl, dt):                                      # ImuPropagator.cpp:98-162 + StateManager.cpp:42-119
        _, accel, gyro = ctrl
        e = self.ext_pose
        self.timestamp += dt
        e.R, e.p, e.v, Phi, G = orc.imu_transition(e.R, e.p, e.v, self.bg.p, self.ba.p, gyro, accel, self.gravity, dt)
        if self.enable_gnss:
            if FS in self.gnss:                                           # ImuPropagator.cpp:139-148: cb += dt * fs
                for t4 in (GPS, GLO, GAL, BDS):
                    if t4 in self.gnss:
                        self.gnss[t4].s = self.gnss[t4].s + dt * self.gnss[FS].s
            gi5 = [self.gnss[t5].idx if t5 in self.gnss else -1 for t5 in (GPS, GLO, GAL, BDS, FS)]
            self.cov.propagate(Phi, G, dt, self.sigma, 1, gi5, self.sigma_cb, self.sigma_rw)
            return
        self.cov.propagate(Phi, G, dt, self.sigma)

    def propagate_until(self, t_end):                                    # ImuPropagator.cpp:232-292
        if not self.has_gravity or t_end <= self.timestamp:
            return
        if not self.imu_buf:
            return
        if self.imu_buf[0][0] > t_end:
            return
        propa_cnt = 0
        last = self.imu_buf[-1]
        for ctrl in self.imu_buf:
            t = ctrl[0]
            if t < self.timestamp:
                propa_cnt += 1
                continue
            if t > t_end:
                break
            propa_cnt += 1
            dt = t - self.timestamp
            if dt < 1e-6:
                continue
            last = ctrl
            self.transition(ctrl, dt)
        if self.timestamp < t_end:
            dt_last = t_end - self.timestamp
            if dt_last > 1e-6:
                self.transition(last, dt_last)
            else:
                self.timestamp = t_end
        del self.imu_buf[:propa_cnt]

    def augment(self):                                                   # StateManager.cpp:253-296
        if self.timestamp in self.sw:
            return
        c = Var("se3", 6)
        e = self.ext_pose
        c.R = e.R @ self.extr.R
        c.p = e.R @ self.extr.p + e.p
        c.idx = self.cov.n
        self.sw[self.timestamp] = c
        self.err_vars.append(c)
        self.cov.augment(e.R)

    def propagate_augment_at_end(self, t_end):                           # ImuPropagator.cpp:294-314
        if not self.has_gravity:
            return
        self.propagate_until(t_end)
        if self.timestamp < t_end or self.timestamp > t_end:
            return
        self.augment()

    # ---- map server -----------------------------------------------------------------------------------------------------
    def collect_meas(self, feats):                                       # MapServerManager.cpp:105-217 (mono and stereo twins)
        ts = self.timestamp
        assert ts in self.sw
        for rec in feats:
            fid, u0, v0 = int(rec[0]), rec[1], rec[2]
            u1, v1 = (rec[3], rec[4]) if self.stereo else (0.0, 0.0)
            fi = self.map.get(fid)
            if fi is None:
                fi = Feature()
                self.map[fid] = fi
            if not fi.obs:
                fi.obs[ts] = (u0, v0, u1, v1)
                fi.id = fid
                fi.is_to_marg = False
                fi.is_tri = False
                fi.anchor = self.sw[ts]
            else:
                if ts in fi.obs:
                    continue
                if fi.slam:                                              # MapServerManager.cpp:136-137 / :178-179: a landmark keeps its latest observation only
                    fi.obs.clear()
                fi.obs[ts] = (u0, v0, u1, v1)
                fi.is_to_marg = False

    def triangulate(self, fi):                                           # MapServerManager.cpp:309-341 over Triangulator.cpp:320-359
        sw = self.sw_sorted()
        C = len(sw)
        cR = np.stack([c.R for _, c in sw])
        cp = np.stack([c.p for _, c in sw])
        uv = np.zeros((C, 4))
        mask = 0
        for s, (t, _) in enumerate(sw):                                  # filterCommonTimestamp: obs stamps that are window stamps
            if t in fi.obs:
                uv[s] = fi.obs[t]
                mask |= 1 << s
        if mask == 0:
            return False
        ok, pf = orc.triangulate(cR, cp, mask, uv, self.stereo, self.R_cl2cr, self.t_cl2cr, **self.tri)      # :274-307 mono / :309-341 stereo
        if not ok or np.any(np.isnan(pf)):
            return False
        fi.num_tri += 1
        body = fi.anchor.R.T @ (pf - fi.anchor.p)
        if body[2] <= 0:
            return False
        fi.pf = pf.copy()
        fi.is_tri = True
        return True

    def flat_frame(self, ids, sel, dof_of):
        """The window + the listed features as the oracle's flattened frame.  sel: None = every observation (RemoveLost), else the
        selected stamps; dof_of(fi) = the dof handed to testChiSquared."""
        sw = self.sw_sorted()
        C = len(sw)
        slot_of = {id(c): s for s, (_, c) in enumerate(sw)}
        F = len(ids)
        fr = dict(clone_idx=[c.idx for _, c in sw], clone_R=np.stack([c.R for _, c in sw]), clone_p=np.stack([c.p for _, c in sw]),
                  pf=np.zeros((F, 3)), anchor=np.zeros(F, dtype=np.int32), obs_mask=np.zeros(F, dtype=np.uint64), uv=np.zeros((F, C, 4)),
                  dof=np.zeros(F, dtype=np.int32), stereo=1 if self.stereo else 0, R_cl2cr=self.R_cl2cr, t_cl2cr=self.t_cl2cr, noise=self.noise)
        max_dof = 1
        for j, fid in enumerate(ids):
            fi = self.map[fid]
            fr["pf"][j] = fi.pf
            fr["anchor"][j] = slot_of[id(fi.anchor)]
            m = 0
            for s, (t, _) in enumerate(sw):
                if t in fi.obs and (sel is None or t in sel):
                    fr["uv"][j, s] = fi.obs[t]
                    m |= 1 << s
            fr["obs_mask"][j] = m
            fr["dof"][j] = dof_of(fi)
            max_dof = max(max_dof, int(fr["dof"][j]))
        fr["chi2_table"] = np.array([0.0] + [self.chi2(d) for d in range(1, max_dof + 1)])
        return fr

    # ---- the three updates ----------------------------------------------------------------------------------------------
    def remove_lost_update(self, tr):                                    # RemoveLostUpdate.cpp:276-405
        ts = self.timestamp
        lost_lm = []
        for fid in sorted(self.map):                                     # markMargStereoFeatures, MapServerManager.cpp:245-273 (mono :225-243)
            fi = self.map[fid]
            if ts not in fi.obs:
                fi.is_to_marg = True
                if fi.slam:
                    lost_lm.append(fid)
        for fid in lost_lm:                                              # a landmark that lost track leaves the state at once
            self.marg_landmark(fid)
            del self.map[fid]
        update_ids, direct = [], []
        for fid in sorted(self.map):
            fi = self.map[fid]
            if fi.is_to_marg and not fi.slam:                            # RemoveLostUpdate.cpp:285: MSCKF type only
                if self.triangulate(fi) and len(fi.obs) >= (3 if self.stereo else 4):        # RemoveLostUpdate.cpp:287 / :51
                    update_ids.append(fid)
                else:
                    direct.append(fid)
        for fid in direct:
            del self.map[fid]
        tr["lost_ids"] = list(update_ids)
        tr["lost_direct"] = list(direct)
        tr["lost_acc"] = []
        tr["lost_rows"] = 0
        if not update_ids:
            return
        fr = self.flat_frame(update_ids, None, lambda fi: len(fi.obs) - 1)                    # :332-333
        dx, acc, gam, m = self.cov.msckf_update(fr, max_accept=self.max_valid_ids, compress_rule=self.compress_rule, selected_variant=0)
        tr["lost_acc"] = [int(a) for a in acc]
        tr["lost_rows"] = int(m)
        if m > 0:                                                        # :399-400
            self.box_plus(dx)
        for fid in update_ids:                                           # :402-403
            del self.map[fid]

    def selected_update(self, sel, dof, tr):                             # SwMargUpdate.cpp:216-365 / KeyframeUpdate.cpp:587-735
        tr["sel_stamps"] = list(sel)
        tr["sel_ids"] = []
        tr["sel_acc"] = []
        tr["sel_rows"] = 0
        for t in sel:
            assert t in self.sw, "selected timestamp not in sw"
        update_ids = []
        for fid in sorted(self.map):
            fi = self.map[fid]
            if fi.slam:                                                  # SwMargUpdate.cpp:240 / KeyframeUpdate.cpp:611
                continue
            if any(t not in fi.obs for t in sel):
                continue
            if self.triangulate(fi):
                update_ids.append(fid)
        tr["sel_ids"] = list(update_ids)
        if not update_ids:
            return
        selset = set(sel)
        fr = self.flat_frame(update_ids, selset, lambda fi: dof)
        dx, acc, gam, m = self.cov.msckf_update(fr, max_accept=0, compress_rule=1, selected_variant=1)
        tr["sel_acc"] = [int(a) for a in acc]
        tr["sel_rows"] = int(m)
        if m > 0:
            self.box_plus(dx)

    def select_sw_timestamps(self, marg_time):                           # SwMargUpdate.cpp:475-497
        if marg_time == INF or marg_time not in self.sw:
            return []
        out = [marg_time]
        cnt = 1
        for t, _ in self.sw_sorted():
            if t <= marg_time:
                continue
            if cnt % self.frame_select_interval == 0:
                out.append(t)
            cnt += 1
        return out

    def get_marg_kfs(self):                                              # KeyframeUpdate.cpp:43-117
        if len(self.sw) < self.max_sw or self.max_sw < 3:
            return []
        if self.timestamp == self.kf_timestamp and self.kfs:
            return list(self.kfs)
        assert len(self.sw) <= self.max_sw, "Current sw poses larger than max size!"
        self.kf_timestamp = self.timestamp
        rem = self.max_sw - 2
        idx1 = 2 + self.select_cnt
        self.select_cnt = (self.select_cnt + 1) % rem
        desc = sorted(self.sw, reverse=True)                             # rbegin() order
        self.kfs = [desc[idx1], desc[1]]
        return list(self.kfs)

    def clean_obs(self, stamps, tr):                                     # SwMargUpdate.cpp:425-446 / KeyframeUpdate.cpp:737-760
        gone = []
        for fid in sorted(self.map):
            fi = self.map[fid]
            for t in stamps:
                fi.obs.pop(t, None)
                if not fi.obs and fid not in gone:
                    gone.append(fid)
        for fid in gone:
            del self.map[fid]
        tr["clean_erased"] = gone

    def change_anchor(self, old_stamps, min_depth, tr):                  # SwMargUpdate.cpp:367-413 / KeyframeUpdate.cpp:280-328
        old = [self.sw[t] for t in old_stamps]
        new_anchor = self.sw[max(self.sw)]
        gone, moved = [], []
        for fid in sorted(self.map):
            fi = self.map[fid]
            if fi.slam:                                                  # SwMargUpdate.cpp:384 / KeyframeUpdate.cpp:301
                continue
            if any(fi.anchor is o for o in old):
                if fi.is_tri:
                    body = new_anchor.R.T @ (fi.pf - new_anchor.p)
                    if body[2] <= min_depth:
                        gone.append(fid)
                        continue
                    fi.anchor = new_anchor                               # resetAnchoredPose(new_anchor, true): the world value stays
                    moved.append(fid)
                else:
                    gone.append(fid)
        for fid in gone:
            del self.map[fid]
        tr["anchor_erased"] = gone
        tr["anchor_moved"] = moved

    def erase_invalid(self, tr):                                         # MapServerManager.cpp:456-491
        gone = []
        for fid in sorted(self.map):
            fi = self.map[fid]
            if not fi.is_tri:
                continue
            if fi.anchor is None:
                gone.append(fid)
                continue
            body = fi.anchor.R.T @ (fi.pf - fi.anchor.p)
            if body[2] <= 0.2:
                gone.append(fid)
        for fid in gone:
            if self.map[fid].slam:                                       # :485-486
                self.marg_landmark(fid)
            del self.map[fid]
        tr["invalid_erased"] = gone

    # ---- in-state SLAM landmarks (max_landmark_features > 0) ---------------------------------------------------------------------------
    def marg_landmark(self, fid):                                        # StateManager::margAnchoredLandmarkInState (StateManager.cpp:340-353)
        if fid not in self.landmarks:
            return
        self.marginalize(self.landmarks[fid])
        del self.landmarks[fid]

    def update_landmarks(self, tr):                                      # LandmarkUpdate.cpp:32-149 (mono) / :688-801 (stereo)
        """Per in-state landmark: the rows of its CURRENT observation w.r.t. [extended pose 9 | extrinsics 6 | anchor 6 | landmark 3]
        (:521-572 / :619-686), a chi^2 gate on the prior with dof = rows (Update.cpp:81-102), the accepted rows stacked with a var_order that
        grows in order of first appearance, one ekfUpdate.  _anchored_landmarks is an unordered_map: the iteration order is unspecified and
        the posterior does not depend on it - ascending id here."""
        tr["lm_upd_ids"], tr["lm_upd_acc"] = [], []
        if not self.landmarks:
            return
        e, x = self.ext_pose, self.extr
        rows_per = 4 if self.stereo else 2
        var_order, col_of = [e, x], {id(e): 0, id(x): 9}
        col_cnt = 15
        Hs, rs = [], []
        for fid in sorted(self.landmarks):
            lm = self.landmarks[fid]
            fi = self.map[fid]
            assert fi.slam and self.timestamp in fi.obs and fi.lm is lm, "landmark in state not tracked to curr time"
            H, res = orc.landmark_rows_epose(e.R, e.p, x.R, x.p, lm.p, np.array(fi.obs[self.timestamp]), self.stereo, self.R_cl2cr, self.t_cl2cr)
            vo = [e, x, lm.anchor, lm]
            gam = self.cov.whiten([v.idx for v in vo], [v.size for v in vo], H, res, self.noise ** 2)
            ok = bool(gam < self.chi2(len(res)))
            tr["lm_upd_ids"].append(fid)
            tr["lm_upd_acc"].append(int(ok))
            if not ok:
                continue
            for v in (lm.anchor, lm):
                if id(v) not in col_of:
                    col_of[id(v)] = col_cnt
                    col_cnt += v.size
                    var_order.append(v)
            Hs.append((H, [col_of[id(v)] for v in vo], [v.size for v in vo]))
            rs.append(res)
        if not Hs:
            return
        m = rows_per * len(Hs)
        Hl = np.zeros((m, col_cnt))
        for k, (H, cols, sizes) in enumerate(Hs):
            c0 = 0
            for c, sz in zip(cols, sizes):
                Hl[rows_per * k:rows_per * (k + 1), c:c + sz] = H[:, c0:c0 + sz]
                c0 += sz
        dx, rc = self.cov.ekf_update([v.idx for v in var_order], [v.size for v in var_order], Hl, np.concatenate(rs), self.noise ** 2)
        self.box_plus(dx)

    def feat_all_obs_rows(self, fi):                                     # LandmarkUpdate.cpp:426-500 (mono) / :803-890 (stereo)
        sw = self.sw_sorted()
        col = {id(c): 6 * k for k, (_, c) in enumerate(sw)}
        per = 4 if self.stereo else 2
        skew = lambda v: np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])
        Hx, Hf, res = [], [], []
        for t in sorted(fi.obs):
            if t not in self.sw:
                continue
            c = self.sw[t]
            q = c.R.T @ (fi.pf - c.p)
            Hp = np.array([[1 / q[2], 0, -q[0] / q[2] ** 2], [0, 1 / q[2], -q[1] / q[2] ** 2]])
            D = np.zeros((3, 6 * len(sw)))
            if c is not fi.anchor:
                D[:, col[id(c)]:col[id(c)] + 3] = c.R.T @ skew(fi.pf)
                D[:, col[id(fi.anchor)]:col[id(fi.anchor)] + 3] = -D[:, col[id(c)]:col[id(c)] + 3]
            D[:, col[id(c)] + 3:col[id(c)] + 6] = -c.R.T
            blocks_x, blocks_f, pred = [Hp @ D], [Hp @ c.R.T], [q[0] / q[2], q[1] / q[2]]
            if self.stereo:
                qr = self.R_cl2cr @ q + self.t_cl2cr
                Hr = np.array([[1 / qr[2], 0, -qr[0] / qr[2] ** 2], [0, 1 / qr[2], -qr[1] / qr[2] ** 2]])
                blocks_x.append(Hr @ self.R_cl2cr @ D)
                blocks_f.append(Hr @ self.R_cl2cr @ c.R.T)
                pred += [qr[0] / qr[2], qr[1] / qr[2]]
            Hx.append(np.vstack(blocks_x)); Hf.append(np.vstack(blocks_f))
            res.append(np.array(fi.obs[t][:per]) - np.array(pred))
        return np.concatenate(res), np.vstack(Hx), np.vstack(Hf)

    def init_new_landmarks(self, tr):                                    # LandmarkUpdate.cpp:363-424 (mono) / :892-956 (stereo); min_init_poses = max_sw
        tr["lm_init_ids"] = []
        if len(self.sw) < self.max_sw:
            return
        vac = self.max_lm - len(self.landmarks)
        if vac <= 0:
            return
        ids = []
        for fid in sorted(self.map):                                     # MapServer is a std::map<int, ...>: ascending id
            if len(ids) >= vac:
                break
            fi = self.map[fid]
            if len(fi.obs) < self.max_sw or fi.slam:
                continue
            if not self.triangulate(fi):
                continue
            ids.append(fid)
        sw = self.sw_sorted()
        for fid in ids:
            fi = self.map[fid]
            res, Hx, Hf = self.feat_all_obs_rows(fi)
            n_old = self.cov.n
            added, dx, chi2 = self.cov.add_variable_delayed([c.idx for _, c in sw], [6] * len(sw), Hx, Hf, res, self.noise, 0.95, True)
            if not added:
                continue
            lm = Var("lm", 3)
            lm.idx, lm.p, lm.anchor, lm.feature = n_old, fi.pf.copy(), fi.anchor, fi
            self.err_vars.append(lm)
            self.landmarks[fid] = lm
            fi.lm, fi.slam = lm, True
            self.box_plus(dx)                                            # the ekfUpdate of the remaining rows (StateManager.cpp:623-624)
            tr["lm_init_ids"].append(fid)

    def change_landmark_anchor(self, old_stamps):                        # LandmarkUpdate.cpp:273-316 (sliding window) / :318-361 (key frames)
        if not old_stamps:
            return
        old = [self.sw[t] for t in old_stamps]
        latest = max(self.sw)
        new_anchor = self.sw[latest]
        to_marg = []
        for fid in sorted(self.map):
            fi = self.map[fid]
            if not fi.slam or not any(fi.lm.anchor is o for o in old):
                continue
            body = new_anchor.R.T @ (fi.lm.p - new_anchor.p)
            if body[2] <= 0:
                to_marg.append(fid)
                continue
            self.change_anchored_pose(fi, latest)
            if fi.lm.anchor is not new_anchor:
                to_marg.append(fid)
        for fid in to_marg:
            self.marg_landmark(fid)
            del self.map[fid]

    def change_anchored_pose(self, fi, target):                          # FeatureInfoManager::changeAnchoredPose, MapServerManager.cpp:343-379
        if len(self.sw) < 2 or target not in self.sw or fi.id not in self.landmarks or not fi.slam:
            return
        if not any(c is fi.lm.anchor for c in self.sw.values()) or self.landmarks[fi.id] is not fi.lm:
            return
        skew = lambda v: np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])
        vo = [fi.lm.anchor, self.sw[target], fi.lm]
        H = np.zeros((3, 15))
        H[:, 0:3] = -skew(fi.lm.p)
        H[:, 6:9] = skew(fi.lm.p)
        H[:, 12:15] = np.eye(3)
        self.cov.replace_var_linear(fi.lm.idx, 3, [v.idx for v in vo], [v.size for v in vo], H)
        fi.lm.anchor = self.sw[target]                                   # resetAnchoredPose(.., true): the world value stays
        fi.anchor = fi.lm.anchor

    # ---- GNSS ---------------------------------------------------------------------------------------------------------------
    def callback_gnss_meas(self, stamp, sats, raw=None):                 # GnssSync::bufferGnssMeas (GnssSync.cpp:27-46)
        """raw: None or dict(eph [n, 25], obs [n, 6], ion [8], doy) - the epoch's ephemerides and observations as GnssProcessor hands them
        on (GnssData.h GnssMeas = (obs, ephems)); only GvioAligner::batchAlign reads them"""
        if len(self.gnss_buf) > 100:
            del self.gnss_buf[:len(self.gnss_buf) - 100]
        self.gnss_buf.append((stamp, sats, raw))

    def callback_spp_meas(self, stamp, pos7, vel4):                      # GnssSync::bufferSppMeas (:48-68)
        if len(self.spp_buf) > 100:
            del self.spp_buf[:len(self.spp_buf) - 100]
        self.spp_buf.append((stamp, np.asarray(pos7, dtype=float), np.asarray(vel4, dtype=float)))

    def _pick(self, buf, target):                                        # getGnssMeasAt / getSppAt (:136-194)
        while buf:
            t = buf[0][0]
            if t < target - self.unsync_thres:
                buf.pop(0)
                continue
            if t >= target + self.unsync_thres:
                break
            return buf.pop(0)
        return None

    def add_gnss_variable(self, gtype, value, cov):                      # StateManager::addGNSSVariable (StateManager.cpp:216-231)
        v = Var("scalar", 1)
        v.s = float(value)
        v.idx = self.cov.append_independent(np.array([[cov]]))
        self.gnss[gtype] = v
        self.err_vars.append(v)

    def _rcv_state(self, cb_over=None, fs_over=None):
        """xyzt / dopp of updateTrackedSys (GnssUpdate.cpp:101-113) and addNewTrackedSys (:343-372)."""
        al, e = self.align, self.ext_pose
        Rw2enu = rot_z(self.gnss[YOF].s)
        xyzt = np.zeros(7)
        xyzt[:3] = al["R_enu2ecef"] @ (Rw2enu @ e.p) + al["anchor_ecef"]                      # getTenu2ecef() * calcTw2enu(yof) * p
        for t4 in (GPS, GLO, GAL, BDS):                                  # GnssManager::getClockbiasVec
            if t4 in self.gnss:
                xyzt[3 + t4] = self.gnss[t4].s
        if cb_over:
            for t4, val in cb_over.items():
                xyzt[3 + t4] = val
        dopp = np.zeros(4)
        dopp[:3] = al["R_enu2ecef"] @ (Rw2enu @ e.v)
        if fs_over is not None:
            dopp[3] = fs_over
        elif FS in self.gnss:
            dopp[3] = self.gnss[FS].s
        return xyzt, dopp, al["R_enu2ecef"] @ Rw2enu

    def update_tracked_sys(self, sats, tr):                              # GnssUpdate.cpp:84-293
        tr["gnss_rows"] = 0
        tr["gnss_keep"] = []
        if not sats:
            return
        if YOF not in self.gnss or FS not in self.gnss or not any(t4 in self.gnss for t4 in (GPS, GLO, GAL, BDS)):      # checkGnssStates
            return
        xyzt, dopp, Rw2ecef = self._rcv_state()
        res_pos, los, el = psr_res(xyzt, sats)
        res_vel = dopp_res(dopp, xyzt[:3], sats)
        e = self.ext_pose
        g = dict(los=los, sys=[o["sys"] for o in sats], res_pos=res_pos, res_vel=res_vel, sin_el=np.sin(el),
                 ura=[o["ura"] for o in sats], psr_std=[o["psr_std"] for o in sats],
                 dopp_std_mps=[o["dopp_std"] * LIGHT_SPEED / o["freq"] for o in sats], R_w2ecef=Rw2ecef, p_w=e.p, v_w=e.v,
                 idx_se23=e.idx, idx_yof=self.gnss[YOF].idx, idx_fs=self.gnss[FS].idx,
                 idx_cb=[self.gnss[t4].idx if t4 in self.gnss else -1 for t4 in (GPS, GLO, GAL, BDS)],
                 psr_amp=self.psr_amp, dopp_amp=self.dopp_amp, chi2_test=self.gnss_chi2_test,
                 chi2_table=np.array([0.0] + [self.chi2(d) for d in range(1, 41)]))
        H, res, Rd, vidx, vsize = orc.gnss_rows(self.cov, g)             # rows with their per-row gates (:148-272), var_order as built there
        m = len(res)
        # which candidate rows survived: the ungated row set, matched by (residual, variance)
        g0 = dict(g); g0["chi2_test"] = 0
        H0, res0, Rd0, _, _ = orc.gnss_rows(self.cov, g0)
        keep, j = [], 0
        for i in range(len(res0)):
            if j < m and res0[i] == res[j] and Rd0[i] == Rd[j]:
                keep.append(1); j += 1
            else:
                keep.append(0)
        assert j == m
        tr["gnss_keep"] = keep
        if m <= 14 and self.gnss_strong_reject:                          # :286: the block gate, dof = rows (0 rows: dof <= 0 -> refused)
            if m <= 0 or not (self.cov.whiten(vidx, vsize, H, res, Rd) < self.chi2(m)):
                return
        dx, _ = self.cov.ekf_update(vidx, vsize, H, res, Rd)             # :289 (zero rows: K has no columns, nothing changes)
        self.box_plus(dx)
        tr["gnss_rows"] = m

    def add_new_tracked_sys(self, sats, pos7, vel4, tr):                 # GnssUpdate.cpp:295-470
        tr["gnss_added"] = []
        if not sats or YOF not in self.gnss:
            return
        spp_sys = ([FS] if abs(vel4[3]) > 1e-3 else []) + [t4 for t4 in (GPS, GLO, GAL, BDS) if abs(pos7[3 + t4]) > 1e-3]      # getSysInSppMeas
        # calcSysToAdd + the loop at :375 iterate std::unordered_set<GNSSType>: with libstdc++ (the reference's toolchain; checked with
        # g++ for all 32 subsets) the two reversals cancel and the order is the insertion order FS, GPS, GLO, GAL, BDS
        to_add = [t for t in spp_sys if t not in self.gnss]
        if not to_add:
            return
        cb_over = {t4: pos7[3 + t4] for t4 in to_add if t4 != FS}
        fs_over = vel4[3] if FS in to_add else None
        xyzt, dopp, Rw2ecef = self._rcv_state(cb_over, fs_over)
        res_pos, los, el = psr_res(xyzt, sats)
        res_vel = dopp_res(dopp, xyzt[:3], sats)
        sin_el = np.sin(el)
        sin_el = np.where(np.abs(sin_el) < 1e-6, 1e-6, sin_el)
        e = self.ext_pose
        x_vidx, x_vsize = [e.idx, self.gnss[YOF].idx], [9, 1]
        for gtype in to_add:
            fs = gtype == FS
            rows = list(range(len(sats))) if fs else [i for i, o in enumerate(sats) if o["sys"] == gtype]      # getResJacobianOfSys :472-521
            if not rows:
                continue
            m = len(rows)
            Hx = np.zeros((m, 10)); Hf = np.ones((m, 1)); res = np.zeros(m)
            x = e.v if fs else e.p                                       # read INSIDE the loop (:389 / :440): an earlier addition's update moved it
            RS = Rw2ecef @ np.array([[0.0, -x[2], x[1]], [x[2], 0.0, -x[0]], [-x[1], x[0], 0.0]])
            avg = 0.0
            for r, i in enumerate(rows):
                u = los[i]
                Hx[r, 0:3] = u @ RS
                Hx[r, (6 if fs else 3):(9 if fs else 6)] = -(u @ Rw2ecef)
                res[r] = -(res_vel[i] if fs else res_pos[i])
                std = sats[i]["dopp_std"] * LIGHT_SPEED / sats[i]["freq"] if fs else sats[i]["psr_std"]
                avg += sats[i]["ura"] * std / (sin_el[i] * sin_el[i])
            noise = (self.dopp_amp if fs else self.psr_amp) * math.sqrt(avg / m)
            if m <= 1:                                                   # addVariableDelayed: H_new.rows() <= H_new.cols() (StateManager.cpp:571-575)
                continue
            added, dx, _ = self.cov.add_variable_delayed(x_vidx, x_vsize, Hx, Hf, res, noise, chi2_mult=0.95, do_chi2=True,
                                                         chi2_check=float(_chi2.ppf(0.95, m)))
            if not added:
                continue
            v = Var("scalar", 1)
            v.s = float(vel4[3] if fs else pos7[3 + gtype])
            v.idx = self.cov.n - 1
            self.err_vars.append(v)
            if m > 1:                                                    # Hup.rows() > 0: the EKF update with the remaining rows (:631-632)
                self.box_plus(dx)
            self.gnss[gtype] = v
            tr["gnss_added"].append(gtype)

    def gnss_block(self, stamp, tr):                                     # IngvioFilter.cpp:329-362
        tr["gnss_rows"], tr["gnss_keep"], tr["gnss_added"], tr["gnss_epoch"] = 0, [], [], -1.0
        if not self.enable_gnss:
            return
        spp = self._pick(self.spp_buf, stamp)
        gm = self._pick(self.gnss_buf, stamp)
        if gm is None:
            return
        tr["gnss_epoch"] = gm[0]
        if spp is not None and self.align is None and gm[2] is not None:  # :344-345: flag && !isAlign() -> batchAlign on the raw epoch
            if self.aligner is None:
                from . import gvio_align
                self.aligner = gvio_align.Aligner(orc, self.gv_batch, self.gv_max_iter, self.gv_eps, self.gv_vel_thres)
            raw = gm[2]
            self.aligner.batch_align(dict(eph=raw["eph"], obs=raw["obs"], doy=raw["doy"]), self.ext_pose.p, self.ext_pose.v, raw["ion"])
            if self.aligner.aligned:
                self.align = dict(yaw_offset=self.aligner.yaw_offset, R_enu2ecef=self.aligner.R_enu2ecef, anchor_ecef=self.aligner.anchor_ecef)
        if self.align is None:
            return                                                       # not aligned (yet): no GNSS update (:347)
        if YOF not in self.gnss:                                         # checkYofStatus :33-43
            self.add_gnss_variable(YOF, self.align["yaw_offset"], self.init_cov_yof)
        self.update_tracked_sys(gm[1], tr)
        if spp is not None:
            self.add_new_tracked_sys(gm[1], spp[1], spp[2], tr)

    # ---- the camera callback ----------------------------------------------------------------------------------------------
    def callback_frame(self, stamp, feats):                              # IngvioFilter.cpp:252-379 (stereo) / :124-250 (mono): the same sequence
        """feats: iterable of (id, u0, v0, u1, v1) or, mono, (id, u0, v0).  Returns the frame's trace dict, or None when the callback
        returned early."""
        if not self.has_image_come:
            self.has_image_come = True
            return None
        if not self.has_init_state:
            return None
        if self.timestamp >= stamp:
            return None
        self.propagate_augment_at_end(stamp)
        if self.timestamp < stamp:
            return None
        tr = dict(stamp=stamp, lm_upd_ids=[], lm_upd_acc=[], lm_init_ids=[])
        lm_before = set(self.landmarks)
        self.collect_meas(feats)
        self.remove_lost_update(tr)
        tr["marg_stamps"] = []
        tr["clean_erased"] = []
        tr["anchor_erased"] = []
        tr["anchor_moved"] = []
        if self.is_key_frame:
            sel = self.get_marg_kfs()
            if sel:
                self.selected_update(sel, 2, tr)                         # KeyframeUpdate.cpp:675-676: dof 2
            else:
                tr.update(sel_stamps=[], sel_ids=[], sel_acc=[], sel_rows=0)
            if self.max_lm > 0:                                          # IngvioFilter.cpp:283-289
                self.update_landmarks(tr)
                self.init_new_landmarks(tr)
            kfs = self.get_marg_kfs()
            self.clean_obs(kfs, tr)                                      # cleanStereoObsAtMargTime
            kfs = self.get_marg_kfs()
            if kfs:
                self.change_anchor(kfs, 0.3, tr)                         # changeMSCKFAnchor, body.z() <= 0.3
            if self.max_lm > 0:                                          # :295-301
                self.change_landmark_anchor(self.get_marg_kfs())
            kfs = self.get_marg_kfs()
            for t in kfs:                                                # margSwPose
                self.marg_sw_pose(t)
            tr["marg_stamps"] = list(kfs)
        else:
            marg_time = self.next_marg_time()
            if marg_time != INF:
                sel = self.select_sw_timestamps(marg_time)
                self.selected_update(sel, len(sel) - 1, tr)              # SwMargUpdate.cpp:306-307
            else:
                tr.update(sel_stamps=[], sel_ids=[], sel_acc=[], sel_rows=0)
            if self.max_lm > 0:                                          # :309-315
                self.update_landmarks(tr)
                self.init_new_landmarks(tr)
            marg_time = self.next_marg_time()
            if marg_time != INF:
                self.clean_obs([marg_time], tr)
            marg_time = self.next_marg_time()
            if marg_time != INF and marg_time in self.sw:
                self.change_anchor([marg_time], 0.0, tr)                 # body.z() <= 0
            marg_time = self.next_marg_time()
            if self.max_lm > 0 and marg_time != INF and marg_time in self.sw:      # :321-322
                self.change_landmark_anchor([marg_time])
            marg_time = self.next_marg_time()
            if marg_time != INF:
                self.marg_sw_pose(marg_time)
                tr["marg_stamps"] = [marg_time]
        self.erase_invalid(tr)
        self.gnss_block(stamp, tr)
        self.frames += 1
        # the state after the frame
        tr["table"] = [(v.idx, v.size) for v in self.err_vars]
        tr["sw_stamps"] = sorted(self.sw)
        tr["map_ids"] = sorted(self.map)
        e = self.ext_pose
        tr["pose"] = np.concatenate([e.R.reshape(-1), e.p, e.v, self.bg.p, self.ba.p, self.extr.R.reshape(-1), self.extr.p])
        tr["gnss_vals"] = np.array([self.gnss[t6].s if t6 in self.gnss else np.nan for t6 in range(6)])      # GPS GLO GAL BDS FS YOF
        tr["align"] = np.array([0.0, 0.0, 0.0, 0.0, 0.0]) if self.align is None else np.r_[1.0, self.align["yaw_offset"], self.align["anchor_ecef"]]
        # in-state landmarks after the frame (ascending id), their world positions, and what left the state during the frame
        tr["lm_ids"] = sorted(self.landmarks)
        tr["lm_vals"] = np.concatenate([self.landmarks[i].p for i in tr["lm_ids"]]) if self.landmarks else np.zeros(0)
        tr["lm_marg_ids"] = sorted((lm_before | set(tr["lm_init_ids"])) - set(self.landmarks))
        P = self.cov.P
        tr["n"] = int(P.shape[0])
        tr["diag"] = np.diag(P).copy()
        tr["norm"] = float(np.linalg.norm(P))
        tr["P"] = P
        return tr


LIST_KEYS_INT = ["lost_ids", "lost_direct", "lost_acc", "sel_ids", "sel_acc", "clean_erased", "anchor_erased", "anchor_moved",
                 "invalid_erased", "map_ids", "gnss_keep", "gnss_added", "lm_upd_ids", "lm_upd_acc", "lm_init_ids", "lm_ids", "lm_marg_ids"]
LIST_KEYS_F64 = ["sel_stamps", "marg_stamps", "sw_stamps", "diag", "gnss_vals", "lm_vals", "align"]
OPTIONAL_KEYS = ("lm_upd_ids", "lm_upd_acc", "lm_init_ids", "lm_ids", "lm_marg_ids", "lm_vals", "align")      # absent from the golden files of rounds 4-5 (no landmarks)
SCALAR_KEYS = ["stamp", "lost_rows", "sel_rows", "n", "norm", "gnss_rows", "gnss_epoch"]


def pack_traces(traces):
    """list of trace dicts -> dict of flat numpy arrays (ragged lists as values + offsets) for np.savez."""
    out = {}
    for k in LIST_KEYS_INT + LIST_KEYS_F64:
        dt = np.int64 if k in LIST_KEYS_INT else np.float64
        vals = [np.asarray(t[k], dtype=dt).reshape(-1) for t in traces]
        out[k] = np.concatenate(vals) if vals else np.zeros(0, dtype=dt)
        out[k + "_off"] = np.concatenate([[0], np.cumsum([len(v) for v in vals])]).astype(np.int64)
    tab = [np.asarray(t["table"], dtype=np.int64).reshape(-1) for t in traces]
    out["table"] = np.concatenate(tab)
    out["table_off"] = np.concatenate([[0], np.cumsum([len(v) for v in tab])]).astype(np.int64)
    for k in SCALAR_KEYS:
        out[k] = np.array([t[k] for t in traces], dtype=np.float64)
    out["pose"] = np.stack([t["pose"] for t in traces])
    return out


def unpack_traces(z):
    """inverse of pack_traces (without the full covariance)."""
    nf = len(z["stamp"])
    out = []
    for f in range(nf):
        t = {}
        for k in LIST_KEYS_INT + LIST_KEYS_F64:
            if k in OPTIONAL_KEYS and k not in z:
                t[k] = np.zeros(0, dtype=np.int64 if k in LIST_KEYS_INT else np.float64)
                continue
            o = z[k + "_off"]
            t[k] = z[k][o[f]:o[f + 1]]
        o = z["table_off"]
        t["table"] = z["table"][o[f]:o[f + 1]].reshape(-1, 2)
        for k in SCALAR_KEYS:
            t[k] = float(z[k][f])
        t["pose"] = z["pose"][f]
        out.append(t)
    return out


def _stamp_from_ns(ns):
    """decodeGnss / decodeSpp (Replay.cpp): m.stamp = 1e-9 * (double)stamp_ns - NOT ros::Time::toSec()."""
    return 1e-9 * float(ns)


def play_recording(path, overrides="", max_frames=None):
    """Plays an INGVIOR1 recording (IMU + STEREO_FRAME records) into a Filter; returns the list of trace dicts."""
    import struct
    flt = None
    traces = []
    raw_pending = None
    with open(path, "rb") as f:
        assert f.read(8) == b"INGVIOR1"
        while True:
            h = f.read(13)
            if not h:
                break
            typ, ns, n = struct.unpack("<BQI", h)
            payload = f.read(n)
            if typ == 0:
                flt = Filter(payload.decode("ascii"), overrides)
            elif typ == 1:
                v = struct.unpack("<6d", payload)
                flt.callback_imu(to_sec(ns), v[0:3], v[3:6])
            elif typ == 8:                                               # GNSS_RAW: doy, ion 8, u32 n, n x { eph 25, obs 6 }: belongs to the GNSS_MEAS that follows
                doy = struct.unpack_from("<d", payload)[0]
                ion = np.array(struct.unpack_from("<8d", payload, 8))
                cnt = struct.unpack_from("<I", payload, 72)[0]
                rec31 = np.array(struct.unpack_from("<%dd" % (31 * cnt), payload, 76)).reshape(cnt, 31)
                raw_pending = (ns, dict(eph=rec31[:, :25].copy(), obs=rec31[:, 25:].copy(), ion=ion, doy=doy))
            elif typ == 4:                                               # GNSS_MEAS: u32 n, n x { i32 sys, f64 x 17 } (Replay.h)
                cnt = struct.unpack_from("<I", payload)[0]
                sats = []
                for i in range(cnt):
                    v = struct.unpack_from("<i17d", payload, 4 + 140 * i)
                    sats.append(dict(sys=v[0], psr=v[1], dopp=v[2], psr_std=v[3], dopp_std=v[4], freq=v[5], sv_pos=np.array(v[6:9]),
                                     sv_vel=np.array(v[9:12]), sv_dt=v[12], sv_ddt=v[13], tgd=v[14], ura=v[15], ion=v[16], tro=v[17]))
                raw = raw_pending[1] if raw_pending is not None and raw_pending[0] == ns else None
                raw_pending = None
                flt.callback_gnss_meas(_stamp_from_ns(ns), sats, raw)
            elif typ == 5:                                               # SPP_MEAS: posSpp 7, velSpp 4
                v = struct.unpack("<11d", payload)
                flt.callback_spp_meas(_stamp_from_ns(ns), v[:7], v[7:])
            elif typ == 6:                                               # ALIGNMENT: aligned, yaw_offset, R_enu2ecef 9 (row-major), anchor_ecef 3
                v = struct.unpack("<14d", payload)
                if v[0] != 0.0:
                    flt.align = dict(yaw_offset=v[1], R_enu2ecef=np.array(v[2:11]).reshape(3, 3), anchor_ecef=np.array(v[11:14]))
            elif typ in (2, 3):
                cnt = struct.unpack_from("<I", payload)[0]
                feats = [struct.unpack_from("<Q4d", payload, 4 + 40 * i) for i in range(cnt)] if typ == 3 else \
                        [struct.unpack_from("<Q2d", payload, 4 + 24 * i) for i in range(cnt)]
                tr = flt.callback_frame(to_sec(ns), feats)
                if tr is not None:
                    traces.append(tr)
                    if max_frames and len(traces) >= max_frames:
                        break
    return traces
