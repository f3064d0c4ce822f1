"""Deterministic synthetic stereo-MSCKF workloads (BASELINE.json configs 2-5).

This is harness code shared by tests and bench.py: it only produces *inputs* (IMU samples,
clone poses, feature observations, index tables).  All filter arithmetic is delegated to the two
callables the caller passes in:

* ``transition(R, p, v, bg, ba, gyro, acc, gravity, dt) -> (R', p', v', Phi, G)`` — the analytic
  IMU transition (ImuPropagator.cpp:98-162); the product one lives in libingvio_host.so
  (``ingvio_amd.host.imu_transition``), tests may pass the oracle's.
* a covariance engine with ``propagate / augment / append_independent / marginalize`` and ``n`` —
  ``ingvio_amd.capi.DeviceCov`` (HIP) or ``oracle.oracle.Cov`` (CPU checker).

Values follow the shipped sports-field stereo configuration
(config/sportsfield/ingvio_stereo.yaml, stereo_{left,right}_config.yaml); only numbers, no code.
"""
import math

import numpy as np

# config/sportsfield/ingvio_stereo.yaml
PARAMS = dict(
    noise_g=0.004, noise_a=0.08, noise_bg=0.0002, noise_ba=0.008,
    # quirk Q1 (State.cpp:51-52): _noise_clockbias <- noise_cb_rw param (0.2), _noise_cb_rw stays 0.2
    sigma_cb=0.2, sigma_rw=0.2,
    init_cov_rot=0.0, init_cov_pos=0.0, init_cov_vel=0.25, init_cov_bg=0.01, init_cov_ba=0.01,
    init_cov_ext_rot=1.8e-2, init_cov_ext_pos=2e-3,
    init_cov_cb=2.0, init_cov_fs=1.0, init_cov_yof=0.015,
    gravity=9.8, visual_noise=0.08, chi2_thres=0.95, chi2_max_dof=150,
)
# stereo_left_config.yaml / stereo_right_config.yaml: R^{imu}_{cam}, t^{imu}_{cam}
R_CL2I = np.array([[0.9999890386957373, -0.0043227774403168, 0.0017989117755288],
                   [0.0043276579084841, 0.9999869417854389, -0.0027180205355500],
                   [-0.0017871388870994, 0.0027257758172719, 0.9999946881262878]])
T_CL2I = np.array([-0.0759472920952561, -0.0039320527565750, -0.0016395029500217])
R_CR2I = np.array([[0.9999014076382304, -0.0133731297219721, 0.0042818692791948],
                   [0.0133731003056063, 0.9999105754655292, 0.0000355022536769],
                   [-0.0042819611512717, 0.0000217631139403, 0.9999908321255077]])
T_CR2I = np.array([0.0341738532732442, -0.0032623030537933, -0.0017782029037505])


def t_cl2cr():
    """T_cl2cr = T_cr2i^-1 * T_cl2i (State.cpp:33)."""
    R = R_CR2I.T @ R_CL2I
    t = R_CR2I.T @ (T_CL2I - T_CR2I)
    return R, t


def chi2_table(max_dof=150, p=0.95):
    """chi2_table[d] = 0.95-quantile of chi^2_d, d = 1..max_dof; [0] unused (Update.cpp:27-34).
    Wilson-Hilferty start + Newton on the regularised lower incomplete gamma: self-contained, equals
    scipy.stats.chi2.ppf to 1e-10 (tests/test_oracle_golden.py)."""
    out = np.zeros(max_dof + 1)
    for k in range(1, max_dof + 1):
        out[k] = _chi2_ppf(p, k)
    return out


def _gammainc_lower_reg(a, x):
    if x <= 0:
        return 0.0
    if x < a + 1.0:  # series
        term = 1.0 / a
        s = term
        n = a
        for _ in range(10000):
            n += 1.0
            term *= x / n
            s += term
            if abs(term) < abs(s) * 1e-17:
                break
        return s * math.exp(-x + a * math.log(x) - math.lgamma(a))
    # continued fraction (Lentz) for Q
    tiny = 1e-300
    b = x + 1.0 - a
    c = 1.0 / tiny
    d = 1.0 / b
    h = d
    for i in range(1, 10000):
        an = -i * (i - a)
        b += 2.0
        d = an * d + b
        if abs(d) < tiny:
            d = tiny
        c = b + an / c
        if abs(c) < tiny:
            c = tiny
        d = 1.0 / d
        delta = d * c
        h *= delta
        if abs(delta - 1.0) < 1e-16:
            break
    q = math.exp(-x + a * math.log(x) - math.lgamma(a)) * h
    return 1.0 - q


def _chi2_ppf(p, k):
    a = 0.5 * k
    # Wilson-Hilferty
    z = 1.6448536269514722 if abs(p - 0.95) < 1e-15 else _norm_ppf(p)
    x = k * (1.0 - 2.0 / (9.0 * k) + z * math.sqrt(2.0 / (9.0 * k))) ** 3
    x = max(x, 1e-8)
    for _ in range(100):
        f = _gammainc_lower_reg(a, 0.5 * x) - p
        pdf = math.exp((a - 1.0) * math.log(0.5 * x) - 0.5 * x - math.lgamma(a)) * 0.5
        step = f / pdf
        xn = x - step
        if xn <= 0:
            xn = 0.5 * x
        if abs(xn - x) < 1e-14 * max(1.0, x):
            x = xn
            break
        x = xn
    return x


def _norm_ppf(p):  # Acklam, refined by one Newton step; only used when p != 0.95
    a = [-3.969683028665376e+01, 2.209460984245205e+02, -2.759285104469687e+02,
         1.383577518672690e+02, -3.066479806614716e+01, 2.506628277459239e+00]
    b = [-5.447609879822406e+01, 1.615858368580409e+02, -1.556989798598866e+02,
         6.680131188771972e+01, -1.328068155288572e+01]
    q = p - 0.5
    r = q * q
    x = (((((a[0] * r + a[1]) * r + a[2]) * r + a[3]) * r + a[4]) * r + a[5]) * q / \
        (((((b[0] * r + b[1]) * r + b[2]) * r + b[3]) * r + b[4]) * r + 1.0)
    e = 0.5 * math.erfc(-x / math.sqrt(2.0)) - p
    x -= e * math.sqrt(2.0 * math.pi) * math.exp(0.5 * x * x)
    return x


# ---------------------------------------------------------------------------------------------
# trajectory: circle r = 5 m at 2 m/s, camera looking along the direction of travel
# ---------------------------------------------------------------------------------------------
RADIUS, SPEED, HEIGHT = 5.0, 2.0, 1.0
OMEGA = SPEED / RADIUS
IMU_DT, IMU_PER_FRAME = 0.005, 10


def true_pose(t):
    """(R_i2w, p, v) of the IMU at time t.  Body axes: x radial-out, y down, z forward."""
    c, s = math.cos(OMEGA * t), math.sin(OMEGA * t)
    p = np.array([RADIUS * c, RADIUS * s, HEIGHT])
    v = SPEED * np.array([-s, c, 0.0])
    R = np.array([[c, 0.0, -s],
                  [s, 0.0, c],
                  [0.0, -1.0, 0.0]])
    return R, p, v


def true_imu(t):
    """Noise-free body-frame (gyro, accel specific force) at time t."""
    R, _, _ = true_pose(t)
    c, s = math.cos(OMEGA * t), math.sin(OMEGA * t)
    a_w = -OMEGA * OMEGA * RADIUS * np.array([c, s, 0.0])
    g_w = np.array([0.0, 0.0, -PARAMS["gravity"]])
    return R.T @ np.array([0.0, 0.0, OMEGA]), R.T @ (a_w - g_w)


def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0.0]])


class Filter:
    """Minimal host-side bookkeeping for the synthetic runs: nominal IMU state, the
    Type::idx()/size() table (State.cpp:62-88, StateManager.cpp:155-231,253-296) and the window of
    clones.  Covariance arithmetic goes to ``cov`` (engine), nominal propagation to ``transition``."""

    def __init__(self, cov_factory, transition, t0=0.0, n_gnss=0, n_landmarks=0, rng=None, lm_sigma=1.0):
        pr = PARAMS
        self.transition = transition
        self.t = t0
        R, p, v = true_pose(t0)
        self.R, self.p, self.v = R.copy(), p.copy(), v.copy()
        self.bg = np.zeros(3)
        self.ba = np.zeros(3)
        self.gravity = np.array([0.0, 0.0, -pr["gravity"]])
        # State ctor + initStateAndCov (State.cpp:60-91,126-167)
        d = np.full(21, 1e-6)
        d[0:3] = pr["init_cov_rot"] ** 2
        d[3:6] = pr["init_cov_pos"] ** 2
        d[6:9] = pr["init_cov_vel"] ** 2
        d[9:12] = pr["init_cov_bg"] ** 2
        d[12:15] = pr["init_cov_ba"] ** 2
        d[15:18] = pr["init_cov_ext_rot"] ** 2
        d[18:21] = pr["init_cov_ext_pos"] ** 2
        self.cov = cov_factory(np.diag(d))
        self.vars = [("SE23", 0, 9), ("bg", 9, 3), ("ba", 12, 3), ("ext", 15, 6)]
        self.gnss_idx = [-1] * 5          # GPS, GLO, GAL, BDS, FS
        self.idx_yof = -1
        self.enable_gnss = 1 if n_gnss else 0
        if n_gnss:
            # addGNSSVariable (StateManager.cpp:216-231) in arrival order
            for g, var in ((0, pr["init_cov_cb"] ** 2), (3, pr["init_cov_cb"] ** 2),
                           (2, pr["init_cov_cb"] ** 2), (1, pr["init_cov_cb"] ** 2),
                           (4, pr["init_cov_fs"] ** 2)):
                self.gnss_idx[g] = self._append("gnss%d" % g, np.array([[var]]))
            self.idx_yof = self._append("yof", np.array([[pr["init_cov_yof"] ** 2]]))
        for i in range(n_landmarks):      # addAnchoredLandmarkInState (StateManager.cpp:298-314), padding
            self._append("lm%d" % i, lm_sigma ** 2 * np.eye(3))
        self.clones = []                  # dicts: name, t, R_c2w, p_c (estimated)

    # -- index bookkeeping ---------------------------------------------------------------
    def _append(self, name, blk):
        idx = self.cov.append_independent(blk)
        self.vars.append((name, idx, blk.shape[0]))
        return idx

    def idx_of(self, name):
        for nm, i, s in self.vars:
            if nm == name:
                return i
        raise KeyError(name)

    def marginalize(self, name):
        """StateManager.cpp:155-192: delete rows/cols, shift later idx by -size."""
        k = [nm for nm, _, _ in self.vars].index(name)
        _, idx, size = self.vars[k]
        self.cov.marginalize(idx, size)
        self.vars = [(nm, i - size if i > idx else i, s) for nm, i, s in self.vars if nm != name]
        for g in range(5):
            if self.gnss_idx[g] > idx:
                self.gnss_idx[g] -= size
        if self.idx_yof > idx:
            self.idx_yof -= size
        self.clones = [c for c in self.clones if c["name"] != name]

    # -- IMU ---------------------------------------------------------------------------------
    def imu_steps(self, rng, k=IMU_PER_FRAME, dt=IMU_DT):
        """k noisy IMU steps ending at t + k*dt; returns the per-step (Phi, G, dt) list and advances
        the nominal state (ImuPropagator.cpp:232-292 with one sample per step)."""
        pr = PARAMS
        steps = []
        # the same samples in raw form (ingvio_frame_step_raw: the transition matrices are then formed on the device)
        self.last_raw = dict(imu=np.zeros((k, 7)), R=np.array(self.R, dtype=float), p=np.array(self.p, dtype=float), v=np.array(self.v, dtype=float),
                             bg=np.array(self.bg, dtype=float), ba=np.array(self.ba, dtype=float), gravity=np.array(self.gravity, dtype=float))
        for q in range(k):
            gyro, acc = true_imu(self.t + dt)
            gyro = gyro + rng.normal(0.0, pr["noise_g"], 3)
            acc = acc + rng.normal(0.0, pr["noise_a"], 3)
            self.last_raw["imu"][q, 0:3] = gyro; self.last_raw["imu"][q, 3:6] = acc; self.last_raw["imu"][q, 6] = dt
            self.R, self.p, self.v, Phi, G = self.transition(self.R, self.p, self.v, self.bg, self.ba,
                                                             gyro, acc, self.gravity, dt)
            self.t += dt
            steps.append((np.asarray(Phi), np.asarray(G), dt))
        return steps

    def sigma(self):
        pr = PARAMS
        return [pr["noise_g"], pr["noise_a"], pr["noise_bg"], pr["noise_ba"]]

    def propagate_cov(self, steps):
        pr = PARAMS
        for Phi, G, dt in steps:
            self.cov.propagate(Phi, G, dt, self.sigma(), self.enable_gnss, self.gnss_idx,
                               pr["sigma_cb"], pr["sigma_rw"])

    def clone(self):
        """augmentSlidingWindowPose (StateManager.cpp:253-296): value T_i2w * T_cl2i, idx = old N."""
        idx = self.cov.augment(self.R)
        name = "clone@%.6f" % self.t
        self.vars.append((name, idx, 6))
        self.clones.append(dict(name=name, t=self.t, R=self.R @ R_CL2I, p=self.p + self.R @ T_CL2I))
        return idx

    def step_dict(self, steps, marg_name=None):
        pr = PARAMS
        return dict(Phi=[s[0] for s in steps], G=[s[1] for s in steps], dt=[s[2] for s in steps],
                    sigma=self.sigma(), enable_gnss=self.enable_gnss, gnss_idx=list(self.gnss_idx),
                    sigma_cb=pr["sigma_cb"], sigma_rw=pr["sigma_rw"], R_i2w=self.R.copy(),
                    marg_idx=-1 if marg_name is None else self.idx_of(marg_name), raw=getattr(self, "last_raw", None))


def true_cam_pose(t):
    R, p, _ = true_pose(t)
    return R @ R_CL2I, p + R @ T_CL2I


def make_features(rng, clone_times, F, noise_px=1e-3, outlier_every=20, stereo=True, pf_sigma=0.02):
    """F landmarks visible from every clone; observations from the TRUE camera poses + pixel noise;
    every ``outlier_every``-th feature gets +0.5 on u0 at its middle observation.
    Returns (pf_est[F,3], uv[F,C,4], outlier[F])."""
    C = len(clone_times)
    Rm, pm = true_cam_pose(clone_times[C // 2])
    Rlr, tlr = t_cl2cr()
    pf = np.zeros((F, 3))
    uv = np.zeros((F, C, 4))
    poses = [true_cam_pose(t) for t in clone_times]
    j = 0
    while j < F:
        depth = rng.uniform(2.0, 20.0)
        x = rng.uniform(-0.5, 0.5) * depth
        y = rng.uniform(-0.35, 0.35) * depth
        pw = Rm @ np.array([x, y, depth]) + pm
        ok = True
        obs = np.zeros((C, 4))
        for c, (Rc, pc) in enumerate(poses):
            q = Rc.T @ (pw - pc)
            qr = Rlr @ q + tlr
            if q[2] < 0.5 or qr[2] < 0.5 or abs(q[0] / q[2]) > 1.2 or abs(q[1] / q[2]) > 1.0:
                ok = False
                break
            obs[c] = [q[0] / q[2], q[1] / q[2], qr[0] / qr[2], qr[1] / qr[2]]
        if not ok:
            continue
        pf[j] = pw
        uv[j] = obs
        j += 1
    uv += rng.normal(0.0, noise_px, uv.shape)
    outlier = np.zeros(F, dtype=bool)
    if outlier_every:
        for j in range(outlier_every - 1, F, outlier_every):
            uv[j, C // 2, 0] += 0.5
            outlier[j] = True
    pf_est = pf + rng.normal(0.0, pf_sigma, pf.shape)
    if not stereo:
        uv[:, :, 2:] = 0.0
    return pf_est, uv, outlier


def frame_from_filter(flt, pf, uv, obs_mask=None, anchor=None, dof=None, stereo=True, noise=None,
                      table=None):
    """Flatten the window + features into the FeatSoA/ClonePose form of the C ABI."""
    C = len(flt.clones)
    F = pf.shape[0]
    Rlr, tlr = t_cl2cr()
    if obs_mask is None:
        obs_mask = np.full(F, (1 << C) - 1, dtype=np.uint64)
    if anchor is None:
        anchor = np.zeros(F, dtype=np.int32)
    if dof is None:  # RemoveLost: #obs - 1 (RemoveLostUpdate.cpp:332-333)
        dof = np.array([bin(int(m)).count("1") - 1 for m in obs_mask], dtype=np.int32)
    return dict(
        clone_idx=np.array([flt.idx_of(c["name"]) for c in flt.clones], dtype=np.int32),
        clone_R=np.stack([c["R"] for c in flt.clones]), clone_p=np.stack([c["p"] for c in flt.clones]),
        pf=np.ascontiguousarray(pf), anchor=np.asarray(anchor, dtype=np.int32),
        obs_mask=np.asarray(obs_mask, dtype=np.uint64), uv=np.ascontiguousarray(uv),
        dof=np.asarray(dof, dtype=np.int32), stereo=1 if stereo else 0, R_cl2cr=Rlr, t_cl2cr=tlr,
        noise=PARAMS["visual_noise"] if noise is None else noise,
        chi2_table=chi2_table() if table is None else table)


def build_case(cov_factory, transition, seed=0, F=150, C=11, n_gnss=6, n_landmarks=52, stereo=True,
               outlier_every=20, table=None, lm_sigma=1.0):
    """Config-2 style case.  Runs C-1 propagate+clone cycles to create a realistic prior, then
    prepares the measured frame: k IMU steps, clone #C, F features seen by all C clones, marginalise
    the oldest clone afterwards.  Returns (flt, step, frame, info) with the covariance engine inside
    ``flt.cov`` holding the PRIOR (N = 21 + n_gnss + 3*n_landmarks + 6*(C-1))."""
    rng = np.random.default_rng(0x1A6F10 + seed)
    flt = Filter(cov_factory, transition, t0=0.1 * seed, n_gnss=n_gnss, n_landmarks=n_landmarks, lm_sigma=lm_sigma)
    for _ in range(C - 1):
        flt.propagate_cov(flt.imu_steps(rng))
        flt.clone()
    # the measured frame: nominal propagation now, covariance work left to the caller
    steps = flt.imu_steps(rng)
    t_new = flt.t
    clone_times = [c["t"] for c in flt.clones] + [t_new]
    oldest = flt.clones[0]["name"]
    step = flt.step_dict(steps, marg_name=oldest)
    # frame inputs need the new clone's pose and idx (= N after propagate, i.e. current n)
    new_idx = flt.cov.n
    pf, uv, outlier = make_features(rng, clone_times, F, stereo=stereo, outlier_every=outlier_every)
    clones = flt.clones + [dict(name="clone@%.6f" % t_new, t=t_new, R=flt.R @ R_CL2I, p=flt.p + flt.R @ T_CL2I)]
    Rlr, tlr = t_cl2cr()
    frame = dict(
        clone_idx=np.array([flt.idx_of(c["name"]) for c in flt.clones] + [new_idx], dtype=np.int32),
        clone_R=np.stack([c["R"] for c in clones]), clone_p=np.stack([c["p"] for c in clones]),
        pf=pf, anchor=np.zeros(F, dtype=np.int32), obs_mask=np.full(F, (1 << C) - 1, dtype=np.uint64),
        uv=uv, dof=np.full(F, C - 1, dtype=np.int32), stereo=1 if stereo else 0,
        R_cl2cr=Rlr, t_cl2cr=tlr, noise=PARAMS["visual_noise"],
        chi2_table=chi2_table() if table is None else table)
    info = dict(outlier=outlier, N_prior=flt.cov.n, N_update=flt.cov.n + 6, new_idx=new_idx,
                marg_idx=step["marg_idx"])
    return flt, step, frame, info


def make_landmarks(rng, flt, frame, n_lm, outlier_every=13, noise=None, untracked=()):
    """In-state SLAM landmarks seen in the CURRENT frame (LandmarkUpdate::updateLandmarkStereo, LandmarkUpdate.cpp:32-149): the
    filter's `lm<i>` blocks become landmarks anchored at clones of the window (never the oldest one, which the frame
    marginalises), placed in view of the current camera, observed with `noise` (every `outlier_every`-th grossly off).
    `frame`: the MSCKF frame dict of the same step (its clone_idx are the anchors' state indices).  Returns the dict
    Context.landmark_stage takes."""
    noise = PARAMS["visual_noise"] if noise is None else noise
    Rlr, tlr = t_cl2cr()
    R_c2w, p_c = flt.R @ R_CL2I, flt.p + flt.R @ T_CL2I
    C = len(frame["clone_idx"])
    pf, uv = np.zeros((n_lm, 3)), np.zeros((n_lm, 4))
    for l in range(n_lm):
        q = np.array([rng.uniform(-1.5, 1.5), rng.uniform(-1.0, 1.0), rng.uniform(3.0, 9.0)])
        pf[l] = R_c2w @ q + p_c
        qr = Rlr @ q + tlr
        uv[l] = [q[0] / q[2], q[1] / q[2], qr[0] / qr[2], qr[1] / qr[2]]
        uv[l] += noise * rng.standard_normal(4) * (30.0 if outlier_every and l % outlier_every == outlier_every - 1 else 1.0)
    tracked = np.ones(n_lm, dtype=np.uint8)
    for l in untracked:
        tracked[l] = 0
    return dict(R_i2w=flt.R.copy(), p_i2w=flt.p.copy(), R_cl2i=R_CL2I.copy(), p_c2i=T_CL2I.copy(), idx_epose=0, idx_ext=15,
                lm_idx=np.array([flt.idx_of("lm%d" % l) for l in range(n_lm)], dtype=np.int32),
                anchor_idx=np.array([int(frame["clone_idx"][1 + l % (C - 1)]) for l in range(n_lm)], dtype=np.int32),
                pf=pf, uv=uv, tracked=tracked)


def make_gnss(rng, flt, n_sat=8, outliers=(5,)):
    """BASELINE config 3 / SURVEY 8(d): 8 satellites on the upper hemisphere (el in [20, 80] deg, az uniform), constellations
    {GPS x4, BDS x2, GAL x2} (gnss_comm::sys2idx GPS 0 / GLO 1 / GAL 2 / BDS 3), ura 2.0, psr_std 1.0, dopp_std 0.5 (cycles/s -> m/s at
    L1), residuals N(0, sigma_row^2); `outliers`: satellites whose pseudo-range residual is gross (the per-row gate must refuse them).
    The values are the OUTPUTS of gnss_comm's psr_res / dopp_res / sat_states: gnss_comm itself is not needed.
    Returns the dict ingvio_amd.host.gnss_rows / the oracle's gnss_rows take (state indices from `flt`)."""
    el = np.deg2rad(rng.uniform(20.0, 80.0, n_sat)); az = rng.uniform(0.0, 2.0 * np.pi, n_sat)
    los = np.stack([np.cos(el) * np.sin(az), np.cos(el) * np.cos(az), np.sin(el)], axis=1)
    sysv = np.array(([0, 0, 0, 0, 3, 3, 2, 2] * ((n_sat + 7) // 8))[:n_sat], dtype=np.int32)
    yaw = rng.uniform(0.0, 2.0 * np.pi)
    # ENU -> ECEF at a mid-latitude point composed with the world yaw offset: any proper rotation serves the covariance path
    lat, lon = np.deg2rad(31.0), np.deg2rad(121.4)
    R_enu2ecef = np.array([[-np.sin(lon), -np.sin(lat) * np.cos(lon), np.cos(lat) * np.cos(lon)],
                           [np.cos(lon), -np.sin(lat) * np.sin(lon), np.cos(lat) * np.sin(lon)],
                           [0.0, np.cos(lat), np.sin(lat)]])
    R_w2enu = np.array([[np.cos(yaw), -np.sin(yaw), 0.0], [np.sin(yaw), np.cos(yaw), 0.0], [0.0, 0.0, 1.0]])
    ura = np.full(n_sat, 2.0); psr_std = np.full(n_sat, 1.0)
    dopp_std_mps = np.full(n_sat, 0.5 * 299792458.0 / 1575.42e6)
    sig_p = np.sqrt(ura * psr_std) / np.sin(el); sig_d = np.sqrt(ura * dopp_std_mps) / np.sin(el)
    res_pos = rng.normal(0.0, 1.0, n_sat) * sig_p; res_vel = rng.normal(0.0, 1.0, n_sat) * sig_d
    for i in outliers:
        if i < n_sat:
            res_pos[i] = 80.0
    return dict(los=los, sys=sysv, res_pos=res_pos, res_vel=res_vel, sin_el=np.sin(el), ura=ura, psr_std=psr_std,
                dopp_std_mps=dopp_std_mps, R_w2ecef=R_enu2ecef @ R_w2enu, p_w=flt.p.copy(), v_w=flt.v.copy(), idx_se23=0,
                idx_yof=flt.idx_yof, idx_cb=np.array(flt.gnss_idx[:4], dtype=np.int32), idx_fs=flt.gnss_idx[4],
                psr_amp=1.0, dopp_amp=1.0)
