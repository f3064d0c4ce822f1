"""INGVIOR1 replay files (ingvio_amd/csrc/host/Replay.h): writer / reader and a synthetic recording of BASELINE config 1 — the
sports-field MONO configuration (config/sportsfield/ingvio_mono.yaml) with max_pts_frame = 150 and an 11-pose sliding window —
as the topics /imu0 and /mono_tracker/mono_feature would carry it.  Harness code: it only produces INPUT files for the C++
replay driver (tools/ingvio_replay.cpp); no filter arithmetic lives here."""
import struct

import numpy as np

MAGIC = b"INGVIOR1"
PARAMS, IMU, MONO_FRAME, STEREO_FRAME, GNSS_MEAS, SPP_MEAS, ALIGNMENT, GROUND_TRUTH = range(8)


class Writer:
    def __init__(self, path):
        self.f = open(path, "wb")
        self.f.write(MAGIC)

    def _put(self, typ, stamp, payload):
        self.f.write(struct.pack("<BQI", typ, int(round(stamp * 1e9)), len(payload)) + payload)

    def params(self, text):
        self._put(PARAMS, 0.0, text.encode("ascii"))

    def imu(self, stamp, gyro, accel):
        self._put(IMU, stamp, struct.pack("<6d", *gyro, *accel))

    def mono(self, stamp, ids, uv):
        b = struct.pack("<I", len(ids)) + b"".join(struct.pack("<Qdd", int(i), float(u), float(v)) for i, (u, v) in zip(ids, uv))
        self._put(MONO_FRAME, stamp, b)

    def stereo(self, stamp, ids, uv4):
        b = struct.pack("<I", len(ids)) + b"".join(struct.pack("<Q4d", int(i), *map(float, q)) for i, q in zip(ids, uv4))
        self._put(STEREO_FRAME, stamp, b)

    def truth(self, stamp, p, q_xyzw):
        self._put(GROUND_TRUTH, stamp, struct.pack("<7d", *p, *q_xyzw))

    def close(self):
        self.f.close()


def read(path):
    """-> list of (type, stamp_seconds, payload bytes)"""
    out = []
    with open(path, "rb") as f:
        assert f.read(8) == MAGIC, "not an INGVIOR1 replay file"
        while True:
            h = f.read(13)
            if not h:
                break
            typ, ns, n = struct.unpack("<BQI", h)
            out.append((typ, ns * 1e-9, f.read(n)))
    return out


def decode_mono(payload):
    n = struct.unpack_from("<I", payload)[0]
    rec = [struct.unpack_from("<Qdd", payload, 4 + 24 * i) for i in range(n)]
    return np.array([r[0] for r in rec], dtype=np.uint64), np.array([[r[1], r[2]] for r in rec])


# ---- BASELINE config 1 -------------------------------------------------------------------------------------------------
def _truth(tau):
    """circle of radius 5 m, angular rate ramping 0 -> 0.4 rad/s over 2 s; camera/IMU axes as in the C++ shim tests (z forward
    along the tangent, y down).  Returns R_i2w, p, v, body angular rate, body specific force."""
    if tau <= 0:
        th, thd, thdd = 0.0, 0.0, 0.0
    elif tau <= 2.0:
        th, thd, thdd = 0.1 * tau * tau, 0.2 * tau, 0.2
    else:
        th, thd, thdd = 0.4 + 0.4 * (tau - 2.0), 0.4, 0.0
    c, s = np.cos(th), np.sin(th)
    R = np.array([[c, 0.0, -s], [s, 0.0, c], [0.0, -1.0, 0.0]])
    p = np.array([5 * c, 5 * s, 1.0])
    v = np.array([-5 * s * thd, 5 * c * thd, 0.0])
    a = np.array([-5 * c * thd * thd - 5 * s * thdd, -5 * s * thd * thd + 5 * c * thdd, 0.0])
    return R, p, v, R.T @ np.array([0.0, 0.0, thd]), R.T @ (a + np.array([0.0, 0.0, 9.8]))


def quat_xyzw(R):
    w = np.sqrt(max(0.0, 1 + R[0, 0] + R[1, 1] + R[2, 2])) / 2
    if w > 1e-6:
        return np.array([(R[2, 1] - R[1, 2]) / (4 * w), (R[0, 2] - R[2, 0]) / (4 * w), (R[1, 0] - R[0, 1]) / (4 * w), w])
    x = np.sqrt(max(0.0, 1 + R[0, 0] - R[1, 1] - R[2, 2])) / 2
    return np.array([x, (R[0, 1] + R[1, 0]) / (4 * x), (R[0, 2] + R[2, 0]) / (4 * x), (R[2, 1] - R[1, 2]) / (4 * x)])


CONFIG1_PARAMS = """%YAML:1.0
# config/sportsfield/ingvio_mono.yaml with max_sliding_window_poses: 11 (BASELINE config 1), GNSS topics absent from the recording
cam_nums: 1
max_sliding_window_poses: 11
is_key_frame: 0
max_landmark_features: 0
enable_gnss: 0
noise_gyro: 0.004
noise_accel: 0.08
noise_bias_gyro: 0.0002
noise_bias_accel: 0.008
init_cov_rot: 0.0
init_cov_pos: 0.0
init_cov_vel: 0.25
init_cov_bg: 0.01
init_cov_ba: 0.01
init_cov_ext_rot: 1.8e-02
init_cov_ext_pos: 2e-03
gravity_norm: 9.8
max_imu_buffer_size: 3000
init_imu_buffer_sp: 300
trans_thres: 0.25
huber_epsilon: 0.01
conv_precision: 5e-08
init_damping: 1e-03
outer_loop_max_iter: 10
inner_loop_max_iter: 10
max_depth: 60.0
min_depth: 0.2
chi2_max_dof: 150
chi2_thres: 0.95
visual_noise: 0.18
frame_select_interval: 28
T_cl2i: 0.9999890386957373 -0.0043227774403168 0.0017989117755288 -0.0759472920952561 0.0043276579084841 0.9999869417854389 -0.0027180205355500 -0.0039320527565750 -0.0017871388870994 0.0027257758172719 0.9999946881262878 -0.0016395029500217
hip_f_max: 160
"""


def write_config1(path, seconds=6.0, max_pts=150, seed=0, pixel_noise=1e-3):
    """IMU at 200 Hz (2 s static for the gravity initialisation, IngvioFilter.cpp:396-406, then the circle), mono frames at 20 Hz
    with up to `max_pts` tracked points (feature_tracker's max_pts_frame, mono_config.yaml:41), ground truth per frame."""
    from . import synth
    rng = np.random.default_rng(seed)
    w = Writer(path)
    w.params(CONFIG1_PARAMS)
    R_ci, t_ci = synth.R_CL2I, synth.T_CL2I
    t0_motion = 2.0
    live = {}                      # id -> world point
    next_id = 1
    n_imu = int(round((t0_motion + seconds) * 200))
    stats = dict(frames=0, features=0, imu=0)
    for k in range(1, n_imu + 1):
        t = k * 0.005
        R, p, v, wb, fb = _truth(t - t0_motion)
        w.imu(t, wb + rng.normal(0, 0.004, 3), fb + rng.normal(0, 0.08, 3))
        stats["imu"] += 1
        if k % 10 == 0:
            Rc, pc = R @ R_ci, p + R @ t_ci
            keep = {}
            for i, pw in live.items():
                q = Rc.T @ (pw - pc)
                if q[2] > 0.5 and abs(q[0] / q[2]) < 0.9 and abs(q[1] / q[2]) < 0.7:
                    keep[i] = pw
            live = keep
            while len(live) < max_pts:
                d = rng.uniform(3.0, 15.0)
                live[next_id] = Rc @ np.array([rng.uniform(-0.8, 0.8) * d, rng.uniform(-0.6, 0.6) * d, d]) + pc
                next_id += 1
            ids = sorted(live)
            uv = []
            for i in ids:
                q = Rc.T @ (live[i] - pc)
                uv.append((q[0] / q[2] + rng.normal(0, pixel_noise), q[1] / q[2] + rng.normal(0, pixel_noise)))
            w.mono(t, ids, uv)
            w.truth(t, p, quat_xyzw(R))
            stats["frames"] += 1; stats["features"] += len(ids)
    w.close()
    return stats
