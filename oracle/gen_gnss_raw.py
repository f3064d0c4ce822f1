"""Raw GNSS epochs for a synthetic recording (TEST INFRASTRUCTURE): rewrites an INGVIOR1 file written by `ingvio_replay --synth ...,gnss=1
--write` so that the filter has to ALIGN ITSELF - the ALIGNMENT record is dropped and every GNSS_MEAS record is preceded by a GNSS_RAW
record (Replay.h type 8: Kepler ephemerides + L1 pseudo-range / Doppler observations + Klobuchar parameters) of the same epoch.

The raw epochs are consistent with the stream's ground truth (ingvio_amd/csrc/host/SynthStream.cpp: the camera on a circle, the world
frame = the truth frame shifted to the starting point, rotated against ENU at (31 N, 121.4 E, 30 m) by the true yaw offset 0.3 rad,
receiver clock biases 150 / 165 / 180 m + 5 m/s drift): GvioAligner::batchAlign must find that yaw and that anchor from them
(GvioAligner.cpp:85-383).  The evaluated satellite observations the GNSS UPDATES use stay the C++ generator's - what is pinned is the
alignment the filter computes and everything that follows from it.  Deterministic: numpy Generator seeded per stream."""
import math
import struct

import numpy as np

from . import gen_gnss_golden as gg
from .gvio_align import geo2rotation, rotz

T_STATIC = 2.0            # SynthStream.cpp:26
YO_TRUE = 0.3             # :100
CB0 = np.array([150.0, 0.0, 165.0, 180.0])      # :103 (GLONASS unused)
FS = 5.0
T0_GPS = 360300.0         # seconds of the GPS week at stream time 0
DOY = 270.4
ION = np.array([0.1118e-07, 0.2235e-07, -0.1192e-06, -0.1192e-06, 0.1167e+06, 0.1802e+06, -0.1311e+06, -0.4588e+06])


def truth_pv(tau):
    """SynthStream.cpp truthAt: position and velocity of the IMU in the truth frame"""
    if tau <= 0:
        th, thd = 0.0, 0.0
    elif tau <= 2.0:
        th, thd = 0.1 * tau * tau, 0.2 * tau
    else:
        th, thd = 0.4 + 0.4 * (tau - 2.0), 0.4
    c, s = math.cos(th), math.sin(th)
    return np.array([5 * c, 5 * s, 1.0]), np.array([-5 * s * thd, 5 * c * thd, 0.0])


def read_records(path):
    out = []
    with open(path, "rb") as f:
        assert f.read(8) == b"INGVIOR1"
        while True:
            h = f.read(13)
            if not h:
                break
            typ, ns, n = struct.unpack("<BQI", h)
            out.append((typ, ns, f.read(n)))
    return out


def add_raw_epochs(path_in, path_out, seed=20261001):
    rng = np.random.default_rng(seed)
    lla = np.array([31.0, 121.4, 30.0])
    anchor = gg.geo2ecef(lla)
    R = geo2rotation(lla)
    Rw = R @ rotz(YO_TRUE)
    eph = gg.make_constellation(rng, anchor, T0_GPS + T_STATIC + 1.5)      # GPS x 4, BDS x 2, GAL x 2, all above 15 degrees
    eph[:, 24] = 3.5                                                    # psr_pos divides the weights by ura - 1 (GPS, BDS) / ura - 2 (GAL)
    p0, _ = truth_pv(0.0)
    recs = read_records(path_in)
    n_raw = 0
    with open(path_out, "wb") as f:
        f.write(b"INGVIOR1")
        for typ, ns, payload in recs:
            if typ == 6:                                                 # ALIGNMENT: the filter aligns itself
                continue
            if typ == 4:
                t = 1e-9 * ns
                p, v = truth_pv(t - T_STATIC)
                rcv = anchor + Rw @ (p - p0)
                vel = Rw @ v
                obs = gg.make_obs(rng, eph, rcv, vel, CB0 + FS * t, FS, ION, DOY, T0_GPS + t, noise=True)
                body = struct.pack("<d8dI", DOY, *ION, len(eph)) + b"".join(struct.pack("<25d6d", *eph[i], *obs[i]) for i in range(len(eph)))
                f.write(struct.pack("<BQI", 8, ns, len(body)) + body)
                n_raw += 1
            f.write(struct.pack("<BQI", typ, ns, len(payload)) + payload)
    return n_raw
