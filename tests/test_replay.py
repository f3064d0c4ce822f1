"""the rosbag-free replay format and BASELINE config 1 (sports-field mono, max_pts_frame = 150, 11
sliding-window poses) driven through callbackIMU / callbackMonoFrame by the C++ replay driver."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

EXE = os.path.join(ROOT, "ingvio_amd", "lib", "ingvio_replay")


def _records_line(out):
    line = [l for l in out.splitlines() if l.startswith("RECORDS")][-1]
    return {k: float(v) for k, v in (kv.split("=") for kv in line.split()[1:])}


def test_replay_file_round_trip_python_and_cpp(tmp_path):
    """the Python writer, the Python reader and the C++ reader (ingvio_replay --dump: no device needed) agree on a recording"""
    from ingvio_amd import replay
    path = str(tmp_path / "config1_short.ingvior")
    st = replay.write_config1(path, seconds=1.0, seed=3)
    recs = replay.read(path)
    kinds = [r[0] for r in recs]
    assert kinds[0] == replay.PARAMS and kinds.count(replay.IMU) == st["imu"] == 600 and kinds.count(replay.MONO_FRAME) == st["frames"] == 60
    stamps = [r[1] for r in recs[1:]]
    assert all(b >= a for a, b in zip(stamps, stamps[1:]))                       # time-ordered like a bag
    ids, uv = replay.decode_mono([r for r in recs if r[0] == replay.MONO_FRAME][-1][2])
    assert len(ids) == 150 and len(set(ids.tolist())) == 150 and np.abs(uv).max() < 1.0
    assert b"max_sliding_window_poses: 11" in recs[0][2] and b"cam_nums: 1" in recs[0][2]
    if not os.path.exists(EXE):
        pytest.skip("ingvio_replay not built")
    out = subprocess.run([EXE, path, "--dump"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    rec = _records_line(out.stdout)
    assert rec["params"] == 1 and rec["imu"] == 600 and rec["mono"] == 60 and rec["truth"] == 60 and rec["features"] == st["features"]
    # a corrupted file is refused, not played
    bad = str(tmp_path / "bad.ingvior")
    with open(path, "rb") as f:
        blob = f.read()
    with open(bad, "wb") as f:
        f.write(blob[:len(blob) // 2 + 5])
    out = subprocess.run([EXE, bad, "--dump"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 1 and "truncated" in out.stderr


@pytest.mark.gpu
def test_config1_replay_end_to_end(tmp_path):
    """BASELINE config 1: 6 s of the recording through the callbacks on the device (mono MSCKF: RemoveLost + SwMarg updates,
    triangulation on the device), odometry against the ground-truth records."""
    from ingvio_amd import replay
    path = str(tmp_path / "config1.ingvior")
    st = replay.write_config1(path, seconds=6.0, seed=1)
    out = subprocess.run([EXE, path], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-2000:])
    rec = _records_line(out.stdout)
    assert rec["mono"] == st["frames"] == 160 and rec["features"] == 150 * 160
    odom = np.array([[float(x) for x in l.split()[1:]] for l in out.stdout.splitlines() if l.startswith("ODOM")])
    # the first image is dropped (IngvioFilter.cpp:257-261) and frames before the gravity initialisation (300 IMU samples) are ignored
    assert 115 <= len(odom) <= 160 and rec["frames_processed"] == len(odom)
    truth = {round(r[1], 6): np.array(__import__("struct").unpack("<7d", r[2])) for r in replay.read(path) if r[0] == replay.GROUND_TRUTH}
    N, clones = odom[:, 11], odom[:, 12]
    assert N.max() <= 21 + 6 * 12 and clones.max() <= 12 and clones[-1] >= 11          # the 11-pose window (+ the clone of the frame being processed)
    # the filter's world frame starts at the origin with the gravity-aligned attitude: compare displacements from the first pose
    p0_true = truth[round(odom[0, 0], 6)][:3]
    err = []
    for row in odom:
        tr = truth[round(row[0], 6)]
        err.append(np.linalg.norm((row[1:4] - odom[0, 1:4]) - (tr[:3] - p0_true)))
    print("config 1 replay: %d frames, final displacement error %.3f m, max %.3f m over %.1f m travelled" % (len(odom), err[-1], max(err), 2.0 * 4.0 + 0.4 * 5))
    assert max(err) < 0.6
    q = odom[-1, 4:8]
    assert abs(np.linalg.norm(q) - 1.0) < 1e-9
