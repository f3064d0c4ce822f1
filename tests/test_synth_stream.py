"""The C++ synthetic stream generator of the host shim (ingvio_amd/csrc/host/SynthStream.{h,cpp}; SURVEY.md 8d: SplitMix64,
seed = 0x1A6F10 + frame id) pinned three ways: an independent Python transcription of the published SplitMix64 + the stream's
sampling order, a committed golden frame (tests/golden/synth_frame.npz, written by `python tests/test_synth_stream.py --regen`
from the C++ tool's output), and the record counts of a written INGVIOR1 file.  No GPU needed (generation only)."""
import math
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TOOL = os.path.join(ROOT, "ingvio_amd", "lib", "ingvio_replay")
GOLDEN = os.path.join(ROOT, "tests", "golden", "synth_frame.npz")
M64 = (1 << 64) - 1
SPEC = dict(feats=150, clones=11, life=10, cohort=0, frames=40, outlier_every=20, pixel_noise=1e-3, seed=0x1A6F10)
GOLDEN_FRAME = 23


class SplitMix64:
    def __init__(self, seed):
        self.s = seed & M64

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & M64
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
        return z ^ (z >> 31)

    def uniform(self, a=0.0, b=1.0):
        return a + (b - a) * ((self.next() >> 11) * (1.0 / 9007199254740992.0))

    def normal(self):
        u1 = 1.0 - self.uniform()
        u2 = self.uniform()
        return math.sqrt(-2.0 * math.log(u1)) * math.cos(6.283185307179586476925286766559 * u2)


def sub(seed, k, tag, i):
    base = SplitMix64((seed + k) & M64)
    return SplitMix64((base.next() + (tag << 40) + i) & M64)


def truth(tau):
    if tau <= 0:
        th, thd = 0.0, 0.0
    elif tau <= 2.0:
        th, thd = 0.1 * tau * tau, 0.2 * tau
    else:
        th, thd = 0.4 + 0.4 * (tau - 2.0), 0.4
    c, s = math.cos(th), math.sin(th)
    return np.array([[c, 0.0, -s], [s, 0.0, c], [0.0, -1.0, 0.0]]), np.array([5 * c, 5 * s, 1.0])


def frame_py(spec, k):
    from ingvio_amd import synth
    F, life, seed = spec["feats"], spec["life"], spec["seed"]

    def cam(kk):
        R, p = truth(kk * 0.05)
        return R @ synth.R_CL2I, p + R @ synth.T_CL2I
    Rc, pc = cam(k)
    Rlr, tlr = synth.t_cl2cr()
    ids, uv = [], []
    for slot in range(F):
        phase = (life + 1 - spec.get("birth_frame", 3)) % life if spec["cohort"] else slot % life
        gen = (k - 1 + phase) // life
        birth = 1 - phase + gen * life
        r = sub(seed, birth, 2, slot)
        d = r.uniform(3.0, 15.0); x = r.uniform(-0.5, 0.5) * d; y = r.uniform(-0.4, 0.4) * d
        Rb, pb = cam(birth)
        pw = pb + Rb @ np.array([x, y, d])
        q = Rc.T @ (pw - pc)
        qr = Rlr @ q + tlr
        r = sub(seed, k, 3, slot)
        n = [r.normal() for _ in range(4)]
        fid = gen * F + slot + 1
        u0 = q[0] / q[2] + spec["pixel_noise"] * n[0]
        if spec["outlier_every"] > 0 and (fid - 1) % spec["outlier_every"] == 0 and k - birth == life // 2:
            u0 += 0.5
        ids.append(fid)
        uv.append([u0, q[1] / q[2] + spec["pixel_noise"] * n[1], qr[0] / qr[2] + spec["pixel_noise"] * n[2],
                   qr[1] / qr[2] + spec["pixel_noise"] * n[3]])
    return np.array(ids, dtype=np.uint64), np.array(uv)


def spec_str(spec):
    return ",".join("%s=%s" % (k, v) for k, v in spec.items())


def frame_cpp(spec, k):
    out = subprocess.run([TOOL, "--synth", spec_str(spec), "--frame", str(k)], capture_output=True, text=True, check=True).stdout
    rows = [l.split()[1:] for l in out.splitlines() if l.startswith("FEAT")]
    stamp = int([l for l in out.splitlines() if l.startswith("STAMP")][0].split()[1])
    return np.array([int(r[0]) for r in rows], dtype=np.uint64), np.array([[float(x) for x in r[1:]] for r in rows]), stamp


needs_tool = pytest.mark.skipif(not os.path.exists(TOOL), reason="ingvio_replay not built")


def test_splitmix64_known_answers():
    """The published test vector of SplitMix64 (seed 1234567: Vigna's reference implementation, first outputs)."""
    r = SplitMix64(1234567)
    assert [r.next() for _ in range(3)] == [6457827717110365317, 3203168211198807973, 9817491932198370423]


@needs_tool
@pytest.mark.parametrize("cohort,k", [(0, 1), (0, 7), (0, 23), (1, 1), (1, 2), (1, 11), (1, 12), (1, 31)])
def test_cpp_generator_matches_python_transcription(cohort, k):
    spec = dict(SPEC, cohort=cohort)
    ids_c, uv_c, stamp = frame_cpp(spec, k)
    ids_p, uv_p = frame_py(spec, k)
    assert stamp == 2_000_000_000 + k * 50_000_000
    assert np.array_equal(ids_c, ids_p)                               # track ids: bit-exact
    assert np.allclose(uv_c, uv_p, rtol=0, atol=1e-13)                # libm vs Python's math: last-ulp differences only
    assert len(set(ids_c.tolist())) == spec["feats"]


@needs_tool
def test_golden_frame():
    g = np.load(GOLDEN)
    ids_c, uv_c, stamp = frame_cpp(SPEC, GOLDEN_FRAME)
    assert np.array_equal(ids_c, g["ids"]) and stamp == int(g["stamp"])
    assert np.array_equal(uv_c, g["uv"])                              # the tool regenerates its own committed frame bit for bit
    # the outlier tracks of this frame carry +0.5 in u0 and nothing else does
    ids_p, uv_p = frame_py(dict(SPEC, outlier_every=0), GOLDEN_FRAME)
    jump = np.abs(uv_c[:, 0] - uv_p[:, 0]) > 0.4
    life = SPEC["life"]
    expect = np.array([(int(i) - 1) % 20 == 0 and (GOLDEN_FRAME - (1 - (s % life) + ((GOLDEN_FRAME - 1 + s % life) // life) * life)) == life // 2
                       for s, i in enumerate(ids_c)])
    assert np.array_equal(jump, expect)


@needs_tool
def test_written_recording_counts(tmp_path):
    from ingvio_amd import replay
    path = str(tmp_path / "s.rec")
    spec = dict(SPEC, frames=12, feats=20)
    subprocess.run([TOOL, "--synth", spec_str(spec), "--write", path], check=True)
    recs = replay.read(path)
    kinds = [r[0] for r in recs]
    # 12 frames + the header-only frame at t = 0 that raises _hasImageCome before the static phase (IngvioFilter.cpp:257-261, :393)
    assert kinds.count(replay.PARAMS) == 1 and kinds.count(replay.IMU) == 400 + 120 and kinds.count(replay.STEREO_FRAME) == 13
    assert recs[1][0] == replay.STEREO_FRAME and recs[1][1] == 0.0 and len(recs[1][2]) == 4
    assert kinds.count(replay.GROUND_TRUTH) == 12
    text = recs[0][2].decode()
    assert "max_sliding_window_poses: 11" in text and "cam_nums: 2" in text
    # every camera frame is preceded by the IMU sample of the same stamp (the filter propagates up to the image time)
    for i, r in enumerate(recs):
        if r[0] == replay.STEREO_FRAME and i > 1:
            assert recs[i - 1][0] == replay.IMU and abs(recs[i - 1][1] - r[1]) < 1e-9


@needs_tool
@pytest.mark.gpu
def test_stream_through_the_callbacks_on_the_device():
    """The cohort stream of bench.py's latency line played into IngvioFilter's callbacks (one filter, key-frame mode, RemoveLost cap
    lifted): the frames on which a cohort is lost carry a RemoveLost update over most of its tracks, the filter follows the circle,
    and the batched triangulation hands the update what the per-feature one would (same accept counts with either)."""
    spec = "feats=150,clones=11,life=10,cohort=1,birth_frame=3,frames=55,key=1"
    sets = ["--set", "hip_max_valid_ids: 0", "--set", "hip_compress_rule: 1"]
    r = subprocess.run([TOOL, "--synth", spec, "--time"] + sets, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    frames = [l.split() for l in r.stdout.splitlines() if l.startswith("FRAME")]
    lat = dict(x.split("=") for x in [l for l in r.stdout.splitlines() if l.startswith("LATENCY")][0].split()[1:])
    heavy = [(int(f[1]), int(f[3]), int(f[4])) for f in frames if int(f[4]) >= 50]      # (frame, rows, accepted)
    print("heavy frames", heavy, "latency", lat)
    assert len(heavy) >= 3 and all(k % 10 == 3 for k, _, _ in heavy)                    # cohorts born at 3, 13, ... are lost at 13, 23, ...
    assert all(rows == 6 * 11 for _, rows, _ in heavy)                                  # top-n compression: n = 6 x window clones rows
    assert int(lat["heavy_frames"]) >= 2 and float(lat["heavy_median_ms"]) > float(lat["other_median_ms"]) > 0.0
    assert float(lat["final_pos_err_m"]) < 0.25                                         # ~4 m travelled
    clones = [int(f[7]) for f in frames[12:]]
    assert set(clones) == {9, 10}                                                       # 11 -> two key frames marginalised -> 9, then 10
    # as written (cap 20, all rows kept): at most 20 features per RemoveLost update
    r2 = subprocess.run([TOOL, "--synth", spec, "--time"], capture_output=True, text=True, timeout=600)
    assert r2.returncode == 0, r2.stderr[-2000:]
    acc2 = [int(l.split()[4]) for l in r2.stdout.splitlines() if l.startswith("FRAME")]
    assert max(acc2) == 20


if __name__ == "__main__" and "--regen" in sys.argv:
    ids, uv, stamp = frame_cpp(SPEC, GOLDEN_FRAME)
    np.savez(GOLDEN, ids=ids, uv=uv, stamp=np.int64(stamp), spec=spec_str(SPEC), frame=GOLDEN_FRAME)
    print("wrote", GOLDEN, ids.shape, uv.shape)
