"""Stream-level parity of the drop-in path: the C++ shim's callbacks on the device against the golden streams.

tests/golden/stream_*.npz hold, for every processed camera frame of fifteen synthetic streams (key-frame and sliding-window mode, 11
and 21 clones, the RemoveLost cap as written and lifted, stereo and mono, raw GNSS epochs; round 6: the 27- / 35-pose windows and
parameter values the reference ships, in-state SLAM landmarks, a stream in which the filter aligns itself), what an INDEPENDENT Python transcription of the reference's policy layer
(oracle/stream_filter.py: IngvioFilter::callbackStereoFrame, RemoveLost / SwMarg / Keyframe selection, anchor change, observation
cleaning, eraseInvalidFeatures, marginalisation) decided while driving the CPU oracle, and the state it ended the frame with.
`ingvio_replay --synth <spec> --trace` plays the same SplitMix64 stream through IngvioFilter on the device.  Compared per frame:
feature ids of every update, accept masks, selected / marginalised stamps, erased ids, the (idx, size) table, window stamps and
map-server ids bit-exact; the nominal state to 1e-9; diag(P) and |P|_F to 1e-6 relative (BASELINE's covariance tolerance)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TOOL = os.path.join(ROOT, "ingvio_amd", "lib", "ingvio_replay")
GOLDEN = os.path.join(ROOT, "tests", "golden")
STREAMS = ["kf11", "kf11_lifted", "sw11", "kf21", "sw21", "kf11_mono", "sw11_mono",      # mono: the MONO callback (BASELINE configs[0])
           "sw11_gnss", "kf11_gnss",                                                       # raw GNSS epochs in the callback (BASELINE configs[2])
           "sw11_lm", "kf11_lm", "sw11_lm_mono",      # in-state SLAM landmarks: delayed initialisation, landmark update, anchor change (SURVEY 8f row f-2 as a stream)
           "kf27", "kf35_mono"]      # the windows and parameter values the reference SHIPS (config/sportsfield/ingvio_stereo.yaml / ingvio_mono.yaml)
SHIPPED = ("kf27", "kf35_mono")      # visual_noise 0.18: the synthetic +0.5 outliers mostly pass the gate there - pinned as they come
POSE_TOL = 1e-9
COV_TOL = 1e-6
LM_TOL = 1e-7        # world positions of in-state landmarks (2 - 20 m away): they start from the triangulator's iterate, which stops at
                     # conv_precision 5e-7 on the cost (Triangulator.cpp:262-270) - two FP64 evaluations of it agree to ~1e-8 m

INT_TAGS = {"LOST_IDS": "lost_ids", "LOST_ACC": "lost_acc", "LOST_DIRECT": "lost_direct", "SEL_IDS": "sel_ids", "SEL_ACC": "sel_acc",
            "CLEAN_ERASED": "clean_erased", "ANCHOR_ERASED": "anchor_erased", "ANCHOR_MOVED": "anchor_moved", "INVALID_ERASED": "invalid_erased",
            "MAP_IDS": "map_ids", "TABLE": "table", "GNSS_KEEP": "gnss_keep",
            "LM_UPD_IDS": "lm_upd_ids", "LM_UPD_ACC": "lm_upd_acc", "LM_INIT_IDS": "lm_init_ids", "LM_IDS": "lm_ids", "LM_MARG_IDS": "lm_marg_ids"}
F64_TAGS = {"SEL_STAMPS": "sel_stamps", "MARG_STAMPS": "marg_stamps", "SW_STAMPS": "sw_stamps", "POSE": "pose", "DIAG": "diag",
            "GNSS_VALS": "gnss_vals", "LM_VALS": "lm_vals", "ALIGN": "align"}
SCALAR_TAGS = {"LOST_ROWS": "lost_rows", "SEL_ROWS": "sel_rows", "NORM": "norm", "GNSS_ROWS": "gnss_rows", "GNSS_ADDED_TOTAL": "gnss_added_total"}


def parse_trace(text):
    """stdout of `ingvio_replay --trace` -> list of per-frame dicts (keys as in oracle/stream_filter.py's trace)."""
    frames, cur = [], None
    for line in text.splitlines():
        tok = line.split()
        if not tok:
            continue
        if tok[0] == "TRACE":
            cur = dict(k=int(tok[1]), stamp=float(tok[2]))
        elif cur is None:
            continue
        elif tok[0] == "END":
            cur["table"] = cur["table"].reshape(-1, 2)
            cur["n"] = len(cur["diag"])
            frames.append(cur)
            cur = None
        elif tok[0] in INT_TAGS:
            cur[INT_TAGS[tok[0]]] = np.array([int(x) for x in tok[1:]], dtype=np.int64)
        elif tok[0] in F64_TAGS:
            cur[F64_TAGS[tok[0]]] = np.array([float(x) for x in tok[1:]], dtype=np.float64)
        elif tok[0] in SCALAR_TAGS:
            cur[SCALAR_TAGS[tok[0]]] = float(tok[1])
    return frames


def compare(gold, got, pose_tol=POSE_TOL, cov_tol=COV_TOL):
    """-> (list of mismatch strings, dict of the largest deviations seen)."""
    bad = []
    worst = dict(pose=0.0, diag=0.0, norm=0.0, gnss=0.0)
    added_total = 0
    if len(gold) != len(got):
        bad.append("frames: golden %d, shim %d" % (len(gold), len(got)))
    for f, (g, s) in enumerate(zip(gold, got)):
        tag = "frame %d (t = %.2f)" % (f, g["stamp"])
        if g["stamp"] != s["stamp"]:
            bad.append("%s: stamp %r vs %r" % (tag, g["stamp"], s["stamp"]))
            break
        for k in ("lost_ids", "lost_acc", "lost_direct", "sel_ids", "sel_acc", "anchor_erased", "anchor_moved", "invalid_erased", "map_ids"):
            if not np.array_equal(np.asarray(g[k], dtype=np.int64), s[k]):
                bad.append("%s: %s differs: golden %s... shim %s..." % (tag, k, np.asarray(g[k])[:12].tolist(), s[k][:12].tolist()))
        # the reference pushes an id once per cleaned stamp (KeyframeUpdate.cpp:749-757): compare the erased SETS
        if sorted(set(np.asarray(g["clean_erased"]).tolist())) != sorted(set(s["clean_erased"].tolist())):
            bad.append("%s: clean_erased differs" % tag)
        for k in ("sel_stamps", "marg_stamps", "sw_stamps"):
            if not np.array_equal(np.asarray(g[k], dtype=np.float64), s[k]):
                bad.append("%s: %s differs: golden %s shim %s" % (tag, k, np.asarray(g[k]).tolist(), s[k].tolist()))
        if not np.array_equal(np.asarray(g["table"], dtype=np.int64).reshape(-1, 2), s["table"]):
            bad.append("%s: (idx, size) table differs" % tag)
        # the alignment the GNSS block works with (given, or found by GvioAligner::batchAlign on the stream's raw epochs): found in the
        # same frame, yaw to 1e-8 rad, anchor to 1e-4 m (the Gauss-Newton iterations stop at gv_align_conv_epsilon = 1e-5)
        if "align" in s and len(g.get("align", [])) == 5:
            ga, sa = np.asarray(g["align"]), s["align"]
            if ga[0] != sa[0]:
                bad.append("%s: aligned %s vs %s" % (tag, ga[0], sa[0]))
            elif ga[0]:
                worst["yaw"] = max(worst.get("yaw", 0.0), abs(ga[1] - sa[1])); worst["anchor"] = max(worst.get("anchor", 0.0), float(np.linalg.norm(ga[2:] - sa[2:])))
                if abs(ga[1] - sa[1]) > 1e-8 or np.linalg.norm(ga[2:] - sa[2:]) > 1e-4:
                    bad.append("%s: alignment differs: yaw %.3e rad, anchor %.3e m" % (tag, abs(ga[1] - sa[1]), np.linalg.norm(ga[2:] - sa[2:])))
        # in-state SLAM landmarks: which ones the update evaluated and accepted, which ones the delayed initialisation added, which ones
        # are in the state after the frame and which ones left it
        for k in ("lm_upd_ids", "lm_upd_acc", "lm_init_ids", "lm_ids", "lm_marg_ids"):
            if k in s and not np.array_equal(np.asarray(g[k], dtype=np.int64), s[k]):
                bad.append("%s: %s differs: golden %s shim %s" % (tag, k, np.asarray(g[k]).tolist(), s[k].tolist()))
        # the GNSS block: rows handed to ekfUpdate, which candidate rows passed their gates, how many variables the delayed
        # initialisations have added so far, which GNSS scalars exist
        added_total += len(g["gnss_added"])
        if "gnss_rows" in s:
            if int(g["gnss_rows"]) != int(s["gnss_rows"]) or not np.array_equal(np.asarray(g["gnss_keep"], dtype=np.int64), s["gnss_keep"]):
                bad.append("%s: GNSS rows %d vs %d, keep %s vs %s" % (tag, int(g["gnss_rows"]), int(s["gnss_rows"]), np.asarray(g["gnss_keep"]).tolist(), s["gnss_keep"].tolist()))
            if added_total != int(s["gnss_added_total"]):
                bad.append("%s: GNSS variables added so far %d vs %d" % (tag, added_total, int(s["gnss_added_total"])))
            if not np.array_equal(np.isnan(np.asarray(g["gnss_vals"])), np.isnan(s["gnss_vals"])):
                bad.append("%s: GNSS scalars in the state differ" % tag)
        # rows handed to the Kalman update: the reference (and the oracle) count the stacked rows it keeps, the device works in
        # information form and always reports n = 6 x window clones (include/ingvio_hip.h, ingvio_msckf_opts::compress_rule) - the
        # same posterior; what must agree is WHETHER an update took place
        for k in ("lost_rows", "sel_rows"):
            if (int(g[k]) > 0) != (int(s[k]) > 0):
                bad.append("%s: %s %d vs %d" % (tag, k, int(g[k]), int(s[k])))
        if bad:
            break                                                          # everything after the first structural difference is noise
        dp = float(np.max(np.abs(np.asarray(g["pose"]) - s["pose"])))
        dd = float(np.max(np.abs(np.asarray(g["diag"]) - s["diag"]) / np.maximum(np.abs(np.asarray(g["diag"])), 1e-300)))
        dn = abs(float(g["norm"]) - s["norm"]) / float(g["norm"])
        worst["pose"] = max(worst["pose"], dp); worst["diag"] = max(worst["diag"], dd); worst["norm"] = max(worst["norm"], dn)
        if "gnss_vals" in s and np.any(~np.isnan(s["gnss_vals"])):
            gv, sv = np.asarray(g["gnss_vals"]), s["gnss_vals"]
            ok = ~np.isnan(gv)
            dg = float(np.max(np.abs(gv[ok] - sv[ok]) / np.maximum(np.abs(gv[ok]), 1.0)))
            worst["gnss"] = max(worst["gnss"], dg)
            if dg > pose_tol:
                bad.append("%s: GNSS scalars off by %.3e" % (tag, dg))
        if "lm_vals" in s and len(s["lm_vals"]):
            dl = float(np.max(np.abs(np.asarray(g["lm_vals"]) - s["lm_vals"])))
            worst["lm"] = max(worst.get("lm", 0.0), dl)
            if dl > LM_TOL:
                bad.append("%s: landmark positions off by %.3e" % (tag, dl))
        if dp > pose_tol:
            bad.append("%s: nominal state off by %.3e" % (tag, dp))
        if dd > cov_tol or dn > cov_tol:
            bad.append("%s: covariance off (diag %.3e, norm %.3e relative)" % (tag, dd, dn))
        if bad:
            break
    return bad, worst


def load_golden(name):
    from oracle import stream_filter as sf
    z = np.load(os.path.join(GOLDEN, "stream_%s.npz" % name))
    return sf.unpack_traces(z), str(z["spec"]), str(z["overrides"])


def run_shim(spec, overrides, extra=(), raw_gnss=False, tmp=None):
    sets = []
    for line in list(overrides.splitlines()) + list(extra):
        if line.strip():
            sets += ["--set", line]
    src = ["--synth", spec]
    if raw_gnss:                                                           # the recording with raw GNSS epochs, rebuilt here (deterministic), played as a FILE
        from oracle import gen_stream_golden as gen
        rec = os.path.join(tmp, "raw_gnss.ingvior")
        gen.write_recording(spec, rec, raw_gnss=True)
        src = [rec]
    r = subprocess.run([TOOL] + src + ["--trace"] + sets, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return parse_trace(r.stdout)


needs_tool = pytest.mark.skipif(not os.path.exists(TOOL), reason="ingvio_replay not built")


# ---- CPU: the golden files themselves ---------------------------------------------------------------------------------------
def test_golden_streams_cover_the_policies():
    """Every stream exercises what it was made for: RemoveLost updates with rejected features, selected-stamp updates, anchor
    changes (moved and erased), cleaning, both marginalisation rules; the as-written stream hits the cap of 20."""
    seen = {}
    for name in STREAMS:
        tr, spec, ov = load_golden(name)
        assert len(tr) >= 60
        if name in SHIPPED:
            assert ("frame_select_interval: %d" % (18 if name == "kf27" else 28)) in ov and "visual_noise: 0.18" in ov
        lost = sum(len(t["lost_ids"]) for t in tr); lost_acc = sum(int(np.sum(t["lost_acc"])) for t in tr)
        sel = sum(len(t["sel_ids"]) for t in tr); sel_acc = sum(int(np.sum(t["sel_acc"])) for t in tr)
        seen[name] = dict(lost=lost, lost_acc=lost_acc, sel=sel, sel_acc=sel_acc, moved=sum(len(t["anchor_moved"]) for t in tr),
                          erased=sum(len(t["anchor_erased"]) for t in tr), direct=sum(len(t["lost_direct"]) for t in tr),
                          margs=sum(len(t["marg_stamps"]) for t in tr))
        assert lost > 0 and sel > 0 and sel_acc > 0 and seen[name]["margs"] > 0 and seen[name]["direct"] > 0, (name, seen[name])
        assert lost_acc < lost or sel_acc < sel or name in SHIPPED, name  # the chi^2 gate (or the cap) refused something
        key = "key=1" in spec
        per_frame = {len(t["marg_stamps"]) for t in tr}
        assert per_frame == ({0, 2} if key else {0, 1}), (name, per_frame)
        n = np.array([t["n"] for t in tr])
        clones = int(spec.split("clones=")[1].split(",")[0])
        n_gnss = 5 if "gnss=1" in spec else 0                              # YOF, FS and three clock biases
        n_lm = int(ov.split("max_landmark_features:")[1].split()[0]) if "max_landmark_features" in ov else 0
        assert n.max() == 21 + n_gnss + 3 * n_lm + 6 * (clones - (1 if key else 0)), (name, n.max())       # state size after the frame's marginalisation
    assert max(int(np.sum(t["lost_acc"])) for t in load_golden("kf11")[0]) == 20          # RemoveLostUpdate.h:38
    assert max(int(np.sum(t["lost_acc"])) for t in load_golden("kf11_lifted")[0]) > 60
    assert seen["kf11"]["moved"] > 0 and seen["kf21"]["erased"] > 0 and seen["sw11"]["moved"] > 0
    tr = load_golden("sw11_lm")[0]                                        # the landmark life cycle happened: initialised, updated every frame, lost / re-anchored, full state
    assert sum(len(t["lm_init_ids"]) for t in tr) >= 20 and sum(len(t["lm_marg_ids"]) for t in tr) >= 15
    assert max(len(t["lm_ids"]) for t in tr) == 6 and sum(int(np.sum(t["lm_upd_acc"])) for t in tr) > 150
    for name in ("sw11_gnss", "kf11_gnss"):                               # the GNSS block did its three things
        tr = load_golden(name)[0]
        assert tr[0]["gnss_added"].tolist() == [4, 0, 2, 3]              # FS, GPS, GAL, BDS by delayed initialisation at the first aligned epoch
        assert max(t["gnss_rows"] for t in tr) == 16 and any(0 in t["gnss_keep"].tolist() for t in tr)      # the outlier's row refused by its gate
        assert [v[0] for v in tr[-1]["table"].tolist()[4:9]] == [21, 22, 23, 24, 25]      # YOF, FS, GPS, GAL, BDS right after the extrinsics
        assert not np.isnan(tr[-1]["gnss_vals"][[0, 2, 3, 4, 5]]).any() and np.isnan(tr[-1]["gnss_vals"][1])


@needs_tool
def test_python_policy_replays_its_own_golden(tmp_path):
    """oracle/stream_filter.py on a freshly written recording reproduces the committed file (first 30 frames of two streams): the
    fixture is what the generator says it is, and the C oracle underneath is deterministic."""
    from oracle import gen_stream_golden as gen, stream_filter as sf
    for name in ("kf11", "sw11", "sw11_gnss"):
        gold, spec, ov = load_golden(name)
        rec = str(tmp_path / (name + ".ingvior"))
        gen.write_recording(spec, rec)
        tr = sf.play_recording(rec, ov, max_frames=30)
        for g, t in zip(gold, tr):
            for k in sf.LIST_KEYS_INT:
                assert np.array_equal(np.asarray(g[k]), np.asarray(t[k], dtype=np.int64)), (name, k)
            assert g["stamp"] == t["stamp"] and np.array_equal(g["sel_stamps"], np.asarray(t["sel_stamps"], dtype=float))
            assert np.array_equal(g["pose"], t["pose"]) and np.array_equal(g["diag"], t["diag"])
            assert np.array_equal(g["gnss_vals"], t["gnss_vals"], equal_nan=True) and g["gnss_rows"] == t["gnss_rows"]


@needs_tool
def test_a_wrong_selection_rule_is_caught_by_the_comparison(tmp_path):
    """The comparison has teeth: the SAME Python filter with frame_select_interval 4 instead of 5 (the deliberately wrong policy
    VERDICT r04 asked to turn the test red) differs from the golden stream in the selected stamps of the first marginalising frame."""
    from oracle import gen_stream_golden as gen, stream_filter as sf
    gold, spec, ov = load_golden("sw11")
    rec = str(tmp_path / "sw11.ingvior")
    gen.write_recording(spec, rec)
    tr = sf.play_recording(rec, ov + "frame_select_interval: 4\n", max_frames=20)
    got = []
    for t in tr:
        s = {k: np.asarray(t[k], dtype=np.int64) for k in sf.LIST_KEYS_INT}
        s.update({k: np.asarray(t[k], dtype=np.float64) for k in sf.LIST_KEYS_F64})
        s.update(stamp=t["stamp"], table=np.asarray(t["table"], dtype=np.int64).reshape(-1, 2), pose=t["pose"], lost_rows=t["lost_rows"],
                 sel_rows=t["sel_rows"], norm=t["norm"], n=t["n"])
        got.append(s)
    bad, _ = compare(gold[:20], got)
    assert bad and "sel_stamps" in " ".join(bad), bad


# ---- GPU: the shim against the golden streams ---------------------------------------------------------------------------------
@needs_tool
@pytest.mark.gpu
def test_shim_aligns_itself_as_the_golden_stream_does(tmp_path):
    """sw11_gnss_align: no ALIGNMENT record, a raw GNSS epoch (ephemerides + observations) in front of every GNSS_MEAS record.  The filter
    buffers gv_align_batch_size epochs, drops the first batch (the camera is still accelerating: horizontal velocity excitation below
    gv_align_vel_thres, GvioAligner.cpp:104-124), aligns on the second - GvioAligner::batchAlign on the device's satellite geodesy
    (host/GvioAligner.cpp over ingvio_gnss_sat_eval) against oracle/gvio_align.py over the C oracle's: aligned in the same frame, then
    checkYofStatus, the delayed initialisations of the clock states and the GNSS updates with the alignment each side FOUND (measured:
    yaw 4e-13 rad, anchor 7e-9 m apart; nominal state 7e-10 - inside the tolerance of every other stream)."""
    gold, spec, ov = load_golden("sw11_gnss_align")
    got = run_shim(spec, ov, raw_gnss=True, tmp=str(tmp_path))
    bad, worst = compare(gold, got)
    print("self-aligning stream: %d frames, yaw %.2e rad, anchor %.2e m, nominal state %.2e, diag(P) %.2e, GNSS scalars %.2e"
          % (len(got), worst.get("yaw", -1), worst.get("anchor", -1), worst["pose"], worst["diag"], worst["gnss"]))
    assert not bad, "\n".join(bad[:10])
    first = min(f for f, t in enumerate(gold) if t["align"][0])
    assert 20 < first < 40 and abs(gold[first]["align"][1] - 0.3) < 0.02                       # the generating yaw offset (SynthStream.cpp:100)
    assert sum(int(t["gnss_rows"]) for t in gold[first + 1:]) > 300                            # and GNSS updates from then on


@needs_tool
@pytest.mark.gpu
def test_self_aligning_stream_with_another_batch_size_goes_red(tmp_path):
    gold, spec, ov = load_golden("sw11_gnss_align")
    got = run_shim(spec, ov, extra=["gv_align_batch_size: 10"], raw_gnss=True, tmp=str(tmp_path))
    bad, _ = compare(gold, got)
    assert bad and "align" in " ".join(bad), bad[:3]


@needs_tool
@pytest.mark.gpu
@pytest.mark.parametrize("name", STREAMS)
def test_shim_stream_matches_golden(name):
    gold, spec, ov = load_golden(name)
    got = run_shim(spec, ov)
    bad, worst = compare(gold, got)
    print("stream %s: %d frames, largest deviation: nominal state %.2e, diag(P) %.2e rel, |P|_F %.2e rel, GNSS scalars %.2e rel, landmarks %.2e"
          % (name, len(got), worst["pose"], worst["diag"], worst["norm"], worst["gnss"], worst.get("lm", 0.0)))
    assert not bad, "\n".join(bad[:10])


@needs_tool
@pytest.mark.gpu
def test_shim_with_a_wrong_frame_select_interval_goes_red():
    """The shim with frame_select_interval 4 against the golden stream made with 5: the comparison must fail, at the selected stamps."""
    gold, spec, ov = load_golden("sw11")
    got = run_shim(spec, ov, extra=["frame_select_interval: 4"])
    bad, _ = compare(gold, got)
    assert bad and "sel_stamps" in " ".join(bad), bad[:3]


@needs_tool
@pytest.mark.gpu
def test_shim_with_another_visual_noise_goes_red_on_the_shipped_stereo_stream():
    """kf27 (sportsfield stereo values) played with the synthetic default visual_noise 0.08 instead of the shipped 0.18: same policy
    decisions at first, another posterior - the comparison must fail on the state or the covariance."""
    gold, spec, ov = load_golden("kf27")
    got = run_shim(spec, ov, extra=["visual_noise: 0.08"])
    bad, _ = compare(gold, got)
    assert bad, "the comparison did not notice a different measurement noise"


@needs_tool
@pytest.mark.gpu
def test_shim_in_sliding_window_mode_goes_red_on_the_shipped_mono_stream():
    """kf35_mono is key-frame mode as shipped (is_key_frame: 1, two clones chosen by KeyframeUpdate::getMargKfs every other frame); the
    shim switched to sliding-window mode marginalises the oldest clone every frame instead: red at the marginalised stamps."""
    gold, spec, ov = load_golden("kf35_mono")
    got = run_shim(spec, ov, extra=["is_key_frame: 0"])
    bad, _ = compare(gold, got)
    assert bad and any(k in " ".join(bad) for k in ("marg_stamps", "sel_stamps", "sel_ids", "table")), bad[:3]


@needs_tool
@pytest.mark.gpu
def test_shim_with_fewer_landmark_slots_goes_red():
    """sw11_lm was made with max_landmark_features 6; the shim with 4 slots initialises fewer landmarks: red at the ids in the state."""
    gold, spec, ov = load_golden("sw11_lm")
    got = run_shim(spec, ov, extra=["max_landmark_features: 4"])
    bad, _ = compare(gold, got)
    assert bad and any(k in " ".join(bad) for k in ("lm_init_ids", "lm_ids", "table")), bad[:3]


@needs_tool
@pytest.mark.gpu
def test_shim_without_the_gnss_row_gate_goes_red():
    """The shim with gnss_chi2_test off against the golden stream made with it on: the outlier pseudo-range (from frame 8 on) is
    fused instead of refused, and the comparison must say so at the GNSS rows."""
    gold, spec, ov = load_golden("sw11_gnss")
    got = run_shim(spec, ov, extra=["gnss_chi2_test: 0"])
    bad, _ = compare(gold, got)
    assert bad and "GNSS rows" in " ".join(bad), bad[:3]


@needs_tool
@pytest.mark.gpu
@pytest.mark.parametrize("name", ["kf11", "sw11_mono"])
def test_shim_stream_with_two_call_remove_lost_matches_golden(name):
    """The golden streams run with RemoveLost's triangulation and update in one device call (hip_fuse_triangulation, the default);
    the two-call form (ingvio_triangulate, host selection, ingvio_msckf_update) stays correct as well."""
    gold, spec, ov = load_golden(name)
    got = run_shim(spec, ov, extra=["hip_fuse_triangulation: 0"])
    bad, _ = compare(gold, got)
    assert not bad, "\n".join(bad[:10])
