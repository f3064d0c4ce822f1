"""Golden STREAMS for the policy layer of the camera callback: tests/golden/stream_*.npz.

TEST INFRASTRUCTURE.  For every stream below the C++ generator (ingvio_replay --synth ... --write: the SplitMix64 stream of
ingvio_amd/csrc/host/SynthStream.cpp, no device needed) writes the INGVIOR1 recording, and oracle/stream_filter.py — the
independent Python transcription of IngvioFilter::callbackStereoFrame and the classes under it, driving the CPU oracle — plays
it.  Committed per processed frame: the feature ids each update consumed, their accept masks, the selected / marginalised clone
stamps, what the anchor change / cleaning / eraseInvalidFeatures removed, the (idx, size) table, the window stamps, the map
server's ids, the nominal state (R, p, v, bg, ba, extrinsics), diag(P) and |P|_F.

    python -m oracle.gen_stream_golden            # regenerates every stream (about a minute of CPU)

tests/test_stream_golden.py replays the same specs through the C++ shim on the device (ingvio_replay --synth ... --trace) and
compares frame by frame; a CPU test re-plays the first frames through stream_filter and checks the committed file."""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TOOL = os.path.join(ROOT, "ingvio_amd", "lib", "ingvio_replay")
GOLDEN = os.path.join(ROOT, "tests", "golden")

# name -> (synth spec, parameter overrides as "key: value" lines)
STREAMS = {
    # key-frame mode, the 11-pose window of BASELINE config 2, cohort tracks (every 10th frame loses 150 tracks at once), as written:
    # RemoveLost accepts at most 20 features (Q3) and keeps every row after the rotation (Q2)
    "kf11": ("feats=150,clones=11,life=10,cohort=1,birth_frame=3,frames=60,key=1", ""),
    # the same with the cap lifted and top-n compression: the heavy frames of bench.py's latency line
    "kf11_lifted": ("feats=150,clones=11,life=10,cohort=1,birth_frame=3,frames=60,key=1", "hip_max_valid_ids: 0\nhip_compress_rule: 1\n"),
    # sliding-window mode (is_key_frame = 0): oldest clone marginalised every frame, selectSwTimestamps with interval 5 -> 3 stamps;
    # staggered track deaths (15 per frame)
    "sw11": ("feats=150,clones=11,life=13,cohort=0,frames=60,key=0", "frame_select_interval: 5\n"),
    # key-frame mode at the window of the reference's shipped stereo config (config/fw_zed2i_f9p/ingvio_stereo.yaml: 21 poses)
    "kf21": ("feats=100,clones=21,life=19,cohort=0,frames=70,key=1,outlier_every=7", ""),
    # sliding-window mode, 21 poses, interval 6 -> 4 stamps
    # BASELINE configs[0] is MONO (sports-field mono, max_pts_frame 150, 11 poses): the mono callback (IngvioFilter.cpp:124-250), min 4
    # observations per RemoveLost feature (RemoveLostUpdate.cpp:51), 2 rows per observation; longer tracks (a mono point needs parallax)
    "kf11_mono": ("feats=150,clones=11,life=10,cohort=0,frames=70,key=1,stereo=0,outlier_every=6", ""),
    "sw11_mono": ("feats=150,clones=11,life=13,cohort=0,frames=70,key=0,stereo=0,outlier_every=6", "frame_select_interval: 5\n"),
    # BASELINE configs[2]: stereo + raw GNSS.  Every frame carries an epoch of 8 satellites (GPS x4, BDS x2, GAL x2: pseudo-range +
    # Doppler, satellite 5 with an 80 m outlier from frame 8 on) and its SPP fix; the alignment is given.  Pins the GNSS block of the
    # callback (IngvioFilter.cpp:329-362): epoch matching, checkYofStatus, updateTrackedSys with gnss_chi2_test, addNewTrackedSys'
    # delayed initialisations - i.e. where the GNSS scalars sit in the (idx, size) table
    "sw11_gnss": ("feats=150,clones=11,life=13,cohort=0,frames=60,key=0,gnss=1", "frame_select_interval: 5\ngnss_chi2_test: 1\ngnss_strong_reject: 1\n"),
    "kf11_gnss": ("feats=150,clones=11,life=10,cohort=0,frames=60,key=1,gnss=1", "gnss_chi2_test: 1\ngnss_strong_reject: 0\n"),
    # the reference's SHIPPED configurations (VERDICT r05 missing 3): config/sportsfield/ingvio_stereo.yaml - key-frame mode, 27 poses,
    # frame_select_interval 18, visual_noise 0.18, trans_thres 0.25, the triangulator's sportsfield values - and ingvio_mono.yaml -
    # 35 poses, interval 28, conv_precision 5e-8, max_depth 60 (MONO callback, two rows per observation)
    "kf27": ("feats=100,clones=27,life=25,cohort=0,frames=80,key=1,outlier_every=7",
             "frame_select_interval: 18\nvisual_noise: 0.18\ntrans_thres: 0.25\nconv_precision: 5e-07\nmax_depth: 40.0\nmin_depth: 0.2\n"
             "max_baseline_ratio: 80.0\nhuber_epsilon: 0.01\ninit_damping: 1e-03\n"),
    "kf35_mono": ("feats=100,clones=35,life=33,cohort=0,frames=90,key=1,stereo=0,outlier_every=6",
                  "frame_select_interval: 28\nvisual_noise: 0.18\ntrans_thres: 0.25\nconv_precision: 5e-08\nmax_depth: 60.0\nmin_depth: 0.2\n"
                  "max_baseline_ratio: 80.0\nhuber_epsilon: 0.01\ninit_damping: 1e-03\n"),
    # in-state SLAM landmarks (round 6; VERDICT r05 missing 3): max_landmark_features 6, sliding-window mode, tracks that outlive the window
    # (a landmark is initialised from a track with >= max_sliding_window_poses observations, LandmarkUpdate.cpp:892-917) - delayed
    # initialisation, the per-frame landmark update with its chi^2 gates, anchor change before the oldest clone goes, landmarks that
    # lose track and leave the state
    "sw11_lm": ("feats=100,clones=11,life=20,cohort=0,frames=70,key=0,outlier_every=9", "frame_select_interval: 5\nmax_landmark_features: 6\n"),
    # the same in key-frame mode (two clones chosen by getMargKfs leave every other frame: changeLandmarkAnchor(state, map, marg_kfs),
    # LandmarkUpdate.cpp:318-361) and through the MONO callback (two rows per landmark, initNewLandmarkMono :363-424)
    "kf11_lm": ("feats=100,clones=11,life=45,cohort=0,frames=80,key=1,outlier_every=9", "max_landmark_features: 5\n"),
    "sw11_lm_mono": ("feats=100,clones=11,life=20,cohort=0,frames=70,key=0,stereo=0,outlier_every=9", "frame_select_interval: 5\nmax_landmark_features: 6\n"),
    # the filter ALIGNS ITSELF (round 6; VERDICT r05 missing 3): the recording carries no ALIGNMENT record but a raw epoch (ephemerides +
    # observations, oracle/gen_gnss_raw.py) in front of every GNSS_MEAS record; GvioAligner::batchAlign (GvioAligner.cpp:85-383) buffers
    # gv_align_batch_size of them, finds yaw offset and ENU anchor, and the GNSS updates start with what it found
    "sw11_gnss_align": ("feats=150,clones=11,life=13,cohort=0,frames=70,key=0,gnss=1",
                        "frame_select_interval: 5\ngnss_chi2_test: 1\ngnss_strong_reject: 1\ngv_align_batch_size: 14\n"),
    "sw21": ("feats=100,clones=21,life=25,cohort=0,frames=70,key=0,outlier_every=7", "frame_select_interval: 6\n"),
}


RAW_GNSS = ("sw11_gnss_align",)      # streams whose recording is rewritten with raw GNSS epochs and without the ALIGNMENT record


def write_recording(spec, path, raw_gnss=False):
    subprocess.run([TOOL, "--synth", spec, "--write", path], check=True)
    if raw_gnss:
        from oracle import gen_gnss_raw
        tmp = path + ".plain"
        os.replace(path, tmp)
        gen_gnss_raw.add_raw_epochs(tmp, path)
        os.remove(tmp)


def generate(name, keep_P_every=0):
    from oracle import stream_filter as sf
    spec, overrides = STREAMS[name]
    with tempfile.TemporaryDirectory() as d:
        rec = os.path.join(d, name + ".ingvior")
        write_recording(spec, rec, raw_gnss=name in RAW_GNSS)
        traces = sf.play_recording(rec, overrides)
    out = sf.pack_traces(traces)
    out["spec"] = np.array(spec)
    out["overrides"] = np.array(overrides)
    return traces, out


def main(argv):
    names = [a for a in argv if not a.startswith("-")] or list(STREAMS)
    for name in names:
        traces, out = generate(name)
        path = os.path.join(GOLDEN, "stream_%s.npz" % name)
        np.savez_compressed(path, **out)
        lost = [len(t["lost_ids"]) for t in traces]
        acc = [int(np.sum(t["lost_acc"])) for t in traces]
        sel = [len(t["sel_ids"]) for t in traces]
        sacc = [int(np.sum(t["sel_acc"])) for t in traces]
        print("%s: %d frames, N %d..%d, RemoveLost features/frame max %d (accepted max %d), selected-update features max %d (accepted max %d), "
              "final |p| %.3f, %d bytes" % (name, len(traces), int(min(t["n"] for t in traces)), int(max(t["n"] for t in traces)), max(lost), max(acc),
                                            max(sel), max(sacc), float(np.linalg.norm(traces[-1]["pose"][9:12])), os.path.getsize(path)))


if __name__ == "__main__":
    main(sys.argv[1:])
