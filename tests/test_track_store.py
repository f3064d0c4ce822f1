"""Device-resident track store (round 6; include/ingvio_hip.h: ingvio_tracks_create / ingvio_frame_stage_tracks): a frame handed over as
a DELTA on the stored observations (one new column per frame, window slots that leave, erased tracks, changed points) + the list of
tracks the update uses + raw IMU samples must leave the context exactly as ingvio_frame_stage does with the whole flattened frame and
the host's transition matrices (MapServerManager.cpp:146-217 builds the same observation sets incrementally; ImuPropagator.cpp:98-162
forms the same Phi / G).  The measurements, masks, points, anchors and dofs the kernels read are bit-identical; Phi / G come from the
device's sin / cos instead of the host's (1-2 ulp), so the posterior is compared to 1e-11 relative."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300))


def build(nb, C, F, seed0, stereo=True):
    from ingvio_amd import capi, host, synth
    N = 21 + 6 + 6 * C
    ctx = capi.Context(batch=nb, n_max=((N + 15) // 16) * 16, c_max=C, f_max=F, m_max=64)
    cases = []
    for b in range(nb):
        flt, step, frame, info = synth.build_case(lambda P, b=b: capi.DeviceCov(ctx, b, P), host.imu_transition, seed=seed0 + b, F=F, C=C,
                                                  n_gnss=6, n_landmarks=0, stereo=stereo)
        rng = np.random.default_rng(900 + b)
        mask = np.zeros(F, dtype=np.uint64); dof = np.zeros(F, dtype=np.int32)
        for j in range(F):                                              # ragged tracks: every feature its own observation set
            k = int(rng.integers(4, C + 1))
            obs = np.sort(rng.choice(C, size=k, replace=False))
            mask[j] = np.uint64(sum(1 << int(o) for o in obs)); dof[j] = k - 1
        frame = dict(frame); frame["obs_mask"] = mask; frame["dof"] = dof
        frame["anchor"] = np.array([int([o for o in range(C) if (int(mask[j]) >> o) & 1][j % 3]) for j in range(F)], dtype=np.int32)
        frame["uv"] = np.array(frame["uv"]) * ((mask[:, None] >> np.arange(C, dtype=np.uint64)[None, :]) & np.uint64(1)).astype(float)[:, :, None]
        cases.append((flt, step, frame, info))
    return ctx, cases


def run_reference(ctx, cases, **kw):
    ctx.snapshot()
    ctx.frame_stage(0, [c[1] for c in cases], [c[2] for c in cases], cases[0][1]["sigma"], 1, 0.2, 0.2, **kw)
    ctx.frame_run(restore_prior=True)
    dx, acc, rows = ctx.frame_fetch()
    return dx.copy(), acc.copy(), rows.copy(), [ctx.cov_get(b) for b in range(len(cases))]


@pytest.mark.parametrize("use_async,selected", [(False, 0), (True, 0), (False, 1)])
def test_frame_from_the_track_store_equals_the_staged_frame(use_async, selected):
    nb, C, F, T = 3, 11, 40, 64
    ctx, cases = build(nb, C, F, 410)
    kw = dict(max_accept=0, compress_rule=1, selected_variant=selected)
    sel = None
    if selected:                                                         # the Selected-timestamp updates see three stamps of every track
        sel = [np.array([np.uint64(int(m) & 0b10000100001 or int(m)) for m in c[2]["obs_mask"]], dtype=np.uint64) for c in cases]
        for c, s in zip(cases, sel):
            c[2]["obs_mask"] = c[2]["obs_mask"] & s
            keep = ((c[2]["obs_mask"][:, None] >> np.arange(C, dtype=np.uint64)[None, :]) & np.uint64(1)).astype(float)
            c[2]["uv_sel"] = np.array(c[2]["uv"]) * keep[:, :, None]
    ref_cases = [(c[0], c[1], dict(c[2], uv=c[2].get("uv_sel", c[2]["uv"])), c[3]) for c in cases]
    dx0, acc0, rows0, P0 = run_reference(ctx, ref_cases, **kw)
    assert acc0[:, :F].sum() > nb * F // 2

    ctx.tracks_create(T)
    perms = [np.random.default_rng(50 + b).permutation(T)[:F].astype(np.int32) for b in range(nb)]
    # the masks BEFORE the selection: the store holds every observation, the selection rides on the frame list
    full_masks = [np.array([int(m) for m in (c[2]["obs_mask"] if not selected else c[2]["obs_mask"])], dtype=np.uint64) for c in cases]
    uv_full = [np.array(c[2]["uv"]) for c in cases]
    if selected:
        full_masks = [np.array([sum(1 << s for s in range(C) if np.any(uv_full[b][j, s] != 0.0)) for j in range(F)], dtype=np.uint64) for b in range(nb)]
    opts_frame = cases[0][2]
    steps = [c[1] for c in cases]

    def stage(deltas, feats=False):
        tfs = []
        for b in range(nb):
            fr = cases[b][2]
            d = dict(deltas[b], clone_idx=fr["clone_idx"], clone_R=fr["clone_R"], clone_p=fr["clone_p"], feat_track=[], feat_anchor=[], feat_dof=[])
            if feats:
                d.update(feat_track=perms[b], feat_anchor=fr["anchor"], feat_dof=fr["dof"], feat_sel=None if sel is None else sel[b],
                         pf_track=perms[b], pf=fr["pf"])
            tfs.append(d)
        call = ctx.frame_stage_tracks_prepare(0, steps, tfs, opts_frame, steps[0]["sigma"], 1, 0.2, 0.2, use_async=use_async, **kw)
        call()

    def column(b, s, slot, extra=None):
        obs = [j for j in range(F) if (int(full_masks[b][j]) >> s) & 1]
        tr = [int(perms[b][j]) for j in obs]
        uv = [uv_full[b][j, s] for j in obs]
        if extra is not None:
            tr.append(extra); uv.append(np.array([9.0, 9.0, 9.0, 9.0]))
        return dict(append=slot, obs_track=tr, obs_uv=np.array(uv).reshape(-1, 4))

    junk_track = [int(np.setdiff1d(np.arange(T), perms[b])[0]) for b in range(nb)]
    # 1. two junk columns at slots 0 and 1 (every track observed, a junk track beside them) ...
    for slot in (0, 1):
        stage([dict(append=slot, obs_track=list(map(int, perms[b])) + [junk_track[b]], obs_uv=np.full((F + 1, 4), 7.0 + slot)) for b in range(nb)])
    # 2. ... leave the window again while the first real column arrives (drop + append in one delta), the junk track is erased
    stage([dict(column(b, 0, 0), drop=[0, 1], free=[junk_track[b]]) for b in range(nb)])
    # 3. the real columns 1 .. C-2, with a column in the middle that is dropped one frame later (the rows close up)
    for s in range(1, C - 1):
        if s == 5:
            stage([dict(append=s, obs_track=list(map(int, perms[b][:7])), obs_uv=np.full((7, 4), 3.0)) for b in range(nb)])
            stage([dict(column(b, s, s), drop=[s]) for b in range(nb)])
        else:
            stage([column(b, s, s) for b in range(nb)])
    # 4. the newest column together with the frame the update uses and the points
    stage([column(b, C - 1, C - 1) for b in range(nb)], feats=True)
    ctx.frame_run(restore_prior=True)
    dx1, acc1, rows1 = ctx.frame_fetch()
    assert np.array_equal(acc1, acc0) and np.array_equal(rows1, rows0)
    for b in range(nb):
        assert rel(ctx.cov_get(b), P0[b]) < 1e-11 and rel(dx1[b], dx0[b]) < 1e-9, (b, rel(ctx.cov_get(b), P0[b]), rel(dx1[b], dx0[b]))
    # 5. a steady-state frame: the newest slot leaves and comes back with the same column - the store is where it was
    stage([dict(column(b, C - 1, C - 1), drop=[C - 1]) for b in range(nb)], feats=True)
    ctx.frame_run(restore_prior=True)
    dx2, acc2, rows2 = ctx.frame_fetch()
    assert np.array_equal(dx2, dx1) and np.array_equal(acc2, acc1)
    ctx.close()


def test_track_store_refuses_inconsistent_deltas():
    from ingvio_amd import capi
    ctx, cases = build(1, 6, 8, 77)
    fr = cases[0][2]
    base = dict(clone_idx=fr["clone_idx"], clone_R=fr["clone_R"], clone_p=fr["clone_p"], feat_track=[], feat_anchor=[], feat_dof=[])
    steps = [cases[0][1]]

    def try_stage(d):
        call = ctx.frame_stage_tracks_prepare(0, steps, [dict(base, **d)], fr, steps[0]["sigma"], 1, 0.2, 0.2)
        call()
    with pytest.raises(capi.IngvioError):                                # no store yet
        try_stage(dict())
    ctx.tracks_create(16)
    try_stage(dict(append=0, obs_track=[0, 1], obs_uv=np.zeros((2, 4))))
    for bad in (dict(append=0, obs_track=[16], obs_uv=np.zeros((1, 4))),          # track out of range
                dict(drop=[3, 2]),                                                # not ascending
                dict(append=6, obs_track=[0], obs_uv=np.zeros((1, 4))),           # slot beyond c_max
                dict(feat_track=[0], feat_anchor=[6], feat_dof=[1]),               # anchor outside the window
                dict(obs_track=[0], obs_uv=np.zeros((1, 4)))):                    # observations without a slot
        with pytest.raises(capi.IngvioError):
            try_stage(bad)
    ctx.close()
