"""Batched SLAM-landmark update (ingvio_landmark_stage / _run / _fetch, kernels_lmbatch.hip + kernels_chol.hip) against the oracle:
rows from orc.landmark_rows_epose (LandmarkUpdate.cpp:521-686 restated, itself pinned by finite differences in test_landmark_path.py),
per-landmark gates with numpy, the stacked update with orc.Cov.ekf_update.  Also the dense-H route of ingvio_ekf_update for row counts
whose S does not fit in LDS."""
import numpy as np
import pytest

from oracle import oracle as orc


def spd(n, rng, scale=1.0):
    A = rng.standard_normal((n, n))
    return scale * (A @ A.T / n + 0.1 * np.eye(n))


def rot(rng, mag=1.0):
    return orc.gamma(mag * rng.standard_normal(3), 0).reshape(3, 3)

CHI2_4, CHI2_2 = 9.487729036781154, 5.991464547107979


def make_filter(rng, C, L, stereo, n_gnss=6, outliers=(), untracked=()):
    """state [epose 9 | biases 6 | ext 6 | gnss | clones 6C | landmarks 3L], everything in view of the camera"""
    i_ext = 15; i_cl = 21 + n_gnss; i_lm = i_cl + 6 * C
    n = i_lm + 3 * L
    R_i2w = rot(rng, 0.3); p_i2w = rng.standard_normal(3)
    R_cl2i = rot(rng, 0.05); p_c2i = 0.05 * rng.standard_normal(3)
    Rlr = rot(rng, 0.01); tlr = np.array([-0.11, 0.002, 0.001])
    R_w2c = R_cl2i.T @ R_i2w.T
    pf, uv = np.zeros((L, 3)), np.zeros((L, 4))
    for l in range(L):
        q = np.array([rng.uniform(-1.5, 1.5), rng.uniform(-1.0, 1.0), rng.uniform(3.0, 9.0)])      # in the left camera
        pf[l] = R_i2w @ (R_cl2i @ q + p_c2i) + p_i2w
        qr = Rlr @ q + tlr
        uv[l] = [q[0] / q[2], q[1] / q[2], qr[0] / qr[2], qr[1] / qr[2]]
        uv[l] += 0.01 * rng.standard_normal(4) * (30.0 if l in outliers else 1.0)
    tracked = np.ones(L, dtype=np.uint8)
    for l in untracked: tracked[l] = 0
    frame = dict(R_i2w=R_i2w, p_i2w=p_i2w, R_cl2i=R_cl2i, p_c2i=p_c2i, idx_epose=0, idx_ext=i_ext,
                 lm_idx=[i_lm + 3 * l for l in range(L)], anchor_idx=[i_cl + 6 * int(rng.integers(0, C)) for _ in range(L)],
                 pf=pf, uv=uv, tracked=tracked)
    return n, frame, Rlr, tlr


def oracle_update(P0, frame, stereo, noise, Rlr, tlr):
    per = 4 if stereo else 2
    thr = CHI2_4 if stereo else CHI2_2
    n = P0.shape[0]
    rows, res, acc, gam = [], [], [], []
    for l, (il, ia) in enumerate(zip(frame["lm_idx"], frame["anchor_idx"])):
        if not frame["tracked"][l]:
            acc.append(0); gam.append(-1.0); continue
        H, r = orc.landmark_rows_epose(frame["R_i2w"], frame["p_i2w"], frame["R_cl2i"], frame["p_c2i"], frame["pf"][l], frame["uv"][l],
                                       stereo, Rlr, tlr)
        Hd = np.zeros((per, n))
        Hd[:, 0:9] += H[:per, 0:9]; Hd[:, 15:21] += H[:per, 9:15]; Hd[:, ia:ia + 6] += H[:per, 15:21]; Hd[:, il:il + 3] += H[:per, 21:24]
        S = Hd @ P0 @ Hd.T + noise ** 2 * np.eye(per)
        g = float(r[:per] @ np.linalg.solve(S, r[:per]))
        gam.append(g)
        if g < thr:
            acc.append(1); rows.append(Hd); res.append(r[:per])
        else:
            acc.append(0)
    oc = orc.Cov(P0)
    dx = np.zeros(n)
    if rows:
        H = np.vstack(rows); r = np.concatenate(res)
        dx, _ = oc.ekf_update([0], [n], H, r, noise ** 2)
    return oc.P, dx, np.array(acc), np.array(gam), per * int(np.sum(acc))


@pytest.mark.gpu
@pytest.mark.parametrize("stereo,L,C", [(True, 52, 11), (False, 20, 11), (True, 3, 11), (True, 64, 11), (True, 40, 30), (False, 64, 21)])
def test_landmark_batch_vs_oracle(stereo, L, C):
    from ingvio_amd import capi
    rng = np.random.default_rng(1000 + L + C)
    B, noise = 3, 0.02
    frames, priors, extras = [], [], []
    n_max = 0
    for b in range(B):
        n, fr, Rlr, tlr = make_filter(rng, C, L, stereo, outliers=(1, 5) if L > 6 else (1,), untracked=(2,) if b == 1 else ())
        frames.append(fr); priors.append(spd(n, rng, 1e-3)); extras.append((Rlr, tlr)); n_max = max(n_max, n)
    Rlr, tlr = extras[0]
    ctx = capi.Context(batch=B, n_max=n_max + 6, c_max=C, f_max=8, m_max=64)
    for b in range(B): ctx.cov_set(b, priors[b])
    ctx.landmark_stage(0, frames, stereo, noise, CHI2_4 if stereo else CHI2_2, Rlr, tlr)
    ctx.landmark_run()
    dx, rows, acc, gam, st = ctx.landmark_fetch()
    assert not st.any()
    for b in range(B):
        Pw, dxw, accw, gamw, mw = oracle_update(priors[b], frames[b], stereo, noise, Rlr, tlr)
        n = priors[b].shape[0]
        assert np.array_equal(acc[b, :L], accw) and rows[b] == mw
        on = gamw >= 0
        assert np.allclose(gam[b, :L][on], gamw[on], rtol=1e-9, atol=1e-12) and (gam[b, :L][~on] == -1).all()
        P = ctx.cov_get(b)
        assert np.linalg.norm(P - Pw) < 1e-9 * np.linalg.norm(Pw) and np.array_equal(P, P.T)
        assert np.linalg.norm(dx[b, :n] - dxw) < 1e-8 * max(1e-6, np.linalg.norm(dxw))
    # replay: staged inputs are read-only
    for b in range(B): ctx.cov_set(b, priors[b])
    ctx.landmark_run()
    dx2, rows2, acc2, _, _ = ctx.landmark_fetch()
    assert np.array_equal(dx, dx2) and np.array_equal(acc, acc2)
    ctx.close()


@pytest.mark.gpu
def test_landmark_batch_nothing_accepted_leaves_state():
    from ingvio_amd import capi
    rng = np.random.default_rng(77)
    n, fr, Rlr, tlr = make_filter(rng, 6, 5, True, untracked=(0, 1, 2, 3, 4))
    P0 = spd(n, rng, 1e-3)
    ctx = capi.Context(batch=1, n_max=n + 6, c_max=6, f_max=8, m_max=64)
    ctx.cov_set(0, P0)
    ctx.landmark_stage(0, [fr], True, 0.02, CHI2_4, Rlr, tlr)
    ctx.landmark_run()
    dx, rows, acc, gam, st = ctx.landmark_fetch()
    assert rows[0] == 0 and not acc.any() and not dx.any() and np.array_equal(ctx.cov_get(0), P0)
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("m,r_kind", [(300, "diag"), (150, "full"), (200, "scalar")])
def test_ekf_update_rows_beyond_lds(m, r_kind):
    """ingvio_ekf_update with more rows than S fits in LDS (round 1: INGVIO_E_CAPACITY above 128 / 136 rows): the dense-H route
    (GEMM + Cholesky sweep with carried rows + downdate) against the oracle's ekfUpdate (StateManager.cpp:359-411)."""
    from ingvio_amd import capi
    rng = np.random.default_rng(m)
    n = 21 + 66 + 30
    P0 = spd(n, rng, 1e-2)
    ctx = capi.Context(batch=2, n_max=n + 3, c_max=11, f_max=8, m_max=64)
    ctx.cov_set(1, P0); ctx.cov_set(0, 2 * P0)
    vo, vs = [0, 27, 87, 21], [9, 30, 30, 6]
    H = rng.standard_normal((m, sum(vs))); r = 0.1 * rng.standard_normal(m)
    if r_kind == "diag": R = rng.uniform(0.5, 2.0, m)
    elif r_kind == "full": A = rng.standard_normal((m, m)); R = A @ A.T / m + 0.5 * np.eye(m)
    else: R = 0.7
    oc = orc.Cov(P0); dxw, _ = oc.ekf_update(vo, vs, H, r, R)
    dx, _ = ctx.ekf_update(1, vo, vs, H, r, R)
    P = ctx.cov_get(1)
    assert np.linalg.norm(P - oc.P) < 1e-10 * np.linalg.norm(oc.P) and np.array_equal(P, P.T)
    assert np.linalg.norm(dx[:n] - dxw) < 1e-9 * np.linalg.norm(dxw)
    assert np.array_equal(ctx.cov_get(0), 2 * P0)                       # the neighbour filter is untouched
    ctx.close()


@pytest.mark.gpu
def test_frame_with_in_state_landmarks_vs_oracle():
    """The frame path carrying REAL in-state landmarks: N = 249 with 52 landmarks that receive rows.  Device:
    ingvio_frame_run = propagate + clone + MSCKF update, then the batched landmark update, then the marginalisation
    (IngvioFilter.cpp:296-322 order; nominal values fixed over the step).  Oracle: the same sequence on orc.Cov."""
    from ingvio_amd import capi, host, synth
    nb, L, noise = 4, 52, synth.PARAMS["visual_noise"]
    ctx = capi.Context(batch=nb, n_max=256, c_max=11, f_max=150, m_max=64)
    cases = []
    for b in range(nb):
        flt, step, frame, info = synth.build_case(lambda P, b=b: capi.DeviceCov(ctx, b, P), host.imu_transition, seed=500 + b, lm_sigma=0.05)
        lm = synth.make_landmarks(np.random.default_rng(40 + b), flt, frame, L, untracked=(3,) if b == 2 else ())
        cases.append((flt, step, frame, info, lm))
    priors = [ctx.cov_get(b) for b in range(nb)]
    Rlr, tlr = synth.t_cl2cr()
    ctx.snapshot()
    ctx.frame_stage(0, [c[1] for c in cases], [c[2] for c in cases], cases[0][1]["sigma"], 1, 0.2, 0.2)
    ctx.landmark_stage(0, [c[4] for c in cases], True, noise, CHI2_4, Rlr, tlr, in_frame=True)
    ctx.frame_run(restore_prior=True)
    dxv, acc, rows = ctx.frame_fetch()
    dxl, lrows, lacc, lgam, st = ctx.landmark_fetch()
    assert not st.any()
    for b in range(nb):
        flt, step, frame, info, lm = cases[b]
        oc = orc.Cov(priors[b], ld=256)
        for Phi, G, dt in zip(step["Phi"], step["G"], step["dt"]):
            oc.propagate(Phi, G, dt, step["sigma"], step["enable_gnss"], step["gnss_idx"], 0.2, 0.2)
        oc.augment(step["R_i2w"])
        dxo, acco, _ = oc.msckf_update(frame, max_accept=0, compress_rule=1)[:3]
        assert np.array_equal(acc[b, :150], acco)
        Pw, dxw, accw, gamw, mw = oracle_update(oc.P, lm, True, noise, Rlr, tlr)
        assert np.array_equal(lacc[b, :L], accw) and lrows[b] == mw and mw >= 2 * L
        oc2 = orc.Cov(Pw, ld=256)
        oc2.marginalize(step["marg_idx"], 6)
        P = ctx.cov_get(b)
        assert ctx.n(b) == 243 and np.linalg.norm(P - oc2.P) < 1e-9 * np.linalg.norm(oc2.P) and np.array_equal(P, P.T)
        assert np.linalg.norm(dxl[b, :249] - dxw) < 1e-8 * np.linalg.norm(dxw)
    P1 = [ctx.cov_get(b) for b in range(nb)]
    ctx.frame_run(restore_prior=True)                                     # repeatable from the restored prior
    for b in range(nb):
        assert np.array_equal(ctx.cov_get(b), P1[b])
    ctx.close()


@pytest.mark.gpu
def test_ekf_update_batch_rows_beyond_lds():
    """ingvio_ekf_update_batch with row counts whose S does not fit LDS: the whole call takes the dense-H route."""
    from ingvio_amd import capi
    rng = np.random.default_rng(4242)
    nb, n = 3, 21 + 66 + 30
    ctx = capi.Context(batch=nb + 1, n_max=n + 3, c_max=11, f_max=8, m_max=64)
    blocks, want = [], []
    for b in range(nb):
        P0 = spd(n, rng, 1e-2)
        ctx.cov_set(b + 1, P0)
        m = (200, 150, 40)[b]
        vo, vs = [0, 27 + 6 * b, 87], [9, 30, 30]
        H = rng.standard_normal((m, sum(vs))); r = 0.1 * rng.standard_normal(m); R = rng.uniform(0.5, 2.0, m)
        blocks.append((vo, vs, H, r, R))
        oc = orc.Cov(P0); dxo, _ = oc.ekf_update(vo, vs, H, r, R)
        want.append((oc.P, dxo))
    dx, st = ctx.ekf_update_batch(1, blocks, diag=True)
    assert not st.any()
    for b in range(nb):
        P = ctx.cov_get(b + 1)
        assert np.linalg.norm(P - want[b][0]) < 1e-10 * np.linalg.norm(want[b][0]) and np.array_equal(P, P.T)
        assert np.linalg.norm(dx[b, :n] - want[b][1]) < 1e-9 * np.linalg.norm(want[b][1])
    ctx.close()
