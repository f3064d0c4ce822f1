"""GPU parity tests that pin K3-K7 WITHOUT the oracle, and the conditioning / config-3 checks VERDICT r01 asked for.

1. Finite differences: the MSCKF measurement Jacobian is obtained by differentiating the stereo / mono projection through the
   reference's retractions (tests/fd_jacobian.py: no analytic Jacobian is written down there) and compared with what the HIP
   kernels produce: the per-feature information H_j^T H_j | H_j^T r_j (factored: gram kernel; dense: R^T R of the TSQR factor),
   the gate value gamma and the posterior.  Both the RemoveLost form and the Selected form (quirk Q10) are covered.
2. sigma x prior-scale sweep of the information-form update (A Pcc + sigma^2 I is the matrix being factorised) against the
   dense path and the oracle.
3. BASELINE config 3 as a frame: 150-feature visual update at N = 249, then the 8-satellite updateTrackedSys with the per-row
   chi^2 gates, everything on the device, against the oracle.
4. RCCL at world size 1 (the only RCCL this box can run)."""
import os

import numpy as np
import pytest

from conftest import rel_err
import fd_jacobian as fd

pytestmark = pytest.mark.gpu

FD_TOL = 2e-9            # five-point differences with h = 1e-3 leave ~1e-11 on H, ~1e-10 on H^T H


def _prior_at_update(ctx, b, flt, step):
    """flt.cov (a DeviceCov on filter b) holds the prior of the frame: propagate k steps and clone, as the frame call would."""
    for Phi, G, dt in zip(step["Phi"], step["G"], step["dt"]):
        flt.cov.propagate(Phi, G, dt, step["sigma"], step["enable_gnss"], step["gnss_idx"], step["sigma_cb"], step["sigma_rw"])
    flt.cov.augment(step["R_i2w"])
    return ctx.cov_get(b)


def _ragged(frame, rng, C, kmin, anchor_in_obs=None):
    F = frame["pf"].shape[0]
    mask = np.zeros(F, dtype=np.uint64); dof = np.zeros(F, dtype=np.int32); anchor = np.zeros(F, dtype=np.int32)
    for j in range(F):
        k = int(rng.integers(kmin, C + 1))
        obs = np.sort(rng.choice(C, size=k, replace=False))
        mask[j] = np.uint64(sum(1 << int(o) for o in obs)); dof[j] = k - 1
        inside = (j % 2 == 0) if anchor_in_obs is None else anchor_in_obs
        anchor[j] = int(rng.choice(obs)) if inside else int(rng.integers(0, C))
    out = dict(frame); out["obs_mask"] = mask; out["dof"] = dof; out["anchor"] = anchor
    return out


def _one_feature(frame, j):
    out = dict(frame)
    for k in ("pf", "anchor", "obs_mask", "uv", "dof"):
        out[k] = np.ascontiguousarray(np.asarray(frame[k])[j:j + 1])
    return out


@pytest.mark.parametrize("method", ["factored", "dense"])
@pytest.mark.parametrize("stereo,selected", [(True, False), (True, True), (False, False), (False, True)])
def test_feature_information_vs_finite_differences(method, stereo, selected):
    """One feature per update: [A | b] of the HIP path = H_j^T H_j | H_j^T r_j must equal the finite-difference Jacobian projected
    on the left nullspace of the finite-difference H_f (basis-free: V V^T = I - Hf (Hf^T Hf)^-1 Hf^T), gamma must equal
    r_j^T (H_j Pcc H_j^T + s^2 I)^-1 r_j and the posterior the plain Kalman update with that H_j."""
    from ingvio_amd import capi, host, synth
    C, F = 11, 24
    ctx = capi.Context(batch=1, n_max=96, c_max=C, f_max=F, m_max=64)
    ctx.set_method(method)
    flt, step, frame, info = synth.build_case(lambda P: capi.DeviceCov(ctx, 0, P), host.imu_transition, seed=11, F=F, C=C,
                                              n_gnss=0, n_landmarks=0, stereo=stereo, outlier_every=0)
    P0 = _prior_at_update(ctx, 0, flt, step)
    frame = _ragged(frame, np.random.default_rng(5), C, 4 if stereo else 5)
    frame["chi2_table"] = np.full(151, 1e300)                  # every feature passes: the gate VALUE is checked below
    worst = dict(A=0.0, b=0.0, g=0.0, P=0.0, dx=0.0)
    for j in range(F):
        f1 = _one_feature(frame, j)
        ctx.cov_set(0, P0)
        dx, acc, gam, rows = ctx.msckf_update(0, [f1] if False else f1, selected_variant=int(selected))
        assert acc[0, 0] == 1 and rows[0] == 6 * C
        A, bvec = ctx.debug_msckf_info(0)
        Afd, bfd, Hj, rj = fd.feature_info_fd(frame, j, selected_variant=selected)
        worst["A"] = max(worst["A"], rel_err(A, Afd)); worst["b"] = max(worst["b"], rel_err(bvec, bfd))
        gfd = fd.gate_gamma(P0, frame, Hj, rj)
        worst["g"] = max(worst["g"], abs(gam[0, 0] - gfd) / max(gfd, 1e-300))
        Pfd, dxfd = fd.ekf_update_dense(P0, frame, Hj, rj)
        worst["P"] = max(worst["P"], rel_err(ctx.cov_get(0), Pfd)); worst["dx"] = max(worst["dx"], rel_err(dx[0, :P0.shape[0]], dxfd))
    print("FD pin (%s, stereo=%s, selected=%s): %s" % (method, stereo, selected, worst))
    assert worst["A"] < FD_TOL and worst["b"] < FD_TOL and worst["g"] < 1e-8 and worst["P"] < 1e-9 and worst["dx"] < 1e-7
    ctx.close()


@pytest.mark.parametrize("method", ["factored", "dense"])
def test_frame_information_vs_finite_differences(method):
    """The whole 150-feature frame of BASELINE config 2 (N = 87 literal): accept mask from finite-difference gammas, stacked
    information = sum of the accepted features' finite-difference H_j^T H_j, posterior = Kalman update with the stacked
    finite-difference H (the reference's formulation before any QR)."""
    from ingvio_amd import capi, host, synth
    C, F = 11, 150
    ctx = capi.Context(batch=1, n_max=96, c_max=C, f_max=F, m_max=64)
    ctx.set_method(method)
    flt, step, frame, info = synth.build_case(lambda P: capi.DeviceCov(ctx, 0, P), host.imu_transition, seed=2, F=F, C=C,
                                              n_gnss=0, n_landmarks=0)
    P0 = _prior_at_update(ctx, 0, flt, step)
    dx, acc, gam, rows = ctx.msckf_update(0, frame)
    A, bvec = ctx.debug_msckf_info(0)
    table = frame["chi2_table"]
    Afd = np.zeros((6 * C, 6 * C)); bfd = np.zeros(6 * C); Hs, rs = [], []
    for j in range(F):
        Aj, bj, Hj, rj = fd.feature_info_fd(frame, j)
        g = fd.gate_gamma(P0, frame, Hj, rj)
        assert abs(gam[0, j] - g) <= 1e-8 * g
        ok = g < table[int(frame["dof"][j])]
        assert bool(acc[0, j]) == bool(ok)
        if ok:
            Afd += Aj; bfd += bj; Hs.append(Hj); rs.append(rj)
    assert np.array_equal(acc[0, :F] == 0, info["outlier"])
    assert rel_err(A, Afd) < FD_TOL and rel_err(bvec, bfd) < FD_TOL
    Pfd, dxfd = fd.ekf_update_dense(P0, frame, np.vstack(Hs), np.concatenate(rs))
    assert rel_err(ctx.cov_get(0), Pfd) < 1e-8 and rel_err(dx[0, :P0.shape[0]], dxfd) < 1e-7
    ctx.close()


def _truth_update_longdouble(P, cols, Aj, bj, acc, var):
    """The update in 80-bit arithmetic (information form with partial pivoting): the yardstick for BOTH the HIP path and the
    oracle where the posterior is dominated by cancellation (P - K H P with a large prior and a small noise)."""
    LD = np.longdouble
    n = len(cols)
    A = np.zeros((n, n), dtype=LD); b = np.zeros(n, dtype=LD)
    for j in np.flatnonzero(acc):
        A += Aj[j]; b += bj[j]
    Pl = P.astype(LD)
    Pc = Pl[:, cols]; Pcc = Pl[np.ix_(cols, cols)]
    aug = np.hstack([A @ Pcc + LD(var) * np.eye(n, dtype=LD), A, b[:, None]])
    for k in range(n):
        p = k + int(np.argmax(np.abs(aug[k:, k])))
        if p != k:
            aug[[k, p]] = aug[[p, k]]
        aug[k] /= aug[k, k]
        f = aug[:, k].copy(); f[k] = 0
        aug -= np.outer(f, aug[k])
    M, t = aug[:, n:2 * n], aug[:, 2 * n]
    Pn = Pl - Pc @ M @ Pc.T
    return (0.5 * (Pn + Pn.T)).astype(np.float64), (Pc @ t).astype(np.float64)


@pytest.mark.parametrize("scale", [1e-4, 1.0, 1e4])
def test_sigma_and_prior_scale_sweep(orc, scale):
    """Conditioning of the factored update.  Sweep the visual noise (shipped configs: 0.18 and 0.08; two much smaller values) and
    the scale of the prior (x 1e-4 / 1 / 1e4) at N = 249, factored and dense methods against the oracle: accept masks exactly
    equal; covariance within the BASELINE tolerance (1e-6 relative, whole matrix).  On the window block alone and on dx the
    reference's own formulation P - K H P cancels up to 10 digits when the prior is inflated and the noise tiny, so there
    the yardstick is an 80-bit evaluation of the same update: the HIP path must be within 1e-6 of it or at least as close to
    it as the oracle is (x 4)."""
    from ingvio_amd import capi, host, synth
    sigmas = [0.18, 0.08, 1e-2, 1e-3]
    nb = len(sigmas)
    ctxs = {}
    for method in ("factored", "dense"):
        c = capi.Context(batch=nb, n_max=256, c_max=11, f_max=150, m_max=64)
        c.set_method(method)
        ctxs[method] = c
    ctx0 = ctxs["factored"]
    flt, step, frame, info = synth.build_case(lambda P: capi.DeviceCov(ctx0, 0, P), host.imu_transition, seed=21)
    P0 = _prior_at_update(ctx0, 0, flt, step)
    cols = np.concatenate([np.arange(i, i + 6) for i in frame["clone_idx"]])
    LD = np.longdouble
    Aj, bj = [], []
    for j in range(150):                                       # per-feature information in 80 bits, once
        Hj, rj = orc.feature_block(frame, j)
        Hl, rl = Hj.astype(LD), rj.astype(LD)
        Aj.append(Hl.T @ Hl); bj.append(Hl.T @ rl)
    post = {}
    for method, ctx in ctxs.items():
        frames = []
        for b, s in enumerate(sigmas):
            ctx.cov_set(b, P0 * scale)
            f = dict(frame); f["noise"] = s
            frames.append(f)
        res = [ctx.msckf_update(b, frames[b]) for b in range(nb)]
        post[method] = [(ctx.cov_get(b), res[b][0][0, :249], res[b][1][0, :150]) for b in range(nb)]
    report = []
    win = np.ix_(cols, cols)
    for b, s in enumerate(sigmas):
        oc = orc.Cov(P0 * scale, ld=256)
        f = dict(frame); f["noise"] = s
        dxo, acco, gamo, m = oc.msckf_update(f, max_accept=0, compress_rule=1)
        Po = oc.P
        Pt, dxt = _truth_update_longdouble(P0 * scale, cols, Aj, bj, acco, s * s)
        o_win, o_dx = rel_err(Po[win], Pt[win]), rel_err(dxo, dxt)
        for method in ("factored", "dense"):
            Pg, dxg, accg = post[method][b]
            assert np.array_equal(accg, acco), (method, s, scale)
            e_full = rel_err(Pg, Po)
            g_win, g_dx = rel_err(Pg[win], Pt[win]), rel_err(dxg, dxt)
            report.append((method, scale, s, int(acco.sum()), e_full, g_win, o_win, g_dx, o_dx))
            assert e_full < 1e-6, (method, s, scale, e_full)
            # Round 3: no exempted regime.  What used to lose the window block at a 1e4 x prior and s <= 1e-2 was not the
            # cancellation in P - (Pc M) Pc^T but the eps |A| violation of A's gauge null space (tools/gpu_gauge_diag.py);
            # the information solve now runs in difference coordinates to a reference clone (kernels_solve.hip), where A has
            # full rank, and both methods are held to the same bound everywhere.
            bound = 1e-6
            assert g_win < max(bound, 4 * o_win), (method, s, scale, g_win, o_win)
            assert g_dx < max(bound, 4 * o_dx), (method, s, scale, g_dx, o_dx)
    for r in report:
        print("sweep %-8s scale %.0e sigma %-6g accepted %3d  cov vs oracle %.1e | window vs 80-bit: hip %.1e oracle %.1e | dx vs 80-bit: hip %.1e oracle %.1e" % r)
    for c in ctxs.values():
        c.close()


@pytest.mark.parametrize("strong_reject", [0, 1])
def test_config3_frame_vs_oracle(orc, strong_reject):
    """BASELINE config 3: 150 feats x 11 clones at N = 249 (6 GNSS scalars in the state), the camera frame (propagate + clone +
    MSCKF update + marginalise) followed by GnssUpdate::updateTrackedSys with 8 satellites, gnss_chi2_test = 1: the 16 per-row
    gates, the compaction, the (optional) block gate and ekfUpdate all run on the device (ingvio_gnss_update_batch), for a
    batch of filters at once.  IngvioFilter.cpp:275-352 order: visual update, marginalisation, then GNSS."""
    from ingvio_amd import capi, host, synth
    nb = 6
    ctx = capi.Context(batch=nb, n_max=256, c_max=11, f_max=150, m_max=64)
    cases = []
    for b in range(nb):
        flt, step, frame, info = synth.build_case(lambda P, b=b: capi.DeviceCov(ctx, b, P), host.imu_transition, seed=300 + b)
        rng = np.random.default_rng(900 + b)
        g = synth.make_gnss(rng, flt, outliers=(5,) if b % 2 == 0 else (1, 6))
        cases.append((flt, step, frame, info, g))
    priors = [ctx.cov_get(b) for b in range(nb)]
    table = cases[0][2]["chi2_table"]
    ctx.snapshot()
    ctx.frame_stage(0, [c[1] for c in cases], [c[2] for c in cases], cases[0][1]["sigma"], 1, 0.2, 0.2)
    ctx.frame_run(restore_prior=True)
    dxv, acc, rows = ctx.frame_fetch()
    blocks = [host.gnss_rows(c[4]) for c in cases]
    if strong_reject:
        # filter 1: gross residuals everywhere and no per-row gate -> the block gate must refuse the whole update
        blocks = [tuple(blk) for blk in blocks]
    dxg, used, keep, gam, st = ctx.gnss_update_batch(0, blocks, table, gate_rows=True, strong_reject=bool(strong_reject))
    for b in range(nb):
        flt, step, frame, info, g = cases[b]
        oc = orc.Cov(priors[b], ld=256)
        dxo, acco, gamo, m = orc.frame_update(oc, step, frame, max_accept=0, compress_rule=1)
        assert np.array_equal(acc[b, :150], acco) and rel_err(dxv[b, :249], dxo) < 1e-9
        go = dict(g); go.update(chi2_test=1, chi2_table=table)
        Ho, ro, Rdo, vio, vso = orc.gnss_rows(oc, go)                       # gated, compacted rows as the reference stacks them
        vidx, vsize, Hc, rc, Rdc = blocks[b]
        kept = np.flatnonzero(keep[b, :len(rc)])
        assert used[b] == len(ro) == len(kept) and np.allclose(rc[kept], ro, rtol=0, atol=0)      # the same rows survive
        outl = (5,) if b % 2 == 0 else (1, 6)
        assert all(keep[b, i] == 0 for i in outl)
        blk_ok = True
        if strong_reject and len(ro) <= 14:
            blk_ok = oc.whiten(vio, vso, Ho, ro, Rdo) < table[len(ro)]
        if blk_ok:
            dxo2, _ = oc.ekf_update(vio, vso, Ho, ro, Rdo)
            assert st[b] == 0 and rel_err(dxg[b, :243], dxo2) < 1e-9
        else:
            assert st[b] == capi.REJECTED and not dxg[b].any()
        P = ctx.cov_get(b)
        assert ctx.n(b) == 243 and rel_err(P, oc.P) < 1e-11 and np.array_equal(P, P.T)
    # block gate: rows with gross residuals, per-row gate off -> refused as a whole, covariance bit-identical
    Pb = [ctx.cov_get(b) for b in range(nb)]
    bad = []
    for vidx, vsize, Hc, rc, Rdc in blocks:
        bad.append((vidx, vsize, Hc[:12], np.full(12, 500.0), Rdc[:12]))
    dxg, used, keep, gam, st = ctx.gnss_update_batch(0, bad, table, gate_rows=False, strong_reject=True)
    assert (st == capi.REJECTED).all() and not dxg.any()
    for b in range(nb):
        assert np.array_equal(ctx.cov_get(b), Pb[b])
    # more than 14 rows: the reference skips the block gate (GnssUpdate.cpp:286) -> the update goes through
    full = [(v, s, H, np.full(len(r), 500.0), Rd) for v, s, H, r, Rd in blocks]
    dxg, used, keep, gam, st = ctx.gnss_update_batch(0, full, table, gate_rows=False, strong_reject=True)
    assert (used == 16).all() and (st == 0).all() and dxg.any()
    ctx.close()


@pytest.mark.parametrize("strong_reject,window", [(0, "small"), (1, "small"), (0, "large")])
def test_config3_in_frame_gnss_vs_oracle(orc, strong_reject, window):
    """The same frame + GNSS update with the GNSS stage marked in-frame (ingvio_gnss_opts::in_frame): ONE ingvio_frame_run applies
    both - the var_order columns of the MSCKF posterior are formed first (k_post_cols), the gates / S / gain read them, and the
    rank-16 downdate rides on the MSCKF write-back (k_info_apply<.., 16>): one read and one write of P.  Must give what the
    sequential calls give: covariance, both dx (the GNSS one indexed AFTER the marginalisation), kept rows, status.  `large`: a
    20-clone window cannot fold the update - ingvio_frame_run then applies it as its own pass, same results."""
    from ingvio_amd import capi, host, synth
    nb = 5
    C, F, n_lm = (11, 150, 52) if window == "small" else (20, 60, 20)
    n_max = ((21 + 6 + 3 * n_lm + 6 * C + 15) // 16) * 16
    ctx = capi.Context(batch=nb, n_max=n_max, c_max=C, f_max=F, m_max=64)
    cases = []
    for b in range(nb):
        flt, step, frame, info = synth.build_case(lambda P, b=b: capi.DeviceCov(ctx, b, P), host.imu_transition, seed=340 + b, F=F, C=C, n_landmarks=n_lm)
        g = synth.make_gnss(np.random.default_rng(940 + b), flt, outliers=(5,) if b % 2 == 0 else (1, 6))
        cases.append((flt, step, frame, info, g))
    priors = [ctx.cov_get(b) for b in range(nb)]
    table = cases[0][2]["chi2_table"]
    ctx.snapshot()
    ctx.frame_stage(0, [c[1] for c in cases], [c[2] for c in cases], cases[0][1]["sigma"], 1, 0.2, 0.2)
    blocks = [host.gnss_rows(c[4]) for c in cases]
    if strong_reject:                                    # filter 1: gross residuals on <= 14 rows, no row may pass on its own merit
        v, s_, H, r, Rd = blocks[1]
        blocks[1] = (v, s_, H[:12], np.full(12, 500.0), Rd[:12])
    ctx.gnss_stage(0, blocks, table, gate_rows=not strong_reject, strong_reject=bool(strong_reject), in_frame=True)
    with pytest.raises(capi.IngvioError):
        ctx.gnss_run()                                   # an in-frame stage is applied by frame_run only
    for rep in range(2):                                 # replayed from the same prior: identical
        ctx.frame_run(restore_prior=True)
        dxv, acc, rows = ctx.frame_fetch()
        dxg, used, keep, gam, st = ctx.gnss_fetch()
        n_post = 21 + 6 + 3 * n_lm + 6 * (C - 1)
        for b in range(nb):
            flt, step, frame, info, g = cases[b]
            oc = orc.Cov(priors[b], ld=n_max)
            dxo, acco, gamo, m = orc.frame_update(oc, step, frame, max_accept=0, compress_rule=1)
            assert np.array_equal(acc[b, :F], acco) and rel_err(dxv[b, :n_post + 6], dxo) < 1e-9 and rows[b] == 6 * C
            vidx, vsize, Hc, rc, Rdc = blocks[b]
            gate = not strong_reject
            if gate:
                go = dict(g); go.update(chi2_test=1, chi2_table=table)
                Ho, ro, Rdo, vio, vso = orc.gnss_rows(oc, go)
                kept = np.flatnonzero(keep[b, :len(rc)])
                assert used[b] == len(ro) == len(kept) and np.array_equal(rc[kept], ro)
            else:
                Ho, ro, Rdo, vio, vso = Hc, rc, Rdc, vidx, vsize
            blk_ok = not (strong_reject and len(ro) <= 14) or oc.whiten(vio, vso, Ho, ro, Rdo) < table[len(ro)]
            if blk_ok:
                dxo2, _ = oc.ekf_update(vio, vso, Ho, ro, Rdo)
                assert st[b] == 0 and rel_err(dxg[b, :n_post], dxo2) < 1e-9, (b, st[b])
                assert not dxg[b, n_post:].any()
            else:
                assert b == 1 and st[b] == capi.REJECTED and not dxg[b].any()
            P = ctx.cov_get(b)
            assert ctx.n(b) == n_post and rel_err(P, oc.P) < 1e-11 and np.array_equal(P, P.T), (b, rel_err(P, oc.P))
    ctx.close()


def test_strip_restore_survives_a_separate_gnss_pass(orc):
    """after a fused frame step the prior survives in the untouched ping-pong half up to the propagation strips, and a
    following frame_run(restore_prior) only rewrites those strips - also when a SEPARATE GNSS pass (k_downdate, in place in the live
    half) ran in between.  frame_run(restore) -> gnss_run -> frame_run(restore) must give exactly what a frame step from a full
    restore gives."""
    from ingvio_amd import capi, host, synth
    nb = 3
    ctx = capi.Context(batch=nb, n_max=256, c_max=11, f_max=150, m_max=64)
    cases = []
    for b in range(nb):
        flt, step, frame, info = synth.build_case(lambda P, b=b: capi.DeviceCov(ctx, b, P), host.imu_transition, seed=360 + b)
        cases.append((flt, step, frame, info, synth.make_gnss(np.random.default_rng(960 + b), flt)))
    table = cases[0][2]["chi2_table"]
    blocks = [host.gnss_rows(c[4]) for c in cases]
    ctx.snapshot()
    ctx.frame_stage(0, [c[1] for c in cases], [c[2] for c in cases], cases[0][1]["sigma"], 1, 0.2, 0.2)
    ctx.frame_run(restore_prior=True)
    ref = [ctx.cov_get(b) for b in range(nb)]                     # the frame step from the full restore
    ctx.gnss_stage(0, blocks, table, gate_rows=True)
    ctx.gnss_run()                                                # separate pass: changes the live half only
    assert not np.array_equal(ctx.cov_get(0), ref[0])
    ctx.frame_run(restore_prior=True)                             # strips only (the shortcut under test)
    for b in range(nb):
        assert np.array_equal(ctx.cov_get(b), ref[b]), b
    ctx.restore()                                                 # and against a FULL restore followed by the same step
    ctx.frame_run(restore_prior=True)
    for b in range(nb):
        assert np.array_equal(ctx.cov_get(b), ref[b]), b
    ctx.close()


def test_in_frame_gnss_stage_is_consumed_by_one_frame(orc):
    """an in-frame GNSS stage belongs to ONE frame.  A non-restoring ingvio_frame_run applies it and consumes it: the
    results stay fetchable, a second frame run (new frame staged, no new GNSS stage) must NOT apply the old rows again, and a fetch
    without any update having run on a fresh stage is refused instead of returning the MSCKF slots."""
    from ingvio_amd import capi, host, synth
    nb = 2
    ctx = capi.Context(batch=nb, n_max=256, c_max=11, f_max=150, m_max=64)
    cases = []
    for b in range(nb):
        flt, step, frame, info = synth.build_case(lambda P, b=b: capi.DeviceCov(ctx, b, P), host.imu_transition, seed=350 + b)
        cases.append((flt, step, frame, info, synth.make_gnss(np.random.default_rng(950 + b), flt)))
    priors = [ctx.cov_get(b) for b in range(nb)]
    table = cases[0][2]["chi2_table"]
    blocks = [host.gnss_rows(c[4]) for c in cases]
    ctx.frame_stage(0, [c[1] for c in cases], [c[2] for c in cases], cases[0][1]["sigma"], 1, 0.2, 0.2)
    ctx.gnss_stage(0, blocks, table, gate_rows=True, in_frame=True)
    with pytest.raises(capi.IngvioError):
        ctx.gnss_fetch()                                 # staged, nothing has run: there is nothing to fetch
    ctx.frame_run(restore_prior=False)
    dxg, used, keep, gam, st = ctx.gnss_fetch()          # this frame's GNSS results, although the stage is consumed
    assert (used > 0).all() and dxg.any()
    ocs = []
    for b in range(nb):
        flt, step, frame, info, g = cases[b]
        oc = orc.Cov(priors[b], ld=256)
        orc.frame_update(oc, step, frame, max_accept=0, compress_rule=1)
        go = dict(g); go.update(chi2_test=1, chi2_table=table)
        Ho, ro, Rdo, vio, vso = orc.gnss_rows(oc, go)
        dxo2, _ = oc.ekf_update(vio, vso, Ho, ro, Rdo)
        assert rel_err(ctx.cov_get(b), oc.P) < 1e-11 and rel_err(dxg[b, :243], dxo2) < 1e-9
        ocs.append(oc)
    with pytest.raises(capi.IngvioError):
        ctx.gnss_run()                                   # consumed: nothing is staged any more
    # the next frame on the moved-on state, no GNSS stage: the MSCKF step alone (the oracle applies no GNSS rows either)
    steps2, frames2 = [], []
    for b in range(nb):
        flt, step, frame, info, g = cases[b]
        steps2.append(step); frames2.append(frame)
    ctx.frame_stage(0, steps2, frames2, cases[0][1]["sigma"], 1, 0.2, 0.2)
    ctx.frame_run(restore_prior=False)
    for b in range(nb):
        flt, step, frame, info, g = cases[b]
        orc.frame_update(ocs[b], step, frame, max_accept=0, compress_rule=1)
        assert ctx.n(b) == ocs[b].n and rel_err(ctx.cov_get(b), ocs[b].P) < 1e-10, (b, rel_err(ctx.cov_get(b), ocs[b].P))
    ctx.close()


def test_gnss_stage_run_is_repeatable(orc):
    """ingvio_gnss_run reads the staged rows only: running it twice from the same restored covariance gives identical bits."""
    from ingvio_amd import capi, host, synth
    ctx = capi.Context(batch=2, n_max=256, c_max=11, f_max=150, m_max=64)
    blocks = []
    for b in range(2):
        flt, step, frame, info = synth.build_case(lambda P, b=b: capi.DeviceCov(ctx, b, P), host.imu_transition, seed=310 + b)
        blocks.append(host.gnss_rows(synth.make_gnss(np.random.default_rng(b), flt)))
    table = synth.chi2_table()
    ctx.snapshot()
    ctx.gnss_stage(0, blocks, table, gate_rows=True)
    outs = []
    for _ in range(2):
        ctx.restore()
        ctx.gnss_run()
        dx, used, keep, gam, st = ctx.gnss_fetch()
        outs.append((dx.copy(), used.copy(), keep.copy(), ctx.cov_get(0), ctx.cov_get(1)))
    for a, b in zip(outs[0], outs[1]):
        assert np.array_equal(a, b)
    assert (outs[0][1] == 15).all()
    ctx.close()


def test_rccl_world_size_1():
    """The timing barrier / MAX all-reduce / all-gather of the multi-GPU harness through RCCL itself (backend "nccl"), at the
    only world size a 1-GPU box can run, together with a covariance context in the same process.  Runs in a fresh interpreter:
    torch must initialise its HIP runtime before libingvio_hip.so is loaded (the order bench.py uses)."""
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29631",
               HSA_ENABLE_IPC_MODE_LEGACY="0", INGVIO_ROOT=ROOT)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "worker_rccl_ws1.py")], env=env, capture_output=True, text=True, timeout=600)
    print(r.stdout[-1500:]); print(r.stderr[-6000:])
    assert r.returncode == 0 and "RCCL_OK" in r.stdout


@pytest.mark.parametrize("scale", [1e-4, 1.0, 1e4])
def test_large_window_sigma_scale_sweep(orc, scale):
    """The same conditioning sweep on a 20-clone window: there stage 2 is the batched Cholesky form of kernels_chol.hip
    (P_cc = L L^T, W = L^T A L + s^2 I = L2 L2^T, no pivoting anywhere).  Accept masks equal to the oracle's, covariance within
    the BASELINE tolerance for every sigma x prior scale."""
    from ingvio_amd import capi, host, synth
    sigmas = [0.18, 0.08, 1e-2, 1e-3]
    C, F, n_lm = 20, 60, 20
    n_max = ((21 + 6 + 3 * n_lm + 6 * C + 15) // 16) * 16
    ctx = capi.Context(batch=len(sigmas), n_max=n_max, c_max=C, f_max=F, m_max=64)
    flt, step, frame, info = synth.build_case(lambda P: capi.DeviceCov(ctx, 0, P), host.imu_transition, seed=33, F=F, C=C, n_landmarks=n_lm)
    P0 = _prior_at_update(ctx, 0, flt, step)
    n = P0.shape[0]
    frames = []
    for b, s in enumerate(sigmas):
        ctx.cov_set(b, P0 * scale)
        f = dict(frame); f["noise"] = s
        frames.append(f)
    res = [ctx.msckf_update(b, frames[b]) for b in range(len(sigmas))]
    for b, s in enumerate(sigmas):
        oc = orc.Cov(P0 * scale, ld=n_max)
        dxo, acco, gamo, m = oc.msckf_update(frames[b], max_accept=0, compress_rule=1)
        Pg = ctx.cov_get(b)
        assert np.array_equal(res[b][1][0, :F], acco), (s, scale)
        e = rel_err(Pg, oc.P)
        print("large-window sweep scale %.0e sigma %-6g accepted %3d  cov vs oracle %.1e" % (scale, s, int(acco.sum()), e))
        assert e < 1e-6, (s, scale, e)
    ctx.close()
