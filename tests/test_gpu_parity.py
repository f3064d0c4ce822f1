"""GPU parity tests: the HIP path (through the C ABI, libingvio_hip.so) against the committed golden
vectors and the CPU oracle on identical seeded inputs.  Tolerances: covariance <= 1e-6 relative
(BASELINE.json north_star; observed ~1e-15), accept masks and clone indices bit-exact."""
import numpy as np
import pytest

from conftest import frame_from_golden, load_golden, rel_err

pytestmark = pytest.mark.gpu

TOL_COV = 1e-6          # BASELINE.json: "<= 1e-6 relative covariance error vs reference"
TIGHT = 1e-11           # what FP64 actually delivers; regressions show up here first


@pytest.fixture(scope="module", params=["factored", "dense"])
def ctx(request):
    """Both evaluation strategies of K4-K11 must reproduce the reference posterior."""
    from ingvio_amd import capi
    c = capi.Context(batch=4, n_max=256, c_max=11, f_max=160, m_max=64)
    c.set_method(request.param)
    yield c
    c.close()


def test_propagate_identity(ctx):
    """TestStateManager.cpp:95-137 through ingvio_propagate (incl. GNSS clock coupling + noise)."""
    z = load_golden("propagate")
    for c in ("c0", "c1", "c2"):
        ctx.cov_set(1, z[c + "_P"])
        ctx.propagate(1, z[c + "_Phi"], z[c + "_G"], float(z[c + "_dt"]), z[c + "_sigma"], 1, z[c + "_gnss_idx"],
                      float(z[c + "_scb"]), float(z[c + "_srw"]))
        P = ctx.cov_get(1)
        assert np.linalg.norm(P - z[c + "_Pn"]) < 1e-10 * max(1.0, np.linalg.norm(z[c + "_Pn"]))
        assert np.array_equal(P, P.T)
    ctx.cov_set(0, z["d_P"])
    ctx.propagate(0, z["d_Phi"], z["d_G"], float(z["d_dt"]), z["d_sigma"])
    assert rel_err(ctx.cov_get(0), z["d_Pn"]) < TIGHT


def test_propagate_fused_equals_stepwise(ctx, orc):
    """K1: one fused launch over k steps == k reference steps (ImuPropagator.cpp:246-289)."""
    rng = np.random.default_rng(1)
    for n, gi, en in ((60, [21, 40, -1, 22, 30], 1), (33, [-1] * 5, 0), (256, [25, -1, -1, -1, 21], 1)):
        A = rng.standard_normal((n, n)); P0 = A @ A.T / n + 0.05 * np.eye(n)
        k = 10
        Phis = np.eye(15) + 0.02 * rng.standard_normal((k, 15, 15)); Gs = rng.uniform(-1, 1, (k, 15, 12))
        dts = rng.uniform(0.004, 0.006, k)
        sig = [0.004, 0.08, 0.0002, 0.008]
        oc = orc.Cov(P0, ld=n + 8)
        for s in range(k):
            oc.propagate(Phis[s], Gs[s], dts[s], sig, en, gi, 0.2, 0.2)
        ctx.cov_set(2, P0)
        ctx.propagate(2, Phis, Gs, dts, sig, en, gi if en else None, 0.2, 0.2, fused=True)
        assert rel_err(ctx.cov_get(2), oc.P) < TIGHT


def test_batched_propagate_and_clone(ctx, orc):
    """All 4 filters of the context in one call each, different sizes and inputs."""
    rng = np.random.default_rng(2)
    ns = [27, 45, 81, 250]
    Ps, ocs = [], []
    Phis = np.eye(15) + 0.02 * rng.standard_normal((4, 3, 15, 15)); Gs = rng.uniform(-1, 1, (4, 3, 15, 12))
    dts = rng.uniform(0.004, 0.006, (4, 3)); Rs = np.stack([orc.gamma(rng.normal(size=3)) for _ in range(4)])
    sig = [0.004, 0.08, 0.0002, 0.008]
    for b, n in enumerate(ns):
        A = rng.standard_normal((n, n)); P0 = A @ A.T / n + 0.05 * np.eye(n)
        ctx.cov_set(b, P0)
        oc = orc.Cov(P0, ld=n + 8)
        for s in range(3):
            oc.propagate(Phis[b, s], Gs[b, s], dts[b, s], sig)
        oc.augment(Rs[b])
        ocs.append(oc)
    ctx.propagate(0, Phis, Gs, dts, sig, fused=True)
    idx = ctx.augment(0, Rs)
    assert list(idx) == ns                                  # new idx == old N, bit-exact
    for b in range(4):
        assert ctx.n(b) == ns[b] + 6
        assert rel_err(ctx.cov_get(b), ocs[b].P) < TIGHT


def test_augment_marginalize_append(ctx):
    z = load_golden("augment")
    ctx.cov_set(0, z["P"])
    idx = ctx.augment(0, z["R_i2w"])
    assert int(idx[0]) == z["P"].shape[0] and ctx.n(0) == z["P"].shape[0] + 6
    P = ctx.cov_get(0)
    assert np.linalg.norm(P - z["Pn"]) < 1e-8 and np.array_equal(P, P.T)       # TestStateManager.cpp:233
    z = load_golden("ekf")
    ctx.cov_set(0, z["P"])
    assert np.array_equal(ctx.marginal(0, z["vidx"], z["vsize"]), z["P_small"])
    ctx.marginalize(0, [int(z["marg_idx"])], int(z["marg_size"]))
    assert np.array_equal(ctx.cov_get(0), z["P_marg"]) and ctx.n(0) == z["P_marg"].shape[0]
    i2 = ctx.append_independent(0, 2.5 * np.eye(3))
    P = ctx.cov_get(0)
    assert int(i2[0]) == z["P_marg"].shape[0]
    assert np.array_equal(P[-3:, -3:], 2.5 * np.eye(3)) and not P[:-3, -3:].any() and not P[-3:, :-3].any()


def test_size_sequence_like_reference(ctx):
    """TestStateManager.cpp:64-93: 21 -> 22 -> 23 -> 24 -> 25 -> 24 -> 25 with bit-exact idx shifts."""
    ctx.cov_set(3, 1e-6 * np.eye(21))
    seq = []
    for var in (4.0, 4.0, 1.0, 1.0):
        seq.append(int(ctx.append_independent(3, np.array([[var]]))[0]))
    assert seq == [21, 22, 23, 24] and ctx.n(3) == 25
    ctx.marginalize(3, [21], 1)
    assert ctx.n(3) == 24
    assert int(ctx.append_independent(3, np.array([[6.0]]))[0]) == 24 and ctx.n(3) == 25
    d = np.diag(ctx.cov_get(3))
    assert np.allclose(d[21:], [4.0, 1.0, 1.0, 6.0])


def test_ekf_update_identity(ctx):
    """TestStateManager.cpp:478-557 — posterior == (I-KH)P for scalar / diagonal / full R."""
    z = load_golden("ekf")
    for R, Pn, dxg in ((0.5, z["Pn"], z["dx"]), (z["Rd"], z["Pn_d"], z["dx_d"]), (z["Rf"], z["Pn_f"], z["dx_f"])):
        ctx.cov_set(3, z["P"])
        dx, rc = ctx.ekf_update(3, z["vidx"], z["vsize"], z["H"], z["res"], R)
        P = ctx.cov_get(3)
        assert rc == 0 and np.linalg.norm(P - Pn) < 1e-8 and np.linalg.norm(dx - dxg) < 1e-8
        assert np.array_equal(P, P.T)
    ctx.cov_set(3, z["P"])
    assert abs(ctx.chi2_gamma(3, z["vidx"], z["vsize"], z["H"], z["res"], 0.5) - float(z["gamma"])) < 1e-10


def test_ekf_rejects_vars_outside_state(ctx):
    from ingvio_amd import capi
    z = load_golden("ekf")
    ctx.cov_set(3, z["P"])
    with pytest.raises(capi.IngvioError) as e:
        ctx.ekf_update(3, [0, 40], [9, 1], z["H"][:, :10], z["res"], 0.5)
    assert e.value.code == capi.E_NOT_IN_STATE


def test_gnss_update(ctx, orc):
    """K13: host-assembled psr/Doppler rows + ekfUpdate with diagonal R (GnssUpdate.cpp:148-290)."""
    z = load_golden("gnss")
    ctx.cov_set(3, z["P"])
    # per-row gate through the C ABI agrees with the golden row selection
    g = {k: z[k] for k in z.files}; g.update(idx_se23=0, chi2_test=1)
    H, res, Rd, vidx, vsize = orc.gnss_rows(orc.Cov(z["P"]), g)
    assert H.shape == z["H"].shape
    dx, rc = ctx.ekf_update(3, z["vidx"], z["vsize"], z["H"], z["res"], z["Rdiag"])
    assert rc == 0 and rel_err(ctx.cov_get(3), z["Pn"]) < TIGHT and rel_err(dx, z["dx"]) < 1e-9
    ctx.cov_set(3, z["P"])
    sub_i, sub_s = [0, int(z["idx_yof"]), int(z["idx_cb"][0])], [9, 1, 1]
    h = np.r_[z["H"][0, :9], 0.0, 1.0][None]
    gam = ctx.chi2_gamma(3, sub_i, sub_s, h, z["res"][:1], z["Rdiag"][:1])
    assert abs(gam - orc.Cov(z["P"]).whiten(sub_i, sub_s, h, z["res"][:1], z["Rdiag"][:1])) < 1e-10 * max(1.0, gam)


@pytest.mark.parametrize("name", ["stereo_ragged", "mono_ragged", "stereo_cap", "selected_q10", "keyframe_like"])
def test_msckf_small(ctx, name):
    """K3-K11 on ragged / mono / capped / selected-timestamp (Q10) frames vs LAPACK golden."""
    z = load_golden("msckf_small")
    fr = frame_from_golden(z, name + "_")
    kw = dict(zip(("max_accept", "compress_rule", "selected_variant"), [int(x) for x in z[name + "_kw"]]))
    ctx.cov_set(0, z[name + "_P"])
    dx, acc, gam, rows = ctx.msckf_update(0, fr, **kw)
    F = len(fr["dof"]); n = z[name + "_P"].shape[0]
    P = ctx.cov_get(0)
    assert np.array_equal(acc[0, :F], z[name + "_acc"])
    ev = ~np.isnan(z[name + "_gamma"])
    assert np.allclose(gam[0, :F][ev], z[name + "_gamma"][ev], rtol=1e-9)
    assert rel_err(P, z[name + "_Pn"]) < TIGHT and np.array_equal(P, P.T)
    assert np.linalg.norm(dx[0, :n] - z[name + "_dx"]) < 1e-9 * max(1.0, np.linalg.norm(z[name + "_dx"]))


def test_msckf_empty_and_all_rejected(ctx):
    from ingvio_amd import capi
    z = load_golden("msckf_small")
    fr = frame_from_golden(z, "stereo_ragged_")
    P0 = z["stereo_ragged_P"]
    ctx.cov_set(0, P0)
    fr2 = dict(fr); fr2["dof"] = np.zeros_like(fr["dof"])            # dof 0 -> every feature rejected
    dx, acc, gam, rows = ctx.msckf_update(0, fr2)
    F = len(fr['dof'])
    assert rows[0] == 0 and not acc[0, :F].any() and not dx[0, :P0.shape[0]].any()
    assert np.array_equal(ctx.cov_get(0), P0)                         # state untouched
    fr3 = dict(fr); fr3["obs_mask"] = np.zeros_like(fr["obs_mask"])   # no observations at all
    dx, acc, gam, rows = ctx.msckf_update(0, fr3)
    assert rows[0] == 0 and np.array_equal(ctx.cov_get(0), P0)
    assert capi.NO_ROWS == 1


def test_config2_frame_golden(ctx):
    """150 feats x 11 clones, literal N=87: update alone, as-written cap 20, then the whole frame
    (propagate x10 fused + clone + update + marginalise) for 4 filters at once."""
    z = load_golden("config2_n87")
    fr = frame_from_golden(z, "fr_")
    ctx.cov_set(1, z["P_pre_update"])
    dx, acc, gam, rows = ctx.msckf_update(1, fr)
    assert rows[0] == 66 and np.array_equal(acc[0, :150], z["acc"]) and np.array_equal(acc[0, :150] == 0, z["outlier"])
    assert np.allclose(gam[0, :150], z["gamma"], rtol=1e-8)
    assert rel_err(ctx.cov_get(1), z["Pn"]) < TIGHT and rel_err(dx[0, :87], z["dx"]) < 1e-9
    ctx.cov_set(1, z["P_pre_update"])
    dx, acc, gam, rows = ctx.msckf_update(1, fr, max_accept=20, compress_rule=0)
    assert np.array_equal(acc[0, :150], z["acc_aw"]) and rel_err(ctx.cov_get(1), z["Pn_aw"]) < 1e-8
    step = dict(Phi=list(z["step_Phi"]), G=list(z["step_G"]), dt=list(z["step_dt"]), sigma=list(z["step_sigma"]),
                R_i2w=z["step_R_i2w"], marg_idx=int(z["step_marg_idx"]))
    for b in range(4):
        ctx.cov_set(b, z["P_prior"])
    ctx.snapshot()
    ctx.frame_stage(0, [step] * 4, [fr] * 4, step["sigma"])
    for _ in range(2):                      # restore makes the step repeatable
        ctx.frame_run(restore_prior=True)
    dx, acc, rows = ctx.frame_fetch()
    for b in range(4):
        assert ctx.n(b) == 81 and rows[b] == 66
        assert rel_err(ctx.cov_get(b), z["P_final"]) < TIGHT and rel_err(dx[b, :87], z["dx"]) < 1e-9


@pytest.mark.parametrize("method", ["factored", "dense"])
def test_full_n249_batch_vs_oracle(orc, method):
    """Config 2 at the BASELINE nominal size: N=249 (6 GNSS scalars + 52 landmark blocks), 8
    different filters in one batch, prior built by the HIP path itself, vs the oracle."""
    from ingvio_amd import capi, host, synth
    ctx2 = capi.Context(batch=8, n_max=256, c_max=11, f_max=150, m_max=64)
    ctx2.set_method(method)
    cases = []
    for b in range(8):
        flt, step, frame, info = synth.build_case(lambda P, b=b: capi.DeviceCov(ctx2, b, P), host.imu_transition, seed=b)
        cases.append((flt, step, frame, info))
    priors = [ctx2.cov_get(b) for b in range(8)]
    ctx2.snapshot()
    ctx2.frame_stage(0, [c[1] for c in cases], [c[2] for c in cases], cases[0][1]["sigma"], 1, 0.2, 0.2)
    ctx2.frame_run(restore_prior=True)
    dx, acc, rows = ctx2.frame_fetch()
    for b in range(8):
        flt, step, frame, info = cases[b]
        assert info["N_update"] == 249 and info["marg_idx"] == 183
        oc = orc.Cov(priors[b], ld=256)
        dxo, acco, gamo, m = orc.frame_update(oc, step, frame, max_accept=0, compress_rule=1)
        assert ctx2.n(b) == 243 and rows[b] == 66 and np.array_equal(acc[b, :150], acco)
        assert np.array_equal(acco == 0, info["outlier"])
        P = ctx2.cov_get(b)
        assert rel_err(P, oc.P) < TIGHT and rel_err(dx[b, :249], dxo) < 1e-9
        assert np.array_equal(P, P.T) and np.diag(P).min() > 0
    # repeatability: the second of two back-to-back restores only rewrites the propagation strips (incl. the GNSS
    # clock rows/columns) of the untouched ping-pong half; the result must be bit-identical
    P_first = [ctx2.cov_get(b) for b in range(8)]
    ctx2.frame_run(restore_prior=True)
    ctx2.frame_run(restore_prior=True)
    for b in range(8):
        assert np.array_equal(ctx2.cov_get(b), P_first[b])
    ctx2.close()


@pytest.mark.parametrize("stereo", [True, False])
def test_full_size_ragged_random_anchor_vs_oracle(orc, stereo):
    """Full-size frames (150 features, 11 clones, N=249) with RAGGED observation sets and RANDOM anchors (an anchor
    need not be one of the observing clones), stereo and mono: whole frame pipeline vs the oracle."""
    from ingvio_amd import capi, host, synth
    nb = 4
    ctx2 = capi.Context(batch=nb, n_max=256, c_max=11, f_max=150, m_max=64)
    cases = []
    for b in range(nb):
        flt, step, frame, info = synth.build_case(lambda P, b=b: capi.DeviceCov(ctx2, b, P), host.imu_transition,
                                                  seed=40 + b, stereo=stereo)
        rng = np.random.default_rng(1000 + b)
        F, C = 150, 11
        mask = np.zeros(F, dtype=np.uint64); dof = np.zeros(F, dtype=np.int32)
        for j in range(F):
            k = int(rng.integers(5, C + 1))                       # 5..11 observing clones
            obs = np.sort(rng.choice(C, size=k, replace=False))
            mask[j] = np.uint64(sum(1 << int(o) for o in obs)); dof[j] = k - 1
        frame = dict(frame); frame["obs_mask"] = mask; frame["dof"] = dof
        frame["anchor"] = rng.integers(0, C, size=F).astype(np.int32)
        cases.append((flt, step, frame, info))
    priors = [ctx2.cov_get(b) for b in range(nb)]
    ctx2.snapshot()
    ctx2.frame_stage(0, [c[1] for c in cases], [c[2] for c in cases], cases[0][1]["sigma"], 1, 0.2, 0.2)
    ctx2.frame_run(restore_prior=True)
    dx, acc, rows = ctx2.frame_fetch()
    for b in range(nb):
        flt, step, frame, info = cases[b]
        oc = orc.Cov(priors[b], ld=256)
        dxo, acco, gamo, m = orc.frame_update(oc, step, frame, max_accept=0, compress_rule=1)
        assert np.array_equal(acc[b, :150], acco) and 0 < acco.sum()
        P = ctx2.cov_get(b)
        assert ctx2.n(b) == 243 and rel_err(P, oc.P) < TIGHT and rel_err(dx[b, :249], dxo) < 1e-8
        assert np.array_equal(P, P.T)
    ctx2.close()


@pytest.mark.parametrize("selected,F,C", [(0, 150, 11), (1, 149, 11), (0, 7, 11), (1, 13, 6), (0, 2, 3)])
def test_gate_mixed_groups_of_four_vs_oracle(orc, selected, F, C):
    """k_feat_gate5 handles FOUR features per wave: groups whose members have different observation sets (a pair lane's block of
    P is shared, every feature takes only the pairs its own mask holds), features without any observation, with a single one (no
    difference coordinates: gamma = |r_perp|^2 / s^2), a reference observation that is not window slot 0, feature counts that are
    not multiples of four, the Q10 (selected-timestamp) variant - gamma and the accept mask of every feature against the oracle's
    dense evaluation, then the posterior."""
    from ingvio_amd import capi, host, synth
    n_lm = 10
    n_max = ((21 + 6 + 3 * n_lm + 6 * C + 15) // 16) * 16
    ctx2 = capi.Context(batch=1, n_max=n_max, c_max=C, f_max=max(F, 4), m_max=64)
    flt, step, frame, info = synth.build_case(lambda P: capi.DeviceCov(ctx2, 0, P), host.imu_transition, seed=70 + F, F=F, C=C, n_landmarks=n_lm)
    # the update of build_case's frame is applied to the state AFTER propagate + clone: bring the device there
    for Phi, G, dt in zip(step["Phi"], step["G"], step["dt"]):
        flt.cov.propagate(Phi, G, dt, step["sigma"], step["enable_gnss"], step["gnss_idx"], step["sigma_cb"], step["sigma_rw"])
    flt.cov.augment(step["R_i2w"])
    P0 = ctx2.cov_get(0)
    rng = np.random.default_rng(5000 + F)
    mask = np.zeros(F, dtype=np.uint64); dof = np.zeros(F, dtype=np.int32)
    for j in range(F):
        kind = j % 8
        if kind == 0: obs = []                                                     # no observation at all
        elif kind == 1: obs = [int(rng.integers(0, C))]                            # one: nobs - 1 = 0 pivots
        elif kind == 2: obs = list(range(1, C))                                    # reference observation = slot 1, not 0
        elif kind == 3: obs = list(range(C))                                       # the full window
        else: obs = sorted(rng.choice(C, size=int(rng.integers(2, C + 1)), replace=False).tolist())
        mask[j] = np.uint64(sum(1 << int(o) for o in obs)); dof[j] = max(len(obs) - 1, 1)
    frame = dict(frame); frame["obs_mask"] = mask; frame["dof"] = dof
    frame["anchor"] = rng.integers(0, C, size=F).astype(np.int32)
    dx, acc, gam, rows = ctx2.msckf_update(0, frame, selected_variant=selected)
    oc = orc.Cov(P0, ld=n_max)
    dxo, acco, gamo, m = oc.msckf_update(frame, max_accept=0, compress_rule=1, selected_variant=selected)
    assert np.array_equal(acc[0, :F], acco), np.flatnonzero(acc[0, :F] != acco)
    g = gam[0, :F]
    empty = np.array([int(mk) == 0 for mk in mask])
    assert np.isnan(g[empty]).all() and not acc[0, :F][empty].any()
    ok = ~empty & np.isfinite(gamo)
    assert np.allclose(g[ok], gamo[ok], rtol=1e-8, atol=1e-10), np.abs(g[ok] - gamo[ok]).max()
    P = ctx2.cov_get(0)
    # (a frame whose only usable feature has ONE observation carries no information: dx is rounding noise on both sides)
    assert rel_err(P, oc.P) < TIGHT and np.linalg.norm(dx[0, :P0.shape[0]] - dxo) < 1e-8 * max(np.linalg.norm(dxo), 1e-6)
    ctx2.close()


@pytest.mark.parametrize("selected,F,C", [(0, 150, 11), (1, 149, 11), (0, 7, 9), (1, 13, 7), (0, 30, 8)])
def test_mono_gate_groups_of_four_vs_oracle(orc, selected, F, C):
    """k_feat_gate5m (round 6): the MONO gate of 7..11-clone windows, four features per wave - the measurement-space gate value as the
    last corner of a quasi-definite bordered LDL^T (2 nobs rows of S, the three columns of Hf as negative pivots, r as the border).
    Groups whose members have different observation sets, features with no / one observation (2 nobs - 3 <= 0: NaN, rejected, as the
    reference's size check), with two (one degree of freedom), an anchor that is or is not among the observing clones, feature counts
    that are not multiples of four, the Q10 (selected-timestamp) variant: gamma and the accept mask of every feature against the
    oracle's dense nullspace evaluation, then the posterior."""
    from ingvio_amd import capi, host, synth
    n_lm = 10
    n_max = ((21 + 6 + 3 * n_lm + 6 * C + 15) // 16) * 16
    ctx2 = capi.Context(batch=1, n_max=n_max, c_max=C, f_max=max(F, 4), m_max=64)
    flt, step, frame, info = synth.build_case(lambda P: capi.DeviceCov(ctx2, 0, P), host.imu_transition, seed=170 + F, F=F, C=C, n_landmarks=n_lm,
                                              stereo=False)
    for Phi, G, dt in zip(step["Phi"], step["G"], step["dt"]):
        flt.cov.propagate(Phi, G, dt, step["sigma"], step["enable_gnss"], step["gnss_idx"], step["sigma_cb"], step["sigma_rw"])
    flt.cov.augment(step["R_i2w"])
    P0 = ctx2.cov_get(0)
    rng = np.random.default_rng(6000 + F)
    mask = np.zeros(F, dtype=np.uint64); dof = np.zeros(F, dtype=np.int32)
    for j in range(F):
        kind = j % 8
        if kind == 0: obs = []                                                     # no observation at all
        elif kind == 1: obs = [int(rng.integers(0, C))]                            # one: 2 rows, nothing left after the projection
        elif kind == 2: obs = sorted(rng.choice(C, size=2, replace=False).tolist())  # two: one degree of freedom
        elif kind == 3: obs = list(range(C))                                       # the full window
        else: obs = sorted(rng.choice(C, size=int(rng.integers(2, C + 1)), replace=False).tolist())
        mask[j] = np.uint64(sum(1 << int(o) for o in obs)); dof[j] = max(2 * len(obs) - 3, 1)
    frame = dict(frame); frame["obs_mask"] = mask; frame["dof"] = dof
    frame["anchor"] = rng.integers(0, C, size=F).astype(np.int32)
    assert not frame["stereo"]
    dx, acc, gam, rows = ctx2.msckf_update(0, frame, selected_variant=selected)
    oc = orc.Cov(P0, ld=n_max)
    dxo, acco, gamo, m = oc.msckf_update(frame, max_accept=0, compress_rule=1, selected_variant=selected)
    assert np.array_equal(acc[0, :F], acco), np.flatnonzero(acc[0, :F] != acco)
    g = gam[0, :F]
    few = np.array([bin(int(mk)).count("1") < 2 for mk in mask])
    assert np.isnan(g[few]).all() and not acc[0, :F][few].any()
    ok = ~few & np.isfinite(gamo)
    assert ok.sum() >= F // 2 and np.allclose(g[ok], gamo[ok], rtol=1e-8, atol=1e-10), np.abs(g[ok] - gamo[ok]).max()
    P = ctx2.cov_get(0)
    assert rel_err(P, oc.P) < TIGHT and np.linalg.norm(dx[0, :P0.shape[0]] - dxo) < 1e-8 * max(np.linalg.norm(dxo), 1e-6)
    ctx2.close()


@pytest.mark.parametrize("C,stereo,method", [(16, True, "factored"), (16, False, "factored"), (6, True, "factored"),
                                             (6, False, "factored"), (13, True, "factored"), (4, True, "factored"),
                                             (12, True, "factored"), (12, False, "factored"),      # 12 clones: the symmetric solve's 72 class (round 6)
                                             (13, True, "dense"), (16, False, "dense"), (4, False, "dense")])
def test_window_size_classes_vs_oracle(orc, C, stereo, method):
    """The kernels are instantiated for window classes 6 / 11 / 16 clones: run whole frames at the class maxima and
    at in-between sizes (13 -> class 16, 4 -> class 6), ragged observations, vs the oracle."""
    from ingvio_amd import capi, host, synth
    nb, F = 3, 60
    n_gnss, n_lm = 6, 10
    N = 21 + n_gnss + 3 * n_lm + 6 * C
    ctx2 = capi.Context(batch=nb, n_max=((N + 15) // 16) * 16, c_max=C, f_max=F, m_max=64)
    ctx2.set_method(method)
    cases = []
    for b in range(nb):
        flt, step, frame, info = synth.build_case(lambda P, b=b: capi.DeviceCov(ctx2, b, P), host.imu_transition,
                                                  seed=70 + b, F=F, C=C, n_gnss=n_gnss, n_landmarks=n_lm, stereo=stereo)
        rng = np.random.default_rng(2000 + b)
        kmin = min(C, 5 if not stereo else 3)
        mask = np.zeros(F, dtype=np.uint64); dof = np.zeros(F, dtype=np.int32)
        for j in range(F):
            k = int(rng.integers(kmin, C + 1))
            obs = np.sort(rng.choice(C, size=k, replace=False))
            mask[j] = np.uint64(sum(1 << int(o) for o in obs)); dof[j] = k - 1
        frame = dict(frame); frame["obs_mask"] = mask; frame["dof"] = dof
        frame["anchor"] = rng.integers(0, C, size=F).astype(np.int32)
        cases.append((flt, step, frame, info))
    ld = ctx2.ldp
    priors = [ctx2.cov_get(b) for b in range(nb)]
    ctx2.snapshot()
    ctx2.frame_stage(0, [c[1] for c in cases], [c[2] for c in cases], cases[0][1]["sigma"], 1, 0.2, 0.2)
    ctx2.frame_run(restore_prior=True)
    dx, acc, rows = ctx2.frame_fetch()
    for b in range(nb):
        flt, step, frame, info = cases[b]
        oc = orc.Cov(priors[b], ld=ld)
        dxo, acco, gamo, m = orc.frame_update(oc, step, frame, max_accept=0, compress_rule=1)
        assert np.array_equal(acc[b, :F], acco)
        P = ctx2.cov_get(b)
        assert P.shape == oc.P.shape and rel_err(P, oc.P) < TIGHT and rel_err(dx[b, :N], dxo) < 1e-8
    ctx2.close()


def test_consecutive_frames_without_restore(orc):
    """Three frame steps in a row WITHOUT restoring the prior: the ping-pong halves alternate, so the fused
    update+marginalise and the zero-copy clone columns run from either half, and the covariance evolves; the oracle
    applies the same three frames sequentially."""
    from ingvio_amd import capi, host, synth
    nb, F, C = 3, 80, 11
    ctx2 = capi.Context(batch=nb, n_max=256, c_max=C, f_max=F, m_max=64)
    cases = []
    for b in range(nb):
        flt, step, frame, info = synth.build_case(lambda P, b=b: capi.DeviceCov(ctx2, b, P), host.imu_transition,
                                                  seed=90 + b, F=F, C=C)
        cases.append((flt, step, frame, info))
    ocs = [orc.Cov(ctx2.cov_get(b), ld=256) for b in range(nb)]
    ctx2.frame_stage(0, [c[1] for c in cases], [c[2] for c in cases], cases[0][1]["sigma"], 1, 0.2, 0.2)
    for it in range(3):
        ctx2.frame_run(restore_prior=False)
        dx, acc, rows = ctx2.frame_fetch()
        for b in range(nb):
            dxo, acco, gamo, m = orc.frame_update(ocs[b], cases[b][1], cases[b][2], max_accept=0, compress_rule=1)
            assert np.array_equal(acc[b, :F], acco), (it, b)
            assert ctx2.n(b) == 243 and rel_err(ctx2.cov_get(b), ocs[b].P) < 1e-10, (it, b)
            assert rel_err(dx[b, :249], dxo) < 1e-7
    ctx2.close()


def test_async_staging_pipeline(orc):
    """ingvio_frame_stage_async: the inputs of the next frame go to the second input set over the copy stream while the
    previous frame's kernels run.  Two different frames (and IMU steps) alternate through the pipeline
    run(i); stage_async(i+1); fetch(i); every fetched result equals the oracle's for the frame that was run, and a
    synchronous stage / staged triangulation afterwards still sees consistent inputs."""
    from ingvio_amd import capi, host, synth
    nb, F, C = 3, 80, 11
    ctx2 = capi.Context(batch=nb, n_max=256, c_max=C, f_max=F, m_max=64)
    sets = []
    for s0 in (300, 400):
        cases = []
        for b in range(nb):
            flt, step, frame, info = synth.build_case(lambda P, b=b: capi.DeviceCov(ctx2, b, P), host.imu_transition,
                                                      seed=s0 + b, F=F, C=C)
            cases.append((step, frame))
        sets.append(cases)
    # both frame sets act on the SAME prior (the one build_case of the last set left on the device)
    priors = [ctx2.cov_get(b) for b in range(nb)]
    ctx2.snapshot()
    want = []
    for cases in sets:
        res = []
        for b in range(nb):
            oc = orc.Cov(priors[b], ld=256)
            dxo, acco, gamo, m = orc.frame_update(oc, cases[b][0], cases[b][1], max_accept=0, compress_rule=1)
            res.append((oc.P, dxo, acco))
        want.append(res)
    stage = [ctx2.frame_stage_prepare(0, [c[0] for c in cases], [c[1] for c in cases], cases[0][0]["sigma"], 1, 0.2, 0.2, use_async=True)
             for cases in sets]
    stage[0]()
    for it in range(5):
        cur = it % 2
        ctx2.frame_run(restore_prior=True)
        stage[1 - cur]()                                  # next frame's inputs while this one computes
        dx, acc, rows = ctx2.frame_fetch()
        for b in range(nb):
            Pw, dxw, accw = want[cur][b]
            assert np.array_equal(acc[b, :F], accw), (it, b)
            assert rel_err(ctx2.cov_get(b), Pw) < TIGHT and rel_err(dx[b, :249], dxw) < 1e-8, (it, b)
    # a synchronous stage after an asynchronous one (in flight) lands in the right order
    ctx2.frame_stage(0, [c[0] for c in sets[0]], [c[1] for c in sets[0]], sets[0][0][0]["sigma"], 1, 0.2, 0.2)
    ctx2.frame_run(restore_prior=True)
    dx, acc, rows = ctx2.frame_fetch()
    for b in range(nb):
        assert np.array_equal(acc[b, :F], want[0][b][2]) and rel_err(ctx2.cov_get(b), want[0][b][0]) < TIGHT
    ctx2.close()


def test_frame_without_marginalisation_in_batch(orc):
    """One filter of the batch keeps its oldest clone (marg_idx = -1): it takes the in-place update path while its
    neighbours take the fused out-of-place one."""
    from ingvio_amd import capi, host, synth
    nb, F, C = 3, 80, 11
    ctx2 = capi.Context(batch=nb, n_max=256, c_max=C, f_max=F, m_max=64)
    cases = []
    for b in range(nb):
        flt, step, frame, info = synth.build_case(lambda P, b=b: capi.DeviceCov(ctx2, b, P), host.imu_transition,
                                                  seed=95 + b, F=F, C=C)
        step = dict(step)
        if b == 1:
            step["marg_idx"] = -1
        cases.append((flt, step, frame, info))
    ocs = [orc.Cov(ctx2.cov_get(b), ld=256) for b in range(nb)]
    ctx2.snapshot()
    ctx2.frame_stage(0, [c[1] for c in cases], [c[2] for c in cases], cases[0][1]["sigma"], 1, 0.2, 0.2)
    for it in range(2):                                   # twice: the second restore must be a full one here
        ctx2.frame_run(restore_prior=True)
    dx, acc, rows = ctx2.frame_fetch()
    for b in range(nb):
        dxo, acco, gamo, m = orc.frame_update(ocs[b], cases[b][1], cases[b][2], max_accept=0, compress_rule=1)
        assert np.array_equal(acc[b, :F], acco)
        assert ctx2.n(b) == (249 if b == 1 else 243)
        assert rel_err(ctx2.cov_get(b), ocs[b].P) < TIGHT and rel_err(dx[b, :249], dxo) < 1e-8
    ctx2.close()


@pytest.mark.parametrize("C,stereo,c_max", [(20, True, 20), (30, True, 30), (35, False, 35), (17, False, 17), (36, True, 36), (22, True, 30), (19, False, 36)])
def test_large_window_vs_oracle(orc, C, stereo, c_max):
    """Windows of 17..36 clones (the reference's shipped configs: 21..35) take the large-window kernels
    (kernels_bigwin.hip): whole frames, ragged observations, random anchors, vs the oracle.  c_max > C: a window that is not
    yet full runs in the context's class (gate class, padded solve size)."""
    from ingvio_amd import capi, host, synth
    nb, F = 2, 48
    n_gnss, n_lm = 6, 4
    N = 21 + n_gnss + 3 * n_lm + 6 * C
    ctx2 = capi.Context(batch=nb, n_max=((N + 15) // 16) * 16, c_max=c_max, f_max=F, m_max=64)
    cases = []
    for b in range(nb):
        flt, step, frame, info = synth.build_case(lambda P, b=b: capi.DeviceCov(ctx2, b, P), host.imu_transition,
                                                  seed=300 + b, F=F, C=C, n_gnss=n_gnss, n_landmarks=n_lm, stereo=stereo)
        rng = np.random.default_rng(3000 + b)
        mask = np.zeros(F, dtype=np.uint64); dof = np.zeros(F, dtype=np.int32)
        for j in range(F):
            k = int(rng.integers(5, C + 1))
            obs = np.sort(rng.choice(C, size=k, replace=False))
            mask[j] = np.uint64(sum(1 << int(o) for o in obs)); dof[j] = k - 1
        frame = dict(frame); frame["obs_mask"] = mask; frame["dof"] = dof
        frame["anchor"] = rng.integers(0, C, size=F).astype(np.int32)
        cases.append((flt, step, frame, info))
    ld = ctx2.ldp
    priors = [ctx2.cov_get(b) for b in range(nb)]
    ctx2.snapshot()
    ctx2.frame_stage(0, [c[1] for c in cases], [c[2] for c in cases], cases[0][1]["sigma"], 1, 0.2, 0.2)
    for _ in range(2):
        ctx2.frame_run(restore_prior=True)
    dx, acc, rows = ctx2.frame_fetch()
    for b in range(nb):
        flt, step, frame, info = cases[b]
        oc = orc.Cov(priors[b], ld=ld)
        dxo, acco, gamo, m = orc.frame_update(oc, step, frame, max_accept=0, compress_rule=1)
        assert np.array_equal(acc[b, :F], acco) and acco.sum() > 0
        P = ctx2.cov_get(b)
        assert rows[b] == 6 * C and P.shape == oc.P.shape
        assert rel_err(P, oc.P) < 1e-10 and rel_err(dx[b, :N], dxo) < 1e-7 and np.array_equal(P, P.T)
    ctx2.close()


@pytest.mark.parametrize("n_lm", [0, 200])
def test_config5_stress_vs_oracle(orc, n_lm):
    """BASELINE config 5: 300 features x 30 clones, stereo (rho = 117 rows per feature, n = 180 columns), at the
    literal state size (N = 207 with the six GNSS scalars) and at the nominal N ~ 800 (+200 landmark blocks):
    whole frame vs the oracle, incl. propagation over several 256-row tiles."""
    from ingvio_amd import capi, host, synth
    C, F, n_gnss = 30, 300, 6
    N = 21 + n_gnss + 3 * n_lm + 6 * C
    ctx2 = capi.Context(batch=1, n_max=((N + 15) // 16) * 16, c_max=C, f_max=F, m_max=64)
    flt, step, frame, info = synth.build_case(lambda P: capi.DeviceCov(ctx2, 0, P), host.imu_transition, seed=5, F=F, C=C,
                                              n_gnss=n_gnss, n_landmarks=n_lm, stereo=True)
    prior = ctx2.cov_get(0)
    oc = orc.Cov(prior, ld=ctx2.ldp)
    dxo, acco, gamo, m = orc.frame_update(oc, step, frame, max_accept=0, compress_rule=1)
    ctx2.snapshot()
    ctx2.frame_stage(0, [step], [frame], step["sigma"], 1, 0.2, 0.2)
    for _ in range(2):
        ctx2.frame_run(restore_prior=True)
    dx, acc, rows = ctx2.frame_fetch()
    P = ctx2.cov_get(0)
    assert np.array_equal(acc[0, :F], acco) and rows[0] == 6 * C and ctx2.n(0) == N - 6
    assert rel_err(P, oc.P) < 1e-10 and rel_err(dx[0, :N], dxo) < 1e-7 and np.array_equal(P, P.T)
    ctx2.close()


@pytest.mark.parametrize("selected,stereo,nb", [(1, True, 1), (0, True, 1), (1, False, 1), (1, True, 3)])
def test_twelve_clone_window_vs_oracle(orc, selected, stereo, nb):
    """An 11-pose window in SLIDING-WINDOW mode (is_key_frame 0) holds 12 clones at update time (SwMargUpdate.cpp:412-419, State.h:83-92):
    window class 72 of k_info_solve (66 columns gauge-reduced for the RemoveLost form, 72 unreduced for the Selected-timestamp form
    with quirk Q10) and of the apply kernels - one filter (the flat few-filter kernels) and a small batch, two consecutive updates on
    the same covariance, posterior and dx against the oracle."""
    from ingvio_amd import capi, host, synth
    C, F, n_gnss, n_lm = 12, 60, 6, 4
    N = 21 + n_gnss + 3 * n_lm + 6 * C
    ctx2 = capi.Context(batch=nb, n_max=((N + 15) // 16) * 16, c_max=C, f_max=F, m_max=64)
    cases = []
    for b in range(nb):
        flt, step, frame, info = synth.build_case(lambda P, b=b: capi.DeviceCov(ctx2, b, P), host.imu_transition, seed=170 + b, F=F, C=C,
                                                  n_gnss=n_gnss, n_landmarks=n_lm, stereo=stereo)
        rng = np.random.default_rng(3000 + b)
        mask = np.zeros(F, dtype=np.uint64); dof = np.zeros(F, dtype=np.int32)
        for j in range(F):
            # the Selected-timestamp updates see 3 stamps of the window (frame_select_interval 5 at 11 poses), RemoveLost whole tracks
            k = int(rng.integers(3 if stereo else 5, 5 if (selected and stereo) else C + 1))
            obs = np.sort(rng.choice(C, size=k, replace=False))
            mask[j] = np.uint64(sum(1 << int(o) for o in obs)); dof[j] = k - 1
        frame = dict(frame); frame["obs_mask"] = mask; frame["dof"] = dof
        frame["anchor"] = np.array([int([o for o in range(C) if (int(mask[j]) >> o) & 1][0]) if j % 2 else int(rng.integers(0, C)) for j in range(F)],
                                   dtype=np.int32)                       # half the anchors observe the feature themselves (Q10 bites there)
        cases.append((step, frame))
    ocs = [orc.Cov(ctx2.cov_get(b), ld=ctx2.ldp) for b in range(nb)]
    ctx2.frame_stage(0, [c[0] for c in cases], [c[1] for c in cases], cases[0][0]["sigma"], 1, 0.2, 0.2, max_accept=0, compress_rule=1,
                     selected_variant=selected)
    for it in range(2):
        ctx2.frame_run(restore_prior=False)
        dx, acc, rows = ctx2.frame_fetch()
        for b in range(nb):
            dxo, acco, gamo, m = orc.frame_update(ocs[b], cases[b][0], cases[b][1], max_accept=0, compress_rule=1, selected_variant=selected)
            assert np.array_equal(acc[b, :F], acco) and acco.sum() > 10
            assert rel_err(ctx2.cov_get(b), ocs[b].P) < 1e-10 and rel_err(dx[b, :N], dxo) < 1e-7, (it, b)
    ctx2.close()


@pytest.mark.parametrize("C,stereo,cap", [(24, True, 0), (27, False, 0), (22, True, 12)])
def test_large_window_selected_variant_and_cap(orc, C, stereo, cap):
    """Large windows with the Selected-timestamp quirk Q10 (anchor block assigned) and the accepted-feature cap,
    two consecutive frames without restore (both ping-pong halves), vs the oracle."""
    from ingvio_amd import capi, host, synth
    F, n_gnss, n_lm = 40, 6, 3
    N = 21 + n_gnss + 3 * n_lm + 6 * C
    ctx2 = capi.Context(batch=1, n_max=((N + 15) // 16) * 16, c_max=C, f_max=F, m_max=64)
    flt, step, frame, info = synth.build_case(lambda P: capi.DeviceCov(ctx2, 0, P), host.imu_transition, seed=77, F=F, C=C,
                                              n_gnss=n_gnss, n_landmarks=n_lm, stereo=stereo)
    rng = np.random.default_rng(77)
    mask = np.zeros(F, dtype=np.uint64); dof = np.zeros(F, dtype=np.int32)
    for j in range(F):
        k = int(rng.integers(6, C + 1))
        obs = np.sort(rng.choice(C, size=k, replace=False))
        mask[j] = np.uint64(sum(1 << int(o) for o in obs)); dof[j] = k - 1
    frame = dict(frame); frame["obs_mask"] = mask; frame["dof"] = dof
    frame["anchor"] = np.array([int(np.sort([o for o in range(C) if (int(mask[j]) >> o) & 1])[j % 3]) if j % 2 else int(rng.integers(0, C))
                                for j in range(F)], dtype=np.int32)          # half the anchors are observing clones
    oc = orc.Cov(ctx2.cov_get(0), ld=ctx2.ldp)
    ctx2.frame_stage(0, [step], [frame], step["sigma"], 1, 0.2, 0.2, max_accept=cap, compress_rule=1, selected_variant=1)
    for it in range(2):
        ctx2.frame_run(restore_prior=False)
        dx, acc, rows = ctx2.frame_fetch()
        dxo, acco, gamo, m = orc.frame_update(oc, step, frame, max_accept=cap, compress_rule=1, selected_variant=1)
        assert np.array_equal(acc[0, :F], acco) and (cap == 0 or acco.sum() <= cap)
        assert rel_err(ctx2.cov_get(0), oc.P) < 1e-10 and rel_err(dx[0, :N], dxo) < 1e-7
    ctx2.close()


@pytest.mark.parametrize("C,stereo", [(22, True), (28, False)])
def test_large_window_update_in_place(orc, C, stereo):
    """Large window without marginalisation (marg_idx = -1): the update runs IN PLACE and needs the copy of the clone
    columns; one of the two filters of the batch, the other takes the fused out-of-place path.  Vs the oracle."""
    from ingvio_amd import capi, host, synth
    nb, F, n_gnss, n_lm = 2, 40, 6, 2
    N = 21 + n_gnss + 3 * n_lm + 6 * C
    ctx2 = capi.Context(batch=nb, n_max=((N + 15) // 16) * 16, c_max=C, f_max=F, m_max=64)
    cases = []
    for b in range(nb):
        flt, step, frame, info = synth.build_case(lambda P, b=b: capi.DeviceCov(ctx2, b, P), host.imu_transition,
                                                  seed=500 + b, F=F, C=C, n_gnss=n_gnss, n_landmarks=n_lm, stereo=stereo)
        step = dict(step)
        if b == 0:
            step["marg_idx"] = -1
        cases.append((flt, step, frame, info))
    ocs = [orc.Cov(ctx2.cov_get(b), ld=ctx2.ldp) for b in range(nb)]
    ctx2.snapshot()
    ctx2.frame_stage(0, [c[1] for c in cases], [c[2] for c in cases], cases[0][1]["sigma"], 1, 0.2, 0.2)
    for _ in range(2):
        ctx2.frame_run(restore_prior=True)
    dx, acc, rows = ctx2.frame_fetch()
    for b in range(nb):
        dxo, acco, gamo, m = orc.frame_update(ocs[b], cases[b][1], cases[b][2], max_accept=0, compress_rule=1)
        assert np.array_equal(acc[b, :F], acco) and rows[b] == 6 * C
        assert ctx2.n(b) == (N if b == 0 else N - 6)
        assert rel_err(ctx2.cov_get(b), ocs[b].P) < 1e-10 and rel_err(dx[b, :N], dxo) < 1e-7
    ctx2.close()


def test_qr_compress(ctx):
    """K7 on an explicit H_large (the SPQR call sites): H_thin^T H_thin == H^T H, H_thin upper triangular;
    also on a rank-deficient matrix (Q9: rank n-6)."""
    rng = np.random.default_rng(3)
    A = rng.standard_normal((500, 66)); b = rng.standard_normal(500)
    Ht, rt = ctx.qr_compress(A, b)
    assert rel_err(Ht.T @ Ht, A.T @ A) < 1e-12 and rel_err(Ht.T @ rt, A.T @ b) < 1e-12
    assert not np.tril(Ht, -1).any()
    A[:, 60:] = A[:, :6] @ rng.standard_normal((6, 6))
    Ht, rt = ctx.qr_compress(A, b)
    assert rel_err(Ht.T @ Ht, A.T @ A) < 1e-12 and np.isfinite(Ht).all()
    A1 = rng.standard_normal((7, 12)); b1 = rng.standard_normal(7)          # fewer rows than columns
    Ht, rt = ctx.qr_compress(A1, b1)
    assert rel_err(Ht.T @ Ht, A1.T @ A1) < 1e-12


def test_ekf_update_batch_vs_oracle(orc):
    """ingvio_ekf_update_batch: one launch for the generic update of several filters (the GNSS update of a batch) with
    different var_orders, row counts and diagonal noise per filter == the oracle's ekfUpdate filter by filter."""
    from ingvio_amd import capi
    nb = 5
    ctx3 = capi.Context(batch=nb + 1, n_max=128, c_max=11, f_max=8, m_max=64)
    rng = np.random.default_rng(77)
    blocks, want = [], []
    for b in range(nb):
        n = 21 + 6 + 6 * (b + 1)
        A = rng.standard_normal((n, n)); P0 = 1e-2 * (A @ A.T / n + 0.1 * np.eye(n))
        ctx3.cov_set(b + 1, P0)
        m = 4 + 3 * b
        vo = [0, 21 + b % 3, 27 + 6 * (b % 2)]; vs = [9, 1, 6]
        H = rng.standard_normal((m, 16)); r = 0.1 * rng.standard_normal(m); R = rng.uniform(0.5, 2.0, m)
        blocks.append((vo, vs, H, r, R))
        oc = orc.Cov(P0); dxo, _ = oc.ekf_update(vo, vs, H, r, R)
        want.append((oc.P, dxo, n))
    dx, st = ctx3.ekf_update_batch(1, blocks, diag=True)
    assert not st.any()
    for b in range(nb):
        Pw, dxw, n = want[b]
        assert rel_err(ctx3.cov_get(b + 1), Pw) < TIGHT and rel_err(dx[b, :n], dxw) < 1e-9
    # scalar noise, and a variable outside one filter's state
    blocks_s = [(vo, vs, H, r, 0.7) for (vo, vs, H, r, R) in blocks]
    dx, st = ctx3.ekf_update_batch(1, blocks_s, diag=False)
    assert np.isfinite(dx).all()
    bad = list(blocks); bad[0] = ([0, 500], [9, 1], np.zeros((2, 10)), np.zeros(2), np.ones(2))
    with pytest.raises(capi.IngvioError) as e:
        ctx3.ekf_update_batch(1, bad, diag=True)
    assert e.value.code == capi.E_NOT_IN_STATE
    ctx3.close()


@pytest.mark.parametrize("m,n", [(500, 65), (300, 130), (1200, 216), (40, 100), (97, 97), (2100, 8), (1600, 45), (6144, 25)])
def test_qr_compress_general_vs_oracle(orc, m, n):
    """Shapes beyond the 96-column TSQR (any n, windows above 16 clones, m < n): blocked Householder QR (kernels_qr.hip)
    with the oracle's reflector convention, so R and Q^T res agree entry by entry, not only up to row signs."""
    from ingvio_amd import capi
    ctx3 = capi.Context(batch=1, n_max=64, c_max=11, f_max=8, m_max=64)
    rng = np.random.default_rng(m * 1000 + n)
    A = rng.standard_normal((m, n)); b = rng.standard_normal(m)
    if n == 130:
        A[:, 100:106] = A[:, :6] @ rng.standard_normal((6, 6))             # rank-deficient (Q9)
    Ht, rt = ctx3.qr_compress(A, b)
    Ro, ro = orc.qr_compress(A, b)
    k = min(m, n)
    scale = np.linalg.norm(Ro)
    if n != 130:      # rank-deficient: the reflector of a numerically zero column is decided by rounding noise, R is not unique
        if m >= 6 * n:   # tall: Cholesky-QR (positive diagonal) - R is unique up to the sign of each row
            sg = np.sign(np.diag(Ro[:k])) * np.sign(np.diag(Ht[:k]))
            Ro = Ro.copy(); ro = ro.copy(); Ro[:k] *= sg[:, None]; ro[:k] *= sg
        assert np.linalg.norm(Ht[:k] - Ro[:k]) < 1e-11 * scale and np.linalg.norm(rt[:k] - ro[:k]) < 1e-11 * max(1.0, np.linalg.norm(ro))
        if m >= 6 * n:
            ctx3.set_qr_method("householder")             # the same shape through the Householder route: the oracle's convention, entry by entry
            Hh, rh = ctx3.qr_compress(A, b)
            Ro2, ro2 = orc.qr_compress(A, b)
            assert np.linalg.norm(Hh[:k] - Ro2[:k]) < 1e-11 * scale and np.linalg.norm(rh[:k] - ro2[:k]) < 1e-11 * max(1.0, np.linalg.norm(ro2))
    assert not np.tril(Ht, -1).any() and not Ht[k:].any() and np.isfinite(Ht).all()
    assert rel_err(Ht.T @ Ht, A.T @ A) < 1e-12 and rel_err(Ht.T @ rt, A.T @ b) < 1e-12
    ctx3.close()


def test_qr_compress_tall_literal_config5_shape(orc):
    """The literal stacked shape of BASELINE config 5: 300 features x 30 clones, stereo -> 35100 x 180 (RemoveLostUpdate.cpp:376-397).
    Taller than the panel kernel's 6144 rows: factorised in row chunks.  R and Q^T r against the oracle's one-shot Householder up
    to row signs; the call must not touch any filter of the context."""
    from ingvio_amd import capi
    ctx3 = capi.Context(batch=2, n_max=64, c_max=11, f_max=8, m_max=64)
    rng = np.random.default_rng(35100)
    P = np.eye(30) * 0.5 + 0.01
    ctx3.cov_set(0, P); ctx3.cov_set(1, 2 * P)
    m, n = 35100, 180
    A = rng.standard_normal((m, n)) * np.logspace(0, -2, n); b = rng.standard_normal(m)
    Ht, rt = ctx3.qr_compress(A, b)
    assert not np.tril(Ht, -1).any() and np.isfinite(Ht).all()
    assert rel_err(Ht.T @ Ht, A.T @ A) < 1e-12 and rel_err(Ht.T @ rt, A.T @ b) < 1e-11
    Ro, ro = orc.qr_compress(A, b)
    sg = np.sign(np.diag(Ro[:n])) * np.sign(np.diag(Ht))
    assert np.linalg.norm(Ht - sg[:, None] * Ro[:n]) < 1e-10 * np.linalg.norm(Ro[:n]) and np.linalg.norm(rt - sg * ro[:n]) < 1e-10 * np.linalg.norm(ro[:n])
    assert np.array_equal(ctx3.cov_get(0), P) and np.array_equal(ctx3.cov_get(1), 2 * P)
    Ht2, rt2 = ctx3.qr_compress(A, b)                                      # cached graph: bit-identical on replay
    assert np.array_equal(Ht, Ht2) and np.array_equal(rt, rt2)
    ctx3.close()


def test_qr_compress_stress_shape():
    """BASELINE config 5's pure-kernel shape: dense 6000 x 800, cond ~ 1e3.  Size-independent properties + LAPACK's R up
    to row signs."""
    from ingvio_amd import capi
    ctx3 = capi.Context(batch=1, n_max=64, c_max=11, f_max=8, m_max=64)
    rng = np.random.default_rng(5)
    m, n = 6000, 800
    U, _ = np.linalg.qr(rng.standard_normal((m, n)))
    V, _ = np.linalg.qr(rng.standard_normal((n, n)))
    A = (U * np.logspace(0, -3, n)) @ V.T                                   # singular values 1 .. 1e-3
    b = rng.standard_normal(m)
    Ht, rt = ctx3.qr_compress(A, b)
    assert not np.tril(Ht, -1).any()
    assert rel_err(Ht.T @ Ht, A.T @ A) < 1e-12 and rel_err(Ht.T @ rt, A.T @ b) < 1e-11
    Rl = np.linalg.qr(A, mode="r")
    sg = np.sign(np.diag(Rl)) * np.sign(np.diag(Ht))
    assert np.linalg.norm(Ht - sg[:, None] * Rl) < 1e-9 * np.linalg.norm(Rl)   # cond 1e3: R is determined to ~1e-13 * cond
    ctx3.close()


def test_round_trip_properties_full_size():
    """Size-independent properties at the bench's full size (512 x N=249 would be the bench; here 32):
    the posterior is symmetric, never larger than the prior on the diagonal of the updated clones,
    restoring the prior makes the step idempotent, and trace decreases."""
    from ingvio_amd import capi, host, synth
    B = 32
    ctx3 = capi.Context(batch=B, n_max=256, c_max=11, f_max=150, m_max=64)
    cases = [synth.build_case(lambda P, b=b: capi.DeviceCov(ctx3, b, P), host.imu_transition, seed=100 + b) for b in range(B)]
    ctx3.snapshot()
    ctx3.frame_stage(0, [c[1] for c in cases], [c[2] for c in cases], cases[0][1]["sigma"], 1, 0.2, 0.2)
    ctx3.frame_run(restore_prior=True)
    P1 = [ctx3.cov_get(b) for b in range(B)]
    dx1, acc1, rows1 = ctx3.frame_fetch()
    ctx3.frame_run(restore_prior=True)
    dx2, acc2, rows2 = ctx3.frame_fetch()
    for b in range(B):
        P2 = ctx3.cov_get(b)
        assert np.array_equal(P1[b], P2)                # bitwise repeatable
        assert np.array_equal(P2, P2.T) and np.linalg.eigvalsh(P2).min() > -1e-10
    assert np.array_equal(dx1, dx2) and np.array_equal(acc1, acc2)
    ctx3.close()
