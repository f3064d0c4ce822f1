"""SLAM-landmark path (delayed initialisation, anchor change, landmark rows).

CPU part: the oracle's restatement of StateManager::addVariableDelayedInvertible / addVariableDelayed / replaceVarLinear
(StateManager.cpp:461-693) is pinned by the identities the reference's own gtests assert (TestStateManager.cpp:594-7xx,
AddDelayedTest.addVarInv / addVar), evaluated here independently with numpy, plus a numpy/LAPACK transcription for
general shapes; the landmark Jacobians (LandmarkUpdate.cpp:521-686) have no reference test and are checked against finite
differences of the measurement under the filter's own retraction (quirk Q12 asserted explicitly).
GPU part: the same operations through the C ABI vs the oracle."""
import numpy as np
import pytest

from oracle import oracle as orc


def spd(n, rng, scale=1.0):
    A = rng.standard_normal((n, n))
    return scale * (A @ A.T / n + 0.1 * np.eye(n))


def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


def rot(rng, mag=1.0):
    return orc.gamma(mag * rng.standard_normal(3), 0).reshape(3, 3)


# ---- AddDelayedTest.addVarInv (TestStateManager.cpp:594-642) --------------------------------------------------------
def test_add_var_invertible_reference_identities():
    rng = np.random.default_rng(1)
    P0 = spd(21, rng)
    H_old = rng.uniform(-1, 1, (1, 9)); H_new = np.array([[1.0]]); noise = 2.0
    c = orc.Cov(P0)
    idx = c.add_variable_delayed_invertible([0], [9], H_old, H_new, noise)
    assert idx == 21 and c.n == 22                                   # ASSERT_EQ(curr_cov_size, cols+1), idx == old rows
    P1 = c.P
    x1 = H_old @ P0[:9, :9] @ H_old.T
    assert abs(P1[21, 21] - (x1[0, 0] + noise ** 2) / H_new[0, 0] ** 2) < 1e-10          # reference: 1e-8
    H_large = np.zeros((1, 21)); H_large[:, :9] = H_old
    assert np.linalg.norm(P1[:21, 21:22] - (-P0 @ H_large.T / H_new[0, 0])) < 1e-10
    assert np.linalg.norm(P1[:21, :21] - P0) < 1e-14 and np.array_equal(P1, P1.T)


# ---- AddDelayedTest.addVar (TestStateManager.cpp:644-7xx) -----------------------------------------------------------
def test_add_var_delayed_reference_identities():
    rng = np.random.default_rng(2)
    P0 = spd(21, rng)
    H_old = rng.uniform(-1, 1, (2, 9)); H_new = np.array([[1.0], [1.0]]); res = rng.uniform(-1, 1, 2); noise = 2.0
    c = orc.Cov(P0)
    added, dx, chi2 = c.add_variable_delayed([0], [9], H_old, H_new, res, noise, 1.0, False)
    assert added and c.n == 22
    QT = np.array([[np.sqrt(2) / 2, np.sqrt(2) / 2], [-np.sqrt(2) / 2, np.sqrt(2) / 2]])
    r2, Ho2, Hn2 = QT @ res, QT @ H_old, QT @ H_new
    x1 = Ho2[0:1] @ P0[:9, :9] @ Ho2[0:1].T
    Pb = np.zeros((22, 22)); Pb[:21, :21] = P0
    Pb[21, 21] = (x1[0, 0] + noise ** 2) / Hn2[0, 0] ** 2
    Hl = np.zeros((1, 21)); Hl[:, :9] = Ho2[0:1]
    Pb[:21, 21:22] = -P0 @ Hl.T / Hn2[0, 0]; Pb[21:22, :21] = Pb[:21, 21:22].T
    Hu = np.zeros((1, 22)); Hu[:, :9] = Ho2[1:2]
    S = Hu @ Pb @ Hu.T + noise ** 2 * np.eye(1)
    K = Pb @ Hu.T @ np.linalg.inv(S)
    Pa = Pb - K @ Hu @ Pb
    assert np.linalg.norm(c.P - 0.5 * (Pa + Pa.T)) < 1e-10            # reference: 1e-8
    assert np.linalg.norm(dx - (K @ r2[1:2]).ravel()) < 1e-12


def _numpy_add_delayed(P, cols, H_old, H_new, res, noise):
    """independent transcription with LAPACK QR (any orthogonal Q gives the same posterior)"""
    m, s = H_new.shape
    Q, _ = np.linalg.qr(H_new, mode="complete")
    Hn, Ho, r = Q.T @ H_new, Q.T @ H_old, Q.T @ res
    n = P.shape[0]
    Hfull = np.zeros((m, n)); Hfull[:, cols] = Ho
    Hx, Hf, Hu = Hfull[:s], Hn[:s, :s], Hfull[s:]
    Su = Hu @ P @ Hu.T + noise ** 2 * np.eye(m - s)
    chi2 = r[s:] @ np.linalg.solve(Su, r[s:])
    Hi = np.linalg.inv(Hf)
    Pb = np.zeros((n + s, n + s)); Pb[:n, :n] = P
    Pb[n:, n:] = Hi @ (Hx @ P @ Hx.T + noise ** 2 * np.eye(s)) @ Hi.T
    Pb[:n, n:] = -P @ Hx.T @ Hi.T; Pb[n:, :n] = Pb[:n, n:].T
    He = np.zeros((m - s, n + s)); He[:, :n] = Hu
    S = He @ Pb @ He.T + noise ** 2 * np.eye(m - s)
    K = Pb @ He.T @ np.linalg.inv(S)
    Pa = Pb - K @ He @ Pb
    return 0.5 * (Pa + Pa.T), K @ r[s:], chi2


@pytest.mark.parametrize("C,nobs,s", [(11, 11, 3), (6, 4, 3), (11, 9, 1)])
def test_add_var_delayed_vs_numpy(C, nobs, s):
    rng = np.random.default_rng(10 + C + nobs)
    n = 21 + 6 * C
    P0 = spd(n, rng, 1e-2)
    vidx = [21 + 6 * i for i in range(C)]; vsize = [6] * C
    cols = np.arange(21, n)
    m = 2 * nobs
    H_old = rng.standard_normal((m, 6 * C)); H_new = rng.standard_normal((m, s)); res = 0.05 * rng.standard_normal(m)
    c = orc.Cov(P0)
    added, dx, chi2 = c.add_variable_delayed(vidx, vsize, H_old, H_new, res, 0.1, 1.0, False)
    Pn, dxn, chi2n = _numpy_add_delayed(P0, cols, H_old, H_new, res, 0.1)
    assert added and c.n == n + s
    assert np.linalg.norm(c.P - Pn) / np.linalg.norm(Pn) < 1e-11
    assert np.linalg.norm(dx - dxn) < 1e-9 * max(1.0, np.linalg.norm(dxn)) and abs(chi2 - chi2n) < 1e-8 * max(1.0, chi2n)
    # the chi2 gate (:614-618) and the shape guard (:571-575)
    c2 = orc.Cov(P0)
    added, _, _ = c2.add_variable_delayed(vidx, vsize, H_old, H_new, 50.0 * np.ones(m), 0.1, 0.95, True)
    assert not added and c2.n == n and np.array_equal(c2.P, P0)
    added, _, _ = c2.add_variable_delayed(vidx, vsize, H_old[:s], H_new[:s], res[:s], 0.1, 1.0, False)
    assert not added and c2.n == n


def test_replace_var_linear_is_a_congruence():
    rng = np.random.default_rng(5)
    C = 5; n = 21 + 6 * C + 3
    P0 = spd(n, rng)
    t = n - 3                                                          # the landmark, appended last
    pf = rng.standard_normal(3)
    H = np.zeros((3, 15)); H[:, 0:3] = -skew(pf); H[:, 6:9] = skew(pf); H[:, 12:15] = np.eye(3)      # MapServerManager.cpp:368-373
    vidx, vsize = [21, 21 + 6 * 4, t], [6, 6, 3]
    c = orc.Cov(P0)
    c.replace_var_linear(t, 3, vidx, vsize, H)
    J = np.eye(n); J[t:t + 3] = 0.0
    J[t:t + 3, 21:27] = H[:, 0:6]; J[t:t + 3, 45:51] = H[:, 6:12]; J[t:t + 3, t:t + 3] = H[:, 12:15]
    assert np.linalg.norm(c.P - J @ P0 @ J.T) < 1e-12 * np.linalg.norm(P0)


# ---- landmark rows vs finite differences under the filter's retraction ----------------------------------------------
def _meas_epose(Ri, pi, Rc, pc, pf, stereo, Rlr, tlr):
    q = Rc.T @ (Ri.T @ (pf - pi) - pc)
    out = [q[0] / q[2], q[1] / q[2]]
    if stereo:
        qr = Rlr @ q + tlr
        out += [qr[0] / qr[2], qr[1] / qr[2]]
    return np.array(out)


@pytest.mark.parametrize("stereo", [False, True])
def test_landmark_rows_epose_finite_differences(stereo):
    rng = np.random.default_rng(7)
    Ri, Rc, Rlr = rot(rng), rot(rng, 0.1), rot(rng, 0.02)
    pi, pc, tlr = rng.standard_normal(3), 0.1 * rng.standard_normal(3), np.array([-0.11, 0.001, 0.002])
    pf = pi + Ri @ (Rc @ np.array([0.4, -0.3, 6.0]) + pc)
    uv = _meas_epose(Ri, pi, Rc, pc, pf, True, Rlr, tlr) + 0.01 * rng.standard_normal(4)
    H, res = orc.landmark_rows_epose(Ri, pi, Rc, pc, pf, uv, stereo, Rlr, tlr)
    rows = 4 if stereo else 2
    h0 = _meas_epose(Ri, pi, Rc, pc, pf, stereo, Rlr, tlr)
    assert np.allclose(res, uv[:rows] - h0, atol=1e-14)
    eps = 1e-6
    Hfd = np.zeros((rows, 24))
    for c in range(24):
        d = np.zeros(24); d[c] = eps
        G = lambda th: orc.gamma(th, 0).reshape(3, 3)
        Ri2 = G(d[0:3]) @ Ri; pi2 = G(d[0:3]) @ pi + d[3:6]                    # SE23 retraction (v block does not enter)
        Rc2 = G(d[9:12]) @ Rc; pc2 = G(d[9:12]) @ pc + d[12:15]                # extrinsics (SE3)
        pf2 = G(d[15:18]) @ pf + d[21:24]                                      # AnchoredLandmark::update with the anchor's dtheta
        Hfd[:, c] = (_meas_epose(Ri2, pi2, Rc2, pc2, pf2, stereo, Rlr, tlr) - h0) / eps
    ok = np.ones(24, dtype=bool)
    assert np.allclose(H[:2][:, ok], Hfd[:2][:, ok], atol=2e-5)                # left rows: every block
    if stereo:
        ok[15:18] = False                                                       # right rows: all but the anchor block (Q12)
        assert np.allclose(H[2:][:, ok], Hfd[2:][:, ok], atol=2e-5)
        q = Rc.T @ (Ri.T @ (pf - pi) - pc); qr = Rlr @ q + tlr
        Hp = np.array([[1 / qr[2], 0, -qr[0] / qr[2] ** 2], [0, 1 / qr[2], -qr[1] / qr[2] ** 2]])
        assert np.allclose(H[2:, 15:18], -Hp @ Rlr @ skew(pf), atol=1e-12)    # as written (LandmarkUpdate.cpp:682)
        assert not np.allclose(H[2:, 15:18], Hfd[2:, 15:18], atol=1e-3)
    assert np.all(H[:, 6:9] == 0) and np.all(H[:, 18:21] == 0)                 # velocity, anchor position: zero blocks


def test_landmark_rows_sw_finite_differences():
    rng = np.random.default_rng(8)
    Rm, pm = rot(rng), rng.standard_normal(3)
    pf = pm + Rm @ np.array([0.2, 0.1, 4.0])
    meas = lambda R, p, f: (lambda q: np.array([q[0] / q[2], q[1] / q[2]]))(R.T @ (f - p))
    uv = meas(Rm, pm, pf) + 0.01
    H, res = orc.landmark_rows_sw(Rm, pm, pf, uv, False, False)
    G = lambda th: orc.gamma(th, 0).reshape(3, 3)
    eps = 1e-6; Hfd = np.zeros((2, 15))
    for c in range(15):
        d = np.zeros(15); d[c] = eps
        Hfd[:, c] = (meas(G(d[0:3]) @ Rm, G(d[0:3]) @ pm + d[3:6], G(d[6:9]) @ pf + d[12:15]) - meas(Rm, pm, pf)) / eps
    assert np.allclose(H, Hfd, atol=2e-5) and np.allclose(res, 0.01, atol=1e-14)
    Ha, _ = orc.landmark_rows_sw(Rm, pm, pf, uv, False, True)                   # current clone IS the anchor (:607-611)
    assert np.all(Ha[:, 0:3] == 0) and np.all(Ha[:, 6:9] == 0) and np.array_equal(Ha[:, 3:6], H[:, 3:6])


# ------------------------------------------------------------------------------------------------------------------
# GPU: the same operations through the C ABI vs the oracle
# ------------------------------------------------------------------------------------------------------------------
def _ctx(n, C, m_max=96):
    from ingvio_amd import capi
    return capi.Context(batch=2, n_max=((n + 15) // 16) * 16 + 16, c_max=C, f_max=32, m_max=m_max)


@pytest.mark.gpu
@pytest.mark.parametrize("C,nobs,s,stereo", [(11, 11, 3, True), (11, 11, 3, False), (6, 5, 3, True), (11, 9, 1, False), (21, 21, 3, True)])
def test_gpu_add_variable_delayed(C, nobs, s, stereo):
    rng = np.random.default_rng(100 + C + nobs + s)
    n = 21 + 6 * C
    P0 = spd(n, rng, 1e-2)
    vidx = [21 + 6 * i for i in range(C)]; vsize = [6] * C
    m = (4 if stereo else 2) * nobs
    H_old = rng.standard_normal((m, 6 * C)); H_new = rng.standard_normal((m, s)); res = 0.05 * rng.standard_normal(m)
    ctx = _ctx(n, C, m_max=max(96, m + 8))
    for b in (0, 1):
        ctx.cov_set(b, P0)
    c = orc.Cov(P0)
    addo, dxo, chi2o = c.add_variable_delayed(vidx, vsize, H_old, H_new, res, 0.1, 1.0, False)
    addg, dxg, chi2g, idx = ctx.add_variable_delayed(1, vidx, vsize, H_old, H_new, res, 0.1, 1.0, False)
    assert addo and addg and idx == n and ctx.n(1) == n + s and ctx.n(0) == n
    Pg = ctx.cov_get(1)
    assert np.linalg.norm(Pg - c.P) / np.linalg.norm(c.P) < 1e-11
    assert np.linalg.norm(dxg - dxo) < 1e-9 * max(1.0, np.linalg.norm(dxo)) and abs(chi2g - chi2o) < 1e-9 * max(1.0, chi2o)
    assert np.array_equal(Pg, Pg.T) and np.array_equal(ctx.cov_get(0), P0)
    # chi2 rejection leaves the state untouched (:614-618); m <= s is refused (:571-575)
    addg, dxg, chi2g, _ = ctx.add_variable_delayed(0, vidx, vsize, H_old, H_new, 50.0 * np.ones(m), 0.1, 0.95, True)
    assert not addg and ctx.n(0) == n and np.array_equal(ctx.cov_get(0), P0)
    addg, _, _, _ = ctx.add_variable_delayed(0, vidx, vsize, H_old[:s], H_new[:s], res[:s], 0.1, 1.0, False)
    assert not addg and ctx.n(0) == n
    ctx.close()


@pytest.mark.gpu
def test_gpu_add_variable_delayed_invertible_and_replace():
    rng = np.random.default_rng(300)
    C = 8; n = 21 + 6 * C
    P0 = spd(n, rng, 1e-2)
    ctx = _ctx(n, C)
    ctx.cov_set(0, P0)
    c = orc.Cov(P0)
    # addVarInv shape of the reference test (scalar on the extended pose), then a 3-vector on two clones
    H1 = rng.uniform(-1, 1, (1, 9))
    assert ctx.add_variable_delayed_invertible(0, [0], [9], H1, np.array([[1.0]]), 2.0) == c.add_variable_delayed_invertible([0], [9], H1, np.array([[1.0]]), 2.0)
    Hx = rng.standard_normal((3, 12)); Hf = rng.standard_normal((3, 3)) + 2 * np.eye(3)
    ig = ctx.add_variable_delayed_invertible(0, [21, 45], [6, 6], Hx, Hf, 0.05)
    io = c.add_variable_delayed_invertible([21, 45], [6, 6], Hx, Hf, 0.05)
    assert ig == io == n + 1 and ctx.n(0) == n + 4
    assert np.linalg.norm(ctx.cov_get(0) - c.P) / np.linalg.norm(c.P) < 1e-12
    # anchor change of that 3-vector (MapServerManager.cpp:363-375)
    pf = rng.standard_normal(3)
    H = np.zeros((3, 15)); H[:, 0:3] = -skew(pf); H[:, 6:9] = skew(pf); H[:, 12:15] = np.eye(3)
    ctx.replace_var_linear(0, ig, 3, [21, 63, ig], [6, 6, 3], H)
    c.replace_var_linear(io, 3, [21, 63, io], [6, 6, 3], H)
    Pg = ctx.cov_get(0)
    assert np.linalg.norm(Pg - c.P) / np.linalg.norm(c.P) < 1e-12 and np.array_equal(Pg, Pg.T)
    # errors: target outside the state, capacity
    from ingvio_amd import capi
    with pytest.raises(capi.IngvioError):
        ctx.replace_var_linear(0, ctx.n(0), 3, [21], [6], np.zeros((3, 6)))
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("stereo", [True, False])
def test_gpu_landmark_update_rows_through_generic_update(stereo):
    """updateLandmark{Mono,Stereo} (LandmarkUpdate.cpp:32-149, 688-8xx) = per-landmark rows, whitenResidual gate, stacked
    ekfUpdate: the covariance arithmetic is ingvio_chi2_gamma + ingvio_ekf_update on the oracle's rows."""
    rng = np.random.default_rng(400)
    C, L = 6, 5
    n = 21 + 6 * C + 3 * L
    P0 = spd(n, rng, 1e-3)
    ctx = _ctx(n, C, m_max=128)
    ctx.cov_set(0, P0)
    c = orc.Cov(P0)
    Ri, Rc, Rlr = rot(rng), rot(rng, 0.1), rot(rng, 0.02)
    pi, pc, tlr = rng.standard_normal(3), 0.1 * rng.standard_normal(3), np.array([-0.11, 0.001, 0.002])
    rows = 4 if stereo else 2
    Hs, rs, order = [], [], [(0, 9), (15, 6)]
    col = {0: 0, 15: 9}; ncol = 15
    blocks = []
    for l in range(L):
        pf = pi + Ri @ (Rc @ np.array([rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(3, 9)]) + pc)
        uv = _meas_epose(Ri, pi, Rc, pc, pf, True, Rlr, tlr) + 0.003 * rng.standard_normal(4)
        H, r = orc.landmark_rows_epose(Ri, pi, Rc, pc, pf, uv, stereo, Rlr, tlr)
        a_idx = 21 + 6 * int(rng.integers(0, C)); l_idx = 21 + 6 * C + 3 * l
        vo = [0, 15, a_idx, l_idx]; vs = [9, 6, 6, 3]
        go = c.whiten(vo, vs, H, r, 0.01 ** 2)
        gg = ctx.chi2_gamma(0, vo, vs, H, r, 0.01 ** 2)
        assert abs(go - gg) < 1e-9 * max(1.0, go)
        for v, sz in ((a_idx, 6), (l_idx, 3)):
            if v not in col:
                col[v] = ncol; ncol += sz; order.append((v, sz))
        blocks.append((H, r, a_idx, l_idx))
    Hl = np.zeros((rows * L, ncol)); rl = np.zeros(rows * L)
    for l, (H, r, a_idx, l_idx) in enumerate(blocks):
        sl = slice(rows * l, rows * l + rows)
        Hl[sl, 0:9] = H[:, 0:9]; Hl[sl, 9:15] = H[:, 9:15]
        Hl[sl, col[a_idx]:col[a_idx] + 6] = H[:, 15:21]; Hl[sl, col[l_idx]:col[l_idx] + 3] = H[:, 21:24]
        rl[sl] = r
    vo = [v for v, _ in order]; vs = [s for _, s in order]
    dxo, _ = c.ekf_update(vo, vs, Hl, rl, 0.01 ** 2)
    dxg, _ = ctx.ekf_update(0, vo, vs, Hl, rl, 0.01 ** 2)
    assert np.linalg.norm(ctx.cov_get(0) - c.P) / np.linalg.norm(c.P) < 1e-11
    assert np.linalg.norm(dxg - dxo) < 1e-9 * max(1.0, np.linalg.norm(dxo))
    ctx.close()


@pytest.mark.gpu
def test_gpu_chi2_gamma_multi():
    """all gates of a frame in one launch == the gates one by one == the oracle's whitenResidual"""
    rng = np.random.default_rng(500)
    C, L = 6, 9
    n = 21 + 6 * C + 3 * L
    P0 = spd(n, rng, 1e-3)
    ctx = _ctx(n, C, m_max=128)
    ctx.cov_set(1, P0)
    c = orc.Cov(P0)
    blocks = []
    for l in range(L):
        rows = 4 if l % 2 == 0 else 2                         # mixed block shapes in one call
        vo = [0, 15, 21 + 6 * int(rng.integers(0, C)), 21 + 6 * C + 3 * l]; vs = [9, 6, 6, 3]
        if l == 4:
            vo, vs = [21, 27], [6, 6]                         # a different var_order altogether
        H = rng.standard_normal((rows, sum(vs))); r = 0.02 * rng.standard_normal(rows)
        blocks.append((vo, vs, H, r))
    g = ctx.chi2_gamma_multi(1, blocks, 0.01 ** 2)
    for l, (vo, vs, H, r) in enumerate(blocks):
        go = c.whiten(vo, vs, H, r, 0.01 ** 2)
        assert abs(g[l] - go) < 1e-9 * max(1.0, go) and abs(ctx.chi2_gamma(1, vo, vs, H, r, 0.01 ** 2) - g[l]) < 1e-9 * max(1.0, go)
    assert len(ctx.chi2_gamma_multi(1, [], 1.0)) == 0
    ctx.close()


@pytest.mark.gpu
def test_gpu_error_paths_of_the_widened_abi():
    """the reference's fatal conditions come back as negative status codes (the shim turns them into its messages + exit)"""
    from ingvio_amd import capi, host, synth
    rng = np.random.default_rng(600)
    C = 4; n = 21 + 6 * C
    ctx = capi.Context(batch=2, n_max=48, c_max=C, f_max=16, m_max=64)           # n_max leaves room for ONE more 3-vector
    ctx.cov_set(0, spd(n, rng, 1e-2))
    vidx, vsize = [21 + 6 * i for i in range(C)], [6] * C
    H_old = rng.standard_normal((8, 6 * C)); H_new = rng.standard_normal((8, 3)); res = 0.01 * rng.standard_normal(8)
    added, _, _, idx = ctx.add_variable_delayed(0, vidx, vsize, H_old, H_new, res, 0.1, 1.0, False)
    assert added and idx == n and ctx.n(0) == n + 3
    with pytest.raises(capi.IngvioError) as e:                                      # no room for a second one: capacity, state untouched
        ctx.add_variable_delayed(0, vidx, vsize, H_old, H_new, res, 0.1, 1.0, False)
    assert e.value.code == capi.E_CAPACITY and ctx.n(0) == n + 3
    with pytest.raises(capi.IngvioError) as e:                                      # checkSubOrder: variable beyond the state
        ctx.chi2_gamma_multi(0, [([0, n + 3], [9, 3], rng.standard_normal((2, 12)), np.zeros(2))], 1.0)
    assert e.value.code == capi.E_NOT_IN_STATE
    with pytest.raises(capi.IngvioError) as e:
        ctx.replace_var_linear(0, n, 3, [21, n + 3], [6, 3], np.zeros((3, 9)))
    assert e.value.code == capi.E_NOT_IN_STATE
    with pytest.raises(capi.IngvioError) as e:                                      # more than 4096 columns: the export's limit
        ctx.qr_compress(np.zeros((8, 4100)), np.zeros(8))
    assert e.value.code == capi.E_CAPACITY
    Ht, rt = ctx.qr_compress(np.ones((6200, 7)), np.ones(6200))                     # any row count (row chunks above 6144)
    assert np.isfinite(Ht).all()
    # asynchronous staging replaces a whole input set: partial batches are refused
    flt, step, frame, info = synth.build_case(lambda P: capi.DeviceCov(ctx, 1, P), host.imu_transition, seed=7, F=16, C=C, n_gnss=0,
                                              n_landmarks=0, stereo=True)
    st = ctx.frame_stage_prepare(1, [step], [frame], step["sigma"], 0, 0.0, 0.0, use_async=True)
    with pytest.raises(capi.IngvioError) as e:
        st()
    assert e.value.code == capi.E_ARG
    ctx.close()
