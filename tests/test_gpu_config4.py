"""BASELINE config 4's per-GPU share as a test: 4096 independent frames over 8 GPUs = 512 filters in ONE
context, 150 features x 11 clones, N = 249, distinct seeds — the batch bench.py times.  The oracle runs on a strided sample
that contains the first and the last filter of every XCD's share (filter b is served by XCD b % 8 in the gate, gram and apply
kernels: kernels_factored.hip, gate_kernel.h); size-independent properties are checked on all 512.

Also here: Cholesky-QR on REAL stacked MSCKF Jacobians (rank n - 6 by construction) at the window sizes that
select it (n >= 128: 22, 30, 35 clones — the reference ships 21 / 25 / 27 / 35)."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu


def test_config4_per_gpu_batch(orc):
    import bench
    from ingvio_amd import capi, synth
    B, F, C, n_gnss, n_lm = 512, 150, 11, 6, 52
    N = 21 + n_gnss + 3 * n_lm + 6 * C
    assert N == 249
    ctx = capi.Context(batch=B, n_max=256, c_max=C, f_max=F, m_max=64)
    filters, steps, frames, infos = bench.build_batch(ctx, B, 0, F, C, n_gnss, n_lm)
    ctx.snapshot()
    pr = synth.PARAMS
    ctx.frame_stage(0, steps, frames, filters[0].sigma(), filters[0].enable_gnss, pr["sigma_cb"], pr["sigma_rw"],
                    max_accept=0, compress_rule=1)
    # the sample: first (b = x) and last (b = 504 + x) filter of every XCD's share, plus a stride through the middle
    sample = sorted(set(list(range(8)) + list(range(B - 8, B)) + list(range(8, B - 8, 29))))
    assert len(sample) >= 32 and all(x in sample for x in (0, 7, 504, 511))
    ctx.restore(); ctx.sync()
    priors = {b: ctx.cov_get(b) for b in sample}
    ctx.frame_run(restore_prior=True)
    dx1, acc1, rows1 = ctx.frame_fetch()
    P1 = [ctx.cov_get(b) for b in range(B)]
    # ---- oracle on the sample ------------------------------------------------------------------------------------------
    worst = dict(cov=0.0, dx=0.0)
    for b in sample:
        assert priors[b].shape == (N - 6, N - 6)
        oc = orc.Cov(priors[b], ld=256)
        dxo, acco, gamo, m = orc.frame_update(oc, steps[b], frames[b], max_accept=0, compress_rule=1)
        assert np.array_equal(acc1[b, :F], acco), b
        assert P1[b].shape == oc.P.shape == (N - 6, N - 6)          # propagate + clone (+6) + marginalise (-6)
        worst["cov"] = max(worst["cov"], rel_err(P1[b], oc.P))
        worst["dx"] = max(worst["dx"], rel_err(dx1[b, :N], dxo))
    print("config 4 per-GPU batch: %d filters, oracle on %d: cov %.1e dx %.1e" % (B, len(sample), worst["cov"], worst["dx"]))
    assert worst["cov"] < 1e-9 and worst["dx"] < 1e-7, worst       # BASELINE tolerance 1e-6; FP64 delivers far better
    # ---- properties on all 512 -----------------------------------------------------------------------------------------
    assert np.isfinite(dx1).all() and (rows1 == 6 * C).all()
    for b in range(B):
        P = P1[b]
        assert np.array_equal(P, P.T), b                             # exactly symmetric
        assert np.linalg.eigvalsh(P).min() > -1e-10, b               # PSD
        assert np.array_equal(acc1[b, :F] == 0, infos[b]["outlier"]), b      # the gate rejects exactly the planted outliers
    # bitwise repeatable: same prior, same inputs -> same bits, whatever the order the workgroups ran in
    ctx.frame_run(restore_prior=True)
    dx2, acc2, rows2 = ctx.frame_fetch()
    assert np.array_equal(dx1, dx2) and np.array_equal(acc1, acc2) and np.array_equal(rows1, rows2)
    for b in range(B):
        assert np.array_equal(P1[b], ctx.cov_get(b)), b
    ctx.close()


def _stacked_msckf_jacobian(orc, C, F, seed):
    """All accepted features' H_j = V^T Hx and r_j = V^T r (orc_msckf_feature_block) of a C-clone window, stacked
    (RemoveLostUpdate.cpp:336-371): (F (4 C - 3)) x 6 C, rank 6 C - 6."""
    from ingvio_amd import synth
    ld = ((21 + 6 * C + 15) // 16) * 16
    flt, step, frame, info = synth.build_case(lambda P: orc.Cov(P, ld=ld), orc.imu_transition, seed=seed, F=F, C=C, n_gnss=0,
                                              n_landmarks=0, outlier_every=0)
    blocks = [orc.feature_block(frame, j) for j in range(F)]
    H = np.vstack([b[0] for b in blocks]); r = np.concatenate([b[1] for b in blocks])
    return H, r, frame


@pytest.mark.parametrize("C", [22, 30, 35])
def test_cholesky_qr_on_rank_deficient_msckf_stack(orc, C):
    from ingvio_amd import capi
    F = 24
    H, r, frame = _stacked_msckf_jacobian(orc, C, F, seed=400 + C)
    m, n = H.shape
    assert n == 6 * C and n >= 128 and m >= 6 * n
    sv = np.linalg.svd(H, compute_uv=False)
    assert (sv[-6:] < 1e-9 * sv[0]).all() and sv[-7] > 1e-6 * sv[0]          # rank n - 6
    N = 21 + n
    ctx = capi.Context(batch=3, n_max=((N + 15) // 16) * 16, c_max=C, f_max=8, m_max=64)      # 6 C update rows: the dense-H route
    rng = np.random.default_rng(C)
    G = rng.standard_normal((N, N)); P0 = 1e-3 * (G @ G.T / N + 0.2 * np.eye(N))
    vidx = [21 + 6 * c for c in range(C)]; vsize = [6] * C
    var = 0.08 ** 2
    thin, post = {}, {}
    for k, method in enumerate(("householder", "cholesky", "auto")):
        ctx.set_qr_method(method)
        Ht, rt = ctx.qr_compress(H, r)
        assert np.isfinite(Ht).all() and np.isfinite(rt).all(), method
        assert not np.tril(Ht, -1).any(), method
        eA, eb = rel_err(Ht.T @ Ht, H.T @ H), rel_err(Ht.T @ rt, H.T @ r)
        print("C=%d %dx%d %-11s  R^T R vs H^T H %.1e   R^T z vs H^T r %.1e" % (C, m, n, method, eA, eb))
        assert eA < 1e-12 and eb < 1e-11, (method, eA, eb)
        thin[method] = (Ht, rt)
        ctx.cov_set(k, P0)
        dx, st = ctx.ekf_update(k, vidx, vsize, Ht, rt, var)
        assert st == 0
        post[method] = (ctx.cov_get(k), dx[:N])
    Ph, dxh = post["householder"]
    # yardstick for all three: StateManager::ekfUpdate (StateManager.cpp:399-411) on the UNCOMPRESSED stack, numpy / LAPACK
    # (the oracle's m x m partial-pivot inverse at m = F (4 C - 3) rows would take minutes)
    cols = np.concatenate([np.arange(i, i + 6) for i in vidx])
    PHt = P0[:, cols] @ H.T
    S = H @ PHt[cols] + var * np.eye(m)
    K = np.linalg.solve(S, PHt.T).T
    Pn = P0 - K @ PHt.T
    Pn = 0.5 * (Pn + Pn.T); dxn = K @ r
    assert rel_err(Ph, Pn) < 1e-9 and rel_err(dxh, dxn) < 1e-8, (rel_err(Ph, Pn), rel_err(dxh, dxn))
    for method in ("cholesky", "auto"):
        Pm, dxm = post[method]
        assert rel_err(Pm, Ph) < 1e-9 and rel_err(dxm, dxh) < 1e-8, (method, rel_err(Pm, Ph), rel_err(dxm, dxh))
    ctx.close()
