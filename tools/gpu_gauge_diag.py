"""diagnostic (GPU): where the factored update loses the window block at an inflated prior and a tiny noise.
Run twice: INGVIO_INFO_GAUGE=off (unreduced solve) and default (gauge-reduced solve, kernels_solve.hip).  Prints, per
(prior scale, sigma): the leak |A U| / |A| of the device's information matrix on the gauge directions U = 1_C (x) I_6, its
distance to the 80-bit A, and the window block / dx of the posterior against the 80-bit update computed (a) from the exact A and
(b) from the DEVICE's A - (b) separates the input error of A from the error of the solve itself."""
import os
import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from ingvio_amd import capi, host, synth
from oracle import oracle as orc
import test_gpu_pinning as T

orc.build()
LD = np.longdouble
rel = lambda a, b: float(np.linalg.norm(np.asarray(a - b, dtype=np.float64)) / max(np.linalg.norm(np.asarray(b, dtype=np.float64)), 1e-300))
sigmas = [0.08, 1e-2, 1e-3]
ctx = capi.Context(batch=len(sigmas), n_max=256, c_max=11, f_max=150, m_max=64)
ctx.set_method("factored")
flt, step, frame, info = synth.build_case(lambda P: capi.DeviceCov(ctx, 0, P), host.imu_transition, seed=21)
P0 = T._prior_at_update(ctx, 0, flt, step)
cols = np.concatenate([np.arange(i, i + 6) for i in frame["clone_idx"]])
n = len(cols); win = np.ix_(cols, cols)
U = np.kron(np.ones((n // 6, 1)), np.eye(6))
Aj, bj = [], []
for j in range(150):
    Hj, rj = orc.feature_block(frame, j)
    Hl, rl = Hj.astype(LD), rj.astype(LD)
    Aj.append(Hl.T @ Hl); bj.append(Hl.T @ rl)
print("INGVIO_INFO_GAUGE =", os.environ.get("INGVIO_INFO_GAUGE", "(default: on)"))
for scale in (1.0, 1e4):
    for b, s in enumerate(sigmas):
        ctx.cov_set(b, P0 * scale)
        f = dict(frame); f["noise"] = s
        dx, acc, gam, rows = ctx.msckf_update(b, f)
        Pg = ctx.cov_get(b); dxg = dx[0, :249]; accg = acc[0, :150]
        A, bv = ctx.debug_msckf_info(b)
        idx = np.flatnonzero(accg)
        At = sum(Aj[j] for j in idx); bt = sum(bj[j] for j in idx)
        Pt, dxt = T._truth_update_longdouble(P0 * scale, cols, Aj, bj, accg, s * s)
        # the 80-bit update fed with the device's A, b
        Pt2, dxt2 = T._truth_update_longdouble(P0 * scale, cols, [A.astype(LD)], [bv.astype(LD)], np.array([1]), s * s)
        oc = orc.Cov(P0 * scale, ld=256)
        dxo, acco, gamo, m = oc.msckf_update(f, max_accept=0, compress_rule=1)
        print("scale %.0e sigma %-6g acc %3d | leak |AU|/|A| %.1e  |A-A80|/|A| %.1e | window vs 80-bit(exact A): hip %.1e oracle %.1e | "
              "window vs 80-bit(device A): %.1e | dx: hip %.1e oracle %.1e (device A: %.1e) | masks equal %s" % (
                  scale, s, len(idx), np.linalg.norm(A @ U) / np.linalg.norm(A), rel(A, At), rel(Pg[win], Pt[win]), rel(oc.P[win], Pt[win]),
                  rel(Pg[win], Pt2[win]), rel(dxg, dxt), rel(dxo, dxt), rel(dxg, dxt2), np.array_equal(accg, acco)))
ctx.close()
