"""debug: the information solve [M | t] of the HIP path against numpy on the same A, b, Pcc."""
import sys
import numpy as np
sys.path.insert(0, ".")
from ingvio_amd import capi, host, synth

C, F = int(sys.argv[1]) if len(sys.argv) > 1 else 11, 40
ctx = capi.Context(batch=1, n_max=96, c_max=C, f_max=F, m_max=64)
flt, step, frame, info = synth.build_case(lambda P: capi.DeviceCov(ctx, 0, P), host.imu_transition, seed=3, F=F, C=C, n_gnss=0, n_landmarks=0)
for Phi, G, dt in zip(step["Phi"], step["G"], step["dt"]):
    flt.cov.propagate(Phi, G, dt, step["sigma"], 0, step["gnss_idx"], 0.2, 0.2)
flt.cov.augment(step["R_i2w"])
P0 = ctx.cov_get(0)
dx, acc, gam, rows = ctx.msckf_update(0, frame)
A, b = ctx.debug_msckf_info(0)
M, t = ctx.debug_info_solution(0)
n = A.shape[0]
cols = np.concatenate([np.arange(i, i + 6) for i in frame["clone_idx"]])
Pcc = P0[np.ix_(cols, cols)]
var = frame["noise"] ** 2
K1 = A @ Pcc + var * np.eye(n)
Mr = np.linalg.solve(K1, A); tr = np.linalg.solve(K1, b)
print("n", n, "|M - Mref| / |Mref|", np.linalg.norm(M[:n, :n] - Mr) / np.linalg.norm(Mr), "t", np.linalg.norm(t[:n] - tr) / np.linalg.norm(tr))
E = np.abs(M[:n, :n] - Mr) / np.abs(Mr).max()
print("worst entries (tile coords):")
blk = np.array([[E[16 * i:16 * i + 16, 16 * j:16 * j + 16].max() if E[16 * i:16 * i + 16, 16 * j:16 * j + 16].size else 0 for j in range((n + 15) // 16)] for i in range((n + 15) // 16)])
np.set_printoptions(precision=1, linewidth=200)
print(blk)
print("M sym err", np.abs(M - M.T).max() / np.abs(M).max(), " pad rows nonzero:", np.abs(M[n:]).max() if M.shape[0] > n else 0)
