"""BASELINE config 5's pure-kernel run: dense 6000 x 800 FP64 QR compression (cond ~ 1e3) through ingvio_qr_compress.
python tools/gpu_qr_bench.py  (under rocprofv3 --kernel-trace --stats for the device time)"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
sys.path.insert(0, "/root/repo")
from ingvio_amd import capi

m, n = 6000, 800
rng = np.random.default_rng(5)
U, _ = np.linalg.qr(rng.standard_normal((m, n))); V, _ = np.linalg.qr(rng.standard_normal((n, n)))
A = np.asfortranarray((U * np.logspace(0, -3, n)) @ V.T); b = rng.standard_normal(m)
ctx = capi.Context(batch=1, n_max=64, c_max=11, f_max=8, m_max=64)
Ht, rt = ctx.qr_compress(A, b)
t0 = time.perf_counter()
for _ in range(3):
    Ht, rt = ctx.qr_compress(A, b)
dt = (time.perf_counter() - t0) / 3
t0 = time.perf_counter(); Rl = np.linalg.qr(A, mode="r"); tl = time.perf_counter() - t0
flops = 2 * m * n * n - 2 / 3 * n ** 3
err = np.linalg.norm(Ht.T @ Ht - A.T @ A) / np.linalg.norm(A.T @ A)
print(f"6000x800: call incl. H2D/D2H {dt*1e3:.2f} ms; numpy/LAPACK dgeqrf on this host {tl*1e3:.1f} ms; {flops/1e9:.2f} GFLOP; "
      f"|R^T R - A^T A|/|A^T A| = {err:.1e}")
ctx.close()
