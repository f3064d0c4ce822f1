"""Shader-clock phases of k_chol_diag (last panel of a 6000 x 800 Cholesky-QR); needs INGVIO_DBG_STAMPS=1 build."""
import sys
import numpy as np
sys.path.insert(0, "/root/repo")
from ingvio_amd import capi
m, n = 6000, 800
rng = np.random.default_rng(5)
A = np.asfortranarray(rng.standard_normal((m, n))); b = rng.standard_normal(m)
ctx = capi.Context(batch=1, n_max=64, c_max=11, f_max=8, m_max=64)
for _ in range(2):
    ctx.qr_compress(A, b)
d = ctx.debug_read(64)[56:]
print("k_chol_diag [gemm, barrier+reduce, factor, scale+store] cycles(100MHz ticks?)", [d[i + 1] - d[i] for i in range(4)], "total", d[4] - d[0])
