"""ingvio_qr_compress on the three shapes VERDICT r02 asks for — the literal config-2 SPQR shape 6150 x 66, the literal config-5
stack 35100 x 180 (both rank n - 6, real stacked MSCKF Jacobians), BASELINE's nominal 6000 x 800 (cond 1e3) — with both methods.
Device time = the HIP-event time of the captured launch graph (profile slot of the QR stage), host time = the whole call incl.
PCIe.  Writes gpurun_out/qr_shapes.json."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from ingvio_amd import capi
from oracle import oracle as orc
import test_gpu_config4 as T4
orc.build()
rng = np.random.default_rng(5)
shapes = []
H, r, _ = T4._stacked_msckf_jacobian(orc, 11, 150, seed=11); shapes.append(("config-2 stack", H, r))
H, r, _ = T4._stacked_msckf_jacobian(orc, 30, 300, seed=30); shapes.append(("config-5 stack", H, r))
m, n = 6000, 800
U, _ = np.linalg.qr(rng.standard_normal((m, n))); V, _ = np.linalg.qr(rng.standard_normal((n, n)))
shapes.append(("nominal 6000x800", (U * np.logspace(0, -3, n)) @ V.T, rng.standard_normal(m)))
ctx = capi.Context(batch=1, n_max=64, c_max=11, f_max=8, m_max=64)
out = []
for name, A, b in shapes:
    A = np.asfortranarray(A); m, n = A.shape
    flops = 2.0 * m * n * n - 2.0 / 3.0 * n ** 3
    for method in ("householder", "cholesky", "auto"):
        ctx.set_qr_method(method)
        Ht, rt = ctx.qr_compress(A, b)                       # builds the graph
        ctx.profile_select(None); ctx.profile_reset(); ctx.profile_enable(True)
        t0 = time.perf_counter()
        for _ in range(3):
            Ht, rt = ctx.qr_compress(A, b)
        host_ms = (time.perf_counter() - t0) / 3 * 1e3
        ctx.profile_enable(False)
        prof = ctx.profile_get()
        dev_ms = sum(ms for k, (ms, calls) in prof.items()) / 3
        G = A.T @ A
        e = float(np.linalg.norm(Ht.T @ Ht - G) / np.linalg.norm(G))
        rec = dict(shape=name, m=m, n=n, method=method, device_ms=dev_ms, host_call_ms=host_ms, tflops=flops / (dev_ms * 1e-3) / 1e12,
                   frac_fp64_peak=flops / (dev_ms * 1e-3) / 1e12 / 78.6, gram_err=e, finite=bool(np.isfinite(Ht).all()))
        out.append(rec)
        print("%-18s %6d x %3d %-11s device %8.3f ms  call %8.2f ms  %6.2f TFLOP/s (%.3f of FP64 peak)  |R^T R - H^T H| %.1e" %
              (name, m, n, method, dev_ms, host_ms, rec["tflops"], rec["frac_fp64_peak"], e))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "qr_shapes.json"), "w"), indent=1)
ctx.close()
