"""Throughput of the batched triangulation on the device next to the oracle on one host thread.
Run on the GPU box: python tools/gpu_tri_bench.py [batch]   (rocprofv3 --kernel-trace --stats for the kernel time)"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from ingvio_amd import capi, host, synth
from oracle import oracle as orc

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
F, C = 150, 11
for stereo in (True, False):
    ctx = capi.Context(batch=B, n_max=96, c_max=C, f_max=F, m_max=32)
    base = []
    for b in range(8):
        flt, step, frame, info = synth.build_case(lambda P, b=b: capi.DeviceCov(ctx, b, P), host.imu_transition, seed=70 + b,
                                                  F=F, C=C, n_gnss=0, n_landmarks=0, stereo=stereo)
        frame = dict(frame)
        frame["uv"] = frame["uv"] + np.random.default_rng(b).normal(0, 2e-3, frame["uv"].shape)
        base.append((step, frame))
    steps = [base[b % 8][0] for b in range(B)]
    frames = [base[b % 8][1] for b in range(B)]
    ctx.frame_stage(0, steps, frames, steps[0]["sigma"])
    kw = dict(stereo=stereo, R_cl2cr=frames[0]["R_cl2cr"], t_cl2cr=frames[0]["t_cl2cr"])
    pf, ok = ctx.triangulate(0, None, **kw)
    t0 = time.perf_counter(); reps = 20
    for _ in range(reps):
        pf, ok = ctx.triangulate(0, None, **kw)
    dt = (time.perf_counter() - t0) / reps
    fr = frames[0]; t0 = time.perf_counter(); n = 0
    for j in range(F):
        oko, pfo = orc.triangulate(fr["clone_R"], fr["clone_p"], int(fr["obs_mask"][j]), fr["uv"][j], stereo, fr["R_cl2cr"], fr["t_cl2cr"])
        assert bool(ok[0, j]) == oko and np.linalg.norm(pf[0, j] - pfo) <= 5e-7 * max(1.0, np.linalg.norm(pfo))
        n += 1
    dc = (time.perf_counter() - t0) / n
    print(f"{'stereo' if stereo else 'mono'}: {B}x{F} features, {int(ok.sum())} ok, device call incl. result copy {dt*1e3:.3f} ms "
          f"= {B*F/dt/1e6:.1f} M features/s; oracle (ctypes, 1 thread) {dc*1e6:.1f} us/feature = {1/dc/1e3:.1f} K features/s")
    ctx.close()
