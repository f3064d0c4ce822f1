"""Experiment: the batch cut into S slices, one context + HIP stream per slice, stream priorities staggered so that the
slices fall out of lock-step and the latency-bound kernels of one slice overlap the gate kernel of another.
python tools/gpu_slice_pipeline.py [slices] [prio-mode]"""
import sys, time, ctypes
import numpy as np
sys.path.insert(0, ".")
import torch
import bench
from ingvio_amd import capi, synth

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4
mode = sys.argv[2] if len(sys.argv) > 2 else "stagger"
Btot, F, C, n_gnss, n_lm = 512, 150, 11, 6, 52
N = 21 + n_gnss + 3 * n_lm + 6 * C
B = Btot // S
torch.cuda.init()
lo, hi = -1, 0
try:
    lo, hi = torch.cuda.Stream.priority_range()      # (least, greatest): smaller number = higher priority
except Exception:
    pass
ctxs, streams = [], []
pr = synth.PARAMS
for s in range(S):
    if mode == "stagger":
        prio = -1 if s % 2 == 0 else 0
    elif mode == "same":
        prio = 0
    else:
        prio = -(s % 3)
    st = torch.cuda.Stream(priority=prio)
    streams.append(st)
    ctx = capi.Context(batch=B, n_max=((N + 15) // 16) * 16, c_max=C, f_max=F, m_max=64, stream=ctypes.c_void_p(st.cuda_stream))
    filters, steps, frames, infos = bench.build_batch(ctx, B, s * B, F, C, n_gnss, n_lm)
    ctx.snapshot()
    ctx.frame_stage(0, steps, frames, filters[0].sigma(), filters[0].enable_gnss, pr["sigma_cb"], pr["sigma_rw"], max_accept=0, compress_rule=1)
    ctxs.append(ctx)
for c in ctxs:
    c.sync()
def run(steps):
    for _ in range(steps):
        for c in ctxs:
            c.frame_run(restore_prior=True)
    for c in ctxs:
        c.sync()
run(5)
t0 = time.perf_counter(); K = 40
run(K)
dt = time.perf_counter() - t0
print(f"slices={S} mode={mode} prio_range=({lo},{hi}): {dt/K*1e3:.3f} ms per step of {Btot} filters = {Btot*K/dt/1e3:.1f} K updates/s")
