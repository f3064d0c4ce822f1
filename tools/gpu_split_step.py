#!/usr/bin/env python3
"""Experiment / regression aid (round 6): the default workload with ingvio_frame_run's batch dealt to 1..4 slices on their own
streams (ingvio_set_frame_parts) - ms per step of 512 filters, and that every split gives the bit-identical posterior.
    python tools/gpu_split_step.py [steps] [batch]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import bench
from ingvio_amd import capi, synth

K = int(sys.argv[1]) if len(sys.argv) > 1 else 40
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
pr = synth.PARAMS
ctx = capi.Context(batch=B, n_max=256, c_max=11, f_max=150, m_max=64)
filters, steps, frames, infos = bench.build_batch(ctx, B, 0, 150, 11, 6, 52)
ctx.snapshot()
ctx.frame_stage(0, steps, frames, filters[0].sigma(), filters[0].enable_gnss, pr["sigma_cb"], pr["sigma_rw"], max_accept=0, compress_rule=1)
ctx.sync()
ref = None
for parts in (1, 2, 3, 4, 2, 1):
    ctx.set_frame_parts(parts)
    for _ in range(5):
        ctx.frame_run(restore_prior=True)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(K):
        ctx.frame_run(restore_prior=True)
    ctx.sync()
    ms = (time.perf_counter() - t0) / K * 1e3
    dx, acc, rows = ctx.frame_fetch()
    P = [ctx.cov_get(b) for b in (0, B // 2 - 1, B // 2, B - 1)]
    if ref is None:
        ref = (dx, acc, rows, P)
    same = bool(np.array_equal(dx, ref[0]) and np.array_equal(acc, ref[1]) and np.array_equal(rows, ref[2]) and all(np.array_equal(a, b) for a, b in zip(P, ref[3])))
    print("parts %d: %.4f ms per step of %d filters = %.0f updates/s   bit-identical to parts 1: %s" % (parts, ms, B, B / ms * 1e3, same), flush=True)
ctx.close()
