#!/usr/bin/env python3
"""Workload for rocprofv3: builds a batch of config-2 filters and runs `steps` frame updates."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from ingvio_amd import capi, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps_n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = capi.Context(batch=B, n_max=256, c_max=11, f_max=150, m_max=64)
filters, steps, frames, infos = bench.build_batch(ctx, B, 0, 150, 11, 6, 52)
ctx.snapshot()
pr = synth.PARAMS
ctx.frame_stage(0, steps, frames, filters[0].sigma(), 1, pr["sigma_cb"], pr["sigma_rw"])
for _ in range(steps_n):
    ctx.frame_run(restore_prior=True)
ctx.sync()
print("done")
