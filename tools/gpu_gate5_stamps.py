import os, sys
sys.path.insert(0, "/root/repo")
import bench
from ingvio_amd import capi, synth
B = 512
ctx = capi.Context(batch=B, n_max=256, c_max=11, f_max=150, m_max=64)
filters, steps, frames, infos = bench.build_batch(ctx, B, 0, 150, 11, 6, 52)
ctx.snapshot(); pr = synth.PARAMS
ctx.frame_stage(0, steps, frames, filters[0].sigma(), 1, pr["sigma_cb"], pr["sigma_rw"])
for _ in range(3):
    ctx.frame_run(restore_prior=True)
d = ctx.debug_read(64)
print("gate5 [front, pair blocks + fill, eliminations, output]", [d[7] - d[5], d[8] - d[7], d[9] - d[8], d[10] - d[9]], "total", d[10] - d[5])
