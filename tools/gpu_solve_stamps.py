import os, sys
sys.path.insert(0, os.getcwd())
import bench
from ingvio_amd import capi, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ctx = capi.Context(batch=B, n_max=256, c_max=11, f_max=150, m_max=64)
filters, steps, frames, infos = bench.build_batch(ctx, B, 0, 150, 11, 6, 52)
ctx.snapshot(); pr = synth.PARAMS
ctx.frame_stage(0, steps, frames, filters[0].sigma(), 1, pr["sigma_cb"], pr["sigma_rw"])
for _ in range(3):
    ctx.frame_run(restore_prior=True)
d = ctx.debug_read(64)
print("solve stamps 0..7 diffs", [d[i + 1] - d[i] for i in range(0, 7)])
for k in range(8):
    b = 8 + 5 * k
    print("panel", k, "[write pan + barrier, block inverse, xinv + emit, MFMA + to next panel]", d[b + 1] - d[b], d[b + 2] - d[b + 1], d[b + 3] - d[b + 2], (d[b + 5] - d[b + 3]) if k < 7 else None)
print("set-up: [kernel start -> tables + trace loads done, -> reference clone chosen, S staged, indices ready (stamp 0)]", d[61] - d[60], d[0] - d[61])
