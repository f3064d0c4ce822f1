"""Shader-clock phases of k_feat_gram3 (needs a kernels_factored build with -DINGVIO_DBG_STAMPS, e.g. tools/build_tu_variant.sh)."""
import sys
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
print("gram3 wave 0, batch 3: [P2 of the next batch, wait at the barrier]", [d[41] - d[36], d[42] - d[41]], " prologue", d[33] - d[32], " total", d[40] - d[32])
