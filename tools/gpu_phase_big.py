"""Shader-clock phases of k_feat_gram_big for ONE filter of BASELINE config 5 (needs an INGVIO_DBG_STAMPS=1 build)."""
import sys
sys.path.insert(0, "/root/repo")
import bench
from ingvio_amd import capi, synth
B, F, C = int(sys.argv[1]) if len(sys.argv) > 1 else 1, 300, 30
FF = int(sys.argv[2]) if len(sys.argv) > 2 else F            # features per frame (the context keeps f_max = 300)
N = 21 + 6 + 600 + 180
ctx = capi.Context(batch=B, n_max=((N + 15) // 16) * 16, c_max=C, f_max=F, m_max=64)
filters, steps, frames, infos = bench.build_batch(ctx, B, 0, FF, C, 6, 200)
ctx.snapshot(); pr = synth.PARAMS
ctx.frame_stage(0, steps, frames, filters[0].sigma(), 1, pr["sigma_cb"], pr["sigma_rw"])
for _ in range(3):
    ctx.frame_run(restore_prior=True)
d = ctx.debug_read(56)
print("gram_big [prologue, batch loop, rank-3 epilogue, sparse epilogue]", [d[i + 1] - d[i] for i in range(48, 52)], "total", d[52] - d[48])
d2 = ctx.debug_read(64)
print("gram_big batch 2 [operand phase (P2) + barrier, MFMA, sparse read-modify-write + barrier -> end of the loop is stamp 50]", [d2[54] - d2[53], d2[55] - d2[54]])
import os
if os.environ.get("INGVIO_DBG_TU") == "b":
    print("gate4_big [front, barrier, pair blocks, tile fill, LDL + gate]", [d[i + 1] - d[i] for i in range(5, 10)], "total", d[10] - d[5])
