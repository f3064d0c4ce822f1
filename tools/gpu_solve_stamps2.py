"""Shader-clock stamps of k_info_solve (build_var/stamps: -DINGVIO_DBG_STAMPS -DINGVIO_DBG_BLOCK=<b>): phases of one workgroup.
    INGVIO_HIP_LIB=build_var/stamps/libingvio_hip.so INGVIO_DBG_TU=s python tools/gpu_solve_stamps2.py [batch] [clones]"""
import os, sys
sys.path.insert(0, os.getcwd())
import bench
from ingvio_amd import capi, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
C = int(sys.argv[2]) if len(sys.argv) > 2 else 11
N = 21 + 6 + 3 * 52 + 6 * C
ctx = capi.Context(batch=B, n_max=(N + 15) // 16 * 16, c_max=C, f_max=150, m_max=64)
filters, steps, frames, infos = bench.build_batch(ctx, B, 0, 150, C, 6, 52)
ctx.snapshot(); pr = synth.PARAMS
ctx.frame_stage(0, steps, frames, filters[0].sigma(), 1, pr["sigma_cb"], pr["sigma_rw"])
for _ in range(3):
    ctx.frame_run(restore_prior=True)
d = ctx.debug_read(64)
names = {60: "kernel start", 61: "tables + trace loads", 0: "S staged, ref clone, indices", 1: "A frags, Pdd tiles, X/Y zeroed", 2: "factorisation 1", 3: "A L", 4: "W tiles",
         5: "factorisation 2 (solve_wave end)", 6: "M product + scatter + gauge block", 7: "Pc copy / outputs"}
order = [60, 61, 0, 1, 2, 3, 4, 5, 6, 7]
print("B = %d, C = %d: k_info_solve phases (cycles of the 100 MHz... shader clock counter units as read)" % (B, C))
for a, b in zip(order[:-1], order[1:]):
    print("  %-40s %8d" % (names[b], d[b] - d[a]))
print("  total %d" % (d[7] - d[60]))
for k in range(8):
    b = 8 + 5 * k
    print("  fact-1 panel", k, "[pan write, block inverse, xinv + emit, MFMA -> next]", d[b + 1] - d[b], d[b + 2] - d[b + 1], d[b + 3] - d[b + 2], (d[b + 5] - d[b + 3]) if k < 7 else None)
