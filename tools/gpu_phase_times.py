#!/usr/bin/env python3
"""Shader-clock phase breakdown of the instrumented kernels (block (0,0) lane 0 only: the oldest wave of its SIMD, i.e.
un-contended latency, not throughput); debugging aid.  Needs a build with INGVIO_DBG_STAMPS=1."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
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
def seg(name, idx):
    print(name, [d[b] - d[a] for a, b in zip(idx[:-1], idx[1:])], "total", d[idx[-1]] - d[idx[0]])
seg("gate3 [front, barrier, pair blocks + tile fill, LDL, border + gate]", [5, 6, 7, 8, 9, 10])
seg("gram2 prologue", [32, 33]); seg("gram2 batch 3 as wave 0 sees it [P3a, barrier, P2 of the next batch, barrier]", [36, 37, 38, 41, 42]); seg("gram2 epilogue", [39, 40]); seg("gram2 total", [32, 40])
seg("propagate [fetch, compose,gnss,strip,AA,fused clone]", [16, 21, 17, 18, 19, 20, 22])
seg("info_solve [deal+load, sweep1, R2+G1, G2, sweep2, G3, Pc copy]", [24, 25, 26, 27, 28, 29, 30, 31])
seg("info_apply [setup, T = Pc M, tile loop]", [11, 12, 13]) if False else None
print("info_apply [T = Pc M, tile loop]", [d[12] - d[11], d[13] - d[12]])
print("info_apply step 4 [MFMAs + stores, stage store, barrier]", [d[45] - d[44], d[46] - d[45], d[47] - d[46]], " start -> T:", d[11] - d[12], " whole:", d[13] - d[12])
print("info_apply step 4 [issue loads + B fragments, MFMAs tile 0, store tile 0, tile 1]", [d[48] - d[44], d[49] - d[48], d[50] - d[49], d[45] - d[50]])
