#!/usr/bin/env python3
"""Host-side cost of every call of the pipelined hand-over loop (512 filters, 150 features x 11 clones), frames vs track-store delta:
    python tools/gpu_handover_breakdown.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import bench
from ingvio_amd import capi, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
pr = synth.PARAMS
ctx = capi.Context(batch=B, n_max=256, c_max=11, f_max=150, m_max=64)
filters, steps, frames, infos = bench.build_batch(ctx, B, 0, 150, 11, 6, 52)
ctx.snapshot()
sg = (filters[0].sigma(), filters[0].enable_gnss, pr["sigma_cb"], pr["sigma_rw"])
kw = dict(max_accept=0, compress_rule=1)
st_async = ctx.frame_stage_prepare(0, steps, frames, *sg, use_async=True, **kw)
tk_async, nbytes = bench.tracks_handover_prepare(ctx, steps, frames, sg, True, kw)

def t(fn, n=10, sync=True):
    ts = []
    for _ in range(n):
        if sync: ctx.sync()
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts))
for name, stage in (("frames", st_async), ("tracks", tk_async)):
    stage(); ctx.frame_run(restore_prior=True); ctx.sync()
    print(name, "stage call (host, device idle): %.3f ms" % t(stage))
    print(name, "frame_run enqueue: %.3f ms" % t(lambda: ctx.frame_run(restore_prior=True)))
    def rs():
        ctx.frame_run(restore_prior=True); ctx.sync()
    print(name, "frame_run + sync: %.3f ms" % t(rs))
    print(name, "fetch_begin: %.3f ms" % t(ctx.frame_fetch_begin))
    ctx.frame_fetch_begin(); ctx.sync()
    def fe():
        ctx.frame_fetch_begin(); ctx.sync(); t0 = time.perf_counter(); ctx.frame_fetch_end(); return time.perf_counter() - t0
    print(name, "fetch_end after the copies landed: %.3f ms" % (1e3 * float(np.median([fe() for _ in range(10)]))))
    def loop(n):
        stage(); ctx.frame_run(restore_prior=True)
        t0 = time.perf_counter()
        for _ in range(n):
            ctx.frame_fetch_begin(); stage(); ctx.frame_run(restore_prior=True); ctx.frame_fetch_end()
        ctx.frame_fetch()
        return (time.perf_counter() - t0) / (n + 1) * 1e3
    loop(3)
    print(name, "pipelined loop: %.3f ms per frame of %d filters" % (loop(20), B))
ctx.close()
