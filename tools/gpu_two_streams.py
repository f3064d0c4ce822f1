#!/usr/bin/env python3
"""Experiment (debugging aid): the default workload as ONE context of 512 filters against TWO contexts of 256 filters whose frame
steps are enqueued alternately on their own streams - does the device overlap the latency-bound kernels of one half (solve, the
tails of apply) with the throughput-bound ones of the other?   python tools/gpu_two_streams.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from ingvio_amd import capi, synth

K = int(sys.argv[1]) if len(sys.argv) > 1 else 40
pr = synth.PARAMS


def make(B, seed0):
    ctx = capi.Context(batch=B, n_max=256, c_max=11, f_max=150, m_max=64)
    filters, steps, frames, infos = bench.build_batch(ctx, B, seed0, 150, 11, 6, 52)
    ctx.snapshot()
    ctx.frame_stage(0, steps, frames, filters[0].sigma(), filters[0].enable_gnss, pr["sigma_cb"], pr["sigma_rw"], max_accept=0, compress_rule=1)
    ctx.sync()
    return ctx


def timed(ctxs, k):
    for _ in range(5):
        for c in ctxs: c.frame_run(restore_prior=True)
    for c in ctxs: c.sync()
    t0 = time.perf_counter()
    for _ in range(k):
        for c in ctxs: c.frame_run(restore_prior=True)
    for c in ctxs: c.sync()
    return (time.perf_counter() - t0) / k * 1e3


one = make(512, 0)
print("one context of 512 filters:      %.4f ms per step of 512" % timed([one], K))
del one
for parts in (2, 4):
    cs = [make(512 // parts, i * (512 // parts)) for i in range(parts)]
    print("%d contexts of %d on %d streams:  %.4f ms per step of 512" % (parts, 512 // parts, parts, timed(cs, K)))
    del cs
