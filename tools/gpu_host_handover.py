"""Host hand-over cost of one batched frame: ingvio_frame_stage (pack + H2D) / ingvio_frame_run / ingvio_frame_fetch (D2H),
serialised, next to the device-resident step time.  python tools/gpu_host_handover.py [batch]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import bench
from ingvio_amd import capi, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
F, C, n_gnss, n_lm = 150, 11, 6, 52
N = 21 + n_gnss + 3 * n_lm + 6 * C
ctx = capi.Context(batch=B, n_max=((N + 15) // 16) * 16, c_max=C, f_max=F, m_max=64)
filters, steps, frames, infos = bench.build_batch(ctx, B, 0, F, C, n_gnss, n_lm)
ctx.snapshot()
pr = synth.PARAMS
stage = ctx.frame_stage_prepare(0, steps, frames, filters[0].sigma(), filters[0].enable_gnss, pr["sigma_cb"], pr["sigma_rw"], max_accept=0, compress_rule=1)
stage(); ctx.frame_run(restore_prior=True); ctx.frame_fetch()
def t(fn, reps=5):
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.sync(); return (time.perf_counter() - t0) / reps
ts = t(stage)
tr = t(lambda: ctx.frame_run(restore_prior=True), 20)
tf = t(lambda: ctx.frame_fetch())
def full():
    stage(); ctx.frame_run(restore_prior=True); ctx.frame_fetch()
ta = t(full)
frame_bytes = sum(np.asarray(v).nbytes for v in frames[0].values() if hasattr(v, "nbytes")) + sum(np.asarray(v).nbytes for v in steps[0].values() if hasattr(v, "nbytes"))
kw = dict(max_accept=0, compress_rule=1)
stage_a = ctx.frame_stage_prepare(0, steps, frames, filters[0].sigma(), filters[0].enable_gnss, pr["sigma_cb"], pr["sigma_rw"], use_async=True, **kw)
def pipelined(stage_fn, reps=20):
    stage_fn(); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.frame_run(restore_prior=True)      # frame i on the device ...
        stage_fn()                             # ... while frame i+1 is packed and sent
        ctx.frame_fetch()
    ctx.sync()
    return (time.perf_counter() - t0) / reps
tp = pipelined(stage); tpa = pipelined(stage_a); tpa = pipelined(stage_a)
dxa, acca, rowsa = ctx.frame_fetch()
stage(); ctx.frame_run(restore_prior=True); dxs, accs, rowss = ctx.frame_fetch()
assert np.array_equal(dxa, dxs) and np.array_equal(acca, accs), "async staging changed the result"
print(f"pipelined run(i); stage(i+1); fetch(i): in-stream {tp*1e3:.2f} ms = {B/tp/1e3:.1f} K updates/s, copy stream + second input set "
      f"{tpa*1e3:.2f} ms = {B/tpa/1e3:.1f} K updates/s")
print(f"B={B}: stage {ts*1e3:.2f} ms ({B*frame_bytes/1e6:.1f} MB in), run {tr*1e3:.2f} ms, fetch {tf*1e3:.2f} ms, stage+run+fetch {ta*1e3:.2f} ms "
      f"= {B/ta/1e3:.1f} K updates/s host-inclusive vs {B/tr/1e3:.1f} K device-resident")
