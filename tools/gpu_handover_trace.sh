#!/bin/bash
# rocprofv3 timeline (kernels + copies) of the pipelined track-store hand-over loop: bash tools/gpu_handover_trace.sh
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/handover_trace
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -o ht -- python - <<'PY' > $OUT/stdout.txt 2>&1
import os, sys, time
sys.path.insert(0, os.getcwd())
import bench
from ingvio_amd import capi, synth
B = 512
pr = synth.PARAMS
ctx = capi.Context(batch=B, n_max=256, c_max=11, f_max=150, m_max=64)
filters, steps, frames, infos = bench.build_batch(ctx, B, 0, 150, 11, 6, 52)
ctx.snapshot()
sg = (filters[0].sigma(), filters[0].enable_gnss, pr["sigma_cb"], pr["sigma_rw"])
kw = dict(max_accept=0, compress_rule=1)
tk, nbytes = bench.tracks_handover_prepare(ctx, steps, frames, sg, True, kw)
tk(); ctx.frame_run(restore_prior=True)
for _ in range(12):
    ctx.frame_fetch_begin(); tk(); ctx.frame_run(restore_prior=True); ctx.frame_fetch_end()
ctx.frame_fetch()
ctx.close()
PY
tail -2 $OUT/stdout.txt
python - <<'PY'
import csv, glob
rows = []
for f in glob.glob("gpurun_out/handover_trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:30], "q" + r["Queue_Id"]))
for f in glob.glob("gpurun_out/handover_trace/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")[:24], ""))
rows.sort()
t0 = rows[-75][0]
for s, e, n, q in rows[-75:]:
    print("%9.1f %9.1f %7.1f %s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, n))
PY
