#!/bin/bash
# kernel timeline of the split frame step (rocprofv3 --kernel-trace): tools/gpu_split_trace.sh <parts> <outdir>
P=${1:-2}; OUT=${2:-gpurun_out/r06a/trace_p$P}
export TMPDIR=/tmp
mkdir -p $OUT
INGVIO_FRAME_PARTS=$P rocprofv3 --kernel-trace --output-format csv -d $OUT -- python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import bench
from ingvio_amd import capi, synth
pr = synth.PARAMS
B = 512
ctx = capi.Context(batch=B, n_max=256, c_max=11, f_max=150, m_max=64)
filters, steps, frames, infos = bench.build_batch(ctx, B, 0, 150, 11, 6, 52)
ctx.snapshot()
ctx.frame_stage(0, steps, frames, filters[0].sigma(), filters[0].enable_gnss, pr["sigma_cb"], pr["sigma_rw"], max_accept=0, compress_rule=1)
ctx.sync()
for _ in range(12):
    ctx.frame_run(restore_prior=True)
ctx.sync()
ctx.close()
PY
find $OUT -name "*kernel_trace.csv" | head
