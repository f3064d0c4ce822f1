#!/bin/bash
# kernel timeline of one config-5 bench run (rocprofv3 --kernel-trace) -> gpurun_out/c5trace_<tag>.csv ; usage: bash tools/gpu_c5_trace.sh <tag> [bench args]
ROOT=$(cd "$(dirname "$0")/.." && pwd); TAG=$1; shift
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/c5tr
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/c5tr -- python $ROOT/bench.py --config 5 --no-cpu --no-profile --no-aux --no-latency --steps 10 --warmup 3 "$@" > /tmp/c5tr.log 2>&1
f=$(ls -S $(find /tmp/c5tr -name "*kernel_trace.csv") | head -1)
cp "$f" $ROOT/gpurun_out/c5trace_$TAG.csv
