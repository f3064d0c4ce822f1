#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/gputest_g.log 2>&1
echo "pytest rc=$?"; tail -12 gpurun_out/gputest_g.log | cut -c1-400
