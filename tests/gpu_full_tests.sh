#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/gputest_full.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/gputest_full.log | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
