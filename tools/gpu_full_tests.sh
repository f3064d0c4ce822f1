#!/bin/bash
# the round-end GPU checks in one call: the whole -m gpu suite, smoke(), the default bench line (+ its detail file), the latency streams
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/gputest_full.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/gputest_full.log | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
T0=$(date +%s); python bench.py --detail gpurun_out/bench_detail.json 2> gpurun_out/bench_default.err | tail -1 > gpurun_out/bench_default.json; echo "bench wall $(( $(date +%s) - T0 )) s, line $(wc -c < gpurun_out/bench_default.json) bytes"
cat gpurun_out/bench_default.json
tail -3 gpurun_out/bench_default.err
