#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pinning.py tests/test_gpu_config4.py -m gpu -q -x > gpurun_out/gputest_u.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/gputest_u.log | cut -c1-300
run() { tag=$1; shift; timeout 600 python bench.py "$@" --steps 20 --warmup 5 --no-cpu --no-aux 2>/dev/null | python -c "
import json,sys
p=json.load(sys.stdin); print('$tag ms/step', round(p['ms_per_step'],4), round(p['value']), p['results_finite'], {k: round(v['avg_ms'],4) for k,v in p['kernels'].items()})"; }
run c2
run c2
run n87 --state literal
