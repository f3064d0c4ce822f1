#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_landmark_batch.py -m gpu -q -x > gpurun_out/gputest_r.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/gputest_r.log | cut -c1-300
for B in 32 1; do
timeout 600 python bench.py --config 5 --batch $B --steps 30 --warmup 5 --no-cpu --no-aux 2>/dev/null | python -c "
import json,sys
p=json.load(sys.stdin); print('default B=$B ms/step', round(p['ms_per_step'],4), round(p['value']), p['results_finite'], {k: round(v['avg_ms'],4) for k,v in p['kernels'].items()})"
done
