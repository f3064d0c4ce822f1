#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py -k "large_window or config5" -m gpu -q -x > gpurun_out/gputest_v.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/gputest_v.log | cut -c1-300
for P in 1 2 4; do
for B in 32 128; do
INGVIO_BIG_PARTS=$P timeout 600 python bench.py --config 5 --batch $B --steps 30 --warmup 5 --no-cpu --no-aux 2>/dev/null | python -c "
import json,sys
p=json.load(sys.stdin); print('parts=$P B=$B ms/step', round(p['ms_per_step'],4), round(p['value']), p['results_finite'], {k: round(v['avg_ms'],4) for k,v in p['kernels'].items()})"
done; done
