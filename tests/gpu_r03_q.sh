#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_host_shim.py tests/test_gpu_config4.py -m gpu -q -x > gpurun_out/gputest_q.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/gputest_q.log | cut -c1-300
ingvio_amd/lib/test_host_shim 2>&1 | grep -i "adjust_yof" | head -5
for i in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu --no-aux 2>/dev/null | python -c "
import json,sys
p=json.load(sys.stdin); print('ms/step', round(p['ms_per_step'],4), round(p['value']), {k: round(v['avg_ms'],4) for k,v in p['kernels'].items()})"
done
