#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pinning.py tests/test_gpu_config4.py -m gpu -q -x > gpurun_out/gputest_k.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/gputest_k.log | cut -c1-300
for A in 2 1; do
INGVIO_APPLY=$A timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu --no-aux 2>/dev/null | python -c "
import json,sys
p=json.load(sys.stdin); print('apply$A ms/step', round(p['ms_per_step'],4), round(p['value']), {k: round(v['avg_ms'],4) for k,v in p['kernels'].items()})"
done
INGVIO_APPLY=2 timeout 600 python bench.py --state literal --steps 20 --warmup 5 --no-cpu --no-aux 2>/dev/null | python -c "
import json,sys
p=json.load(sys.stdin); print('N87 apply2 ms/step', round(p['ms_per_step'],4), round(p['value']), {k: round(v['avg_ms'],4) for k,v in p['kernels'].items()})"
INGVIO_APPLY=1 timeout 600 python bench.py --state literal --steps 20 --warmup 5 --no-cpu --no-aux 2>/dev/null | python -c "
import json,sys
p=json.load(sys.stdin); print('N87 apply1 ms/step', round(p['ms_per_step'],4), round(p['value']), {k: round(v['avg_ms'],4) for k,v in p['kernels'].items()})"
