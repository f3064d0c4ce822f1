#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pinning.py tests/test_sharded_filter.py -m gpu -q -x -k "large or big or window or sweep or config5 or sharded or qr" > gpurun_out/gputest_j.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/gputest_j.log | cut -c1-300
timeout 600 python bench.py --config 5 --steps 20 --warmup 5 --no-cpu --no-aux 2>/dev/null | python -c "
import json,sys
p=json.load(sys.stdin); print('c5 B32 ms/step', round(p['ms_per_step'],4), round(p['value']), {k: round(v['avg_ms'],4) for k,v in p['kernels'].items()})"
timeout 600 python bench.py --config 5 --batch 1 --steps 50 --warmup 5 --no-cpu --no-aux 2>/dev/null | python -c "
import json,sys
p=json.load(sys.stdin); print('c5 single filter ms/step', round(p['ms_per_step'],4), {k: round(v['avg_ms'],4) for k,v in p['kernels'].items()})"
( export INGVIO_HIP_LIB=$PWD/build_var/stamps/libingvio_hip.so INGVIO_DBG_TU=b; python tests/gpu_phase_big.py 32 2>&1 | tail -1 )
