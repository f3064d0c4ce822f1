#!/bin/bash
# Landmark path: register-resident solve (kernels_lmchol.hip) against the sweep out of L2 (INGVIO_LM_SOLVE=sweep)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_landmark_batch.py tests/test_landmark_path.py tests/test_gpu_parity.py tests/test_gpu_config4.py -m gpu -q -x > gpurun_out/gputest_l.log 2>&1
echo "pytest rc=$?"; tail -6 gpurun_out/gputest_l.log | cut -c1-300
for A in sweep regs; do
INGVIO_LM_SOLVE=$A timeout 600 python bench.py --landmarks real --steps 20 --warmup 5 --no-cpu --no-aux 2>gpurun_out/bench_l_$A.err | python -c "
import json,sys
p=json.load(sys.stdin); print('$A ms/step', round(p['ms_per_step'],4), round(p['value']), p['results_finite'], {k: round(v['avg_ms'],4) for k,v in p['kernels'].items()})"
done
