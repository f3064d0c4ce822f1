#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_landmark_batch.py tests/test_landmark_path.py tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/gputest_l.log 2>&1
echo "pytest rc=$?"; tail -6 gpurun_out/gputest_l.log | cut -c1-300
for A in split fused; do
INGVIO_LM_FRONT=$A timeout 600 python bench.py --landmarks real --steps 20 --warmup 5 --no-cpu --no-aux 2>gpurun_out/bench_l_$A.err | python -c "
import json,sys
p=json.load(sys.stdin); print('$A ms/step', round(p['ms_per_step'],4), round(p['value']), p['results_finite'], {k: round(v['avg_ms'],4) for k,v in p['kernels'].items()})"
done
R=$PWD
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_lm -- python $R/bench.py --landmarks real --steps 20 --warmup 5 --no-cpu --no-aux --no-profile > $R/gpurun_out/prof_lm.log 2>&1
cd $R
find gpurun_out/prof_lm -name "*kernel_stats.csv" -exec cp {} gpurun_out/kernel_stats_lm.csv \;
cut -d, -f1-4 gpurun_out/kernel_stats_lm.csv | sed 's/(CovView[^"]*"/"/; s/(anonymous namespace):://' | cut -c1-110 | head -16
