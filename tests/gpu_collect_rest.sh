#!/bin/bash
# Remaining small-window workloads: counters, then the bench lines computed against the refreshed counters.
cd /root/repo; mkdir -p gpurun_out
bash tests/gpu_counters.sh c2n93 c2_B512_F150_C11_N93 --config 2 --state gnss 2>&1 | tail -1
python tests/merge_counters.py c2_B512_F150_C11_N93
PER_STEP=k_chol_step=6 bash tests/gpu_counters.sh c2lm c2_B512_F150_C11_N249_lmreal --config 2 --landmarks real 2>&1 | tail -1
python tests/merge_counters.py c2_B512_F150_C11_N249_lmreal
run() { tag=$1; shift; python bench.py "$@" 2> gpurun_out/bench_$tag.err | tail -1 > gpurun_out/bench_$tag.json; python -c "import json; d=json.load(open('gpurun_out/bench_$tag.json')); print('$tag', round(d['value']), round(d['ms_per_step'],4), d['results_finite'], (d.get('roofline') or {}).get('frac'))"; }
run c2_n87 --state literal --no-cpu
run c2_n93 --state gnss --no-cpu
run c3 --config 3 --no-cpu
run c2_lmreal --landmarks real --no-cpu
