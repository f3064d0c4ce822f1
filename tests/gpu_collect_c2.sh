#!/bin/bash
# Headline workload: counters + kernel stats + the small-window bench lines in one GPU call.
cd /root/repo; mkdir -p gpurun_out
bash tests/gpu_counters.sh c2 c2_B512_F150_C11_N249 --config 2
run() { tag=$1; shift; python bench.py "$@" 2> gpurun_out/bench_$tag.err | tail -1 > gpurun_out/bench_$tag.json; python -c "import json; d=json.load(open('gpurun_out/bench_$tag.json')); print('$tag', round(d['value']), round(d['ms_per_step'],4), d['results_finite'], (d.get('roofline') or {}).get('frac'))"; }
run c2_n87 --state literal --no-cpu
run c2_n93 --state gnss --no-cpu
run c3 --config 3 --no-cpu
run c2_lmreal --landmarks real --no-cpu
