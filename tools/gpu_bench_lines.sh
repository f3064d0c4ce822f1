#!/bin/bash
# All bench lines of the round in one GPU call (each into gpurun_out/bench_<tag>.json).
cd /root/repo; mkdir -p gpurun_out
run() { tag=$1; shift; python bench.py "$@" 2> gpurun_out/bench_$tag.err | tail -1 > gpurun_out/bench_$tag.json; python -c "import json; d=json.load(open('gpurun_out/bench_$tag.json')); print('$tag', round(d['value']), round(d['ms_per_step'],4), d['results_finite'], (d.get('roofline') or {}).get('frac'))"; }
run c2
run c2_n87 --state literal --no-cpu --no-aux
run c2_n93 --state gnss --no-cpu --no-aux
run c3 --config 3 --no-cpu --no-aux
run c5_n807 --config 5 --no-aux
run c5_n201 --config 5 --state literal --no-cpu --no-aux
run c5_n807_b128 --config 5 --batch 128 --no-cpu --no-aux
run c5_n807_b1 --config 5 --batch 1 --no-cpu --steps 50 --no-aux
run c2_lmreal --landmarks real --no-cpu --no-aux
python tools/gpu_qr_shapes.py > gpurun_out/qr_shapes.log 2>&1; tail -5 gpurun_out/qr_shapes.log
