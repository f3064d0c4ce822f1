#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
run() { tag=$1; shift; python bench.py "$@" 2> gpurun_out/bench_$tag.err | tail -1 > gpurun_out/bench_$tag.json; python -c "import json; d=json.load(open('gpurun_out/bench_$tag.json')); print('$tag', round(d['value']), round(d['ms_per_step'],4), d['results_finite'], (d.get('roofline') or {}).get('frac'))"; }
run c2_n87 --state literal --no-cpu --no-aux
run c2_n93 --state gnss --no-cpu --no-aux
run c5_n201 --config 5 --state literal --no-cpu --no-aux
