cd /root/repo
PER_STEP=k_chol_step=10,k_chol_first=2,k_chol_carried=2,k_gemm=4 bash tests/gpu_counters.sh c5 c5_B32_F300_C30_N807 --config 5
PER_STEP=k_chol_step=10,k_chol_first=2,k_chol_carried=2,k_gemm=4 bash tests/gpu_counters.sh c5lit c5_B32_F300_C30_N201 --config 5 --state literal
cp profiles/counters.json /tmp/counters_old.json 2>/dev/null
run() { tag=$1; shift; python bench.py "$@" 2> gpurun_out/bench_$tag.err | tail -1 > gpurun_out/bench_$tag.json; python -c "import json; d=json.load(open('gpurun_out/bench_$tag.json')); print('$tag', round(d['value']), round(d['ms_per_step'],4), d['results_finite'], (d.get('roofline') or {}).get('frac'))"; }
run c5_n807 --config 5
run c5_n201 --config 5 --state literal --no-cpu
run c5_n807_b128 --config 5 --batch 128 --no-cpu
run c5_n807_b1 --config 5 --batch 1 --no-cpu --steps 50
