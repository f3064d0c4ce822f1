cd /root/repo
bash tests/gpu_collect_r03.sh c2lm
python bench.py --landmarks real --no-cpu --no-aux 2> gpurun_out/bench_c2_lmreal.err | tail -1 > gpurun_out/bench_c2_lmreal.json
python -c "import json; d=json.load(open('gpurun_out/bench_c2_lmreal.json')); print(round(d['value']), round(d['ms_per_step'],4), {k: round(v['avg_ms'],4) for k,v in d['kernels'].items()})"
