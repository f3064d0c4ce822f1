#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/gputest_i.log 2>&1
echo "pytest rc=$?"; tail -8 gpurun_out/gputest_i.log | cut -c1-300
for G in 4 3; do
INGVIO_GATE=$G timeout 600 python bench.py --config 5 --steps 20 --warmup 5 --no-cpu --no-aux > gpurun_out/bench_c5_g$G.json 2> gpurun_out/bench_c5_g$G.err
python - <<PY
import json
try:
    p = json.load(open("gpurun_out/bench_c5_g$G.json"))
    print("gate$G c5 value", round(p["value"]), "ms/step", round(p["ms_per_step"], 4), {k: round(v["avg_ms"], 4) for k, v in p["kernels"].items()})
except Exception as e:
    print("bench parse failed", e); print(open("gpurun_out/bench_c5_g$G.err").read()[-1500:])
PY
done
timeout 600 python bench.py --config 5 --batch 1 --steps 50 --warmup 5 --no-cpu --no-aux 2>/dev/null | python -c "
import json,sys
p=json.load(sys.stdin); print('c5 single filter ms/step', round(p['ms_per_step'],4), {k: round(v['avg_ms'],4) for k,v in p['kernels'].items()})"
timeout 600 python tests/gpu_qr_shapes.py 2>&1 | grep auto
