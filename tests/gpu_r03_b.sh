#!/bin/bash
# round 3: full GPU suite + bench A/B of the gate generations
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -s > gpurun_out/gputest_b.log 2>&1
echo "pytest rc=$?"; tail -15 gpurun_out/gputest_b.log
grep -h "^sweep\|config 4 per-GPU\|R^T R vs\|large-window sweep\|FD pin" gpurun_out/gputest_b.log | tail -40
for G in 4 3; do
  INGVIO_GATE=$G timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu --no-aux > gpurun_out/bench_gate$G.json 2> gpurun_out/bench_gate$G.err
  python - <<PY
import json
try:
    p = json.load(open("gpurun_out/bench_gate$G.json"))
    print("gate$G value", round(p["value"]), "ms/step", round(p["ms_per_step"], 4), {k: round(v["avg_ms"], 4) for k, v in p["kernels"].items()})
except Exception as e:
    print("bench parse failed", e); print(open("gpurun_out/bench_gate$G.err").read()[-1500:])
PY
done
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r03b.json 2> gpurun_out/bench_r03b.err
python - <<'PY'
import json
try:
    p = json.load(open("gpurun_out/bench_r03b.json"))
    print("value", p["value"], "ms/step", p["ms_per_step"], "parity", p["parity_vs_oracle"])
    for c, a in p.get("aux_configs", {}).items():
        print(c, a["value"], a["ms_per_step"], a["parity_vs_oracle"], {k: round(v["avg_ms"], 4) for k, v in a["kernels"].items()})
except Exception as e:
    print("bench parse failed", e)
PY
