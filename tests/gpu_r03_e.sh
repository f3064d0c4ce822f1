#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/gputest_e.log 2>&1
echo "pytest rc=$?"; tail -6 gpurun_out/gputest_e.log
INGVIO_HIP_LIB=$PWD/build_var/stamps/libingvio_hip.so timeout 300 python tests/gpu_phase_times.py 512 2>&1 | grep "propagate\|gate3"
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu --no-aux > gpurun_out/bench_e.json 2> gpurun_out/bench_e.err
python - <<'PY'
import json
try:
    p = json.load(open("gpurun_out/bench_e.json"))
    print("value", round(p["value"]), "ms/step", round(p["ms_per_step"], 4), {k: round(v["avg_ms"], 4) for k, v in p["kernels"].items()})
    print("roofline", {k: v for k, v in p["roofline"].items() if k in ("kernel", "frac", "achieved", "lane_utilisation", "useful_frac", "avg_launch_ms")})
except Exception as e:
    print("bench parse failed", e); print(open("gpurun_out/bench_e.err").read()[-1500:])
PY
