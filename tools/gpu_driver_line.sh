#!/bin/bash
# the driver's command, N times, with the figures of the line that matter: bash tools/gpu_driver_line.sh [n] [extra bench args]
cd "$(dirname "$0")/.."
N=${1:-2}; shift
for i in $(seq $N); do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 "$@" 2>/dev/null | tail -1 > /tmp/line.json
  python3 - <<'PY'
import json
d = json.load(open("/tmp/line.json"))
print("value", d["value"], "ms/step", d["ms_per_step"], "cond", d.get("device_conditioning"),
      "aux", {k: (v["value"], v["ms_per_step"]) for k, v in (d.get("aux_configs") or {}).items()},
      "roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], "bytes", len(open("/tmp/line.json").read()))
PY
done
