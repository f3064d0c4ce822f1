#!/bin/bash
# A/B of one library under two environments on a bench config (per-kernel HIP-event times), interleaved in one call.
# usage (GPU box): bash tools/gpu_env_ab.sh "VAR=value" [config] [extra bench args]
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
KV=$1; CFG=${2:-2}; shift 2
for rep in 1 2 3; do for v in new old; do
  if [ $v = old ]; then export "$KV"; else unset "${KV%%=*}"; fi
  python bench.py --config $CFG --no-cpu --no-aux --no-latency "$@" --detail gpurun_out/ab_$v.json 2>/dev/null | tail -1 > gpurun_out/ab_line_$v.json
  python - $v <<'PY'
import json, sys
v = sys.argv[1]
d = json.load(open("gpurun_out/ab_line_%s.json" % v)); k = json.load(open("gpurun_out/ab_%s.json" % v))["kernels"]
print(v, "ms/step", round(d["ms_per_step"], 4), " ".join("%s %.1f" % (n.replace("k_", "").replace("feat_", ""), 1e3 * e["avg_ms"]) for n, e in k.items()))
PY
done; done
