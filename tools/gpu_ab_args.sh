#!/bin/bash
# A/B of the regular library against another build on any bench arguments (per-kernel HIP-event times), interleaved in one call.
# usage (GPU box): bash tools/gpu_ab_args.sh build_var/<old>/libingvio_hip.so <bench args...>
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
OLD=$1; shift
for rep in 1 2 3; do for v in new old; do
  if [ $v = old ]; then export INGVIO_HIP_LIB=/root/repo/$OLD; else unset INGVIO_HIP_LIB; fi
  python bench.py "$@" --no-cpu --no-aux --no-latency --detail gpurun_out/ab_$v.json 2>/dev/null | tail -1 > gpurun_out/ab_line_$v.json
  python - $v <<'PY'
import json, sys
v = sys.argv[1]
d = json.load(open("gpurun_out/ab_line_%s.json" % v)); k = json.load(open("gpurun_out/ab_%s.json" % v))["kernels"]
print(v, "ms/step", round(d["ms_per_step"], 4), " ".join("%s %.1f" % (n.replace("k_", "").replace("feat_", ""), 1e3 * e["avg_ms"]) for n, e in k.items()))
PY
done; done
