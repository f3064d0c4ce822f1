#!/bin/bash
# A/B of several library builds (build_var/<name>/libingvio_hip.so, tools/build_variant.sh) on the config-5 bench: step and the Gram / gate
# kernels' HIP-event times, interleaved in one call (box-to-box spread is larger than most single steps).
# usage (GPU box): bash tools/gpu_gram_ab.sh <variant> [<variant> ...]        e.g.  bash tools/gpu_gram_ab.sh alt stamps
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for rep in 1 2; do for v in "$@"; do
INGVIO_HIP_LIB=$PWD/build_var/$v/libingvio_hip.so python bench.py --config 5 --no-cpu --no-aux --no-latency --detail gpurun_out/gab_$v.json 2>/dev/null | tail -1 > gpurun_out/gab_line_$v.json
python - $v <<'PY'
import json, sys
v = sys.argv[1]
d = json.load(open("gpurun_out/gab_line_%s.json" % v)); k = json.load(open("gpurun_out/gab_%s.json" % v))["kernels"]
print(v, "ms/step", round(d["ms_per_step"], 4), "gram us", round(1e3 * k["gram"]["avg_ms"], 1), "gate us", round(1e3 * k["gate"]["avg_ms"], 1))
PY
done; done
