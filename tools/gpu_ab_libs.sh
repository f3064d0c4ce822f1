#!/bin/bash
# A/B/C of library builds on one bench configuration (ms per step + per-kernel HIP-event times), interleaved, in one call.
# usage (GPU box): bash tools/gpu_ab_libs.sh <config> <reps> new build_var/<a>/libingvio_hip.so build_var/<b>/libingvio_hip.so ... [-- bench args]
cd "$(dirname "$0")/.."
CFG=$1; REPS=$2; shift 2
LIBS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done; [ "$1" = "--" ] && shift
mkdir -p gpurun_out
for rep in $(seq 1 $REPS); do for v in "${LIBS[@]}"; do
  if [ $v = new ]; then unset INGVIO_HIP_LIB; else export INGVIO_HIP_LIB=/root/repo/$v; fi
  python bench.py --config $CFG --no-cpu --no-aux --no-latency --detail gpurun_out/ab_tmp.json "$@" 2>/dev/null | tail -1 > gpurun_out/ab_line_tmp.json
  python - "$v" <<'PY'
import json, sys
v = sys.argv[1]
d = json.load(open("gpurun_out/ab_line_tmp.json")); k = json.load(open("gpurun_out/ab_tmp.json"))["kernels"]
print("%-40s ms/step %.4f  " % (v[-40:], d["ms_per_step"]), " ".join("%s %.1f" % (n.replace("k_", "").replace("feat_", ""), 1e3 * e["avg_ms"]) for n, e in k.items()), flush=True)
PY
done; done
