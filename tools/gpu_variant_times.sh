#!/bin/bash
# per-stage kernel times of the default workload for a list of library variants (build_var/<name>, tools/build_tu_variant.sh):
#   tools/gpu_variant_times.sh [bench args --] name1 name2 ...     ("base" = the regular library)
cd "$(dirname "$0")/.."
ROOT=$(pwd)
mkdir -p gpurun_out
ARGS=""
if [[ " $* " == *" -- "* ]]; then
  while [ "$1" != "--" ]; do ARGS="$ARGS $1"; shift; done; shift
fi
for v in "$@"; do
  if [ "$v" = base ]; then unset INGVIO_HIP_LIB; else export INGVIO_HIP_LIB=$ROOT/build_var/$v/libingvio_hip.so; fi
  python bench.py --no-cpu --no-aux --no-latency --steps 20 --warmup 3 $ARGS --detail gpurun_out/var_$v.json > gpurun_out/var_$v.log 2>&1
  python - "$v" <<'PY'
import json, sys
v = sys.argv[1]
try:
    d = json.load(open("gpurun_out/var_%s.json" % v))
    print("%-14s step %.4f ms  " % (v, d["ms_per_step"]) + "  ".join("%s %.4f" % (k.replace("k_", ""), e["avg_ms"]) for k, e in d["kernels"].items()), " finite", d["results_finite"],
          " parity", (d.get("parity_vs_oracle") or {}).get("max_rel_cov_err"))
except Exception as e:
    print(v, "FAILED", e); print(open("gpurun_out/var_%s.log" % v).read()[-1500:])
PY
done
