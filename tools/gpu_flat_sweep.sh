#!/bin/bash
# Where the one-wave-per-tile apply stops paying: config-2 bench at 64 / 128 / 256 filters with the product library (threshold 64) and
# with a variant whose threshold is 256.  Build the variant first:  bash tools/build_variant.sh flat256 -DINGVIO_APPLY_FLAT_NB=256
# usage (GPU box): bash tools/gpu_flat_sweep.sh
cd /root/repo
python -m pytest tests/test_gpu_alternatives.py -x -q -m gpu -k "few or FEW" 2>&1 | tail -3
for B in 64 128 256; do
  for lib in ingvio_amd/lib build_var/flat256; do
    INGVIO_HIP_LIB=/root/repo/$lib/libingvio_hip.so python bench.py --config 2 --batch $B --steps 30 --warmup 5 --no-cpu --no-aux --no-latency --detail gpurun_out/few/dd.json 2>/dev/null | tail -1 > gpurun_out/few/bb.json
    python - $B $lib <<'PY'
import json, sys
a = json.load(open("gpurun_out/few/bb.json")); d = json.load(open("gpurun_out/few/dd.json"))
print("B", sys.argv[1], sys.argv[2], "ms/step", a["ms_per_step"], "apply us", round(1e3 * d["kernels"]["apply"]["avg_ms"], 1), "solve us", round(1e3 * d["kernels"]["solve"]["avg_ms"], 1))
PY
  done
done
