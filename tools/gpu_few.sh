#!/bin/bash
# The few-filter path (k_chunk_sum + flat apply): parity tests, then the B = 1 / 8 / 32 step against the same kernels a full batch takes
# (alt library, INGVIO_FEW=off).  usage (GPU box): bash tools/gpu_few.sh
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/few
python -m pytest tests/test_gpu_parity.py tests/test_gpu_pinning.py tests/test_stream_golden.py -x -q -m gpu 2>&1 | tail -5
for B in 1 8 32; do
  python bench.py --config 2 --batch $B --steps 30 --warmup 5 --no-cpu --no-aux --no-latency --detail gpurun_out/few/d_$B.json 2>/dev/null | tail -1 > gpurun_out/few/b_$B.json
  INGVIO_HIP_LIB=/root/repo/build_var/alt/libingvio_hip.so INGVIO_FEW=off python bench.py --config 2 --batch $B --steps 30 --warmup 5 --no-cpu --no-aux --no-latency --detail gpurun_out/few/d0_$B.json 2>/dev/null | tail -1 > gpurun_out/few/b0_$B.json
  python - $B <<'PY'
import json, sys
B = sys.argv[1]
a = json.load(open("gpurun_out/few/b_%s.json" % B)); o = json.load(open("gpurun_out/few/b0_%s.json" % B))
d = json.load(open("gpurun_out/few/d_%s.json" % B)); d0 = json.load(open("gpurun_out/few/d0_%s.json" % B))
print("B", B, "ms/step few", a["ms_per_step"], "as a full batch", o["ms_per_step"], "parity", (a.get("parity_vs_oracle") or {}).get("max_rel_cov_err"))
for k in d["kernels"]:
    print("   %-16s %8.1f us   (full-batch kernels %8.1f)" % (k, 1e3 * d["kernels"][k]["avg_ms"], 1e3 * d0["kernels"].get(k, {}).get("avg_ms", float("nan"))))
PY
done
