#!/bin/bash
# The single-filter figures of a round in one call: device-resident step of ONE filter (configs 2 and 5) and of 128 filters (config 5)
# with per-kernel HIP-event times, rocprofv3 kernel stats of the same, kernel + copy trace of the shim's streams.
# usage (GPU box): bash tools/gpu_final_b1.sh
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/final_b1
export TMPDIR=/tmp
for spec in "2 1 c2_b1" "5 1 c5_b1" "5 128 c5_b128"; do
  set -- $spec
  python bench.py --config $1 --batch $2 --no-cpu --no-aux --no-latency --detail gpurun_out/final_b1/bench_$3_detail.json 2>/dev/null | tail -1 > gpurun_out/final_b1/bench_$3.json
  python - $3 <<'PY'
import json, sys
d = json.load(open("gpurun_out/final_b1/bench_%s.json" % sys.argv[1])); print(sys.argv[1], "value", d["value"], "ms/step", d["ms_per_step"])
PY
done
bash tools/gpu_b1_profile.sh > gpurun_out/final_b1/b1_profile.txt 2>&1
cp gpurun_out/b1/kernel_stats_c2.csv gpurun_out/final_b1/kernel_stats_c2_b1.csv; cp gpurun_out/b1/kernel_stats_c5.csv gpurun_out/final_b1/kernel_stats_c5_b1.csv
for w in config2 kf21 kf27 sw11 kf35_mono; do
  bash tools/gpu_replay_trace.sh $w > gpurun_out/final_b1/replay_$w.txt 2>&1
  cp gpurun_out/replay_trace_$w/rt_kernel_stats.csv gpurun_out/final_b1/replay_kernel_stats_$w.csv
  head -1 gpurun_out/final_b1/replay_$w.txt | cut -c1-200
done
