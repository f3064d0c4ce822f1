#!/bin/bash
# quick iteration on the large-window path: its parity tests, then the config-5 bench line with the per-kernel table
# usage (GPU box): bash tools/gpu_c5_iter.sh [extra bench args]
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config4.py tests/test_gpu_pinning.py -q -m gpu -x -k "config5 or big or large or window or clones" 2>&1 | tail -5
python bench.py --config 5 --no-aux --no-latency --no-cpu --detail gpurun_out/bench_c5_iter_detail.json "$@" 2> gpurun_out/bench_c5_iter.err | tail -1 > gpurun_out/bench_c5_iter.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_c5_iter.json"))
print("c5 value", round(d["value"]), "ms/step", round(d["ms_per_step"], 4), "parity", d.get("parity_vs_oracle"))
det = json.load(open("gpurun_out/bench_c5_iter_detail.json"))
for k in ("kernels", "per_kernel", "kernel_table", "profile"):
    if k in det:
        print(k, json.dumps(det[k])[:3000])
PY
tail -3 gpurun_out/bench_c5_iter.err
# per-kernel times of the same command (rocprofv3 --kernel-trace --stats)
ROOT=$(pwd); export TMPDIR=/tmp
( cd /tmp && rm -rf /tmp/c5stats && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c5stats -- python $ROOT/bench.py --config 5 --no-cpu --no-profile --no-aux --no-latency --steps 20 --warmup 5 "$@" > /tmp/c5stats.log 2>&1 )
find /tmp/c5stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/kernel_stats_c5_iter.csv \;
python - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/kernel_stats_c5_iter.csv")))
for r in rows[:22]:
    print("%-70s calls %5s avg_us %8.2f pct %5s" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
