#!/bin/bash
# Kernel trace of the single-filter step (B = 1) for config 2 and config 5: which kernels make the latency.
# usage (GPU box): bash tools/gpu_b1_profile.sh
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/b1
export TMPDIR=/tmp
for c in 2 5; do
  rocprofv3 --kernel-trace --stats -d gpurun_out/b1/c$c -o b1 --output-format csv -- python bench.py --config $c --batch 1 --steps 20 --warmup 3 --no-cpu --no-aux --no-latency --no-profile --detail gpurun_out/b1/detail_c$c.json > gpurun_out/b1/bench_c$c.json 2> gpurun_out/b1/err_c$c.txt
  f=$(find gpurun_out/b1/c$c -name '*kernel_stats.csv' | head -1)
  cp "$f" gpurun_out/b1/kernel_stats_c$c.csv
  python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%-44s calls %4s avg %9.1f us min %9.1f" % (r['Name'][:44], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
PY
  tail -c 400 gpurun_out/b1/bench_c$c.json | head -c 400; echo
  rm -rf gpurun_out/b1/c$c
done
