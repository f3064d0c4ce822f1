#!/bin/bash
# B = 1, config 2: kernel times against the number of Gram chunks (INGVIO_GRAM_CHUNKS).  usage (GPU box): bash tools/gpu_b1_chunks.sh
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/b1
export TMPDIR=/tmp
for g in 1 2 4 8 16; do
  echo "== chunks $g"
  INGVIO_GRAM_CHUNKS=$g rocprofv3 --kernel-trace --stats -d gpurun_out/b1/g$g -o b1 --output-format csv -- python bench.py --config 2 --batch 1 --steps 20 --warmup 3 --no-cpu --no-aux --no-latency --no-profile --detail gpurun_out/b1/detail_g$g.json > gpurun_out/b1/bench_g$g.json 2> gpurun_out/b1/err_g$g.txt
  f=$(find gpurun_out/b1/g$g -name '*kernel_stats.csv' | head -1)
  python - "$f" gpurun_out/b1/bench_g$g.json <<'PY'
import csv, sys, json
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r['Name'] for k in ('solve', 'gram', 'gate', 'apply')):
        print("  %-44s avg %9.1f us" % (r['Name'][:44], float(r['AverageNs'])/1e3))
print("  ms_per_step", json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])['ms_per_step'])
PY
  rm -rf gpurun_out/b1/g$g
done
