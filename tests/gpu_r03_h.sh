#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tests/gpu_qr_shapes.py 2>&1 | tail -12
R=$PWD
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c5 -- python $R/bench.py --config 5 --no-cpu --no-profile --no-aux --steps 20 --warmup 5 > $R/gpurun_out/prof_c5.log 2>&1
cd $R; f=$(ls -t $(find gpurun_out/prof_c5 -name "*kernel_stats.csv") | head -1); cp $f gpurun_out/kernel_stats_c5.csv; python - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:22]:
    print(r["Name"][:70].ljust(70), r["Calls"].rjust(5), "%9.1f us avg %10.1f us total" % (float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e3))
PY
tail -1 gpurun_out/prof_c5.log | cut -c1-300
