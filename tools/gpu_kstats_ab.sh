#!/bin/bash
# per-kernel average durations (rocprofv3 --kernel-trace --stats) of the regular library and another build on the same bench arguments
# usage (GPU box): bash tools/gpu_kstats_ab.sh build_var/<old>/libingvio_hip.so "<kernel name regex>" <bench args...>
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OLD=$1; PAT=$2; shift 2
export TMPDIR=/tmp
cd /tmp
for v in new old; do
  if [ $v = old ]; then export INGVIO_HIP_LIB=$ROOT/$OLD; else unset INGVIO_HIP_LIB; fi
  rm -rf /tmp/ks_$v
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$v -- python $ROOT/bench.py --no-cpu --no-profile --no-aux --no-latency --steps 20 --warmup 5 "$@" > /tmp/ks_$v.log 2>&1
  f=$(find /tmp/ks_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"
  python - "$f" "$PAT" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pat = re.compile(sys.argv[2])
for r in rows:
    if pat.search(r["Name"]):
        print("  %-60s calls %5s avg %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
