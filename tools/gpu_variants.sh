#!/bin/bash
# Build-flag experiments: bench lines of libraries built with other scheduler strategies (build_var/<name>/libingvio_hip.so).
cd /root/repo
line() { python bench.py "$@" --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],4), d['results_finite'], {k:round(v['avg_ms'],3) for k,v in d['kernels'].items()})"; }
for v in ${VARIANTS:-base maxilp memclause}; do
  if [ $v = base ]; then unset INGVIO_HIP_LIB; else export INGVIO_HIP_LIB=/root/repo/build_var/$v/libingvio_hip.so; fi
  echo "== $v"; for a in "$@"; do line $a; done
done
