#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for V in default pnt default pnt; do
L=""; if [ $V != default ]; then L=$PWD/build_var/$V/libingvio_hip.so; fi
INGVIO_HIP_LIB=$L timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu --no-aux 2>/dev/null | python -c "
import json,sys
p=json.load(sys.stdin); print('$V ms/step', round(p['ms_per_step'],4), round(p['value']), p['results_finite'], {k: round(v['avg_ms'],4) for k,v in p['kernels'].items()})"
done
