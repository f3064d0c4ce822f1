#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for G in 1 2 3 4; do
INGVIO_GRAM_CHUNKS=$G timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu --no-aux 2>/dev/null | python -c "
import json,sys
p=json.load(sys.stdin); print('G=$G ms/step', round(p['ms_per_step'],4), round(p['value']), p['results_finite'], {k: round(v['avg_ms'],4) for k,v in p['kernels'].items()})"
done
