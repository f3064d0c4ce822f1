#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for i in 1 2 3 4; do
python bench.py 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.load(sys.stdin)
print('run $i c2', round(d['value']), round(d['kernels']['k_feat_gate3']['avg_ms'],4), {k:(round(v['value']), round(v['kernels']['k_feat_gate3']['avg_ms'],4)) for k,v in d['aux_configs'].items()})"
done
