#!/bin/bash
# ms per step against the number of timed steps (fixed overhead of the timed region vs steady state): bash tools/gpu_steps_sweep.sh [bench args]
cd "$(dirname "$0")/.."
for rep in 1 2; do for K in 20 50 100 200 1000; do for P in "" "--no-profile"; do
  python bench.py --config 2 --no-cpu --no-aux --no-latency --steps $K --warmup 5 $P "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('K=%5d %-13s ms/step %.4f  total %.3f ms' % ($K, '$P', d['ms_per_step'], d['ms_per_step']*$K))"
done; done; done
