#!/bin/bash
# probe of the fused flip (DESIGN 4.4 / HISTORY 10): launch = k_post_marg behind the write-back; flip = the solve flips; dummy = the solve flips and an empty launch follows
cd "$(dirname "$0")/.."
for rep in 1 2; do for v in launch dummy flip; do
  unset INGVIO_FLIP INGVIO_FLIP_DUMMY
  if [ $v = launch ]; then export INGVIO_FLIP=launch; fi
  if [ $v = dummy ]; then export INGVIO_FLIP_DUMMY=1; fi
  python bench.py --config 2 --no-cpu --no-aux --no-latency --detail gpurun_out/ab_$v.json 2>/dev/null | tail -1 > gpurun_out/ab_line_$v.json
  python - $v <<'PY'
import json, sys
v = sys.argv[1]
d = json.load(open("gpurun_out/ab_line_%s.json" % v)); k = json.load(open("gpurun_out/ab_%s.json" % v))["kernels"]
print(v, "ms/step", round(d["ms_per_step"], 4), " ".join("%s %.1f" % (n.replace("k_", "").replace("feat_", ""), 1e3 * e["avg_ms"]) for n, e in k.items()))
PY
done; done
