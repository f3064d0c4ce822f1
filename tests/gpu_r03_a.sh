#!/bin/bash
# round 3, first GPU call: gauge diagnostic (off / on), full GPU suite, bench with aux configs, counter availability
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
( INGVIO_INFO_GAUGE=off timeout 300 python tests/gpu_gauge_diag.py ) > gpurun_out/gauge_off.txt 2>&1
( timeout 300 python tests/gpu_gauge_diag.py ) > gpurun_out/gauge_on.txt 2>&1
tail -8 gpurun_out/gauge_off.txt; tail -8 gpurun_out/gauge_on.txt
timeout 900 python -m pytest tests -m gpu -x -q -s > gpurun_out/gputest_a.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/gputest_a.log
grep -h "^sweep\|config 4 per-GPU\|R^T R vs\|large-window sweep" gpurun_out/gputest_a.log | tail -60
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r03a.json 2> gpurun_out/bench_r03a.err
echo "bench rc=$?"; tail -3 gpurun_out/bench_r03a.err
python - <<'PY'
import json
try:
    p = json.load(open("gpurun_out/bench_r03a.json"))
    print("value", p["value"], "ms/step", p["ms_per_step"], "parity", p["parity_vs_oracle"])
    for k, v in p["kernels"].items(): print("  ", k, round(v["avg_ms"], 4))
    for c, a in p.get("aux_configs", {}).items():
        print(c, a["value"], a["ms_per_step"], a["parity_vs_oracle"], a["results_finite"])
        for k, v in a["kernels"].items(): print("  ", k, round(v["avg_ms"], 4))
except Exception as e:
    print("bench parse failed", e)
PY
(cd /tmp && rocprofv3 -L 2>/dev/null | grep -i -E "THREAD_CYCLES|ACTIVE_INST_VALU|INST_CYCLES_VALU|VALUUtil|SQ_INSTS_VALU " | head -20) > gpurun_out/counters_avail.txt 2>&1
cat gpurun_out/counters_avail.txt | head -20
