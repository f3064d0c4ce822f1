#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pinning.py tests/test_gpu_config4.py -m gpu -q -x > gpurun_out/gputest_f.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/gputest_f.log
INGVIO_HIP_LIB=$PWD/build_var/stamps/libingvio_hip.so timeout 300 python tests/gpu_phase_times.py 512 2>&1 | grep "propagate"
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu --no-aux > gpurun_out/bench_f.json 2> gpurun_out/bench_f.err
python - <<'PY'
import json
try:
    p = json.load(open("gpurun_out/bench_f.json"))
    print("value", round(p["value"]), "ms/step", round(p["ms_per_step"], 4), {k: round(v["avg_ms"], 4) for k, v in p["kernels"].items()})
except Exception as e:
    print("bench parse failed", e); print(open("gpurun_out/bench_f.err").read()[-1500:])
PY
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/prof_f -- python $OLDPWD/bench.py --no-cpu --no-profile --no-aux --steps 20 --warmup 5 > $OLDPWD/gpurun_out/prof_f.log 2>&1
cd $OLDPWD; find gpurun_out/prof_f -name "*kernel_stats.csv" -exec head -8 {} \;
