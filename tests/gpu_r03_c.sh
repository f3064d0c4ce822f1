#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29631 HSA_ENABLE_IPC_MODE_LEGACY=0 INGVIO_ROOT=$PWD timeout 300 python tests/gpu_rccl_ws1.py > gpurun_out/rccl_ws1.log 2>&1
echo "rccl rc=$?"; grep -v "^$" gpurun_out/rccl_ws1.log | tail -25
timeout 1200 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_pinning.py::test_rccl_world_size_1 > gpurun_out/gputest_c.log 2>&1
echo "pytest rc=$?"; tail -6 gpurun_out/gputest_c.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu --no-aux > gpurun_out/bench_c.json 2> gpurun_out/bench_c.err
python - <<'PY'
import json
try:
    p = json.load(open("gpurun_out/bench_c.json"))
    print("value", round(p["value"]), "ms/step", round(p["ms_per_step"], 4), {k: round(v["avg_ms"], 4) for k, v in p["kernels"].items()})
except Exception as e:
    print("bench parse failed", e); print(open("gpurun_out/bench_c.err").read()[-1500:])
PY
