#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29631 HSA_ENABLE_IPC_MODE_LEGACY=0 INGVIO_ROOT=$PWD timeout 300 python tests/gpu_rccl_ws1.py > gpurun_out/rccl_ws1.log 2>&1
echo "rccl rc=$?"; grep -v "^$" gpurun_out/rccl_ws1.log | tail -8
INGVIO_HIP_LIB=$PWD/build_var/stamps/libingvio_hip.so timeout 300 python tests/gpu_phase_times.py 512 > gpurun_out/phases_512.txt 2>&1
tail -9 gpurun_out/phases_512.txt
bash tests/gpu_counters.sh c2 c2_B512_F150_C11_N249 --config 2
python - <<'PY'
import json
J = json.load(open("gpurun_out/counters.json"))
for w, d in J["workloads"].items():
    for k, c in d["kernels"].items():
        print(w, k, {x: (round(v, 1) if isinstance(v, float) else v) for x, v in c.items() if x in ("SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_INSTS_VALU_MFMA_MOPS_F64", "FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU_FMA_F64")})
PY
head -12 gpurun_out/kernel_stats_c2.csv
