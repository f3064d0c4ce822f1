#!/bin/bash
# Round-2 profile collection for the workloads whose kernels changed late in the round (run on the GPU box through gpurun).
cd /root/repo
PER_STEP=k_chol_step=10,k_chol_first=2,k_chol_carried=2,k_gemm=4 bash tests/gpu_counters.sh c5 c5_B32_F300_C30_N807 --config 5
PER_STEP=k_chol_step=10,k_chol_first=2,k_chol_carried=2,k_gemm=4 bash tests/gpu_counters.sh c5lit c5_B32_F300_C30_N201 --config 5 --state literal
PER_STEP=k_chol_step=6 bash tests/gpu_counters.sh c2lm c2_B512_F150_C11_N249_lmreal --config 2 --landmarks real
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/qrstats -- python /root/repo/tests/gpu_qr_bench.py > /root/repo/gpurun_out/qr_bench.log 2>&1
cd /root/repo; find gpurun_out/qrstats -name "*kernel_stats.csv" -exec cp {} gpurun_out/kernel_stats_qr.csv \;
tail -1 gpurun_out/qr_bench.log
cd /root/repo
bash tests/gpu_counters.sh c2 c2_B512_F150_C11_N249 --config 2
bash tests/gpu_counters.sh c3 c3_B512_F150_C11_N249 --config 3
bash tests/gpu_counters.sh c2n87 c2_B512_F150_C11_N87 --config 2 --state literal
bash tests/gpu_counters.sh c2n93 c2_B512_F150_C11_N93 --config 2 --state gnss
