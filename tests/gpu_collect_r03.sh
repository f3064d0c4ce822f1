#!/bin/bash
# Round-3 profile collection (run on the GPU box through gpurun): per-kernel times + PMC counters of the bench workloads into
# gpurun_out/ (kernel_stats_<tag>.csv, counters_<tag>.csv, counters.json), then the bench lines and the QR shapes.
cd /root/repo
W=${1:-all}
if [ $W = all ] || [ $W = c2 ]; then bash tests/gpu_counters.sh c2 c2_B512_F150_C11_N249 --config 2; fi
if [ $W = all ] || [ $W = c3 ]; then bash tests/gpu_counters.sh c3 c3_B512_F150_C11_N249 --config 3; fi
if [ $W = all ] || [ $W = c5 ]; then PER_STEP=k_chol_step=11,k_chol_first=2,k_gemm=4,k_big_gauge_fix=2 bash tests/gpu_counters.sh c5 c5_B32_F300_C30_N807 --config 5; fi
if [ $W = all ] || [ $W = c2lm ]; then bash tests/gpu_counters.sh c2lm c2_B512_F150_C11_N249_lmreal --config 2 --landmarks real; fi
