#!/bin/bash
# counters + traces of the remaining state-size variants with the round-3 kernels (their r02 entries still named k_feat_gate3)
cd /root/repo
bash tests/gpu_counters.sh c2n87 c2_B512_F150_C11_N87 --config 2 --state literal
bash tests/gpu_counters.sh c2n93 c2_B512_F150_C11_N93 --config 2 --state gnss
PER_STEP=k_chol_step=11,k_chol_first=2,k_gemm=4,k_big_gauge_fix=2 bash tests/gpu_counters.sh c5lit c5_B32_F300_C30_N201 --config 5 --state literal
