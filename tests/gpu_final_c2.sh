#!/bin/bash
# final collection of the round for the workloads whose kernels changed last (k_info_solve): counters first, then (after the merge on
# the build box) the bench lines
cd /root/repo
bash tests/gpu_collect_r03.sh c2
bash tests/gpu_collect_r03.sh c3
