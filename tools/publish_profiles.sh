#!/bin/bash
# copies the summaries of a collection (tools/gpu_collect.sh, tools/gpu_full_tests.sh, tools/gpu_final_b1.sh) from gpurun_out/ (scratch)
# into profiles/ (tracked) under the round's prefix: bash tools/publish_profiles.sh r06
cd "$(dirname "$0")/.."
R=${1:-r06}; G=gpurun_out; P=profiles
cp $G/counters.json $P/counters.json
for w in c2 c3 c5 c2n87 c2n93 c5lit c2lm c2mono c2c12; do
  [ -f $G/bench_$w.json ] && cp $G/bench_$w.json $P/${R}_bench_$w.json
  [ -f $G/bench_${w}_detail.json ] && cp $G/bench_${w}_detail.json $P/${R}_bench_${w}_detail.json
  [ -f $G/kernel_stats_$w.csv ] && cp $G/kernel_stats_$w.csv $P/${R}_rocprofv3_kernel_stats_$w.csv
  [ -f $G/counters_$w.csv ] && cp $G/counters_$w.csv $P/${R}_pmc_$w.csv
done
[ -f $G/bench_default.json ] && cp $G/bench_default.json $P/${R}_bench_default_line.json
[ -f $G/bench_detail.json ] && cp $G/bench_detail.json $P/${R}_bench_default_detail.json
for t in c2_b1 c5_b1 c5_b128; do
  [ -f $G/final_b1/bench_$t.json ] && cp $G/final_b1/bench_$t.json $P/${R}_bench_$t.json
  [ -f $G/final_b1/bench_${t}_detail.json ] && cp $G/final_b1/bench_${t}_detail.json $P/${R}_bench_${t}_detail.json
done
for t in c2_b1 c5_b1; do [ -f $G/final_b1/kernel_stats_$t.csv ] && cp $G/final_b1/kernel_stats_$t.csv $P/${R}_rocprofv3_kernel_stats_$t.csv; done
for w in config2 kf21 kf27 sw11 kf35_mono; do [ -f $G/final_b1/replay_kernel_stats_$w.csv ] && cp $G/final_b1/replay_kernel_stats_$w.csv $P/${R}_replay_kernel_stats_$w.csv; done
python tools/kernel_resources.py > $P/${R}_kernel_resources.txt 2>/dev/null
git status --short $P | wc -l
