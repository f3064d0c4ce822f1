#!/bin/bash
# samples the shader clock (rocm-smi) while the bench of a configuration runs: what "peak" means under this load
# usage (GPU box): bash tools/gpu_clock_sample.sh <config> [bench args]
cd "$(dirname "$0")/.."
CFG=$1; shift
python bench.py --config $CFG --no-cpu --no-aux --no-latency --no-profile --steps 4000 --warmup 20 "$@" > /tmp/clk_bench.json 2>/dev/null &
BP=$!
sleep 6
for i in 1 2 3 4 5 6; do
  /opt/rocm/bin/rocm-smi --showclocks 2>/dev/null | grep -i -E "sclk|mclk|fclk" | tr -s ' ' | head -4
  /opt/rocm/bin/rocm-smi --showpower 2>/dev/null | grep -i -E "power" | head -2
  sleep 1
done
wait $BP
tail -1 /tmp/clk_bench.json | cut -c1-200
