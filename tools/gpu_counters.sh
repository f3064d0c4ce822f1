#!/bin/bash
# Collects, on the GPU box, everything profiles/ needs for one bench workload:
#   1. rocprofv3 --kernel-trace --stats of the bench command        -> gpurun_out/prof_<tag>/stats
#   2. rocprofv3 --pmc passes (counters only, no trace domains)      -> gpurun_out/prof_<tag>/pmc_*  -> profiles/counters.json
# usage: [PER_STEP=k_chol_step=12,k_gemm=4] tools/gpu_counters.sh <tag> <counters key> <bench args...>      e.g.  tools/gpu_counters.sh c2 c2_B512_F150_C11_N249 --config 2
#   PER_STEP: kernels launched several times per bench step (stages made of several launches), see tools/pmc_summary.py
set -u
TAG=$1; KEY=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
BENCH="python $ROOT/bench.py --no-cpu --no-profile --no-aux --no-latency --no-condition --steps 3 --warmup 1 $*"      # counters are per launch: no conditioning steps (they only grow the csv)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python $ROOT/bench.py --no-cpu --no-profile --no-aux --no-latency --steps 20 --warmup 5 "$@" > "$OUT/stats.log" 2>&1
PASSES=(
 "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES"
 "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU"
 "FETCH_SIZE"
 "WRITE_SIZE"
 "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INST_CYCLES_VALU"
)
i=0
DIRS=""
for P in "${PASSES[@]}"; do
  timeout 600 rocprofv3 --pmc $P --output-format csv -d "$OUT/pmc_$i" -- $BENCH > "$OUT/pmc_$i.log" 2>&1
  DIRS="$DIRS $OUT/pmc_$i"
  i=$((i+1))
done
cd "$ROOT"
python tools/pmc_summary.py --key "$KEY" --last 3 --per-step "${PER_STEP:-}" --out gpurun_out/counters.json --csv "gpurun_out/counters_$TAG.csv" \
  --source "rocprofv3 --pmc (5 passes) -- python bench.py --no-cpu --no-profile --steps 3 --warmup 1 $*; last 3 dispatches per kernel" $DIRS
find "$OUT/stats" -name "*kernel_stats.csv" -exec cp {} "gpurun_out/kernel_stats_$TAG.csv" \;
rm -rf $DIRS      # the raw per-dispatch counter files: folded above (gpurun copies at most 64 MiB back)
tail -2 "$OUT/stats.log"
