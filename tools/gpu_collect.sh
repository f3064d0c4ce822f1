#!/bin/bash
# Profile collection of a round in ONE GPU call: for every workload rocprofv3 kernel stats + the PMC passes (tools/gpu_counters.sh ->
# gpurun_out/counters.json with the library's build id, kernel_stats_<tag>.csv, counters_<tag>.csv), then the bench line of the same
# workload priced with those counters (gpurun_out/bench_<tag>.json + bench_<tag>_detail.json).
#   tools/gpu_collect.sh [c2 c3 c5 c2n87 c2n93 c5lit c2lm ...]      (default: c2 c3 c5)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
W="${*:-c2 c3 c5}"
args() { case $1 in
  c2) echo "--config 2";; c3) echo "--config 3";; c5) echo "--config 5";;
  c2n87) echo "--config 2 --state literal";; c2n93) echo "--config 2 --state gnss";; c5lit) echo "--config 5 --state literal";;
  c2lm) echo "--config 2 --landmarks real";; c2mono) echo "--config 2 --mono";; c2c12) echo "--config 2 --clones 12";; esac; }
key() { case $1 in
  c2) echo c2_B512_F150_C11_N249;; c3) echo c3_B512_F150_C11_N249;; c5) echo c5_B32_F300_C30_N807;;
  c2n87) echo c2_B512_F150_C11_N87;; c2n93) echo c2_B512_F150_C11_N93;; c5lit) echo c5_B32_F300_C30_N201;;
  c2lm) echo c2_B512_F150_C11_N249_lmreal;; c2mono) echo c2_B512_F150_C11_N249_mono;; c2c12) echo c2_B512_F150_C12_N255;; esac; }
for w in $W; do
  PS=""
  case $w in c5|c5lit) PS="k_chol_step=11,k_chol_first=2,k_gemm64=3";; esac
  T0=$(date +%s)
  PER_STEP=$PS bash tools/gpu_counters.sh $w $(key $w) $(args $w)
  EXTRA="--no-aux --no-latency"
  [ $w = c2 ] && EXTRA=""
  [ $w = c2 ] || EXTRA="$EXTRA --no-cpu"
  python bench.py $(args $w) $EXTRA --counters gpurun_out/counters.json --detail gpurun_out/bench_${w}_detail.json 2> gpurun_out/bench_$w.err | tail -1 > gpurun_out/bench_$w.json
  python - $w <<'PY'
import json, sys
w = sys.argv[1]
d = json.load(open("gpurun_out/bench_%s.json" % w)); r = d.get("roofline") or {}
print(w, "value", round(d["value"]), "ms/step", round(d["ms_per_step"], 4), "roofline", r.get("kernel"), "frac", r.get("frac"), "useful", r.get("useful_frac"),
      "stale", r.get("counters_stale"), "whole", d.get("whole_step_frac_fp64_peak"), "parity", (d.get("parity_vs_oracle") or {}).get("max_rel_cov_err"))
PY
  echo "  $w took $(( $(date +%s) - T0 )) s"
done
