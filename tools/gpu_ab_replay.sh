#!/bin/bash
# A/B of two builds of libingvio_hip.so under the C++ shim's latency streams, interleaved in ONE call (boxes and clocks differ between
# calls by several percent).  usage (GPU box): bash tools/gpu_ab_replay.sh build_var/<name>/libingvio_hip.so [repetitions]
cd "$(dirname "$0")/.."
ALT=$1; N=${2:-3}
run() { # label, preload, spec, extra...
  local label=$1 pre=$2 spec=$3; shift 3
  local out
  out=$(LD_PRELOAD=$pre ingvio_amd/lib/ingvio_replay --synth "$spec" --time "$@" | tail -1)
  echo "$label $(echo "$out" | grep -o 'median_ms=[0-9.]*') $(echo "$out" | grep -o 'heavy_median_ms=[0-9.]*') $(echo "$out" | grep -o 'other_median_ms=[0-9.]*')"
}
for i in $(seq $N); do
  for spec in "feats=100,clones=21,life=19,cohort=0,frames=90,key=1" "feats=150,clones=27,life=25,cohort=0,frames=110,key=1"; do
    run "product ${spec:10:9}" "" "$spec"
    run "variant ${spec:10:9}" "$PWD/$ALT" "$spec"
  done
  spec="feats=150,clones=11,life=10,cohort=1,birth_frame=3,frames=75,key=1"
  run "product cohort11 " "" "$spec" --set "hip_max_valid_ids: 0" --set "hip_compress_rule: 1"
  run "variant cohort11 " "$PWD/$ALT" "$spec" --set "hip_max_valid_ids: 0" --set "hip_compress_rule: 1"
done
