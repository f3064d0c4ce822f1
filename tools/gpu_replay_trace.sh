#!/bin/bash
# rocprofv3 kernel + memory-copy trace of the single-filter latency stream through the C++ shim (what the device does per camera callback).
# usage (GPU box): bash tools/gpu_replay_trace.sh [config2|config5]
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
W=${1:-config2}
case $W in
  config2) SPEC="feats=150,clones=11,life=10,cohort=1,birth_frame=3,frames=45,key=1";;
  config5) SPEC="feats=300,clones=30,life=28,cohort=1,birth_frame=2,frames=95,key=1";;
  kf27) SPEC="feats=150,clones=27,life=25,cohort=0,frames=110,key=1";;      # the sports-field stereo window, staggered track deaths (6 lost features per frame)
  kf21) SPEC="feats=100,clones=21,life=19,cohort=0,frames=90,key=1";;
  sw11) SPEC="feats=150,clones=11,life=13,cohort=0,frames=90,key=0";;      # sliding-window mode at 11 poses: 12 clones at update time
  kf35_mono) SPEC="feats=150,clones=35,life=33,cohort=0,frames=120,key=1,stereo=0";;
esac
EXTRA=()
case $W in sw11) EXTRA=(--set "frame_select_interval: 5");; esac
case $W in config2|config5) EXTRA=(--set "hip_max_valid_ids: 0" --set "hip_compress_rule: 1");; esac
OUT=gpurun_out/replay_trace_$W
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --memory-copy-trace --stats -d $OUT -o rt --output-format csv -- ingvio_amd/lib/ingvio_replay --synth "$SPEC" --time "${EXTRA[@]}" > $OUT/stdout.txt 2> $OUT/stderr.txt
tail -1 $OUT/stdout.txt
f=$(find $OUT -name '*kernel_stats.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%-52s calls %4s avg %8.1f us min %8.1f max %8.1f" % (r['Name'][:52], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
f=$(find $OUT -name '*memory_copy_stats.csv' | head -1); [ -n "$f" ] && cat "$f" | cut -c1-160
