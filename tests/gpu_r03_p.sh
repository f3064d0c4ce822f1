#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for B in 1 512; do
echo "== B=$B"
INGVIO_HIP_LIB=$PWD/build_var/dbg/libingvio_hip.so timeout 300 python tests/gpu_phase_times.py $B 2>&1 | tail -9
done
