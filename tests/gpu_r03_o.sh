#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for B in 1 512; do
INGVIO_DBG_TU=m INGVIO_HIP_LIB=$PWD/build_var/dbg/libingvio_hip.so timeout 300 python tests/gpu_phase_lm.py $B 2>&1 | tail -1
done
