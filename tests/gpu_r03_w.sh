#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
echo "== power-of-two stride (INGVIO_P_PAD=0)"
INGVIO_P_PAD=0 timeout 900 python tests/gpu_alloc_sensitivity.py 6 2>&1 | tail -7
echo "== padded stride"
timeout 900 python tests/gpu_alloc_sensitivity.py 6 2>&1 | tail -7
