#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
for A in split fused; do
echo "== front $A"
INGVIO_LM_FRONT=$A timeout 600 python -m pytest tests/test_landmark_batch.py -m gpu -q 2>&1 | tail -8 | cut -c1-200
done
