#!/bin/bash
# builds build_var/<name>/libingvio_hip.so from the objects of the regular build (ingvio_amd/lib/*.o) with ONE translation unit
# recompiled with extra hipcc flags: tools/build_tu_variant.sh <name> <tu without .hip> [flags...]
# select it at run time with INGVIO_HIP_LIB=$ROOT/build_var/<name>/libingvio_hip.so
NAME=$1; TU=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/build_var/$NAME
mkdir -p "$OUT"
EXTRA=""
if [ $TU = kernels_bigwin ]; then EXTRA="-mllvm -amdgpu-sched-strategy=max-ilp"; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value $EXTRA "$@" -c "$ROOT/ingvio_amd/csrc/$TU.hip" -o "$OUT/$TU.o" || exit 1
OBJS="$OUT/$TU.o"
for f in "$ROOT"/ingvio_amd/lib/*.o; do
  [ "$(basename "$f")" = "$TU.o" ] || OBJS="$OBJS $f"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libingvio_hip.so" $OBJS && ls -la "$OUT/libingvio_hip.so"
