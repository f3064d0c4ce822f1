#!/bin/bash
# builds a variant of libingvio_hip.so into build_var/<name>/ with extra hipcc flags (e.g. -DINGVIO_DBG_STAMPS for the
# shader-clock probes); select it at run time with INGVIO_HIP_LIB=/root/repo/build_var/<name>/libingvio_hip.so
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/build_var/$NAME
mkdir -p "$OUT"
cd "$ROOT/ingvio_amd/csrc"
OBJS=""
for f in kernels_cov kernels_msckf kernels_ekf kernels_factored kernels_solve kernels_bigwin kernels_tri kernels_lm kernels_qr kernels_chol kernels_lmbatch kernels_lmchol kernels_gnss kernels_tracks capi; do
  EXTRA=""
  if [ $f = kernels_bigwin ]; then EXTRA="-mllvm -amdgpu-sched-strategy=max-ilp"; fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value $EXTRA "$@" -c $f.hip -o "$OUT/$f.o" &
  OBJS="$OBJS $OUT/$f.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libingvio_hip.so" $OBJS "$ROOT/ingvio_amd/lib/build_id.o"      # the regular build's id object (ingvio_build_id): run python ingvio_amd/build.py first
ls -la "$OUT/libingvio_hip.so"
