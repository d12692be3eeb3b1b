#!/bin/bash
# Experiment build of the library with extra -D flags next to the product one (same ABI):
#   bash tools/build_variant.sh d8 -DSIMRANK_DEPTH8     ->  gpurun_variants/libsimrank_hip_d8.so
# run with  SIMRANK_LIB=$PWD/gpurun_variants/libsimrank_hip_d8.so python tools/leg_only.py ...
set -e
TAG=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_variants
mkdir -p $OUT/obj_$TAG
for f in api spmm dense blockdense fused fused2 half planprep plan biplan; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$ROOT/include -fvisibility=hidden -DSIMRANK_BUILD "$@" \
    -c $ROOT/simrank_amd/csrc/$f.hip -o $OUT/obj_$TAG/$f.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OUT/obj_$TAG/*.o -lpthread -o $OUT/libsimrank_hip_$TAG.so
ls -la $OUT/libsimrank_hip_$TAG.so
