#!/bin/bash
# Experiment build of the library with extra -D flags next to the product one (same ABI):
#   bash tools/build_variant.sh d8 -DSIMRANK_DEPTH8     ->  gpurun_variants/libsimrank_hip_d8.so
#   bash tools/build_variant.sh f2 -DSIMRANK_EXPERIMENT_FUSED2   (the persistent leg 1 of round 4; its tests:
#        SIMRANK_LIB=$PWD/gpurun_variants/libsimrank_hip_f2.so python -m pytest tools/experiments/test_gpu_fused2.py -m gpu)
# run with  SIMRANK_LIB=$PWD/gpurun_variants/libsimrank_hip_d8.so python tools/leg_only.py ...
set -e
TAG=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_variants
mkdir -p $OUT/obj_$TAG
SRC=$ROOT/simrank_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$ROOT/include -I$SRC -fvisibility=hidden -DSIMRANK_BUILD"
for f in api spmm dense blockdense fused half planprep plan biplan shardplan handback; do
  /opt/rocm/bin/hipcc $FLAGS "$@" -c $SRC/$f.hip -o $OUT/obj_$TAG/$f.o &
done
# experiments that lost their A/B live outside the product library (tools/experiments/): the persistent leg 1
# (round 4, fuse = 2) is compiled in only with -DSIMRANK_EXPERIMENT_FUSED2
case " $* " in *SIMRANK_EXPERIMENT_FUSED2*)
  /opt/rocm/bin/hipcc $FLAGS "$@" -c $ROOT/tools/experiments/fused2.hip -o $OUT/obj_$TAG/fused2.o & ;;
esac
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OUT/obj_$TAG/*.o -lpthread -o $OUT/libsimrank_hip_$TAG.so
ls -la $OUT/libsimrank_hip_$TAG.so
