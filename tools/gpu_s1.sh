#!/bin/bash
# round-4 session 1: slots per CU, grouping, timeline of the one-launch leg 1
set -u
export PYTHONUNBUFFERED=1
O=gpurun_out/r04_s1.log
: > $O
FST=$PWD/gpurun_variants/libsimrank_hip_fst.so
echo "== product lib" >> $O
python tools/leg_only.py --workload pl32768d32 --steps 5 >> $O 2>&1
python tools/leg_only.py --workload pl32768d32 --steps 5 --set fuse_group=1 >> $O 2>&1
python tools/leg_only.py --workload pl32768d32 --steps 5 --set fuse_group=2 >> $O 2>&1
echo "== stamps lib, LDS pad (workgroups per CU)" >> $O
for pad in 0 20000 46000 90000; do
  echo "pad $pad" >> $O
  SIMRANK_LIB=$FST SIMRANK_FUSED_PAD=$pad python tools/leg_only.py --workload pl32768d32 --steps 5 >> $O 2>&1
done
echo "pad 20000 group 1" >> $O
SIMRANK_LIB=$FST SIMRANK_FUSED_PAD=20000 python tools/leg_only.py --workload pl32768d32 --steps 5 --set fuse_group=1 >> $O 2>&1
echo "pad 46000 group 1" >> $O
SIMRANK_LIB=$FST SIMRANK_FUSED_PAD=46000 python tools/leg_only.py --workload pl32768d32 --steps 5 --set fuse_group=1 >> $O 2>&1
echo "== timeline" >> $O
SIMRANK_LIB=$FST SIMRANK_FST_BASE=40320 python tools/fused_stamps.py pl32768d32 --out gpurun_out/r04_fst_def.npz >> $O 2>&1
SIMRANK_LIB=$FST SIMRANK_FST_BASE=40320 python tools/fused_stamps.py pl32768d32 --set fuse_group=1 --out gpurun_out/r04_fst_g1.npz >> $O 2>&1
tail -60 $O
