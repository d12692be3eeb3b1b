#!/bin/bash
# rocprofv3 kernel stats of the sharded path on one GPU: eight virtual ranks of pl32768d32 (Python driver, half-form leg 2),
# then config 5 on fp16-held matrices through the C sharded loop
set -u
OUT=$PWD/gpurun_out/prof_shards
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
PS=8 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/f32 -- python3 $GRAFT_REPO_ROOT/tools/emulate_shards.py pl32768d32 > $OUT/f32.log 2>&1
STORAGE=fp16 PP=1 PS=8 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/fp16 -- python3 $GRAFT_REPO_ROOT/tools/emulate_shards_c.py pl65536 > $OUT/fp16.log 2>&1
cd - > /dev/null
for d in f32 fp16; do
  f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1)
  echo "== $d"; head -12 "$f" | cut -c1-220
  cp "$f" gpurun_out/r04_kernel_stats_shards_$d.csv
done
