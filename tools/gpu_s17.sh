#!/bin/bash
# leg 1 / leg 2 of the shards under different row pads (channel hashing of power-of-two pitches)
set -o pipefail
out=gpurun_out/r04_shard_pads.log
: > $out
for pad in 32 96 160 224; do
  echo "== SIMRANK_ROW_PAD=$pad SIMRANK_PITCH_PAD=$pad" >> $out
  SIMRANK_ROW_PAD=$pad SIMRANK_PITCH_PAD=$pad PS=4,8 timeout -k 10 250 python tools/emulate_shards.py pl32768d32 >> $out 2>&1 || exit 1
done
tail -30 $out | cut -c1-260
