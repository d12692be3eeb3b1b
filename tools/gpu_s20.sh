#!/bin/bash
# single rank, panel-blocked layout: rows appended to every panel (the distance between panels modulo the channel hash)
out=gpurun_out/r04_block_pad_rows.log
: > $out
for pad in 8 24 40 72 136 264 520; do
  echo "== SIMRANK_BLOCK_PAD_ROWS=$pad" >> $out
  SIMRANK_BLOCK_PAD_ROWS=$pad timeout -k 10 200 python tools/leg_only.py --workload pl32768d32 --steps 5 >> $out 2>&1 || exit 1
done
cat $out
