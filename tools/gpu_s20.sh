#!/bin/bash
# single rank, panel-blocked layout: rows appended to every panel — 8 against 40 / 72, alternating on one box
out=gpurun_out/r04_block_pad_rows2.log
: > $out
for rep in 1 2 3; do for pad in 8 72 40; do
  echo -n "pad $pad: " >> $out
  SIMRANK_BLOCK_PAD_ROWS=$pad timeout -k 10 200 python tools/leg_only.py --workload pl32768d32 --steps 8 >> $out 2>&1 || exit 1
done; done
for pad in 8 72; do
  echo -n "pl65536 pp pad $pad: " >> $out
  SIMRANK_BLOCK_PAD_ROWS=$pad timeout -k 10 200 python tools/leg_only.py --workload pl65536 --pp --steps 4 >> $out 2>&1 || exit 1
  echo -n "er8192 pad $pad: " >> $out
  SIMRANK_BLOCK_PAD_ROWS=$pad timeout -k 10 200 python tools/leg_only.py --workload er8192 --steps 20 >> $out 2>&1 || exit 1
done
cat $out
