#!/bin/bash
# On a node with several MI355X: the bench at 1, 2, 4, 8 GPUs, sharded leg 2 in both forms.
# Never run on the one-GPU box.   usage: bash tools/scale_forms.sh [steps]
set -u
STEPS=${1:-20}
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
for n in 1 2 4 8; do
  for form in full half; do
    [ $n = 1 ] && [ $form = half ] && continue
    echo "== $n GPU(s), leg 2 in its $form form"
    python bench.py --gpus $n --steps $STEPS --warmup 3 --no-extras --no-cpu-baseline --shard-form $form \
      | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k: d.get(k) for k in ('value', 'ms_per_step', 'n_gpus', 'rccl_ranks', 'exchange_ms')}, d['roofline']['ms'], d['roofline_other']['ms'])"
  done
done
