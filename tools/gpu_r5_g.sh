#!/bin/bash
# round 5, call G: the one-command multi-rank bench rehearsed on one GPU (a one-rank RCCL world), then the whole GPU suite
set -u
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
O=gpurun_out/r5g.log
: > $O
timeout -k 10 600 python bench.py --gpus 1 --force-dist --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r5g_dist1.json 2> gpurun_out/r5g_dist1.err || { tail -20 gpurun_out/r5g_dist1.err >> $O; }
python - >> $O <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r5g_dist1.json"))
    print("dist1:", d["value"], d["headline_loop"], d.get("rccl_ranks"))
    for k, v in d["sharded_c_loop"]["variants"].items():
        print("  ", k, {x: v.get(x) for x in ("value", "ms_per_step", "skipped", "error")}, v.get("rank0_events_ms"))
    print("  config5:", json.dumps(d["sharded_c_loop"].get("config5"))[:900])
    print("  form:", d["sharded_c_loop"].get("form_measured"), d.get("shard_form_measured"))
except Exception as e:
    print("dist1 failed:", e)
PY
echo "== full suite ==" >> $O
timeout -k 10 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -12 > gpurun_out/r5g_suite.log
tail -8 gpurun_out/r5g_suite.log >> $O
cat $O
