#!/usr/bin/env python3
"""Write the one-launch plan of leg 1 (unit records + the id streams of its gather phase + the dense sets' ids) of a
workload to a file, exactly as `simrank_plan_create` builds it — the input of tools/micro/gather_depth.hip.

    python3 tools/dump_fused_plan.py [--workload pl32768d32] --out gpurun_out/plan_pl32768d32.bin
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="pl32768d32")
ap.add_argument("--out", required=True)
args = ap.parse_args()
os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
os.environ["SIMRANK_DUMP_FUSED_PLAN"] = os.path.abspath(args.out)

from simrank_amd import ingest, synth                              # noqa: E402
from simrank_amd.engine import HipOps, Plan                        # noqa: E402

df, kind = synth.WORKLOADS[args.workload]
assert kind == "directed"
_, csr = ingest.directed(df(), False, "from", "to", "weight")
plan = Plan(HipOps(0), csr, coef=0.8)
plan.free()
print("wrote", args.out, os.path.getsize(args.out), "bytes", flush=True)
