#!/usr/bin/env python3
"""A few updates of one workload with given tuning knobs — the program rocprofv3 wraps when a
counter pass is about ONE setting (tools/pmc_knob.sh).

    python3 tools/leg_only.py [--workload pl32768] [--steps 3] [--set balance=1,dense_min=4]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simrank_amd import ingest, synth                              # noqa: E402
from tests.pydriver import LocalWorld, SideSpec, Solver        # noqa: E402
from simrank_amd.engine import HipOps                              # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="pl32768")
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--set", default="")
ap.add_argument("--storage", default="f32", choices=["f32", "fp16"])
ap.add_argument("--pp", action="store_true", help="SimRank++ (evidence counts in the epilogue)")
ap.add_argument("--exact", action="store_true", help="exact convergence count (every element compared): what bench.py's headline times")
args = ap.parse_args()
if "probe" in args.set:
    os.environ["SIMRANK_ENABLE_PROBES"] = "1"       # diagnostic knobs (wrong results, timing only)
ops = HipOps(0)
if args.set:
    ops.set_tuning(**{k: int(v) for k, v in (kv.split("=") for kv in args.set.split(","))})
df, kind = synth.WORKLOADS[args.workload]
df = df()
if kind == "bipartite":            # BASELINE config 3: BipartiteSimRankPP (corrected evidence), both groups per step
    _, _, _, _, g12, g21 = ingest.bipartite(df, False, "user", "item", "weight")
    specs = [SideSpec(g12, g12.rowscale, 0.8, evidence_from=g12), SideSpec(g21, g21.rowscale, 0.8, evidence_from=g21)]
else:
    _, csr = ingest.directed(df, False, "from", "to", "weight")
    spec = SideSpec(csr, csr.rowscale, 0.8, storage=args.storage)
    if args.pp:
        spec = SideSpec(csr, ingest.spread(csr) * csr.rowscale, 0.8, evidence_from=csr, storage=args.storage)
    specs = [spec]
s = Solver(lambda r: ops, LocalWorld(1), specs, "sparse")
s.exact_count = args.exact
import time                                                        # noqa: E402
s.reset()
s.step(0.0)
ops.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):                                        # wall clock, no events
    s.step(0.0)
ops.synchronize()
wall = (time.perf_counter() - t0) / args.steps * 1e3
s.enable_timing()
for _ in range(args.steps):
    s.step(0.0)
ops.synchronize()
print(args.set, args.storage, {k: round(v[0], 3) for k, v in s.leg_times().items()}, f"wall {wall:.3f} ms/step", flush=True)
