#!/usr/bin/env python3
"""simrank_biplan_create on the MovieLens-shaped graph, three calls (SIMRANK_TIME_BUILD=1: the builders' durations on stderr)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simrank_amd import ingest, synth
from simrank_amd.engine import BiPlan, HipOps
ops = HipOps(0)
_, _, _, _, g12, g21 = ingest.bipartite(synth.WORKLOADS["ml1m"][0](), False, "user", "item", "weight")
rs1, rs2 = ingest.spread(g12) * g12.rowscale, ingest.spread(g21) * g21.rowscale
for i in range(3):
    t0 = time.perf_counter()
    plan = BiPlan(ops, g12, rs1, rs2, evidence=True)
    ops.synchronize()
    t1 = time.perf_counter()
    done, conv = plan.run(100, 1e-4)
    t2 = time.perf_counter()
    plan.free()
    t3 = time.perf_counter()
    print(f"call {i}: create {t1 - t0:.4f} s, run {t2 - t1:.4f} s ({done} updates), free {t3 - t2:.4f} s", flush=True)
