#!/usr/bin/env python3
"""simrank_plan_create at config 5 (N = 65536 SimRank++), three calls (the first pays the allocator)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simrank_amd import ingest, synth
from simrank_amd.engine import HipOps, Plan
ops = HipOps(0)
csr = ingest.directed(synth.WORKLOADS["pl65536"][0](), False, "from", "to", "weight")[1]
scale = ingest.spread(csr) * csr.rowscale
for i in range(3):
    t0 = time.perf_counter()
    plan = Plan(ops, csr, scale, coef=0.8, evidence=True)
    ops.synchronize()
    t1 = time.perf_counter()
    done, conv = plan.run(100, 1e-4)
    t2 = time.perf_counter()
    idx, val = plan.topk(10)
    t3 = time.perf_counter()
    plan.free()
    t4 = time.perf_counter()
    print(f"call {i}: create {t1 - t0:.4f} s, run {t2 - t1:.4f} s ({done} updates), top-10 {t3 - t2:.4f} s, free {t4 - t3:.4f} s", flush=True)
