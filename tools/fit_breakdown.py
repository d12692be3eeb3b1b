#!/usr/bin/env python3
"""Where the wall-clock of a whole fit goes (second call: allocator warm).
    python3 tools/fit_breakdown.py [f32|fp16] [workload = pl65536] [pp|plain] [top10|full] [plan|python]
(plan = cplan.PlanSolver, what fit() runs on one GPU since round 5; python = driver.Solver)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simrank_amd import ingest, synth
from tests.pydriver import LocalWorld, SideSpec, Solver
from simrank_amd.engine import HipOps

storage = sys.argv[1] if len(sys.argv) > 1 else "f32"
ops = HipOps(0)
wl = sys.argv[2] if len(sys.argv) > 2 else "pl65536"
pp = (sys.argv[3] if len(sys.argv) > 3 else "pp") == "pp"
full = (sys.argv[4] if len(sys.argv) > 4 else "top10") == "full"
use_plan = (sys.argv[5] if len(sys.argv) > 5 else "plan") == "plan"
from simrank_amd.cplan import PlanSolver
df = synth.WORKLOADS[wl][0]()
for rep in range(2):
    t = [time.perf_counter()]
    def lap(name):
        ops.synchronize(); t.append(time.perf_counter()); print(f"  {name:28s} {t[-1] - t[-2]:.3f} s", flush=True)
    print("call", rep, storage)
    _, csr = ingest.directed(df, False, "from", "to", "weight"); lap("ingest")
    spec = (SideSpec(csr, ingest.spread(csr) * csr.rowscale, 0.8, evidence_from=csr, storage=storage) if pp
            else SideSpec(csr, csr.rowscale, 0.8, storage=storage)); lap("spread weights")
    s = (PlanSolver(ops, LocalWorld(1), [spec]) if use_plan
         else Solver(lambda r: ops, LocalWorld(1), [spec], "auto")); lap("solver (graph, evidence, S)")
    k = s.run(100, 1e-4); lap(f"run to eps (k={k})")
    if full:
        import pandas as pd
        res = s.result(0); lap("result (f64, caller's order)")
        frame = pd.DataFrame(res, index=_, columns=_); lap("DataFrame")
        del res, frame; lap("free the host copies")
    else:
        idx, val = s.topk(0, 10); lap("top-10 hand-back")
    s.release(); lap("release")
    print(f"  total {t[-1] - t[0]:.3f} s")
