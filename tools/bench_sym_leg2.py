#!/usr/bin/env python3
"""Leg 2 of a symmetric update as ONE launch (tuning `fuse_sym`: fused.hip, SYM) against the two-launch leg (dense_tiles +
gather3<kSym>), through the loops fit() runs: ms per update of the C-level plans at fuse_sym = 0 / 1 / -1 (automatic).
`python3 tools/bench_sym_leg2.py [ml1m,pl32768d32,er8192,...]`."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simrank_amd import ingest, synth                      # noqa: E402
from simrank_amd.engine import BiPlan, HipOps, Plan        # noqa: E402

names = (sys.argv[1] if len(sys.argv) > 1 else "ml1m,er8192,pl32768d32").split(",")
ops = HipOps(0)
for name in names:
    df = synth.WORKLOADS[name][0]()
    for sym in (0, 1, -1, 0, 1):
        ops.set_tuning(fuse_sym=sym)
        if name == "ml1m":
            _, _, _, _, g12, g21 = ingest.bipartite(df, False, "user", "item", "weight")
            plan = BiPlan(ops, g12, ingest.spread(g12) * g12.rowscale, ingest.spread(g21) * g21.rowscale, evidence=True)
            reps = 200
        else:
            _, csr = ingest.directed(df, False, "from", "to", "weight")
            plan = Plan(ops, csr, rowscale=csr.rowscale)
            reps = 300 if csr.n_rows <= 8192 else 30
        plan.run(5, 0.0)
        best = 1e9
        for _ in range(3):
            plan.reset()
            ops.synchronize()
            t0 = time.perf_counter()
            done, _ = plan.run(reps, 0.0)                       # (eps = 0: ends early at a bit-wise fixed point)
            ops.synchronize()
            best = min(best, (time.perf_counter() - t0) / max(1, done))
        print(f"{name} fuse_sym={sym:2d}: {best * 1e3:.3f} ms per update ({1 / best:.1f} it/s; {done} updates per run)", flush=True)
        plan.free()
ops.set_tuning(fuse_sym=-1)
