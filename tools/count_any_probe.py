#!/usr/bin/env python3
"""Leg 2 with the exact convergence count, with the short-circuit test (count_any) and without any
comparison (no previous iterate): what the reads of the previous iterate cost."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simrank_amd import ingest, synth                              # noqa: E402
from tests.pydriver import LocalWorld, Side, SideSpec, Solver  # noqa: E402
from simrank_amd.engine import HipOps                              # noqa: E402

ops = HipOps(0)
w = sys.argv[1] if len(sys.argv) > 1 else "pl32768"
df = synth.WORKLOADS[w][0]()
_, csr = ingest.directed(df, False, "from", "to", "weight")
s = Solver(lambda r: ops, LocalWorld(1), [SideSpec(csr, csr.rowscale, 0.8)], "sparse")
s.reset()
for _ in range(2):
    s.step(0.0)
orig = Side._ep
for name in ("exact", "any", "none", "exact", "any", "none"):
    s.exact_count = name == "exact"
    if name == "none":
        Side._ep = lambda self, S_prev, eps: dict(orig(self, S_prev, eps), previous=None)
    else:
        Side._ep = orig
    s.step(0.0)
    s.enable_timing()
    counts = [s.step(0.0) for _ in range(5)]
    t = s.leg_times()
    print(name, {k: round(v[0], 3) for k, v in t.items() if k.startswith("leg")}, "count", counts[-1], flush=True)
