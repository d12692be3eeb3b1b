#!/usr/bin/env python3
"""Where the set-up time of a SimRank++ fit goes (SimRank.py:311-337 + _create_graph): ingest, graph
objects (host: transposed pattern, dense / fused plans), evidence counts (device), live-segment count.

    python3 tools/setup_breakdown.py [--workload pl65536]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simrank_amd import ingest, synth                              # noqa: E402
from simrank_amd.engine import HipOps                              # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="pl65536")
ap.add_argument("--set", default="")
args = ap.parse_args()
ops = HipOps(0)
if args.set:
    ops.set_tuning(**{k: int(v) for k, v in (kv.split("=") for kv in args.set.split(","))})
df, kind = synth.WORKLOADS[args.workload][0](), synth.WORKLOADS[args.workload][1]
t = time.perf_counter()
if kind == "directed":
    _, csr = ingest.directed(df, False, "from", "to", "weight")
    csrs = [csr]
else:
    _, _, _, _, g12, g21 = ingest.bipartite(df, False, "user", "item", "weight")
    csrs = [g12, g21]
print(f"ingest {time.perf_counter() - t:.3f} s", flush=True)
for c in csrs:
    outdeg = np.bincount(c.col, minlength=c.n_cols).astype(np.float64)
    print(f"pattern {c.n_rows} x {c.n_cols}, nnz {c.nnz}: 2-hop paths sum outdeg^2 = {np.sum(outdeg ** 2):.3e}")
    t = time.perf_counter()
    g = ops.graph(c)
    ops.synchronize()
    print(f"  graph object (host plans + upload) {time.perf_counter() - t:.3f} s", flush=True)
    ev = ops.matrix(c.n_rows, c.n_rows, np.uint8, blocked=True)
    ops.synchronize()
    for rep in range(2):
        t = time.perf_counter()
        ops.evidence_counts(g, 0, ev)
        ops.synchronize()
        print(f"  evidence counts (device) {time.perf_counter() - t:.3f} s", flush=True)
    t = time.perf_counter()
    live = ops.evidence_live_fraction(ev)
    print(f"  live segments {live:.3f}: {time.perf_counter() - t:.3f} s", flush=True)
