#!/usr/bin/env python3
"""Where does the gather leg's time go: the popular source rows or the tail?

Splits the graph's entries by the popularity (in-degree as a source) of their column: the H
most referenced columns ("hubs") against the rest, and times leg 1 on each part alone.  If
the hub part dominates, serving hubs from LDS pays; if the tail does, it cannot.

    python tools/split_probe.py [--workload pl32768] [--hubs 1024,4096,8192]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simrank_amd import ingest, synth                              # noqa: E402
from tests.pydriver import LocalWorld, SideSpec, Solver        # noqa: E402
from simrank_amd.engine import HipOps                              # noqa: E402


def subset(csr, keep):
    rows = np.repeat(np.arange(csr.n_rows), np.diff(csr.rowptr))[keep]
    col = csr.col[keep]
    rowptr = np.zeros(csr.n_rows + 1, dtype=np.int64)
    np.add.at(rowptr, rows + 1, 1)
    return ingest.CSR(csr.n_rows, csr.n_cols, np.cumsum(rowptr).astype(np.int32), col.astype(np.int32),
                      csr.rowscale)


def legs(ops, csr):
    s = Solver(lambda r: ops, LocalWorld(1), [SideSpec(csr, csr.rowscale, 0.8)], "sparse")
    s.reset()
    for _ in range(2):
        s.step(0.0)
    s.enable_timing()
    for _ in range(4):
        s.step(0.0)
    t = s.leg_times()
    out = t["leg1.0"][0], t["leg2.0"][0]
    s.release()
    return out


ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="pl32768")
ap.add_argument("--hubs", default="1024,4096,8192")
args = ap.parse_args()

ops = HipOps(0)
df = synth.WORKLOADS[args.workload][0]()
_, csr = ingest.directed(df, False, "from", "to", "weight")
pop = np.bincount(csr.col, minlength=csr.n_cols)
order = np.argsort(-pop, kind="stable")
rank = np.empty_like(order)
rank[order] = np.arange(order.size)
l1, l2 = legs(ops, csr)
print(f"# {args.workload}: N={csr.n_rows} nnz={csr.nnz}  full graph: leg1 {l1:.3f} ms  leg2 {l2:.3f} ms",
      flush=True)
for h in [int(v) for v in args.hubs.split(",")]:
    is_hub = rank[csr.col] < h
    share = is_hub.mean()
    a = legs(ops, subset(csr, is_hub))
    b = legs(ops, subset(csr, ~is_hub))
    print(f"hubs={h:6d}: {100 * share:5.1f}% of entries | hub part leg1 {a[0]:7.3f} leg2 {a[1]:7.3f} ms"
          f" | tail part leg1 {b[0]:7.3f} leg2 {b[1]:7.3f} ms", flush=True)
