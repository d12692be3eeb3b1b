#!/usr/bin/env python3
"""Would a K-split of leg 1 pay?  Times the transposed gather leg on the bench graph restricted to
the entries whose column is in one K-range (half / quarter panels fit the XCD's L2), against the
whole pattern."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simrank_amd import ingest, synth
from simrank_amd.ingest import CSR
from tests.pydriver import SideSpec, reorder_specs
from simrank_amd.engine import HipOps

ops = HipOps(0)
w = sys.argv[1] if len(sys.argv) > 1 else "pl32768"
df = synth.WORKLOADS[w][0]()
_, csr = ingest.directed(df, False, "from", "to", "weight")
specs, _ = reorder_specs([SideSpec(csr, csr.rowscale, 0.8)])
c = specs[0].csr
n = c.n_rows
X = ops.matrix(n, n); Y = ops.matrix(n, n)
ops.fill_identity(X, 0)
rows = np.repeat(np.arange(n), np.diff(c.rowptr))

def restricted(lo, hi):
    keep = (c.col >= lo) & (c.col < hi)
    rp = np.concatenate([[0], np.cumsum(np.bincount(rows[keep], minlength=n))]).astype(np.int32)
    return CSR(n, n, rp, c.col[keep].astype(np.int32), c.rowscale)

def timed(g, reps=5):
    for _ in range(2):
        ops.spmm(g, X, Y, transpose_out=True)
    e0, e1 = ops.event(), ops.event()
    ops.record(e0)
    for _ in range(reps):
        ops.spmm(g, X, Y, transpose_out=True)
    ops.record(e1)
    return ops.elapsed_ms(e0, e1) / reps

for dmin in (0, 4):
    ops.set_tuning(dense_min=dmin)
    full = timed(ops.graph(c))
    print(f"{w} dense_min={dmin}: whole pattern {full:.3f} ms", flush=True)
    for parts in (2, 4):
        ts = []
        for k in range(parts):
            g = ops.graph(restricted(n * k // parts, n * (k + 1) // parts))
            ts.append(timed(g))
        print(f"   {parts} K-ranges: " + " + ".join(f"{t:.3f}" for t in ts) + f" = {sum(ts):.3f} ms", flush=True)
ops.set_tuning(dense_min=4)
