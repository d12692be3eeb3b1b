#!/usr/bin/env python3
"""Cost of cutting a rank's leg 1 into pipeline stages: one transposed launch over the rank's
columns against 2 / 4 / 8 launches over slices (what TorchWorld(stages=k) queues), on one GPU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simrank_amd import ingest, synth
from tests.pydriver import SideSpec, reorder_specs
from simrank_amd.engine import HipOps

ops = HipOps(0)
df = synth.WORKLOADS["pl32768"][0]()
_, csr = ingest.directed(df, False, "from", "to", "weight")
specs, _ = reorder_specs([SideSpec(csr, csr.rowscale, 0.8)])
c = specs[0].csr
n = c.n_rows
g = ops.graph(c)
for P in (2, 4, 8):
    L = n // P
    X = ops.matrix(n, L)
    ops.fill_identity(X, 0)
    Y = ops.matrix(L, n)
    for stages in (1, 2, 4, 8):
        w = L // stages
        def run():
            for s in range(stages):
                ops.spmm(g, X, Y, n_cols=w, transpose_out=True, x_col0=s * w, y_offset=s * w * Y.ld)
        run(); run()
        e0, e1 = ops.event(), ops.event()
        ops.record(e0)
        for _ in range(5):
            run()
        ops.record(e1)
        print(f"P={P} (L={L}) stages={stages}: leg 1 {ops.elapsed_ms(e0, e1) / 5:.3f} ms", flush=True)
    X.free(); Y.free()
