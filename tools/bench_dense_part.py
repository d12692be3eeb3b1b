#!/usr/bin/env python3
"""The dense-tile MFMA kernel alone (simrank_dense_part) by operand width, for the bench graph."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simrank_amd import ingest, synth
from tests.pydriver import SideSpec, reorder_specs
from simrank_amd.engine import HipOps

ops = HipOps(0)
w = sys.argv[1] if len(sys.argv) > 1 else "pl32768"
df = synth.WORKLOADS[w][0]()
_, csr = ingest.directed(df, False, "from", "to", "weight")
specs, _ = reorder_specs([SideSpec(csr, csr.rowscale, 0.8)])
g = ops.graph(specs[0].csr)
nt, dk, cov = ops.dense_stats(g)
n = csr.n_rows
X = ops.matrix(n, n)
ops.fill_identity(X, 0)
for L in (n, n // 2, n // 4, n // 8, n // 16, n // 32):
    for _ in range(2):
        ops.dense_part(g, X, L)
    e0, e1 = ops.event(), ops.event()
    ops.record(e0)
    for _ in range(5):
        ops.dense_part(g, X, L)
    ops.record(e1)
    ms = ops.elapsed_ms(e0, e1) / 5
    print(f"{w} dense part: {nt} row blocks, {dk} columns ({cov / csr.nnz:.3f} of the entries); L={L}: {ms:.3f} ms, "
          f"{2 * 3 * 128 * dk * L / ms / 1e9:.0f} TFLOP/s bf16", flush=True)
