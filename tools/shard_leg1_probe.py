#!/usr/bin/env python3
"""Leg 1 of ONE rank of a P-rank world on pl32768d32 (N x N/P column block), timed in the layouts it could use:
(a) panel-blocked operand, panel-blocked transposed result (what a single rank iterates in);
(b) row-major operand, chunked transposed result (what a rank of the sharded path holds: driver.Side);
(c) row-major operand, plain pitched result.  HIP events, 5 launches each."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simrank_amd import ingest, synth                           # noqa: E402
from tests.pydriver import SideSpec, reorder_specs, row_pad  # noqa: E402
from simrank_amd.engine import HipOps                           # noqa: E402

ops = HipOps(0)
w = sys.argv[1] if len(sys.argv) > 1 else "pl32768d32"
P = int(sys.argv[2]) if len(sys.argv) > 2 else 8
df = synth.WORKLOADS[w][0]()
_, csr = ingest.directed(df, False, "from", "to", "weight")
n = csr.n_rows
L = n // P
(spec,), _ = reorder_specs([SideSpec(csr, csr.rowscale, 0.8)], P)     # the solver's node order (dealt to P shards)
csr = spec.csr
g = ops.graph(csr, spec.rowscale)


def timed(fn, reps=5):
    fn()
    e0, e1 = ops.event(), ops.event()
    ops.record(e0)
    for _ in range(reps):
        fn()
    ops.record(e1)
    ops.event_synchronize(e1)
    return ops.elapsed_ms(e0, e1) / reps


xb = ops.matrix(n, L, blocked=True)
ops.fill_identity(xb, 0)
yb = ops.matrix(L, n, blocked=True)
a = timed(lambda: ops.spmm(g, xb, yb, transpose_out=True))
xr = ops.matrix(n, L)
ops.permute(xb, xr)
mb = n // P
pad = row_pad(mb)
size = P * L * (mb + pad)
yc = ops.matrix(1, size, ld=size)
b = timed(lambda: ops.spmm(g, xr, yc, n_cols=L, transpose_out=True, t_block=mb, t_pad=pad))
yr = ops.matrix(L, n)
c = timed(lambda: ops.spmm(g, xr, yr, n_cols=L, transpose_out=True))
print(f"{w} P={P} one rank's leg 1 ({n} x {L} block, row pitch {xr.ld}, chunk pad {pad}): blocked -> blocked {a:.3f} ms; "
      f"row-major -> chunks {b:.3f} ms; row-major -> pitched {c:.3f} ms", flush=True)

# the row pitch of the row-major block (HipOps.pitch_pad floats beyond a power of two): where does the gap to (a) go?
for padf in [int(v) for v in os.environ.get("PITCH_PADS", "0,32,64,96,160,288,544,1056").split(",")]:
    xp = ops.matrix(n, L, ld=L + padf)
    ops.permute(xb, xp)
    t = timed(lambda: ops.spmm(g, xp, yc, n_cols=L, transpose_out=True, t_block=mb, t_pad=pad))
    print(f"  row pitch {L + padf} floats ({4 * (L + padf)} B): row-major -> chunks {t:.3f} ms", flush=True)
    xp.free()
