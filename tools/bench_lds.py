#!/usr/bin/env python3
"""LDS-tiled leg vs the L2-gather legs on graphs with <= 8192 source rows."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simrank_amd import ingest, synth                     # noqa: E402
from simrank_amd.engine import HipOps                     # noqa: E402

ops = HipOps(0)


def timed(fn, reps=20):
    fn(); fn()
    a, b = ops.event(), ops.event()
    ops.record(a)
    for _ in range(reps):
        fn()
    ops.record(b)
    return ops.elapsed_ms(a, b) / reps


for name, df_csr in (("er8192", None), ("ml1m", None)):
    df = synth.WORKLOADS[name][0]()
    if name == "ml1m":
        _, _, _, _, csr, _ = ingest.bipartite(df, False, "user", "item", "weight")
    else:
        _, csr = ingest.directed(df, False, "from", "to", "weight")
    M, K = csr.n_rows, csr.n_cols
    g = ops.graph(csr)
    print(f"{name}: M={M} K={K} nnz={csr.nnz} lds_supported={ops.lds_supported(g)}", flush=True)
    xb, zt, ob = ops.b4_matrix(K, K), ops.b4_matrix(K, M), ops.b4_matrix(M, M)
    ops.b4_identity(xb, K)
    t1 = timed(lambda: ops.spmm_lds(g, xb, K, zt))
    t2 = timed(lambda: ops.spmm_lds(g, zt, M, ob, epilogue=dict(coef=0.8, previous=ob if M == K else None,
                                                                 eps=0.0, diag_col0=0)))
    x, tt, o = ops.matrix(K, K), ops.matrix(K, M), ops.matrix(M, M)
    ops.fill_identity(x, 0)
    g1 = timed(lambda: ops.spmm(g, x, tt, transpose_out=True))
    g2 = timed(lambda: ops.spmm(g, tt, o, epilogue=dict(coef=0.8, previous=o, eps=0.0, diag_col0=0,
                                                        symmetric=True)))
    gb = 4e-9 * csr.nnz
    print(f"  LDS legs    : leg1 {t1:.3f} ms ({gb * K / t1:.1f} TB/s gathered)  leg2 {t2:.3f} ms ({gb * M / t2:.1f} TB/s)")
    print(f"  gather legs : leg1 {g1:.3f} ms ({gb * K / g1:.1f} TB/s gathered)  leg2 {g2:.3f} ms (upper triangle + mirror)", flush=True)
