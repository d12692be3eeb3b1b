#!/usr/bin/env python3
"""What does the gather kernel reach when every operand line it asks for is an L2 hit?

Rectangular random patterns: M output rows x K source rows, `deg` entries per row, operand
K x L.  With 32-float panels a panel's slice of the operand is K x 128 B: 4 MiB (= one XCD's L2)
at K = 32768, 1 MiB at K = 8192.  The rate at small K is the ceiling a K-blocked leg could reach
at K = 32768 (round 2).

    python tools/l2_probe.py [--M 32768] [--L 32768] [--deg 24] [--K 2048,4096,8192,16384,32768]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simrank_amd.engine import HipOps          # noqa: E402
from simrank_amd.ingest import CSR             # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--M", type=int, default=32768)
ap.add_argument("--L", type=int, default=32768)
ap.add_argument("--deg", type=str, default="24")
ap.add_argument("--K", type=str, default="2048,4096,8192,16384,32768")
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--ld", type=int, default=0, help="leading dimension of the operand (0 = pitched L): a narrow "
                                                  "operand (L = 256: one panel per XCD) with ld = 256 keeps "
                                                  "a panel slice inside 32 MB of addresses, with ld = 32800 "
                                                  "it is spread over 4 GB like the real matrix")
ap.add_argument("--set", default="")
args = ap.parse_args()

ops = HipOps(0)
ops.set_tuning(dense_min=0)
if args.set:
    ops.set_tuning(**{k: int(v) for k, v in (kv.split("=") for kv in args.set.split(","))})
rng = np.random.default_rng(1)
M, L = args.M, args.L
for deg in [int(v) for v in args.deg.split(",")]:
    for K in [int(v) for v in args.K.split(",")]:
        # one column per stratum of K/deg source rows: distinct and ascending inside a row
        edges = (np.arange(deg + 1) * K) // deg
        col = (edges[:-1] + (rng.random((M, deg)) * np.diff(edges)).astype(np.int64)).astype(np.int32).ravel()
        rowptr = (np.arange(M + 1, dtype=np.int64) * deg).astype(np.int32)
        csr = CSR(M, K, rowptr, col, np.full(M, 1.0 / deg))
        g = ops.graph(csr)
        X = ops.matrix(K, L, ld=args.ld or None)
        Yt = ops.matrix(L, M)
        Y = ops.matrix(M, L)
        ops.fill_identity(X, 0)
        res = []
        for trans, out in ((True, Yt), (False, Y)):
            for _ in range(2):
                ops.spmm(g, X, out, transpose_out=trans)
            e0, e1 = ops.event(), ops.event()
            ops.record(e0)
            for _ in range(args.reps):
                ops.spmm(g, X, out, transpose_out=trans)
            ops.record(e1)
            ops.synchronize()
            ms = ops.elapsed_ms(e0, e1) / args.reps
            gb = 4e-9 * M * deg * L
            res.append(f"{'trans' if trans else 'plain'} {ms:7.3f} ms {gb / ms:6.2f} TB/s gathered "
                       f"({gb / ms / 256 * 1000:5.1f} GB/s per CU)")
        print(f"M={M} K={K:6d} deg={deg:3d} L={L} ld={X.ld}: slice {K * 128 / 2**20:4.2f} MiB | " + " | ".join(res),
              flush=True)
        for m in (X, Yt, Y):
            m.free()
        g.free()
ops.set_tuning(dense_min=4)
