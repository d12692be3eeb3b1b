#!/usr/bin/env python3
"""Leg times under a relabelling of the nodes by the length of their rows of W.

The update is equivariant under a node permutation, so the solver may work in any order.
Ascending row length packs equal rows into a wave's passes (fewer masked gathers) and, for
leg 2's upper triangle, gives the long rows the short column ranges.

    python tools/order_probe.py [--workload pl32768,er8192]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simrank_amd import ingest, synth                              # noqa: E402
from tests.pydriver import LocalWorld, SideSpec, Solver        # noqa: E402
from simrank_amd.engine import HipOps                              # noqa: E402


def relabel(csr, perm, columns=True):
    """Node i of the new graph is node perm[i] of the old one (rows and, unless
    ``columns`` is False — a timing experiment, not a SimRank — columns)."""
    inv = np.empty_like(perm)
    inv[perm] = np.arange(perm.size)
    if not columns:
        inv = np.arange(perm.size)
    lens = np.diff(csr.rowptr)[perm]
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    col = np.empty(csr.nnz, dtype=np.int32)
    for i, old in enumerate(perm):
        seg = inv[csr.col[csr.rowptr[old]:csr.rowptr[old + 1]]]
        col[rowptr[i]:rowptr[i + 1]] = np.sort(seg)
    return ingest.CSR(csr.n_rows, csr.n_cols, rowptr, col, csr.rowscale[perm])


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
ap.add_argument("--workload", default="pl32768,er8192")
args = ap.parse_args()
ops = HipOps(0)
pass
for w in args.workload.split(","):
    df = synth.WORKLOADS[w][0]()
    _, csr = ingest.directed(df, False, "from", "to", "weight")
    lens = np.diff(csr.rowptr)
    for name, perm in (("natural", np.arange(csr.n_rows)),
                       ("ascending", np.argsort(lens, kind="stable")),
                       ("descending", np.argsort(-lens, kind="stable"))):
        for columns in (True, False):
            l1, l2 = legs(ops, relabel(csr, perm.astype(np.int64), columns))
            print(f"{w} {name:10s} columns relabelled={columns!s:5s}: leg1 {l1:7.3f} ms  leg2 {l2:7.3f} ms  "
                  f"sum {l1 + l2:7.3f} ms", flush=True)
