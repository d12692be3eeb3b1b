#!/usr/bin/env python3
"""Error of fits with fp16-held matrices against the float64 oracle (profiles/r03_half_errors.log)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simrank_amd.SimRank as SRA
from oracle import simrank_oracle as O
from simrank_amd import synth

def stats(a, b):
    err = np.abs(a - b)
    pos = b > 0
    rel = err[pos] / b[pos]
    big = b > 1e-3
    relb = err[big] / b[big]
    return (f"max_abs {err.max():.3e} max_rel {rel.max():.3e} p99 {np.quantile(rel, .99):.3e} median {np.median(rel):.3e} | "
            f"b>1e-3: n {big.sum()} max {relb.max():.3e} median {np.median(relb):.3e}")

for n, cls in ((2048, "SimRank"), (2048, "SimRankPP"), (4096, "SimRankPP")):
    df = synth.powerlaw_directed(n, 24, seed=12)
    for fixed in (False, True):           # to convergence (eps 1e-4), and exactly 10 updates on both sides
        okw = dict(iterations=10, eps=1e-30) if fixed else {}
        want = (O.fit_simrank if cls == "SimRank" else O.fit_simrank_pp)(df, verbose=False, **okw)
        for kw in ({}, dict(storage_precision="fp16")):
            est = getattr(SRA, cls)()
            got = est.fit(df, verbose=False, **okw, **kw)
            print(n, cls, "10 updates" if fixed else "to eps", kw, "k", est.converged_at, "oracle k", want["k"],
                  stats(got.values, want["S"]), flush=True)
