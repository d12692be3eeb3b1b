#!/usr/bin/env python3
"""Soak of whole fits on fp16-HELD matrices (fit(storage_precision="fp16"), csrc/half.hip): random graphs, every
class, six updates on both sides, against the float64 oracle with the bar that mode states (a few fp16
roundings).  `python3 tools/soak_fits_half.py [first_seed] [count]`."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simrank_amd.SimRank as SRA                      # noqa: E402
from oracle import simrank_oracle as O                 # noqa: E402
from simrank_amd import synth                          # noqa: E402
from tests.graphs import bipartite_random              # noqa: E402
from tests.pydriver import LocalWorld                  # noqa: E402  (the two-matrix classes on fp16-held matrices run through
                                                       # the tests' Python choreography: the C-level two-matrix plan is f32)

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
t0 = time.time()
worst_rel = worst_abs = 0.0


def check(got, want):
    global worst_rel, worst_abs
    assert np.array_equal(got, got.T) and np.all(np.diag(got) == 1.0)
    err = np.abs(got - want)
    big = want > 1e-4
    rel = float((err[big] / want[big]).max()) if big.any() else 0.0
    assert err.max() < 8e-4 and rel < 6e-3, (float(err.max()), rel)
    worst_rel, worst_abs = max(worst_rel, rel), max(worst_abs, float(err.max()))


for seed in range(first, first + count):
    rng = np.random.default_rng(5000 + seed)
    n = int(rng.integers(2, 900))
    kind = ["er", "pl", "bip"][seed % 3]
    weighted = bool(rng.integers(0, 2))
    it = dict(iterations=6, eps=1e-30, verbose=False, weighted=weighted)
    if kind == "bip":
        n2 = int(rng.integers(2, 400))
        df = bipartite_random(n, n2, float(rng.uniform(0.01, 0.3)), seed=seed)
        pp = bool(rng.integers(0, 2))
        extra = dict(strict_reference=False) if pp else {}
        s1, s2 = (SRA.BipartiteSimRankPP if pp else SRA.BipartiteSimRank)().fit(
            df, storage_precision="fp16", C1=0.7, C2=0.85, world=LocalWorld(1), **it, **extra)
        want = (O.fit_bipartite_pp if pp else O.fit_bipartite)(df, C1=0.7, C2=0.85, **it, **extra)
        check(s1.values, want["S1"])
        check(s2.values, want["S2"])
    else:
        df = (synth.er_directed(n, float(rng.uniform(0.005, 0.3)), seed) if kind == "er"
              else synth.powerlaw_directed(n, float(rng.uniform(1, 30)), seed))
        if len(df) == 0:
            continue
        cls = str(rng.choice(["SimRank", "SimRankPP", "AprioriSimRank"]))
        kw = dict(C=float(rng.uniform(0.5, 0.9)), **it)
        if cls == "AprioriSimRank":
            m = len(O.fit_simrank(df, iterations=0, verbose=False)["labels"])
            prior = rng.random((m, m))
            prior = (prior + prior.T) / 2
            got = SRA.AprioriSimRank().fit(df, prior, lbd=0.3, storage_precision="fp16", **kw)
            want = O.fit_simrank_pp(df, apriori=prior, lbd=0.3, **kw)
        else:
            got = getattr(SRA, cls)().fit(df, storage_precision="fp16", **kw)
            want = (O.fit_simrank if cls == "SimRank" else O.fit_simrank_pp)(df, **kw)
        assert list(got.index) == want["labels"]
        check(got.values, want["S"])
    if (seed - first) % 25 == 24:
        print(f"{seed - first + 1} cases, {time.time() - t0:.0f} s, worst rel (values > 1e-4) {worst_rel:.2e}, "
              f"worst abs {worst_abs:.2e}", flush=True)
print(f"soak_fits_half: seeds {first}..{first + count - 1} passed; worst relative error on values > 1e-4 "
      f"{worst_rel:.2e}, worst absolute error {worst_abs:.2e}")
