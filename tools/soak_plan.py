#!/usr/bin/env python3
"""Soak of the C-level loop (simrank_plan_*: create / run / result) on a real GPU: random directed graphs,
SimRank and SimRank++ (a prior every third case, every sixth one NOT symmetric: asymmetric iterates), against the float64 oracle at 1e-5 with the
reference's convergence iteration.  `python3 tools/soak_plan.py [first_seed] [count]`."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import simrank_oracle as O                 # noqa: E402
from simrank_amd import ingest, synth                  # noqa: E402
from simrank_amd.engine import HipOps, Plan            # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
ops = HipOps(0)
# SOAK_KNOBS="fuse_sym=1,fuse_steps=1": library knobs for the whole run (e.g. the one-launch leg 2 on every graph that has a set)
if os.environ.get("SOAK_KNOBS"):
    from simrank_amd.engine import HipOps as _H
    _H(0).set_tuning(**{k: int(v) for k, v in (kv.split("=") for kv in os.environ["SOAK_KNOBS"].split(","))})
    print("knobs:", os.environ["SOAK_KNOBS"], flush=True)
t0 = time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(9000 + seed)
    n = int(rng.integers(2, 700))
    df = (synth.er_directed(n, float(rng.uniform(0.005, 0.2)), seed) if seed % 2
          else synth.powerlaw_directed(n, float(rng.uniform(1, 25)), seed))
    if len(df) == 0:
        continue
    pp = bool(rng.integers(0, 2)) or seed % 3 == 0           # (the prior classes of the reference are SimRank++ ones)
    coef = float(rng.uniform(0.5, 0.9))
    nodes, csr = ingest.directed(df, False, "from", "to", "weight")
    scale = ingest.spread(csr) * csr.rowscale if pp else csr.rowscale
    kw = {}
    prior = None
    if seed % 3 == 0:
        prior = rng.random((csr.n_rows, csr.n_rows)).astype(np.float32)
        if seed % 6:
            prior = ((prior + prior.T) / 2).astype(np.float32)
        kw = dict(apriori=prior.astype(np.float64), lbd=0.3)
    want = (O.fit_simrank_pp if pp else O.fit_simrank)(df, C=coef, verbose=False, **kw)
    plan = Plan(ops, csr, rowscale=scale, coef=coef, evidence=pp, apriori=prior, lbd=0.3 if prior is not None else 0.0)
    done, conv = plan.run(100, 1e-4)
    got = plan.result()
    plan.free()
    assert (conv if conv is not None else -1) == (want["k"] if want["k"] is not None else -1), (seed, conv, want["k"])
    np.testing.assert_allclose(got, want["S"], rtol=1e-5, atol=1e-30, err_msg=str(seed))
    if (seed - first) % 25 == 24:
        print(f"{seed - first + 1} cases, {time.time() - t0:.0f} s", flush=True)
print(f"soak_plan: seeds {first}..{first + count - 1} passed (directed)")
# the bipartite loop (simrank_biplan_*): plain and SimRank++ in its corrected form, n1 != n2
from simrank_amd.engine import BiPlan                  # noqa: E402
from tests.graphs import bipartite_random              # noqa: E402
from tests.test_gpu_parity import _csr_and_scales      # noqa: E402
t0 = time.time()
for seed in range(first, first + count // 2):
    rng = np.random.default_rng(19000 + seed)
    n1, n2 = int(rng.integers(1, 500)), int(rng.integers(1, 300))
    df = bipartite_random(n1, n2, float(rng.uniform(0.005, 0.3)), seed=seed)
    pp = bool(rng.integers(0, 2))
    c1, c2 = float(rng.uniform(0.5, 0.9)), float(rng.uniform(0.5, 0.9))
    pri = {}
    if len(df) == 0:
        continue
    n1, n2 = df["user"].nunique(), df["item"].nunique()     # (nodes without an edge do not exist for the reference)
    if pp and seed % 3 == 0:                     # BipartitleAprioriSimRank; every other one with a prior that is not symmetric
        a1, a2 = rng.random((n1, n1)).astype(np.float32), rng.random((n2, n2)).astype(np.float32)
        a1 = ((a1 + a1.T) / 2).astype(np.float32)
        if seed % 2:
            a2 = ((a2 + a2.T) / 2).astype(np.float32)
        pri = dict(apriori1=a1.astype(np.float64), apriori2=a2.astype(np.float64), lbd1=0.25, lbd2=0.4)
    if pp:
        want = O.fit_bipartite_pp(df, verbose=False, strict_reference=False, C1=c1, C2=c2, **pri)
        m1, m2 = want["W1"], want["W2"]
    else:
        want = O.fit_bipartite(df, verbose=False, C1=c1, C2=c2)
        m1, m2 = want["G12"], want["G21"]
    csr, _, _ = _csr_and_scales(want["G12"], want["G21"])
    rs1, rs2 = np.asarray(m1).max(axis=1), np.asarray(m2).max(axis=1)     # one value per row
    plan = BiPlan(ops, csr, rs1, rs2, c1=c1, c2=c2, evidence=pp, **pri)
    done, conv = plan.run(100, 1e-4)
    s1, s2 = plan.result()
    plan.free()
    assert (conv if conv is not None else -1) == (want["k"] if want["k"] is not None else -1), (seed, conv, want["k"])
    np.testing.assert_allclose(s1, want["S1"], rtol=1e-5, atol=1e-30, err_msg=str(seed))
    np.testing.assert_allclose(s2, want["S2"], rtol=1e-5, atol=1e-30, err_msg=str(seed))
    if (seed - first) % 25 == 24:
        print(f"{seed - first + 1} bipartite cases, {time.time() - t0:.0f} s", flush=True)
print(f"soak_plan: {count // 2} bipartite graphs passed")
