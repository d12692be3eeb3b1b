#!/usr/bin/env python3
"""Soak of the sharded loop behind the C ABI (simrank_shardplan_*, in-process groups of 1..8 virtual ranks) on a real
GPU: random directed graphs, SimRank / SimRank++ (a prior every third case, every sixth one NOT symmetric: a second all-to-all per update), random rank counts (uneven and
empty blocks included), both forms of leg 2 where the size allows, exchange 1 in 1..4 stages — against the float64
oracle at 1e-5 with the reference's convergence iteration; every fourth case on the fp16 wire (looser bound, no
convergence index).  `python3 tools/soak_shardplan.py [first_seed] [count]`."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import simrank_oracle as O                 # noqa: E402
from simrank_amd import ingest, synth                  # noqa: E402
from simrank_amd.engine import HipOps, ShardPlans      # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
ops = HipOps(0)
t0 = time.time()
halves = wires = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(29000 + seed)
    world = int(rng.integers(1, 9))
    if seed % 2:
        n = 32 * world * int(rng.integers(1, max(2, 640 // (32 * world))))      # sizes the half form accepts
    else:
        n = int(rng.integers(2, 700))
    df = (synth.er_directed(n, float(rng.uniform(0.005, 0.2)), seed) if seed % 4 < 2
          else synth.powerlaw_directed(n, float(rng.uniform(1, 25)), seed))
    if len(df) == 0:
        continue
    pp = bool(rng.integers(0, 2)) or seed % 3 == 0
    coef = float(rng.uniform(0.5, 0.9))
    nodes, csr = ingest.directed(df, False, "from", "to", "weight")
    scale = ingest.spread(csr) * csr.rowscale if pp else csr.rowscale
    kw, prior = {}, None
    if seed % 3 == 0:
        prior = rng.random((csr.n_rows, csr.n_rows)).astype(np.float32)
        if seed % 6:
            prior = ((prior + prior.T) / 2).astype(np.float32)
        kw = dict(apriori=prior.astype(np.float64), lbd=0.3)
    want = (O.fit_simrank_pp if pp else O.fit_simrank)(df, C=coef, verbose=False, **kw)
    asym = prior is not None and seed % 6 == 0
    form = int(rng.integers(0, 2)) if csr.n_rows % (32 * world) == 0 and not asym else 0
    wire = seed % 4 == 3 and not asym
    if seed % 5 == 1 and csr.n_rows % (64 * world) == 0 and prior is None:
        # fp16-held matrices on every rank: the bars of tests/test_gpu_half.py (may end later than the reference, never
        # earlier by more than one update; within eps C / (1 - C) + the arithmetic error of the mode)
        sp = ShardPlans(ops, csr, rowscale=scale, world=world, coef=coef, evidence=pp, stages=int(rng.integers(1, 5)), storage="fp16")
        done, conv = sp.run(100, 1e-4)
        got = sp.result()
        sp.free()
        assert conv is not None and conv >= want["k"] - 1, (seed, conv, want["k"])
        assert np.abs(got - want["S"]).max() < 1e-4 * coef / (1 - coef) + 6e-4, (seed, np.abs(got - want["S"]).max())
        halfs = globals().get("halfs", 0) + 1
        continue
    sp = ShardPlans(ops, csr, rowscale=scale, world=world, coef=coef, evidence=pp, apriori=prior,
                    lbd=0.3 if prior is not None else 0.0, leg2_form=form, stages=int(rng.integers(1, 5)), wire_fp16=wire)
    done, conv = sp.run(100, 1e-4)
    got = sp.result()
    sp.free()
    halves += form
    wires += wire
    if wire:
        # (one fp16 rounding of the transposed product per update, and the loop may pass its eps = 1e-4 test one index
        # earlier or later than the reference: bounded in absolute terms, and relative to elements that are not small)
        big = want["S"] > 0.05
        rel = (np.abs(got - want["S"])[big] / want["S"][big]).max() if big.any() else 0.0
        assert rel < 1e-2 and np.abs(got - want["S"]).max() < 1e-3, (seed, rel, np.abs(got - want["S"]).max())
    else:
        assert (conv if conv is not None else -1) == (want["k"] if want["k"] is not None else -1), (seed, conv, want["k"])
        np.testing.assert_allclose(got, want["S"], rtol=1e-5, atol=1e-30, err_msg=str(seed))
    if (seed - first) % 25 == 24:
        print(f"{seed - first + 1} cases, {time.time() - t0:.0f} s", flush=True)
print(f"soak_shardplan: seeds {first}..{first + count - 1} passed ({halves} in the half form, {wires} on the fp16 wire, "
      f"{globals().get('halfs', 0)} on fp16-held matrices)")
