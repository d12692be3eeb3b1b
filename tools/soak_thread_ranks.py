#!/usr/bin/env python3
"""Soak of the sharded loops behind the C ABI on CONCURRENT ranks (simrank_comm_thread_group: one host thread per rank on the
code path of an RCCL rank, engine.ThreadRanks): random directed and bipartite graphs, every class, 2..8 ranks (uneven and empty
blocks included), both forms of leg 2 where the size allows, exchanges in 1..4 stages, priors (symmetric / not: the second
all-to-all), the fp16 wire, fp16-held matrices — every case BIT-EQUAL to the in-process group (the same ranks advanced one after
another by one thread, device copies), the same loop index on every rank, root's gather = the blocks put together.  An ordering
bug between the kernel stream and the exchange stream, a count reduced across ranks that are in different updates, a send matched
with the wrong receive: this is where they would show.   `python3 tools/soak_thread_ranks.py [first_seed] [count]`."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simrank_amd import ingest, synth                                              # noqa: E402
from simrank_amd.engine import HipOps, ShardBiPlans, ShardPlans, ThreadRanks       # noqa: E402
from tests.graphs import bipartite_random                                          # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
ops = HipOps(0)
t0 = time.time()
kinds = {}
for seed in range(first, first + count):
    rng = np.random.default_rng(41000 + seed)
    world = int(rng.integers(2, 9))
    stages = int(rng.integers(1, 5))
    iters, eps = (100, 1e-4) if seed % 2 else (int(rng.integers(1, 7)), 0.0)
    if seed % 5 == 4:
        # the two-matrix loop
        n1 = 32 * world * int(rng.integers(1, 4)) if seed % 2 else int(rng.integers(3, 300))
        n2 = 32 * world * int(rng.integers(1, 3)) if seed % 2 else int(rng.integers(3, 200))
        df = bipartite_random(n1, n2, float(rng.uniform(0.03, 0.3)), seed=seed)
        _, _, _, _, g12, g21 = ingest.bipartite(df, False, "user", "item", "weight")
        pp = bool(rng.integers(0, 2))
        rs = (ingest.spread(g12) * g12.rowscale, ingest.spread(g21) * g21.rowscale) if pp else (g12.rowscale, g21.rowscale)
        opts = dict(c1=float(rng.uniform(0.5, 0.9)), c2=float(rng.uniform(0.5, 0.9)), evidence=pp, stages=stages,
                    leg2_form=int(rng.integers(-1, 2)) if seed % 2 else 0)
        if seed % 3 == 0:
            p1, p2 = rng.random((g12.n_rows, g12.n_rows)), rng.random((g21.n_rows, g21.n_rows))
            if seed % 6:
                p1, p2 = (p1 + p1.T) / 2, (p2 + p2.T) / 2
            else:
                opts["leg2_form"] = 0
            opts.update(apriori1=p1, apriori2=p2, lbd1=0.3, lbd2=0.2)

        def rank(r, rops, comm):
            bp = ShardBiPlans(rops, g12, rs[0], rs[1], world=world, comm=comm, **opts)
            res = bp.run(iters, eps)
            out = res, bp.result(1, root=0, i_am_root=(r == 0)), bp.result(2, root=0, i_am_root=(r == 0))
            bp.free()
            return out
        tr = ThreadRanks(world)
        try:
            outs = tr.run(rank, timeout=300.0)
        finally:
            tr.close()
        bp = ShardBiPlans(ops, g12, rs[0], rs[1], world=world, **opts)
        ref = bp.run(iters, eps)
        r1, r2 = bp.result(1), bp.result(2)
        bp.free()
        assert all(o[0] == ref for o in outs), (seed, [o[0] for o in outs], ref)
        assert np.array_equal(outs[0][1], r1) and np.array_equal(outs[0][2], r2), seed
        kinds["bipartite"] = kinds.get("bipartite", 0) + 1
    else:
        if seed % 2:
            n = 64 * world * int(rng.integers(1, max(2, 640 // (64 * world))))      # sizes the half form and fp16-held shards accept
        else:
            n = int(rng.integers(2, 700))
        df = (synth.er_directed(n, float(rng.uniform(0.005, 0.2)), seed) if seed % 4 < 2
              else synth.powerlaw_directed(n, float(rng.uniform(1, 25)), seed))
        if len(df) == 0:
            continue
        nodes, csr = ingest.directed(df, False, "from", "to", "weight")
        pp = bool(rng.integers(0, 2)) or seed % 3 == 0
        scale = ingest.spread(csr) * csr.rowscale if pp else csr.rowscale
        kw = dict(coef=float(rng.uniform(0.5, 0.9)), evidence=pp, stages=stages)
        fits_half = csr.n_rows % (32 * world) == 0
        kind = "plain"
        if seed % 3 == 0:
            prior = rng.random((csr.n_rows, csr.n_rows)).astype(np.float32)
            if seed % 6:
                prior = ((prior + prior.T) / 2).astype(np.float32)
                kw["leg2_form"] = int(rng.integers(0, 2)) if fits_half else 0
                kind = "prior"
            else:
                kw["leg2_form"] = 0
                kind = "asymmetric prior"
            kw.update(apriori=prior, lbd=0.3)
        elif seed % 7 == 1 and csr.n_rows % (64 * world) == 0:
            kw.update(storage="fp16", leg2_form=0)
            kind = "fp16-held"
        else:
            kw["leg2_form"] = int(rng.integers(0, 2)) if fits_half else 0
            if seed % 4 == 3:
                kw["wire_fp16"] = True
                kind = "fp16 wire"
        if kw.get("leg2_form") == 1:
            kind += ", half form"

        def rank(r, rops, comm):
            sp = ShardPlans(rops, csr, rowscale=scale, world=world, comm=comm, **kw)
            res = sp.run(iters, eps)
            full = sp.result(root=0, i_am_root=(r == 0))
            blk, ids = sp.block(0)
            sp.free()
            return res, full, blk, ids
        tr = ThreadRanks(world)
        try:
            outs = tr.run(rank, timeout=300.0)
        finally:
            tr.close()
        sp = ShardPlans(ops, csr, rowscale=scale, world=world, **kw)
        ref = sp.run(iters, eps)
        rfull = sp.result()
        sp.free()
        assert all(o[0] == ref for o in outs), (seed, [o[0] for o in outs], ref)
        assert np.array_equal(outs[0][1], rfull), (seed, kind)
        put = np.full_like(rfull, np.nan)
        for o in outs:
            put[:, o[3]] = o[2]
        assert np.array_equal(put, rfull), (seed, kind)
        kinds[kind] = kinds.get(kind, 0) + 1
    if (seed - first) % 25 == 24:
        print(f"{seed - first + 1} cases, {time.time() - t0:.0f} s", flush=True)
print(f"soak_thread_ranks: seeds {first}..{first + count - 1}: every case bit-equal to the in-process group; {kinds}")
