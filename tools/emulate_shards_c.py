#!/usr/bin/env python3
"""The sharded loop behind the C ABI (simrank_shardplan_*) on ONE GPU: an in-process group of P virtual ranks advances in
lockstep (exchanges = device copies), so (time of one collective step) / P is what each of P real GPUs would spend in
kernels + the copies that stand in for the links.  Prints it next to the Python driver's sharded path on the same graph
and checks that both paths produce the same bits after three updates (sampled columns)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simrank_amd import ingest, synth                           # noqa: E402
from tests.pydriver import LocalWorld, SideSpec, Solver     # noqa: E402
from simrank_amd.engine import HipOps, ShardPlans               # noqa: E402

ops = HipOps(0)
w = sys.argv[1] if len(sys.argv) > 1 else "pl32768d32"
df = synth.WORKLOADS[w][0]()
_, csr = ingest.directed(df, False, "from", "to", "weight")
n = csr.n_rows
pp = os.environ.get("PP") == "1"                 # SimRank++ (evidence, spread weights): config 5's class
scale = ingest.spread(csr) * csr.rowscale if pp else csr.rowscale
if os.environ.get("STORAGE") == "fp16":
    # fp16-held matrices on every rank (options.storage_fp16; the Python driver has no such sharded mode): timing only,
    # next to the f32 plan of the same form
    for P in [int(v) for v in os.environ.get("PS", "8").split(",")]:
        for storage in ("f32", "fp16"):
            sp = ShardPlans(ops, csr, rowscale=scale, world=P, evidence=pp, leg2_form=0, stages=1, storage=storage)
            sp.step(0.0)
            ops.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                sp.step(0.0, exact_count=False)
            ops.synchronize()
            ms = (time.perf_counter() - t0) / 3 * 1e3
            sp.free()
            per = n * (n // P) * (2 if storage == "fp16" else 4) * (P - 1) / P / 2**20
            print(f"{w}{' SimRank++' if pp else ''} P={P} {storage} matrices, leg 2 in its full form: C loop {ms / P:.3f} ms per rank "
                  f"and update ({ms:.2f} ms for the {P} virtual ranks, device copies included); exchange payload per rank "
                  f"{per:.0f} MiB", flush=True)
    sys.exit(0)
STAGES = int(os.environ.get("STAGES", "1"))       # exchange 1 (and, in the half form, leg 2 + exchange 2) in that many stages
for P in [int(v) for v in os.environ.get("PS", "8,4").split(",")]:
    for form in (0, 1):
        sp = ShardPlans(ops, csr, world=P, leg2_form=form, stages=STAGES)
        sp.step(0.0)
        ops.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            sp.step(0.0, exact_count=False)
        ops.synchronize()
        c_ms = (time.perf_counter() - t0) / 3 * 1e3
        blk, ids = sp.block(P - 1)
        sp.free()
        s = Solver(lambda r: ops, LocalWorld(P, symmetric_shards=bool(form)), [SideSpec(csr, csr.rowscale, 0.8)], "sparse")
        s.reset()
        s.step(0.0)
        ops.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            s.step(0.0)
        ops.synchronize()
        py_ms = (time.perf_counter() - t0) / 3 * 1e3
        # the last rank's block of the Python solver, rows into the caller's order
        r = P - 1
        rows = np.asarray(s.inv[0])[ids[:3]]                     # positions of three of the block's own nodes
        got = ops.download_rows(s.cur[0][r], [int(v) for v in rows])          # (solver-order rows) x (block columns)
        same = np.array_equal(got.astype(np.float64), blk[ids[:3]][:, :])
        s.release()
        del s
        print(f"{w} P={P} stages={STAGES} leg 2 in its {'half' if form else 'full'} form: C loop {c_ms / P:.3f} ms per rank and update "
              f"({c_ms:.2f} ms for the {P} virtual ranks, device copies included), Python driver {py_ms / P:.3f}; "
              f"four updates, sampled rows of the last rank's block bit-equal: {same}", flush=True)
