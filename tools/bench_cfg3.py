#!/usr/bin/env python3
"""cfg3 (MovieLens-1M-shaped bipartite SimRank++): per-iteration time by mode and by the knobs of
the block-dense part.   python tools/bench_cfg3.py [--modes sparse,dense] [--knobs k=v,k=v;k=v]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simrank_amd import ingest, synth                      # noqa: E402
from tests.pydriver import LocalWorld, SideSpec, Solver  # noqa: E402
from simrank_amd.engine import HipOps                      # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--modes", default="sparse")
ap.add_argument("--knobs", default="dense_min=0;dense_min=4,dense_cols=128")
args = ap.parse_args()
ops = HipOps(0)
df = synth.WORKLOADS["ml1m"][0]()
s1, s2, l1, l2, g12, g21 = ingest.bipartite(df, False, "user", "item", "weight")
print(f"cfg3: n1={g12.n_rows} n2={g12.n_cols} nnz={g12.nnz} density={g12.density:.4f}", flush=True)
for mode in args.modes.split(","):
    for knobs in args.knobs.split(";"):
        kv = {k: int(v) for k, v in (x.split("=") for x in knobs.split(",") if x)}
        ops.set_tuning(**kv)
        specs = [SideSpec(g12, g12.rowscale, 0.8, evidence_from=g12),
                 SideSpec(g21, g21.rowscale, 0.8, evidence_from=g21)]
        sol = Solver(lambda r: ops, LocalWorld(1), specs, mode)
        stats = [ops.dense_stats(next(iter(s.values())).graph) for s in sol.sides]
        sol.reset()
        sol.step(0.0)
        sol.enable_timing()
        ops.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            sol.step(0.0)
        ops.synchronize()
        dt = (time.perf_counter() - t0) / 5
        legs = {k: round(v[0], 3) for k, v in sol.leg_times().items()}
        print(f"cfg3 mode={mode} {knobs}: {dt * 1e3:.2f} ms/iteration, legs {legs}, dense sets {stats}", flush=True)
        sol.release()
        del sol
        ops.set_tuning(dense_min=4, dense_cols=128, dense_sym=-1)
