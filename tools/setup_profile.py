#!/usr/bin/env python3
"""cProfile of the solver set-up of a SimRank++ fit (what bench.py reports as setup_s_graph_and_evidence).
    python3 tools/setup_profile.py [--workload pl65536]"""
import argparse, cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simrank_amd import ingest, synth                              # noqa: E402
from tests.pydriver import LocalWorld, SideSpec, Solver        # noqa: E402
from simrank_amd.engine import HipOps                              # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="pl65536")
args = ap.parse_args()
ops = HipOps(0)
df = synth.WORKLOADS[args.workload][0]()
_, csr = ingest.directed(df, False, "from", "to", "weight")
spec = SideSpec(csr, ingest.spread(csr) * csr.rowscale, 0.8, evidence_from=csr)
for rep in range(2):
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    s = Solver(lambda r: ops, LocalWorld(1), [spec], "auto")
    ops.synchronize()
    pr.disable()
    print(f"== set-up {rep}: {time.perf_counter() - t0:.3f} s")
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
    t0 = time.perf_counter()
    s.reset(); ops.synchronize()
    print(f"reset {time.perf_counter() - t0:.3f} s")
    s.release(); del s
