#!/usr/bin/env python3
"""cProfile of a whole fit through the class surface (second call: allocator and host frames warm): where the host side of
`SimRank().fit(edges)` spends its time next to the C calls.   python3 tools/fit_profile.py [workload = pl32768d32]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simrank_amd.SimRank as SRA          # noqa: E402
from simrank_amd import synth              # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "pl32768d32"
df = synth.WORKLOADS[wl][0]()
bip = synth.WORKLOADS[wl][1] == "bipartite"       # (config 3: BipartiteSimRankPP, as bench.py's fit_wall)


def fit():
    return SRA.BipartiteSimRankPP().fit(df, verbose=False, strict_reference=False) if bip else SRA.SimRank().fit(df, verbose=False)


for rep in range(2):
    t0 = time.perf_counter()
    res = fit()
    print(f"call {rep}: {time.perf_counter() - t0:.3f} s", flush=True)
    del res
pr = cProfile.Profile()
pr.enable()
res = fit()
pr.disable()
pstats.Stats(pr).sort_stats("cumtime").print_stats(28)
