#!/usr/bin/env python3
"""Support-restricted SimRank++: leg times of the SimRank++ update with the
restricted instantiation of leg 2 forced off / on, and the live-segment fraction that decides.

    python tools/bench_restrict.py [er8192,pl32768,er:4096:0.002]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tests.pydriver as drv                                    # noqa: E402
from simrank_amd import ingest, synth                              # noqa: E402
from tests.pydriver import LocalWorld, SideSpec, Solver        # noqa: E402
from simrank_amd.engine import HipOps                              # noqa: E402

ops = HipOps(0)
for w in (sys.argv[1] if len(sys.argv) > 1 else "er8192,pl32768").split(","):
    if w.startswith("er:"):
        _, n_, p_ = w.split(":")
        df = synth.er_directed(int(n_), float(p_), 1)
    else:
        df = synth.WORKLOADS[w][0]()
    _, csr = ingest.directed(df, False, "from", "to", "weight")
    scale = ingest.spread(csr) * csr.rowscale
    for name, below in (("off", 0.0), ("on", 2.0)):
        drv.RESTRICT_BELOW = below
        s = Solver(lambda r: ops, LocalWorld(1), [SideSpec(csr, scale, 0.8, evidence_from=csr)], "sparse")
        side = next(iter(s.sides[0].values()))
        s.reset()
        for _ in range(4):                       # (S fills in over the first iterations)
            s.step(0.0)
        s.enable_timing()
        for _ in range(8):
            s.step(0.0)
        t = s.leg_times()
        print(f"{w}: N={csr.n_rows} nnz={csr.nnz} live 32-column segments of E {side.ev_live:.3f} | restricted leg 2 "
              f"{name:3s}: leg1 {t['leg1.0'][0]:7.3f} ms  leg2 {t['leg2.0'][0]:7.3f} ms", flush=True)
        s.release()
        del s
drv.RESTRICT_BELOW = 0.5
