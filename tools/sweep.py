#!/usr/bin/env python3
"""Leg timings over the tuning knobs (panel width, XCD map, pitch padding).
usage: python tools/sweep.py [workload ...]   -> table on stdout"""
import itertools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simrank_amd import ingest, synth          # noqa: E402
from simrank_amd.driver import LocalWorld, SideSpec, Solver   # noqa: E402
from simrank_amd.engine import HipOps           # noqa: E402

ops = HipOps(0)
workloads = sys.argv[1:] or ["er8192", "pl32768"]
for w in workloads:
    df = synth.WORKLOADS[w][0]()
    _, csr = ingest.directed(df, False, "from", "to", "weight")
    print(f"# {w}: N={csr.n_rows} nnz={csr.nnz}", flush=True)
    for tile, _unused, panel in itertools.product([16, 32, 64], [0], [16, 32, 64]):
        ops.set_tuning(panel=panel, tile=tile)
        s = Solver(lambda r: ops, LocalWorld(1), [SideSpec(csr, csr.rowscale, 0.8)], "sparse")
        s.reset()
        for _ in range(2):
            s.step(0.0)
        s.enable_timing()
        for _ in range(4):
            s.step(0.0)
        t = s.leg_times()
        l1, l2 = t["leg1.0"][0], t["leg2.0"][0]
        gb = 4e-9 * csr.nnz * csr.n_rows
        print(f"{w} tile={tile:2d} panel={panel:3d}  leg1 {l1:8.3f} ms ({gb / l1:6.2f} TB/s gathered)"
              f"  leg2 {l2:8.3f} ms ({gb / l2:6.2f} TB/s)", flush=True)
        s.release()
        del s
