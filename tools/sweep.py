#!/usr/bin/env python3
"""Leg timings over the tuning knobs of the gather kernel (include/simrank_hip.h,
simrank_set_tuning).  Used for the profiles/sweep_r01_*.log tables.

    python tools/sweep.py [--workload pl32768] [--panel 32,64] [--tile 16,32,64]
                          [--xcd 1,0] [--huge 512] [--triangle 1]
"""
import argparse
import itertools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simrank_amd import ingest, synth                              # noqa: E402
from tests.pydriver import LocalWorld, SideSpec, Solver        # noqa: E402
from simrank_amd.engine import HipOps                              # noqa: E402


def ints(text):
    return [int(v) for v in text.split(",")]


ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="er8192,pl32768")
ap.add_argument("--panel", type=ints, default=[0])
ap.add_argument("--tile", type=ints, default=[0])
ap.add_argument("--xcd", type=ints, default=[1])
ap.add_argument("--huge", type=ints, default=[512])
ap.add_argument("--triangle", type=ints, default=[1])
ap.add_argument("--mode", default="sparse")
ap.add_argument("--nt", type=ints, default=[1])
ap.add_argument("--balance", type=ints, default=[4])
ap.add_argument("--dense-min", type=ints, default=[4])
ap.add_argument("--dense-cols", type=ints, default=[128])
args = ap.parse_args()

ops = HipOps(0)
for w in args.workload.split(","):
    if w.startswith("er:"):                      # er:N:p  ad-hoc Erdos-Renyi graph
        _, n_, p_ = w.split(":")
        df = synth.er_directed(int(n_), float(p_), 1)
    else:
        df = synth.WORKLOADS[w][0]()
    _, csr = ingest.directed(df, False, "from", "to", "weight")
    print(f"# {w}: N={csr.n_rows} nnz={csr.nnz}", flush=True)
    for panel, tile, xcd, huge, tri, dmin, dcols, nt_, bal in itertools.product(
            args.panel, args.tile, args.xcd, args.huge, args.triangle, args.dense_min, args.dense_cols, args.nt,
            args.balance):
        ops.set_tuning(panel=panel, tile=tile, xcd_map=xcd, huge=huge, triangle=tri, dense_min=dmin,
                       dense_cols=dcols, stream_nt=nt_, balance=bal)
        s = Solver(lambda r: ops, LocalWorld(1), [SideSpec(csr, csr.rowscale, 0.8)], args.mode)
        s.reset()
        for _ in range(2):
            s.step(0.0)
        s.enable_timing()
        for _ in range(4):
            s.step(0.0)
        t = s.leg_times()
        l1, l2 = t["leg1.0"][0], t["leg2.0"][0]
        gb = 4e-9 * csr.nnz * csr.n_rows
        nt, dk, cov = ops.dense_stats(next(iter(s.sides[0].values())).graph)
        print(f"{w} mode={s.mode} panel={panel:3d} tile={tile:2d} xcd={xcd} huge={huge} triangle={tri} "
              f"dense_min={dmin} dense_cols={dcols} nt={nt_} balance={bal} [tiles {nt} cols {dk} covered {cov / max(1, csr.nnz):.3f}]  "
              f"leg1 {l1:8.3f} ms ({gb / l1:6.2f} TB/s gathered)  leg2 {l2:8.3f} ms  "
              f"total {l1 + l2:8.3f} ms", flush=True)
        s.release()
        del s
ops.set_tuning(panel=0, tile=0, xcd_map=1, huge=512, triangle=1, dense_min=4, dense_cols=128, stream_nt=1, balance=2)
