#!/usr/bin/env python3
"""Where a wave of the gather kernel spends its cycles: s_memtime stamps at the phase boundaries
(diagnostic build only: bash tools/build_variant.sh stamps -DSIMRANK_STAMPS).

    SIMRANK_LIB=$PWD/gpurun_variants/libsimrank_hip_stamps.so python tools/stamps.py [workload]
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simrank_amd import ingest, synth                              # noqa: E402
from tests.pydriver import LocalWorld, SideSpec, Solver        # noqa: E402
from simrank_amd.engine import HipOps                              # noqa: E402

NAMES = ["prologue (args, tile, row pointers, order)", "phase A0 (huge rows over 4 waves)",
         "phase A (long rows, 8 lane groups per row)", "phase B (one row per lane group)",
         "barrier before the tile store", "tile store + epilogue tail"]
ops = HipOps(0)
lib = ops.lib
lib.simrank_read_stamps.argtypes = [C.c_void_p, C.c_int32]
w = sys.argv[1] if len(sys.argv) > 1 else "pl32768"
df = synth.WORKLOADS[w][0]()
_, csr = ingest.directed(df, False, "from", "to", "weight")
s = Solver(lambda r: ops, LocalWorld(1), [SideSpec(csr, csr.rowscale, 0.8)], "sparse")
s.reset()
for _ in range(3):
    s.step(0.0)
side = s.sides[0][0]
buf = (C.c_ulonglong * 16)()


def report(tag):
    lib.simrank_read_stamps(buf, 1)
    tot = sum(buf[:6])
    print(f"{w} {tag}: wave-cycles by phase (share of the stamped total)")
    for i, n in enumerate(NAMES):
        print(f"   {n:48s} {buf[i]:16d}  {100.0 * buf[i] / max(1, tot):5.1f} %")


lib.simrank_read_stamps(buf, 1)
for _ in range(3):
    side.leg1(s.cur[0][0])
ops.synchronize()
report("leg 1 (transposed store)")
for _ in range(3):
    side.leg2(s.cur[0][0], s.nxt[0][0], 0.0)
ops.synchronize()
report("leg 2 (upper triangle, epilogue)")
