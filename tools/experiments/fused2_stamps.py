#!/usr/bin/env python3
"""Timeline of the items of the persistent leg 1 (diagnostic build only:
bash tools/build_variant.sh f2st -DSIMRANK_F2_STAMPS).

    SIMRANK_LIB=$PWD/gpurun_variants/libsimrank_hip_f2st.so python tools/fused2_stamps.py [workload] [--set k=v,...]

Per item (wave 0 of its workgroup): top of the loop, prefetches issued, matrix-core phase + sum done, gather phase
done, barrier passed, end (store issued, hand-over barrier passed); XCC, panel, block, piece, quads, rounds.
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simrank_amd import ingest, synth                              # noqa: E402
from tests.pydriver import LocalWorld, SideSpec, Solver        # noqa: E402
from simrank_amd.engine import HipOps                              # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("workload", nargs="?", default="pl32768d32")
ap.add_argument("--set", default="fuse=2")
ap.add_argument("--out", default="")
args = ap.parse_args()
ops = HipOps(0)
ops.set_tuning(**{k: int(v) for k, v in (kv.split("=") for kv in args.set.split(","))})
lib = ops.lib
lib.simrank_read_fused2_stamps.argtypes = [C.c_void_p, C.c_int64, C.c_int32]
df = synth.WORKLOADS[args.workload][0]()
_, csr = ingest.directed(df, False, "from", "to", "weight")
s = Solver(lambda r: ops, LocalWorld(1), [SideSpec(csr, csr.rowscale, 0.8)], "sparse")
s.reset()
for _ in range(2):
    s.step(0.0)
ops.synchronize()
lib.simrank_read_fused2_stamps(None, 0, 1)
side = s.sides[0][0]
side.leg1(s.cur[0][0])
ops.synchronize()
cap = 1 << 17
buf = np.zeros(cap * 8, dtype=np.uint64)
n = lib.simrank_read_fused2_stamps(buf.ctypes.data_as(C.c_void_p), cap, 1)
st = buf.reshape(-1, 8)[:n]
if args.out:
    np.savez_compressed(args.out, st=st)
t = st[:, :6].astype(np.int64)
xcd = (st[:, 6] >> np.uint64(56)).astype(np.int64)
panel = ((st[:, 6] >> np.uint64(32)) & np.uint64(0xFFFFFF)).astype(np.int64)
blk = ((st[:, 6] >> np.uint64(8)) & np.uint64(0xFFFFFF)).astype(np.int64)
piece = (st[:, 6] & np.uint64(0xFF)).astype(np.int64)
nq = (st[:, 7] >> np.uint64(32)).astype(np.int64)
nr = ((st[:, 7] >> np.uint64(8)) & np.uint64(0xFFFFFF)).astype(np.int64)
npc = (st[:, 7] & np.uint64(0xFF)).astype(np.int64)
dur = t[:, 5] - t[:, 0]
tot = dur.sum()
names = ["prefetch issue", "matrix cores + sum", "gather", "barrier wait", "publish / store + hand-over"]
print(f"# {args.workload} {args.set}: {n} items stamped; mean item {dur.mean() / 1e3:.1f} kcycles, p50 {np.median(dur) / 1e3:.1f}, "
      f"p90 {np.percentile(dur, 90) / 1e3:.1f}, max {dur.max() / 1e3:.1f}")
for i, nm in enumerate(names):
    print(f"   {nm:30s} {100.0 * (t[:, i + 1] - t[:, i]).sum() / tot:5.1f} %")
for label, m in (("items with a matrix-core phase", nq > 0), ("items without", nq == 0), ("pieces of split blocks", npc > 1)):
    if m.any():
        d = dur[m]
        print(f"{label}: {m.sum()} ({100.0 * d.sum() / tot:.1f} % of the time), mean {d.mean() / 1e3:.1f} kcycles; "
              + ", ".join(f"{nm} {(t[m, i + 1] - t[m, i]).mean() / 1e3:.1f}" for i, nm in enumerate(names)))
m = nq > 0
if m.any():
    steps = 4 * ((nq[m] + 3) // 4)          # steps of wave 0
    print(f"matrix-core phase: {((t[m, 2] - t[m, 1]).sum() / steps.sum()):.0f} cycles per step of wave 0")
m = (nq == 0) & (nr > 0)
if m.any():
    print(f"gather phase: {((t[m, 3] - t[m, 2]).sum() / nr[m].sum()):.0f} cycles per round of wave 0 (items without matrix-core phase)")
for x in sorted(set(xcd.tolist()))[:2]:
    mm = xcd == x
    lo, hi = np.percentile(t[mm, 0], 20), np.percentile(t[mm, 5], 80)
    ts = np.linspace(lo, hi, 200)
    live = [len(set(panel[mm & (t[:, 0] <= tt) & (t[:, 5] > tt)].tolist())) for tt in ts]
    ps = sorted(set(panel[mm].tolist()))
    med = [np.median(t[mm & (panel == p), 0]) for p in ps]
    print(f"xcc {x}: panels in flight mean {np.mean(live):.2f} (max {max(live)}); panel period "
          f"{np.mean(np.diff(med[1:-1])) / 1e3 if len(med) > 3 else 0:.1f} kcycles; items per panel {mm.sum() / max(1, len(ps)):.0f}")
