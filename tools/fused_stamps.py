#!/usr/bin/env python3
"""Timeline of the workgroups of the one-launch leg 1 (diagnostic build only:
bash tools/build_variant.sh fst -DSIMRANK_FUSED_STAMPS).

    SIMRANK_LIB=$PWD/gpurun_variants/libsimrank_hip_fst.so SIMRANK_FST_BASE=<first block> \
        python tools/fused_stamps.py [workload] [--set knob=v,...] [--out file.npz]

Per stamped workgroup: start, end of the prologue, end of the matrix-core phase, last barrier before the tile
store, end (s_memtime of wave 0), XCC id, hardware id, panel, unit.  Prints the phase shares, the duration by
unit rank and how many panels an XCD has in flight.
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
ap.add_argument("--set", default="")
ap.add_argument("--out", default="")
ap.add_argument("--cap", type=int, default=1 << 15)
args = ap.parse_args()
if "probe" in args.set:
    os.environ["SIMRANK_ENABLE_PROBES"] = "1"
ops = HipOps(0)
if args.set:
    ops.set_tuning(**{k: int(v) for k, v in (kv.split("=") for kv in args.set.split(","))})
lib = ops.lib
lib.simrank_read_fused_stamps.argtypes = [C.c_void_p, C.c_int64]
df = synth.WORKLOADS[args.workload][0]()
_, csr = ingest.directed(df, False, "from", "to", "weight")
s = Solver(lambda r: ops, LocalWorld(1), [SideSpec(csr, csr.rowscale, 0.8)], "sparse")
s.reset()
for _ in range(3):
    s.step(0.0)
ops.synchronize()
buf = np.zeros(args.cap * 8, dtype=np.uint64)
lib.simrank_read_fused_stamps(buf.ctypes.data_as(C.c_void_p), args.cap)
st = buf.reshape(-1, 8)
st = st[st[:, 0] > 0]
t0, t1, t2, t3, t4 = (st[:, i].astype(np.int64) for i in range(5))
xcc = (st[:, 5] >> np.uint64(32)).astype(np.int64)
hw = (st[:, 5] & np.uint64(0xFFFFFFFF)).astype(np.int64)
panel = (st[:, 6] >> np.uint64(32)).astype(np.int64)
unit = (st[:, 6] & np.uint64(0xFFFFFFFF)).astype(np.int64)
if args.out:
    np.savez_compressed(args.out, st=st)
base = t0.min()
dur = t4 - t0
print(f"# {args.workload} {args.set}: {len(st)} workgroups stamped, panels {panel.min()}..{panel.max()}, "
      f"window {(t4.max() - base) / 1e3:.1f} kcycles (100 MHz ticks? no: shader clocks)")
tot = dur.sum()
print(f"phase shares of workgroup time: prologue {100 * (t1 - t0).sum() / tot:.1f} %, matrix cores + sum "
      f"{100 * (t2 - t1).sum() / tot:.1f} %, gather {100 * (t3 - t2).sum() / tot:.1f} %, barrier + store "
      f"{100 * (t4 - t3).sum() / tot:.1f} %")
print(f"xcc ids seen: {sorted(set(xcc.tolist()))}; blockIdx % 8 == xcc for "
      f"{100 * np.mean((panel % 8) == xcc):.1f} % of the workgroups (panel % 8 vs XCC)")
nu = unit.max() + 1
print("duration by unit rank (launch order; kcycles): mean over panels")
for lo in list(range(0, min(nu, 16))) + list(range(16, nu, max(1, nu // 12))):
    m = unit == lo
    if m.any():
        print(f"   unit {lo:4d}: total {dur[m].mean() / 1e3:8.1f}  prologue {(t1 - t0)[m].mean() / 1e3:6.1f}  mfma "
              f"{(t2 - t1)[m].mean() / 1e3:8.1f}  gather {(t3 - t2)[m].mean() / 1e3:8.1f}  store {(t4 - t3)[m].mean() / 1e3:6.1f}")
print(f"all units: mean {dur.mean() / 1e3:.1f} kcycles, p50 {np.median(dur) / 1e3:.1f}, p90 {np.percentile(dur, 90) / 1e3:.1f}, "
      f"max {dur.max() / 1e3:.1f}")
# panels in flight per XCD over time
for x in sorted(set(xcc.tolist()))[:2]:
    m = xcc == x
    ts = np.linspace(np.percentile(t0[m], 20), np.percentile(t4[m], 80), 200)
    live = []
    wgs = []
    for t in ts:
        a = m & (t0 <= t) & (t4 > t)
        live.append(len(set(panel[a].tolist())))
        wgs.append(a.sum())
    print(f"xcc {x}: panels in flight mean {np.mean(live):.2f} (min {min(live)}, max {max(live)}), workgroups resident "
          f"mean {np.mean(wgs):.1f}; panels seen {len(set(panel[m].tolist()))}")
    # time per panel: spacing of the median start time of consecutive panels
    ps = sorted(set(panel[m].tolist()))
    med = [np.median(t0[m & (panel == p)]) for p in ps]
    if len(med) > 2:
        print(f"   panel period (median start to median start): {np.mean(np.diff(med)) / 1e3:.1f} kcycles")
# CU occupancy: workgroups per (xcc, cu) at a time
cu = (hw >> 8) & 0xF
se = (hw >> 13) & 0x7
sh = (hw >> 12) & 1
print(f"hw id fields: cu 0..{cu.max()}, sh 0..{sh.max()}, se 0..{se.max()}")
