#!/usr/bin/env python3
"""Timeline of ONE launch of the one-launch legs on a side of the MovieLens-shaped graph (or any directed workload): what a
16-column step of the matrix-core phase costs a wave (diagnostic build: bash tools/build_variant.sh fst -DSIMRANK_FUSED_STAMPS).

    SIMRANK_LIB=$PWD/gpurun_variants/libsimrank_hip_fst.so python3 tools/stamps_leg.py [ml1m:1|ml1m:2|workload] [--set k=v,..]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simrank_amd import ingest, synth                              # noqa: E402
from simrank_amd.engine import HipOps                              # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("what", nargs="?", default="ml1m:2")
ap.add_argument("--set", default="")
ap.add_argument("--cap", type=int, default=1 << 15)
args = ap.parse_args()
if "probe" in args.set:
    os.environ["SIMRANK_ENABLE_PROBES"] = "1"
ops = HipOps(0)
if args.set:
    ops.set_tuning(**{k: int(v) for k, v in (kv.split("=") for kv in args.set.split(","))})
lib = ops.lib
lib.simrank_read_fused_stamps.argtypes = [C.c_void_p, C.c_int64]
name, _, side = args.what.partition(":")
df = synth.WORKLOADS[name][0]()
if synth.WORKLOADS[name][1] == "bipartite":
    _, _, _, _, g12, g21 = ingest.bipartite(df, False, "user", "item", "weight")
    csr = g12 if side in ("", "1") else g21
else:
    _, csr = ingest.directed(df, False, "from", "to", "weight")
g = ops.graph(csr)
steps, cov, rem = ops.fused_stats(g)
M, K = csr.n_rows, csr.n_cols
L = K
X = ops.matrix(K, L, blocked=True)
ops.upload(X, (np.random.default_rng(0).random((K, L)) ** 3).astype(np.float32))
Y = ops.matrix(L, M, blocked=True)
for _ in range(3):
    ops.spmm(g, X, Y, transpose_out=True)
ops.synchronize()
buf = np.zeros(args.cap * 8, dtype=np.uint64)
lib.simrank_read_fused_stamps(buf.ctypes.data_as(C.c_void_p), args.cap)
st = buf.reshape(-1, 8)
st = st[st[:, 0] > 0]
t0, t1, t2, t3, t4 = (st[:, i].astype(np.int64) for i in range(5))
unit = (st[:, 6] & np.uint64(0xFFFFFFFF)).astype(np.int64)
panels = (L + 31) // 32
print(f"# {args.what} {args.set}: M={M} K={K} L={L}: {steps} steps, {cov} covered, {rem} gathered per panel; {len(st)} workgroups "
      f"stamped of {panels} panels x {unit.max() + 1} units; launch window {(t4.max() - t0.min()) / 1e3:.1f} kcycles")
dur = t4 - t0
tot = dur.sum()
print(f"shares of workgroup time: prologue {100 * (t1 - t0).sum() / tot:.1f} %, matrix cores + sum {100 * (t2 - t1).sum() / tot:.1f} %, "
      f"gather {100 * (t3 - t2).sum() / tot:.1f} %, barrier + store {100 * (t4 - t3).sum() / tot:.1f} %")
# the wave with most quads makes ceil(ceil(steps_unit / 4) / 4) quads = that x 4 steps: cycles per step of wave 0
nu = unit.max() + 1
print("unit: workgroups, mean total / prologue / mfma / gather / store (kcycles)")
for u in range(nu):
    m = unit == u
    if m.any() and (u < 6 or u % max(1, nu // 8) == 0 or u == nu - 1):
        print(f"  {u:4d}: {m.sum():5d}  {dur[m].mean() / 1e3:8.1f} {(t1 - t0)[m].mean() / 1e3:7.1f} {(t2 - t1)[m].mean() / 1e3:8.1f} "
              f"{(t3 - t2)[m].mean() / 1e3:8.1f} {(t4 - t3)[m].mean() / 1e3:7.1f}")
mf = (t2 - t1).sum()
print(f"matrix-core phase: {mf / 1e6:.1f} Mcycles of workgroup time for {steps * panels} steps (x 12 MFMAs x 32 cycles = "
      f"{steps * panels * 384 / 1e6:.1f} Mcycles of one SIMD's matrix pipe; 4 waves share a workgroup's steps): "
      f"{mf * 4 / (steps * panels):.0f} wave-cycles per step")
