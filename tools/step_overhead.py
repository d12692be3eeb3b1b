#!/usr/bin/env python3
"""Host-side cost of one Solver.step on a small workload: with / without leg timing, and the
read-back of the convergence count."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simrank_amd import ingest, synth
from tests.pydriver import LocalWorld, SideSpec, Solver
from simrank_amd.engine import HipOps

ops = HipOps(0)
w = sys.argv[1] if len(sys.argv) > 1 else "er8192"
df = synth.WORKLOADS[w][0]()
_, csr = ingest.directed(df, False, "from", "to", "weight")
s = Solver(lambda r: ops, LocalWorld(1), [SideSpec(csr, csr.rowscale, 0.8)], "sparse")
s.reset()
for _ in range(5):
    s.step(0.0)
def run(n=200):
    ops.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        s.step(0.0)
    ops.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
print(f"{w}: step without timing {run():.4f} ms")
s.enable_timing()
print(f"{w}: step with leg timing {run():.4f} ms; legs {s.leg_times()}")
s.events = None
t0 = time.perf_counter()
for _ in range(200):
    ops.read_changed()
print(f"read_changed alone {(time.perf_counter() - t0) / 200 * 1e3:.4f} ms")
side = s.sides[0][0]
def legs_only(n=200):
    ops.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        side.leg1(s.cur[0][0])
        side.leg2(s.cur[0][0], s.nxt[0][0], 0.0)
    ops.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
print(f"legs back to back, no read-back {legs_only():.4f} ms")
