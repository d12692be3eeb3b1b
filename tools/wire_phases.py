#!/usr/bin/env python3
"""The fp16 wire of the sharded loop on eight virtual ranks of ONE GPU (device copies stand in for the links): ms per rank and
update of the f32 and fp16 wires in both forms of leg 2, with the plan's own phase timings (all ranks' launches of a phase are
queued back to back: a phase's time / P = one rank's).   python3 tools/wire_phases.py [workload] [P] [stages]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simrank_amd import ingest, synth                           # noqa: E402
from simrank_amd.engine import HipOps, ShardPlans               # noqa: E402

w = sys.argv[1] if len(sys.argv) > 1 else "pl32768d32"
P = int(sys.argv[2]) if len(sys.argv) > 2 else 8
stages = int(sys.argv[3]) if len(sys.argv) > 3 else 1
ops = HipOps(0)
_, csr = ingest.directed(synth.WORKLOADS[w][0](), False, "from", "to", "weight")
for form in (1, 0):
    for wire in (False, True):
        sp = ShardPlans(ops, csr, world=P, leg2_form=form, stages=stages, wire_fp16=wire)
        sp.step(0.0)
        ops.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            sp.step(0.0, exact_count=False)
        ops.synchronize()
        ms = (time.perf_counter() - t0) / 4 * 1e3
        sp.set_timing(4)
        for _ in range(4):
            sp.step(0.0, exact_count=False)
        ops.synchronize()
        t = sp.timings()
        sp.free()
        print(f"{w} P={P} stages={stages} leg 2 {'half' if form else 'full'} form, {'fp16' if wire else 'f32 '} wire: "
              f"{ms / P:.3f} ms per rank and update; phases / P: " + ", ".join(f"{k} {v / P:.3f}" for k, v in t.items() if k != "updates"),
              flush=True)
