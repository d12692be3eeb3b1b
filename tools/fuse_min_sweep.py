#!/usr/bin/env python3
"""The threshold of the dense sets (`fuse_min`: a column goes to the matrix cores when that many rows of a 128-row block
reference it) against the legs' times: per workload and threshold the plan's 16-column steps, the entries they cover, the
gathered remainder, and the HIP-event times of both legs.  `python3 tools/fuse_min_sweep.py [workloads] [thresholds | knob lists] [pp[,fp16]]`."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simrank_amd import ingest, synth                              # noqa: E402
from simrank_amd.engine import HipOps                              # noqa: E402
from tests.pydriver import LocalWorld, SideSpec, Solver            # noqa: E402

names = (sys.argv[1] if len(sys.argv) > 1 else "ml1m,er8192,pl32768d32").split(",")
# settings: thresholds ("2,3,4") or knob lists separated by ';' ("fuse_min=3;fuse_min=0,fuse_pays=256")
arg = sys.argv[2] if len(sys.argv) > 2 else "2,3,4,5,6,8"
ths = ([dict(kv.split("=") for kv in st.split(",")) for st in arg.split(";")] if "=" in arg
       else [dict(fuse_min=x) for x in arg.split(",")])
mode = sys.argv[3] if len(sys.argv) > 3 else ""             # "pp" (SimRank++), "pp,fp16" (... on fp16-held matrices)
storage = "fp16" if "fp16" in mode else "f32"
ops = HipOps(0)
for name in names:
    make, kind = synth.WORKLOADS[name]
    df = make()
    for t in ths:
        ops.set_tuning(**{k: int(v) for k, v in t.items()})
        if kind == "bipartite":
            _, _, _, _, g12, g21 = ingest.bipartite(df, False, "user", "item", "weight")
            specs = [SideSpec(g12, g12.rowscale, 0.8, evidence_from=g12), SideSpec(g21, g21.rowscale, 0.8, evidence_from=g21)]
        else:
            _, csr = ingest.directed(df, False, "from", "to", "weight")
            specs = [SideSpec(csr, csr.rowscale, 0.8, storage=storage)]
            if "pp" in mode:
                specs = [SideSpec(csr, ingest.spread(csr) * csr.rowscale, 0.8, evidence_from=csr, storage=storage)]
        s = Solver(lambda r: ops, LocalWorld(1), specs, "sparse")
        s.exact_count = True
        stats = [ops.fused_stats(next(iter(sd.values())).graph) for sd in s.sides]
        s.reset()
        s.step(0.0)
        s.enable_timing()
        for _ in range(10 if kind == "bipartite" else 4):
            s.step(0.0)
        ops.synchronize()
        legs = {k: round(v[0], 3) for k, v in s.leg_times().items() if k.startswith("leg")}
        print(f"{name} {mode} {t}: (steps, covered, remainder) {stats} legs {legs} sum {sum(legs.values()):.3f} ms", flush=True)
        s.release()
        del s
        ops.set_tuning(fuse_min=0, fuse_pays=-1, fuse_unit=48, fuse_rows=8192, fuse_group=3)
