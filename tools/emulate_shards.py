#!/usr/bin/env python3
"""Per-rank kernel times of the sharded path, emulated on ONE GPU: LocalWorld(P) runs the P
column shards one after another (exchange = device copies), so the mean leg time per virtual
rank is what each of P real GPUs would spend computing; the all-to-all itself is not timed."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simrank_amd import ingest, synth                           # noqa: E402
from tests.pydriver import LocalWorld, SideSpec, Solver     # noqa: E402
from simrank_amd.engine import HipOps                           # noqa: E402

ops = HipOps(0)
w = sys.argv[1] if len(sys.argv) > 1 else "pl32768"
df = synth.WORKLOADS[w][0]()
_, csr = ingest.directed(df, False, "from", "to", "weight")
n = csr.n_rows
import itertools
huges = [int(x) for x in os.environ.get('HUGE', '512').split(',')]
dmins = [int(x) for x in os.environ.get('DMIN', '4').split(',')]
for huge, dmin, P in itertools.product(huges, dmins, (1, 8) if len(huges) > 1 else [int(v) for v in os.environ.get('PS', '1,2,4,8').split(',')]):
    for half, st2 in (((False, 1),) if P == 1 else ((False, 1), (True, 1), (True, int(os.environ.get("LEG2_STAGES", "2"))))):
        ops.set_tuning(huge=huge, dense_min=dmin)
        s = Solver(lambda r: ops, LocalWorld(P, symmetric_shards=half, leg2_stages=st2),
                   [SideSpec(csr, csr.rowscale, 0.8)], "sparse")
        s.reset()
        s.step(0.0)
        s.enable_timing()
        for _ in range(3):
            s.step(0.0)
        t = s.leg_times()
        l1, l2 = t["leg1.0"][0], t["leg2.0"][0]
        un = t.get("unpack.0", (0.0, 0))[0]
        xfer = 4.0 * n * n / P * (P - 1) / P
        form = "full form"
        if half and s.sides[0][0].shard_sym:
            form = "half form" + (f" in {len(s.sides[0][0].sh_stages)} stages" if s.sides[0][0].sh_stages else "")
            xfer += 4.0 * s.sides[0][0].sh_chunk * (P - 1)
        print(f"{w} huge={huge} dense_min={dmin} P={P} leg 2 in its {form}: per-rank leg1 {l1:.3f} ms, "
              f"leg2 {l2:.3f} ms, unpack {un:.3f} ms -> compute {l1 + l2 + un:.3f} ms/iteration; "
              f"all-to-all payload per rank {xfer / 2**20:.0f} MiB out + in", flush=True)
        s.release()
        del s
