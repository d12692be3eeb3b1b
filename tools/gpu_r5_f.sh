#!/bin/bash
# round 5, call F: evidence counts with the hub columns on the matrix cores: tests, then the kernel times at config 5 / config 3
set -u
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
O=gpurun_out/r5f.log
: > $O
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "evidence or counts" 2>&1 | tail -15 >> $O || { tail -40 $O; exit 1; }
bash tools/gpu_prof_one.sh ev python3 $PWD/tools/ev_probe.py pl65536 > /dev/null 2>&1; python3 - >> $O <<PYX
import csv,glob
for f in glob.glob("gpurun_out/prof_ev/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(r["Name"][:44], r["Calls"], round(float(r["AverageNs"])/1e6,3), "ms")
PYX
timeout -k 10 600 python - >> $O 2>&1 <<'PY'
import sys, time
sys.path.insert(0, ".")
import numpy as np
from simrank_amd import ingest, synth
from simrank_amd.engine import HipOps
ops = HipOps(0)
for wl in ("pl65536", "pl32768d32", "ml1m"):
    df = synth.WORKLOADS[wl][0]()
    if wl == "ml1m":
        _, _, _, _, g12, g21 = ingest.bipartite(df, False, "user", "item", "weight")
        cases = [("ml1m users", g12), ("ml1m items", g21)]
    else:
        cases = [(wl, ingest.directed(df, False, "from", "to", "weight")[1])]
    for name, csr in cases:
        ref = None
        for hub in (0, 14, 7, 28):
            ops.set_tuning(ev_hub=hub)
            g = ops.graph(csr)
            cnt = ops.matrix(csr.n_rows, csr.n_rows, np.uint8, blocked=True)
            ops.evidence_counts(g, 0, cnt)
            ops.synchronize()
            e0, e1 = ops.event(), ops.event()
            ops.record(e0)
            for _ in range(3):
                ops.evidence_counts(g, 0, cnt)
            ops.record(e1)
            ops.synchronize()
            ms = ops.elapsed_ms(e0, e1) / 3
            got = ops.download_rows(cnt, [0, 5, csr.n_rows // 2, csr.n_rows - 1])
            if ref is None:
                ref = got
            same = bool(np.array_equal(ref, got))
            print(f"{name}: ev_hub {hub}: {ms:.2f} ms, sampled rows equal to ev_hub 0: {same}", flush=True)
            cnt.free(); g.free()
ops.set_tuning(ev_hub=14)
PY
cat $O
