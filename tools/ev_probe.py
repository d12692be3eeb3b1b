#!/usr/bin/env python3
"""The evidence counts of one workload a few times (the program rocprofv3 wraps for the per-kernel split).
    python3 tools/ev_probe.py [workload = pl65536] [ev_hub = 14]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from simrank_amd import ingest, synth
from simrank_amd.engine import HipOps
wl = sys.argv[1] if len(sys.argv) > 1 else "pl65536"
ops = HipOps(0)
if len(sys.argv) > 2:
    ops.set_tuning(ev_hub=int(sys.argv[2]))
csr = ingest.directed(synth.WORKLOADS[wl][0](), False, "from", "to", "weight")[1]
g = ops.graph(csr)
cnt = ops.matrix(csr.n_rows, csr.n_rows, np.uint8, blocked=True)
for _ in range(4):
    ops.evidence_counts(g, 0, cnt)
ops.synchronize()
print("done", wl, int(ops.download_rows(cnt, [csr.n_rows - 1]).sum()))
