#!/bin/bash
set -u
export PYTHONUNBUFFERED=1
O=gpurun_out/r04_s11.log
: > $O
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "evidence" > gpurun_out/r04_s11_pytest.log 2>&1
echo "pytest evidence rc $?" >> $O; tail -5 gpurun_out/r04_s11_pytest.log >> $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "PP or pp or evidence or Apriori" > gpurun_out/r04_s11_pytest2.log 2>&1
echo "pytest pp rc $?" >> $O; tail -5 gpurun_out/r04_s11_pytest2.log >> $O
python - >> $O 2>&1 <<'PY'
import numpy as np
from simrank_amd import ingest, synth
from simrank_amd.engine import HipOps
ops = HipOps(0)
for w in ("pl65536", "ml1m"):
    df = synth.WORKLOADS[w][0]()
    if w == "ml1m":
        _, _, _, _, csr, _ = ingest.bipartite(df, False, "user", "item", "weight")
    else:
        _, csr = ingest.directed(df, False, "from", "to", "weight")
    for tri in (0, 1):
        ops.set_tuning(ev_tri=tri)
        g = ops.graph(csr)
        cnt = ops.matrix(csr.n_rows, csr.n_rows, np.uint8, blocked=True)
        ops.evidence_counts(g, 0, cnt)
        e0, e1 = ops.event(), ops.event()
        ops.record(e0)
        for _ in range(3):
            ops.evidence_counts(g, 0, cnt)
        ops.record(e1); ops.synchronize()
        print(w, "ev_tri", tri, ops.elapsed_ms(e0, e1) / 3, "ms")
        cnt.free(); g.free()
PY
tail -20 $O
