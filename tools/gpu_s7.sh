#!/bin/bash
set -u
export PYTHONUNBUFFERED=1
O=gpurun_out/r04_s7.log
: > $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -15 >> $O
echo "pytest rc ${PIPESTATUS[0]}" >> $O
python - >> $O 2>&1 <<'PY'
import time, numpy as np
from simrank_amd import ingest, synth
from simrank_amd.driver import LocalWorld, SideSpec, Solver
from simrank_amd.engine import HipOps, Plan
ops = HipOps(0)
for w in ("er8192", "pl32768d32"):
    df = synth.WORKLOADS[w][0]()
    _, csr = ingest.directed(df, False, "from", "to", "weight")
    s = Solver(lambda r: ops, LocalWorld(1), [SideSpec(csr, csr.rowscale, 0.8)], "sparse")
    K = 200 if w == "er8192" else 40
    s.run(5, 0.0); ops.synchronize()
    t0 = time.perf_counter(); s.run(K, 0.0); ops.synchronize(); t1 = time.perf_counter()
    print(w, "Solver.run (speculative):", K / (t1 - t0), "it/s")
    s.reset(); ops.synchronize()
    t0 = time.perf_counter()
    for _ in range(K): s.step(0.0)
    ops.synchronize(); t1 = time.perf_counter()
    print(w, "Solver.step loop:", K / (t1 - t0), "it/s")
    s.release()
    p = Plan(ops, csr, csr.rowscale)
    p.run(5, 0.0)
    t0 = time.perf_counter(); p.run(K, 0.0); t1 = time.perf_counter()
    print(w, "simrank_plan_run:", K / (t1 - t0), "it/s")
    p.free()
PY
tail -30 $O
