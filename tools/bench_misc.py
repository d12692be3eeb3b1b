#!/usr/bin/env python3
"""Side measurements: f32 MFMA GEMM rate, and cfg3 (MovieLens-shaped bipartite SimRank++)
per-iteration time in sparse / dense / hybrid mode."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simrank_amd import ingest, synth                      # noqa: E402
from tests.pydriver import LocalWorld, SideSpec, Solver  # noqa: E402
from simrank_amd.engine import HipOps                      # noqa: E402

ops = HipOps(0)
rng = np.random.default_rng(0)
for n in (2048, 4096, 8192):
    a = ops.matrix(n, n); b = ops.matrix(n, n); c = ops.matrix(n, n)
    ops.upload(a, (rng.random((n, n)) - 0.5).astype(np.float32))
    ops.upload(b, (rng.random((n, n)) - 0.5).astype(np.float32))
    for _ in range(2):
        ops.gemm_nt(a, b, c, n, n, n)
    e0, e1 = ops.event(), ops.event()
    ops.record(e0)
    reps = 5
    for _ in range(reps):
        ops.gemm_nt(a, b, c, n, n, n)
    ops.record(e1)
    ms = ops.elapsed_ms(e0, e1) / reps
    print(f"gemm_nt f32 MFMA {n}^3: {ms:.3f} ms  {2 * n**3 / ms / 1e9:.1f} TFLOP/s", flush=True)
    for m in (a, b, c):
        m.free()

if "--cfg3" in sys.argv:
    df = synth.WORKLOADS["ml1m"][0]()
    s1, s2, l1, l2, g12, g21 = ingest.bipartite(df, False, "user", "item", "weight")
    print(f"cfg3: n1={g12.n_rows} n2={g12.n_cols} nnz={g12.nnz} density={g12.density:.4f}", flush=True)
    for mode in ("sparse", "hybrid", "dense"):
        t0 = time.perf_counter()
        specs = [SideSpec(g12, g12.rowscale, 0.8, evidence_from=g12),
                 SideSpec(g21, g21.rowscale, 0.8, evidence_from=g21)]
        sol = Solver(lambda r: ops, LocalWorld(1), specs, mode)
        ops.synchronize()
        t_setup = time.perf_counter() - t0
        sol.reset()
        sol.step(0.0)
        sol.enable_timing()
        ops.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            sol.step(0.0)
        ops.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print(f"cfg3 BipartiteSimRankPP mode={mode}: setup (graphs + evidence) {t_setup:.2f} s, "
              f"{dt * 1e3:.2f} ms/iteration, legs {sol.leg_times()}", flush=True)
        sol.release()

if "--cfg5" in sys.argv:
    # config 5 shape: N = 65536 SimRank++ (evidence), one GPU, f32 gather legs
    df = synth.WORKLOADS["pl65536"][0]()
    _, csr = ingest.directed(df, False, "from", "to", "weight")
    print(f"cfg5: N={csr.n_rows} nnz={csr.nnz}", flush=True)
    t0 = time.perf_counter()
    spec = SideSpec(csr, ingest.spread(csr) * csr.rowscale, 0.8, evidence_from=csr)
    sol = Solver(lambda r: ops, LocalWorld(1), [spec], "sparse")
    ops.synchronize()
    t_setup = time.perf_counter() - t0
    sol.reset()
    sol.step(0.0)
    sol.enable_timing()
    ops.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        sol.step(0.0)
    ops.synchronize()
    dt = (time.perf_counter() - t0) / 3
    t0 = time.perf_counter()
    k = sol.run(100, 1e-4)
    ops.synchronize()
    print(f"cfg5 SimRankPP N=65536: setup (graph + evidence counts) {t_setup:.2f} s, "
          f"{dt * 1e3:.1f} ms/iteration, legs {sol.leg_times()}, converged at {k} in "
          f"{time.perf_counter() - t0:.2f} s", flush=True)
    sol.release()
