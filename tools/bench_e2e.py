#!/usr/bin/env python3
"""End-to-end fit() wall time (host DataFrame in, host DataFrame out): ingest, device setup,
iterations, PCIe hand-back — the PCIe-inclusive view of DESIGN.md §6."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simrank_amd.SimRank as SRA                      # noqa: E402
from simrank_amd import ingest, synth                  # noqa: E402

for w in sys.argv[1:] or ["bts300", "er8192", "pl32768"]:
    df = synth.WORKLOADS[w][0]()
    t0 = time.perf_counter()
    nodes, csr = ingest.directed(df, False, "from", "to", "weight")
    t_ingest = time.perf_counter() - t0
    S = None
    for rep in range(2):                                 # second run: warm library / allocator
        est = SRA.SimRank()
        del S                                            # (freeing an 8 GiB result takes 0.3 s: not part of fit)
        t0 = time.perf_counter()
        S = est.fit(df, verbose=False)
        t_fit = time.perf_counter() - t0
    print(f"{w}: N={csr.n_rows} edges={len(df)} ingest (edge list -> CSR) {t_ingest * 1e3:.1f} ms; "
          f"fit() end to end {t_fit:.3f} s for {est.converged_at} iterations "
          f"({S.values.nbytes / 2**30:.2f} GiB float64 result)", flush=True)
    del S

if "--topk" in sys.argv or True:
    df = synth.WORKLOADS["pl32768"][0]()
    top = None
    for rep in range(2):
        del top
        t0 = time.perf_counter()
        top = SRA.SimRank().fit(df, verbose=False, top_k=10)
        t_fit = time.perf_counter() - t0
    print(f"pl32768 fit(top_k=10): {t_fit:.3f} s end to end, {len(top)} rows "
          f"({top.memory_usage(deep=True).sum() / 2**20:.0f} MiB instead of 8 GiB)", flush=True)
