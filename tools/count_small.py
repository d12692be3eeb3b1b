#!/usr/bin/env python3
"""How many entries of the converged S of a workload lie under the absolute floor of the parity comparisons
(atol = 1e-30 beside rtol = 1e-5: entries below ~1e-25 are effectively exempt from the relative bar)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simrank_amd.SimRank as SRA
from simrank_amd import synth
wl = sys.argv[1] if len(sys.argv) > 1 else "pl32768d32"
df = synth.WORKLOADS[wl][0]()
est = SRA.SimRank()
S = est.fit(df, verbose=False).values
n = S.shape[0]
print(f"{wl}: N={n}, converged at {est.converged_at}; entries: total {S.size}, == 0: {(S == 0).sum()}, "
      f"0 < s < 1e-25: {((S > 0) & (S < 1e-25)).sum()}, 0 < s < 1e-12: {((S > 0) & (S < 1e-12)).sum()}, "
      f"smallest positive {S[S > 0].min():.3e}, median off-diagonal {np.median(S[~np.eye(n, dtype=bool)]):.3e}")
