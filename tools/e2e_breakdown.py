import sys, time
sys.path.insert(0, ".")
import numpy as np, pandas as pd
from simrank_amd import ingest, synth
from tests.pydriver import LocalWorld, SideSpec, Solver
from simrank_amd.engine import HipOps
for w in (sys.argv[1:] or ["bts300", "er8192", "pl32768"]):
    df = synth.WORKLOADS[w][0]()
    out = S = None
    for rep in range(2):
        del out, S                       # (freeing the previous 8 GiB result is not part of a fit)
        t = [time.perf_counter()]
        nodes, csr = ingest.directed(df, False, "from", "to", "weight"); t.append(time.perf_counter())
        ops = HipOps(0); t.append(time.perf_counter())
        sol = Solver(lambda r: ops, LocalWorld(1), [SideSpec(csr, csr.rowscale, 0.8)], "auto"); ops.synchronize(); t.append(time.perf_counter())
        k = sol.run(100, 1e-4); ops.synchronize(); t.append(time.perf_counter())
        S = sol.result(0); t.append(time.perf_counter())
        sol.release(); t.append(time.perf_counter())
        out = pd.DataFrame(S, index=nodes, columns=nodes); t.append(time.perf_counter())
        ops.close()
    names = ["ingest", "HipOps()", "Solver setup", f"run ({k} it, mode {sol.mode})", "download", "release", "DataFrame"]
    print(w, " | ".join(f"{n} {1e3*(b-a):.1f} ms" for n, a, b in zip(names, t, t[1:])))
