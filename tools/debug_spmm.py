import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.test_gpu_kernels import random_csr, dense64, put
from simrank_amd.engine import HipOps
ops = HipOps(0)
M, K, L = 300, 257, 100
csr = random_csr(M, K, 9, seed=M + L, heavy={1: 200, 3: 70})
X = np.random.default_rng(1).random((K, L)).astype(np.float32)
g = ops.graph(csr); x = put(ops, X); y = ops.matrix(M, L)
ops.spmm(g, x, y)
got = ops.download(y); want = dense64(csr) @ X.astype(np.float64)
bad = np.abs(got - want) > 1e-5 * np.abs(want) + 1e-12
deg = np.diff(csr.rowptr)
rows = np.flatnonzero(bad.any(axis=1))
print("bad rows", len(rows), "of", M)
for r in rows[:20]:
    cols = np.flatnonzero(bad[r])
    print(r, "deg", deg[r], "rs", csr.rowscale[r], "badcols", cols[:8], len(cols), "got/want", got[r, cols[0]], want[r, cols[0]])
