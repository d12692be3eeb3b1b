import sys, time
sys.path.insert(0, '.')
import numpy as np
from simrank_amd.engine import HipOps
ops = HipOps(0)
n = 32768
a = ops.matrix(n, n); b = ops.matrix(n, n)
ops.fill_identity(a, 0)
for name, fn, nbytes in (("d2d copy", lambda: ops.copy(b, a), 2 * a.nbytes),
                         ("fill_identity", lambda: ops.fill_identity(b, 0), a.nbytes)):
    fn(); fn()
    e0, e1 = ops.event(), ops.event()
    ops.record(e0)
    for _ in range(5): fn()
    ops.record(e1)
    ms = ops.elapsed_ms(e0, e1) / 5
    print(f"{name}: {ms:.3f} ms, {nbytes / ms / 1e9:.2f} TB/s total traffic")
idx = ops.index_vector(np.arange(n))
for name, fn in (("permute identity rows+cols", lambda: ops.permute(a, b, idx, idx)), ("permute rows only", lambda: ops.permute(a, b, idx, None))):
    fn(); fn()
    e0, e1 = ops.event(), ops.event()
    ops.record(e0)
    for _ in range(5): fn()
    ops.record(e1)
    ms = ops.elapsed_ms(e0, e1) / 5
    print(f"{name}: {ms:.3f} ms, {2 * a.nbytes / ms / 1e9:.2f} TB/s total traffic")
