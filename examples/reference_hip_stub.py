"""The binding a maintainer of ysong1231/SimRank would add: replaces the
loop of SimRank.py:124-140 by calls into libsimrank_hip.so, using nothing but ctypes + NumPy.
`iterate(G, C, iterations, eps)` takes the reference's dense `self.Graph.values`."""
import ctypes as C
import os

import numpy as np

_lib = C.CDLL(os.environ.get("SIMRANK_HIP_LIB", os.path.join(
    os.path.dirname(os.path.abspath(__file__)), "..", "simrank_amd", "libsimrank_hip.so")))
_lib.simrank_last_error.restype = C.c_char_p
i64, vp = C.c_int64, C.c_void_p
SLOTS = 1024                                   # SIMRANK_CHANGED_SLOTS


class Epilogue(C.Structure):                   # struct simrank_epilogue
    _fields_ = [("coef", C.c_float), ("lbd", C.c_float),
                ("evidence", vp), ("ld_evidence", i64),
                ("apriori", vp), ("ld_apriori", i64),
                ("previous", vp), ("ld_previous", i64),
                ("eps", C.c_double), ("n_changed", vp),
                ("diag_col0", i64), ("set_diag", C.c_int32), ("symmetric", C.c_int32),
                ("restrict_support", C.c_int32), ("count_any", C.c_int32)]


def _ok(rc):
    if rc:
        raise RuntimeError(_lib.simrank_last_error().decode())


def iterate(G, C_, iterations, eps):
    """-> (S as float64 ndarray, k or None); drop-in for SimRank.py:124-140."""
    n = len(G)
    rows, cols = np.nonzero(G)                 # every row of G is constant-valued (1/in-degree)
    rowptr = np.zeros(n + 1, np.int32)
    np.cumsum(np.bincount(rows, minlength=n), out=rowptr[1:])
    col = np.ascontiguousarray(cols, dtype=np.int32)
    scale = np.zeros(n, np.float32)
    scale[rows] = G[rows, cols]
    g = vp()
    _ok(_lib.simrank_graph_create(i64(n), i64(n), i64(col.size), vp(rowptr.ctypes.data),
                                  vp(col.ctypes.data), vp(scale.ctypes.data), C.byref(g)))
    ld = (n + 3) // 4 * 4
    S, Tt, Sn, cnt = vp(), vp(), vp(), vp()
    for b in (S, Tt, Sn):
        _ok(_lib.simrank_malloc(C.byref(b), C.c_size_t(4 * n * ld)))
    _ok(_lib.simrank_malloc(C.byref(cnt), C.c_size_t(8 * SLOTS)))
    _ok(_lib.simrank_fill_identity(S, i64(n), i64(n), i64(ld), i64(0), None))
    changed, k_conv = (n if eps < 1 else 0), None
    for k in range(iterations):
        if changed == 0:
            k_conv = k
            break
        _ok(_lib.simrank_spmm(g, S, i64(ld), i64(n), Tt, i64(ld), C.c_int32(1), i64(0), i64(0),
                              None, None))                        # Tt = (G.S)^T
        ep = Epilogue(coef=C_, previous=S.value, ld_previous=ld, eps=eps,
                      n_changed=cnt.value, set_diag=1)
        _ok(_lib.simrank_spmm(g, Tt, i64(ld), i64(n), Sn, i64(ld), C.c_int32(0), i64(0), i64(0),
                              C.byref(ep), None))                 # S' = C.G.Tt, diag <- 1
        slots = (C.c_ulonglong * SLOTS)()
        _ok(_lib.simrank_memcpy_d2h(slots, cnt, C.c_size_t(8 * SLOTS), None))
        changed = sum(slots)
        S, Sn = Sn, S
    out = np.empty((n, n))
    _ok(_lib.simrank_download_f64(vp(out.ctypes.data), i64(n), S, i64(ld), i64(n), i64(n), None))
    for b in (S, Tt, Sn, cnt):
        _lib.simrank_free(b)
    _lib.simrank_graph_destroy(g)
    return out, k_conv
