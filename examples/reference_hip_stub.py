"""The binding a maintainer of ysong1231/SimRank would add: replaces the
loop of SimRank.py:124-140 by three calls into libsimrank_hip.so — create a plan, run it, download —
using nothing but ctypes + NumPy.  `iterate(G, C, iterations, eps)` takes the reference's dense
`self.Graph.values`; `evidence=True` is SimRankPP.fit's loop (:346-362)."""
import ctypes as C
import os

import numpy as np

_lib = C.CDLL(os.environ.get("SIMRANK_HIP_LIB", os.path.join(
    os.path.dirname(os.path.abspath(__file__)), "..", "simrank_amd", "libsimrank_hip.so")))
_lib.simrank_last_error.restype = C.c_char_p
i64, vp = C.c_int64, C.c_void_p


class PlanOptions(C.Structure):                # struct simrank_plan_options
    _fields_ = [("coef", C.c_float), ("lbd", C.c_float), ("apriori", vp), ("ld_apriori", i64),
                ("evidence", C.c_int32), ("reorder", C.c_int32),
                ("storage_fp16", C.c_int32), ("dense_terms", C.c_int32)]   # storage_fp16 = 1: config 5's reduced precision


def _ok(rc):
    if rc:
        raise RuntimeError(_lib.simrank_last_error().decode())


def iterate(G, C_, iterations, eps, evidence=False):
    """-> (S as float64 ndarray, k or None); drop-in for SimRank.py:124-140."""
    n = len(G)
    rows, cols = np.nonzero(G)                 # every row of G is constant-valued (1/in-degree)
    rowptr = np.zeros(n + 1, np.int32)
    np.cumsum(np.bincount(rows, minlength=n), out=rowptr[1:])
    col = np.ascontiguousarray(cols, dtype=np.int32)
    scale = np.zeros(n, np.float32)
    scale[rows] = G[rows, cols]
    opt = PlanOptions(coef=C_, evidence=int(evidence), reorder=1)
    plan = vp()
    _ok(_lib.simrank_plan_create(i64(n), i64(col.size), vp(rowptr.ctypes.data), vp(col.ctypes.data),
                                 vp(scale.ctypes.data), C.byref(opt), None, C.byref(plan)))
    done, conv = C.c_int32(0), C.c_int32(-1)
    _ok(_lib.simrank_plan_run(plan, C.c_int32(iterations), C.c_double(eps), C.byref(done), C.byref(conv)))
    out = np.empty((n, n))
    _ok(_lib.simrank_plan_result_f64(plan, vp(out.ctypes.data), i64(n)))
    _lib.simrank_plan_destroy(plan)
    return out, (None if conv.value < 0 else conv.value)


class BiPlanOptions(C.Structure):              # struct simrank_biplan_options
    _fields_ = [("c1", C.c_float), ("c2", C.c_float), ("lbd1", C.c_float), ("lbd2", C.c_float),
                ("apriori1", vp), ("ld_apriori1", i64), ("apriori2", vp), ("ld_apriori2", i64),
                ("evidence", C.c_int32), ("reorder", C.c_int32), ("strict_reference", C.c_int32)]


def iterate_bipartite(G12, G21, C1, C2, iterations, eps):
    """-> (S1, S2 as float64 ndarrays, k or None); drop-in for SimRank.py:280-302 (both dense graphs as the
    reference holds them: every row constant-valued; G21's pattern is G12's transposed)."""
    n1, n2 = G12.shape
    rows, cols = np.nonzero(G12)
    rowptr = np.zeros(n1 + 1, np.int32)
    np.cumsum(np.bincount(rows, minlength=n1), out=rowptr[1:])
    col = np.ascontiguousarray(cols, dtype=np.int32)
    s1 = np.zeros(n1, np.float32)
    s1[rows] = G12[rows, cols]
    r2, c2 = np.nonzero(G21)
    s2 = np.zeros(n2, np.float32)
    s2[r2] = G21[r2, c2]
    opt = BiPlanOptions(c1=C1, c2=C2, evidence=0, reorder=1, strict_reference=1)
    plan = vp()
    _ok(_lib.simrank_biplan_create(i64(n1), i64(n2), i64(col.size), vp(rowptr.ctypes.data), vp(col.ctypes.data),
                                   vp(s1.ctypes.data), vp(s2.ctypes.data), C.byref(opt), None, C.byref(plan)))
    done, conv = C.c_int32(0), C.c_int32(-1)
    _ok(_lib.simrank_biplan_run(plan, C.c_int32(iterations), C.c_double(eps), C.byref(done), C.byref(conv)))
    out1, out2 = np.empty((n1, n1)), np.empty((n2, n2))
    _ok(_lib.simrank_biplan_result_f64(plan, C.c_int32(1), vp(out1.ctypes.data), i64(n1)))
    _ok(_lib.simrank_biplan_result_f64(plan, C.c_int32(2), vp(out2.ctypes.data), i64(n2)))
    _lib.simrank_biplan_destroy(plan)
    return out1, out2, (None if conv.value < 0 else conv.value)
