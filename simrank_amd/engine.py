"""Device layer: thin Python objects over the C ABI (include/simrank_hip.h).

``HipOps`` is the operation set the iteration driver (``driver.py``) is written against:
allocate / upload / download matrices, upload a graph, and launch the kernels.  It owns
no algorithm.  It fails loudly when the library or the GPU is missing.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np

from . import _lib, hostpool
from ._lib import Epilogue, SimRankHipError, check
from .ingest import CSR


CHANGED_SLOTS = 1024      # SIMRANK_CHANGED_SLOTS of include/simrank_hip.h


class Matrix:
    """Device matrix (float32, uint8 or int32): row-major with a leading dimension, or
    PANEL-BLOCKED (``blocked=True``): 32-column panels of ``rows_pad`` rows, element (r, c) at
    ((c >> 5) * rows_pad + r) * 32 + (c & 31) — the layout the single-rank solver iterates in
    (include/simrank_hip.h, "PANEL-BLOCKED operands").  A float16 matrix exists only panel-blocked and
    with 64-column panels (a row segment is one 128-byte line either way; "FP16 STORAGE")."""

    def __init__(self, ops, rows: int, cols: int, dtype, ld: int | None = None,
                 external=None, blocked: bool = False):
        """``external``: an object with ``data_ptr()`` (a torch tensor) whose memory is
        used instead of allocating; it is kept alive with the matrix."""
        self.ops = ops
        self.rows, self.cols = int(rows), int(cols)
        self.dtype = np.dtype(dtype)
        self.blocked = bool(blocked)
        self.scale = 1.0                   # float16 matrices: stored value = value x scale (HipOps.HALF_SCALE in the solver)
        if self.blocked:
            assert external is None and ld is None
            # (+8 rows: panels are not a power of two apart, and every panel starts 16-byte aligned)
            self.rows_pad = -(-max(1, self.rows) // 8) * 8 + BLOCK_PAD_ROWS
            self.ld = 64 if self.dtype == np.float16 else 32
            self.panels = -(-max(1, self.cols) // self.ld)
            self.nbytes = self.panels * self.rows_pad * self.ld * self.dtype.itemsize
        else:
            assert self.dtype != np.float16, "float16 matrices are panel-blocked"
            self.rows_pad = 0
            self.ld = int(ld if ld is not None else ops.pitch(cols, self.dtype))
            self.nbytes = max(1, self.rows) * self.ld * self.dtype.itemsize
        self.external = external
        self.ptr = external.data_ptr() if external is not None else ops._malloc(self.nbytes)
        if self.blocked and self.dtype.itemsize == 1:
            # evidence counts: rows without in-edges are not visited by the counting kernel and must read as zero.  The f32 /
            # fp16 matrices of an update are written whole before they are read (identity fill, leg 1, leg 2) and the
            # padding rows / columns of a panel are read by nobody: no memset (three of them were 10 ms of a config-5 fit)
            check(ops.lib.simrank_memset(C.c_void_p(self.ptr), 0, self.nbytes, ops.stream), "simrank_memset")

    def free(self):
        if self.ptr and self.external is None:
            self.ops._free(self.ptr)
        self.ptr = 0
        self.external = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Graph:
    def __init__(self, ops, csr: CSR, rowscale: np.ndarray | None = None, counting=None):
        """``counting`` = (counts Matrix, col0): the evidence counts of the pattern (columns col0 ... of it) are queued as
        soon as the pattern is on the device and run beside the host plan builders (simrank_graph_create_counting)."""
        lib = ops.lib
        rs = np.ascontiguousarray(csr.rowscale if rowscale is None else rowscale,
                                  dtype=np.float32)
        rowptr = np.ascontiguousarray(csr.rowptr, dtype=np.int32)
        col = np.ascontiguousarray(csr.col, dtype=np.int32)
        h = C.c_void_p()
        if counting is not None:
            cnt, col0 = counting
            check(lib.simrank_graph_create_counting(csr.n_rows, csr.n_cols, col.size, rowptr.ctypes.data,
                                                    col.ctypes.data if col.size else None, rs.ctypes.data, int(col0),
                                                    cnt.cols, cnt.ptr, cnt.ld, cnt.rows_pad if cnt.blocked else 0,
                                                    ops.stream, C.byref(h)), "simrank_graph_create_counting")
        else:
            check(lib.simrank_graph_create(csr.n_rows, csr.n_cols, col.size, rowptr.ctypes.data,
                                           col.ctypes.data if col.size else None, rs.ctypes.data,
                                           C.byref(h)), "simrank_graph_create")
        self.ops, self.handle = ops, h
        self.n_rows, self.n_cols, self.nnz = csr.n_rows, csr.n_cols, int(col.size)

    def free(self):
        if self.handle:
            self.ops.lib.simrank_graph_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class PlanOptions(C.Structure):                # struct simrank_plan_options
    _fields_ = [("coef", C.c_float), ("lbd", C.c_float), ("apriori", C.c_void_p), ("ld_apriori", C.c_int64),
                ("evidence", C.c_int32), ("reorder", C.c_int32), ("storage_fp16", C.c_int32), ("dense_terms", C.c_int32)]


def _progress_callback(on_iteration, on_converged):
    """The console hooks of a C-level loop as a simrank_progress_fn (+ the list a Python exception raised inside one is
    kept in: the loop is told to stop and the caller re-raises)."""
    raised = []

    def hook(_user, k, converged):
        try:
            fn = on_converged if converged else on_iteration
            if fn is not None:
                fn(int(k))
            return 0
        except BaseException as e:            # (ctypes would print and swallow it)
            raised.append(e)
            return 1
    return _lib.PROGRESS_FN(hook), raised


class Plan:
    """The reference loop on one GPU behind the C ABI (simrank_plan_*: SimRank.py:124-141, :346-363,
    :440-455): CSR + per-row scale in the caller's node order in, S (float64, caller's order) out."""

    def __init__(self, ops, csr: CSR, rowscale=None, coef: float = 0.8, evidence: bool = False,
                 apriori=None, lbd: float = 0.0, reorder: bool = True, storage: str = "f32", dense_terms: int = 3):
        """``storage``: "f32", or "fp16" (matrices held in fp16: config 5's reduced-precision mode).
        ``dense_terms``: 3 = exact products on the matrix cores, 1 = one fp16 operand term."""
        assert storage in ("f32", "fp16")
        self.ops = ops
        rs = np.ascontiguousarray(csr.rowscale if rowscale is None else rowscale, dtype=np.float32)
        rowptr = np.ascontiguousarray(csr.rowptr, dtype=np.int32)
        col = np.ascontiguousarray(csr.col, dtype=np.int32)
        ap = None if apriori is None else np.ascontiguousarray(apriori, dtype=np.float32)
        opt = PlanOptions(coef=coef, lbd=lbd, apriori=None if ap is None else ap.ctypes.data,
                          ld_apriori=0 if ap is None else ap.shape[1], evidence=int(evidence), reorder=int(reorder),
                          storage_fp16=int(storage == "fp16"), dense_terms=int(dense_terms))
        h = C.c_void_p()
        with HipOps._knob_lock:          # (the graph inside snapshots the process-wide knobs: not while another
            check(ops.lib.simrank_plan_create(csr.n_rows, col.size, rowptr.ctypes.data,       # thread has per-graph ones set)
                                              col.ctypes.data if col.size else None, rs.ctypes.data, C.byref(opt),
                                              ops.stream, C.byref(h)), "simrank_plan_create")
        self.handle, self.n = h, csr.n_rows

    def run(self, iterations: int, eps: float, on_iteration=None, on_converged=None):
        """-> (updates applied, loop index at which the convergence test passed or None).  ``on_iteration(k)`` /
        ``on_converged(k)``: the reference's console hooks (SimRank.py:131-135), called from inside the C loop."""
        done, conv = C.c_int32(0), C.c_int32(-1)
        if on_iteration is None and on_converged is None:
            check(self.ops.lib.simrank_plan_run(self.handle, int(iterations), float(eps), C.byref(done), C.byref(conv)),
                  "simrank_plan_run")
        else:
            cb, raised = _progress_callback(on_iteration, on_converged)
            check(self.ops.lib.simrank_plan_run_cb(self.handle, int(iterations), float(eps), cb, None, C.byref(done),
                                                   C.byref(conv)), "simrank_plan_run_cb")
            if raised:
                raise raised[0]
        return done.value, (None if conv.value < 0 else conv.value)

    def set_timing(self, updates: int):
        """HIP events around both legs of the next ``updates`` updates, on the plan's stream (0 = off)."""
        check(self.ops.lib.simrank_plan_set_timing(self.handle, int(updates)), "simrank_plan_set_timing")

    def leg_times(self):
        """(mean ms of leg 1, mean ms of leg 2, updates stamped) since ``set_timing`` / the last call."""
        a, b, n = C.c_double(0), C.c_double(0), C.c_int32(0)
        check(self.ops.lib.simrank_plan_leg_times(self.handle, C.byref(a), C.byref(b), C.byref(n)), "simrank_plan_leg_times")
        return a.value, b.value, n.value

    def graph_handle(self):
        """The plan's graph object (for simrank_graph_fused_stats / _dense_stats); owned by the plan."""
        g = C.c_void_p()
        check(self.ops.lib.simrank_plan_info(self.handle, None, None, C.byref(g)), "simrank_plan_info")
        return g

    def evidence_counts(self) -> np.ndarray:
        """uint8 [n, n] common in-neighbour counts (saturated at 255) in the caller's order."""
        out = np.empty((self.n, self.n), dtype=np.uint8)
        check(self.ops.lib.simrank_plan_evidence_u8(self.handle, out.ctypes.data, self.n), "simrank_plan_evidence_u8")
        return out

    def trim(self):
        """Release the matrices of the loop; the evidence counts stay readable."""
        if self.handle:
            check(self.ops.lib.simrank_plan_trim(self.handle), "simrank_plan_trim")

    def reset(self):
        check(self.ops.lib.simrank_plan_reset(self.handle), "simrank_plan_reset")

    def step(self, eps: float, exact_count: bool = True) -> int:
        c = C.c_int64(0)
        check(self.ops.lib.simrank_plan_step(self.handle, float(eps), int(exact_count), C.byref(c)), "simrank_plan_step")
        return c.value

    def result(self) -> np.ndarray:
        out = hostpool.empty_f64(self.n, self.n)        # (a frame the caller dropped earlier, when there is one)
        check(self.ops.lib.simrank_plan_result_f64(self.handle, out.ctypes.data, self.n), "simrank_plan_result_f64")
        return out

    def rows(self, rows) -> np.ndarray:
        """float32 [len(rows), n]: those rows of the current similarity matrix, caller's node order on both axes."""
        ids = np.ascontiguousarray(rows, dtype=np.int32)
        out = np.empty((ids.size, self.n), dtype=np.float32)
        check(self.ops.lib.simrank_plan_rows_f32(self.handle, ids.ctypes.data, int(ids.size), out.ctypes.data, self.n),
              "simrank_plan_rows_f32")
        return out

    def topk(self, k: int, exclude_diag: bool = True):
        """(ids int32 [n, k], values float32 [n, k]): the k most similar nodes of every node, caller's ids."""
        idx = np.empty((self.n, k), dtype=np.int32)
        val = np.empty((self.n, k), dtype=np.float32)
        check(self.ops.lib.simrank_plan_topk(self.handle, int(k), int(exclude_diag), idx.ctypes.data, val.ctypes.data),
              "simrank_plan_topk")
        return idx, val

    def free(self):
        if self.handle:
            self.ops.lib.simrank_plan_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class ShardPlanOptions(C.Structure):           # struct simrank_shardplan_options
    _fields_ = [("coef", C.c_float), ("lbd", C.c_float), ("apriori", C.c_void_p), ("ld_apriori", C.c_int64),
                ("evidence", C.c_int32), ("reorder", C.c_int32), ("leg2_form", C.c_int32), ("stages", C.c_int32),
                ("wire_fp16", C.c_int32), ("storage_fp16", C.c_int32)]


class ShardPlans:
    """This process's share of a SHARDED fit behind the C ABI (simrank_shardplan_*: K10, S split by column block over
    ``world`` ranks): ONE plan when ``comm`` is an RCCL communicator of a multi-process world (``rccl_comm``), or all
    ``world`` plans of an in-process group of virtual ranks on this device (``comm=None``: the exchanges are device
    copies — tests and single-GPU emulation).  ``run`` / ``step`` / ``result`` are collective."""

    def __init__(self, ops, csr: CSR, rowscale=None, world: int = 1, comm=None, coef: float = 0.8, evidence: bool = False,
                 apriori=None, lbd: float = 0.0, reorder: bool = True, leg2_form: int = -1, stages: int = 0,
                 wire_fp16: bool = False, storage: str = "f32"):
        """``storage="fp16"``: matrices held in fp16 on every rank (config 5's reduced precision on shards)."""
        assert storage in ("f32", "fp16")
        self.ops, self.n = ops, csr.n_rows
        lib = ops.lib
        rs = np.ascontiguousarray(csr.rowscale if rowscale is None else rowscale, dtype=np.float32)
        rowptr = np.ascontiguousarray(csr.rowptr, dtype=np.int32)
        col = np.ascontiguousarray(csr.col, dtype=np.int32)
        ap = None if apriori is None else np.ascontiguousarray(apriori, dtype=np.float32)
        opt = ShardPlanOptions(coef=coef, lbd=lbd, apriori=None if ap is None else ap.ctypes.data,
                               ld_apriori=0 if ap is None else ap.shape[1], evidence=int(evidence), reorder=int(reorder),
                               leg2_form=int(leg2_form), stages=int(stages), wire_fp16=int(wire_fp16),
                               storage_fp16=int(storage == "fp16"))
        self.own_comms = comm is None
        if comm is None:
            arr = (C.c_void_p * world)()
            check(lib.simrank_comm_local_group(int(world), arr), "simrank_comm_local_group")
            self.comms = [C.c_void_p(arr[r]) for r in range(world)]
        else:
            self.comms = [comm]
        self.plans = []
        with HipOps._knob_lock:
            for c in self.comms:
                h = C.c_void_p()
                check(lib.simrank_shardplan_create(csr.n_rows, col.size, rowptr.ctypes.data,
                                                   col.ctypes.data if col.size else None, rs.ctypes.data, C.byref(opt),
                                                   c, ops.stream, C.byref(h)), "simrank_shardplan_create")
                self.plans.append(h)
        self._arr = (C.c_void_p * len(self.plans))(*[h.value for h in self.plans])

    @staticmethod
    def rccl_unique_id(lib) -> bytes:
        buf = C.create_string_buffer(128)
        check(lib.simrank_comm_unique_id(buf), "simrank_comm_unique_id")
        return buf.raw

    @staticmethod
    def rccl_comm(lib, unique_id: bytes, rank: int, world: int):
        h = C.c_void_p()
        check(lib.simrank_comm_create(C.c_char_p(unique_id), int(rank), int(world), C.byref(h)), "simrank_comm_create")
        return h

    def info(self, i: int = 0) -> dict:
        n, lo, hi = C.c_int64(), C.c_int64(), C.c_int64()
        half, stages, updates = C.c_int32(), C.c_int32(), C.c_int32()
        check(self.ops.lib.simrank_shardplan_info(self.plans[i], C.byref(n), C.byref(lo), C.byref(hi), C.byref(half),
                                                  C.byref(stages), C.byref(updates)), "simrank_shardplan_info")
        return dict(n=n.value, col_lo=lo.value, col_hi=hi.value, half_form=bool(half.value), stages=stages.value,
                    updates=updates.value)

    TIMING_KEYS = ("leg1_ms", "exchange1_ms", "wait_before_leg2_ms", "leg2_ms", "count_and_exchange2_ms", "update_ms")

    def set_timing(self, updates: int):
        """HIP events at the boundaries of the next ``updates`` updates of the first local plan (0 = off)."""
        check(self.ops.lib.simrank_shardplan_set_timing(self.plans[0], int(updates)), "simrank_shardplan_set_timing")

    def timings(self) -> dict:
        """Mean milliseconds per update since ``set_timing`` (keys: TIMING_KEYS; + "updates")."""
        ms = (C.c_double * 6)()
        n = C.c_int32(0)
        check(self.ops.lib.simrank_shardplan_timings(self.plans[0], ms, 6, C.byref(n)), "simrank_shardplan_timings")
        out = {k: float(ms[i]) for i, k in enumerate(self.TIMING_KEYS)}
        out["updates"] = n.value
        return out

    def reset(self):
        check(self.ops.lib.simrank_shardplan_reset(self._arr, len(self.plans)), "simrank_shardplan_reset")

    def step(self, eps: float, exact_count: bool = True) -> int:
        c = C.c_int64(0)
        check(self.ops.lib.simrank_shardplan_step(self._arr, len(self.plans), float(eps), int(exact_count), C.byref(c)),
              "simrank_shardplan_step")
        return c.value

    def run(self, iterations: int, eps: float):
        """-> (updates applied, loop index at which the convergence test passed or None); the same on every rank."""
        done, conv = C.c_int32(0), C.c_int32(-1)
        check(self.ops.lib.simrank_shardplan_run(self._arr, len(self.plans), int(iterations), float(eps), C.byref(done),
                                                 C.byref(conv)), "simrank_shardplan_run")
        return done.value, (None if conv.value < 0 else conv.value)

    def block(self, i: int = 0):
        """(float64 [n, columns of plan i] in the caller's row order, the caller's node ids of those columns)."""
        inf = self.info(i)
        w = inf["col_hi"] - inf["col_lo"]
        out = np.empty((self.n, w), dtype=np.float64)
        ids = np.empty(w, dtype=np.int32)
        check(self.ops.lib.simrank_shardplan_block_f64(self.plans[i], out.ctypes.data, max(1, w)), "simrank_shardplan_block_f64")
        check(self.ops.lib.simrank_shardplan_columns(self.plans[i], ids.ctypes.data), "simrank_shardplan_columns")
        return out, ids

    def result(self, root: int = 0, i_am_root: bool = True):
        """The whole matrix (float64, caller's order) on rank ``root``; None elsewhere.  Collective."""
        out = hostpool.empty_f64(self.n, self.n) if i_am_root else None
        check(self.ops.lib.simrank_shardplan_result_f64(self._arr, len(self.plans), int(root),
                                                        out.ctypes.data if out is not None else None, self.n),
              "simrank_shardplan_result_f64")
        return out

    def topk(self, k: int, exclude_diag: bool = True, root: int = 0, i_am_root: bool = True):
        """(ids int32 [n, k], values float32 [n, k]) on rank ``root``: the k most similar nodes of every node.  Collective."""
        idx = np.empty((self.n, k), dtype=np.int32) if i_am_root else None
        val = np.empty((self.n, k), dtype=np.float32) if i_am_root else None
        check(self.ops.lib.simrank_shardplan_topk(self._arr, len(self.plans), int(root), int(k), int(exclude_diag),
                                                  idx.ctypes.data if i_am_root else None,
                                                  val.ctypes.data if i_am_root else None), "simrank_shardplan_topk")
        return idx, val

    def free(self):
        lib = self.ops.lib
        for h in self.plans:
            lib.simrank_shardplan_destroy(h)
        self.plans = []
        if self.own_comms:
            for c in self.comms:
                lib.simrank_comm_destroy(c)
        self.comms = []

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass



class ThreadRanks:
    """``world`` ranks of a THREAD group (simrank_comm_thread_group) on one device: ``run(fn)`` calls ``fn(rank, ops, comm)``
    on ``world`` host threads at once, each with its own engine (its own streams) — the ranks of an RCCL world as threads of
    this process, over the library's in-process transport.  What ``fn`` builds with ``comm`` (``ShardPlans(..., comm=comm)``,
    ``ShardBiPlans(..., comm=comm)``) takes the code path of an RCCL rank; its calls are collectives between the threads."""

    def __init__(self, world: int, device: int = 0):
        self.world, self.device = int(world), int(device)
        self.lib = _lib.load()
        arr = (C.c_void_p * self.world)()
        check(self.lib.simrank_comm_thread_group(self.world, arr), "simrank_comm_thread_group")
        self.comms = [C.c_void_p(arr[r]) for r in range(self.world)]

    def run(self, fn, timeout: float = 300.0):
        """-> [fn(rank, ops, comm) for every rank].  The first exception of any rank is raised; ranks still running after
        ``timeout`` seconds raise TimeoutError (the transport's own watchdog, SIMRANK_THREAD_COMM_TIMEOUT, turns a rank that
        waits for a peer into an error on every rank well before that)."""
        import threading
        out, err = [None] * self.world, [None] * self.world

        def body(r):
            try:
                out[r] = fn(r, HipOps(self.device), self.comms[r])
            except BaseException as e:          # noqa: BLE001 — handed to the caller
                err[r] = e
        ts = [threading.Thread(target=body, args=(r,), daemon=True) for r in range(self.world)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(timeout)
        if any(t.is_alive() for t in ts):
            raise TimeoutError(f"thread ranks still running after {timeout} s: {[r for r, t in enumerate(ts) if t.is_alive()]}")
        for e in err:
            if e is not None:
                raise e
        return out

    def close(self):
        for c in self.comms:
            self.lib.simrank_comm_destroy(c)
        self.comms = []

class ShardBiPlans:
    """This process's share of a SHARDED two-matrix fit behind the C ABI (simrank_shardbiplan_*: the loops of
    SimRank.py:288-302, :410-424, :478-492 with S1 and S2 split by column block over ``world`` ranks): one pair of plans
    over an RCCL communicator, or all ``world`` pairs of an in-process group (``comm=None``).  Collective calls."""

    def __init__(self, ops, csr12: CSR, rowscale1, rowscale2, world: int = 1, comm=None, c1: float = 0.8, c2: float = 0.8,
                 evidence: bool = False, apriori1=None, apriori2=None, lbd1: float = 0.0, lbd2: float = 0.0,
                 reorder: bool = True, strict_reference: bool = False, leg2_form: int = -1, stages: int = 0,
                 wire_fp16: bool = False):
        self.ops, self.n1, self.n2 = ops, csr12.n_rows, csr12.n_cols
        lib = ops.lib
        rowptr = np.ascontiguousarray(csr12.rowptr, dtype=np.int32)
        col = np.ascontiguousarray(csr12.col, dtype=np.int32)
        rs1 = np.ascontiguousarray(rowscale1, dtype=np.float32)
        rs2 = np.ascontiguousarray(rowscale2, dtype=np.float32)
        a1 = None if apriori1 is None else np.ascontiguousarray(apriori1, dtype=np.float32)
        a2 = None if apriori2 is None else np.ascontiguousarray(apriori2, dtype=np.float32)
        opt = BiPlanOptions(c1=c1, c2=c2, lbd1=lbd1, lbd2=lbd2,
                            apriori1=None if a1 is None else a1.ctypes.data, ld_apriori1=0 if a1 is None else a1.shape[1],
                            apriori2=None if a2 is None else a2.ctypes.data, ld_apriori2=0 if a2 is None else a2.shape[1],
                            evidence=int(evidence), reorder=int(reorder), strict_reference=int(strict_reference))
        self.own_comms = comm is None
        if comm is None:
            arr = (C.c_void_p * world)()
            check(lib.simrank_comm_local_group(int(world), arr), "simrank_comm_local_group")
            self.comms = [C.c_void_p(arr[r]) for r in range(world)]
        else:
            self.comms = [comm]
        self.pairs = []
        with HipOps._knob_lock:
            for c in self.comms:
                h = C.c_void_p()
                check(lib.simrank_shardbiplan_create(csr12.n_rows, csr12.n_cols, col.size, rowptr.ctypes.data,
                                                     col.ctypes.data if col.size else None, rs1.ctypes.data, rs2.ctypes.data,
                                                     C.byref(opt), int(leg2_form), int(stages), int(wire_fp16), c, ops.stream,
                                                     C.byref(h)), "simrank_shardbiplan_create")
                self.pairs.append(h)
        self._arr = (C.c_void_p * len(self.pairs))(*[h.value for h in self.pairs])
        self._sides = {}
        for group in (1, 2):
            hs = []
            for h in self.pairs:
                sp = C.c_void_p()
                check(lib.simrank_shardbiplan_side(h, group, C.byref(sp)), "simrank_shardbiplan_side")
                hs.append(sp)
            self._sides[group] = (C.c_void_p * len(hs))(*[x.value for x in hs])

    def reset(self):
        check(self.ops.lib.simrank_shardbiplan_reset(self._arr, len(self.pairs)), "simrank_shardbiplan_reset")

    def step(self, eps: float, exact_count: bool = True):
        c1, c2 = C.c_int64(0), C.c_int64(0)
        check(self.ops.lib.simrank_shardbiplan_step(self._arr, len(self.pairs), float(eps), int(exact_count), C.byref(c1),
                                                    C.byref(c2)), "simrank_shardbiplan_step")
        return c1.value, c2.value

    def run(self, iterations: int, eps: float):
        """-> (loop bodies applied, loop index at which the convergence test passed or None); the same on every rank."""
        done, conv = C.c_int32(0), C.c_int32(-1)
        check(self.ops.lib.simrank_shardbiplan_run(self._arr, len(self.pairs), int(iterations), float(eps), C.byref(done),
                                                   C.byref(conv)), "simrank_shardbiplan_run")
        return done.value, (None if conv.value < 0 else conv.value)

    def side_info(self, group: int, i: int = 0) -> dict:
        n, lo, hi = C.c_int64(), C.c_int64(), C.c_int64()
        half, stages, updates = C.c_int32(), C.c_int32(), C.c_int32()
        check(self.ops.lib.simrank_shardplan_info(C.c_void_p(self._sides[group][i]), C.byref(n), C.byref(lo), C.byref(hi),
                                                  C.byref(half), C.byref(stages), C.byref(updates)), "simrank_shardplan_info")
        return dict(n=n.value, col_lo=lo.value, col_hi=hi.value, half_form=bool(half.value), stages=stages.value,
                    updates=updates.value)

    def result(self, group: int, root: int = 0, i_am_root: bool = True):
        """Group 1 | 2's whole matrix (float64, caller's order) on rank ``root``; None elsewhere.  Collective."""
        n = self.n1 if group == 1 else self.n2
        out = hostpool.empty_f64(n, n) if i_am_root else None
        check(self.ops.lib.simrank_shardplan_result_f64(self._sides[group], len(self.pairs), int(root),
                                                        out.ctypes.data if out is not None else None, n),
              "simrank_shardplan_result_f64")
        return out

    def topk(self, group: int, k: int, exclude_diag: bool = True, root: int = 0, i_am_root: bool = True):
        n = self.n1 if group == 1 else self.n2
        idx = np.empty((n, k), dtype=np.int32) if i_am_root else None
        val = np.empty((n, k), dtype=np.float32) if i_am_root else None
        check(self.ops.lib.simrank_shardplan_topk(self._sides[group], len(self.pairs), int(root), int(k), int(exclude_diag),
                                                  idx.ctypes.data if i_am_root else None,
                                                  val.ctypes.data if i_am_root else None), "simrank_shardplan_topk")
        return idx, val

    def free(self):
        lib = self.ops.lib
        for h in self.pairs:
            lib.simrank_shardbiplan_destroy(h)
        self.pairs = []
        if self.own_comms:
            for c in self.comms:
                lib.simrank_comm_destroy(c)
        self.comms = []

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class BiPlanOptions(C.Structure):              # struct simrank_biplan_options
    _fields_ = [("c1", C.c_float), ("c2", C.c_float), ("lbd1", C.c_float), ("lbd2", C.c_float),
                ("apriori1", C.c_void_p), ("ld_apriori1", C.c_int64), ("apriori2", C.c_void_p),
                ("ld_apriori2", C.c_int64), ("evidence", C.c_int32), ("reorder", C.c_int32),
                ("strict_reference", C.c_int32)]


class BiPlan:
    """The bipartite loop on one GPU behind the C ABI (simrank_biplan_*: SimRank.py:288-302, :410-424,
    :478-492): the group-1 CSR (columns = group-2 ids) and both groups' row scales in, (S1, S2) out."""

    def __init__(self, ops, csr12: CSR, rowscale1, rowscale2, c1: float = 0.8, c2: float = 0.8,
                 evidence: bool = False, apriori1=None, apriori2=None, lbd1: float = 0.0, lbd2: float = 0.0,
                 reorder: bool = True, strict_reference: bool = False):
        self.ops = ops
        rowptr = np.ascontiguousarray(csr12.rowptr, dtype=np.int32)
        col = np.ascontiguousarray(csr12.col, dtype=np.int32)
        rs1 = np.ascontiguousarray(rowscale1, dtype=np.float32)
        rs2 = np.ascontiguousarray(rowscale2, dtype=np.float32)
        assert rs1.size == csr12.n_rows and rs2.size == csr12.n_cols
        a1 = None if apriori1 is None else np.ascontiguousarray(apriori1, dtype=np.float32)
        a2 = None if apriori2 is None else np.ascontiguousarray(apriori2, dtype=np.float32)
        opt = BiPlanOptions(c1=c1, c2=c2, lbd1=lbd1, lbd2=lbd2,
                            apriori1=None if a1 is None else a1.ctypes.data, ld_apriori1=0 if a1 is None else a1.shape[1],
                            apriori2=None if a2 is None else a2.ctypes.data, ld_apriori2=0 if a2 is None else a2.shape[1],
                            evidence=int(evidence), reorder=int(reorder), strict_reference=int(strict_reference))
        h = C.c_void_p()
        with HipOps._knob_lock:
            check(ops.lib.simrank_biplan_create(csr12.n_rows, csr12.n_cols, col.size, rowptr.ctypes.data,
                                                col.ctypes.data if col.size else None, rs1.ctypes.data, rs2.ctypes.data,
                                                C.byref(opt), ops.stream, C.byref(h)), "simrank_biplan_create")
        self.handle, self.n1, self.n2 = h, csr12.n_rows, csr12.n_cols

    def run(self, iterations: int, eps: float, on_iteration=None, on_converged=None):
        """-> (loop bodies applied, loop index at which the convergence test passed or None); hooks as ``Plan.run``."""
        done, conv = C.c_int32(0), C.c_int32(-1)
        if on_iteration is None and on_converged is None:
            check(self.ops.lib.simrank_biplan_run(self.handle, int(iterations), float(eps), C.byref(done), C.byref(conv)),
                  "simrank_biplan_run")
        else:
            cb, raised = _progress_callback(on_iteration, on_converged)
            check(self.ops.lib.simrank_biplan_run_cb(self.handle, int(iterations), float(eps), cb, None, C.byref(done),
                                                     C.byref(conv)), "simrank_biplan_run_cb")
            if raised:
                raise raised[0]
        return done.value, (None if conv.value < 0 else conv.value)

    def result_group(self, group: int) -> np.ndarray:
        n = self.n1 if group == 1 else self.n2
        m = hostpool.empty_f64(n, n)
        check(self.ops.lib.simrank_biplan_result_f64(self.handle, group, m.ctypes.data, n), "simrank_biplan_result_f64")
        return m

    def rows(self, group: int, rows) -> np.ndarray:
        """float32 [len(rows), n_group]: those rows of group 1 | 2's current similarity matrix, caller's order on both axes."""
        n = self.n1 if group == 1 else self.n2
        ids = np.ascontiguousarray(rows, dtype=np.int32)
        out = np.empty((ids.size, n), dtype=np.float32)
        check(self.ops.lib.simrank_biplan_rows_f32(self.handle, int(group), ids.ctypes.data, int(ids.size), out.ctypes.data, n),
              "simrank_biplan_rows_f32")
        return out

    def topk(self, group: int, k: int, exclude_diag: bool = True):
        """(ids int32 [n, k], values float32 [n, k]) of group 1 | 2, caller's ids."""
        n = self.n1 if group == 1 else self.n2
        idx = np.empty((n, k), dtype=np.int32)
        val = np.empty((n, k), dtype=np.float32)
        check(self.ops.lib.simrank_biplan_topk(self.handle, int(group), int(k), int(exclude_diag), idx.ctypes.data,
                                               val.ctypes.data), "simrank_biplan_topk")
        return idx, val

    def evidence_counts(self, group: int) -> np.ndarray:
        """uint8 [n, n] counts gating group 1 | 2's update, caller's order."""
        n = self.n1 if group == 1 else self.n2
        out = np.empty((n, n), dtype=np.uint8)
        check(self.ops.lib.simrank_biplan_evidence_u8(self.handle, int(group), out.ctypes.data, n),
              "simrank_biplan_evidence_u8")
        return out

    def trim(self):
        if self.handle:
            check(self.ops.lib.simrank_biplan_trim(self.handle), "simrank_biplan_trim")

    def reset(self):
        check(self.ops.lib.simrank_biplan_reset(self.handle), "simrank_biplan_reset")

    def step(self, eps: float, exact_count: bool = True):
        c1, c2 = C.c_int64(0), C.c_int64(0)
        check(self.ops.lib.simrank_biplan_step(self.handle, float(eps), int(exact_count), C.byref(c1), C.byref(c2)),
              "simrank_biplan_step")
        return c1.value, c2.value

    def result(self):
        return self.result_group(1), self.result_group(2)

    def free(self):
        if self.handle:
            self.ops.lib.simrank_biplan_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


# rows appended to every panel of a panel-blocked matrix (panels are then not a power of two apart); a measurement knob
BLOCK_PAD_ROWS = int(os.environ.get("SIMRANK_BLOCK_PAD_ROWS", "8")) // 8 * 8 or 8


class HipOps:
    """Kernel launcher for one device.  ``pitch_pad``: extra elements added to the leading
    dimension of matrices it allocates when that dimension is a large power of two (keeps
    a panel's row segments from landing on one L2/HBM channel)."""

    name = "hip"
    supports_blocked = True      # the solver may keep its matrices panel-blocked (Matrix.blocked)
    supports_half_storage = True      # fp16-held matrices on 64-column panels (half.hip)
    HALF_SCALE = 16384.0              # what the solver's fp16 matrices are scaled by (include/simrank_hip.h, SCALE)
    supports_shard_symmetric = True   # sharded leg 2 in its half form (spmm_shard / shard_unpack)
    supports_counting_graph = True    # graph(..., counting=): evidence counts queued during graph creation

    def __init__(self, device: int | None = None, stream: int | None = None):
        self.lib = _lib.load()
        n = _lib.device_count()
        if n <= 0:
            raise SimRankHipError(
                "no HIP device visible: the SimRank engine needs an MI355X (gfx950); "
                "there is no CPU fallback")
        self.device = 0 if device is None else int(device)
        check(self.lib.simrank_set_device(self.device), "simrank_set_device")
        self._own_stream = stream is None
        if stream is None:
            s = C.c_void_p()
            check(self.lib.simrank_stream_create(C.byref(s)), "simrank_stream_create")
            self.stream = s
        else:
            self.stream = C.c_void_p(stream)
        self._counter = self._malloc(8 * CHANGED_SLOTS)
        self._counter_set = None          # pinned slots for fetch_changed / wait_changed: this engine's own
        self.pitch_pad = int(os.environ.get("SIMRANK_PITCH_PAD", "96"))

    # ---- memory ----
    # hipMalloc maps a 17 GiB matrix in 0.4-0.5 s (profiles/r03_setup_before_pool.log: 1.6-2.0 s of a config-5
    # solver set-up, against 0.12 s for its four iterations), so blocks of at least 64 MiB are pooled INSIDE the
    # library (csrc/api.hip: simrank_malloc / simrank_free; best fit within 1/8, least recently freed first out,
    # an allocation that fails empties the pool and is tried again; the C-level plans use the same pool).
    # Pooled memory is invisible to other allocators of the process (torch's among them): trim_pool() hands
    # it back; SIMRANK_POOL_GIB bounds what may rest per device (default 56: one config-5 plan).
    def _malloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        check(self.lib.simrank_malloc(C.byref(p), max(16, int(nbytes))), "simrank_malloc")
        return p.value

    def _free(self, ptr: int):
        self.lib.simrank_free(C.c_void_p(ptr))

    @classmethod
    def trim_pool(cls, device: int | None = None):
        """Give the library's cached device blocks back to the driver (all devices, or one), and the host frames at rest
        (hostpool.py) back to the system."""
        _lib.load().simrank_pool_trim(-1 if device is None else int(device))
        hostpool.trim()

    @classmethod
    def pool_stats(cls, device: int | None = None):
        """(bytes at rest, blocks at rest, limit in bytes) of the library's device block pool."""
        a, b, c = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        _lib.load().simrank_pool_stats(-1 if device is None else int(device), C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value

    def pitch(self, cols: int, dtype) -> int:
        unit = 16 // np.dtype(dtype).itemsize          # 16-byte rows for the vector kernels
        ld = -(-cols // unit) * unit
        if ld >= 4096 and (ld & (ld - 1)) == 0:
            ld += self.pitch_pad
        return ld

    def matrix(self, rows, cols, dtype=np.float32, ld=None, external=None, blocked=False) -> Matrix:
        return Matrix(self, rows, cols, dtype, ld, external, blocked)

    def exchange_buffer(self, n_floats: int):
        """Flat float32 buffer usable by torch.distributed collectives on this device."""
        import torch
        return torch.empty(max(1, n_floats), dtype=torch.float32, device=f"cuda:{self.device}")

    def exchange_buffer_h(self, n: int):
        """Flat float16 buffer: the wire format of an exchange (driver: exchange_precision="fp16")."""
        import torch
        return torch.empty(max(8, n), dtype=torch.float16, device=f"cuda:{self.device}")

    WIRE_SCALE = 16384.0        # what values on the wire are multiplied by (as fp16-held matrices: similarities of a
                                # large sparse graph lie mostly below fp16's normal range)

    def narrow_t(self, src_t, dst_t, off: int, n: int):
        """dst_t[off:off+n] (float16 tensor) = src_t[off:off+n] (float32 tensor) x WIRE_SCALE, on the engine's stream."""
        if n:
            check(self.lib.simrank_narrow_h16(C.c_void_p(src_t.data_ptr() + 4 * off), C.c_void_p(dst_t.data_ptr() + 2 * off),
                                              int(n), C.c_float(self.WIRE_SCALE), self.stream), "simrank_narrow_h16")

    def widen_t(self, src_t, dst_t, off: int, n: int):
        """dst_t[off:off+n] (float32) = src_t[off:off+n] (float16) / WIRE_SCALE, on the engine's stream."""
        if n:
            check(self.lib.simrank_widen_h16(C.c_void_p(src_t.data_ptr() + 2 * off), C.c_void_p(dst_t.data_ptr() + 4 * off),
                                             int(n), C.c_float(self.WIRE_SCALE), self.stream), "simrank_widen_h16")

    def round_trip_h16(self, m: "Matrix", n: int):
        """The first n floats of ``m`` as they would arrive over an fp16 wire (virtual ranks: no wire to save bytes on)."""
        if n:
            tmp = self._malloc(2 * n + 16)
            check(self.lib.simrank_narrow_h16(C.c_void_p(m.ptr), C.c_void_p(tmp), int(n), C.c_float(self.WIRE_SCALE),
                                              self.stream), "simrank_narrow_h16")
            check(self.lib.simrank_widen_h16(C.c_void_p(tmp), C.c_void_p(m.ptr), int(n), C.c_float(self.WIRE_SCALE),
                                             self.stream), "simrank_widen_h16")
            self.lib.simrank_stream_synchronize(self.stream)
            self._free(tmp)

    def copy_bytes(self, dst_ptr: int, src_ptr: int, nbytes: int):
        if nbytes:
            check(self.lib.simrank_memcpy_d2d(C.c_void_p(dst_ptr), C.c_void_p(src_ptr), nbytes,
                                              self.stream), "d2d")

    def upload(self, m: Matrix, host: np.ndarray):
        host = np.asarray(host)
        assert host.shape == (m.rows, m.cols), (host.shape, m.rows, m.cols)
        if m.dtype == np.float16:           # laid out on the host (tests and warm starts: small)
            buf = np.zeros((m.panels, m.rows_pad, 64), dtype=np.float16)
            for pnl in range(m.panels):
                w = min(64, m.cols - 64 * pnl)
                buf[pnl, :m.rows, :w] = host[:, 64 * pnl:64 * pnl + w].astype(np.float32) * np.float32(m.scale)
            check(self.lib.simrank_memcpy_h2d(m.ptr, buf.ctypes.data, buf.nbytes, self.stream),
                  "simrank_memcpy_h2d")
            return
        if m.blocked:                       # through a row-major copy on the device
            tmp = self.matrix(m.rows, m.cols, m.dtype)
            self.upload(tmp, host)
            self.permute(tmp, m)
            self.synchronize()
            tmp.free()
            return
        buf = np.zeros((m.rows, m.ld), dtype=m.dtype)
        buf[:, :m.cols] = host
        check(self.lib.simrank_memcpy_h2d(m.ptr, buf.ctypes.data, buf.nbytes, self.stream),
              "simrank_memcpy_h2d")

    def widen(self, src: Matrix) -> Matrix:
        """float32 panel-blocked copy of a float16 matrix (what the hand-back entries read)."""
        assert src.dtype == np.float16 and src.blocked
        dst = self.matrix(src.rows, src.cols, blocked=True)
        check(self.lib.simrank_widen_blocked_h16(src.ptr, src.rows_pad, dst.ptr, dst.rows_pad, src.rows, src.cols,
                                                 src.scale, self.stream), "simrank_widen_blocked_h16")
        return dst

    def download(self, m: Matrix) -> np.ndarray:
        """Device matrix -> host array of its own dtype."""
        if m.dtype == np.float16:
            wide = self.widen(m)
            out = self.download(wide)       # float32: the values the matrix stands for (stored / scale, exact)
            wide.free()
            return out
        if m.blocked:
            tmp = self.matrix(m.rows, m.cols, m.dtype)
            self.permute(m, tmp)
            out = self.download(tmp)
            tmp.free()
            return out
        buf = np.empty((m.rows, m.ld), dtype=m.dtype)
        check(self.lib.simrank_memcpy_d2h(buf.ctypes.data, m.ptr, buf.nbytes, self.stream),
              "simrank_memcpy_d2h")
        return buf[:, :m.cols] if m.ld == m.cols else np.ascontiguousarray(buf[:, :m.cols])

    def download_rows(self, m: Matrix, rows) -> np.ndarray:
        """Selected rows of a device matrix (partial hand-back: no N x N host copy)."""
        if m.dtype == np.float16:
            wide = self.widen(m)
            out = self.download_rows(wide, rows)
            wide.free()
            return out
        if m.blocked:
            idx = self.index_vector(rows)
            tmp = self.matrix(len(rows), m.cols, m.dtype)
            self.permute(m, tmp, row_idx=idx)
            out = self.download(tmp)
            tmp.free()
            idx.free()
            return out
        out = np.empty((len(rows), m.cols), dtype=m.dtype)
        isz = m.dtype.itemsize
        for i, r in enumerate(rows):
            check(self.lib.simrank_memcpy_d2h(out[i].ctypes.data, C.c_void_p(m.ptr + int(r) * m.ld * isz),
                                              m.cols * isz, self.stream), "simrank_memcpy_d2h")
        return out

    def topk_rows(self, m: Matrix, k: int, col0: int = 0, exclude_diag: bool = True,
                  col_ids: Matrix | None = None):
        """(column ids int32 [rows, k], values float32 [rows, k]) of the k largest entries of
        every row, computed on the device; -1 / 0 where a row has fewer.  The id of block
        column c is ``col0 + c`` unless ``col_ids`` (an ``index_vector``) names them."""
        k = int(k)
        idx = self.matrix(m.rows, k, np.int32, ld=k)
        val = self.matrix(m.rows, k, np.float32, ld=k)
        ids = col_ids.ptr if col_ids is not None else None
        if m.blocked:
            check(self.lib.simrank_topk_rows_blocked(m.ptr, m.rows_pad, m.rows, m.cols, int(col0), ids, k,
                                                     1 if exclude_diag else 0, idx.ptr, val.ptr,
                                                     self.stream), "simrank_topk_rows_blocked")
        else:
            check(self.lib.simrank_topk_rows_ids(m.ptr, m.ld, m.rows, m.cols, int(col0), ids, k,
                                                 1 if exclude_diag else 0, idx.ptr, val.ptr,
                                                 self.stream), "simrank_topk_rows_ids")
        out = self.download(idx), self.download(val)
        idx.free()
        val.free()
        return out

    def index_vector(self, values) -> Matrix:
        """int32 index list on the device (for ``permute`` and ``topk_rows``)."""
        v = np.ascontiguousarray(values, dtype=np.int32).reshape(1, -1)
        m = self.matrix(1, v.shape[1], np.int32, ld=v.shape[1])
        self.upload(m, v)
        return m

    def permute(self, src: Matrix, dst: Matrix, row_idx: Matrix | None = None,
                col_idx: Matrix | None = None):
        """dst[i, j] = src[row_idx[i], col_idx[j]] (None = identity); float32 / int32 or uint8;
        either side may be panel-blocked (this is also how a matrix changes layout)."""
        assert src.dtype == dst.dtype and src.dtype.itemsize in (1, 4)
        check(self.lib.simrank_permute_layout(src.ptr, src.ld, src.rows_pad, dst.ptr, dst.ld, dst.rows_pad,
                                              dst.rows, dst.cols,
                                              row_idx.ptr if row_idx is not None else None,
                                              col_idx.ptr if col_idx is not None else None,
                                              src.dtype.itemsize, self.stream), "simrank_permute_layout")

    def download_f64(self, m: Matrix, out: np.ndarray | None = None) -> np.ndarray:
        """float32 device matrix -> float64 host array (pinned, pipelined staging)."""
        if out is None:
            out = hostpool.empty_f64(m.rows, m.cols)
        assert out.dtype == np.float64 and out.flags.c_contiguous and not m.blocked
        check(self.lib.simrank_download_f64(out.ctypes.data, out.shape[1], m.ptr, m.ld, m.rows,
                                            m.cols, self.stream), "simrank_download_f64")
        return out

    def handback_f64(self, m: Matrix, idx: Matrix | None = None, out: np.ndarray | None = None,
                     symmetric: bool = False) -> np.ndarray:
        """Square float32 device matrix (either layout) -> float64 host array with rows and columns taken in the order
        ``idx`` (an ``index_vector``; None = as stored): out[i][j] = m[idx[i]][idx[j]], pipelined band by band
        (csrc/handback.hip).  ``symmetric``: only the upper triangle crosses PCIe when the matrix proves mirror-equal
        outside its 32 x 32 diagonal blocks (same bits; pays only where the host side has CPUs to spare)."""
        assert m.rows == m.cols and m.dtype == np.float32
        if out is None:
            out = hostpool.empty_f64(m.rows, m.cols)
        assert out.dtype == np.float64 and out.flags.c_contiguous and out.shape == (m.rows, m.cols)
        check(self.lib.simrank_handback_f64(out.ctypes.data, out.shape[1], m.ptr, m.ld, m.rows_pad if m.blocked else 0,
                                            m.rows, idx.ptr if idx is not None else None, int(bool(symmetric)), self.stream),
              "simrank_handback_f64")
        return out

    def copy(self, dst: Matrix, src: Matrix):
        assert dst.nbytes == src.nbytes
        check(self.lib.simrank_memcpy_d2d(dst.ptr, src.ptr, src.nbytes, self.stream), "d2d")

    def synchronize(self):
        check(self.lib.simrank_stream_synchronize(self.stream), "stream_synchronize")

    def torch_stream(self):
        """The engine's stream as a torch stream: a collective issued under
        ``torch.cuda.stream(ops.torch_stream())`` is ordered after the kernels already queued on
        the engine's stream, and ``work.wait()`` makes that stream (not the host) wait for it."""
        import torch
        if getattr(self, "_torch_stream", None) is None:
            self._torch_stream = torch.cuda.ExternalStream(self.stream.value, device=self.device)
        return self._torch_stream

    def counter_tensor(self):
        """The convergence counters as a torch int64 tensor (allocated by torch from then on), so
        that a distributed world can reduce them on the device."""
        import torch
        if getattr(self, "_counter_t", None) is None:
            t = torch.zeros(CHANGED_SLOTS, dtype=torch.int64, device=f"cuda:{self.device}")
            torch.cuda.synchronize(self.device)
            if self._counter:
                self._free(self._counter)
            self._counter_t, self._counter = t, t.data_ptr()
        return self._counter_t

    def collective_done(self):
        """Make the engine's stream wait for a torch.distributed collective: the collective
        was enqueued on torch's current stream, the kernels run on the engine's own."""
        import torch
        torch.cuda.current_stream(self.device).synchronize()

    # ---- graph + kernels ----
    _knob_lock = threading.RLock()    # graph creation with per-graph knobs: set, create, restore as one step

    def graph(self, csr: CSR, rowscale=None, dense_terms: int = 3, knobs: dict | None = None, counting=None) -> Graph:
        """``dense_terms``: operand terms of the matrix-core part for this graph (3 exact, 1 = one fp16 term).
        ``knobs``: tuning values for THIS graph only (a graph keeps the knobs it was created with): set, create
        and restore under a lock that every graph creation of this module takes."""
        with HipOps._knob_lock:
            saved = {k: self.get_tuning(k) for k in (knobs or {})}
            try:
                if knobs:
                    self.set_tuning(**knobs)
                g = Graph(self, csr, rowscale, counting)
            finally:
                if saved:
                    self.set_tuning(**saved)
        if dense_terms != 3:
            check(self.lib.simrank_graph_set_dense_terms(g.handle, int(dense_terms)), "simrank_graph_set_dense_terms")
        return g

    def fill_identity(self, S: Matrix, col0: int):
        if S.rows and S.cols and S.dtype == np.float16:
            check(self.lib.simrank_fill_identity_blocked_h16(S.ptr, S.rows, S.cols, S.rows_pad, col0, S.scale,
                                                             self.stream), "simrank_fill_identity_blocked_h16")
        elif S.rows and S.cols and S.blocked:
            check(self.lib.simrank_fill_identity_blocked(S.ptr, S.rows, S.cols, S.rows_pad, col0,
                                                         self.stream), "simrank_fill_identity_blocked")
        elif S.rows and S.cols:
            check(self.lib.simrank_fill_identity(S.ptr, S.rows, S.cols, S.ld, col0, self.stream),
                  "simrank_fill_identity")

    def _epilogue(self, coef, evidence=None, apriori=None, lbd=0.0, previous=None, eps=0.0,
                  diag_col0=0, set_diag=True, symmetric=False, restrict_support=False,
                  count_any=False) -> Epilogue:
        ep = Epilogue()
        ep.coef = float(coef)
        ep.lbd = float(lbd)
        if evidence is not None:
            ep.evidence, ep.ld_evidence = evidence.ptr, evidence.ld
        if apriori is not None:
            ep.apriori, ep.ld_apriori = apriori.ptr, apriori.ld
        if previous is not None:
            ep.previous, ep.ld_previous = previous.ptr, previous.ld
            ep.n_changed = self._counter
        ep.eps = float(eps)
        ep.diag_col0 = int(diag_col0)
        ep.set_diag = 1 if set_diag else 0
        ep.symmetric = 1 if symmetric else 0
        ep.restrict_support = 1 if (restrict_support and evidence is not None) else 0
        ep.count_any = 1 if count_any else 0
        return ep

    def spmm(self, g: Graph, X: Matrix, Y: Matrix, n_cols: int | None = None,
             transpose_out: bool = False, t_block: int = 0, t_pad: int = 0,
             epilogue: dict | None = None, x_col0: int = 0, y_offset: int = 0):
        """Y = diag(rowscale).A.X (+ fused epilogue); see simrank_spmm.  ``x_col0`` selects a
        column sub-block of X (first column), ``y_offset`` an element offset into Y."""
        n_cols = X.cols if n_cols is None else n_cols
        ep = self._epilogue(**epilogue) if epilogue is not None else None
        if X.dtype == np.float16:
            # fp16 storage: both legs on 64-column panels (half.hip); evidence and prior keep their f32-era layout
            assert Y.dtype == np.float16 and not x_col0 and not y_offset and n_cols == X.cols
            assert X.scale == Y.scale
            aux_pad = 0
            for name in ("evidence", "apriori"):
                m = (epilogue or {}).get(name)
                assert m is None or m.blocked, name
                aux_pad = m.rows_pad if m is not None else aux_pad
            prev = (epilogue or {}).get("previous")
            assert prev is None or (prev.dtype == np.float16 and prev.rows_pad == Y.rows_pad and prev.scale == Y.scale)
            check(self.lib.simrank_spmm_blocked_h16(g.handle, X.ptr, X.rows_pad, n_cols, Y.ptr, Y.rows_pad,
                                                    1 if transpose_out else 0,
                                                    C.byref(ep) if ep is not None else None, aux_pad, Y.scale,
                                                    self.stream), "simrank_spmm_blocked_h16")
            return
        if X.blocked:
            assert Y.blocked and not x_col0 and not y_offset and n_cols == X.cols
            for name in ("evidence", "apriori", "previous"):
                m = (epilogue or {}).get(name)
                assert m is None or (m.blocked and m.rows_pad == Y.rows_pad), name
            check(self.lib.simrank_spmm_blocked(g.handle, X.ptr, X.rows_pad, n_cols, Y.ptr, Y.rows_pad,
                                                1 if transpose_out else 0,
                                                C.byref(ep) if ep is not None else None, self.stream),
                  "simrank_spmm_blocked")
            return
        check(self.lib.simrank_spmm(g.handle, X.ptr + 4 * int(x_col0), X.ld, n_cols,
                                    Y.ptr + 4 * int(y_offset), Y.ld,
                                    1 if transpose_out else 0, int(t_block), int(t_pad),
                                    C.byref(ep) if ep is not None else None, self.stream),
              "simrank_spmm")

    def spmm_shard(self, g: Graph, X: Matrix, Y: Matrix, epilogue: dict, rank: int, world: int,
                   send: Matrix, chunk_floats: int):
        """Leg 2 of one rank of a sharded symmetric update in its half form: tiles (h, i) x (rank, j)
        with i <= j into Y, the transposed tiles i < j into Y (h == rank) or ``send`` (packed, one
        chunk per destination rank); see simrank_spmm_shard."""
        ep = self._epilogue(**epilogue)
        check(self.lib.simrank_spmm_shard(g.handle, X.ptr, X.ld, Y.ptr, Y.ld, C.byref(ep), int(rank),
                                          int(world), send.ptr, int(chunk_floats), self.stream),
              "simrank_spmm_shard")

    def shard_unpack(self, Y: Matrix, recv: Matrix, chunk_floats: int, rank: int, world: int,
                     n_rows: int):
        """The mirrored tiles received from the other ranks go to their places in Y."""
        check(self.lib.simrank_shard_unpack(Y.ptr, Y.ld, recv.ptr, int(chunk_floats), int(rank),
                                            int(world), int(n_rows), self.stream),
              "simrank_shard_unpack")

    def spmm_shard_stage(self, g: Graph, X: Matrix, Y: Matrix, epilogue: dict, rank: int, world: int,
                         send: Matrix, send_off: int, stage_chunk: int, tile_lo: int, tile_hi: int, zero_counters: bool):
        """One stage of ``spmm_shard``: column tiles [tile_lo, tile_hi); their packed mirrored tiles go to
        ``send`` from float offset ``send_off`` on, ``stage_chunk`` floats per destination rank."""
        ep = self._epilogue(**epilogue)
        check(self.lib.simrank_spmm_shard_stage(g.handle, X.ptr, X.ld, Y.ptr, Y.ld, C.byref(ep), int(rank), int(world),
                                                send.ptr + 4 * int(send_off), int(stage_chunk), int(tile_lo),
                                                int(tile_hi), int(bool(zero_counters)), self.stream),
              "simrank_spmm_shard_stage")

    def shard_unpack_stage(self, Y: Matrix, recv: Matrix, recv_off: int, stage_chunk: int, rank: int, world: int,
                           n_rows: int, tile_lo: int, tile_hi: int):
        check(self.lib.simrank_shard_unpack_stage(Y.ptr, Y.ld, recv.ptr + 4 * int(recv_off), int(stage_chunk), int(rank),
                                                  int(world), int(n_rows), int(tile_lo), int(tile_hi), self.stream),
              "simrank_shard_unpack_stage")

    def epilogue_apply(self, Q: Matrix, Y: Matrix, n_rows: int, n_cols: int, epilogue: dict):
        """Y = epilogue(Q) element-wise; see simrank_epilogue_apply."""
        ep = self._epilogue(**epilogue)
        if Q.blocked:
            assert Y.blocked and Y.rows_pad == Q.rows_pad
            check(self.lib.simrank_epilogue_apply_blocked(Q.ptr, Y.ptr, n_rows, n_cols, Q.rows_pad,
                                                          C.byref(ep), self.stream),
                  "simrank_epilogue_apply_blocked")
            return
        check(self.lib.simrank_epilogue_apply(Q.ptr, Q.ld, Y.ptr, Y.ld, n_rows, n_cols,
                                              C.byref(ep), self.stream), "simrank_epilogue_apply")

    def gemm_nt(self, A: Matrix, B: Matrix, Cm: Matrix, M: int, N: int, K: int,
                epilogue: dict | None = None):
        ep = self._epilogue(**epilogue) if epilogue is not None else None
        check(self.lib.simrank_gemm_nt(M, N, K, A.ptr, A.ld, B.ptr, B.ld, Cm.ptr, Cm.ld,
                                       C.byref(ep) if ep is not None else None, self.stream),
              "simrank_gemm_nt")

    def densify(self, g: Graph, Wd: Matrix):
        check(self.lib.simrank_graph_densify(g.handle, Wd.ptr, Wd.ld, self.stream),
              "simrank_graph_densify")

    def dense_stats(self, g: Graph):
        """(row blocks with a dense set, total size of the sets, entries covered) of the
        block-dense part of a graph (simrank_graph_dense_stats)."""
        a, b, c = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        check(self.lib.simrank_graph_dense_stats(g.handle, C.byref(a), C.byref(b), C.byref(c)),
              "simrank_graph_dense_stats")
        return a.value, b.value, c.value

    def fused_stats(self, g: Graph):
        """(16-column matrix-core steps per panel, entries on the matrix cores, entries gathered) of the
        one-launch leg 1 of a graph (simrank_graph_fused_stats); (0, 0, nnz) when tuning "fuse" was 0."""
        a, b, c = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        check(self.lib.simrank_graph_fused_stats(g.handle, C.byref(a), C.byref(b), C.byref(c)),
              "simrank_graph_fused_stats")
        return a.value, b.value, c.value

    def dense_part(self, g: Graph, X: Matrix, n_cols: int | None = None):
        """The matrix-core part of ``spmm`` alone (measurement; see simrank_dense_part)."""
        check(self.lib.simrank_dense_part(g.handle, X.ptr, -X.rows_pad if X.blocked else X.ld,
                                          X.cols if n_cols is None else n_cols, self.stream),
              "simrank_dense_part")

    def evidence_counts(self, g: Graph, col0: int, out: Matrix):
        if out.cols and out.blocked:
            check(self.lib.simrank_evidence_counts_blocked(g.handle, col0, out.cols, out.ptr, out.rows_pad,
                                                           self.stream), "simrank_evidence_counts_blocked")
        elif out.cols:
            check(self.lib.simrank_evidence_counts(g.handle, col0, out.cols, out.ptr, out.ld,
                                                   self.stream), "simrank_evidence_counts")

    def evidence_live_fraction(self, counts: Matrix) -> float:
        """Fraction of the aligned 32-column segments of a count block that hold a nonzero count:
        what a support-restricted SimRank++ update (epilogue ``restrict_support``) still gathers."""
        if not counts.cols:
            return 1.0
        live, total = C.c_int64(0), C.c_int64(0)
        check(self.lib.simrank_evidence_live_segments(counts.ptr, counts.ld, counts.rows_pad, counts.rows,
                                                      counts.cols, C.byref(live), C.byref(total), self.stream),
              "simrank_evidence_live_segments")
        return live.value / max(1, total.value)

    def read_changed(self) -> int:
        """Value of the convergence counter of the last epilogue with ``previous``."""
        total = C.c_ulonglong(0)
        check(self.lib.simrank_read_counters(self._counter, CHANGED_SLOTS, C.byref(total),
                                             self.stream), "simrank_read_counters")
        return int(total.value)

    def fetch_changed(self, slot: int):
        """Queue the read-back of the convergence counter into pinned slot ``slot`` (0..3); the stream goes on."""
        if self._counter_set is None:
            h = C.c_void_p()
            check(self.lib.simrank_counters_create(C.byref(h)), "simrank_counters_create")
            self._counter_set = h
        check(self.lib.simrank_counters_fetch(self._counter_set, self._counter, CHANGED_SLOTS, int(slot), self.stream),
              "simrank_counters_fetch")

    def wait_changed(self, slot: int) -> int:
        """The counter value a ``fetch_changed(slot)`` read: waits for that copy only."""
        total = C.c_ulonglong(0)
        check(self.lib.simrank_counters_wait(self._counter_set, int(slot), C.byref(total)), "simrank_counters_wait")
        return int(total.value)

    # ---- timing (HIP events on the engine's own stream) ----
    def event(self) -> int:
        e = C.c_void_p()
        check(self.lib.simrank_event_create(C.byref(e)), "event_create")
        return e.value

    def record(self, ev: int):
        check(self.lib.simrank_event_record(C.c_void_p(ev), self.stream), "event_record")

    def event_synchronize(self, ev: int):
        check(self.lib.simrank_event_synchronize(C.c_void_p(ev)), "event_synchronize")

    def event_destroy(self, ev: int):
        self.lib.simrank_event_destroy(C.c_void_p(ev))

    def elapsed_ms(self, start: int, stop: int) -> float:
        ms = C.c_float(0)
        check(self.lib.simrank_event_elapsed_ms(C.c_void_p(start), C.c_void_p(stop),
                                                C.byref(ms)), "event_elapsed")
        return float(ms.value)

    def close(self):
        """Release the stream and the counter (matrices and graphs free themselves)."""
        if getattr(self, "_counter_set", None) is not None:
            self.lib.simrank_counters_destroy(self._counter_set)
            self._counter_set = None
        if getattr(self, "_counter", 0):
            if getattr(self, "_counter_t", None) is None:
                self._free(self._counter)
            self._counter, self._counter_t = 0, None
        if getattr(self, "_own_stream", False) and self.stream:
            self.lib.simrank_stream_destroy(self.stream)
            self.stream = None
            self._own_stream = False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def get_tuning(self, key: str) -> int:
        v = C.c_int64(0)
        check(self.lib.simrank_get_tuning(key.encode(), C.byref(v)), f"get_tuning({key})")
        return v.value

    def set_tuning(self, **kw):
        for k, v in kw.items():
            check(self.lib.simrank_set_tuning(k.encode(), int(v)), f"set_tuning({k})")

    def device_info(self) -> dict:
        name = C.create_string_buffer(256)
        arch = C.create_string_buffer(64)
        mem = C.c_int64(0)
        cus = C.c_int(0)
        check(self.lib.simrank_device_info(self.device, name, 256, C.byref(mem), C.byref(cus),
                                           arch, 64), "device_info")
        return dict(name=name.value.decode(), arch=arch.value.decode(), total_bytes=mem.value,
                    compute_units=cus.value)
