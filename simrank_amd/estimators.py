"""The reference's class surface, backed by the MI355X engine.

Same class names, ``fit`` signatures (positional order and defaults), return types,
post-fit attributes and console text as ysong1231/SimRank (SimRank/SimRank.py:8, :143,
:305, :365, :427, :457), so ``from simrank_amd import SimRank`` is a drop-in for
``from SimRank import SimRank``.  What differs is *how*: the edge list becomes a CSR graph
(``ingest.py``), the loop runs as HIP kernels behind the C ABI (``cplan.py`` / ``cshard.py`` / ``csrc/``), the similarity
matrix never exists densely on the host until the result is handed back, and the dense
attributes the reference keeps (``Graph``, ``Weight``, ``Evidence`` …) are materialised only
when read.

Extra keyword-only arguments (defaults keep the reference's behaviour):
    mode              "auto" | "sparse": the gather legs ("dense" | "hybrid", BASELINE's literal dense-GEMM leg, left fit()
                      in round 6: they run through the tests' Python choreography and ``bench.py`` only)
    device            HIP device ordinal (default: LOCAL_RANK or 0)
    world             ``driver.LocalWorld`` / ``driver.TorchWorld`` (sharded runs; in a multi-process
                      world the dense result goes to rank 0 only unless TorchWorld(handback="all"))
    top_k             return, instead of the dense matrix, a long-format frame (node, rank,
                      neighbor, similarity) with the k most similar other nodes of every node,
                      selected on the device (no N x N transfer)
    dense_precision   "f32" (default) | "fp16": operand precision of the matrix-core part of the legs;
                      "fp16" is BASELINE.json config 5's reduced-precision dense leg, outside the parity bar
    strict_reference  bipartite classes only; True keeps quirks Q1 (set-order labels on
                      sorted-order data) and Q2 (Evidence_N1 on the group-2 update, a
                      ValueError when n1 != n2); False labels correctly and uses Evidence_N2
"""
from __future__ import annotations

import contextlib
import os
import dataclasses
import threading
import time

import numpy as np
import pandas as pd

from . import ingest
from .driver import LocalWorld, SideSpec
from .progress import announce_converged, update_progress


_DENSE_TERMS = {"f32": 3, "fp16": 1}
_precision_now = threading.local()          # per thread: two threads may fit at different precisions


@contextlib.contextmanager
def _precision(dense_precision, storage_precision="f32"):
    """``fit(dense_precision=...)``: "f32" (default) keeps the matrix-core part of the legs exact
    (operand split into three bf16 terms); "fp16" rounds its operand to one fp16 term — BASELINE.json
    config 5's reduced-precision dense leg: faster, NOT within the 1e-5 parity bar.
    ``fit(storage_precision=...)``: "f32" (default), or "fp16": the similarity matrices and the
    intermediate product are HELD in fp16 (f32 sums and epilogue, one rounding per stored value; gather
    mode, symmetric iterates; one GPU — or, for SimRank / SimRank++ without a prior, the ranks of a
    ``LocalWorld(P)`` / RCCL ``TorchWorld``: ``cshard.py``, the sharded loop behind the C ABI) — half the
    bytes and half the gathered lines per update (and half the bytes on the links), the
    reduced-precision mode that pays on config 5; NOT within the parity bar either.  In that mode the
    convergence test is NOT the reference's `_converged` (SimRank.py:54-77): an element counts as moved only
    when it moved by more than eps + half an fp16 spacing at its stored value, so ``converged_at`` and the
    number of updates are not comparable with the reference's (the loop usually ends one or two updates
    later on SimRank++ matrices, much later on matrices full of values above 1/8)."""
    if dense_precision not in _DENSE_TERMS:
        raise ValueError(f"dense_precision must be one of {sorted(_DENSE_TERMS)}, not {dense_precision!r}")
    if storage_precision not in ("f32", "fp16"):
        raise ValueError(f"storage_precision must be 'f32' or 'fp16', not {storage_precision!r}")
    stack = _precision_now.__dict__.setdefault("stack", [("f32", "f32")])
    stack.append((dense_precision, storage_precision))
    try:
        yield
    finally:
        stack.pop()


PYTHON_SOLVER = None      # tests/pydriver.py installs its kernel-by-kernel Python choreography here (a test double; never set by the product)


def _make_solver(ops_factory, device, world, specs, mode):
    """The solver with the graphs created at the precision asked for: it travels in the specs and is set
    per graph object (simrank_graph_set_dense_terms), not through the process-wide tuning defaults.

    TWO choreographies, both behind the C ABI: one rank -> ``cplan.PlanSolver`` (simrank_plan_* / simrank_biplan_*: every
    class the reference has, fp16-held matrices, asymmetric priors in f32); several ranks — virtual (``LocalWorld(P)``) or
    RCCL (``TorchWorld``) -> ``cshard.CShardSolver`` (simrank_shardplan_* / simrank_shardbiplan_*).  What neither runs is
    refused with the reason.  (``PYTHON_SOLVER``: the tests' double — an injected engine (``_ops_factory``), the GEMM modes
    "dense" / "hybrid", a world with ``loop="python"``, a gloo world — runs ``tests/pydriver.Solver`` when that module is
    loaded, and only then.)"""
    from . import cplan, cshard
    from .driver import TorchWorld
    dense, storage = _precision_now.__dict__.get("stack", [("f32", "f32")])[-1]
    terms = _DENSE_TERMS[dense]
    if terms != 3 or storage != "f32":
        specs = [dataclasses.replace(s, dense_terms=terms, storage=storage) for s in specs]
    if mode not in ("auto", "sparse", "dense", "hybrid"):
        raise ValueError(f"mode must be 'auto' or 'sparse', not {mode!r}")
    loop = getattr(world, "loop", "c")
    gloo = isinstance(world, TorchWorld) and world.dist.get_backend(world.group) != "nccl"
    shards_fp16 = storage == "fp16" and world.size > 1      # (fp16-held shards exist behind the C ABI only: cshard says why not)
    if (ops_factory is not None or mode in ("dense", "hybrid") or loop == "python" or (gloo and loop != "c")) and not shards_fp16:
        if PYTHON_SOLVER is None:
            what = ("an injected engine" if ops_factory is not None else f"mode={mode!r}" if mode in ("dense", "hybrid")
                    else "loop='python'" if loop == "python" else "a gloo world")
            raise ValueError(f"{what} needs the Python choreography of tests/pydriver.py (a test double: import it first); "
                             "fit() itself runs the gather legs behind the C ABI (mode 'auto' / 'sparse') on one GPU, on "
                             "LocalWorld(P) or on an RCCL TorchWorld")
        if world.size == 1 and not isinstance(world, LocalWorld) and storage == "fp16":
            world = LocalWorld(1, loop="python")
        return PYTHON_SOLVER(ops_factory, device, world, specs, mode)
    factory = ops_factory or _default_ops_factory(device)
    if world.size == 1 and (isinstance(world, LocalWorld) or loop != "c"):
        # (a one-rank process group holds everything: the single-GPU plan, unless the caller asks for the sharded loop's
        # RCCL path on one rank — TorchWorld(loop="c"), how that path is exercised on one GPU)
        ops = factory(0)
        one = world if isinstance(world, LocalWorld) else LocalWorld(1)
        if not cplan.applies(None, one, specs, "sparse") or not cplan.lean_knobs(ops):
            raise ValueError("this fit has no C-level plan: the plans behind the C ABI run the default kernel knobs, one storage "
                             "precision, fp16-held matrices for the one-matrix classes with symmetric priors only, exact "
                             "products in the two-matrix classes")
        return cplan.PlanSolver(ops, one, specs)
    why = cshard.applies(world, specs, "sparse")
    if why is not None:
        raise ValueError(("storage_precision='fp16' on several ranks: " if storage == "fp16" else "fit on several ranks: ") + why)
    return cshard.CShardSolver(factory, world, specs)


_engines = threading.local()        # the default engine of a (thread, device): its stream, counters and slots made once


def _default_ops_factory(device):
    from .engine import HipOps          # raises if the library or the GPU is missing
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
    cache = _engines.__dict__.setdefault("by_device", {})
    ops = cache.get(device)
    if ops is None:
        # (4 ms per fit otherwise — stream, counter set, pinned slots; one per thread, so that two threads fitting at once never
        # share a stream or a convergence counter)
        ops = cache[device] = HipOps(device)
    return lambda rank: ops


def _square_frame(S, labels):
    """``pd.DataFrame(S, index=labels, columns=labels)`` (SimRank.py:141) with the labels converted once instead of twice
    (7 ms for 32768 of them); the two axes are separate Index objects, as the reference's."""
    idx = pd.Index(labels)
    return pd.DataFrame(S, index=idx, columns=idx.copy())


def _solve(specs, iterations, eps, verbose, mode, device, world, ops_factory=None):
    world = world or LocalWorld(1)
    solver = _make_solver(ops_factory, device, world, specs, mode)
    talk = verbose and world.is_root
    if talk:
        print("Start iterating...")
    k = solver.run(iterations, eps,
                   on_iteration=(lambda i: update_progress(i / iterations)) if talk else None,
                   on_converged=announce_converged if talk else None)
    return solver, k


class _Lazy:
    """Attribute that is computed on first read and can be overwritten like a plain one."""

    def __init__(self, name, default):
        self.name, self.default = "_lazy_" + name, default

    def __get__(self, obj, owner):
        if obj is None:
            return self
        v = obj.__dict__.get(self.name)
        if v is None:
            return self.default()
        if callable(v):
            v = v()
            obj.__dict__[self.name] = v
        return v

    def __set__(self, obj, value):
        obj.__dict__[self.name] = value


def _topk_frame(solver, j, k, labels):
    """Long-format hand-back: one row per (node, rank) with the k most similar OTHER nodes."""
    idx, val = solver.topk(j, k)
    n, kk = idx.shape
    lab = pd.Index(labels)
    keep = idx.ravel() >= 0
    return pd.DataFrame({
        "node": lab.take(np.repeat(np.arange(n), kk)[keep]),
        "rank": np.tile(np.arange(1, kk + 1), n)[keep],
        "neighbor": lab.take(idx.ravel()[keep]),
        "similarity": val.ravel()[keep]})


def _is_symmetric(prior) -> bool:
    """The fused two-gather update needs symmetric iterates; everything the reference builds
    is symmetric except what a user-supplied prior (SimRank.py:453) brings in."""
    if prior is None:
        return True
    a = np.asarray(prior)
    return a.ndim == 2 and a.shape[0] == a.shape[1] and bool(np.array_equal(a, a.T))


class SimRank(object):
    """SimRank on a directed, optionally weighted graph (SimRank.py:8-141).

    Attributes after ``fit``: ``Nodes`` (set), ``Graph`` (N x N DataFrame, built on read),
    ``converged_at`` (loop index of convergence or None), ``engine_mode``.
    """
    Graph = _Lazy("Graph", pd.DataFrame)

    def __init__(self):
        self.Nodes = set()
        self.Graph = pd.DataFrame()

    # -- ingest ------------------------------------------------------------------------
    def _create_graph(self, data, weighted, from_node_column, to_node_column, weight_column):
        nodes, csr = ingest.directed(data, weighted, from_node_column, to_node_column,
                                     weight_column)
        self.Nodes = set(nodes)
        self._order = nodes
        self._csr = csr
        self.Graph = lambda: pd.DataFrame(csr.dense(), index=nodes, columns=nodes)
        return csr

    def _side(self, csr, C):
        return SideSpec(csr, csr.rowscale, C)

    def _finish(self, solver, k, top_k=None):
        self.converged_at = k
        self.engine_mode = solver.mode
        if top_k:
            out = _topk_frame(solver, 0, top_k, self._order)
            solver.release()
            return out
        S = solver.result(0)
        solver.release()
        if S is None:                     # multi-process world, root-only hand-back: not the root
            return None
        return _square_frame(S, self._order)

    def fit(self, data, C=0.8, weighted=False, from_node_column="from", to_node_column="to",
            weight_column="weight", iterations=100, eps=1e-4, verbose=True, *,
            mode="auto", device=None, world=None, top_k=None, dense_precision="f32", storage_precision="f32",
            _ops_factory=None):
        with _precision(dense_precision, storage_precision):
            csr = self._create_graph(data, weighted, from_node_column, to_node_column, weight_column)
            solver, k = _solve([self._side(csr, C)], iterations, eps, verbose, mode, device, world,
                               _ops_factory)
            return self._finish(solver, k, top_k)


class SimRankPP(SimRank):
    """SimRank++: evidence and spread weighting (SimRank.py:305-363)."""
    Evidence = _Lazy("Evidence", pd.DataFrame)
    Weight = _Lazy("Weight", pd.DataFrame)

    def __init__(self):
        super(SimRankPP, self).__init__()
        self.Evidence = pd.DataFrame()
        self.Weight = pd.DataFrame()

    def _cal_Weight(self, csr, verbose):
        """spread[v] . rowscale[v]: the per-row scale of W (SimRank.py:322-337)."""
        if verbose:
            print("Initializing Weight matrix...")
        start = time.time()
        scale = ingest.spread(csr) * csr.rowscale
        if verbose:
            print(f"Finished in {time.time() - start}s!")
        return scale

    def _pp_side(self, csr, C, verbose, apriori=None, lbd=0.0, evidence_from=None):
        scale = self._cal_Weight(csr, verbose)
        return SideSpec(csr, scale, C, evidence_from=evidence_from or csr, apriori=apriori,
                        lbd=lbd, symmetric=_is_symmetric(apriori))

    def _fit_pp(self, data, C, weighted, from_node_column, to_node_column, weight_column,
                iterations, eps, verbose, mode, device, world, ops_factory, apriori=None,
                lbd=0.0, top_k=None):
        csr = self._create_graph(data, weighted, from_node_column, to_node_column, weight_column)
        talk = verbose and (world is None or world.is_root)
        spec = self._pp_side(csr, C, talk, apriori, lbd)
        self.Weight = lambda: csr.dense(spec.rowscale)
        if talk:
            print("Initializing Evidence matrix...")
        start = time.time()
        # the counts are produced on the device while the solver is set up
        world = world or LocalWorld(1)
        solver = _make_solver(ops_factory, device, world, [spec], mode)
        for o in solver.ops.values():
            o.synchronize()
        if talk:
            print(f"Finished in {time.time() - start}s!")
            print("Start iterating...")
        k = solver.run(iterations, eps,
                       on_iteration=(lambda i: update_progress(i / iterations)) if talk else None,
                       on_converged=announce_converged if talk else None)
        self.Evidence = _lazy_evidence(world, solver, 0, csr)
        return self._finish(solver, k, top_k)

    def fit(self, data, C=0.8, weighted=False, from_node_column="from", to_node_column="to",
            weight_column="weight", iterations=100, eps=1e-4, verbose=True, *,
            mode="auto", device=None, world=None, top_k=None, dense_precision="f32", storage_precision="f32",
            _ops_factory=None):
        with _precision(dense_precision, storage_precision):
            return self._fit_pp(data, C, weighted, from_node_column, to_node_column, weight_column,
                                iterations, eps, verbose, mode, device, world, _ops_factory,
                                top_k=top_k)


class AprioriSimRank(SimRankPP):
    """SimRank++ blended with a prior similarity matrix (SimRank.py:427-455)."""

    def __init__(self):
        super(AprioriSimRank, self).__init__()

    def fit(self, data, AprioriSim, C=0.8, lbd=0.5, weighted=False, from_node_column="from",
            to_node_column="to", weight_column="weight", iterations=100, eps=1e-4,
            verbose=True, *, mode="auto", device=None, world=None, top_k=None,
            dense_precision="f32", storage_precision="f32", _ops_factory=None):
        with _precision(dense_precision, storage_precision):
            if not isinstance(AprioriSim, np.ndarray):
                # the reference fails at np.fill_diagonal for anything but an ndarray
                raise AttributeError(f"'{type(AprioriSim).__name__}' object has no attribute 'flat'")
            return self._fit_pp(data, C, weighted, from_node_column, to_node_column, weight_column,
                                iterations, eps, verbose, mode, device, world, _ops_factory,
                                apriori=AprioriSim, lbd=lbd, top_k=top_k)


# ----------------------------------------------------------------------------------------
# bipartite
# ----------------------------------------------------------------------------------------
class BipartiteSimRank(object):
    """SimRank on a bipartite graph: two similarity matrices updated alternately, the
    second from the just-updated first (SimRank.py:143-303)."""
    Graph_N1_N2 = _Lazy("Graph_N1_N2", pd.DataFrame)
    Graph_N2_N1 = _Lazy("Graph_N2_N1", pd.DataFrame)

    def __init__(self):
        self.NodesGroup1 = set()
        self.NodesGroup2 = set()
        self.Graph_N1_N2 = pd.DataFrame()
        self.Graph_N2_N1 = pd.DataFrame()

    def _create_graph(self, data, weighted, node_group1_column, node_group2_column,
                      weight_column):
        set1, set2, lab1, lab2, g12, g21 = ingest.bipartite(
            data, weighted, node_group1_column, node_group2_column, weight_column)
        self.NodesGroup1, self.NodesGroup2 = set(set1), set(set2)
        self._set_order = (set1, set2)
        self._sorted = (lab1, lab2)
        self._csr = (g12, g21)
        self.Graph_N1_N2 = lambda: pd.DataFrame(g12.dense(), index=lab1, columns=lab2)
        self.Graph_N2_N1 = lambda: pd.DataFrame(g21.dense(), index=lab2, columns=lab1)
        return g12, g21

    def _finish(self, solver, k, strict_reference, top_k=None):
        self.converged_at = k
        self.engine_mode = solver.mode
        l1, l2 = self._set_order if strict_reference else map(list, self._sorted)
        if top_k:
            out = (_topk_frame(solver, 0, top_k, l1), _topk_frame(solver, 1, top_k, l2))
            solver.release()
            return out
        S1, S2 = solver.result(0), solver.result(1)
        solver.release()
        if S1 is None:                    # multi-process world, root-only hand-back: not the root
            return None
        return (_square_frame(S1, l1), _square_frame(S2, l2))

    def fit(self, data, C1=0.8, C2=0.8, weighted=False, node_group1_column="user",
            node_group2_column="item", weight_column="weight", iterations=100, eps=1e-4,
            verbose=True, *, mode="auto", device=None, world=None, strict_reference=True,
            top_k=None, dense_precision="f32", storage_precision="f32", _ops_factory=None):
        with _precision(dense_precision, storage_precision):
            g12, g21 = self._create_graph(data, weighted, node_group1_column, node_group2_column,
                                          weight_column)
            specs = [SideSpec(g12, g12.rowscale, C1), SideSpec(g21, g21.rowscale, C2)]
            solver, k = _solve(specs, iterations, eps, verbose, mode, device, world, _ops_factory)
            return self._finish(solver, k, strict_reference, top_k)


class BipartiteSimRankPP(SimRankPP):
    """Bipartite SimRank++ (SimRank.py:365-425)."""
    Graph_N1_N2 = _Lazy("Graph_N1_N2", pd.DataFrame)
    Graph_N2_N1 = _Lazy("Graph_N2_N1", pd.DataFrame)
    Evidence_N1 = _Lazy("Evidence_N1", pd.DataFrame)
    Evidence_N2 = _Lazy("Evidence_N2", pd.DataFrame)
    Weight_N1 = _Lazy("Weight_N1", pd.DataFrame)
    Weight_N2 = _Lazy("Weight_N2", pd.DataFrame)

    def __init__(self):
        self.NodesGroup1 = set()
        self.NodesGroup2 = set()
        for name in ("Graph_N1_N2", "Graph_N2_N1", "Evidence_N1", "Evidence_N2", "Weight_N1",
                     "Weight_N2"):
            setattr(self, name, pd.DataFrame())

    _create_graph = BipartiteSimRank._create_graph
    _finish = BipartiteSimRank._finish

    def _fit_bpp(self, data, C1, C2, weighted, node_group1_column, node_group2_column,
                 weight_column, iterations, eps, verbose, mode, device, world, strict_reference,
                 ops_factory, priors=(None, None), lbds=(0.0, 0.0), top_k=None):
        g12, g21 = self._create_graph(data, weighted, node_group1_column, node_group2_column,
                                      weight_column)
        world = world or LocalWorld(1)
        talk = verbose and world.is_root
        w1 = self._cal_Weight(g12, talk)                               # SimRank.py:396
        w2 = self._cal_Weight(g21, talk)                               # :397
        self.Weight_N1 = lambda: g12.dense(w1)
        self.Weight_N2 = lambda: g21.dense(w2)
        # quirk Q2 (:423, :491): the group-2 update is gated by Evidence_N1
        ev2 = g12 if strict_reference else g21
        # a non-symmetric prior on either side makes BOTH iterates non-symmetric
        sym = _is_symmetric(priors[0]) and _is_symmetric(priors[1])
        specs = [SideSpec(g12, w1, C1, evidence_from=g12, apriori=priors[0], lbd=lbds[0],
                          symmetric=sym),
                 SideSpec(g21, w2, C2, evidence_from=ev2, apriori=priors[1], lbd=lbds[1],
                          symmetric=sym)]
        if talk:
            print("Initializing Evidence matrix...")
        start = time.time()
        solver = _make_solver(ops_factory, device, world, specs, mode)
        for o in solver.ops.values():
            o.synchronize()
        if talk:
            print(f"Finished in {time.time() - start}s!")
            print("Initializing Evidence matrix...")
            print(f"Finished in {0.0}s!")
            print("Start iterating...")
        k = solver.run(iterations, eps,
                       on_iteration=(lambda i: update_progress(i / iterations)) if talk else None,
                       on_converged=announce_converged if talk else None)
        self.Evidence_N1 = _lazy_evidence(world, solver, 0, g12)
        self.Evidence_N2 = ((lambda: _host_evidence(g21)) if strict_reference
                            else _lazy_evidence(world, solver, 1, g21))
        return self._finish(solver, k, strict_reference, top_k)

    def fit(self, data, C1=0.8, C2=0.8, weighted=False, node_group1_column="user",
            node_group2_column="item", weight_column="weight", iterations=100, eps=1e-4,
            verbose=True, *, mode="auto", device=None, world=None, strict_reference=True,
            top_k=None, dense_precision="f32", storage_precision="f32", _ops_factory=None):
        with _precision(dense_precision, storage_precision):
            return self._fit_bpp(data, C1, C2, weighted, node_group1_column, node_group2_column,
                                 weight_column, iterations, eps, verbose, mode, device, world,
                                 strict_reference, _ops_factory, top_k=top_k)


class BipartitleAprioriSimRank(BipartiteSimRankPP):
    """Bipartite SimRank++ with priors (SimRank.py:457-493; the class name is the
    reference's spelling)."""

    def __init__(self):
        super(BipartitleAprioriSimRank, self).__init__()

    def fit(self, data, AprioriSim1, AprioriSim2, C1=0.8, C2=0.8, lbd1=0.5, lbd2=0.5,
            weighted=False, node_group1_column="user", node_group2_column="item",
            weight_column="weight", iterations=100, eps=1e-4, verbose=True, *, mode="auto",
            device=None, world=None, strict_reference=True, top_k=None, dense_precision="f32", storage_precision="f32",
            _ops_factory=None):
        with _precision(dense_precision, storage_precision):
            for a in (AprioriSim1, AprioriSim2):
                if not isinstance(a, np.ndarray):
                    raise AttributeError(f"'{type(a).__name__}' object has no attribute 'flat'")
            return self._fit_bpp(data, C1, C2, weighted, node_group1_column, node_group2_column,
                                 weight_column, iterations, eps, verbose, mode, device, world,
                                 strict_reference, _ops_factory, priors=(AprioriSim1, AprioriSim2),
                                 lbds=(lbd1, lbd2), top_k=top_k)


def _lazy_evidence(world, solver, j, csr):
    """Reader for an ``Evidence`` attribute: 1 - 0.5**count as float64 (SimRank.py:316; the
    device counts saturate at 255, and 0.5**54 already rounds 1 - x to 1.0, so saturation is
    exact).  With every shard in this process the counts come back from the device; in a
    multi-process world a lazy read must not be a collective (only some ranks may read
    it), so it is recomputed from the CSR."""
    if isinstance(world, LocalWorld) and hasattr(solver, "evidence"):
        return lambda: solver.evidence(j)
    return lambda: _host_evidence(csr)


def _host_evidence(csr):
    """Evidence of a pattern computed on the host from the CSR (only for the attribute the
    reference computes but never uses, Evidence_N2 in strict mode)."""
    import scipy.sparse as sp
    live = np.repeat(csr.rowscale > 0, np.diff(csr.rowptr))
    pat = sp.csr_matrix((live.astype(np.int64), csr.col, csr.rowptr),
                        shape=(csr.n_rows, csr.n_cols))
    return 1 - 0.5 ** np.asarray((pat @ pat.T).todense(), dtype=np.float64)


# spellings used by the reference README (README.md:16) and BASELINE.json
BipartitleSimRank = BipartiteSimRank
BipartitleSimRankPP = BipartiteSimRankPP
BipartiteAprioriSimRank = BipartitleAprioriSimRank
