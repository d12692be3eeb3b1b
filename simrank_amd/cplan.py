"""The estimators' solver on ONE GPU: the loops of ``fit`` behind the C ABI.

``PlanSolver`` gives ``engine.Plan`` (``simrank_plan_*``: SimRank.py:129-141, :351-363, :443-455) and ``engine.BiPlan``
(``simrank_biplan_*``: :288-303, :410-425, :478-493) the few methods ``estimators.py`` asks of a solver — ``run`` with
the reference's console hooks, ``result``, ``topk``, ``evidence``, ``release`` — so that what a user imports runs the
fastest loop the library has: both legs and the count of an update queued by one C call, update k + 1 queued before the
count of update k is read on small graphs, and a banded hand-back (csrc/handback.hip: full form by default; the
upper-triangle form is opt-in, ``SIMRANK_SYM_HANDBACK=1``, and checks on the device that the result is symmetric).

An asymmetric prior (``SimRank.py:453``: asymmetric iterates) runs in the same plans with leg 2 stored transposed and the
epilogue as a pass of its own.  ``driver.Solver`` (the same choreography in Python) stays for what a plan does not run: the
dense / hybrid GEMM modes, virtual or real ranks (``LocalWorld(P > 1)``, ``TorchWorld``), non-default
kernel knobs, and the NumPy test double of the CPU tests.
"""
from __future__ import annotations

import numpy as np

from .driver import LocalWorld, lean_knobs


def applies(ops_factory, world, specs, mode) -> bool:
    """Can the C-level plan run these specs?  One rank of this process, the gather legs, symmetric iterates, the HIP
    engine (``ops_factory`` None = the default engine) with its kernel knobs at their defaults."""
    if ops_factory is not None or not isinstance(world, LocalWorld) or world.size != 1:
        return False
    if mode not in ("auto", "sparse"):
        return False
    if not all(s.symmetric for s in specs) and any(s.storage != "f32" for s in specs):
        return False                 # (asymmetric priors: the f32 plans run them un-fused; fp16-held matrices do not)
    if len({s.storage for s in specs}) != 1 or len({s.dense_terms for s in specs}) != 1:
        return False
    if len(specs) == 2:
        a, b = specs
        if a.storage != "f32" or a.dense_terms != 3:
            return False             # (the two-matrix plan is f32 with exact products only)
        if (a.evidence_from is None) != (b.evidence_from is None):
            return False
        if a.evidence_from is not None and (a.evidence_from is not a.csr or
                                            not (b.evidence_from is a.csr or b.evidence_from is b.csr)):
            return False
        if a.csr.n_rows != b.csr.n_cols or a.csr.n_cols != b.csr.n_rows or a.csr.nnz != b.csr.nnz:
            return False
    else:
        (s,) = specs
        if s.evidence_from is not None and s.evidence_from is not s.csr:
            return False
        if s.csr.n_rows != s.csr.n_cols:
            return False
    return True


def _prior32(spec):
    """The prior as the plans take it: float32, C-contiguous (SimRank.py:453 blends a float64 array; the rounding to
    float32 is the engine's storage precision, as in driver.Side)."""
    if spec.apriori is None:
        return None
    a = np.asarray(spec.apriori)
    n = spec.csr.n_rows
    if a.shape != (n, n):
        raise ValueError(f"operands could not be broadcast together with shapes ({n},{n}) {a.shape} ")
    return np.ascontiguousarray(a, dtype=np.float32)


class PlanSolver:
    """``driver.Solver``'s surface over ``engine.Plan`` / ``engine.BiPlan``."""

    mode = "sparse"

    def __init__(self, ops, world, specs):
        from .engine import BiPlan, Plan
        if not lean_knobs(ops):
            raise ValueError("the C-level plans need the default kernel knobs")
        self.world = world
        self.ops = {0: ops}
        self.specs = specs
        self.bipartite = len(specs) == 2
        self.storage = specs[0].storage
        self.broadcast_error = None
        if self.bipartite:
            a, b = specs
            self.n = [a.csr.n_rows, b.csr.n_rows]
            evidence = a.evidence_from is not None
            strict = evidence and b.evidence_from is a.csr           # quirk Q2: Evidence_N1 gates both updates
            if strict and self.n[0] != self.n[1] and self.n[0] != 1:
                # NumPy raises when the first group-2 update RUNS (SimRank.py:423, :491), not at set-up
                self.broadcast_error = ValueError(
                    f"operands could not be broadcast together with shapes "
                    f"({self.n[0]},{self.n[0]}) ({self.n[1]},{self.n[1]}) ")
            self.plan = BiPlan(ops, a.csr, a.rowscale, b.rowscale, c1=a.coef, c2=b.coef, evidence=evidence,
                               apriori1=_prior32(a), apriori2=_prior32(b), lbd1=a.lbd, lbd2=b.lbd,
                               strict_reference=strict)
        else:
            (s,) = specs
            self.n = [s.csr.n_rows]
            ap = _prior32(s)
            if ap is not None and s.storage == "fp16" and not (np.isfinite(ap).all() and float(np.abs(ap).max()) < 3.99):
                raise ValueError("storage_precision='fp16' needs prior values below 4 in magnitude")
            self.plan = Plan(ops, s.csr, s.rowscale, coef=s.coef, evidence=s.evidence_from is not None, apriori=ap,
                             lbd=s.lbd, storage=s.storage, dense_terms=s.dense_terms)

    def run(self, iterations, eps, on_iteration=None, on_converged=None):
        """The loop of SimRank.py:129-140 / :288-302.  Returns k (loop index at which the test passed) or None when
        ``iterations`` updates were applied."""
        if self.broadcast_error is not None and iterations > 0 and 1.0 > eps:
            if on_iteration:
                on_iteration(0)                 # (the reference has printed its first progress line and updated S1)
            raise self.broadcast_error
        _, k = self.plan.run(iterations, eps, on_iteration, on_converged)
        return k

    def result(self, j=0):
        return self.plan.result_group(j + 1) if self.bipartite else self.plan.result()

    def topk(self, j, k, exclude_diag=True):
        n = self.n[j]
        k = int(min(k, max(1, n - (1 if exclude_diag else 0))))
        idx, val = self.plan.topk(j + 1, k, exclude_diag) if self.bipartite else self.plan.topk(k, exclude_diag)
        return idx, val.astype(np.float64)

    def evidence(self, j=0):
        """Evidence matrix of side j (1 - 0.5**count, SimRank.py:316) as float64 in the caller's node order."""
        cnt = self.plan.evidence_counts(j + 1) if self.bipartite else self.plan.evidence_counts()
        return 1 - 0.5 ** cnt.astype(np.float64)

    def release(self):
        """Free the matrices of the loop; the evidence counts stay (the ``Evidence`` attributes read them lazily)."""
        self.plan.trim()
