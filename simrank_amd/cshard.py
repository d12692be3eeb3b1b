"""fp16-held matrices on SHARDS through the reference's class surface: ``fit(storage_precision="fp16", world=...)`` with
more than one rank (BASELINE.json config 5 in its stated form: N = 65536 SimRank++, reduced precision, 8 GPUs).

The Python driver's solver keeps fp16-held matrices to one rank; the sharded loop on such matrices lives behind the C ABI
(``simrank_shardplan_*``, csrc/shardplan.hip: leg 1 and a full-form leg 2 of half.hip on every rank's column block, the fp16
panels themselves on the links).  ``CShardSolver`` gives that loop the few methods the estimators ask of a solver —
``run`` with the reference's progress callbacks (SimRank.py:129-140), ``result``, ``topk``, ``release`` — over

* ``LocalWorld(P)``: an in-process group of P virtual ranks on one device (tests, single-GPU emulation), or
* ``TorchWorld`` on RCCL ranks: the library's own RCCL communicator, made from an id rank 0 broadcasts through
  ``torch.distributed``; the library is pointed at the RCCL build torch itself loaded, so one process never runs two.

Since round 5 the same solver runs the f32 (parity-grade) sharded fits of an RCCL world as well — every class with
class, the two-matrix ones through ``simrank_shardbiplan_*``, asymmetric priors with a second all-to-all and an un-fused
epilogue — so that what a multi-GPU user's ``fit`` runs is the C loop, not a second choreography in Python;
``driver.Solver`` keeps the GEMM modes and the CPU rehearsal over gloo with the NumPy test double.
"""
from __future__ import annotations

import os

import numpy as np

from .driver import LocalWorld, TorchWorld


def applies(world, specs, mode) -> str | None:
    """None when the C sharded loop can run these specs, else the reason it cannot."""
    if mode not in ("auto", "sparse"):
        return "the sharded C loop runs the gather legs only (mode 'sparse' or 'auto')"
    if not all(s.symmetric for s in specs):
        # asymmetric iterates: leg 2's product goes round a second all-to-all, the epilogue is a pass of its own (f32 only)
        if any(s.storage != "f32" for s in specs):
            return "an asymmetric prior needs f32 matrices"
    if len({s.storage for s in specs}) != 1 or any(s.dense_terms != 3 for s in specs):
        return "one storage precision for every matrix, exact products on the matrix cores"
    fp16 = specs[0].storage == "fp16"
    if isinstance(world, TorchWorld) and world.dist.get_backend(world.group) != "nccl":
        return "the sharded C loop exchanges over RCCL (one GPU per process)"
    if len(specs) == 2:
        a, b = specs
        if fp16:
            return "the bipartite classes keep fp16-held matrices to one GPU"
        if (a.evidence_from is None) != (b.evidence_from is None):
            return "evidence on one group only"
        if a.evidence_from is not None and (a.evidence_from is not a.csr or
                                            not (b.evidence_from is a.csr or b.evidence_from is b.csr)):
            return "evidence of a foreign pattern"
        if a.csr.n_rows != b.csr.n_cols or a.csr.n_cols != b.csr.n_rows or a.csr.nnz != b.csr.nnz:
            return "the two patterns are not each other's transpose"
        return None
    s = specs[0]
    if s.evidence_from is not None and s.evidence_from is not s.csr:
        return "evidence of a foreign pattern"
    if fp16 and s.apriori is not None:
        return "a prior keeps fp16-held matrices to one GPU"
    if fp16 and s.csr.n_rows % (64 * world.size):
        return f"fp16-held matrices on {world.size} ranks need the node count to be a multiple of {64 * world.size}"
    return None


def _rccl_comm(world, ops):
    """The library's communicator for this torch world (made once per world)."""
    comm = getattr(world, "_c_comm", None)
    if comm is not None:
        return comm
    from .engine import ShardPlans
    if "SIMRANK_RCCL_LIB" not in os.environ:
        import torch
        bundled = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        if os.path.exists(bundled):              # the RCCL torch runs on (and the HIP runtime it was built for)
            os.environ["SIMRANK_RCCL_LIB"] = bundled
    box = [ShardPlans.rccl_unique_id(ops.lib) if world.rank == 0 else None]
    world.dist.broadcast_object_list(box, src=0, group=world.group)
    # (RCCL prints a version banner on STDOUT when a communicator is made outside torch; a program that promised one JSON
    # line there — bench.py — must not carry it: the banner goes to stderr)
    import sys
    sys.stdout.flush()
    saved = os.dup(1)
    try:
        os.dup2(2, 1)
        comm = ShardPlans.rccl_comm(ops.lib, box[0], world.rank, world.size)
    finally:
        os.dup2(saved, 1)
        os.close(saved)
    world._c_comm = comm
    return comm


def _prior32(spec):
    if spec.apriori is None:
        return None
    a = np.asarray(spec.apriori)
    n = spec.csr.n_rows
    if a.shape != (n, n):
        raise ValueError(f"operands could not be broadcast together with shapes ({n},{n}) {a.shape} ")
    return np.ascontiguousarray(a, dtype=np.float32)


class CShardSolver:
    """The estimators' view of ``engine.ShardPlans`` / ``engine.ShardBiPlans``: the sharded loops behind the C ABI
    (csrc/shardplan.hip) — every class, f32 (the parity path) or, for SimRank / SimRank++ without a prior, fp16-held
    matrices."""

    mode = "sparse"

    def __init__(self, make_ops, world, specs):
        from .engine import ShardBiPlans, ShardPlans
        if not isinstance(specs, (list, tuple)):
            specs = [specs]
        self.world = world
        self.specs = specs
        self.bipartite = len(specs) == 2
        self.storage = specs[0].storage
        self.n = [s.csr.n_rows for s in specs]
        self.ops = {r: make_ops(r) for r in world.local_ranks}
        ops = self.ops[world.local_ranks[0]]
        if self.storage == "fp16" and not getattr(ops, "supports_half_storage", False):
            raise ValueError("storage_precision='fp16' needs the HIP engine (matrices held in fp16: csrc/half.hip)")
        local = isinstance(world, LocalWorld)
        sym = getattr(world, "symmetric_shards", True)
        form = -1 if sym == "auto" else (1 if sym else 0)
        if self.storage == "fp16" or not all(s.symmetric for s in specs):
            form = 0                 # (no mirror image to share: fp16-held blocks, asymmetric iterates)
        common = dict(world=world.size, comm=None if local else _rccl_comm(world, ops),
                      stages=0 if local else getattr(world, "stages", 0),
                      wire_fp16=getattr(world, "exchange_precision", "f32") == "fp16")
        self.broadcast_error = None
        if self.bipartite:
            a, b = specs
            evidence = a.evidence_from is not None
            strict = evidence and b.evidence_from is a.csr              # quirk Q2: Evidence_N1 gates both updates
            if strict and self.n[0] != self.n[1] and self.n[0] != 1:
                self.broadcast_error = ValueError(
                    f"operands could not be broadcast together with shapes "
                    f"({self.n[0]},{self.n[0]}) ({self.n[1]},{self.n[1]}) ")
            def fits(n):
                return n % (32 * world.size) == 0
            if form == 1 and not (fits(self.n[0]) or fits(self.n[1])):
                form = 0
            self.plans = ShardBiPlans(ops, a.csr, a.rowscale, b.rowscale, c1=a.coef, c2=b.coef, evidence=evidence,
                                      apriori1=_prior32(a), apriori2=_prior32(b), lbd1=a.lbd, lbd2=b.lbd,
                                      strict_reference=strict, leg2_form=form, **common)
        else:
            (s,) = specs
            if form == 1 and s.csr.n_rows % (32 * world.size):
                form = 0
            self.plans = ShardPlans(ops, s.csr, rowscale=s.rowscale, coef=s.coef, evidence=s.evidence_from is not None,
                                    apriori=_prior32(s), lbd=s.lbd, storage=self.storage, leg2_form=form, **common)
        self.root = local or world.rank == 0

    def run(self, iterations, eps, on_iteration=None, on_converged=None):
        """The loop of SimRank.py:129-140 / :288-302 (the count of every update is read before the next one is queued)."""
        if self.broadcast_error is not None and iterations > 0 and 1.0 > eps:
            if on_iteration:
                on_iteration(0)
            raise self.broadcast_error
        self.plans.reset()
        changed = sum(self.n) if 1.0 > eps else 0
        for k in range(iterations):
            if changed == 0:
                if on_converged:
                    on_converged(k)
                return k
            if on_iteration:
                on_iteration(k)
            c = self.plans.step(eps, exact_count=False)
            changed = sum(c) if self.bipartite else c
        return None

    def _share(self, value):
        """Root's hand-back to the ranks that asked for one (TorchWorld(handback="all"), top-k)."""
        box = [value]
        self.world.dist.broadcast_object_list(box, src=0, group=self.world.group)
        return box[0]

    def result(self, j=0):
        full = (self.plans.result(j + 1, root=0, i_am_root=self.root) if self.bipartite
                else self.plans.result(root=0, i_am_root=self.root))
        if isinstance(self.world, LocalWorld) or getattr(self.world, "handback", "root") == "root":
            if not self.root:
                import warnings
                warnings.warn("TorchWorld(handback='root'): only rank 0 receives the similarity matrix, fit() returns "
                              "None on this rank (pass handback='all', or fit(top_k=k), to get results on every rank)",
                              RuntimeWarning, stacklevel=4)
            return full
        return self._share(full)

    def topk(self, j, k, exclude_diag=True):
        n = self.n[j]
        k = int(min(k, max(1, n - (1 if exclude_diag else 0))))
        idx, val = (self.plans.topk(j + 1, k, exclude_diag, root=0, i_am_root=self.root) if self.bipartite
                    else self.plans.topk(k, exclude_diag, root=0, i_am_root=self.root))
        if not isinstance(self.world, LocalWorld):
            idx, val = self._share((idx, val))
        return idx, val.astype(np.float64)

    def release(self):
        self.plans.free()
