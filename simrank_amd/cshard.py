"""fp16-held matrices on SHARDS through the reference's class surface: ``fit(storage_precision="fp16", world=...)`` with
more than one rank (BASELINE.json config 5 in its stated form: N = 65536 SimRank++, reduced precision, 8 GPUs).

The Python driver's solver keeps fp16-held matrices to one rank; the sharded loop on such matrices lives behind the C ABI
(``simrank_shardplan_*``, csrc/shardplan.hip: leg 1 and a full-form leg 2 of half.hip on every rank's column block, the fp16
panels themselves on the links).  ``CShardSolver`` gives that loop the few methods the estimators ask of a solver —
``run`` with the reference's progress callbacks (SimRank.py:129-140), ``result``, ``topk``, ``release`` — over

* ``LocalWorld(P)``: an in-process group of P virtual ranks on one device (tests, single-GPU emulation), or
* ``TorchWorld`` on RCCL ranks: the library's own RCCL communicator, made from an id rank 0 broadcasts through
  ``torch.distributed``; the library is pointed at the RCCL build torch itself loaded, so one process never runs two.

Only what that entry point runs: one symmetric side (SimRank, SimRank++), no prior, n a multiple of 64 x ranks.
"""
from __future__ import annotations

import os

import numpy as np

from .driver import LocalWorld, TorchWorld


def applies(world, specs, mode) -> str | None:
    """None when the C sharded loop can run these specs on fp16-held matrices, else the reason it cannot."""
    if len(specs) != 1:
        return "the bipartite classes keep fp16-held matrices to one GPU"
    s = specs[0]
    if not s.symmetric or s.apriori is not None:
        return "a prior keeps fp16-held matrices to one GPU"
    if mode not in ("auto", "sparse"):
        return "fp16-held matrices exist for the gather legs only (mode 'sparse' or 'auto')"
    if s.csr.n_rows % (64 * world.size):
        return f"fp16-held matrices on {world.size} ranks need the node count to be a multiple of {64 * world.size}"
    if isinstance(world, TorchWorld) and world.dist.get_backend(world.group) != "nccl":
        return "fp16-held matrices on shards need RCCL ranks (one GPU per process)"
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
    comm = ShardPlans.rccl_comm(ops.lib, box[0], world.rank, world.size)
    world._c_comm = comm
    return comm


class CShardSolver:
    """The estimators' view of ``engine.ShardPlans`` (fp16-held matrices on every rank)."""

    mode = "sparse"
    storage = "fp16"

    def __init__(self, make_ops, world, spec):
        from .engine import ShardPlans
        self.world = world
        self.n = [spec.csr.n_rows]
        self.ops = {r: make_ops(r) for r in world.local_ranks}
        ops = self.ops[world.local_ranks[0]]
        if not getattr(ops, "supports_half_storage", False):
            raise ValueError("storage_precision='fp16' needs the HIP engine (matrices held in fp16: csrc/half.hip)")
        evidence = spec.evidence_from is not None
        if evidence and spec.evidence_from is not spec.csr:
            raise ValueError("fp16-held matrices on shards take the evidence of the graph itself")
        kw = dict(coef=spec.coef, evidence=evidence, storage="fp16", leg2_form=0, stages=getattr(world, "stages", 0))
        if isinstance(world, LocalWorld):
            self.plans = ShardPlans(ops, spec.csr, rowscale=spec.rowscale, world=world.size, **kw)
        else:
            self.plans = ShardPlans(ops, spec.csr, rowscale=spec.rowscale, world=world.size,
                                    comm=_rccl_comm(world, ops), **kw)
        self.root = isinstance(world, LocalWorld) or world.rank == 0

    def run(self, iterations, eps, on_iteration=None, on_converged=None):
        """The loop of SimRank.py:129-140 (the count of every update is read before the next one is queued: a rank's
        update on matrices this mode is for takes milliseconds)."""
        self.plans.reset()
        changed = self.n[0] if 1.0 > eps else 0
        for k in range(iterations):
            if changed == 0:
                if on_converged:
                    on_converged(k)
                return k
            if on_iteration:
                on_iteration(k)
            changed = self.plans.step(eps, exact_count=False)
        return None

    def _share(self, value):
        """Root's hand-back to the ranks that asked for one (TorchWorld(handback="all"), top-k)."""
        box = [value]
        self.world.dist.broadcast_object_list(box, src=0, group=self.world.group)
        return box[0]

    def result(self, j=0):
        full = self.plans.result(root=0, i_am_root=self.root)
        if isinstance(self.world, LocalWorld) or getattr(self.world, "handback", "root") == "root":
            return full
        return self._share(full)

    def topk(self, j, k, exclude_diag=True):
        n = self.n[0]
        k = int(min(k, max(1, n - (1 if exclude_diag else 0))))
        idx, val = self.plans.topk(k, exclude_diag, root=0, i_am_root=self.root)
        if not isinstance(self.world, LocalWorld):
            idx, val = self._share((idx, val))
        return idx, val.astype(np.float64)

    def release(self):
        self.plans.free()
