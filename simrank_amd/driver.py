"""Worlds and specs of a fit: WHO runs the update (one GPU, P ranks) and WHAT it is (``SideSpec``).

The loops themselves live behind the C ABI (``csrc/plan.hip``, ``biplan.hip``, ``shardplan.hip``); ``estimators.py`` picks
one of two solvers for a world:

    one rank                      ``cplan.PlanSolver``   -> simrank_plan_* / simrank_biplan_*
    several ranks                 ``cshard.CShardSolver`` -> simrank_shardplan_* / simrank_shardbiplan_*
                                  (``LocalWorld(P)``: P virtual ranks of this process on one device, device copies for
                                  links; ``TorchWorld``: one process per GPU, the library's own RCCL communicator)

Until round 6 this module also held a kernel-by-kernel Python choreography of the same update (``Solver``, ``Side``).
It now lives under ``tests/pydriver.py`` — the engine of the NumPy test double, of the gloo rehearsals on CPU and of
the dense / hybrid GEMM modes, which left ``fit()`` with it — and plugs in through ``estimators.PYTHON_SOLVER``.

Sharding: every similarity matrix is split by COLUMN block over the ranks; leg 1 needs only the rank's
own columns, its product leaves transposed in per-destination chunks, ONE all-to-all per update delivers the operand
of leg 2 (+ a half-size one in the half form of leg 2), one all-reduce of an integer the convergence count
(SimRank.py:129-140 over shards; the reference is one NumPy process and has no counterpart).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .ingest import CSR


def lean_knobs(ops) -> bool:
    """The C-level plans exist for the lean gather kernel (32-column panels, 32-row tiles): with the measurement knobs
    set otherwise (``set_tuning(lean=0)``, ``--panel 64`` ...) a plan would ask the library for a combination it refuses."""
    get = getattr(ops, "get_tuning", None)
    if get is None:
        return True
    return get("lean") == 1 and get("panel") in (0, 32) and get("tile") in (0, 32)


def permute_columns(csr: CSR, perm: np.ndarray) -> CSR:
    """The same pattern with column j renamed perm[j] (columns re-sorted inside each row)."""
    rows = np.repeat(np.arange(csr.n_rows, dtype=np.int64), np.diff(csr.rowptr))
    key = rows * csr.n_cols + perm[csr.col]
    key.sort()
    return CSR(csr.n_rows, csr.n_cols, csr.rowptr, (key % csr.n_cols).astype(np.int32), csr.rowscale)


@dataclass
class SideSpec:
    """One similarity matrix of a fit: S <- coef . W . S_in . W^T (.* E) (+ lbd . A), diag <- 1."""
    csr: CSR                     # W = diag(rowscale) . pattern, M x K
    rowscale: np.ndarray         # float64 [M] actually used (weight-adjusted for SimRank++)
    coef: float
    evidence_from: CSR | None = None   # pattern whose common-neighbour counts gate the update
    apriori: np.ndarray | None = None  # M x M prior
    lbd: float = 0.0
    symmetric: bool = True       # False: a non-symmetric prior makes the iterates asymmetric
    dense_terms: int = 3         # operand terms of the matrix-core part (3 exact; 1 = one fp16 term, config 5)
    storage: str = "f32"         # "fp16": S and the transposed product held in fp16 (config 5)


class LocalWorld:
    """P ranks inside this process on one device (P = 1 is the ordinary single-GPU case; P > 1: virtual ranks — the
    sharded C loop on an in-process group whose exchanges are device copies: tests and single-GPU emulation)."""

    def __init__(self, size: int = 1, symmetric_shards=True, leg2_stages: int = 1, exchange_precision: str = "f32",
                 loop: str = "c"):
        """``symmetric_shards``: sharded symmetric updates run leg 2 in its half form when the node count allows it; False
        keeps the full form, whose results are bit-equal to a single rank's full form.  ``exchange_precision``: "fp16"
        rounds what the ranks hand each other the way the fp16 wire format does.  ``loop``: "c" (the loops behind the C
        ABI); "python" needs a world that brings the Python choreography (``tests/pydriver.py``)."""
        if exchange_precision not in ("f32", "fp16"):
            raise ValueError("exchange_precision must be 'f32' or 'fp16'")
        if loop not in ("python", "c"):
            raise ValueError("loop must be 'python' or 'c'")
        self.loop = loop
        self.exchange_precision = exchange_precision
        self.size = int(size)
        self.local_ranks = list(range(self.size))
        self.is_root = True
        self.stages = 1
        self.leg2_stages = int(leg2_stages)
        self.symmetric_shards = symmetric_shards if symmetric_shards == "auto" else bool(symmetric_shards)


class TorchWorld:
    """One rank per process over torch.distributed — backend "nccl" = RCCL over xGMI: the sharded C loop over the
    library's own RCCL communicator, made from an id rank 0 broadcasts through the process group (``cshard.py``)."""

    def __init__(self, group=None, stages: int = 0, stage_single_rank: bool = False, handback: str = "root",
                 symmetric_shards="auto", measure_single_rank: bool = False, exchange_precision: str = "f32",
                 loop: str = "auto"):
        """``stages``: leg 1 (and the half-form leg 2) in that many column slices, each exchanged behind its own kernel
        (0 = the library's rule).  ``handback``: "root" (default): rank 0 receives the dense result, the other ranks'
        ``fit`` returns None; "all": every rank gets it (small N only); ``fit(top_k=k)`` hands k columns per row to every
        rank either way.  ``symmetric_shards``: True / False / "auto" (the half form of leg 2 from 8 ranks on — it trades
        50 % more bytes on the links for a third less compute per rank; ``bench.py --gpus N`` times both).
        ``exchange_precision="fp16"``: both all-to-alls move fp16 (value x 2^14) — half the bytes on the links, one fp16
        rounding of the transposed product per update: outside the 1e-5 parity bar, never the default.
        ``loop``: "auto" | "c": the C loop ("c" also in a one-rank world); "python": a world of ``tests/pydriver.py``."""
        import torch.distributed as dist
        if handback not in ("root", "all"):
            raise ValueError("handback must be 'root' or 'all'")
        if loop not in ("auto", "c", "python"):
            raise ValueError("loop must be 'auto', 'c' or 'python'")
        if exchange_precision not in ("f32", "fp16"):
            raise ValueError("exchange_precision must be 'f32' or 'fp16'")
        self.loop = loop
        self.exchange_precision = exchange_precision
        self.handback = handback
        self.dist = dist
        self.group = group
        self.size = dist.get_world_size(group)
        self.symmetric_shards = symmetric_shards if symmetric_shards in ("auto", "force") else bool(symmetric_shards)
        self.form_measured = None
        self.measure_single_rank = bool(measure_single_rank)
        self.rank = dist.get_rank(group)
        self.local_ranks = [self.rank]
        self.is_root = self.rank == 0
        self.stages = max(0, int(stages)) if (self.size > 1 or stage_single_rank) else 1
        self.leg2_stages = 0

    def close(self):
        """Destroy the library's own RCCL communicator of this world, if a fit made one (cshard.py)."""
        comm = getattr(self, "_c_comm", None)
        if comm is not None:
            self._c_comm = None
            try:
                from . import _lib
                _lib.load().simrank_comm_destroy(comm)
            except Exception:
                pass

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
