"""The update of SimRank.py:129-140 / :288-302 / :351-362 / :410-424 / :443-454 / :478-492 as a kernel-by-kernel PYTHON
choreography on top of an operation set (``engine.HipOps`` through the C ABI's kernel-level entries, or the NumPy double
``tests/cpu_ops.py``).  Until round 6 this was ``simrank_amd/driver.py`` and what ``fit()`` ran; the product now runs the
loops behind the C ABI only (``cplan.PlanSolver``, ``cshard.CShardSolver``) and this module is TEST INFRASTRUCTURE:

* the engine of the NumPy double (CPU tests of the host logic, world_size-2..4 ``gloo`` rehearsals of the sharded update),
* an independent second implementation the C loops are compared with bit for bit on the GPU,
* the dense / hybrid GEMM modes (BASELINE's literal "dense MFMA leg"), which left ``fit()`` and stay a bench entry.

Importing it installs ``estimators.PYTHON_SOLVER``: ``fit(_ops_factory=...)``, ``fit(mode="dense"|"hybrid")`` and worlds of
this module with ``loop="python"`` (its default) then run ``Solver`` below.

One similarity update   S_out = coef . W . S_in . W^T (.*E) (+ lbd.A), diag <- 1   is two
row-gather SpMMs because S_in is symmetric:

    leg 1   Tt = (W . S_in)^T        stored transposed, in per-destination blocks
    leg 2   S_out = W . Tt  + fused epilogue (also counts |S_out - S_prev| > eps)

Sharding: column blocks, one all-to-all of (N/P) x (N/P) tiles per update, one integer reduced.
The driver is SPMD over "virtual ranks": with a real process group every process runs one;
the loopback world runs P of them on one device.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, replace

import numpy as np

import simrank_amd.driver as _product
from simrank_amd.driver import SideSpec, lean_knobs, permute_columns  # noqa: F401  (re-exported: the specs are the product's)
from simrank_amd.ingest import CSR, partition, relabel

PAD_MIN_ROWS, PAD_MULTIPLE = 1024, 256   # when exchanged chunk rows get padded (row_pad)
# by how many floats (whole 128-byte lines).  Three lines, not one: with ONE line the row stride is 2^k + 1 lines and the
# channel hash of the memory system (XOR of address fields) still sends the rows of a panel to few channels — a rank's
# leg 1 at P = 8 on pl32768d32: 1.04 ms unpadded, 0.80 with one line, 0.63 with three or five
# (profiles/r04_shard_leg1_probe.log; SIMRANK_ROW_PAD / SIMRANK_PITCH_PAD are the measurement knobs)
ROW_PAD = int(os.environ.get("SIMRANK_ROW_PAD", "96"))
RESTRICT_BELOW = 0.5     # SimRank++: leg 2 skips evidence-dead 32-column segments when fewer than this
                         # fraction of them is live (ER N=8192: 0.24 live; the power-law graphs: 0.9)
HALF_FORM_FROM = 8      # TorchWorld(symmetric_shards="auto") when nothing is measured: half-form leg 2 from this many ranks on
MEASURE_FORM_FROM_N = 4096   # ... with at least this many nodes (and > 1 rank) both forms are TIMED and the faster one taken
SPECULATE_BELOW_N = 16384   # run() queues loop body k + 1 before reading the count of body k only below this many nodes
DEAL_UNIT = 128         # nodes are dealt to the shards in runs of this many (dealt_order)
STAGE_ALIGN = 32         # stage widths of a pipelined exchange are multiples of this (panels)


# --------------------------------------------------------------------------------------
# worlds
# --------------------------------------------------------------------------------------
@dataclass
class Xfer:
    """One rank's part of the all-to-all that transposes a column-sharded product.

    The rank computed ``ncols`` columns (global offset ``col_lo``, out of ``col_dim``) for all
    ``row_dim`` rows and stored them with simrank_spmm(transpose_out=1, t_block=mb): block h
    of ``send`` is the (ncols x rows of rank h) transpose meant for rank h.  After the
    exchange ``recv`` is col_dim x (rows of this rank), row-major."""
    ops: object
    rank: int
    send: object
    recv: object
    ncols: int
    col_lo: int
    col_dim: int
    row_dim: int
    mb: int
    nrows: int                 # rows of this rank (width of recv)
    pad: int = 0               # floats appended to every row of a chunk (see row_pad)
    stages: list | None = None  # pipelined exchange: per stage (x_col0, ncols, send_off, recv_off,
                                # in_splits, out_splits, event); None = one all-to-all
    send_t: object = None      # torch views of send / recv (TorchWorld)
    recv_t: object = None
    wire: object = None        # fp16 shadows of the two (exchange_precision="fp16")


class Wire:
    """fp16 shadows of one pair of exchange buffers (``exchange_precision="fp16"``): what the kernels wrote in f32 is
    narrowed (value x 2^14, nearest even, saturating) into ``send_h`` right before a collective moves it, and what
    arrives in ``recv_h`` is widened into the f32 buffer the next leg reads.  Half the bytes on the links; the
    transposed product (and the mirrored tiles of the half form) take one fp16 rounding per update — outside the
    1e-5 parity bar, never the default."""

    def __init__(self, ops, send_t, recv_t):
        self.ops, self.send_t, self.recv_t = ops, send_t, recv_t
        self.send_h = ops.exchange_buffer_h(send_t.numel())
        self.recv_h = ops.exchange_buffer_h(recv_t.numel())

    def pack(self, off, n):
        self.ops.narrow_t(self.send_t, self.send_h, off, n)

    def unpack(self, off, n):
        self.ops.widen_t(self.recv_h, self.recv_t, off, n)


def _wire_view(t):
    """fp16 tensors travel as bytes (every backend moves those; split sizes double)."""
    import torch
    return t.view(torch.uint8)


def row_pad(block_rows: int) -> int:
    """Padding of the rows of an exchanged chunk.  A chunk row has `rows of the receiving
    rank` floats and becomes a row of the receiver's gather operand; when that length is a
    large power-of-two multiple, consecutive rows of a panel fall on the same L2 channels
    and sets, so ``ROW_PAD`` floats (three 128-byte lines) are appended.  Every rank derives it from the
    full block size, so senders and receivers agree."""
    return ROW_PAD if block_rows >= PAD_MIN_ROWS and block_rows % PAD_MULTIPLE == 0 else 0


def auto_stages(k_dim: int, world: int) -> int:
    """Pipeline depth of the exchange when the caller does not fix it: slices of at least 2048
    columns, at most 4.  Narrower slices cost more in short launches than their overlap gives
    back (pl32768, one rank's leg 1 of 4096 columns: 0.89 ms in one launch, 1.05 in two, 1.38 in
    four — tools/stage_probe.py).  Every rank derives it from the largest block, so all agree."""
    return int(max(1, min(4, -(-k_dim // world) // 2048)))


def stage_widths(n_cols: int, n_stages: int):
    """Column counts of the stages a rank cuts its n_cols product columns into: equal
    pieces rounded up to 32 columns (whole gather panels), the last one shorter or empty."""
    q = -(-n_cols // n_stages)
    q = -(-q // STAGE_ALIGN) * STAGE_ALIGN
    return [max(0, min(q, n_cols - s * q)) for s in range(n_stages)]


def staged_row_order(k_dim: int, world: int, n_stages: int) -> np.ndarray:
    """Row order of the leg-2 operand when the exchange runs in stages: stage by stage, inside
    a stage rank by rank (what consecutive all_to_all_single calls deliver).  Returns
    perm with perm[k] = row of the operand that holds global row k."""
    widths = [stage_widths(partition(k_dim, world, h)[1] - partition(k_dim, world, h)[0], n_stages)
              for h in range(world)]
    perm = np.empty(k_dim, dtype=np.int64)
    row = 0
    for s in range(n_stages):
        for h in range(world):
            lo = partition(k_dim, world, h)[0] + sum(widths[h][:s])
            n = widths[h][s]
            perm[lo:lo + n] = np.arange(row, row + n)
            row += n
    return perm


class LocalWorld(_product.LocalWorld):
    """P virtual ranks inside this process (P = 1 is the ordinary single-GPU case) WITH the exchanges of the Python
    choreography; ``loop="python"`` (default here): ``fit`` on this world runs ``Solver`` below."""

    def __init__(self, size: int = 1, symmetric_shards: bool = True, leg2_stages: int = 1,
                 exchange_precision: str = "f32", loop: str = "python"):
        """``symmetric_shards``: sharded symmetric updates run leg 2 in its half form when the node
        count allows it (``Side.shard_sym``); False keeps the full form, whose results are bit-equal
        to a single rank's full form.  ``leg2_stages``: the half-form leg 2 (and its exchange) cut into
        that many stages of column tiles, as ``TorchWorld`` does to overlap the second all-to-all.
        ``exchange_precision``: "fp16" rounds what the virtual ranks hand each other the way ``TorchWorld``'s fp16
        wire format does (no bytes to save here: this is how that mode's arithmetic is tested on one GPU)."""
        if exchange_precision not in ("f32", "fp16"):
            raise ValueError("exchange_precision must be 'f32' or 'fp16'")
        if loop not in ("python", "c"):
            raise ValueError("loop must be 'python' or 'c'")
        # "c": the virtual ranks run the sharded loop behind the C ABI (an in-process group of csrc/shardplan.hip — what
        # the ranks of an RCCL world run, with device copies for links); "python": driver.Solver's own choreography
        self.loop = loop
        self.exchange_precision = exchange_precision
        self.size = int(size)
        self.local_ranks = list(range(self.size))
        self.is_root = True
        self.stages = 1
        self.leg2_stages = int(leg2_stages)
        self.symmetric_shards = bool(symmetric_shards)

    def exchange(self, parts):
        """All-to-all of the transposed tiles between the virtual ranks (device copies)."""
        if self.size == 1:
            return                               # recv aliases send
        if self.exchange_precision == "fp16":
            for src in parts:
                src.ops.round_trip_h16(src.send, self.size * src.ncols * (src.mb + src.pad))
        for src in parts:
            for dst in parts:
                w = dst.nrows + src.pad          # padded row of a chunk
                n = src.ncols * w                # tile (columns of src) x (rows of dst)
                src.ops.copy_bytes(dst.recv.ptr + 4 * src.col_lo * w,
                                   src.send.ptr + 4 * dst.rank * src.ncols * (src.mb + src.pad),
                                   4 * n)

    def exchange_mirrors(self, sides):
        """All-to-all of the packed mirrored tiles of a half-form leg 2 (equal chunks; stage by stage when
        the leg was cut: a stage's range of every chunk sits together, ``Side.sh_stages``)."""
        if self.exchange_precision == "fp16":
            for sd in sides:
                sd.ops.round_trip_h16(sd.sh_send, self.size * sd.sh_chunk)
        for st in sides[0].sh_stages or [dict(off=0, chunk=sides[0].sh_chunk)]:
            n = 4 * st["chunk"]
            for src in sides:
                for dst in sides:
                    if src is not dst and n:
                        src.ops.copy_bytes(dst.sh_recv.ptr + 4 * st["off"] + n * src.rank,
                                           src.sh_send.ptr + 4 * st["off"] + n * dst.rank, n)

    def sum_int(self, values):
        return int(sum(values))

    def gather_list(self, per_rank):
        return [per_rank[r] for r in self.local_ranks]

    def gather_columns(self, blocks, n_rows, n_cols):
        if self.size == 1:
            return blocks[0]                     # the one block is the matrix: no host copy
        out = np.empty((n_rows, n_cols), dtype=np.float64)
        for r, blk in blocks.items():
            lo, hi = partition(n_cols, self.size, r)
            out[:, lo:hi] = blk
        return out


class TorchWorld(_product.TorchWorld):
    """One rank per process over torch.distributed (backend "nccl" = RCCL over xGMI on the
    GPU box; "gloo" in the CPU tests) WITH the collectives of the Python choreography."""

    def __init__(self, group=None, stages: int = 0, stage_single_rank: bool = False,
                 handback: str = "root", symmetric_shards="auto", measure_single_rank: bool = False,
                 exchange_precision: str = "f32", loop: str = "python"):
        """``stages``: leg 1 is cut into that many column slices, each exchanged by its own
        all_to_all_single as soon as its kernel has finished, so the transfers over xGMI overlap
        the remaining leg-1 kernels (1 = one exchange after the whole leg; 0 = by the width of a
        rank's column block, see ``auto_stages``).
        ``handback``: who receives the dense result of a fit.  "root" (default): the float32
        column blocks are gathered to rank 0 on the device, put back into the caller's node order
        there and downloaded once; the other ranks' ``fit`` returns None.  "all": every rank gets
        the full float64 matrix (pickled all-gather: N^2 x 8 B x P per node — small N only).
        ``fit(top_k=k)`` hands k columns per row to every rank either way.
        ``symmetric_shards``: as for ``LocalWorld``; "auto" (default): the half form trades 50 % more bytes on
        the links for 30-35 % less compute per rank, which pays once a rank spreads its exchange over enough
        xGMI links — so the first solver built on a world of several ranks times one update in
        each form on the real links and every rank adopts the faster one (``resolve_shard_form``; below
        ``MEASURE_FORM_FROM_N`` nodes, or where the half form does not apply, the rule of thumb "from
        ``HALF_FORM_FROM`` ranks on" decides)."""
        import torch.distributed as dist
        if handback not in ("root", "all"):
            raise ValueError("handback must be 'root' or 'all'")
        if loop not in ("auto", "c", "python"):
            raise ValueError("loop must be 'auto', 'c' or 'python'")
        # which choreography a fit on this world runs: "auto" = the sharded loop behind the C ABI (csrc/shardplan.hip over
        # the library's own RCCL communicator) on several RCCL ranks wherever it applies, this module's Solver otherwise
        # (gloo, asymmetric priors, GEMM modes); "c" asks for the C loop also in a one-rank RCCL world (how its RCCL path
        # is exercised on one GPU); "python" keeps the Solver
        self.loop = loop
        if exchange_precision not in ("f32", "fp16"):
            raise ValueError("exchange_precision must be 'f32' or 'fp16'")
        # "fp16": both all-to-alls move fp16 (value x 2^14) instead of f32 — HALF the bytes on the xGMI links, where
        # the exchanges, not the kernels, set the pace of a sharded update from N = 65536 on; the
        # kernels on both sides stay f32, the transposed product takes one fp16 rounding per update: outside the
        # 1e-5 parity bar, never the default (BASELINE config 5's "reduced precision" on the links)
        self.exchange_precision = exchange_precision
        self.handback = handback
        self.dist = dist
        self.group = group
        self.size = dist.get_world_size(group)
        # ("force": also in a one-rank world — how the half-form path, RCCL call included, is
        # exercised on a single GPU)
        # "auto" stays unresolved until a solver is built on this world: with more than one rank the solver
        # TIMES one update in each form and all ranks adopt the faster one (resolve_shard_form)
        self.symmetric_shards = (symmetric_shards if symmetric_shards in ("auto", "force")
                                 else bool(symmetric_shards))
        self.form_measured = None        # filled by resolve_shard_form: {"half_ms", "full_ms", "chosen"}
        # (tests: run the measurement in a one-rank world too — how its RCCL calls are exercised on one GPU)
        self.measure_single_rank = bool(measure_single_rank)
        self.rank = dist.get_rank(group)
        self.local_ranks = [self.rank]
        self.is_root = self.rank == 0
        # (a one-rank world stages only on request: that is how the path is exercised on one GPU)
        self.stages = max(0, int(stages)) if (self.size > 1 or stage_single_rank) else 1
        self.leg2_stages = 0               # 0: the half-form leg 2 is cut like leg 1 (Side.n_stages)

    def close(self):
        """Destroy the library's own RCCL communicator of this world, if a fit over the C loop made one (cshard.py)."""
        comm = getattr(self, "_c_comm", None)
        if comm is not None:
            self._c_comm = None
            try:
                from simrank_amd import _lib
                _lib.load().simrank_comm_destroy(comm)
            except Exception:
                pass

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def stream_ordered(self):
        """RCCL collectives can be ordered on the engine's stream: no host synchronisation
        between the legs and the exchange (gloo works on host tensors and needs the syncs)."""
        import os
        if os.environ.get("SIMRANK_HOST_SYNC_EXCHANGE") == "1":      # debugging aid: host-driven pipeline
            return False
        return self.dist.get_backend(self.group) == "nccl"

    def _a2a(self, wire, recv_t, send_t, roff, rn, soff, sn, out_splits=None, in_splits=None, async_op=False):
        """all_to_all_single of send_t[soff:soff+sn] into recv_t[roff:roff+rn], through the fp16 wire buffers when
        there are any.  -> (work or None, what to run once the data has arrived)."""
        if wire is None:
            w = self.dist.all_to_all_single(recv_t[roff:roff + rn], send_t[soff:soff + sn], out_splits, in_splits,
                                            group=self.group, async_op=async_op)
            return w, None
        wire.pack(soff, sn)
        if not self.stream_ordered:
            wire.ops.synchronize()                # (the collective reads the fp16 buffer from another stream / the host)
        twice = lambda v: None if v is None else [2 * int(e) for e in v]
        w = self.dist.all_to_all_single(_wire_view(wire.recv_h[roff:roff + rn]), _wire_view(wire.send_h[soff:soff + sn]),
                                        twice(out_splits), twice(in_splits), group=self.group, async_op=async_op)
        return w, (lambda: wire.unpack(roff, rn))

    def begin_stage(self, x, st):
        """Stream-ordered pipeline: called right after the kernels of one stage of leg 1 were
        queued.  The all-to-all of that stage is issued at once; RCCL's stream waits for what is
        on the engine's stream so far (this stage) and runs beside the kernels queued next."""
        import torch
        # every rank joins every stage's collective, also with all-zero splits (a rank that owns an
        # empty block at small N): skipping it on local data would leave the peers waiting
        with torch.cuda.stream(x.ops.torch_stream()):
            st["work"], st["post"] = self._a2a(x.wire, x.recv_t, x.send_t, st["recv_off"], sum(st["out_splits"]),
                                               st["send_off"], sum(st["in_splits"]), st["out_splits"], st["in_splits"],
                                               async_op=True)

    def exchange(self, parts):
        (x,) = parts
        if self.stream_ordered:
            import torch
            with torch.cuda.stream(x.ops.torch_stream()):
                if x.stages is not None:
                    for st in x.stages:               # issued by begin_stage during leg 1
                        w = st.pop("work", None)
                        post = st.pop("post", None)
                        if w is not None:
                            w.wait()                  # the engine's stream waits, not the host
                        if post is not None:
                            post()
                    return
                span = lambda n, h: partition(n, self.size, h)[1] - partition(n, self.size, h)[0]
                in_splits = [x.ncols * (span(x.row_dim, h) + x.pad) for h in range(self.size)]
                out_splits = [span(x.col_dim, h) * (x.nrows + x.pad) for h in range(self.size)]
                _, post = self._a2a(x.wire, x.recv_t, x.send_t, 0, sum(out_splits), 0, sum(in_splits), out_splits, in_splits)
                if post is not None:
                    post()
            return
        if x.stages is not None:
            works = []
            for st in x.stages:
                x.ops.event_synchronize(st["event"])      # this slice's kernel has finished
                # (zero-size splits are legal; every rank must issue every stage's collective)
                works.append(self._a2a(x.wire, x.recv_t, x.send_t, st["recv_off"], sum(st["out_splits"]), st["send_off"],
                                       sum(st["in_splits"]), st["out_splits"], st["in_splits"], async_op=True))
            for w, _ in works:
                w.wait()
            x.ops.collective_done()
            for _, post in works:
                if post is not None:
                    post()
            return
        x.ops.synchronize()                      # the producing leg finished on the engine's stream
        span = lambda n, h: partition(n, self.size, h)[1] - partition(n, self.size, h)[0]
        in_splits = [x.ncols * (span(x.row_dim, h) + x.pad) for h in range(self.size)]
        out_splits = [span(x.col_dim, h) * (x.nrows + x.pad) for h in range(self.size)]
        _, post = self._a2a(x.wire, x.recv_t, x.send_t, 0, sum(out_splits), 0, sum(in_splits), out_splits, in_splits)
        x.ops.collective_done()
        if post is not None:
            post()

    def begin_mirror_stage(self, sd, st):
        """Stream-ordered pipeline of exchange 2: called right after the kernel of one stage of the half-form
        leg 2 was queued; that stage's mirrored tiles leave while the next stage computes."""
        import torch
        n = self.size * st["chunk"]
        with torch.cuda.stream(sd.ops.torch_stream()):
            st["work"], st["post"] = self._a2a(sd.sh_wire, sd.sh_recv_t, sd.sh_send_t, st["off"], n, st["off"], n,
                                               async_op=True)

    def exchange_mirrors(self, sides):
        """All-to-all of the packed mirrored tiles of a half-form leg 2: equal chunks, the one a
        rank addresses to itself is empty on purpose (its own mirrors were stored in place).  When the leg
        was cut into stages the all-to-alls were issued stage by stage during the leg (stream-ordered
        worlds) and this only makes the engine's stream wait for them."""
        (sd,) = sides
        stages = sd.sh_stages
        if self.stream_ordered:
            import torch
            with torch.cuda.stream(sd.ops.torch_stream()):
                if stages is None:
                    _, post = self._a2a(sd.sh_wire, sd.sh_recv_t, sd.sh_send_t, 0, sd.sh_recv_t.numel(), 0,
                                        sd.sh_send_t.numel())
                    if post is not None:
                        post()
                    return
                for st in stages:
                    w, post = st.pop("work", None), st.pop("post", None)
                    if w is None:                      # (not issued during the leg)
                        n = self.size * st["chunk"]
                        w, post = self._a2a(sd.sh_wire, sd.sh_recv_t, sd.sh_send_t, st["off"], n, st["off"], n,
                                            async_op=True)
                    w.wait()                           # the engine's stream waits, not the host
                    if post is not None:
                        post()
            return
        sd.ops.synchronize()
        posts = []
        if stages is None:
            posts.append(self._a2a(sd.sh_wire, sd.sh_recv_t, sd.sh_send_t, 0, sd.sh_recv_t.numel(), 0,
                                   sd.sh_send_t.numel())[1])
        else:
            for st in stages:
                n = self.size * st["chunk"]
                posts.append(self._a2a(sd.sh_wire, sd.sh_recv_t, sd.sh_send_t, st["off"], n, st["off"], n)[1])
        sd.ops.collective_done()
        for post in posts:
            if post is not None:
                post()

    def sum_changed(self, ops, active=True):
        """Global convergence count of the update just queued, stream-ordered: the striped
        counters are summed and all-reduced on the engine's stream, one read-back (the only host
        synchronisation of an update)."""
        import torch
        t = ops.counter_tensor()
        with torch.cuda.stream(ops.torch_stream()):
            s = t.sum().reshape(1) if active else torch.zeros(1, dtype=torch.int64, device=t.device)
            self.dist.all_reduce(s, group=self.group)
            return int(s.item())

    def sum_int(self, values):
        import torch
        (v,) = values
        t = torch.tensor([v], dtype=torch.int64, device=self._dev())
        self.dist.all_reduce(t, group=self.group)
        return int(t.item())

    def max_float(self, v: float) -> float:
        import torch
        t = torch.tensor([v], dtype=torch.float64, device=self._dev())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return float(t.item())

    def barrier(self):
        self.dist.barrier(group=self.group)

    def _dev(self):
        import torch
        if self.dist.get_backend(self.group) == "nccl":
            return torch.device("cuda", torch.cuda.current_device())
        return torch.device("cpu")

    def gather_list(self, per_rank):
        (mine,) = per_rank.values()
        parts = [None] * self.size
        self.dist.all_gather_object(parts, mine, group=self.group)
        return parts

    def gather_columns(self, blocks, n_rows, n_cols):
        (blk,) = blocks.values()
        parts = [None] * self.size
        self.dist.all_gather_object(parts, blk, group=self.group)
        return np.concatenate(parts, axis=1)

    def gather_to_root(self, ops, src, n, rows, col_inv):
        """Dense hand-back to rank 0: ``src`` is this rank's n x L float32 block (solver order);
        its rows are taken in the order ``rows`` (an index vector or None) while it is copied into
        a padded n x ceil(n/P) send buffer; one ``gather`` of those device buffers; on the root
        the blocks are laid side by side, the columns put into the caller's order (``col_inv``)
        and the matrix downloaded once as float64.  Other ranks return None."""
        import torch
        P, mb = self.size, -(-n // self.size)
        send = ops.exchange_buffer(n * mb)
        if src.cols:
            ops.permute(src, ops.matrix(n, src.cols, ld=mb, external=send), rows, None)
        ops.synchronize()
        big = ops.exchange_buffer(P * n * mb) if self.is_root else None
        self.dist.gather(send, list(big.view(P, n * mb)) if self.is_root else None,
                         dst=self.dist.get_global_rank(self.group, 0) if self.group else 0,
                         group=self.group)
        if not self.is_root:
            import warnings
            warnings.warn("TorchWorld(handback='root'): only rank 0 receives the similarity matrix, fit() returns "
                          "None on this rank (pass handback='all', or fit(top_k=k), to get results on every rank)",
                          RuntimeWarning, stacklevel=4)       # (shown once per call site by the warnings filter)
            return None
        side = big.view(P, n, mb).permute(1, 0, 2).reshape(n, P * mb).contiguous()   # [i][h*mb + j]
        if side.is_cuda:
            torch.cuda.current_stream(side.device).synchronize()
        del big
        full = ops.matrix(n, n, ld=P * mb, external=side)
        if col_inv is not None:
            final = ops.matrix(n, n)
            ops.permute(full, final, None, col_inv)
            out = ops.download_f64(final)
            final.free()
        else:
            out = ops.download_f64(full)
        return out


# --------------------------------------------------------------------------------------
# one side of an update, for one (virtual) rank
# --------------------------------------------------------------------------------------
class Side:
    def __init__(self, ops, spec: SideSpec, rank: int, world: int, mode: str, torch_buffers: bool,
                 stages: int = 1, blocked: bool = False, shard_symmetric: bool = True, leg2_stages: int = 1,
                 wire_fp16: bool = False):
        self.ops, self.spec, self.rank, self.world, self.mode = ops, spec, rank, world, mode
        self.wire_fp16 = bool(wire_fp16) and torch_buffers       # fp16 shadows of the exchange buffers (TorchWorld)
        self.sh_wire = None
        self.blocked = blocked         # single rank, gather legs: every matrix panel-blocked
        self.sdtype = np.float16 if spec.storage == "fp16" else np.float32    # S and the transposed product
        csr = spec.csr
        if stages == 0:
            stages = auto_stages(csr.n_cols, world)
        self.n_stages = stages if (torch_buffers and mode == "sparse") else 1
        self.M, self.K = csr.n_rows, csr.n_cols
        self.m_lo, self.m_hi = partition(self.M, world, rank)
        self.k_lo, self.k_hi = partition(self.K, world, rank)
        self.Lm, self.Lk = self.m_hi - self.m_lo, self.k_hi - self.k_lo
        self.mb = -(-self.M // world)
        knobs = {}
        if spec.storage == "fp16" and hasattr(ops, "get_tuning"):
            knobs["fuse_unit"] = 1 << 20               # (half.hip runs whole blocks: no units whose sums meet in memory)
        if spec.storage == "fp16" and hasattr(ops, "get_tuning") and ops.get_tuning("fuse_min") == 3:
            # one fp16 MFMA term instead of three bf16 ones, but each operand segment serves 64 columns, so the
            # gathers got cheaper still: the break-even moves up by one (measured: 4 is 3 % faster than 3, 2 is
            # 12 % slower).  Only while the knob is at its default.
            knobs["fuse_min"] = 4
            if ops.get_tuning("fuse_group") == 3:      # (and groups of four: 7 % faster than three at config 5 here,
                knobs["fuse_group"] = 4                #  where three is 5 % faster than four in f32)
        elif (spec.storage == "fp16" and hasattr(ops, "get_tuning") and ops.get_tuning("fuse_min") == 0
              and ops.get_tuning("fuse_pays") < 0):
            knobs["fuse_pays"] = 256                   # (round 6: dense sets by quads that pay — the same step up, as plan.hip)
            if ops.get_tuning("fuse_group") == 3:
                knobs["fuse_group"] = 4
        # SimRank++ on the graph's own pattern (the directed classes): the evidence counts are queued while the graph's
        # plans are still being built on the host (engine.Graph(counting=)); counted below otherwise
        self.ev = None
        counted = False
        gkw = dict(knobs=knobs) if knobs else {}
        if (spec.evidence_from is csr and self.M == self.K and self.m_hi > self.m_lo and
                getattr(ops, "supports_counting_graph", False)):
            self.ev = (ops.matrix(self.M, self.m_hi - self.m_lo, np.uint8, blocked=True) if blocked
                       else ops.matrix(self.M, self.m_hi - self.m_lo, np.uint8))
            gkw["counting"] = (self.ev, self.m_lo)
            counted = True
        self.graph = ops.graph(csr, spec.rowscale, dense_terms=spec.dense_terms, **gkw)
        self.symmetric = spec.symmetric
        self.x1 = self.x2 = None
        self.broadcast_error = None
        self.ev_live, self.restrict = 1.0, False
        # `_converged` (SimRank.py:74-77) uses its sum as a truth value: the fused test may stop comparing
        # at the first difference (epilogue count_any).  True: exact counts (Solver.exact_count)
        self.exact_count = False
        if mode == "sparse":
            # exchange 1: leg-1 product (M rows x my Lk of K columns) -> K x Lm
            self.x1 = self._xfer(self.Lk, self.k_lo, self.K, torch_buffers)
            self.send, self.recv = self.x1.send, self.x1.recv
            self.graph2 = self.graph
            if self.n_stages > 1:
                self._plan_stages()
            if not self.symmetric and world > 1:
                # exchange 2: raw leg-2 product (M rows x my Lm of M columns) -> M x Lm
                self.x2 = self._xfer(self.Lm, self.m_lo, self.M, torch_buffers)
        else:                                                    # dense / hybrid: one rank only
            assert world == 1, "dense and hybrid modes run on one rank"
            self.wd = ops.matrix(self.M, self.K)
            ops.densify(self.graph, self.wd)
            self.t = ops.matrix(self.M, self.K)
        # Half-form leg 2 of a sharded symmetric update (simrank_spmm_shard): of the 32 x 32 tiles
        # (shard h, tile i) x (my column tile j) only i <= j is computed, the transposed tiles i < j go to
        # the ranks that own them in a second, half-size all-to-all.  Needs equal shards of whole tiles.
        self.sh_stages = None          # half-form leg 2 in stages: [{tile_lo, tile_hi, off, chunk}], heaviest first
        self.shard_sym = (mode == "sparse" and (world > 1 or shard_symmetric == "force") and
                          self.symmetric and bool(shard_symmetric) and
                          self.M % (32 * world) == 0 and getattr(ops, "supports_shard_symmetric", False) and
                          lean_knobs(ops))
        if self.shard_sym:
            t = self.mb // 32
            self.sh_chunk = max(1, t * (t - 1) // 2 * 1024)
            self.sh_send_t = self.sh_recv_t = None
            if torch_buffers:
                self.sh_send_t = ops.exchange_buffer(world * self.sh_chunk)
                self.sh_recv_t = ops.exchange_buffer(world * self.sh_chunk)
                self.sh_send = ops.matrix(world, self.sh_chunk, ld=self.sh_chunk, external=self.sh_send_t)
                self.sh_recv = ops.matrix(world, self.sh_chunk, ld=self.sh_chunk, external=self.sh_recv_t)
                if self.wire_fp16:
                    self.sh_wire = Wire(ops, self.sh_send_t, self.sh_recv_t)
            else:
                self.sh_send = ops.matrix(world, self.sh_chunk, ld=self.sh_chunk)
                self.sh_recv = ops.matrix(world, self.sh_chunk, ld=self.sh_chunk)
            # The leg (and exchange 2) in stages of column tiles: a column tile j packs j mirrored tiles per
            # source shard, slots j (j - 1) / 2 ..., so tiles [lo, hi) own a contiguous slot range of every
            # chunk; the buffers hold them stage-major (per stage: one piece per rank), the stages are cut at
            # T sqrt(k / S) for equal slot counts and run heaviest (last tiles) first.
            want = leg2_stages if leg2_stages > 0 else self.n_stages
            want = min(want, t // 2)
            if want > 1:
                cuts = sorted({min(t, max(2, int(round(t * (k / want) ** 0.5)))) for k in range(1, want)} | {t})
                lo, self.sh_stages = 0, []
                for hi in cuts:
                    s_lo, s_hi = lo * (lo - 1) // 2, hi * (hi - 1) // 2
                    self.sh_stages.append(dict(tile_lo=lo, tile_hi=hi, off=world * s_lo * 1024,
                                               chunk=(s_hi - s_lo) * 1024))
                    lo = hi
                self.sh_stages.reverse()
        if counted:
            # support density of E decides between the two instantiations of leg 2
            self.ev_live = ops.evidence_live_fraction(self.ev)
            self.restrict = self.ev_live < RESTRICT_BELOW
        elif spec.evidence_from is not None:
            ev = spec.evidence_from
            self.ev = (ops.matrix(self.M, self.Lm, np.uint8, blocked=True) if self.blocked
                       else ops.matrix(self.M, self.Lm, np.uint8))
            if ev.n_rows == self.M:
                eg = self.graph if ev is csr else ops.graph(ev)
                ops.evidence_counts(eg, self.m_lo, self.ev)
                # support density of E decides between the two instantiations of leg 2
                self.ev_live = ops.evidence_live_fraction(self.ev)
                self.restrict = self.ev_live < RESTRICT_BELOW
            elif ev.n_rows == 1:
                # quirk Q2 with a single group-1 node: NumPy broadcasts the 1 x 1 Evidence_N1 over
                # the n2 x n2 update (SimRank.py:423), i.e. one count gates every element
                live = ev.rowscale[0] > 0
                cnt = min(255, int(ev.rowptr[1] - ev.rowptr[0])) if live else 0
                ops.upload(self.ev, np.full((self.M, self.Lm), cnt, dtype=np.uint8))
            else:
                # quirk Q2: the reference multiplies the n2 x n2 update by the n1 x n1 Evidence_N1 and
                # NumPy raises when that update RUNS (not at set-up: iterations=0 or eps >= 1 return
                # the identity matrices); `leg2` raises it at the first group-2 update
                self.broadcast_error = ValueError(
                    f"operands could not be broadcast together with shapes "
                    f"({ev.n_rows},{ev.n_rows}) ({self.M},{self.M}) ")
        self.ap = None
        if spec.apriori is not None:
            a = np.asarray(spec.apriori)
            if a.shape != (self.M, self.M):
                raise ValueError(f"operands could not be broadcast together with shapes "
                                 f"({self.M},{self.M}) {a.shape} ")
            if self.sdtype == np.float16 and not (np.isfinite(a).all() and float(np.abs(a).max()) < 3.99):
                # (fp16-held matrices store value x 2^14: anything from 4 up is out of fp16's range)
                raise ValueError("storage_precision='fp16' needs prior values below 4 in magnitude")
            self.ap = ops.matrix(self.M, self.Lm, blocked=True) if self.blocked else ops.matrix(self.M, self.Lm)
            ops.upload(self.ap, a[:, self.m_lo:self.m_hi].astype(np.float32))

    def _xfer(self, ncols, col_lo, col_dim, torch_buffers) -> Xfer:
        o = self.ops
        x = Xfer(o, self.rank, None, None, ncols, col_lo, col_dim, self.M, self.mb, self.Lm)
        if self.world == 1 and not torch_buffers:
            # one rank: recv aliases send (pitched rows, or panel-blocked with the solver's matrices)
            x.send = x.recv = (o.matrix(col_dim, self.M, self.sdtype, blocked=True) if self.blocked
                               else o.matrix(col_dim, self.M))
            if self.sdtype == np.float16:
                x.send.scale = o.HALF_SCALE
            return x
        x.pad = row_pad(self.mb)
        send_ld = self.M + self.world * x.pad                    # floats per column, all chunks
        recv_ld = max(1, self.Lm + x.pad)
        if torch_buffers:                                        # chunks of (M_h + pad)-float rows
            x.send_t = o.exchange_buffer(send_ld * ncols)
            x.recv_t = o.exchange_buffer(col_dim * recv_ld)
            x.send = o.matrix(max(1, ncols), send_ld, ld=send_ld, external=x.send_t)
            x.recv = o.matrix(col_dim, self.Lm, ld=recv_ld, external=x.recv_t)
            if self.wire_fp16:
                x.wire = Wire(o, x.send_t, x.recv_t)
        else:
            x.send = o.matrix(max(1, ncols), send_ld, ld=send_ld)
            x.recv = o.matrix(col_dim, self.Lm, ld=recv_ld)
        return x

    def _plan_stages(self):
        """Cut exchange 1 into stages (see TorchWorld): per stage the slice of my columns, where
        its chunks sit in the send buffer, where the received slices land, and the split
        sizes.  Leg 2 then gathers from an operand whose rows are in stage order, so it gets
        the graph with its columns renamed accordingly."""
        P, S, x = self.world, self.n_stages, self.x1
        span = lambda n, h: partition(n, P, h)[1] - partition(n, P, h)[0]
        widths = [stage_widths(span(self.K, h), S) for h in range(P)]
        mine = widths[self.rank]
        chunk_rows = [span(self.M, h) + x.pad for h in range(P)]     # floats per column, per peer
        x.stages, send_off, recv_off, col0 = [], 0, 0, 0
        for s in range(S):
            in_splits = [mine[s] * chunk_rows[h] for h in range(P)]
            out_splits = [widths[h][s] * (self.Lm + x.pad) for h in range(P)]
            x.stages.append(dict(x_col0=col0, ncols=mine[s], send_off=send_off, recv_off=recv_off,
                                 in_splits=in_splits, out_splits=out_splits, event=self.ops.event()))
            send_off += sum(in_splits)
            recv_off += sum(out_splits)
            col0 += mine[s]
        perm = staged_row_order(self.K, P, S)
        self.graph2 = self.ops.graph(permute_columns(self.spec.csr, perm), self.spec.rowscale,
                                     dense_terms=self.spec.dense_terms)

    # S_in: K x Lk block of the input similarity
    def leg1(self, S_in, stage_hook=None):
        """``stage_hook(xfer, stage)``: called after the kernels of each stage of a pipelined
        exchange were queued (a stream-ordered world issues that stage's all-to-all there);
        without it an event marks the end of the stage for the host to wait on."""
        o = self.ops
        if self.mode == "sparse" and self.x1.stages is not None:
            for st in self.x1.stages:
                if st["ncols"]:
                    o.spmm(self.graph, S_in, self.send, n_cols=st["ncols"], transpose_out=True,
                           t_block=self.mb, t_pad=self.x1.pad, x_col0=st["x_col0"],
                           y_offset=st["send_off"])
                if stage_hook is not None:
                    stage_hook(self.x1, st)
                else:
                    o.record(st["event"])
        elif self.mode == "sparse":
            if self.Lk:
                o.spmm(self.graph, S_in, self.send, n_cols=self.Lk, transpose_out=True,
                       t_block=self.mb, t_pad=self.x1.pad)
        elif self.mode == "hybrid":
            o.spmm(self.graph, S_in, self.t)                          # T = W.S, plain store
        else:
            o.gemm_nt(self.wd, S_in, self.t, self.M, self.K, self.K)  # T = Wd.S^T, S symmetric

    def _ep(self, S_prev, eps):
        return dict(coef=self.spec.coef, evidence=self.ev, apriori=self.ap, lbd=self.spec.lbd,
                    previous=S_prev, eps=eps, diag_col0=self.m_lo, set_diag=True,
                    restrict_support=self.restrict, count_any=not self.exact_count)

    def leg2(self, S_prev, S_out, eps, mirror_hook=None):
        """Symmetric iterates: S_out = W . Tt with the fused epilogue.
        Otherwise only the raw product, stored transposed (what W . Tt yields is the
        TRANSPOSE of the wanted block); ``finish`` applies the epilogue after exchange 2."""
        o = self.ops
        if self.broadcast_error is not None:
            raise self.broadcast_error
        if not self.Lm:
            return
        if self.mode != "sparse":
            o.gemm_nt(self.t, self.wd, S_out, self.M, self.M, self.K, epilogue=self._ep(S_prev, eps))
        elif self.shard_sym and self.sh_stages is not None:
            ep = self._ep(S_prev, eps)
            for k, st in enumerate(self.sh_stages):
                o.spmm_shard_stage(self.graph2, self.recv, S_out, ep, self.rank, self.world, self.sh_send,
                                   st["off"], st["chunk"], st["tile_lo"], st["tile_hi"], k == 0)
                if mirror_hook is not None:
                    mirror_hook(self, st)          # this stage's mirrored tiles leave while the next computes
        elif self.shard_sym:
            o.spmm_shard(self.graph2, self.recv, S_out, self._ep(S_prev, eps), self.rank, self.world,
                         self.sh_send, self.sh_chunk)
        elif self.symmetric:
            ep = self._ep(S_prev, eps)
            # one rank holds the whole symmetric matrix: upper triangle + mirror image
            ep["symmetric"] = self.world == 1
            o.spmm(self.graph2, self.recv, S_out, n_cols=self.Lm, epilogue=ep)
        elif self.world == 1:
            o.spmm(self.graph2, self.recv, S_out, n_cols=self.Lm, transpose_out=True)
        else:
            o.spmm(self.graph2, self.recv, self.x2.send, n_cols=self.Lm, transpose_out=True,
                   t_block=self.mb, t_pad=self.x2.pad)

    def unpack(self, S_out):
        """Half-form leg 2, after the exchange of the mirrored tiles: put the received ones in place."""
        if self.sh_stages is None:
            self.ops.shard_unpack(S_out, self.sh_recv, self.sh_chunk, self.rank, self.world, self.M)
            return
        for st in self.sh_stages:
            if st["chunk"]:
                self.ops.shard_unpack_stage(S_out, self.sh_recv, st["off"], st["chunk"], self.rank, self.world,
                                            self.M, st["tile_lo"], st["tile_hi"])

    def finish(self, S_prev, S_out, eps):
        """Second half of an update with asymmetric iterates: the stand-alone epilogue."""
        if self.symmetric or not self.Lm:
            return
        raw = S_out if self.world == 1 else self.x2.recv
        self.ops.epilogue_apply(raw, S_out, self.M, self.Lm, self._ep(S_prev, eps))


def choose_mode(mode: str, csrs, world: int, symmetric: bool = True) -> str:
    if mode not in ("auto", "sparse", "dense", "hybrid"):
        raise ValueError(f"mode must be auto, sparse, dense or hybrid, not {mode!r}")
    if not symmetric:
        return "sparse"       # the NT GEMM legs rely on S == S^T; the gather legs do not
    if mode == "auto":
        # The gather legs serve every density since the dense blocks of a pattern go to the
        # matrix cores inside simrank_spmm (bf16x3 MFMA, csrc/blockdense.hip): measured 2-3x
        # faster than the f32 MFMA GEMM legs on dense graphs too (profiles/modes_r01.log).
        # "dense" / "hybrid" remain as explicit choices.
        return "sparse"
    if mode != "sparse" and world != 1:
        raise ValueError("dense and hybrid modes are single-rank; use mode='sparse' when sharded")
    return mode


def resolve_shard_form(make_ops, world, specs, mode, reorder=True):
    """``TorchWorld(symmetric_shards="auto")``: which form of the sharded leg 2 this world runs.

    The half form computes a third less per rank and sends half as much again over the links; which one is
    faster depends on what RCCL's all-to-all reaches on the node's xGMI links, so it is measured: a solver in
    each form, one warm-up update and ``steps`` timed ones bracketed by barriers, the MAX over the ranks of
    each time, the smaller one wins — every rank sees the same two numbers and takes the same decision.
    Where the half form cannot run (asymmetric prior, N not a multiple of 32 x ranks, dense modes) or the
    problem is small, ``HALF_FORM_FROM`` decides without a measurement.  Leaves the result in
    ``world.symmetric_shards`` (bool) and the timings in ``world.form_measured``."""
    import time
    if getattr(world, "symmetric_shards", None) != "auto":
        return
    n_min = min(s.csr.n_rows for s in specs)
    single = world.size == 1 and getattr(world, "measure_single_rank", False)
    applies = ((world.size > 1 or single) and mode in ("auto", "sparse") and all(s.symmetric for s in specs) and
               all(s.csr.n_rows % (32 * world.size) == 0 for s in specs))
    if not applies or n_min < MEASURE_FORM_FROM_N:
        world.symmetric_shards = world.size >= HALF_FORM_FROM
        world.form_measured = None
        return
    times, failed = {}, None
    try:
        for half in (True, False):
            world.symmetric_shards = ("force" if single else True) if half else False
            solver = None
            try:
                solver = Solver(make_ops, world, specs, mode, reorder)
                solver.reset()
                solver.step(0.0)
            except Exception as e:                       # (e.g. out of memory on one rank: every rank must learn of it)
                failed = e
            # a rank that failed still joins the collectives its peers are in, with a time no measurement reaches
            bad = world.max_float(1.0 if failed is not None else 0.0)
            if bad:
                if solver is not None:
                    solver.release()
                break
            world.barrier()
            t0 = time.perf_counter()
            for _ in range(2):
                solver.step(0.0)
            world.barrier()
            times[half] = world.max_float((time.perf_counter() - t0) / 2)
            solver.release()
            del solver
    finally:
        if len(times) == 2:
            world.symmetric_shards = (("force" if single else True) if times[True] < times[False] else False)
            world.form_measured = {"half_ms": times[True] * 1e3, "full_ms": times[False] * 1e3,
                                   "chosen": "half" if times[True] < times[False] else "full"}
        else:
            # the measurement did not complete on some rank: the rule of thumb, the same on every rank
            world.symmetric_shards = world.size >= HALF_FORM_FROM
            world.form_measured = None
    if failed is not None:
        raise failed


# --------------------------------------------------------------------------------------
# solvers
# --------------------------------------------------------------------------------------
def length_order(csr: CSR) -> np.ndarray:
    """Rows of ``csr`` by ascending number of entries (stable)."""
    return np.argsort(np.diff(csr.rowptr), kind="stable")


def dealt_order(order: np.ndarray, world: int) -> np.ndarray:
    """Ascending-length order dealt to ``world`` shards in tiles of 32 nodes: tile t goes to shard
    t mod world, so every shard holds the same mix of short and long rows, ascending inside.  That is
    what makes the half-form sharded leg 2 (``Side.shard_sym``) — tile i of any shard against column
    tile j of mine only when i <= j — cut every rank's gathers the way the triangle does on one
    rank, and balances leg 1 over the ranks as well.  Unchanged when the tiles do not divide evenly.
    (The reference iterates in list(self.Nodes) order, SimRank.py:43/:141; the update is equivariant
    under renaming, results are handed back in that order.)"""
    n = order.size
    if world <= 1 or n % (32 * world):
        return order
    # (whole 128-row blocks when they divide evenly: a block of the matrix-core part then holds
    # consecutive rows of the ascending order, as on one rank — leg 1 at P = 8: 0.96 -> 0.90 ms)
    unit = DEAL_UNIT if n % (DEAL_UNIT * world) == 0 else 32
    return order.reshape(n // (unit * world), world, unit).transpose(1, 0, 2).reshape(-1)


def reorder_specs(specs, deal: int = 1):
    """The update is equivariant under a renaming of the nodes, so the solver is free to pick
    the order it iterates in: every node set goes by ASCENDING ROW LENGTH of its graph.
    The rows a wave gathers together then have equal lengths (no masked gathers), and in the
    upper-triangle form of leg 2 the long rows own the short column ranges — on a power-law
    graph a quarter of the gathers of the natural order (HISTORY.md §4.7).
    Returns (specs in the new order, [order of node set j]); results are handed back in the
    caller's order by ``Solver.result`` / ``topk`` / ``evidence``."""
    orders = [dealt_order(length_order(sp.csr), deal) for sp in specs]
    out = []
    for j, sp in enumerate(specs):
        cols = orders[0] if len(specs) == 1 else orders[1 - j]
        csr = relabel(sp.csr, orders[j], cols)
        ev = sp.evidence_from
        if ev is sp.csr:
            ev = csr
        elif ev is not None and ev.n_rows == sp.csr.n_rows:
            ev = relabel(ev, orders[j], None)          # common-neighbour counts ignore column names
        ap = sp.apriori
        if ap is not None and np.shape(ap) == (sp.csr.n_rows, sp.csr.n_rows):
            ap = np.asarray(ap)[orders[j]][:, orders[j]]
        out.append(replace(sp, csr=csr, rowscale=np.asarray(sp.rowscale)[orders[j]],
                           evidence_from=ev, apriori=ap))
    return out, orders


class Solver:
    """Runs the reference loop for one or two coupled similarity matrices.

    ``sides``: [spec] for the directed classes (S <- f(S)); [spec1, spec2] for the bipartite
    ones (S1 <- f1(S2), then S2 <- f2(S1_new)).  ``make_ops(virtual_rank)`` returns the
    kernel launcher of a virtual rank.  Inside, nodes are in ``reorder_specs`` order
    (``reorder=False`` keeps the caller's); everything handed back is in the caller's.
    """

    def __init__(self, make_ops, world, specs, mode="auto", reorder=True):
        resolve_shard_form(make_ops, world, specs, mode, reorder)     # ("auto" worlds only, once)
        self.world = world
        self.order = [None] * len(specs)
        if reorder:
            sym = getattr(world, "symmetric_shards", True) and all(s.symmetric for s in specs)
            specs, self.order = reorder_specs(specs, world.size if sym else 1)
        self.inv = [None if o is None else np.argsort(o) for o in self.order]
        self._index = {}
        self.specs = specs
        self.bipartite = len(specs) == 2
        if mode == "auto" and any(s.storage == "fp16" for s in specs):
            mode = "sparse"              # fp16-held matrices exist for the gather legs only
        self.mode = choose_mode(mode, [s.csr for s in specs], world.size,
                                all(s.symmetric for s in specs))
        torch_buffers = isinstance(world, TorchWorld)
        self.ops = {r: make_ops(r) for r in world.local_ranks}
        # One rank running the gather legs keeps S, the transposed product, evidence and prior
        # PANEL-BLOCKED (engine.Matrix): a panel's slice of the operand is then contiguous instead of
        # one 128-byte segment every 128 KiB, which is what the gathers need at N >= 16384 (TLB reach;
        # HISTORY.md §4.9).  Sharded ranks hold N x N/P blocks whose rows are close together already.
        self.blocked = (world.size == 1 and not torch_buffers and self.mode == "sparse" and
                        all(getattr(o, "supports_blocked", False) and lean_knobs(o) for o in self.ops.values()))
        # fp16 storage (SideSpec.storage): the panel-blocked single-rank gather solver only, symmetric iterates
        self.storage = specs[0].storage
        if any(s.storage != self.storage for s in specs):
            raise ValueError("every side must use the same storage precision")
        if self.storage == "fp16":
            if not (self.blocked and all(s.symmetric for s in specs) and
                    all(getattr(o, "supports_half_storage", False) for o in self.ops.values())):
                raise ValueError("storage_precision='fp16' needs one GPU, the gather legs (mode 'sparse' or 'auto' "
                                 "choosing it), default kernel knobs and a symmetric prior")
        elif self.storage != "f32":
            raise ValueError(f"storage {self.storage!r}")
        self.sdtype = np.float16 if self.storage == "fp16" else np.float32
        self.sides = [{r: Side(self.ops[r], sp, r, world.size, self.mode, torch_buffers,
                               getattr(world, "stages", 1), self.blocked,
                               getattr(world, "symmetric_shards", True), getattr(world, "leg2_stages", 1),
                               getattr(world, "exchange_precision", "f32") == "fp16")
                       for r in world.local_ranks} for sp in specs]
        # similarity matrices: index j -> size n_j; S_j is n_j x (block of n_j), ping-pong
        if self.bipartite:
            self.n = [specs[0].csr.n_rows, specs[1].csr.n_rows]
        else:
            self.n = [specs[0].csr.n_rows]
        self.cur, self.nxt = [], []
        for n in self.n:
            c, x = {}, {}
            for r in world.local_ranks:
                lo, hi = partition(n, world.size, r)
                kw = dict(blocked=True) if self.blocked else {}
                c[r] = self.ops[r].matrix(n, hi - lo, self.sdtype, **kw)
                x[r] = self.ops[r].matrix(n, hi - lo, self.sdtype, **kw)
                if self.storage == "fp16":      # stored value = value x 2^14 (fp16's range is too short below)
                    c[r].scale = x[r].scale = self.ops[r].HALF_SCALE
            self.cur.append(c)
            self.nxt.append(x)
        self.events = None
        self.leg_ms = []

    @property
    def exact_count(self):
        """False (default): ``step`` returns 0 exactly when no element moved by more than eps and some
        positive number otherwise — all the reference's loop asks (`_converged`, SimRank.py:74-77,
        :130) — and the kernels stop reading the previous iterate once a difference is known.
        True: the exact number of such elements."""
        return all(sd.exact_count for side in self.sides for sd in side.values())

    @exact_count.setter
    def exact_count(self, value):
        for side in self.sides:
            for sd in side.values():
                sd.exact_count = bool(value)

    def reset(self):
        for j, n in enumerate(self.n):
            for r in self.world.local_ranks:
                lo, _ = partition(n, self.world.size, r)
                self.ops[r].fill_identity(self.cur[j][r], lo)

    def enable_timing(self, steps: int = 0):
        """Record HIP events around every leg on the engine's stream (rank-local).  ``steps``:
        events for that many steps are created now, outside the region being timed."""
        self.events = []
        self._event_pool = {r: [self.ops[r].event() for _ in range(10 * len(self.sides) * steps)]
                            for r in self.world.local_ranks}

    def _timed(self, r, fn, tag):
        if self.events is None:
            return fn()
        o = self.ops[r]
        pool = getattr(self, "_event_pool", {}).get(r, [])
        a = pool.pop() if pool else o.event()
        b = pool.pop() if pool else o.event()
        o.record(a)
        out = fn()
        o.record(b)
        self.events.append((tag, r, a, b))
        return out

    def _update(self, side_idx, in_idx, out_idx, eps, defer=None):
        """One similarity update on every local virtual rank.  Returns 0 when no element moved by more than
        eps and a positive number otherwise — the exact global count of such elements only with
        ``exact_count`` set (the default short-circuit test stops comparing once one has been found).
        ``defer`` (a pinned counter slot, single-rank fused updates only): the count is not read; its copy is
        queued behind the update and ``ops.wait_changed(defer)`` returns it later (``run``)."""
        sides = self.sides[side_idx]
        hook = self.world.begin_stage if getattr(self.world, "stream_ordered", False) else None
        for r in self.world.local_ranks:
            self._timed(r, lambda: sides[r].leg1(self.cur[in_idx][r], hook), f"leg1.{side_idx}")
        local = self.world.local_ranks
        if self.mode == "sparse":
            # (timed on the first local rank's stream: with a stream-ordered world this is the time that
            # stream waits for the all-to-all beyond what leg 1 already hid)
            self._timed(local[0], lambda: self.world.exchange([sides[r].x1 for r in local]),
                        f"exchange1.{side_idx}")
        fused = sides[local[0]].symmetric
        device_sum = fused and hook is not None        # stream-ordered world: reduce on the device
        counts = []
        for r in local:
            if device_sum:
                self.ops[r].counter_tensor()            # (before the epilogue takes its address)
            mhook = getattr(self.world, "begin_mirror_stage", None) if hook is not None else None
            self._timed(r, lambda: sides[r].leg2(self.cur[out_idx][r], self.nxt[out_idx][r], eps, mhook),
                        f"leg2.{side_idx}")
            if fused and not device_sum:   # virtual ranks may share one device counter: read it per launch
                if defer is not None:
                    self.ops[r].fetch_changed(defer)
                else:
                    counts.append(self.ops[r].read_changed() if sides[r].Lm else 0)
        if sides[local[0]].shard_sym:
            # (the convergence counters were read above / are reduced below: the mirrored tiles were
            # counted by the rank that computed them)
            self._timed(local[0], lambda: self.world.exchange_mirrors([sides[r] for r in local]),
                        f"exchange2.{side_idx}")
            for r in local:
                self._timed(r, lambda: sides[r].unpack(self.nxt[out_idx][r]), f"unpack.{side_idx}")
        if not fused:
            if self.world.size > 1:
                self.world.exchange([sides[r].x2 for r in local])
            for r in local:
                sides[r].finish(self.cur[out_idx][r], self.nxt[out_idx][r], eps)
                counts.append(self.ops[r].read_changed() if sides[r].Lm else 0)
        for r in self.world.local_ranks:
            self.cur[out_idx][r], self.nxt[out_idx][r] = self.nxt[out_idx][r], self.cur[out_idx][r]
        if device_sum:
            (r,) = local
            return self.world.sum_changed(self.ops[r], bool(sides[r].Lm))
        if defer is not None:
            return None
        return self.world.sum_int(counts)

    def _can_defer(self):
        """The count of an update may be read one update late: one rank, fused (symmetric) updates whose counter
        this process reads itself, no per-leg timing, an engine with pinned counter slots."""
        if self.world.size != 1 or len(self.world.local_ranks) != 1 or self.events is not None:
            return False
        if max(self.n) >= SPECULATE_BELOW_N:
            return False          # (an update takes milliseconds: the round trip is noise, a dropped update is not)
        if getattr(self.world, "stream_ordered", False):
            return False
        r = self.world.local_ranks[0]
        return (hasattr(self.ops[r], "fetch_changed") and all(s[r].symmetric and not s[r].shard_sym and s[r].Lm for s in self.sides))

    def _step_deferred(self, eps, parity):
        """One loop body queued; -> the pinned slots its counts will land in."""
        if self.bipartite:
            self._update(0, 1, 0, eps, defer=2 * parity)
            self._update(1, 0, 1, eps, defer=2 * parity + 1)
            return (2 * parity, 2 * parity + 1)
        self._update(0, 0, 0, eps, defer=2 * parity)
        return (2 * parity,)

    def _undo_step(self):
        """Drop the loop body queued last (its results sit in the buffers of the iterates before last)."""
        for j in range(len(self.sides)):
            for r in self.world.local_ranks:
                self.cur[j][r], self.nxt[j][r] = self.nxt[j][r], self.cur[j][r]

    def step(self, eps=0.0):
        """One loop body of the reference (both updates for the bipartite classes); the return value as
        ``_update``: 0 = nothing moved by more than eps."""
        if self.bipartite:
            c1 = self._update(0, 1, 0, eps)      # S1 <- f1(S2)           SimRank.py:297-299
            c2 = self._update(1, 0, 1, eps)      # S2 <- f2(S1 new)       SimRank.py:300-302
            return c1 + c2
        return self._update(0, 0, 0, eps)        #                        SimRank.py:138-140

    def run(self, iterations, eps, on_iteration=None, on_converged=None):
        """The loop of SimRank.py:129-140.  Returns k (loop index at which the test passed)
        or None when ``iterations`` updates were applied."""
        self.reset()
        # test at k = 0 compares S_0 = I with S_-1 = 0: the diagonal differs by 1
        changed = sum(self.n) if 1.0 > eps else 0
        if iterations > 1 and changed and self._can_defer():
            # Loop body k + 1 is queued BEFORE the count of body k is read (csrc/plan.hip does the same behind the C
            # ABI): the device never idles while the host learns whether it may go on; when the count says
            # "converged" the speculative body is dropped — it wrote the buffers of the iterates before last.
            o = self.ops[self.world.local_ranks[0]]
            if on_iteration:
                on_iteration(0)
            slots = self._step_deferred(eps, 1)                   # body 1
            for k in range(1, iterations):
                nxt_slots = self._step_deferred(eps, (k + 1) & 1)   # body k + 1, speculative
                if sum(o.wait_changed(s) for s in slots) == 0:
                    self._undo_step()
                    if on_converged:
                        on_converged(k)
                    return k
                if on_iteration:
                    on_iteration(k)
                slots = nxt_slots
            return None
        for k in range(iterations):
            if changed == 0:
                if on_converged:
                    on_converged(k)
                return k
            if on_iteration:
                on_iteration(k)
            changed = self.step(eps)
        return None

    def _index_vector(self, r, key, values):
        """Device copy of an index list, made once per (rank, key)."""
        if (r, key) not in self._index:
            self._index[(r, key)] = self.ops[r].index_vector(values)
        return self._index[(r, key)]

    def result(self, j=0):
        """Full similarity matrix j as float64 on the host, in the caller's node order.  In a
        multi-process world with ``handback="root"`` only rank 0 gets it (None elsewhere)."""
        inv = self.inv[j]
        if getattr(self.world, "handback", "all") == "root" and self.world.size > 1:
            (r,) = self.world.local_ranks
            rows = None if inv is None else self._index_vector(r, ("inv", j), inv)
            return self.world.gather_to_root(self.ops[r], self.cur[j][r], self.n[j], rows, rows)
        blocks = {}
        for r in self.world.local_ranks:
            o, src = self.ops[r], self.cur[j][r]
            tmp = wide = None
            if src.dtype == np.float16:
                src = wide = o.widen(src)
            if self.blocked and hasattr(o, "handback_f64") and src.rows == src.cols:
                # out of the panel-blocked layout and the solver's node order band by band, PCIe and the widening
                # on the host overlapping (csrc/handback.hip)
                rows = None if inv is None else self._index_vector(r, ("inv", j), inv)
                blocks[r] = o.handback_f64(src, rows)
                if wide is not None:
                    wide.free()
                continue
            if self.blocked:
                # out of the panel-blocked layout and the solver's node order in one pass
                rows = None if inv is None else self._index_vector(r, ("inv", j), inv)
                tmp = o.matrix(src.rows, src.cols)
                o.permute(src, tmp, rows, rows)
                src = tmp
            elif inv is not None and src.cols:
                # un-permute on the device into the idle ping-pong partner: rows always, and
                # columns too when this rank holds all of them
                rows = self._index_vector(r, ("inv", j), inv)
                o.permute(src, self.nxt[j][r], rows, rows if self.world.size == 1 else None)
                src = self.nxt[j][r]
            blocks[r] = o.download_f64(src)
            for m in (tmp, wide):
                if m is not None:
                    m.free()
        full = self.world.gather_columns(blocks, self.n[j], self.n[j])
        if inv is not None and self.world.size > 1:
            full = np.ascontiguousarray(full[:, inv])
        return full

    def evidence(self, j=0):
        """Evidence matrix of side j (1 - 0.5**count, SimRank.py:316) as float64 in the
        caller's node order; local shards only (see estimators._lazy_evidence)."""
        inv = self.inv[j]
        blocks = {}
        for r, side in self.sides[j].items():
            o, cnt = side.ops, side.ev
            if (inv is not None or self.blocked) and cnt.cols:
                tmp = o.matrix(cnt.rows, cnt.cols, np.uint8)
                rows = None if inv is None else self._index_vector(r, ("inv", j), inv)
                o.permute(cnt, tmp, rows, rows if self.world.size == 1 else None)
                blocks[r] = 1 - 0.5 ** o.download(tmp).astype(np.float64)
                tmp.free()
            else:
                blocks[r] = 1 - 0.5 ** o.download(cnt).astype(np.float64)
        full = self.world.gather_columns(blocks, self.n[j], self.n[j])
        if inv is not None and self.world.size > 1:
            full = np.ascontiguousarray(full[:, inv])
        return full

    def topk(self, j, k, exclude_diag=True):
        """k most similar columns of every row of similarity matrix j, found on the device
        shard by shard and merged on the host: (column ids [n, k], values [n, k]); -1 / 0
        where a row has fewer than k other columns.  Moves n.k.P values instead of n^2."""
        n = self.n[j]
        k = int(min(k, max(1, n - (1 if exclude_diag else 0))))
        per_rank = {}
        for r in self.world.local_ranks:
            lo, hi = partition(n, self.world.size, r)
            if hi > lo:
                ids = None
                if self.order[j] is not None:      # report (and break ties by) the caller's ids
                    ids = self._index_vector(r, ("ids", j), self.order[j][lo:hi])
                src, tmp, wide = self.cur[j][r], None, None
                if src.dtype == np.float16:
                    src = wide = self.ops[r].widen(src)
                if getattr(src, "blocked", False) and not (k <= 32 and self.ops[r].name == "hip"):
                    # a row of a panel-blocked matrix is 128-byte pieces 4 MiB apart: the k selection rounds of
                    # the many-pass kernel (k > 32) re-read it, so they run on a row-major copy; up to k = 32 the
                    # one-pass kernel reads the panel-blocked matrix itself, eight rows per wave
                    tmp = self.ops[r].matrix(src.rows, src.cols)
                    self.ops[r].permute(src, tmp)
                    src = tmp
                per_rank[r] = self.ops[r].topk_rows(src, min(k, hi - lo), col0=lo,
                                                    exclude_diag=exclude_diag, col_ids=ids)
                for m in (tmp, wide):
                    if m is not None:
                        m.free()
            else:
                per_rank[r] = (np.full((n, 1), -1, np.int32), np.zeros((n, 1), np.float32))
        parts = self.world.gather_list(per_rank)
        if len(parts) == 1 and parts[0][0].shape[1] == k:
            idx, val = parts[0][0], parts[0][1].astype(np.float64)        # one rank: the kernel's order is the answer
        else:
            idx = np.concatenate([p[0] for p in parts], axis=1)
            val = np.concatenate([p[1] for p in parts], axis=1).astype(np.float64)
            key = np.where(idx >= 0, val, -np.inf)
            order = np.lexsort((idx, -key), axis=1)[:, :k]
            rows = np.arange(n)[:, None]
            idx, val = idx[rows, order], np.where(idx[rows, order] >= 0, val[rows, order], 0.0)
        if self.inv[j] is not None:                # rows back into the caller's order
            idx, val = idx[self.inv[j]], val[self.inv[j]]
        return idx, val

    def release(self):
        """Free the work buffers; the evidence counts stay (the ``Evidence`` attributes of
        the estimators read them lazily)."""
        for group in (self.cur, self.nxt):
            for per_rank in group:
                for m in per_rank.values():
                    m.free()
        for sides in self.sides:
            for s in sides.values():
                for name in ("wd", "t", "ap", "sh_send", "sh_recv"):
                    m = getattr(s, name, None)
                    if m is not None:
                        m.free()
                s.sh_send_t = s.sh_recv_t = None
                for x in (s.x1, s.x2):
                    if x is not None:
                        x.send.free()
                        x.recv.free()
                        x.send_t = x.recv_t = None
                        for st in x.stages or []:
                            s.ops.event_destroy(st["event"])
                        x.stages = None

    def leg_times(self):
        """Mean milliseconds per tag from the recorded events."""
        acc = {}
        for tag, r, a, b in self.events or []:
            acc.setdefault(tag, []).append(self.ops[r].elapsed_ms(a, b))
            if hasattr(self.ops[r], "event_destroy"):
                self.ops[r].event_destroy(a)
                self.ops[r].event_destroy(b)
        if self.events:
            self.events = []
        return {t: (float(np.mean(v)), len(v)) for t, v in acc.items()}


# --------------------------------------------------------------------------------------
# the plug into fit()
# --------------------------------------------------------------------------------------
def make_solver(ops_factory, device, world, specs, mode):
    """What ``estimators._make_solver`` calls when a fit needs the Python choreography."""
    from simrank_amd import estimators
    if type(world) is _product.LocalWorld:       # (the product's plain world object: give it the exchanges)
        world = LocalWorld(world.size, world.symmetric_shards, world.leg2_stages, world.exchange_precision, loop="python")
    return Solver(ops_factory or estimators._default_ops_factory(device), world, specs, mode)


def _install():
    from simrank_amd import estimators
    estimators.PYTHON_SOLVER = make_solver


_install()
