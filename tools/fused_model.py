#!/usr/bin/env python3
"""CPU model of the one-launch leg 1 (fused.hip) on ONE XCD: unit costs, lines per panel, and an LRU
model of the XCD's L2 with the units of consecutive panels resident the way the dispatcher keeps them
(`slots` workgroups, each replaced by the next unit in launch order when it finishes).

    python tools/fused_model.py [--workload pl32768d32] [--block 128] [--thr 3] [--panels 4]
                                [--slots 128] [--cap 32768] [--streams 1] [--stores 1] [--one-panel 0]

Lines are keyed (panel, operand row); `--streams` adds the id / pattern / row-record bytes a unit reads
(streamed once, 128-byte lines) and `--stores` the 4 KiB tiles it writes, both through the same LRU.
`--one-panel 1`: a unit of panel p + 1 starts only when every unit of panel p has finished (what a
panel-at-a-time schedule would do to the hit rate; the idle slots are reported as `fill`).
"""
import argparse
import heapq
import os
import sys
from collections import OrderedDict

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simrank_amd import ingest, synth                       # noqa: E402
from tests.pydriver import SideSpec, reorder_specs      # noqa: E402


def plan(rowptr, col, n, block, thr, min_steps=8):
    """Per block: (set columns, remainder ids in stream order)."""
    out = []
    nblk = (n + block - 1) // block
    for b in range(nblk):
        lo, hi = b * block, min(n, (b + 1) * block)
        c = col[rowptr[lo]:rowptr[hi]]
        u, cnt = np.unique(c, return_counts=True)
        dense = u[cnt >= thr]
        if (len(dense) + 15) // 16 < min_steps:
            dense = dense[:0]
        rem = c[~np.isin(c, dense)]
        out.append((dense, rem))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="pl32768d32")
    ap.add_argument("--block", type=int, default=128)
    ap.add_argument("--thr", type=int, default=3)
    ap.add_argument("--panels", type=int, default=4)
    ap.add_argument("--slots", type=int, default=128)
    ap.add_argument("--cap", type=int, default=32768)
    ap.add_argument("--streams", type=int, default=1)
    ap.add_argument("--stores", type=int, default=1)
    ap.add_argument("--one-panel", type=int, default=0)
    ap.add_argument("--group", type=int, default=3)
    ap.add_argument("--sim", type=int, default=1)
    ap.add_argument("--split", type=float, default=0, help="units costlier than this are cut into equal pieces")
    ap.add_argument("--mfma-cost", type=float, default=24.0, help="cost units per 16-column step and 128 rows")
    ap.add_argument("--line-cost", type=float, default=1.0, help="cost units per gathered line")
    args = ap.parse_args()

    df = synth.WORKLOADS[args.workload][0]()
    _, csr = ingest.directed(df, False, "from", "to", "weight")
    specs, _ = reorder_specs([SideSpec(csr, csr.rowscale, 0.8)])
    c = specs[0].csr
    n = c.n_rows
    rowptr, col = c.rowptr.astype(np.int64), c.col.astype(np.int64)
    blocks = plan(rowptr, col, n, args.block, args.thr)
    pairs = sum(len(d) for d, _ in blocks)
    rem = sum(len(r) for _, r in blocks)
    steps = sum((len(d) + 15) // 16 for d, _ in blocks)
    print(f"# {args.workload}: N={n} nnz={len(col)} block={args.block} thr={args.thr}: pairs={pairs} steps={steps} "
          f"remainder={rem} lines/panel={pairs + rem} covered={1 - rem / len(col):.3f} "
          f"mfma tiles/panel={steps * args.block // 32}")
    # units: a block with a set alone; blocks without one grouped
    units = []
    b = 0
    while b < len(blocks):
        d, r = blocks[b]
        if len(d):
            units.append([b])
            b += 1
        else:
            e, ent = b + 1, len(r)
            while e < len(blocks) and e - b < args.group and not len(blocks[e][0]) and ent + len(blocks[e][1]) <= 6144:
                ent += len(blocks[e][1])
                e += 1
            units.append(list(range(b, e)))
            b = e
    ucost, ulines = [], []
    for u in units:
        st = sum((len(blocks[b][0]) + 15) // 16 for b in u)
        ln = sum(len(blocks[b][1]) for b in u)
        ucost.append(st * args.mfma_cost * (args.block / 128) + ln * 2.5 * args.line_cost + 100)
        lines = np.concatenate([np.concatenate([blocks[b][0], blocks[b][1]]) for b in u])
        ulines.append(lines)
    if args.split > 0:
        nc, nl = [], []
        for cst, ln in zip(ucost, ulines):
            k = max(1, int(np.ceil(cst / args.split)))
            for piece in np.array_split(ln, k):
                nc.append(cst / k)
                nl.append(piece)
        ucost, ulines = nc, nl
        units = [[0]] * len(ucost)
    ucost = np.array(ucost)
    order = np.argsort(-ucost, kind="stable")
    print(f"# units/panel={len(units)} cost: max={ucost.max():.0f} mean={ucost.mean():.0f} "
          f"p50={np.median(ucost):.0f} p90={np.percentile(ucost, 90):.0f}; sum/slots={ucost.sum() / args.slots:.0f} "
          f"(max/(sum/slots) = {ucost.max() / (ucost.sum() / args.slots):.2f})")
    top = order[:8] if args.split <= 0 else []
    print("# heaviest units: " + ", ".join(f"{ucost[i]:.0f}(set {sum(len(blocks[b][0]) for b in units[i])}, "
                                             f"rem {sum(len(blocks[b][1]) for b in units[i])})" for i in top))
    if not args.sim:
        return
    # event simulation: each resident unit issues its lines at a rate of 1 line per (cost / lines) time units
    rng = np.random.default_rng(0)
    lru = OrderedDict()
    hits = np.zeros(args.panels, dtype=np.int64)
    total = np.zeros(args.panels, dtype=np.int64)
    extra_miss = np.zeros(args.panels, dtype=np.int64)

    def touch(key, p, count=True):
        if key in lru:
            lru.move_to_end(key)
            if count:
                hits[p] += 1
        else:
            lru[key] = None
            if len(lru) > args.cap:
                lru.popitem(last=False)
            if not count:
                extra_miss[p] += 1
        if count:
            total[p] += 1

    dispatch = [(p, int(u)) for p in range(args.panels) for u in order]
    # heap of (next event time, seq, state)
    heap = []
    seq = 0
    nxt = 0
    now = 0.0
    busy = 0.0
    live_panel = {}     # panel -> resident units

    def start(t):
        nonlocal nxt, seq
        p, u = dispatch[nxt]
        nxt += 1
        lines = ulines[u]
        k = len(lines)
        nstream = 0
        if args.streams:
            # ids 2 B per gathered line, pattern bits 1 KiB per quad, row records
            nb = 0 if args.split > 0 else sum(len(blocks[b][0]) for b in units[u])
            nstream = int((2 * k + nb // 64 * 1024 + 2 * nb + 1024 * len(units[u])) // 128) + 1
        nstore = 32 * len(units[u]) * args.block // 128 if args.stores else 0
        dt = ucost[u] / max(1, k)
        live_panel[p] = live_panel.get(p, 0) + 1
        heapq.heappush(heap, (t + dt, seq, [p, u, 0, dt, nstream, nstore, 0, 0]))
        seq += 1

    t = 0.0
    slots_free = args.slots
    panel_done = {p: 0 for p in range(args.panels)}
    cur_panel = 0
    idle = 0.0
    last_t = 0.0
    while True:
        while slots_free > 0 and nxt < len(dispatch) and (not args.one_panel or dispatch[nxt][0] <= cur_panel):
            start(t)
            slots_free -= 1
        if not heap:
            break
        t, _, st = heapq.heappop(heap)
        idle += slots_free * (t - last_t)
        last_t = t
        p, u, pos, dt, nstream, nstore, sdone, wdone = st
        lines = ulines[u]
        k = len(lines)
        touch((p << 20) | int(lines[pos]), p)
        pos += 1
        # streams and stores interleaved proportionally
        while sdone < nstream * pos // max(1, k):
            touch((1 << 40) | (u << 20) | sdone, p, count=False)
            sdone += 1
        while wdone < nstore * pos // max(1, k):
            touch((2 << 40) | (p << 28) | (u << 12) | wdone, p, count=False)
            wdone += 1
        if pos < k:
            st[2], st[6], st[7] = pos, sdone, wdone
            heapq.heappush(heap, (t + dt, seq, st))
            seq += 1
        else:
            slots_free += 1
            panel_done[p] += 1
            if panel_done[p] == len(units):
                cur_panel = max(cur_panel, p + 1)
    mid = slice(1, args.panels - 1) if args.panels > 2 else slice(0, args.panels)
    h, tt = hits[mid].sum(), total[mid].sum()
    np_ = hits[mid].size
    print(f"block={args.block} thr={args.thr} slots={args.slots} one_panel={args.one_panel} streams={args.streams} "
          f"stores={args.stores}: accesses/panel={total[0]} hit rate {h / tt:.3f} operand misses/panel {(tt - h) / np_:.0f} "
          f"(compulsory {n}) stream+store fills/panel {extra_miss[mid].sum() / np_:.0f}; makespan {t:.0f}, "
          f"ideal {ucost.sum() * args.panels / args.slots:.0f}, slot idle {idle / (t * args.slots):.3f}")


if __name__ == "__main__":
    main()
