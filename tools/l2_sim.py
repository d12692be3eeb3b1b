#!/usr/bin/env python3
"""CPU model of ONE XCD's L2 during the gather leg: which order of the work keeps a panel's
operand lines resident?  (round 2, VERDICT item 2: 17.9 GB of fills for a 4.3 GB operand.)

Model: an LRU set of `cap` 128-byte lines; `slots` workgroups resident (4 waves each, one 32-row
tile per wave, tiles from the balanced tiling); every resident wave issues one batch of up to 8
lines per turn (round-robin); a workgroup that finishes is replaced by the next one in dispatch
order (panels x, x+8, ... of this XCD; inside a panel heavy workgroups first).  Lines are keyed
(panel, source row).  Variants: `slabs` = the K range is walked in that many slabs inside every
wave tile (accumulators stay in registers), 1 = today's kernel.

    python tools/l2_sim.py [--workload pl32768] [--panels 4] [--slabs 1,2,4,8] [--slots 224]
"""
import argparse
import os
import sys
from collections import OrderedDict

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simrank_amd import ingest, synth                       # noqa: E402
from tests.pydriver import SideSpec, reorder_specs      # noqa: E402


def dense_remainder(rowptr, col, n, dmin=4, dcols=128):
    """Entries left to the gather kernel after the block-dense selection (blockdense.hip)."""
    rows = np.repeat(np.arange(n), np.diff(rowptr))
    blk = rows // 128
    key = blk.astype(np.int64) * n + col
    u, inv, cnt = np.unique(key, return_inverse=True, return_counts=True)
    per_blk = np.bincount((u // n)[cnt >= dmin], minlength=(n + 127) // 128)
    dense = (cnt[inv] >= dmin) & (per_blk[blk] >= dcols)
    keep = ~dense
    rp = np.concatenate([[0], np.cumsum(np.bincount(rows[keep], minlength=n))])
    return rp.astype(np.int64), col[keep]


def build_tiles(rowptr, n, balance=4):
    nnz = int(rowptr[-1])
    nblk = (n + 31) // 32
    limit = max(balance * ((nnz + nblk - 1) // nblk), 256)
    tiles = []
    for b in range(nblk):
        stack = [(b * 32, min(n, b * 32 + 32))]
        while stack:
            lo, hi = stack.pop()
            if rowptr[hi] - rowptr[lo] <= limit or hi - lo <= 1:
                tiles.append(lo)
            else:
                mid = lo + (hi - lo + 1) // 2
                stack.append((mid, hi))
                stack.append((lo, mid))
    tiles.append(n)
    return np.array(tiles)


def build_tiles_equal(rowptr, n, target, max_rows=64):
    """Tiles of about `target` entries: consecutive rows, at most max_rows, a row heavier than the
    target alone (such rows are split over the waves of a workgroup in the kernel)."""
    tiles, lo = [], 0
    while lo < n:
        hi = lo + 1
        while hi < n and hi - lo < max_rows and rowptr[hi + 1] - rowptr[lo] <= target:
            hi += 1
        tiles.append(lo)
        lo = hi
    tiles.append(n)
    return np.array(tiles)


def wave_batches(rowptr, col, lo, hi, slabs, K, part=0, parts=1):
    """Line batches (arrays of source rows) one wave issues for rows [lo, hi), in issue order.
    parts > 1: the wave takes the part-th share of every row's entries (cooperative tile)."""
    out = []
    edges = [K * s // slabs for s in range(slabs + 1)]
    lens = rowptr[lo + 1:hi + 1] - rowptr[lo:hi]
    order = np.argsort(-lens, kind="stable")
    for s in range(slabs):
        segs = []
        for r in order:
            c = col[rowptr[lo + r]:rowptr[lo + r + 1]]
            if parts > 1:
                q = -(-len(c) // parts)
                c = c[part * q:(part + 1) * q]
            if slabs > 1:
                c = c[(c >= edges[s]) & (c < edges[s + 1])]
            segs.append(c)
        segs.sort(key=lambda c: -len(c))
        heavy = [c for c in segs if len(c) >= 64]
        rest = [c for c in segs if 0 < len(c) < 64]
        for c in heavy:
            for k in range(0, len(c), 8):
                out.append(c[k:k + 8])
        for p in range(0, len(rest), 8):
            grp = rest[p:p + 8]
            for j in range(len(grp[0])):
                out.append(np.array([c[j] for c in grp if j < len(c)]))
    return out


ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="pl32768")
ap.add_argument("--panels", type=int, default=4)
ap.add_argument("--slabs", default="1,2,4,8")
ap.add_argument("--slots", type=int, default=224)
ap.add_argument("--cap", type=int, default=30000)
ap.add_argument("--dense-min", type=int, default=4)
ap.add_argument("--equal", type=int, default=0, help="equal-weight tiles of about this many entries (0 = balanced 32-row tiles)")
ap.add_argument("--max-rows", type=int, default=64)
ap.add_argument("--balance", type=int, default=4)
ap.add_argument("--coop", type=int, default=0, help="tiles heavier than this are shared by the 4 waves of a workgroup")
ap.add_argument("--merge", type=int, default=0, help="a wave takes consecutive tiles until it holds about this many entries")
args = ap.parse_args()

df = synth.WORKLOADS[args.workload][0]()
_, csr = ingest.directed(df, False, "from", "to", "weight")
specs, _ = reorder_specs([SideSpec(csr, csr.rowscale, 0.8)])
c = specs[0].csr
n = c.n_rows
rp, col = (dense_remainder(c.rowptr.astype(np.int64), c.col.astype(np.int64), n, args.dense_min)
           if args.dense_min else (c.rowptr.astype(np.int64), c.col.astype(np.int64)))
tiles = build_tiles_equal(rp, n, args.equal, args.max_rows) if args.equal else build_tiles(rp, n, args.balance)
n_tiles = len(tiles) - 1
# wave work lists: wave j processes tiles [wt[j], wt[j+1])
wt = [0]
while wt[-1] < n_tiles:
    t = wt[-1] + 1
    while args.merge and t < n_tiles and rp[tiles[t + 1]] - rp[tiles[wt[-1]]] <= args.merge:
        t += 1
    wt.append(t)
n_waves = len(wt) - 1
# workgroups: lists of (wave list index, part, parts)
wgs, cur = [], []
for j in range(n_waves):
    w = rp[tiles[wt[j + 1]]] - rp[tiles[wt[j]]]
    if args.coop and w > args.coop:
        if cur:
            wgs.append(cur)
            cur = []
        wgs.append([(j, k, 4) for k in range(4)])
    else:
        cur.append((j, 0, 1))
        if len(cur) == 4:
            wgs.append(cur)
            cur = []
if cur:
    wgs.append(cur)
n_wg = len(wgs)
print(f"# {args.workload}: N={n} remainder nnz={len(col)} tiles={n_tiles} waves={n_waves} workgroups/panel={n_wg}")

for slabs in [int(v) for v in args.slabs.split(",")]:
    # per workgroup: its 4 waves' batch lists (the same for every panel)
    wg_waves = []
    for g in range(n_wg):
        waves = []
        for (j, part, parts) in wgs[g]:
            bl = []
            for t in range(wt[j], wt[j + 1]):
                bl += wave_batches(rp, col, tiles[t], tiles[t + 1], slabs, n, part, parts)
            waves.append(bl)
        wg_waves.append(waves)
    dispatch = [(p, g) for p in range(args.panels) for g in range(n_wg - 1, -1, -1)]
    lru = OrderedDict()
    hits = np.zeros(args.panels, dtype=np.int64)
    total = np.zeros(args.panels, dtype=np.int64)
    active = []            # [panel, waves(list of batch lists), positions]
    nxt = 0
    turns = 0
    while True:
        while len(active) < args.slots and nxt < len(dispatch):
            p, g = dispatch[nxt]
            nxt += 1
            active.append([p, wg_waves[g], [0] * len(wg_waves[g])])
        if not active:
            break
        still = []
        for wg in active:
            p, waves, pos = wg
            alive = False
            for w, bl in enumerate(waves):
                if pos[w] < len(bl):
                    for line in bl[pos[w]]:
                        key = (p << 20) | int(line)
                        total[p] += 1
                        if key in lru:
                            hits[p] += 1
                            lru.move_to_end(key)
                        else:
                            lru[key] = None
                            if len(lru) > args.cap:
                                lru.popitem(last=False)
                    pos[w] += 1
                    alive = alive or pos[w] < len(bl)
            if alive:
                still.append(wg)
        active = still
        turns += 1
    compulsory = n  # first touch of every line of a panel at most
    mid = slice(1, args.panels - 1) if args.panels > 2 else slice(0, args.panels)
    h, t = hits[mid].sum(), total[mid].sum()
    print(f"slabs={slabs}: turns={turns} accesses/panel={total[0]} hit rate (inner panels) {h / t:.3f} "
          f"misses/panel {(t - h) / max(1, hits[mid].size):.0f} (compulsory <= {compulsory})", flush=True)
