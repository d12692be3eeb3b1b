"""Seeded synthetic edge lists for the measurement harness.

The reference ships no data (its notebook's csv files are git-ignored), so every
workload BASELINE.json names is generated here: int64 labels 0..N-1, duplicate edges
dropped, every id forced to appear so N is exact, an integer ``weight`` column in 1..5.
"""
from __future__ import annotations

import numpy as np
import pandas as pd


def _finish_directed(src, dst, n, rng, columns=("from", "to")):
    key = np.unique(src.astype(np.int64) * n + dst.astype(np.int64))
    src, dst = key // n, key % n
    seen = np.zeros(n, dtype=bool)
    seen[src] = True
    seen[dst] = True
    missing = np.flatnonzero(~seen)
    if missing.size:                       # ring edge v -> v+1 for ids that never appeared
        src = np.concatenate([src, missing])
        dst = np.concatenate([dst, (missing + 1) % n])
        key = np.unique(src * n + dst)
        src, dst = key // n, key % n
    order = rng.permutation(src.size)      # edge lists arrive unordered in practice
    return pd.DataFrame({columns[0]: src[order], columns[1]: dst[order],
                         "weight": rng.integers(1, 6, size=src.size)})


def er_directed(n: int, p: float, seed: int) -> pd.DataFrame:
    """Erdos-Renyi directed graph (config 2: n=8192, p=0.001, seed=8192)."""
    rng = np.random.default_rng(seed)
    m = rng.binomial(n * n, p)
    src = rng.integers(0, n, size=m)
    dst = rng.integers(0, n, size=m)
    return _finish_directed(src, dst, n, rng)


def powerlaw_directed(n: int, avg_deg: float, seed: int, exponent: float = 2.1,
                      exact: bool = False) -> pd.DataFrame:
    """Chung-Lu directed graph with power-law in- and out-degrees (config 4:
    n=32768, avg_deg=32, seed=32768).  Expected degree of rank-i node ~ (i+1)^(-1/(g-1)).
    ``exact``: n.avg_deg distinct edges (default: that many draws, duplicates dropped)."""
    rng = np.random.default_rng(seed)
    w = (np.arange(n) + 1.0) ** (-1.0 / (exponent - 1.0))
    w /= w.sum()
    m = int(n * avg_deg)
    ps = rng.permutation(n)                       # (draw order fixed: the round-1 graphs stay the same)
    src = ps[rng.choice(n, size=m, p=w)]
    pd_ = rng.permutation(n)
    dst = pd_[rng.choice(n, size=m, p=w)]
    if exact:
        # Chung-Lu sampling with replacement loses the duplicates (a quarter of the draws at
        # n = 32768, avg_deg = 32): keep drawing until m DISTINCT edges exist, so the mean degree
        # after de-duplication is the one asked for
        key = np.unique(src.astype(np.int64) * n + dst)
        while key.size < m:
            extra = int((m - key.size) * 1.5) + 1024
            s2 = ps[rng.choice(n, size=extra, p=w)]
            d2 = pd_[rng.choice(n, size=extra, p=w)]
            new = np.setdiff1d(np.unique(s2.astype(np.int64) * n + d2), key)
            key = np.concatenate([key, rng.permutation(new)[:m - key.size]])
        src, dst = key // n, key % n
    return _finish_directed(src, dst, n, rng)


def hub_and_spoke(n: int = 300, seed: int = 300, hubs: int = 12) -> pd.DataFrame:
    """BTS-flights-like directed graph (config 1): a few hubs connected to everybody in
    both directions, spokes with a handful of routes."""
    rng = np.random.default_rng(seed)
    hub = rng.choice(n, size=hubs, replace=False)
    a = np.repeat(hub, n)
    b = np.tile(np.arange(n), hubs)
    keep = (a != b) & (rng.random(a.size) < 0.6)
    extra_s = rng.integers(0, n, size=4 * n)
    extra_d = rng.integers(0, n, size=4 * n)
    src = np.concatenate([a[keep], b[keep], extra_s])
    dst = np.concatenate([b[keep], a[keep], extra_d])
    ok = src != dst
    return _finish_directed(src[ok], dst[ok], n, rng)


def bipartite_zipf(n1: int, n2: int, nnz: int, seed: int,
                   columns=("user", "item")) -> pd.DataFrame:
    """MovieLens-1M-shaped bipartite graph (config 3: 6040 x 3706, ~1.0 M ratings):
    Zipf-skewed item popularity, log-normal user activity, ratings 1..5."""
    rng = np.random.default_rng(seed)
    pu = rng.lognormal(0.0, 1.0, size=n1)
    pu /= pu.sum()
    pi = (np.arange(n2) + 1.0) ** -0.9
    pi /= pi.sum()
    pi = pi[rng.permutation(n2)]
    key = np.empty(0, dtype=np.int64)
    while key.size < nnz:
        need = int((nnz - key.size) * 1.3) + 1024
        u = rng.choice(n1, size=need, p=pu)
        i = rng.choice(n2, size=need, p=pi)
        key = np.unique(np.concatenate([key, u.astype(np.int64) * n2 + i]))
    key = rng.permutation(key)[:nnz]
    u, i = key // n2, key % n2
    mu = np.flatnonzero(np.bincount(u, minlength=n1) == 0)
    mi = np.flatnonzero(np.bincount(i, minlength=n2) == 0)
    u = np.concatenate([u, mu, rng.integers(0, n1, size=mi.size)])
    i = np.concatenate([i, rng.integers(0, n2, size=mu.size), mi])
    key = np.unique(u * n2 + i)
    u, i = key // n2, key % n2
    order = rng.permutation(u.size)
    return pd.DataFrame({columns[0]: u[order], columns[1]: i[order],
                         "weight": rng.integers(1, 6, size=u.size)})


WORKLOADS = {
    # name: (factory, kind) — the configurations of BASELINE.json
    "bts300": (lambda: hub_and_spoke(300, 300), "directed"),
    "er8192": (lambda: er_directed(8192, 0.001, 8192), "directed"),
    "ml1m": (lambda: bipartite_zipf(6040, 3706, 1_000_209, 1), "bipartite"),
    "pl32768": (lambda: powerlaw_directed(32768, 32, 32768), "directed"),
    "pl65536": (lambda: powerlaw_directed(65536, 32, 65536), "directed"),
    # config 4 with the stated mean degree AFTER de-duplication (1 048 576 distinct edges; the plain
    # "pl32768" recipe of SURVEY.md §8d keeps 783 100 of its 1 048 576 draws)
    "pl32768d32": (lambda: powerlaw_directed(32768, 32, 32768, exact=True), "directed"),
    # not a BASELINE.json configuration: same size and average degree as pl32768 without the skew
    "er32768": (lambda: er_directed(32768, 32 / 32768, 32768), "directed"),
    # not a BASELINE.json configuration either: config 4's recipe at half the nodes — a panel's operand slice is 2 MiB, half an
    # XCD's L2 (what leg 1 does when the slice fits with room to spare)
    "pl16384d32": (lambda: powerlaw_directed(16384, 32, 16384, exact=True), "directed"),
}
