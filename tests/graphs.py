"""Small seeded graphs used by the golden generator and the parity tests."""
import numpy as np
import pandas as pd

from simrank_amd.synth import er_directed, powerlaw_directed  # noqa: F401  (re-exported)

# labels printed by the reference notebook for its K(10,10) example
# (examples/basic_examples.ipynb:1286-1728)
K10_USERS = [131904, 125794, 34576, 82418, 83090, 59477, 8405, 118205, 74142, 121535]
K10_MOVIES = [480, 260, 356, 296, 2571, 589, 110, 527, 593, 318]


def toy5():
    """5-node directed toy of SURVEY.md Appendix B (node 5 has no in-edges)."""
    e = [(1, 2), (2, 3), (3, 1), (4, 1), (1, 3), (2, 4), (5, 1)]
    return pd.DataFrame({"from": [a for a, _ in e], "to": [b for _, b in e],
                         "weight": [2, 1, 3, 1, 5, 2, 4]})


def quirky_directed():
    """Self-loops, zero in-degree nodes, single-in-neighbour rows, a sink, one heavy row."""
    e = [(0, 0), (0, 1), (1, 2), (2, 2), (3, 2), (4, 5), (5, 4), (6, 7), (8, 7), (9, 7),
         (10, 7), (11, 7), (0, 7), (1, 7), (2, 9), (3, 9), (7, 10), (6, 6), (11, 3), (4, 3)]
    rng = np.random.default_rng(12)
    return pd.DataFrame({"from": [a for a, _ in e], "to": [b for _, b in e],
                         "weight": rng.integers(1, 6, size=len(e))})


def complete_bipartite(users, items):
    u = np.repeat(np.asarray(users), len(items))
    i = np.tile(np.asarray(items), len(users))
    rng = np.random.default_rng(10)
    return pd.DataFrame({"user": u, "item": i, "weight": rng.integers(1, 6, size=u.size)})


def bipartite_random(n1, n2, p, seed):
    rng = np.random.default_rng(seed)
    m = rng.random((n1, n2)) < p
    m[np.arange(n1), rng.integers(0, n2, size=n1)] = True      # every user rates something
    m[rng.integers(0, n1, size=n2), np.arange(n2)] = True      # every item is rated
    u, i = np.nonzero(m)
    order = rng.permutation(u.size)
    return pd.DataFrame({"user": u[order] + 1000, "item": i[order] + 1,
                         "weight": rng.integers(1, 6, size=u.size)})


def relabel_big_ints(df, cols, seed, shared=None):
    """Replace labels by large sparse ints so Python-set order != sorted order (Q0/Q1).
    Directed graphs (from/to) use one label space for both endpoints."""
    rng = np.random.default_rng(seed)
    out = df.copy()
    if shared is None:
        shared = set(cols) == {"from", "to"}
    groups = [list(cols)] if shared else [[c] for c in cols]
    for g in groups:
        old = np.unique(np.concatenate([df[c].to_numpy() for c in g]))
        new = rng.choice(10**6, size=old.size, replace=False) + 17
        m = dict(zip(old, new))
        for c in g:
            out[c] = df[c].map(m).to_numpy()
    return out
