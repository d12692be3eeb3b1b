"""CPU oracle for the SimRank / SimRank++ power-iteration hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``simrank_amd/`` imports this module; it is
used by ``tests/``, by ``__graft_entry__.smoke()`` (as the checker) and by the
``cpu_baseline`` leg of ``bench.py`` (as the timed CPU port).  The product path is the HIP
library and fails loudly without it.

What this is: a dense float64 NumPy restatement of what ysong1231/SimRank computes
(reference file ``SimRank/SimRank.py``; every function cites the lines it follows).  It
is written from the reference's *behaviour* — function-style, vectorised ingest, one
shared iteration core — not from its text.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the untouched reference
in the build container (through a 12-line pandas-compat shim), runs
every class on seeded edge lists and stores inputs + outputs under ``tests/golden/``;
``tests/test_oracle_golden.py`` checks this module against every one of those vectors
(label order, S to 1e-12, convergence iteration, stdout text, Evidence/Weight), including
the two cases anchored to the reference notebook's printed numbers (K(10,10):
0.285637/0.285658 at iteration 13, ``examples/basic_examples.ipynb:1325,1520,1728``).
"""
from __future__ import annotations

import io
import time

import numpy as np
import pandas as pd

BAR_LENGTH = 30  # Helper.py:3


# --------------------------------------------------------------------------------------
# stdout text (Helper.py:5-19 and SimRank.py:128,132-134)
# --------------------------------------------------------------------------------------
def progress_text(progress: float) -> str:
    """Text one ``update_progress`` call writes (Helper.py:5-19)."""
    status = ""
    if isinstance(progress, int):
        progress = float(progress)
    if not isinstance(progress, float):
        raise ValueError("Progress must be float")
    if progress < 0:
        raise ValueError("Progress below 0")
    if progress >= 1:
        progress, status = 1, "Done...\r\n"
    filled = int(round(BAR_LENGTH * progress))
    bar = "#" * filled + "-" * (BAR_LENGTH - filled)
    return f"\rPercent: [{bar}] {round(progress * 100, 1)}% {status}"


def converged_text(k: int) -> str:
    """Text written when the test at loop index k passes (SimRank.py:132)."""
    return f'\rPercent: [{"#" * BAR_LENGTH}] 100% Complete! \n\rConverged at iteration {k}'


# --------------------------------------------------------------------------------------
# ingest
# --------------------------------------------------------------------------------------
def _inv_or_zero(x: np.ndarray) -> np.ndarray:
    """1/x with +-inf -> 0 (SimRank.py:49, :197-198)."""
    with np.errstate(divide="ignore"):
        r = 1.0 / np.asarray(x, dtype=np.float64)
    r[~np.isfinite(r)] = 0.0
    return r


def _check_unique_pairs(rows: np.ndarray, cols: np.ndarray, ncols: int) -> None:
    """pivot() refuses duplicate (index, column) pairs (SimRank.py:50, :199-200)."""
    key = rows.astype(np.int64) * int(ncols) + cols.astype(np.int64)
    if np.unique(key).size != key.size:
        raise ValueError("Index contains duplicate entries, cannot reshape")


def directed_graph(data: pd.DataFrame, weighted=False, from_node_column="from",
                   to_node_column="to", weight_column="weight"):
    """Edge list -> (node order, dense G).  SimRank.py:42-52.

    Node order is the iteration order of the Python set built at SimRank.py:42 (quirk
    Q0).  G[v, u] = 1/indeg(v) for an edge u->v, or 1/sum of incoming weights when
    ``weighted`` (quirk Q3: not w/sum w).
    """
    src = data[from_node_column]
    dst = data[to_node_column]
    nodes = list(set(src.unique()) | set(dst.unique()))          # :42
    pos = pd.Index(nodes)
    r = pos.get_indexer(dst)
    c = pos.get_indexer(src)
    n = len(nodes)
    if weighted:
        per_target = data.groupby(to_node_column)[weight_column].sum()       # :45
    else:
        per_target = data.groupby(to_node_column)[from_node_column].count()  # :47
    scale = _inv_or_zero(per_target.reindex(dst).to_numpy())                  # :48-49
    _check_unique_pairs(r, c, n)                                              # :50
    G = np.zeros((n, n))
    G[r, c] = scale                                                           # :51-52
    return nodes, G


def bipartite_graph(data: pd.DataFrame, weighted=False, node_group1_column="user",
                    node_group2_column="item", weight_column="weight"):
    """Edge list -> set-order labels, sorted labels and the two rectangular graphs.

    SimRank.py:186-200 (identical copy at :377-391).  The matrices come out of
    ``pivot`` and are therefore in *sorted* label order, while the sets keep Python set
    order (quirk Q1).  G12[a, i] = 1/deg(a), G21[i, a] = 1/deg(i) (quirk Q9).
    """
    c1 = data[node_group1_column]
    c2 = data[node_group2_column]
    set1 = list(set(c1.unique()))                                             # :186
    set2 = list(set(c2.unique()))                                             # :187
    lab1 = pd.Index(np.sort(c1.unique()))
    lab2 = pd.Index(np.sort(c2.unique()))
    i1 = lab1.get_indexer(c1)
    i2 = lab2.get_indexer(c2)
    if weighted:
        d1 = data.groupby(node_group1_column)[weight_column].sum()            # :191
        d2 = data.groupby(node_group2_column)[weight_column].sum()            # :192
    else:
        d1 = data.groupby(node_group1_column)[node_group2_column].count()     # :194
        d2 = data.groupby(node_group2_column)[node_group1_column].count()     # :195
    s1 = _inv_or_zero(d1.reindex(c1).to_numpy())                              # :197
    s2 = _inv_or_zero(d2.reindex(c2).to_numpy())                              # :198
    _check_unique_pairs(i1, i2, len(lab2))                                    # :199
    G12 = np.zeros((len(lab1), len(lab2)))
    G21 = np.zeros((len(lab2), len(lab1)))
    G12[i1, i2] = s1                                                          # :199
    G21[i2, i1] = s2                                                          # :200
    return set1, set2, list(lab1), list(lab2), G12, G21


# --------------------------------------------------------------------------------------
# SimRank++ precompute
# --------------------------------------------------------------------------------------
def evidence(G: np.ndarray) -> np.ndarray:
    """E = 1 - 0.5**(number of common in-neighbours).  SimRank.py:315-316."""
    pat = (G > 0).astype(np.int64)
    common = pat @ pat.T
    return 1 - 0.5 ** common


def weight(G: np.ndarray) -> np.ndarray:
    """W = diag(exp(-var_ddof1(non-zero entries of the row))) . G.  SimRank.py:326-333.

    A row with fewer than two non-zeros has NaN variance -> 0 -> spread 1 (quirk Q4).
    """
    nz = G != 0
    cnt = nz.sum(axis=1)
    with np.errstate(invalid="ignore", divide="ignore"):
        mean = np.where(cnt > 0, G.sum(axis=1) / np.maximum(cnt, 1), 0.0)
        dev = np.where(nz, G - mean[:, None], 0.0)
        var = (dev * dev).sum(axis=1) / (cnt - 1)
    var = np.where(cnt > 1, var, 0.0)                                         # NaN -> 0
    spread = np.exp(-var)
    return spread[:, None] * G                                                # :333


# --------------------------------------------------------------------------------------
# iteration core
# --------------------------------------------------------------------------------------
def converged(s1: np.ndarray, s2: np.ndarray, eps: float) -> bool:
    """All-elements strict test, diagonal included.  SimRank.py:74 (:222)."""
    return int((np.abs(s1 - s2) > eps).sum()) == 0


def update(W: np.ndarray, S: np.ndarray, coef: float, E=None, apriori=None, lbd=None):
    """One similarity update: the bodies of SimRank.py:139-140, :298-299, :361-362,
    :420-421, :453-454, :488-489.  Evaluation order follows the reference expression."""
    prod = W.dot(S).dot(W.T)
    if apriori is not None:
        new = (1 - lbd) * E * coef * prod + lbd * apriori                     # :453
    elif E is not None:
        new = E * coef * prod                                                 # :361
    else:
        new = coef * prod                                                     # :139
    np.fill_diagonal(new, 1)
    return new


def update_rows(W: np.ndarray, S: np.ndarray, coef: float, rows: slice):
    """Rows ``rows`` of ``update(W, S, coef)`` — the same expression restricted to a slab of
    output rows (used by bench.py to time a bounded sample of one CPU iteration)."""
    new = coef * W[rows].dot(S).dot(W.T)
    idx = np.arange(rows.start, rows.stop)
    new[idx - rows.start, idx] = 1
    return new


def iterate_directed(W, C=0.8, iterations=100, eps=1e-4, E=None, apriori=None, lbd=None,
                     out: io.StringIO | None = None):
    """Loop of SimRank.py:124-140 (PP :346-362, Apriori :438-454).

    Returns (S, k) with k = loop index at which the convergence test passed, or None
    when ``iterations`` updates were applied without a passing test.
    """
    n = W.shape[0]
    old = np.zeros((n, n))
    new = np.zeros((n, n))
    np.fill_diagonal(new, 1)
    if out is not None:
        out.write("Start iterating...\n")
    for k in range(iterations):
        if converged(old, new, eps):
            if out is not None:
                out.write(converged_text(k))
            return new, k
        if out is not None:
            out.write(progress_text(k / iterations))
        old = new
        new = update(W, old, C, E, apriori, lbd)
    return new, None


def iterate_bipartite(W12, W21, C1=0.8, C2=0.8, iterations=100, eps=1e-4, E1=None, E2=None,
                      apriori1=None, apriori2=None, lbd1=None, lbd2=None,
                      out: io.StringIO | None = None):
    """Loop of SimRank.py:280-302 (PP :402-424, Apriori :470-492): Gauss-Seidel, the
    group-2 update consumes the group-1 matrix of the same iteration (:301)."""
    n1, n2 = W12.shape
    old1, new1 = np.zeros((n1, n1)), np.eye(n1)
    old2, new2 = np.zeros((n2, n2)), np.eye(n2)
    if out is not None:
        out.write("Start iterating...\n")
    for k in range(iterations):
        if converged(old1, new1, eps) and converged(old2, new2, eps):          # :289
            if out is not None:
                out.write(converged_text(k))
            return new1, new2, k
        if out is not None:
            out.write(progress_text(k / iterations))
        old1 = new1
        new1 = update(W12, new2, C1, E1, apriori1, lbd1)                       # :298
        old2 = new2
        new2 = update(W21, new1, C2, E2, apriori2, lbd2)                       # :301
    return new1, new2, None


# --------------------------------------------------------------------------------------
# fit-level restatements: (labels, S, k, stdout text, extras)
# --------------------------------------------------------------------------------------
def _pp_log(out, what, fn, G):
    if out is not None:
        out.write(f"Initializing {what} matrix...\n")
    t0 = time.time()
    r = fn(G)
    if out is not None:
        out.write(f"Finished in {time.time() - t0}s!\n")
    return r


def fit_simrank(data, C=0.8, weighted=False, from_node_column="from", to_node_column="to",
                weight_column="weight", iterations=100, eps=1e-4, verbose=True):
    """SimRank.fit, SimRank.py:79-141."""
    out = io.StringIO() if verbose else None
    nodes, G = directed_graph(data, weighted, from_node_column, to_node_column, weight_column)
    S, k = iterate_directed(G, C, iterations, eps, out=out)
    return dict(labels=nodes, S=S, k=k, stdout=out.getvalue() if out else "", G=G)


def fit_simrank_pp(data, C=0.8, weighted=False, from_node_column="from", to_node_column="to",
                   weight_column="weight", iterations=100, eps=1e-4, verbose=True,
                   apriori=None, lbd=0.5):
    """SimRankPP.fit (SimRank.py:339-363); with ``apriori`` AprioriSimRank.fit (:431-455)."""
    out = io.StringIO() if verbose else None
    nodes, G = directed_graph(data, weighted, from_node_column, to_node_column, weight_column)
    W = _pp_log(out, "Weight", weight, G)                                      # :342
    E = _pp_log(out, "Evidence", evidence, G)                                  # :344
    S, k = iterate_directed(W, C, iterations, eps, E=E, apriori=apriori,
                            lbd=lbd if apriori is not None else None, out=out)
    return dict(labels=nodes, S=S, k=k, stdout=out.getvalue() if out else "", G=G, W=W, E=E)


def fit_bipartite(data, C1=0.8, C2=0.8, weighted=False, node_group1_column="user",
                  node_group2_column="item", weight_column="weight", iterations=100,
                  eps=1e-4, verbose=True):
    """BipartiteSimRank.fit, SimRank.py:227-303.  ``labels*`` are the set-order labels
    the reference attaches to the sorted-order arrays (quirk Q1); ``sorted*`` is the
    order the arrays are really in."""
    out = io.StringIO() if verbose else None
    set1, set2, lab1, lab2, G12, G21 = bipartite_graph(
        data, weighted, node_group1_column, node_group2_column, weight_column)
    S1, S2, k = iterate_bipartite(G12, G21, C1, C2, iterations, eps, out=out)
    return dict(labels1=set1, labels2=set2, sorted1=lab1, sorted2=lab2, S1=S1, S2=S2, k=k,
                stdout=out.getvalue() if out else "", G12=G12, G21=G21)


def fit_bipartite_pp(data, C1=0.8, C2=0.8, weighted=False, node_group1_column="user",
                     node_group2_column="item", weight_column="weight", iterations=100,
                     eps=1e-4, verbose=True, strict_reference=True,
                     apriori1=None, apriori2=None, lbd1=0.5, lbd2=0.5):
    """BipartiteSimRankPP.fit (SimRank.py:393-425); with priors
    BipartitleAprioriSimRank.fit (:461-493).

    ``strict_reference`` keeps quirk Q2: the group-2 update is multiplied by
    Evidence_N1 (:423, :491) — a broadcasting ValueError when n1 != n2.  With
    ``strict_reference=False`` Evidence_N2 is used (the evident intent).
    """
    out = io.StringIO() if verbose else None
    set1, set2, lab1, lab2, G12, G21 = bipartite_graph(
        data, weighted, node_group1_column, node_group2_column, weight_column)
    W1 = _pp_log(out, "Weight", weight, G12)                                   # :396
    W2 = _pp_log(out, "Weight", weight, G21)                                   # :397
    E1 = _pp_log(out, "Evidence", evidence, G12)                               # :399
    E2 = _pp_log(out, "Evidence", evidence, G21)                               # :400
    E_for_2 = E1 if strict_reference else E2                                   # :423
    use_prior = apriori1 is not None
    S1, S2, k = iterate_bipartite(
        W1, W2, C1, C2, iterations, eps, E1=E1, E2=E_for_2,
        apriori1=apriori1, apriori2=apriori2,
        lbd1=lbd1 if use_prior else None, lbd2=lbd2 if use_prior else None, out=out)
    return dict(labels1=set1, labels2=set2, sorted1=lab1, sorted2=lab2, S1=S1, S2=S2, k=k,
                stdout=out.getvalue() if out else "", G12=G12, G21=G21, W1=W1, W2=W2,
                E1=E1, E2=E2)
