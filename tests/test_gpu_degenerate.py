"""Degenerate inputs through the class surface (the C loops behind it) against the oracle: one node with a self-loop, two
nodes, one edge with string labels, stars, a chain, a clique; iterations = 0 / 1, eps = 0 / >= 1, a small C; symmetric and
asymmetric priors on those graphs; one-edge, one-user, one-item, square and rectangular bipartite graphs — every class."""
import numpy as np
import pandas as pd
import pytest

import simrank_amd.SimRank as SRA
from oracle import simrank_oracle as O

pytestmark = pytest.mark.gpu


def test_degenerate_graphs_and_loop_bounds_against_the_oracle():
    ok = 0
    graphs = {
        "self_loop": pd.DataFrame({"from": [7], "to": [7]}),
        "two_nodes": pd.DataFrame({"from": [1, 2], "to": [2, 1]}),
        "one_edge": pd.DataFrame({"from": ["a"], "to": ["b"]}),
        "star": pd.DataFrame({"from": [0] * 9, "to": list(range(1, 10))}),
        "in_star": pd.DataFrame({"from": list(range(1, 10)), "to": [0] * 9}),
        "chain": pd.DataFrame({"from": list(range(40)), "to": list(range(1, 41))}),
        "clique5": pd.DataFrame([(i, j) for i in range(5) for j in range(5) if i != j], columns=["from", "to"]),
    }
    for name, df in graphs.items():
        for kw in (dict(), dict(iterations=0), dict(iterations=1), dict(eps=1.0), dict(eps=0.0, iterations=7), dict(C=0.3)):
            for cls, ofit in ((SRA.SimRank, O.fit_simrank), (SRA.SimRankPP, O.fit_simrank_pp)):
                est = cls()
                got = est.fit(df, verbose=False, **kw)
                want = ofit(df, verbose=False, **kw)
                assert list(got.index) == want["labels"], (name, kw)
                np.testing.assert_allclose(got.values, want["S"], rtol=1e-5, atol=1e-30, err_msg=f"{name} {kw} {cls.__name__}")
                assert est.converged_at == want["k"], (name, kw, est.converged_at, want["k"])
                ok += 1
            n = len(set(df["from"]) | set(df["to"]))
            rng = np.random.default_rng(len(name))
            for sym in (True, False):
                A = rng.random((n, n))
                if sym:
                    A = (A + A.T) / 2
                got = SRA.AprioriSimRank().fit(df, A, lbd=0.4, verbose=False, **kw)
                want = O.fit_simrank_pp(df, apriori=A, lbd=0.4, verbose=False, **kw)
                np.testing.assert_allclose(got.values, want["S"], rtol=1e-5, atol=1e-30, err_msg=f"{name} {kw} prior sym={sym}")
                ok += 1
    bip = {
        "one_edge": pd.DataFrame({"user": ["u"], "item": ["i"]}),
        "one_user": pd.DataFrame({"user": [1] * 6, "item": list(range(6))}),
        "one_item": pd.DataFrame({"user": list(range(6)), "item": [3] * 6}),
        "square": pd.DataFrame([(u, i) for u in range(6) for i in range(6) if (u + i) % 3], columns=["user", "item"]),
        "rect": pd.DataFrame([(u, i) for u in range(9) for i in range(4) if (u * i + u) % 2 == 0], columns=["user", "item"]),
    }
    for name, df in bip.items():
        for kw in (dict(), dict(iterations=0), dict(iterations=2), dict(eps=1.0)):
            got = SRA.BipartiteSimRank().fit(df, verbose=False, **kw)
            want = O.fit_bipartite(df, verbose=False, **kw)
            np.testing.assert_allclose(got[0].values, want["S1"], rtol=1e-5, atol=1e-30, err_msg=f"{name} {kw}")
            np.testing.assert_allclose(got[1].values, want["S2"], rtol=1e-5, atol=1e-30, err_msg=f"{name} {kw}")
            got = SRA.BipartiteSimRankPP().fit(df, verbose=False, strict_reference=False, **kw)
            want = O.fit_bipartite_pp(df, verbose=False, strict_reference=False, **kw)
            np.testing.assert_allclose(got[0].values, want["S1"], rtol=1e-5, atol=1e-30, err_msg=f"{name} {kw} pp")
            np.testing.assert_allclose(got[1].values, want["S2"], rtol=1e-5, atol=1e-30, err_msg=f"{name} {kw} pp")
            ok += 2
    assert ok == 208
