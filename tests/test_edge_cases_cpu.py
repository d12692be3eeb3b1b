"""Edge-case graphs through the estimators with the NumPy test double, against the oracle."""
import numpy as np
import pytest

import simrank_amd.SimRank as SRA
from oracle import simrank_oracle as O
from tests.cpu_ops import NumpyOps
from tests.edge_cases import bipartite_cases, directed_cases

D, B = directed_cases(), bipartite_cases()


def _f():
    ops = NumpyOps()
    return lambda r: ops


@pytest.mark.parametrize("name", sorted(D))
@pytest.mark.parametrize("cls", ["SimRank", "SimRankPP"])
def test_directed_edge_cases(name, cls):
    df = D[name]
    est = getattr(SRA, cls)()
    got = est.fit(df, verbose=False, _ops_factory=_f())
    want = (O.fit_simrank if cls == "SimRank" else O.fit_simrank_pp)(df, verbose=False)
    assert list(got.index) == want["labels"]
    np.testing.assert_allclose(got.values, want["S"], rtol=1e-5, atol=1e-30)
    assert est.converged_at == want["k"]


@pytest.mark.parametrize("name", sorted(B))
def test_bipartite_edge_cases(name):
    df = B[name]
    est = SRA.BipartiteSimRank()
    s1, s2 = est.fit(df, verbose=False, _ops_factory=_f())
    want = O.fit_bipartite(df, verbose=False)
    assert list(s1.index) == want["labels1"] and list(s2.index) == want["labels2"]
    np.testing.assert_allclose(s1.values, want["S1"], rtol=1e-5, atol=1e-30)
    np.testing.assert_allclose(s2.values, want["S2"], rtol=1e-5, atol=1e-30)
    assert est.converged_at == want["k"]
