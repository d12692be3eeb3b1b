"""The oracle (oracle/simrank_oracle.py) against every golden vector the reference produced.

This is what pins the oracle: label order exact, S within 1e-12, convergence iteration
and stdout text exact (wall times masked), Evidence / Weight exact.
"""
import re

import numpy as np
import pytest

from oracle import simrank_oracle as O
from tests.conftest import Golden, golden_names

TIME_RE = re.compile(r"Finished in [0-9.e+-]+s!")
TOL = dict(rtol=1e-12, atol=1e-14)


def _run(g: Golden):
    kw = dict(g.kwargs)
    if g.cls == "SimRank":
        return O.fit_simrank(g.frame, **kw)
    if g.cls == "SimRankPP":
        return O.fit_simrank_pp(g.frame, **kw)
    if g.cls == "AprioriSimRank":
        return O.fit_simrank_pp(g.frame, apriori=g.args[0], **kw)
    if g.cls == "BipartiteSimRank":
        return O.fit_bipartite(g.frame, **kw)
    if g.cls == "BipartiteSimRankPP":
        return O.fit_bipartite_pp(g.frame, strict_reference=True, **kw)
    if g.cls == "BipartitleAprioriSimRank":
        return O.fit_bipartite_pp(g.frame, strict_reference=True,
                                  apriori1=g.args[0], apriori2=g.args[1], **kw)
    raise AssertionError(g.cls)


@pytest.mark.parametrize("name", golden_names())
def test_oracle_matches_reference(name):
    g = Golden(name)
    if g.raises:
        with pytest.raises(ValueError):
            _run(g)
        return
    r = _run(g)
    if g.kwargs.get("verbose", True):      # k is parsed from stdout: unknown when quiet
        assert (r["k"] if r["k"] is not None else -1) == g.k
    assert TIME_RE.sub("Finished in <t>s!", r["stdout"]) == g.stdout
    if "S" in g.out:
        assert list(r["labels"]) == list(g.out["labels"])
        np.testing.assert_allclose(r["S"], g.out["S"], **TOL)
        np.testing.assert_array_equal(r["G"], g.out["G"])
        if "E" in g.out:
            np.testing.assert_array_equal(r["E"], g.out["E"])
            np.testing.assert_allclose(r["W"], g.out["W"], rtol=1e-15, atol=0)
    else:
        assert list(r["labels1"]) == list(g.out["labels1"])
        assert list(r["labels2"]) == list(g.out["labels2"])
        assert list(r["sorted1"]) == list(g.out["sorted1"])
        assert list(r["sorted2"]) == list(g.out["sorted2"])
        np.testing.assert_allclose(r["S1"], g.out["S1"], **TOL)
        np.testing.assert_allclose(r["S2"], g.out["S2"], **TOL)
        np.testing.assert_array_equal(r["G12"], g.out["G12"])
        np.testing.assert_array_equal(r["G21"], g.out["G21"])
        if "E1" in g.out:
            np.testing.assert_array_equal(r["E1"], g.out["E1"])
            np.testing.assert_array_equal(r["E2"], g.out["E2"])
            np.testing.assert_allclose(r["W1"], g.out["W1"], rtol=1e-15, atol=0)
            np.testing.assert_allclose(r["W2"], g.out["W2"], rtol=1e-15, atol=0)


def test_notebook_known_answers():
    """KA1 / KA2 of SURVEY.md §4: the numbers printed in the reference's notebook."""
    g = Golden("BipartiteSimRank_k10")
    r = _run(g)
    assert r["k"] == 13                                     # ipynb:1325
    assert round(r["S1"][0, 1], 6) == 0.285637              # ipynb:1520
    assert round(r["S2"][0, 1], 6) == 0.285658              # ipynb:1728
    assert list(r["labels1"]) == [131904, 125794, 34576, 82418, 83090, 59477, 8405,
                                  118205, 74142, 121535]
    r = _run(Golden("BipartiteSimRankPP_k10"))
    assert r["k"] == 13                                     # ipynb:2234
    assert round(r["S1"][0, 1], 6) == 0.284645              # ipynb:2430
    assert round(r["S2"][0, 1], 6) == 0.284666              # ipynb:2638


def test_toy_known_answers():
    """SURVEY.md Appendix B toy, values verified against the reference at survey time."""
    r = _run(Golden("SimRank_toy5"))
    S = r["S"]
    assert list(r["labels"]) == [1, 2, 3, 4, 5]
    for (i, j), v in {(0, 1): 0.057712, (0, 2): 0.091398, (0, 3): 0.125084,
                      (1, 2): 0.423064, (2, 3): 0.423064, (1, 3): 0.046129}.items():
        assert round(S[i, j], 6) == v
    assert np.array_equal(S[4], np.eye(5)[4])


def test_progress_text():
    assert O.progress_text(0.0) == "\rPercent: [" + "-" * 30 + "] 0.0% "
    assert O.progress_text(1 / 3).startswith("\rPercent: [##########")
    with pytest.raises(ValueError):
        O.progress_text(-0.5)
    with pytest.raises(ValueError):
        O.progress_text("x")
