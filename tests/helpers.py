"""Shared checks: run one of the package's estimators on a golden case and compare with what
the reference produced (labels exact, stdout exact modulo wall times, S within tolerance)."""
import contextlib
import io
import re

import numpy as np

import simrank_amd.SimRank as SRA

TIME_RE = re.compile(r"Finished in [0-9.e+-]+s!")
RTOL = 1e-5      # north_star: fp32 S entries within 1e-5 relative of the NumPy reference


def run_estimator(g, ops_factory=None, **extra):
    est = getattr(SRA, g.cls)()
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        res = est.fit(g.frame, *g.args, **g.kwargs, _ops_factory=ops_factory, **extra)
    return est, res, TIME_RE.sub("Finished in <t>s!", buf.getvalue())


def assert_close(got, want, rtol=RTOL):
    """Relative on every entry; exact zeros must stay exactly zero (SimRank++ support)."""
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, rtol=rtol, atol=1e-30)


def check_against_golden(g, est, res, text, rtol=RTOL, check_attrs=True):
    assert text == g.stdout
    if g.kwargs.get("verbose", True):
        assert (est.converged_at if est.converged_at is not None else -1) == g.k
    if "S" in g.out:
        assert list(res.index) == list(g.out["labels"])
        assert list(res.columns) == list(g.out["labels"])
        assert res.values.dtype == np.float64
        assert_close(res.values, g.out["S"], rtol)
        if check_attrs:
            assert est.Nodes == set(g.out["labels"].tolist())
            np.testing.assert_array_equal(est.Graph.values, g.out["G"])
            assert list(est.Graph.index) == list(g.out["labels"])
            if "E" in g.out:
                np.testing.assert_array_equal(est.Evidence, g.out["E"])
                np.testing.assert_allclose(est.Weight, g.out["W"], rtol=1e-15, atol=0)
                assert isinstance(est.Evidence, np.ndarray) and isinstance(est.Weight, np.ndarray)
    else:
        s1, s2 = res
        assert list(s1.index) == list(g.out["labels1"]) == list(s1.columns)
        assert list(s2.index) == list(g.out["labels2"]) == list(s2.columns)
        assert_close(s1.values, g.out["S1"], rtol)
        assert_close(s2.values, g.out["S2"], rtol)
        if check_attrs:
            assert list(est.Graph_N1_N2.index) == list(g.out["sorted1"])
            assert list(est.Graph_N1_N2.columns) == list(g.out["sorted2"])
            assert est.Graph_N1_N2.index.name == g.kwargs.get("node_group1_column", "user")
            assert est.Graph_N1_N2.columns.name == g.kwargs.get("node_group2_column", "item")
            np.testing.assert_array_equal(est.Graph_N1_N2.values, g.out["G12"])
            np.testing.assert_array_equal(est.Graph_N2_N1.values, g.out["G21"])
            if "E1" in g.out:
                np.testing.assert_array_equal(est.Evidence_N1, g.out["E1"])
                np.testing.assert_array_equal(est.Evidence_N2, g.out["E2"])
                np.testing.assert_allclose(est.Weight_N1, g.out["W1"], rtol=1e-15, atol=0)
                np.testing.assert_allclose(est.Weight_N2, g.out["W2"], rtol=1e-15, atol=0)
