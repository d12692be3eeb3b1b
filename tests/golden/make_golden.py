#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the UNTOUCHED reference.

Runs only in the build container (needs /root/reference).  It imports the reference
module as-is and rebinds its ``pd`` global to a proxy whose ``DataFrame(...)`` turns a
``set`` passed as index/columns into ``list(set)`` — what the pandas the reference was
written for did implicitly (pandas 2.3.3 raises "index cannot be a set").
Nothing else of the reference is altered, and none of its text is stored: each fixture
holds inputs (edge list, kwargs) and outputs (label order, S, convergence iteration,
stdout, and for SimRank++ Evidence / Weight).

    python tests/golden/make_golden.py          # rewrites tests/golden/*.npz + manifest.json
"""
import contextlib
import io
import json
import os
import re
import sys
import types
import warnings

import numpy as np
import pandas as _pd

warnings.filterwarnings("ignore")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
from SimRank import SimRank as SR  # noqa: E402  (the reference, untouched)


class _PandasProxy(types.ModuleType):
    def __getattr__(self, k):
        return getattr(_pd, k)


def _frame(data=None, index=None, columns=None, **kw):
    if isinstance(index, (set, frozenset)):
        index = list(index)
    if isinstance(columns, (set, frozenset)):
        columns = list(columns)
    return _pd.DataFrame(data, index=index, columns=columns, **kw)


_proxy = _PandasProxy("pandas_proxy")
_proxy.DataFrame = _frame
SR.pd = _proxy

sys.path.insert(0, os.path.join(HERE, "..", ".."))
from tests.graphs import (  # noqa: E402
    K10_USERS, K10_MOVIES, toy5, er_directed, powerlaw_directed, quirky_directed,
    bipartite_random, complete_bipartite, relabel_big_ints)

TIME_RE = re.compile(r"Finished in [0-9.e+-]+s!")


def run(cls_name, df, args=(), **kw):
    obj = getattr(SR, cls_name)()
    buf = io.StringIO()
    try:
        with contextlib.redirect_stdout(buf):
            res = obj.fit(df, *args, **kw)
    except Exception as e:  # recorded as the expected behaviour
        return dict(raises=type(e).__name__, message=str(e))
    text = TIME_RE.sub("Finished in <t>s!", buf.getvalue())
    m = re.search(r"Converged at iteration (\d+)", text)
    out = dict(stdout=text, k=int(m.group(1)) if m else -1)
    if isinstance(res, tuple):
        out["labels1"] = np.asarray(res[0].index)
        out["labels2"] = np.asarray(res[1].index)
        out["S1"] = res[0].values
        out["S2"] = res[1].values
        out["sorted1"] = np.asarray(obj.Graph_N1_N2.index)
        out["sorted2"] = np.asarray(obj.Graph_N1_N2.columns)
        out["G12"] = obj.Graph_N1_N2.values
        out["G21"] = obj.Graph_N2_N1.values
        if hasattr(obj, "Evidence_N1") and isinstance(obj.Evidence_N1, np.ndarray):
            out["E1"], out["E2"] = obj.Evidence_N1, obj.Evidence_N2
            out["W1"], out["W2"] = obj.Weight_N1, obj.Weight_N2
    else:
        out["labels"] = np.asarray(res.index)
        out["S"] = res.values
        out["G"] = obj.Graph.values
        if hasattr(obj, "Evidence") and isinstance(obj.Evidence, np.ndarray):
            out["E"], out["W"] = obj.Evidence, obj.Weight
    return out


def cases():
    """(name, class, DataFrame, positional extras, kwargs)."""
    rng = np.random.default_rng(2024)
    out = []

    def add(name, cls, df, args=(), **kw):
        out.append((name, cls, df, args, kw))

    t5 = toy5()
    er64 = er_directed(64, 0.08, seed=64)
    er128 = er_directed(128, 0.05, seed=128)
    er256 = er_directed(256, 0.03, seed=256)
    pl256 = powerlaw_directed(256, 12, seed=7)
    qk = quirky_directed()
    erbig = relabel_big_ints(er_directed(48, 0.1, seed=5), ["from", "to"], seed=11)

    for cls in ("SimRank", "SimRankPP"):
        add(f"{cls}_toy5", cls, t5)
        add(f"{cls}_toy5_iter0", cls, t5, iterations=0)
        add(f"{cls}_toy5_eps1", cls, t5, eps=1.0)
        add(f"{cls}_toy5_iter3", cls, t5, iterations=3)
        add(f"{cls}_toy5_quiet", cls, t5, verbose=False)
        add(f"{cls}_er64", cls, er64)
        add(f"{cls}_er64_weighted", cls, er64, weighted=True)
        add(f"{cls}_er64_C06", cls, er64, C=0.6, eps=1e-6)
        add(f"{cls}_er128", cls, er128)
        add(f"{cls}_pl256", cls, pl256, iterations=6)
        add(f"{cls}_quirky", cls, qk)
        add(f"{cls}_quirky_weighted", cls, qk, weighted=True)
        add(f"{cls}_bigints", cls, erbig)
        ren = er64.rename(columns={"from": "src", "to": "dst", "weight": "w"})
        add(f"{cls}_er64_cols", cls, ren, from_node_column="src", to_node_column="dst",
            weight_column="w", weighted=True)
    add("SimRank_er256", "SimRank", er256, iterations=8)
    dup = _pd.concat([t5, t5.iloc[:1]], ignore_index=True)
    add("SimRank_dup_edges", "SimRank", dup)

    pr5 = np.full((5, 5), 0.1)
    add("AprioriSimRank_toy5", "AprioriSimRank", t5, (pr5,))
    pr64 = rng.random((len(set(er64["from"]) | set(er64["to"])),) * 2)
    pr64 = (pr64 + pr64.T) / 2
    add("AprioriSimRank_er64", "AprioriSimRank", er64, (pr64,), lbd=0.3)
    # a prior that is NOT symmetric: the iterates stop being symmetric (SimRank.py:453)
    add("AprioriSimRank_er64_asym", "AprioriSimRank", er64, (rng.random(pr64.shape),), lbd=0.4)
    add("AprioriSimRank_quirky_asym", "AprioriSimRank", qk, (rng.random((12, 12)),), lbd=0.25,
        iterations=7)

    k10 = complete_bipartite(K10_USERS, K10_MOVIES)
    b40 = bipartite_random(40, 40, 0.15, seed=40)
    b5030 = bipartite_random(50, 30, 0.15, seed=50)
    b40big = relabel_big_ints(bipartite_random(24, 24, 0.2, seed=24), ["user", "item"], seed=3)
    for cls in ("BipartiteSimRank", "BipartiteSimRankPP"):
        add(f"{cls}_k10", cls, k10)
        add(f"{cls}_b40", cls, b40)
        add(f"{cls}_b40_weighted", cls, b40, weighted=True)
        add(f"{cls}_b40_iter0", cls, b40, iterations=0)
        add(f"{cls}_b40_iter2", cls, b40, iterations=2)
        add(f"{cls}_b40_C", cls, b40, C1=0.7, C2=0.9, eps=1e-6)
        add(f"{cls}_b5030", cls, b5030)          # PP: Q2 -> ValueError in the reference
        add(f"{cls}_bigints", cls, b40big)
        ren = b40.rename(columns={"user": "u", "item": "i", "weight": "w"})
        add(f"{cls}_b40_cols", cls, ren, node_group1_column="u", node_group2_column="i",
            weight_column="w", weighted=True)
    p1 = rng.random((40, 40)); p1 = (p1 + p1.T) / 2
    p2 = rng.random((40, 40)); p2 = (p2 + p2.T) / 2
    add("BipartitleAprioriSimRank_b40", "BipartitleAprioriSimRank", b40, (p1, p2),
        lbd1=0.4, lbd2=0.2)
    add("BipartitleAprioriSimRank_b40_asym", "BipartitleAprioriSimRank", b40,
        (rng.random((40, 40)), p2), lbd1=0.3, lbd2=0.3)
    # round 2 (appended: the cases above keep their random streams)
    # quirk Q2 fires only when the group-2 update RUNS: no iteration / eps >= 1 return identities,
    # and a single group-1 node gives a 1 x 1 Evidence_N1 that NumPy broadcasts
    add("BipartiteSimRankPP_b5030_iter0", "BipartiteSimRankPP", b5030, iterations=0)
    add("BipartiteSimRankPP_b5030_eps1", "BipartiteSimRankPP", b5030, eps=1.0)
    one = _pd.DataFrame({"user": [7] * 6, "item": [3, 9, 4, 12, 5, 8], "weight": [1, 2, 3, 1, 2, 5]})
    add("BipartiteSimRankPP_one_user", "BipartiteSimRankPP", one)
    # BASELINE.json configs[0]: BTS-flights-like directed graph (~300 airports), C = 0.8, 10 iterations
    from simrank_amd.synth import hub_and_spoke
    bts = hub_and_spoke(300, 300)
    add("SimRank_bts300", "SimRank", bts, C=0.8, iterations=10)
    add("SimRankPP_bts300", "SimRankPP", bts, C=0.8, iterations=10)
    return out


def main():
    """``--only a,b``: (re)generate just those cases and merge them into the manifest."""
    only = None
    if "--only" in sys.argv:
        only = set(sys.argv[sys.argv.index("--only") + 1].split(","))
    manifest = {}
    if only is not None:
        with open(os.path.join(HERE, "manifest.json")) as f:
            manifest = json.load(f)
    else:
        for f in os.listdir(HERE):
            if f.endswith(".npz"):
                os.remove(os.path.join(HERE, f))
    for name, cls, df, args, kw in cases():
        if only is not None and name not in only:
            continue
        res = run(cls, df, args, **kw)
        arrays = {f"in_{c}": df[c].to_numpy() for c in df.columns}
        for i, a in enumerate(args):
            arrays[f"arg{i}"] = np.asarray(a)
        meta = dict(cls=cls, kwargs=kw, columns=list(df.columns), nargs=len(args))
        for k, v in res.items():
            if isinstance(v, np.ndarray):
                arrays[f"out_{k}"] = v
            else:
                meta[k] = v
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrays)
        manifest[name] = meta
        print(f"{name:40s} {meta.get('raises', 'k=%s' % meta.get('k'))}")
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
