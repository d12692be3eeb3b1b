"""Host logic above the C ABI, on CPU: ingest -> CSR, the iteration driver, console text,
quirk handling, sharded exchange layout.  The kernels are replaced by the NumPy test double
of tests/cpu_ops.py (the real kernels are checked by the `-m gpu` tests)."""
import numpy as np
import pandas as pd
import pytest

from oracle import simrank_oracle as O
from simrank_amd import ingest
from tests.pydriver import LocalWorld
from tests.conftest import Golden, golden_names
from tests.cpu_ops import NumpyOps
from tests.helpers import check_against_golden, run_estimator


def _factory():
    ops = NumpyOps()
    return lambda rank: ops


@pytest.mark.parametrize("name", golden_names())
def test_estimators_match_reference_on_cpu_double(name):
    g = Golden(name)
    if g.raises:
        with pytest.raises(ValueError):
            run_estimator(g, _factory())
        return
    est, res, text = run_estimator(g, _factory())
    check_against_golden(g, est, res, text)


@pytest.mark.parametrize("mode", ["sparse", "dense", "hybrid"])
@pytest.mark.parametrize("name", ["SimRank_er64", "SimRankPP_er64_weighted",
                                  "AprioriSimRank_er64", "AprioriSimRank_er64_asym", "BipartiteSimRank_b5030",
                                  "BipartiteSimRankPP_b40", "BipartitleAprioriSimRank_b40", "BipartitleAprioriSimRank_b40_asym"])
def test_every_mode_gives_the_same_answer(name, mode):
    g = Golden(name)
    est, res, text = run_estimator(g, _factory(), mode=mode)
    assert est.engine_mode == ("sparse" if name.endswith("_asym") else mode)
    check_against_golden(g, est, res, text, check_attrs=False)


@pytest.mark.parametrize("world", [2, 3, 5])
@pytest.mark.parametrize("name", ["SimRank_er64", "SimRankPP_quirky", "AprioriSimRank_er64", "AprioriSimRank_er64_asym",
                                  "BipartiteSimRank_b5030", "BipartiteSimRankPP_b40",
                                  "BipartitleAprioriSimRank_b40", "SimRank_toy5"])
def test_virtual_ranks_loopback(name, world):
    """P column shards with the all-to-all done by copies must reproduce P = 1."""
    g = Golden(name)
    est, res, text = run_estimator(g, _factory(), world=LocalWorld(world), mode="sparse")
    check_against_golden(g, est, res, text)


def test_ingest_matches_oracle_dense_graphs():
    for name in golden_names("SimRank"):
        g = Golden(name)
        if g.raises:
            continue
        kw = {k: v for k, v in g.kwargs.items() if k.endswith("_column") or k == "weighted"}
        nodes, csr = ingest.directed(g.frame, kw.get("weighted", False),
                                     kw.get("from_node_column", "from"),
                                     kw.get("to_node_column", "to"),
                                     kw.get("weight_column", "weight"))
        onodes, G = O.directed_graph(g.frame, **kw)
        assert nodes == onodes
        np.testing.assert_array_equal(csr.dense(), G)
        assert (np.diff(csr.col.astype(np.int64))[np.diff(
            np.repeat(np.arange(csr.n_rows), np.diff(csr.rowptr))) == 0] > 0).all()


def test_duplicate_edges_raise_like_pivot():
    d = pd.DataFrame({"from": [1, 1, 2], "to": [2, 2, 1]})
    with pytest.raises(ValueError, match="duplicate entries"):
        ingest.directed(d, False, "from", "to", "weight")
    b = pd.DataFrame({"user": [1, 1], "item": [5, 5]})
    with pytest.raises(ValueError, match="duplicate entries"):
        ingest.bipartite(b, False, "user", "item", "weight")


def test_missing_column_is_a_keyerror():
    import simrank_amd.SimRank as SRA
    with pytest.raises(KeyError):
        SRA.SimRank().fit(pd.DataFrame({"a": [1], "b": [2]}), _ops_factory=_factory())


def test_input_frame_is_not_mutated():
    g = Golden("SimRankPP_er64_weighted")
    before = g.frame.copy(deep=True)
    run_estimator(g, _factory())
    pd.testing.assert_frame_equal(g.frame, before)


def test_strict_reference_off_corrects_labels_and_evidence():
    """Q1: correct labels; Q2: Evidence_N2 on the group-2 update (runs when n1 != n2)."""
    g = Golden("BipartiteSimRankPP_b5030")
    with pytest.raises(ValueError, match="broadcast"):
        run_estimator(g, _factory())
    est, (s1, s2), _ = run_estimator(g, _factory(), strict_reference=False)
    want = O.fit_bipartite_pp(g.frame, strict_reference=False, **g.kwargs)
    assert list(s1.index) == want["sorted1"] and list(s2.index) == want["sorted2"]
    np.testing.assert_allclose(s1.values, want["S1"], rtol=1e-5, atol=1e-30)
    np.testing.assert_allclose(s2.values, want["S2"], rtol=1e-5, atol=1e-30)
    np.testing.assert_array_equal(est.Evidence_N2, want["E2"])
    # with the evidence switched off by an all-ones prior trick: PP with E == 1 is plain
    gb = Golden("BipartiteSimRank_bigints")
    _, (p1, _), _ = run_estimator(gb, _factory(), strict_reference=False)
    assert list(p1.index) == sorted(p1.index)


def test_apriori_must_be_ndarray():
    import simrank_amd.SimRank as SRA
    g = Golden("AprioriSimRank_toy5")
    with pytest.raises(AttributeError, match="flat"):
        SRA.AprioriSimRank().fit(g.frame, pd.DataFrame(g.args[0]), _ops_factory=_factory())


def test_partition_covers_everything():
    for n in (1, 5, 64, 1000):
        for w in (1, 2, 3, 8):
            blocks = [ingest.partition(n, w, r) for r in range(w)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
            b = -(-n // w)
            assert all(hi - lo == b for lo, hi in blocks if hi < n)


def test_reference_class_names_and_signatures():
    import inspect
    import simrank_amd.SimRank as SRA
    for name in ("SimRank", "SimRankPP", "BipartiteSimRank", "BipartiteSimRankPP",
                 "AprioriSimRank", "BipartitleAprioriSimRank", "BipartitleSimRank",
                 "BipartitleSimRankPP"):
        assert inspect.isclass(getattr(SRA, name))
    sig = inspect.signature(SRA.SimRank.fit)
    assert list(sig.parameters)[:10] == ["self", "data", "C", "weighted", "from_node_column",
                                         "to_node_column", "weight_column", "iterations", "eps",
                                         "verbose"]
    d = {k: v.default for k, v in sig.parameters.items()}
    assert (d["C"], d["weighted"], d["iterations"], d["eps"], d["verbose"]) == (0.8, False, 100, 1e-4, True)
    sig = inspect.signature(SRA.BipartiteSimRankPP.fit)
    assert list(sig.parameters)[:11] == ["self", "data", "C1", "C2", "weighted",
                                         "node_group1_column", "node_group2_column",
                                         "weight_column", "iterations", "eps", "verbose"]
    sig = inspect.signature(SRA.BipartitleAprioriSimRank.fit)
    assert list(sig.parameters)[:8] == ["self", "data", "AprioriSim1", "AprioriSim2", "C1", "C2",
                                        "lbd1", "lbd2"]
    assert SRA.BAR_LENGTH == 30 and callable(SRA.update_progress)


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("name", ["SimRank_er64", "AprioriSimRank_er64_asym", "BipartiteSimRank_b5030"])
def test_virtual_ranks_with_padded_chunks(name, world, monkeypatch):
    """Exchanged chunk rows padded by 32 floats (what large power-of-two blocks get)."""
    import tests.pydriver as drv
    monkeypatch.setattr(drv, "PAD_MIN_ROWS", 1)
    monkeypatch.setattr(drv, "PAD_MULTIPLE", 1)
    g = Golden(name)
    est, res, text = run_estimator(g, _factory(), world=LocalWorld(world), mode="sparse")
    check_against_golden(g, est, res, text)


@pytest.mark.parametrize("world", [1, 3])
def test_top_k_hand_back(world):
    """top_k=... returns the k most similar other nodes per node instead of the dense matrix;
    it must agree with sorting the dense result (ties: lower position first)."""
    g = Golden("SimRank_er64")
    import simrank_amd.SimRank as SRA
    dense = SRA.SimRank().fit(g.frame, verbose=False, _ops_factory=_factory())
    top = SRA.SimRank().fit(g.frame, verbose=False, top_k=5, world=LocalWorld(world), mode="sparse",
                            _ops_factory=_factory())
    assert list(top.columns) == ["node", "rank", "neighbor", "similarity"]
    assert len(top) == 5 * len(dense)
    labels = list(dense.index)
    for node in labels[:10]:
        row = dense.loc[node].drop(node)
        order = sorted(range(len(row)), key=lambda i: (-np.float32(row.iloc[i]), labels.index(row.index[i])))
        want = [row.index[i] for i in order[:5]]
        got = top[top.node == node].sort_values("rank")
        assert list(got.neighbor) == want
        np.testing.assert_allclose(got.similarity, [row[w] for w in want], rtol=1e-6)
    b = Golden("BipartiteSimRank_b5030")
    t1, t2 = SRA.BipartiteSimRank().fit(b.frame, verbose=False, top_k=3, _ops_factory=_factory())
    assert len(t1) == 3 * 50 and len(t2) == 3 * 30


def test_stage_plan_is_a_permutation():
    from tests.pydriver import stage_widths, staged_row_order, permute_columns
    for k, world, stages in [(64, 2, 3), (1000, 3, 4), (32768, 8, 4), (5, 2, 4)]:
        perm = staged_row_order(k, world, stages)
        assert sorted(perm.tolist()) == list(range(k))
        for h in range(world):
            lo, hi = ingest.partition(k, world, h)
            assert sum(stage_widths(hi - lo, stages)) == hi - lo
    g = Golden("SimRank_er64")
    _, csr = ingest.directed(g.frame, False, "from", "to", "weight")
    perm = staged_row_order(csr.n_cols, 2, 3)
    p = permute_columns(csr, perm)
    d = csr.dense()
    np.testing.assert_array_equal(p.dense()[:, perm], d)


def test_relabel_is_the_permuted_matrix():
    rng = np.random.default_rng(5)
    dense = (rng.random((37, 23)) < 0.2)
    dense[4] = False                                   # an empty row
    rows, cols = np.nonzero(dense)
    rowptr = np.concatenate([[0], np.cumsum(dense.sum(axis=1))]).astype(np.int32)
    csr = ingest.CSR(37, 23, rowptr, cols.astype(np.int32), rng.random(37) + 0.5)
    ro, co = rng.permutation(37), rng.permutation(23)
    for r, c in ((ro, co), (ro, None), (None, co), (None, None)):
        out = ingest.relabel(csr, r, c)
        want = csr.dense()
        want = want if r is None else want[r]
        want = want if c is None else want[:, c]
        np.testing.assert_array_equal(out.dense(), want)
        for a in range(37):                            # ids ascending inside every row
            seg = out.col[out.rowptr[a]:out.rowptr[a + 1]]
            assert np.all(np.diff(seg) > 0)


@pytest.mark.parametrize("world", [1, 2, 3])
@pytest.mark.parametrize("name", ["SimRank_er128", "SimRankPP_quirky", "AprioriSimRank_er64_asym",
                                  "BipartiteSimRankPP_b40", "BipartiteSimRank_b5030"])
def test_solver_node_order_is_invisible(name, world):
    """The solver iterates with every node set sorted by row length (driver.reorder_specs);
    results, evidence and top-k come back in the caller's order and match a solver that
    keeps the caller's order throughout."""
    from tests import pydriver as driver
    g = Golden(name)
    made = []
    orig = driver.Solver.__init__
    outs = {}
    for reorder in (True, False):
        def init(self, make_ops, w, specs, mode="auto", reorder=reorder):
            orig(self, make_ops, w, specs, mode, reorder=reorder)
            made.append(self)
        driver.Solver.__init__ = init
        try:
            est, res, _ = run_estimator(g, _factory(), world=LocalWorld(world), mode="sparse")
            ev = [getattr(est, a) for a in ("Evidence", "Evidence_N1", "Evidence_N2") if hasattr(est, a)]
            outs[reorder] = (res, ev)
        finally:
            driver.Solver.__init__ = orig
    assert any(s.order[0] is not None and not np.array_equal(s.order[0], np.arange(s.n[0])) for s in made)
    a, b = outs[True], outs[False]
    for x, y in zip(a[0] if isinstance(a[0], tuple) else (a[0],), b[0] if isinstance(b[0], tuple) else (b[0],)):
        assert list(x.index) == list(y.index)
        np.testing.assert_allclose(x.values, y.values, rtol=1e-6, atol=1e-30)
    for x, y in zip(a[1], b[1]):
        np.testing.assert_array_equal(x, y)
    check_against_golden(g, *run_estimator(g, _factory(), world=LocalWorld(world), mode="sparse"))


def test_top_k_ties_follow_the_callers_order():
    """Equal similarities are ranked by position in the caller's node order even though the
    device works in its own order (simrank_topk_rows_ids)."""
    import simrank_amd.SimRank as SRA
    # a star: every leaf is equally similar to every other leaf
    df = pd.DataFrame({"from": [0] * 12, "to": list(range(1, 13))})
    df = pd.concat([df, pd.DataFrame({"from": [3, 3, 5], "to": [0, 7, 7]})], ignore_index=True)
    dense = SRA.SimRank().fit(df, verbose=False, _ops_factory=_factory())
    top = SRA.SimRank().fit(df, verbose=False, top_k=4, _ops_factory=_factory(), mode="sparse")
    labels = list(dense.index)
    for node in labels:
        row = dense.loc[node].drop(node)
        order = sorted(range(len(row)), key=lambda i: (-np.float32(row.iloc[i]), labels.index(row.index[i])))
        got = top[top.node == node].sort_values("rank")
        assert list(got.neighbor) == [row.index[i] for i in order[:4]]


def test_bf16x3_truncation_split_is_exact_in_numpy():
    """The dense-tile kernel (csrc/blockdense.hip) feeds the matrix cores three bf16 terms per f32
    operand: hi = top 16 bits of x, mid = top 16 bits of x - hi, lo = x - hi - mid.  The same
    arithmetic in NumPy: every term is a bf16 number (low 16 bits zero), the three add up to x
    bit for bit, for both signs and every exponent whose last term is still a normal number."""
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(200_000) * np.exp(rng.uniform(-70, 80, 200_000))).astype(np.float32)
    x = x[(np.abs(x) >= 1e-33) & np.isfinite(x)]  # below ~7.7e-34 the last term would be subnormal
    x = np.concatenate([x, np.float32([0.0, 1.0, -1.0, 1 + 2.0 ** -23, 16777215.0, -3.0000002, 1e-30])])

    def trunc(v):
        return (v.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)

    hi = trunc(x)
    r = x - hi                                   # exact: both share the exponent of x
    mid = trunc(r)
    lo = r - mid                                 # at most 8 significant bits are left
    for t in (hi, mid, lo):
        assert not np.any(t.view(np.uint32) & np.uint32(0xFFFF))          # representable in bf16
    assert np.array_equal((hi + mid) + lo, x)
    assert np.array_equal(hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64),
                          x.astype(np.float64))


def _fit_all(cls_name, frame, world, **kw):
    import simrank_amd.SimRank as SRA
    ops = NumpyOps()                 # the virtual ranks of a LocalWorld share one "device"
    est = getattr(SRA, cls_name)()
    res = est.fit(frame, verbose=False, world=world, mode="sparse", _ops_factory=lambda r: ops, **kw)
    return est, (res if isinstance(res, tuple) else (res,)), [ops]


@pytest.mark.parametrize("cls_name,world", [("SimRank", 2), ("SimRank", 4), ("SimRankPP", 2),
                                            ("AprioriSimRank", 4), ("BipartiteSimRank", 2),
                                            ("BipartiteSimRankPP", 2)])
def test_half_form_sharded_leg2(cls_name, world):
    """Node counts divisible by 32 x P: leg 2 of every rank computes only the tiles i <= j and
    the mirrored ones travel in a second all-to-all (driver.Side.shard_sym).  The result is the
    one-rank result up to float32 summation order; with ``symmetric_shards=False`` the full
    form runs instead."""
    from simrank_amd import synth
    if cls_name.startswith("Bipart"):
        # (the ++ class only runs with equally large groups: quirk Q2)
        frame = synth.bipartite_zipf(128, 128 if cls_name.endswith("PP") else 192, 1500, 7)
        kw = dict(node_group1_column="user", node_group2_column="item")
    else:
        frame = synth.powerlaw_directed(256, 6, 11)
        kw = dict(from_node_column="from", to_node_column="to")
    if cls_name == "AprioriSimRank":
        rng = np.random.default_rng(5)
        a = rng.random((256, 256))
        kw["AprioriSim"] = (a + a.T) / 2                   # symmetric prior: fused path
    if cls_name.endswith("PP"):
        kw["weighted"] = True
    _, one, _ = _fit_all(cls_name, frame, LocalWorld(1), **kw)
    _, half, ops = _fit_all(cls_name, frame, LocalWorld(world), **kw)
    assert any(c[0] == "spmm_shard" for o in ops for c in o.calls)
    assert any(c[0] == "shard_unpack" for o in ops for c in o.calls)
    for x, y in zip(one, half):
        assert list(x.index) == list(y.index)
        np.testing.assert_allclose(y.values, x.values, rtol=2e-5, atol=1e-30)
    _, full, ops = _fit_all(cls_name, frame, LocalWorld(world, symmetric_shards=False), **kw)
    assert not any(c[0] == "spmm_shard" for o in ops for c in o.calls)
    for x, y in zip(one, full):
        np.testing.assert_allclose(y.values, x.values, rtol=2e-5, atol=1e-30)
    # the half form cut into stages of column tiles (exchange 2 overlapped with leg 2 on a real node): every
    # tile is computed by exactly one stage from the same operands, so the result is the unstaged one, bit for bit
    if world == 2 and not cls_name.startswith("Bipart"):
        _, staged, ops = _fit_all(cls_name, frame, LocalWorld(world, leg2_stages=3), **kw)
        n_launch = sum(c[0] == "spmm_shard" for o in ops for c in o.calls)
        n_plain = sum(c[0] == "spmm_shard" for o in _fit_all(cls_name, frame, LocalWorld(world), **kw)[2] for c in o.calls)
        assert n_launch > n_plain                       # really several launches per update
        for x, y in zip(half, staged):
            assert np.array_equal(x.values, y.values)


def test_dealt_order():
    from tests import pydriver as driver
    from tests.pydriver import dealt_order
    o = np.arange(1024)
    d = dealt_order(o, 4)                 # runs of 128: 0..127 -> shard 0, 128..255 -> shard 1, ...
    assert sorted(d) == list(o) and list(d[:128]) == list(range(128))
    assert list(d[128:256]) == list(range(512, 640)) and list(d[256:384]) == list(range(128, 256))
    o = np.arange(256)
    d = dealt_order(o, 4)                 # too few nodes for runs of 128: tiles of 32
    assert sorted(d) == list(o)
    assert list(d[:32]) == list(range(32)) and list(d[32:64]) == list(range(128, 160))
    assert list(d[64:96]) == list(range(32, 64))          # shard 1 starts with tile 1
    assert dealt_order(np.arange(100), 4) is not None and list(dealt_order(np.arange(100), 4)) == list(range(100))


def test_storage_precision_is_validated_on_the_host():
    """`fit(storage_precision=...)` (config 5's reduced-precision mode, csrc/half.hip): unknown values are
    refused before anything touches a device, and an engine without fp16-held matrices (this NumPy stand-in;
    sharded worlds; the dense legs) refuses the mode instead of silently running f32."""
    import simrank_amd.SimRank as SRA
    df = pd.DataFrame({"from": [0, 1, 2, 3], "to": [1, 2, 3, 0], "weight": [1.0, 1.0, 1.0, 1.0]})
    with pytest.raises(ValueError, match="storage_precision must be"):
        SRA.SimRank().fit(df, verbose=False, storage_precision="bf16", _ops_factory=lambda r: NumpyOps())
    with pytest.raises(ValueError, match="storage_precision='fp16' needs"):
        SRA.SimRank().fit(df, verbose=False, storage_precision="fp16", _ops_factory=lambda r: NumpyOps())
    # several ranks: the sharded loop behind the C ABI takes the mode (simrank_amd/cshard.py) — where its conditions
    # hold (whole 64-column panels per rank, one symmetric side, no prior) and the engine is the HIP one
    with pytest.raises(ValueError, match="on several ranks: .* multiple of 128"):
        SRA.SimRank().fit(df, verbose=False, storage_precision="fp16", world=LocalWorld(2),
                          _ops_factory=lambda r: NumpyOps())
    ring = pd.DataFrame({"from": np.arange(128), "to": (np.arange(128) + 1) % 128, "weight": np.ones(128)})
    with pytest.raises(ValueError, match="storage_precision='fp16' needs the HIP engine"):
        SRA.SimRank().fit(ring, verbose=False, storage_precision="fp16", world=LocalWorld(2),
                          _ops_factory=lambda r: NumpyOps())
    with pytest.raises(ValueError, match="on several ranks: a prior"):
        SRA.AprioriSimRank().fit(ring, np.eye(128), verbose=False, storage_precision="fp16", world=LocalWorld(2),
                                 _ops_factory=lambda r: NumpyOps())
    with pytest.raises(ValueError, match="on several ranks: the bipartite classes"):
        SRA.BipartiteSimRank().fit(pd.DataFrame({"user": [0, 1], "item": [0, 1], "weight": [1.0, 1.0]}), verbose=False,
                                   storage_precision="fp16", world=LocalWorld(2), _ops_factory=lambda r: NumpyOps())
    ok = SRA.SimRank().fit(df, verbose=False, storage_precision="f32", _ops_factory=lambda r: NumpyOps())
    assert ok.shape == (4, 4)


def test_host_frames_come_back_when_the_last_view_is_gone(monkeypatch):
    """hostpool: the float64 frame of a dense hand-back is a mapping of the library's own that returns to a pool when
    the caller's last reference to the array AND to every view of it (a DataFrame holds one) is gone, and is handed out
    again for the next result of that size; SIMRANK_HOST_POOL_GIB bounds what rests, 0 disables."""
    import gc
    from simrank_amd import hostpool as hp
    monkeypatch.setattr(hp, "MIN_BYTES", 1 << 10)
    hp.trim()
    a = hp.empty_f64(300, 300)
    assert a.shape == (300, 300) and a.dtype == np.float64 and a.flags.c_contiguous and a.flags.writeable
    a[:] = 7.0
    frame = pd.DataFrame(a, index=range(300), columns=range(300))
    del a
    gc.collect()
    assert hp.stats() == (0, 0)                       # the frame still holds it
    view = frame.values[5:9]
    del frame
    gc.collect()
    assert hp.stats() == (0, 0) and view[0, 0] == 7.0   # ... and so does a slice the caller kept
    del view
    gc.collect()
    assert hp.stats() == (300 * 300 * 8, 1)
    b = hp.empty_f64(300, 300)
    assert hp.stats() == (0, 0) and b[0, 0] == 7.0    # the same pages (np.empty semantics: contents are arbitrary)
    c = hp.empty_f64(300, 300)                        # a second one while the first is out: a new mapping
    c[:] = 1.0
    assert b[0, 0] == 7.0
    del b, c
    gc.collect()
    assert hp.stats()[1] == 2
    hp.trim()
    assert hp.stats() == (0, 0)
    monkeypatch.setenv("SIMRANK_HOST_POOL_GIB", "0")
    d = hp.empty_f64(300, 300)
    assert d.flags.owndata                            # plain np.empty
    monkeypatch.setenv("SIMRANK_HOST_POOL_GIB", "0.0001")     # ~107 kB: a 720 kB frame is not kept
    e = hp.empty_f64(300, 300)
    assert e.flags.owndata
