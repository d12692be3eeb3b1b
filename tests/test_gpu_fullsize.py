"""BASELINE.json's configurations AT FULL SIZE through the loops that SHIP (round 6): `fit()` of the class surface, the
C-level plans it runs (`simrank_plan_*`, `simrank_biplan_*`) and the sharded C loops an 8-GPU run executes
(`simrank_shardplan_*`, `simrank_shardbiplan_*`, eight virtual ranks on the one GPU).

* config 2 (N = 8192) is small enough for the float64 oracle: the WHOLE fit — every element, the labels, the convergence
  index (SimRank.py:129-141);
* configs 4 and 5 (N = 32768 / 65536) are not: rows of one more update are recomputed on the host in float64 from rows of
  the device's own S_k (evidence factor included), top-k against a host selection on those rows, the size-independent
  properties (unit diagonal, range, symmetry, support, identity rows);
* the sharded loops: sampled columns bit-equal to the Python choreography that makes the same launches, within 1e-5 of
  the float64 recomputation.
"""
import numpy as np
import pytest
import scipy.sparse as sp

import simrank_amd.SimRank as SRA
from oracle import simrank_oracle as O
from simrank_amd import ingest, synth
from tests.helpers import RTOL, assert_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from simrank_amd.engine import HipOps
    return HipOps(0)


def _pattern(csr):
    return sp.csr_matrix((np.ones(csr.col.size), csr.col, csr.rowptr), shape=(csr.n_rows, csr.n_cols))


def _sampled(csr, extra=19):
    """24 rows: the ends, a third in, the longest row, random ones."""
    n = csr.n_rows
    return sorted({0, 1, n // 3, n - 1, int(np.argmax(np.diff(csr.rowptr)))} |
                  set(int(r) for r in np.random.default_rng(n).choice(n, size=extra, replace=False)))


def _want_rows(csr, scale, coef, rows, fetch, evidence=False):
    """Rows `rows` of the NEXT iterate in float64 from rows of the current one (`fetch(ids) -> float32 [len, n]`, caller's
    order): S'[a, :] = E[a, :] * coef * (W[a, :] . S) . W^T, diag <- 1 (SimRank.py:139-140, :361-362); -> (want, counts)."""
    n = csr.n_rows
    rs = np.asarray(scale, dtype=np.float32).astype(np.float64)
    Pat = _pattern(csr)
    W = sp.diags(rs) @ Pat
    need = sorted(set(int(i) for a in rows for i in csr.col[csr.rowptr[a]:csr.rowptr[a + 1]]))
    pos = {i: p for p, i in enumerate(need)}
    part = fetch(need).astype(np.float64) if need else np.zeros((0, n))
    live = sp.diags((rs > 0).astype(np.float64)) @ Pat if evidence else None
    want, counts = {}, {}
    for a in rows:
        t = rs[a] * part[[pos[int(i)] for i in csr.col[csr.rowptr[a]:csr.rowptr[a + 1]]]].sum(axis=0)
        w = coef * (W @ t)
        if evidence:
            cnt = np.asarray((live[a] @ live.T).todense()).ravel()
            w = w * (1.0 - 0.5 ** cnt)
            counts[a] = cnt
        w[a] = 1.0
        want[a] = w
    return want, counts


def _host_topk(row, a, k):
    cand = np.array([c for c in range(row.size) if c != a])
    order = cand[np.lexsort((cand, -row[cand]))][:k]
    return order, row[order]


# ---------------------------------------------------------------------------------------------------------------------
# config 2: the whole fit against the oracle
# ---------------------------------------------------------------------------------------------------------------------
def test_config2_er8192_whole_fit_against_the_oracle(ops):
    """BASELINE.json configs[1], N = 8192, end to end: `SimRank().fit` (ingest -> simrank_plan_run_cb -> f64 hand-back) and the
    plan driven directly, against `oracle.fit_simrank` — EVERY element within 1e-5, the labels, the convergence index."""
    from simrank_amd.engine import Plan
    df = synth.WORKLOADS["er8192"][0]()
    want = O.fit_simrank(df, verbose=False)
    est = SRA.SimRank()
    got = est.fit(df, verbose=False)
    assert list(got.index) == list(want["labels"]) == list(got.columns)
    assert est.converged_at == want["k"] and want["k"] is not None
    assert_close(got.values, want["S"])
    assert np.array_equal(np.diag(got.values), np.ones(len(got)))
    nodes, csr = ingest.directed(df, False, "from", "to", "weight")
    plan = Plan(ops, csr, coef=0.8)
    done, conv = plan.run(100, 1e-4)
    assert conv == want["k"]
    direct = plan.result()
    rows = _sampled(csr)
    assert np.array_equal(plan.rows(rows).astype(np.float64), direct[rows])
    plan.free()
    assert np.array_equal(direct, got.values)          # fit() IS this loop: the same bits
    # (SimRank++ of this size is not repeated here: the oracle's evidence is an int64 dense matmul, SimRank.py:315 — four
    # minutes at N = 8192 on the box's 16 CPUs; its whole fits are compared at N <= 4096 in test_gpu_parity.py, and config 5
    # below checks the evidence factor at N = 65536 on sampled rows)


# ---------------------------------------------------------------------------------------------------------------------
# configs 2 / 4 at full size through the C plan: sampled rows of one more update, properties
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("workload,iters", [("er8192", 8), ("pl32768", 3), ("pl32768d32", 3)])
def test_full_size_rows_through_the_c_plan(ops, workload, iters):
    """`simrank_plan_*` (what fit() runs on one GPU) at BASELINE's sizes: `iters` updates, then 24 rows of one more update
    against their float64 recomputation from the device's own S_k; diagonal, range, symmetry, identity rows."""
    from simrank_amd.engine import Plan
    df = synth.WORKLOADS[workload][0]()
    nodes, csr = ingest.directed(df, False, "from", "to", "weight")
    n = csr.n_rows
    if workload == "pl32768d32":
        assert csr.nnz == 32 * 32768
    plan = Plan(ops, csr, coef=0.8)
    assert plan.run(iters, 0.0) == (iters, None)
    rows = _sampled(csr)
    want, _ = _want_rows(csr, csr.rowscale, 0.8, rows, plan.rows)
    assert plan.step(0.0) > 0
    got = plan.rows(rows).astype(np.float64)
    for k, a in enumerate(rows):
        np.testing.assert_allclose(got[k], want[a], rtol=RTOL, atol=1e-30)
        assert got[k][a] == 1.0 and got[k].min() >= 0.0 and got[k].max() <= 1.0
    # symmetry: both triangles come from different tiles and summation orders
    lo, hi = list(range(0, 256)), list(range(n - 256, n))
    A, B = plan.rows(lo)[:, hi], plan.rows(hi)[:, lo]
    np.testing.assert_allclose(A, B.T, rtol=RTOL, atol=1e-30)
    lonely = np.flatnonzero(np.diff(csr.rowptr) == 0)[:5]        # nodes without in-edges keep the identity row (quirk Q6)
    if lonely.size:
        assert np.all(plan.rows(lonely).sum(axis=1) == 1.0)
    plan.free()


# ---------------------------------------------------------------------------------------------------------------------
# config 5: N = 65536 SimRank++ through the C plan and through fit(top_k=10), f32 and fp16-held
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("storage", ["f32", "fp16"])
def test_config5_pl65536_simrank_pp_through_the_plan_and_fit(ops, storage):
    """BASELINE.json configs[4] on one GPU through the shipped loops: `engine.Plan(evidence=True)` — rows of one more update
    against float64 incl. the evidence factor 1 - 2^-|common in-neighbours|, support inside supp(E), symmetry — and
    `SimRankPP().fit(top_k=10)`: its convergence index is the plan's, its top-10 lists are a host selection on the plan's
    rows.  fp16-held matrices with the bar that mode states (a few fp16 roundings: 2e-3 relative)."""
    from simrank_amd.engine import Plan
    df = synth.WORKLOADS["pl65536"][0]()
    nodes, csr = ingest.directed(df, False, "from", "to", "weight")
    n = csr.n_rows
    assert n == 65536
    scale = ingest.spread(csr) * csr.rowscale                      # _cal_Weight, SimRank.py:322-337
    tol = dict(rtol=RTOL, atol=1e-30) if storage == "f32" else dict(rtol=2e-3, atol=1e-9)
    plan = Plan(ops, csr, scale, coef=0.8, evidence=True, storage=storage)
    assert plan.run(3, 0.0) == (3, None)
    rows = _sampled(csr)
    want, counts = _want_rows(csr, scale, 0.8, rows, plan.rows, evidence=True)
    assert plan.step(0.0) > 0
    got = plan.rows(rows).astype(np.float64)
    for k, a in enumerate(rows):
        np.testing.assert_allclose(got[k], want[a], **tol)
        outside = counts[a] == 0                                   # S stays inside supp(E), quirk Q6
        outside[a] = False
        assert np.all(got[k][outside] == 0.0)
        assert got[k].min() >= 0.0 and got[k].max() <= 1.0 and got[k][a] == 1.0
    lo, hi = list(range(0, 256)), list(range(n - 256, n))
    A, B = plan.rows(lo)[:, hi], plan.rows(hi)[:, lo]
    np.testing.assert_allclose(A, B.T, rtol=RTOL, atol=1e-30)
    # to eps, then top-10 on the device against a host selection on the same rows
    done, conv = plan.run(100, 1e-4)
    final = plan.rows(rows)
    idx, val = plan.topk(10)
    for k, a in enumerate(rows):
        order, vals = _host_topk(final[k], a, 10)
        live = vals > 0                                            # (ties among exact zeros: any order of ids)
        assert list(idx[a][live]) == list(order[live])
        np.testing.assert_array_equal(val[a], vals)
    plan.free()
    # the class surface: the same loop, the same lists
    est = SRA.SimRankPP()
    frame = est.fit(df, verbose=False, top_k=10, storage_precision=storage)
    assert est.converged_at == conv
    lab = list(nodes)
    for k, a in enumerate(rows):
        mine = frame[frame["node"] == lab[a]].sort_values("rank")
        order, vals = _host_topk(final[k], a, 10)
        keep = vals > 0
        assert list(mine["neighbor"][: int(keep.sum())]) == [lab[i] for i in order[keep]]
        np.testing.assert_array_equal(mine["similarity"].to_numpy()[: int(keep.sum())], vals[keep].astype(mine["similarity"].dtype))


# ---------------------------------------------------------------------------------------------------------------------
# the sharded C loops at full size, eight virtual ranks
# ---------------------------------------------------------------------------------------------------------------------
def _columns_of(sp_, ranks):
    """{caller's column id: float64 column} of the blocks of `ranks`."""
    out = {}
    for r in ranks:
        blk, ids = sp_.block(r)
        for j, c in enumerate(ids[:: max(1, len(ids) // 6)]):       # six columns per block
            out[int(c)] = blk[:, list(ids).index(c)]
    return out


@pytest.mark.parametrize("workload,pp", [("pl32768d32", False), ("pl65536", True)])
@pytest.mark.parametrize("form,stages,wire,storage", [(0, 1, False, "f32"), (1, 2, False, "f32"), (1, 1, True, "f32"), (0, 1, False, "fp16")])
def test_sharded_c_loop_at_full_size_on_eight_ranks(ops, workload, pp, form, stages, wire, storage):
    """`simrank_shardplan_*` on eight virtual ranks at BASELINE configs 4 and 5 (f32 full / half form staged, fp16 wire,
    fp16-held): three updates; sampled columns of the first and last rank against the float64 recomputation from the plan's
    own previous iterate (S is symmetric: a column is a row), and the convergence counts against the single-rank plan's
    within the counts' own tolerance; in f32 the full form is BIT-EQUAL to the Python choreography (driver.Solver) that makes
    the same launches."""
    from simrank_amd.engine import Plan, ShardPlans
    if storage == "fp16" and form:
        pytest.skip("fp16-held shards run leg 2 in its full form")
    df = synth.WORKLOADS[workload][0]()
    nodes, csr = ingest.directed(df, False, "from", "to", "weight")
    n = csr.n_rows
    scale = ingest.spread(csr) * csr.rowscale if pp else csr.rowscale
    P = 8
    sp8 = ShardPlans(ops, csr, rowscale=scale, world=P, evidence=pp, leg2_form=form, stages=stages, wire_fp16=wire, storage=storage)
    assert sp8.info()["half_form"] == bool(form)
    counts = [sp8.step(1e-4, exact_count=True) for _ in range(2)]
    prev = _columns_of(sp8, (0, P - 1))                               # columns of S_2 ...
    ids = sorted(prev)
    # ... whose neighbours' columns the recomputation needs: fetch every column the sampled ones gather (S symmetric:
    # column c of S_2 = row c), from the single-rank plan at the same iterate — the two agree to rounding, checked below
    one = Plan(ops, csr, scale, coef=0.8, evidence=pp, storage=storage)
    one.run(2, 0.0)
    same = one.rows(ids).astype(np.float64)
    loose = storage == "fp16" or wire
    for k, c in enumerate(ids):
        np.testing.assert_allclose(prev[c], same[k], **(dict(rtol=5e-3, atol=1e-6) if loose else dict(rtol=RTOL, atol=1e-30)))
    want, _ = _want_rows(csr, scale, 0.8, ids, one.rows, evidence=pp)
    counts.append(sp8.step(1e-4, exact_count=True))
    got = _columns_of(sp8, (0, P - 1))
    for c in ids:
        np.testing.assert_allclose(got[c], want[c], **(dict(rtol=5e-3, atol=1e-6) if loose else dict(rtol=RTOL, atol=1e-30)))
        assert got[c][c] == 1.0
    one_counts = []
    one.reset()
    for _ in range(3):
        one_counts.append(one.step(1e-4, exact_count=True))
    one.free()
    assert all(c > 0 for c in counts)
    if not loose:          # moved-element counts of the two loops: equal up to elements within rounding of eps
        for a, b in zip(counts, one_counts):
            assert abs(a - b) <= max(64, a // 100000), (counts, one_counts)
    if storage == "f32" and not wire and workload == "pl32768d32":
        # the Python choreography (kernel by kernel through the ABI) makes the same launches on the same node orders: the
        # last rank's block carries the same BITS (three of its own nodes' rows x all its columns), in both forms
        from tests.pydriver import LocalWorld, SideSpec, Solver
        blk, bids = sp8.block(P - 1)
        s = Solver(lambda r: ops, LocalWorld(P, symmetric_shards=bool(form)),
                   [SideSpec(csr, scale, 0.8, evidence_from=csr if pp else None)], "sparse")
        s.reset()
        for _ in range(3):
            s.step(1e-4)
        pos = np.asarray(s.inv[0])[bids[:3]]
        py = ops.download_rows(s.cur[0][P - 1], [int(v) for v in pos]).astype(np.float64)
        s.release()
        assert np.array_equal(py, blk[bids[:3]]), (form, stages)
    sp8.free()


def test_config3_ml1m_sharded_bipartite_on_eight_ranks(ops):
    """BASELINE.json configs[2] (MovieLens-1M-shaped, BipartiteSimRankPP with the corrected Evidence_N2) through
    `simrank_shardbiplan_*` on eight virtual ranks, full and half form: both matrices against the single-GPU two-matrix plan
    (`simrank_biplan_*`, what fit() runs) on sampled rows after three loop bodies, and its convergence index to eps."""
    from simrank_amd.engine import BiPlan, ShardBiPlans
    df = synth.WORKLOADS["ml1m"][0]()
    _, _, _, _, g12, g21 = ingest.bipartite(df, False, "user", "item", "weight")
    rs1, rs2 = ingest.spread(g12) * g12.rowscale, ingest.spread(g21) * g21.rowscale
    bp = BiPlan(ops, g12, rs1, rs2, evidence=True)
    assert bp.run(3, 0.0) == (3, None)
    rows1, rows2 = _sampled(g12, 7), _sampled(g21, 7)
    want1, want2 = bp.rows(1, rows1), bp.rows(2, rows2)
    k_one = bp.run(100, 1e-4)
    bp.free()
    for form in (0, 1):
        sb = ShardBiPlans(ops, g12, rs1, rs2, world=8, evidence=True, leg2_form=form, stages=2)
        assert sb.run(3, 0.0) == (3, None)
        s1, s2 = sb.result(1), sb.result(2)
        np.testing.assert_allclose(s1[rows1], want1, rtol=RTOL, atol=1e-30)
        np.testing.assert_allclose(s2[rows2], want2, rtol=RTOL, atol=1e-30)
        assert np.array_equal(np.diag(s1), np.ones(g12.n_rows)) and np.array_equal(np.diag(s2), np.ones(g21.n_rows))
        assert sb.run(100, 1e-4) == k_one
        sb.free()
