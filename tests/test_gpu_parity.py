"""End-to-end parity on a real MI355X: the estimators (class surface -> ingest -> C ABI ->
HIP kernels) against (1) every golden vector the reference produced, (2) the oracle on
seeded graphs at sizes it finishes in seconds, (3) size-independent properties at
BASELINE.json's full sizes.  float32 S entries within 1e-5 relative; labels, convergence
iteration, console text and the SimRank++ Evidence/Weight attributes exact."""
import numpy as np
import pytest
import scipy.sparse as sp

import simrank_amd.SimRank as SRA
from oracle import simrank_oracle as O
from simrank_amd import ingest, synth
from tests.pydriver import LocalWorld, SideSpec, Solver
from tests.conftest import Golden, free_port, golden_names
from tests.graphs import bipartite_random
from tests.helpers import RTOL, assert_close, check_against_golden, run_estimator

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from simrank_amd.engine import HipOps
    return HipOps(0)


@pytest.mark.parametrize("name", golden_names())
def test_golden_vectors(name):
    g = Golden(name)
    if g.raises:
        with pytest.raises(ValueError):
            run_estimator(g)
        return
    est, res, text = run_estimator(g)
    check_against_golden(g, est, res, text)


@pytest.mark.parametrize("mode", ["sparse", "dense", "hybrid"])
@pytest.mark.parametrize("name", ["SimRank_er128", "SimRank_pl256", "SimRankPP_er64_weighted",
                                  "SimRankPP_quirky", "AprioriSimRank_er64", "AprioriSimRank_er64_asym",
                                  "BipartiteSimRank_b5030", "BipartiteSimRank_k10",
                                  "BipartiteSimRankPP_b40", "BipartitleAprioriSimRank_b40", "BipartitleAprioriSimRank_b40_asym"])
def test_golden_vectors_in_every_mode(name, mode):
    g = Golden(name)
    est, res, text = run_estimator(g, mode=mode)
    assert est.engine_mode == ("sparse" if name.endswith("_asym") else mode)
    check_against_golden(g, est, res, text, check_attrs=False)


def test_auto_mode_is_the_gather_legs_at_every_density():
    """mode="auto": simrank_spmm for both legs (dense blocks of the pattern go to the matrix
    cores inside it); the f32 GEMM legs are an explicit choice."""
    est, _, _ = run_estimator(Golden("BipartiteSimRank_k10"))       # complete bipartite
    assert est.engine_mode == "sparse"
    est, _, _ = run_estimator(Golden("SimRank_er256"))               # 3 % dense
    assert est.engine_mode == "sparse"


@pytest.mark.parametrize("world", [2, 3, 4])
@pytest.mark.parametrize("name", ["SimRank_er128", "SimRankPP_quirky", "AprioriSimRank_er64", "AprioriSimRank_er64_asym",
                                  "BipartiteSimRank_b5030", "BipartiteSimRankPP_b40",
                                  "SimRank_toy5"])
def test_logical_shards_on_one_gpu(name, world):
    """P column shards on one device, all-to-all by device copies."""
    g = Golden(name)
    est, res, text = run_estimator(g, world=LocalWorld(world), mode="sparse")
    check_against_golden(g, est, res, text)


def test_logical_shards_are_bitwise_equal_to_one_shard():
    from simrank_amd.engine import HipOps
    df = synth.er_directed(1024, 0.01, seed=1)
    knob = HipOps(0)
    # P = 1 without the single-rank shortcut of leg 2 (no upper triangle); leg 1 is the one-launch leg on one rank
    # and on every shard (round 4: the same plan on a rank's row-major column block = the same bits)
    knob.set_tuning(triangle=0)
    try:
        one = SRA.SimRank().fit(df, iterations=6, eps=0, verbose=False, mode="sparse")
    finally:
        knob.set_tuning(triangle=1)
    knob.set_tuning(fuse_shards=0)      # (the gather kernels the shards ran until round 3: other summation order)
    try:
        old = SRA.SimRank().fit(df, iterations=6, eps=0, verbose=False, mode="sparse", world=LocalWorld(4, symmetric_shards=False))
    finally:
        knob.set_tuning(fuse_shards=1)
    np.testing.assert_allclose(old.values, one.values, rtol=1e-6, atol=1e-30)
    tri = SRA.SimRank().fit(df, iterations=6, eps=0, verbose=False, mode="sparse")
    # mirrored tiles carry the same bits (only the diagonal tiles of the solver's own node
    # order are computed on both sides): nearly all pairs (i, j), (j, i) are bit-equal
    assert (tri.values == tri.values.T).mean() > 0.95
    np.testing.assert_allclose(tri.values, one.values, rtol=1e-6, atol=1e-30)
    for world in (2, 4, 8):
        many = SRA.SimRank().fit(df, iterations=6, eps=0, verbose=False, mode="sparse",
                                 world=LocalWorld(world, symmetric_shards=False))
        assert np.array_equal(one.values, many.values)
        # N = 1024 is a multiple of 32 x P: by default leg 2 runs in its half form (nodes dealt to
        # the shards, tiles i <= j, mirrored tiles exchanged) — other summation order, same values
        half = SRA.SimRank().fit(df, iterations=6, eps=0, verbose=False, mode="sparse",
                                 world=LocalWorld(world))
        np.testing.assert_allclose(half.values, one.values, rtol=1e-6, atol=1e-30)
        # (tiles i == j of two different shards are computed on both sides: 1/4 of them at P = 8)
        assert (half.values == half.values.T).mean() > 0.7
        # the half form in stages of column tiles (how exchange 2 is overlapped on a real node): same bits
        if 1024 // (32 * world) >= 4:
            staged = SRA.SimRank().fit(df, iterations=6, eps=0, verbose=False, mode="sparse",
                                       world=LocalWorld(world, leg2_stages=3))
            assert np.array_equal(staged.values, half.values)


def test_runs_are_bitwise_reproducible():
    df = synth.powerlaw_directed(2048, 16, seed=3)
    a = SRA.SimRankPP().fit(df, iterations=5, eps=0, verbose=False)
    b = SRA.SimRankPP().fit(df, iterations=5, eps=0, verbose=False)
    assert np.array_equal(a.values, b.values)


@pytest.mark.parametrize("kind", ["er", "powerlaw"])
@pytest.mark.parametrize("cls", ["SimRank", "SimRankPP"])
def test_midsize_against_oracle(kind, cls):
    df = (synth.er_directed(2048, 0.004, seed=11) if kind == "er"
          else synth.powerlaw_directed(2048, 24, seed=12))
    got = getattr(SRA, cls)().fit(df, verbose=False)
    want = (O.fit_simrank if cls == "SimRank" else O.fit_simrank_pp)(df, verbose=False)
    assert list(got.index) == want["labels"]
    assert_close(got.values, want["S"])


def test_two_different_graphs_back_to_back_on_pooled_blocks():
    """Blocks of >= 64 MiB (N >= 4096) come back from the library's pool UN-ZEROED: two different graphs of equal N
    — a power-law SimRank++ (evidence) and an Erdos-Renyi AprioriSimRank (symmetric prior) — fitted one after the
    other in one process, through ``fit`` and through the C-level plan, each against the oracle's loop
    (SimRank.py:351-362, :443-454).  A stale byte in a padding row or a reused evidence / prior buffer shows here."""
    from simrank_amd.engine import HipOps, Plan
    n, its = 4096, 3
    rng = np.random.default_rng(5)
    cases = []
    for kind in ("pl", "er"):
        df = synth.powerlaw_directed(n, 16, seed=41) if kind == "pl" else synth.er_directed(n, 0.0025, seed=42)
        labels, G = O.directed_graph(df)
        W = O.weight(G)
        pat = (G > 0).astype(np.float64)
        E = 1 - 0.5 ** (pat @ pat.T)                       # (= O.evidence: the counts are exact in float64; BLAS instead of an int64 matmul)
        prior = None
        if kind == "er":
            prior = rng.random((n, n))
            prior = (prior + prior.T) / 2
        want, _ = O.iterate_directed(W, C=0.8, iterations=its, eps=0.0, E=E, apriori=prior, lbd=0.3 if prior is not None else None)
        cases.append((kind, df, labels, prior, want))
    ops = HipOps(0)
    ops.trim_pool()

    def fit(kind, df, labels, prior, want):
        if prior is None:
            got = SRA.SimRankPP().fit(df, iterations=its, eps=0.0, verbose=False)
        else:
            got = SRA.AprioriSimRank().fit(df, prior, lbd=0.3, iterations=its, eps=0.0, verbose=False)
        assert list(got.index) == list(labels)
        assert_close(got.values, want)

    def plan(kind, df, labels, prior, want):
        _, csr = ingest.directed(df, False, "from", "to", "weight")
        pl = Plan(ops, csr, ingest.spread(csr) * csr.rowscale, coef=0.8, evidence=True, apriori=prior, lbd=0.3 if prior is not None else 0.0)
        pl.run(its, 0.0)
        got = pl.result()
        pl.free()
        # (plan results are in the CSR's node order = ingest's = the oracle's label order)
        assert_close(got, want)

    fit(*cases[0])
    # the first fit's matrices are at rest in the pool (SIMRANK_POOL_GIB=0 turns the pool off: then the rest still holds)
    assert ops.pool_stats()[1] > 0 or ops.pool_stats()[2] == 0
    fit(*cases[1])
    plan(*cases[0])
    plan(*cases[1])
    fit(*cases[0])
    plan(*cases[1])


def test_midsize_weighted_and_prior_against_oracle():
    df = synth.er_directed(1500, 0.006, seed=13)
    rng = np.random.default_rng(0)
    prior = rng.random((1500, 1500))
    prior = (prior + prior.T) / 2
    got = SRA.AprioriSimRank().fit(df, prior, lbd=0.2, weighted=True, verbose=False)
    want = O.fit_simrank_pp(df, apriori=prior, lbd=0.2, weighted=True, verbose=False)
    assert_close(got.values, want["S"])


def test_midsize_bipartite_against_oracle():
    df = bipartite_random(900, 500, 0.03, seed=14)
    est = SRA.BipartiteSimRankPP()
    s1, s2 = est.fit(df, verbose=False, strict_reference=False)
    want = O.fit_bipartite_pp(df, verbose=False, strict_reference=False)
    assert list(s1.index) == want["sorted1"]
    assert_close(s1.values, want["S1"])
    assert_close(s2.values, want["S2"])
    assert est.converged_at == want["k"]
    plain = SRA.BipartiteSimRank().fit(df, verbose=False)
    wantp = O.fit_bipartite(df, verbose=False)
    assert_close(plain[0].values, wantp["S1"])
    assert_close(plain[1].values, wantp["S2"])


# ---------------------------------------------------------------------------------------
# BASELINE.json sizes: properties that do not need a dense float64 oracle run
# ---------------------------------------------------------------------------------------
def _sampled_rows_check(ops, solver, csr, rows, coef):
    """Rows of one more update recomputed on the host in float64 from the device's S_k:
    S_{k+1}[a, :] = coef . (W[a, :] . S_k) . W^T,  diag <- 1."""
    need = sorted(set(np.concatenate([csr.col[csr.rowptr[a]:csr.rowptr[a + 1]] for a in rows])))
    pos = {int(i): p for p, i in enumerate(need)}
    part = ops.download_rows(solver.cur[0][0], need).astype(np.float64)
    rs = csr.rowscale.astype(np.float32).astype(np.float64)
    W = sp.diags(rs) @ sp.csr_matrix((np.ones(csr.col.size), csr.col, csr.rowptr),
                                     shape=(csr.n_rows, csr.n_cols))
    t_rows = {}
    for a in rows:                         # t = W[a, :] . S_k : the rows a gathers
        idx = [pos[int(i)] for i in csr.col[csr.rowptr[a]:csr.rowptr[a + 1]]]
        t_rows[a] = rs[a] * part[idx].sum(axis=0)
    solver.step(0.0)
    S_n = ops.download(solver.cur[0][0])
    for a in rows:
        want = coef * (W @ t_rows[a])
        want[a] = 1.0
        np.testing.assert_allclose(S_n[a], want, rtol=RTOL, atol=1e-30)
    return S_n


@pytest.mark.parametrize("workload,iters", [("er8192", 8), ("pl32768", 3), ("pl32768d32", 3)])
def test_full_size_properties(ops, workload, iters):
    """BASELINE.json configs[1] and configs[3] at full size; pl32768d32 is config 4 as it is stated (mean
    degree 32 after de-duplication: 1 048 576 edges), pl32768 the lighter graph of the SURVEY.md recipe."""
    df = synth.WORKLOADS[workload][0]()
    nodes, csr = ingest.directed(df, False, "from", "to", "weight")
    n = csr.n_rows
    assert n == int(workload[2:].replace("d32", ""))
    if workload == "pl32768d32":
        assert csr.nnz == 32 * 32768
    solver = Solver(lambda r: ops, LocalWorld(1), [SideSpec(csr, csr.rowscale, 0.8)], "sparse")
    csr = solver.specs[0].csr       # the device buffers are in the solver's own node order
    assert np.all(np.diff(np.diff(csr.rowptr)) >= 0)            # ascending row length
    solver.reset()
    for _ in range(iters):
        solver.step(0.0)
    # 24 sampled rows of one more update within 1e-5 of a float64 recomputation (the ends, the longest row, random ones)
    rows = sorted({0, 1, n // 3, n - 1, int(np.argmax(np.diff(csr.rowptr)))} |
                  set(int(r) for r in np.random.default_rng(n).choice(n, size=19, replace=False)))
    S = _sampled_rows_check(ops, solver, csr, rows, 0.8)
    assert np.array_equal(np.diag(S), np.ones(n, dtype=np.float32))
    assert S.min() >= 0.0 and S.max() <= 1.0
    # symmetry: both triangles come from different summation orders
    blk = slice(0, 4096)
    np.testing.assert_allclose(S[blk, :], S[:, blk].T, rtol=RTOL, atol=1e-30)
    # nodes without in-edges keep the identity row (quirk Q6)
    lonely = np.flatnonzero(np.diff(csr.rowptr) == 0)[:5]
    for a in lonely:
        assert S[a].sum() == 1.0
    solver.release()


def test_n98304_beyond_16_bit_ids_and_32_bit_element_offsets(ops):
    """N = 98 304: more than 65 536 nodes (neighbour ids no longer fit the 16-bit id stream) and
    N^2 = 9.7e9 elements per matrix (element offsets beyond 2^32; at N = 65 536 they still fit).
    Three updates, then rows of one more update recomputed on the host in float64 from the device's
    S_k, the unit diagonal and symmetry on sampled rows/columns."""
    n = 98304
    df = synth.er_directed(n, 8.0 / n, seed=98304)
    nodes, csr = ingest.directed(df, False, "from", "to", "weight")
    assert csr.n_rows == n
    solver = Solver(lambda r: ops, LocalWorld(1), [SideSpec(csr, csr.rowscale, 0.8)], "sparse")
    assert solver.blocked
    csr = solver.specs[0].csr
    solver.reset()
    for _ in range(3):
        solver.step(0.0)
    rows = [0, 1, 40000, 65535, 65536, 70001, n - 2, n - 1]
    need = sorted(set(np.concatenate([csr.col[csr.rowptr[a]:csr.rowptr[a + 1]] for a in rows])))
    pos = {int(i): p for p, i in enumerate(need)}
    part = ops.download_rows(solver.cur[0][0], need).astype(np.float64)
    rs = csr.rowscale.astype(np.float32).astype(np.float64)
    W = sp.diags(rs) @ sp.csr_matrix((np.ones(csr.col.size), csr.col, csr.rowptr), shape=(n, n))
    t_rows = {a: rs[a] * part[[pos[int(i)] for i in csr.col[csr.rowptr[a]:csr.rowptr[a + 1]]]].sum(axis=0)
              for a in rows}
    changed = solver.step(0.0)
    got = ops.download_rows(solver.cur[0][0], rows)
    for k, a in enumerate(rows):
        want = 0.8 * (W @ t_rows[a])
        want[a] = 1.0
        np.testing.assert_allclose(got[k], want, rtol=RTOL, atol=1e-30)
        assert got[k][a] == 1.0
    assert 0 < changed <= n * n
    # symmetry across the 2^32-element line: column a of the sampled rows against row a
    sub = got[:, rows]
    np.testing.assert_allclose(sub, sub.T, rtol=RTOL, atol=1e-30)
    top = solver.topk(0, 3)
    assert top[0].shape == (n, 3)
    solver.release()


@pytest.mark.parametrize("storage", ["f32", "fp16"])
def test_config5_pl65536_simrank_pp_properties(ops, storage):
    """(storage = "fp16": the same checks on matrices HELD in fp16 — config 5's reduced-precision mode,
    csrc/half.hip — with the bar that mode states: a few fp16 roundings, 2e-3 relative.)
    BASELINE.json configs[4] shape: N = 65536 power-law graph, SimRank++ with evidence
    (SimRank.py:351-362, evidence :311-320), f32, on one GPU.  The dense f64 oracle would need
    100 GB and hours, so: rows of one more update recomputed on the host in float64 from the
    device's S_k — evidence factor 1 - 2^-|common in-neighbours| included — plus the
    size-independent properties (unit diagonal, range, symmetry, support inside supp(E),
    identity rows of nodes without in-edges)."""
    df = synth.WORKLOADS["pl65536"][0]()
    nodes, csr = ingest.directed(df, False, "from", "to", "weight")
    n = csr.n_rows
    assert n == 65536
    scale = ingest.spread(csr) * csr.rowscale                      # _cal_Weight, SimRank.py:322-337
    solver = Solver(lambda r: ops, LocalWorld(1),
                    [SideSpec(csr, scale, 0.8, evidence_from=csr, storage=storage)], "sparse")
    tol = dict(rtol=RTOL, atol=1e-30) if storage == "f32" else dict(rtol=2e-3, atol=1e-9)
    csr = solver.specs[0].csr                                      # the solver's own node order
    rs = np.asarray(solver.specs[0].rowscale, dtype=np.float32).astype(np.float64)
    solver.reset()
    for _ in range(3):
        solver.step(0.0)
    rows = [0, 5, n // 2, n - 2, n - 1]                            # short rows ... the longest rows
    need = sorted(set(np.concatenate([csr.col[csr.rowptr[a]:csr.rowptr[a + 1]] for a in rows])))
    pos = {int(i): p for p, i in enumerate(need)}
    part = ops.download_rows(solver.cur[0][0], need).astype(np.float64)
    Pat = sp.csr_matrix((np.ones(csr.col.size), csr.col, csr.rowptr), shape=(n, n))
    W = sp.diags(rs) @ Pat
    live = sp.diags((rs > 0).astype(np.float64)) @ Pat            # evidence counts ignore dead rows
    t_rows = {a: rs[a] * part[[pos[int(i)] for i in csr.col[csr.rowptr[a]:csr.rowptr[a + 1]]]].sum(axis=0)
              for a in rows}
    solver.step(0.0)
    got = ops.download_rows(solver.cur[0][0], rows).astype(np.float64)
    for k, a in enumerate(rows):
        cnt = np.asarray((live[a] @ live.T).todense()).ravel()
        want = 0.8 * (W @ t_rows[a]) * (1.0 - 0.5 ** cnt)
        want[a] = 1.0
        np.testing.assert_allclose(got[k], want, **tol)
        outside = cnt == 0                                         # S stays inside supp(E), quirk Q6
        outside[a] = False
        assert np.all(got[k][outside] == 0.0)
        assert got[k].min() >= 0.0 and got[k].max() <= 1.0 and got[k][a] == 1.0
    # symmetry between two far-apart row blocks (both triangles come from different tiles)
    lo, hi = list(range(0, 256)), list(range(n - 256, n))
    A = ops.download_rows(solver.cur[0][0], lo)[:, hi]
    B = ops.download_rows(solver.cur[0][0], hi)[:, lo]
    np.testing.assert_allclose(A, B.T, rtol=RTOL, atol=1e-30)
    if storage == "fp16":
        assert np.array_equal(A, B.T)                              # one triangle is the other's mirror image
    lonely = np.flatnonzero(np.diff(csr.rowptr) == 0)[:3]
    if lonely.size:
        L = ops.download_rows(solver.cur[0][0], lonely)
        assert np.all(L.sum(axis=1) == 1.0)
    solver.release()


@pytest.mark.parametrize("workload", ["pl32768d32", "pl32768"])
def test_config4_pl32768_eight_shards_bitwise(ops, workload):
    """BASELINE.json configs[3] in its stated form — S sharded 8 ways — at full size (pl32768d32: mean
    degree 32 after de-duplication; pl32768: the lighter graph of rounds 1-2): eight virtual ranks on the
    one GPU (all-to-all by device copies) against one rank on the same kernels; sampled rows must carry
    the same bits."""
    df = synth.WORKLOADS[workload][0]()
    nodes, csr = ingest.directed(df, False, "from", "to", "weight")
    n = csr.n_rows
    rows = [0, 1, n // 3, n // 2 + 7, n - 129, n - 1]
    ops.set_tuning(triangle=0)   # one rank on the kernels the shards run (no triangle leg 2; the one-launch leg 1 on both)
    try:
        one = Solver(lambda r: ops, LocalWorld(1), [SideSpec(csr, csr.rowscale, 0.8)], "sparse")
        one.reset()
        for _ in range(3):
            one.step(0.0)
        want = ops.download_rows(one.cur[0][0], rows)
        one.release()
    finally:
        ops.set_tuning(triangle=1)
    inv_one = one.inv[0]
    del one
    world = LocalWorld(8, symmetric_shards=False)
    many = Solver(lambda r: ops, world, [SideSpec(csr, csr.rowscale, 0.8)], "sparse")
    many.reset()
    for _ in range(3):
        many.step(0.0)
    got = np.concatenate([ops.download_rows(many.cur[0][r], rows) for r in world.local_ranks], axis=1)
    many.release()
    assert got.shape == want.shape
    assert np.array_equal(got, want)
    assert np.all(want[np.arange(len(rows)), rows] == 1.0)
    # the single rank as the solver really runs it (one-launch leg 1 + upper-triangle leg 2): same values
    # up to float32 summation order
    fast = Solver(lambda r: ops, LocalWorld(1), [SideSpec(csr, csr.rowscale, 0.8)], "sparse")
    fast.reset()
    for _ in range(3):
        fast.step(0.0)
    np.testing.assert_allclose(ops.download_rows(fast.cur[0][0], rows), want, rtol=RTOL, atol=1e-30)
    fast.release()
    # the same with leg 2 in its half form (tiles i <= j + a second, half-size exchange of the mirrored
    # tiles; nodes dealt to the shards in tiles of 32): same values up to float32 summation order
    world = LocalWorld(8)
    half = Solver(lambda r: ops, world, [SideSpec(csr, csr.rowscale, 0.8)], "sparse")
    assert all(sd.shard_sym for sd in half.sides[0].values())
    half.reset()
    for _ in range(3):
        half.step(0.0)
    nodes_ = np.argsort(inv_one)[rows]                # the caller's ids of the sampled rows
    hrows = [int(half.inv[0][a]) for a in nodes_]
    got = np.concatenate([ops.download_rows(half.cur[0][r], hrows) for r in world.local_ranks], axis=1)
    half.release()
    got = got[:, half.inv[0]][:, np.argsort(inv_one)]   # columns: dealt order -> caller's -> ascending
    np.testing.assert_allclose(got, want, rtol=RTOL, atol=1e-30)


@pytest.mark.parametrize("wire", ["f32", "fp16"])
def test_config5_pl65536_simrank_pp_eight_shards(ops, wire):
    """BASELINE.json configs[4] in its stated sharded form: N = 65536 SimRank++ (evidence counts, spread
    weights) over eight virtual ranks on the one GPU, leg 2 in its half form with the second exchange —
    against the single-rank run on sampled rows (same values up to float32 summation order), and the
    convergence counts of both must be non-zero together.  ``wire="fp16"``: the config's reduced precision on
    the links (exchange buffers travel as fp16 x 2^14, the kernels stay f32) — two fp16 roundings per update
    (transposed product, mirrored tiles), so the sampled rows agree to a few fp16 spacings instead of 1e-5."""
    df = synth.WORKLOADS["pl65536"][0]()
    nodes, csr = ingest.directed(df, False, "from", "to", "weight")
    n = csr.n_rows
    scale = ingest.spread(csr) * csr.rowscale

    def run(world):
        s = Solver(lambda r: ops, world, [SideSpec(csr, scale, 0.8, evidence_from=csr)], "sparse")
        s.reset()
        moved = [s.step(1e-4) for _ in range(2)]
        return s, moved
    one, moved_one = run(LocalWorld(1))
    rows = [0, 3, n // 2, n - 200, n - 1]                      # positions in the single-rank order
    want = ops.download_rows(one.cur[0][0], rows)
    order_one, inv_one = one.order[0], one.inv[0]
    one.release()
    del one
    world = LocalWorld(8, exchange_precision=wire)
    half, moved_half = run(world)
    assert all(sd.shard_sym for sd in half.sides[0].values())
    hrows = [int(half.inv[0][a]) for a in order_one[rows]]      # the same nodes in the dealt order
    got = np.concatenate([ops.download_rows(half.cur[0][r], hrows) for r in world.local_ranks], axis=1)
    inv_half = half.inv[0]
    half.release()
    got = got[:, inv_half][:, order_one]                         # columns: dealt -> caller's -> single-rank order
    if wire == "fp16":
        # (values below 2^-14 / 2^14 = 2^-28 sit in fp16's subnormal range on the wire: absolute spacing 2^-38)
        np.testing.assert_allclose(got, want, rtol=3e-3, atol=2.0 ** -36)
        assert np.abs(got - want).max() > 0                     # really rounded
    else:
        np.testing.assert_allclose(got, want, rtol=RTOL, atol=1e-30)
    assert np.all(got[np.arange(len(rows)), rows] == 1.0)
    assert all(a > 0 for a in moved_one) and all(b > 0 for b in moved_half)


def test_config3_movielens_shaped_bipartite_pp(ops):
    """6040 x 3706, ~1.0 M ratings: runs only with the corrected
    Evidence_N2 (the reference raises, quirk Q2); checked through sampled rows of one update
    recomputed on the host in float64 and the identity PP-with-E==1 == plain."""
    df = synth.WORKLOADS["ml1m"][0]()
    with pytest.raises(ValueError, match="broadcast"):
        SRA.BipartiteSimRankPP().fit(df, iterations=1, verbose=False)
    est = SRA.BipartiteSimRankPP()
    s1, s2 = est.fit(df, iterations=2, eps=0, verbose=False, strict_reference=False)
    assert s1.shape == (6040, 6040) and s2.shape == (3706, 3706)
    g12, g21 = est._csr

    def W(c):
        return sp.diags(c.rowscale.astype(np.float32).astype(np.float64)) @ sp.csr_matrix(
            (np.ones(c.col.size), c.col, c.rowptr), shape=(c.n_rows, c.n_cols))
    W12, W21 = W(g12), W(g21)
    # iteration 1 from identity, then iteration 2, for sampled rows of S1 (float64 host)
    P12 = (W12 != 0).astype(np.float64)
    P21 = (W21 != 0).astype(np.float64)
    rows = [0, 17, 3000, 6039]
    S2_0 = sp.identity(3706, format="csr")
    S1_1 = 0.8 * (W12 @ S2_0 @ W12.T).toarray()
    E1 = 1 - 0.5 ** (P12 @ P12.T).toarray()
    S1_1 *= E1
    np.fill_diagonal(S1_1, 1)
    S2_1 = 0.8 * (W21 @ sp.csr_matrix(S1_1) @ W21.T).toarray()
    E2 = 1 - 0.5 ** (P21 @ P21.T).toarray()
    S2_1 *= E2
    np.fill_diagonal(S2_1, 1)
    for a in rows:
        want = 0.8 * (W12 @ (W12[a] @ S2_1).T).ravel() * E1[a]
        want[a] = 1.0
        np.testing.assert_allclose(s1.values[a], want, rtol=RTOL, atol=1e-30)
    # the Gauss-Seidel half (SimRank.py:300-302): S2 of iteration 2 is computed from the NEW S1 of iteration 2
    S1_2 = 0.8 * (W12 @ (W12 @ S2_1).T) * E1          # (S2_1 symmetric: W12 S2 W12^T = W12 (W12 S2)^T)
    np.fill_diagonal(S1_2, 1)
    for b in [0, 5, 1800, 3705]:
        want = 0.8 * (W21 @ (W21[b] @ S1_2).T).ravel() * E2[b]
        want[b] = 1.0
        np.testing.assert_allclose(s2.values[b], want, rtol=RTOL, atol=1e-30)
    assert est.engine_mode in ("sparse", "dense")
    # a 1500 x 900 sub-sample of the same graph run to convergence against the oracle (both matrices, every
    # element, the convergence iteration)
    sub = df[(df["user"] < 1500) & (df["item"] < 900)]
    est2 = SRA.BipartiteSimRankPP()
    t1, t2 = est2.fit(sub, verbose=False, strict_reference=False)
    want = O.fit_bipartite_pp(sub, verbose=False, strict_reference=False)
    assert list(t1.index) == want["sorted1"]
    assert_close(t1.values, want["S1"])
    assert_close(t2.values, want["S2"])
    assert est2.converged_at == want["k"]


def test_integration_stub_of_the_reference_binding():
    """examples/reference_hip_stub.py: ctypes-only binding fed with the
    reference's dense Graph; must reproduce the reference's S and convergence iteration."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                        "examples", "reference_hip_stub.py")
    spec = importlib.util.spec_from_file_location("reference_hip_stub", path)
    stub = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(stub)
    for name in ("SimRank_toy5", "SimRank_er128", "SimRank_quirky_weighted"):
        g = Golden(name)
        S, k = stub.iterate(g.out["G"], g.kwargs.get("C", 0.8), g.kwargs.get("iterations", 100),
                            g.kwargs.get("eps", 1e-4))
        assert_close(S, g.out["S"])
        assert k == g.k
    for name in ("BipartiteSimRank_b40", "BipartiteSimRank_b5030"):
        g = Golden(name)
        S1, S2, k = stub.iterate_bipartite(g.out["G12"], g.out["G21"], g.kwargs.get("C1", 0.8),
                                           g.kwargs.get("C2", 0.8), g.kwargs.get("iterations", 100),
                                           g.kwargs.get("eps", 1e-4))
        assert_close(S1, g.out["S1"])
        assert_close(S2, g.out["S2"])
        assert k == g.k


def test_rccl_world_of_one_rank():
    """bench.py --force-dist under torch.distributed.run with one rank: the TorchWorld path
    (RCCL init, torch-owned exchange buffers handed to the C ABI, three staged asynchronous
    all_to_all_single calls per update, all_reduce, stream hand-over) on the single GPU that is available; its result must be the
    same iterations/s order as the local world and the run must exit cleanly."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.join(root, "bench.py"),
           "--gpus", "1", "--steps", "3", "--warmup", "1", "--workload", "er8192", "--force-dist",
           "--stages", "3", "--no-cpu-baseline"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["value"] > 100 and out["converge"]["iterations"] == 5


def test_rccl_world_values():
    """Values, labels, convergence iteration and stdout through the RCCL world of one rank
    (staged and unstaged stream-ordered exchange), see tests/dist_gpu_worker.py."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(root, "tests", "dist_gpu_worker.py")]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
    assert p.returncode == 0 and "RCCL WORLD ok" in p.stdout, p.stdout[-3000:] + p.stderr[-3000:]


def test_rccl_world_of_two_ranks():
    """The same worker on two GPUs (real all-to-all over xGMI, root-only hand-back); skipped on
    the one-GPU box this suite usually runs on."""
    import os
    import subprocess
    import sys
    from simrank_amd import _lib
    if _lib.device_count() < 2:
        pytest.skip("needs two GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(root, "tests", "dist_gpu_worker.py")]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert p.returncode == 0 and p.stdout.count("RCCL WORLD ok") == 2, p.stdout[-3000:] + p.stderr[-3000:]


@pytest.mark.parametrize("cls", ["SimRank", "SimRankPP"])
def test_directed_edge_cases_on_gpu(cls):
    """Degenerate and extreme graphs (N = 1, stars with one 899-entry row, complete graph,
    chain, string labels) through the HIP path, against the oracle."""
    from tests.edge_cases import directed_cases
    for name, df in directed_cases().items():
        est = getattr(SRA, cls)()
        got = est.fit(df, verbose=False)
        want = (O.fit_simrank if cls == "SimRank" else O.fit_simrank_pp)(df, verbose=False)
        assert list(got.index) == want["labels"], name
        np.testing.assert_allclose(got.values, want["S"], rtol=RTOL, atol=1e-30, err_msg=name)
        assert est.converged_at == want["k"], name
        if name == "complete_40":
            assert est.engine_mode == "sparse"


def test_bipartite_edge_cases_on_gpu():
    from tests.edge_cases import bipartite_cases
    for name, df in bipartite_cases().items():
        est = SRA.BipartiteSimRankPP()
        s1, s2 = est.fit(df, verbose=False, strict_reference=False)
        want = O.fit_bipartite_pp(df, verbose=False, strict_reference=False)
        np.testing.assert_allclose(s1.values, want["S1"], rtol=RTOL, atol=1e-30, err_msg=name)
        np.testing.assert_allclose(s2.values, want["S2"], rtol=RTOL, atol=1e-30, err_msg=name)
        assert est.converged_at == want["k"], name


@pytest.mark.parametrize("seed", range(12))
def test_randomized_against_oracle(seed):
    """Random size / density / class / sharding / mode, every result against the oracle."""
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(33, 700))
    kind = ["er", "pl", "bip"][seed % 3]
    # (virtual ranks: odd seeds through the sharded loops behind the C ABI, even ones through the Python driver)
    world = LocalWorld(int(rng.integers(1, 5)), loop="c" if seed % 2 else "python")
    mode = "sparse" if world.size > 1 else str(rng.choice(["auto", "sparse", "dense", "hybrid"]))
    weighted = bool(rng.integers(0, 2))
    if kind == "bip":
        n2 = int(rng.integers(20, 300))
        df = bipartite_random(n, n2, float(rng.uniform(0.01, 0.2)), seed=seed)
        pp = bool(rng.integers(0, 2))
        est = (SRA.BipartiteSimRankPP if pp else SRA.BipartiteSimRank)()
        kw = dict(weighted=weighted, verbose=False, C1=0.7, C2=0.85)
        extra = dict(strict_reference=False) if pp else {}
        s1, s2 = est.fit(df, world=world, mode=mode, **kw, **extra)
        want = (O.fit_bipartite_pp if pp else O.fit_bipartite)(df, **kw, **extra)
        sorted_labels = pp
        assert list(s1.index) == (want["sorted1"] if sorted_labels else want["labels1"])
        assert_close(s1.values, want["S1"])
        assert_close(s2.values, want["S2"])
    else:
        df = (synth.er_directed(n, float(rng.uniform(0.005, 0.15)), seed) if kind == "er"
              else synth.powerlaw_directed(n, float(rng.uniform(2, 30)), seed))
        cls = str(rng.choice(["SimRank", "SimRankPP", "AprioriSimRank"]))
        kw = dict(weighted=weighted, verbose=False, C=float(rng.uniform(0.5, 0.9)))
        if cls == "AprioriSimRank":
            prior = rng.random((n, n))
            if rng.integers(0, 2):
                prior = (prior + prior.T) / 2
            got = SRA.AprioriSimRank().fit(df, prior, lbd=0.3, world=world, mode=mode, **kw)
            want = O.fit_simrank_pp(df, apriori=prior, lbd=0.3, **kw)
        else:
            est = getattr(SRA, cls)()
            got = est.fit(df, world=world, mode=mode, **kw)
            want = (O.fit_simrank if cls == "SimRank" else O.fit_simrank_pp)(df, **kw)
            assert est.converged_at == want["k"]
        assert list(got.index) == want["labels"]
        assert_close(got.values, want["S"])


@pytest.mark.parametrize("seed", range(10))
def test_randomized_half_form_against_oracle(seed, monkeypatch):
    """Node counts that are multiples of 32 x P, so that leg 2 of every virtual rank runs in its half
    form (simrank_spmm_shard + the exchange of the mirrored tiles + simrank_shard_unpack): random P,
    size, density and class — bipartite, evidence (also support-restricted), symmetric prior —
    against the oracle; the convergence iteration must match too (the counter counts mirrored
    elements twice)."""
    import tests.pydriver as drv
    rng = np.random.default_rng(7000 + seed)
    P = int(rng.choice([2, 3, 4, 8]))
    n = 32 * P * int(rng.integers(1, 7 if P < 8 else 4))
    made = []
    orig = drv.Solver.__init__

    def spy(self, *a, **k):
        orig(self, *a, **k)
        made.append(self)
    monkeypatch.setattr(drv.Solver, "__init__", spy)
    if seed % 3 == 2:
        monkeypatch.setattr(drv, "RESTRICT_BELOW", 2.0)          # always the support-restricted leg 2
    world = LocalWorld(P)
    weighted = bool(rng.integers(0, 2))
    if seed % 5 == 4:
        n2 = 32 * P * int(rng.integers(1, 4))
        pp = bool(rng.integers(0, 2)) and n2 == n                 # (the ++ class needs equal groups: Q2)
        df = bipartite_random(n, n2, float(rng.uniform(0.02, 0.2)), seed=seed)
        est = (SRA.BipartiteSimRankPP if pp else SRA.BipartiteSimRank)()
        kw = dict(weighted=weighted, verbose=False, C1=0.7, C2=0.85)
        s1, s2 = est.fit(df, world=world, mode="sparse", **kw)
        want = (O.fit_bipartite_pp if pp else O.fit_bipartite)(df, **kw)
        assert_close(s1.values, want["S1"])
        assert_close(s2.values, want["S2"])
        assert est.converged_at == want["k"]
    else:
        df = (synth.er_directed(n, float(rng.uniform(0.01, 0.15)), seed) if seed % 2
              else synth.powerlaw_directed(n, float(rng.uniform(3, 30)), seed))
        cls = ["SimRank", "SimRankPP", "AprioriSimRank"][seed % 3]
        kw = dict(weighted=weighted, verbose=False, C=float(rng.uniform(0.5, 0.9)))
        if cls == "AprioriSimRank":
            prior = rng.random((n, n))
            prior = (prior + prior.T) / 2                         # symmetric: the fused path
            got = SRA.AprioriSimRank().fit(df, prior, lbd=0.3, world=world, mode="sparse", **kw)
            want = O.fit_simrank_pp(df, apriori=prior, lbd=0.3, **kw)
        else:
            est = getattr(SRA, cls)()
            got = est.fit(df, world=world, mode="sparse", **kw)
            want = (O.fit_simrank if cls == "SimRank" else O.fit_simrank_pp)(df, **kw)
            assert est.converged_at == want["k"]
        assert list(got.index) == want["labels"]
        assert_close(got.values, want["S"])
    assert made and all(sd.shard_sym for side in made[-1].sides for sd in side.values())


@pytest.mark.parametrize("world", [2, 4])
def test_logical_shards_with_padded_chunks(world, monkeypatch):
    """The padded chunk layout (t_pad) through the real kernels, forced on at small size and
    natural at N = 4096 (blocks of 2048 / 1024 rows)."""
    import tests.pydriver as drv
    g = Golden("SimRankPP_er128")
    monkeypatch.setattr(drv, "PAD_MIN_ROWS", 1)
    monkeypatch.setattr(drv, "PAD_MULTIPLE", 1)
    est, res, text = run_estimator(g, world=LocalWorld(world), mode="sparse")
    check_against_golden(g, est, res, text)
    monkeypatch.undo()
    df = synth.er_directed(4096, 0.003, seed=2)
    one = SRA.SimRank().fit(df, iterations=4, eps=0, verbose=False, mode="sparse")
    many = SRA.SimRank().fit(df, iterations=4, eps=0, verbose=False, mode="sparse",
                             world=LocalWorld(world))
    np.testing.assert_allclose(many.values, one.values, rtol=1e-6, atol=1e-30)


def test_top_k_hand_back_on_gpu():
    """fit(top_k=k): device-side selection, single and sharded, against the dense result."""
    df = synth.powerlaw_directed(1500, 12, seed=21)
    dense = SRA.SimRankPP().fit(df, verbose=False)
    labels = list(dense.index)
    pos = {l: i for i, l in enumerate(labels)}
    for world in (LocalWorld(1), LocalWorld(4)):
        top = SRA.SimRankPP().fit(df, verbose=False, top_k=8, world=world, mode="sparse")
        assert len(top) == 8 * len(labels)
        for node in labels[::97]:
            row = dense.loc[node].drop(node)
            got = top[top.node == node].sort_values("rank")
            # same similarity values as the 8 largest of the dense row (ties may reorder labels)
            want = np.sort(row.values)[::-1][:8]
            np.testing.assert_allclose(got.similarity.values, want, rtol=1e-6, atol=1e-30)
            for nb, sim in zip(got.neighbor, got.similarity):
                np.testing.assert_allclose(dense.iloc[pos[node], pos[nb]], sim, rtol=1e-6, atol=1e-30)


# ---------------------------------------------------------------------------------------
# graphs with a dense corner: part of every leg runs on the matrix cores (csrc/blockdense.hip)
# ---------------------------------------------------------------------------------------
def _dense_corner_graph():
    from simrank_amd import ingest
    from tests.pydriver import SideSpec, reorder_specs
    from simrank_amd.engine import HipOps
    df = synth.powerlaw_directed(4096, 32, seed=21)
    _, csr = ingest.directed(df, False, "from", "to", "weight")
    specs, _ = reorder_specs([SideSpec(csr, csr.rowscale, 0.8)])
    ops = HipOps(0)
    blocks, cols, covered = ops.dense_stats(ops.graph(specs[0].csr))
    assert blocks >= 1 and cols >= 128 and covered > 0.1 * csr.nnz      # the path under test is live
    return df


@pytest.mark.parametrize("cls", ["SimRank", "SimRankPP"])
def test_fit_with_dense_sets_against_oracle(cls):
    df = _dense_corner_graph()
    est = getattr(SRA, cls)()
    got = est.fit(df, verbose=False)
    want = (O.fit_simrank if cls == "SimRank" else O.fit_simrank_pp)(df, verbose=False)
    assert list(got.index) == want["labels"]
    assert_close(got.values, want["S"])
    assert est.converged_at == want["k"]


@pytest.mark.parametrize("world", [1, 3])
def test_support_restricted_simrank_pp_is_bit_equal(world, monkeypatch):
    """SimRank++ keeps S inside supp(E) (SimRank.py:315-316, :361).  The restricted instantiation
    of leg 2 skips the gathers of 32-column segments whose evidence counts are all zero; it must
    give the bits of the unrestricted one (single rank: upper-triangle form; sharded: plain form),
    on a sparse-evidence graph (where the driver picks it by itself) and on one with dense sets."""
    import tests.pydriver as drv
    from simrank_amd.engine import HipOps
    # (fuse_sym = 0: the subject is gather3's restricted instantiation; where the dense sets dominate, the unrestricted leg 2 of
    # a single rank is otherwise the one-launch form of fused.hip — another order of the sums)
    HipOps(0).set_tuning(fuse_sym=0)
    try:
        for df in (synth.er_directed(4096, 0.001, seed=11), _dense_corner_graph()):
            out = {}
            for name, below in (("never", 0.0), ("always", 2.0), ("auto", drv.RESTRICT_BELOW)):
                monkeypatch.setattr(drv, "RESTRICT_BELOW", below)
                est = SRA.SimRankPP()
                S = est.fit(df, verbose=False, world=LocalWorld(world), mode="sparse")
                out[name] = (S.values, est.converged_at)
            assert np.array_equal(out["never"][0], out["always"][0])
            assert np.array_equal(out["never"][0], out["auto"][0])
            assert out["never"][1] == out["always"][1] == out["auto"][1]
    finally:
        HipOps(0).set_tuning(fuse_sym=-1)
    # the sparse-evidence graph is the case the driver restricts by itself
    _, csr = ingest.directed(synth.er_directed(4096, 0.001, seed=11), False, "from", "to", "weight")
    ops = HipOps(0)
    cnt = ops.matrix(csr.n_rows, csr.n_rows, np.uint8)
    ops.evidence_counts(ops.graph(csr), 0, cnt)
    frac = ops.evidence_live_fraction(cnt)
    c = ops.download(cnt)
    segs = c.reshape(csr.n_rows, -1, 32).any(axis=2)
    assert abs(frac - segs.mean()) < 1e-12 and frac < drv.RESTRICT_BELOW
    # the panel-blocked layout (what the single-rank solver holds) takes another path through the kernel:
    # a width off the 32-column grid, a few isolated counts, one of them in the last, partial segment
    rng = np.random.default_rng(0)
    c2 = np.zeros((777, 777), dtype=np.uint8)
    c2[rng.integers(0, 777, 300), rng.integers(0, 777, 300)] = 1
    c2[5, 776] = 3
    blk = ops.matrix(777, 777, np.uint8, blocked=True)
    ops.upload(blk, c2)
    padded = np.zeros((777, 25 * 32), dtype=np.uint8)
    padded[:, :777] = c2
    assert abs(ops.evidence_live_fraction(blk) - padded.reshape(777, 25, 32).any(axis=2).mean()) < 1e-12


def test_fp16_dense_blocks_are_an_explicit_reduced_precision_choice():
    """fit(dense_precision="fp16"): BASELINE.json config 5's reduced-precision dense leg — the
    operand of the matrix-core part rounded to one fp16 term.  Close to the oracle (fp16 has 11
    significant bits; sums average the rounding), but NOT within the 1e-5 bar, and never the
    default."""
    df = _dense_corner_graph()
    want = O.fit_simrank(df, iterations=6, eps=0, verbose=False)["S"]
    exact = SRA.SimRank().fit(df, iterations=6, eps=0, verbose=False)
    low = SRA.SimRank().fit(df, iterations=6, eps=0, verbose=False, dense_precision="fp16")
    assert_close(exact.values, want)
    pos = want > 0
    rel = np.abs(low.values[pos] - want[pos]) / want[pos]
    assert 1e-6 < rel.max() < 5e-3, rel.max()
    again = SRA.SimRank().fit(df, iterations=6, eps=0, verbose=False)          # the knob does not stick
    assert np.array_equal(again.values, exact.values)
    with pytest.raises(ValueError, match="dense_precision"):
        SRA.SimRank().fit(df, verbose=False, dense_precision="fp8")


def test_dense_sets_shards_equal_one_shard_bit_for_bit():
    """The dense part works per column block with a fixed summation order: P column shards give
    the same bits as one shard (upper-triangle shortcut off for the comparison)."""
    from simrank_amd.engine import HipOps
    df = _dense_corner_graph()
    knob = HipOps(0)
    # (the dense_tiles + gather path on both sides: one-launch leg off on the single rank AND on the shards)
    knob.set_tuning(triangle=0, fuse=0)
    try:
        one = SRA.SimRank().fit(df, iterations=4, eps=0, verbose=False, mode="sparse")
        for world in (2, 4):
            many = SRA.SimRank().fit(df, iterations=4, eps=0, verbose=False, mode="sparse",
                                     world=LocalWorld(world, symmetric_shards=False))
            assert np.array_equal(one.values, many.values)
    finally:
        knob.set_tuning(triangle=1, fuse=1)
    knob.set_tuning(dense_min=0)
    try:
        plain = SRA.SimRank().fit(df, iterations=4, eps=0, verbose=False, mode="sparse")
    finally:
        knob.set_tuning(dense_min=4)
    np.testing.assert_allclose(one.values, plain.values, rtol=2e-6, atol=1e-30)


# ---------------------------------------------------------------------------------------
# the C-level plan (simrank_plan_*): the reference loop behind create / run / result
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["SimRank_toy5", "SimRank_er128", "SimRank_quirky_weighted", "SimRank_bts300",
                                  "SimRankPP_toy5", "SimRankPP_er128", "SimRankPP_bts300"])
def test_plan_api_reproduces_the_golden_vectors(ops, name):
    """create -> run -> result against the vectors generated from the reference: S to 1e-5, the
    "Converged at iteration k" index exactly, iterations = 0 and eps >= 1 included."""
    from simrank_amd.engine import Plan
    g = Golden(name)
    G = g.out["G"]
    n = len(G)
    rows, cols = np.nonzero(G)
    rowptr = np.zeros(n + 1, np.int32)
    np.cumsum(np.bincount(rows, minlength=n), out=rowptr[1:])
    scale = np.zeros(n)
    scale[rows] = G[rows, cols]
    csr = ingest.CSR(n, n, rowptr, cols.astype(np.int32), scale)
    pp = name.startswith("SimRankPP")
    plan = Plan(ops, csr, coef=g.kwargs.get("C", 0.8), evidence=pp)
    done, conv = plan.run(g.kwargs.get("iterations", 100), g.kwargs.get("eps", 1e-4))
    assert (conv if conv is not None else -1) == (g.k if g.k is not None else -1)
    assert_close(plan.result(), g.out["S"])
    # the same plan again, step by step with the exact count, against the one-call loop
    plan.reset()
    eps = g.kwargs.get("eps", 1e-4)
    counts = [plan.step(eps, exact_count=True) for _ in range(done)]
    if conv is not None and done:
        assert counts[-1] == 0 and all(c > 0 for c in counts[:-1])     # the passing test is the first zero count
    assert_close(plan.result(), g.out["S"])
    assert plan.run(0, 1e-4) == (0, None)
    np.testing.assert_array_equal(plan.result(), np.eye(n))
    assert plan.run(5, 1.0) == (0, 0)
    plan.free()


def test_plan_topk_in_the_callers_ids(ops):
    """simrank_plan_topk: the k most similar nodes per node from the plan's own matrices (f32 and fp16-held),
    ids and tie order the caller's, against a sort of the dense result."""
    from simrank_amd.engine import Plan
    df = synth.powerlaw_directed(600, 7, seed=9)
    _, csr = ingest.directed(df, False, "from", "to", "weight")
    for storage in ("f32", "fp16"):
        plan = Plan(ops, csr, coef=0.8, evidence=True, storage=storage)
        plan.run(6, 1e-30)
        S = plan.result()
        idx, val = plan.topk(7)
        plan.free()
        n = csr.n_rows
        for a in range(n):
            cand = np.array([c for c in range(n) if c != a])
            order = cand[np.lexsort((cand, -S[a, cand]))][:7]
            assert list(idx[a]) == list(order)
            np.testing.assert_array_equal(val[a].astype(np.float64), S[a, order])


def test_plan_api_with_a_prior_and_in_the_callers_order(ops):
    """AprioriSimRank's loop (SimRank.py:443-454) through the plan: symmetric prior, nodes re-ordered inside
    and handed back in the caller's order; an asymmetric prior (asymmetric iterates: leg 2 stored transposed, the
    epilogue as a pass of its own) against the oracle too, step by step with its exact counts; refused only for
    fp16-held matrices."""
    from simrank_amd.engine import Plan
    from simrank_amd._lib import SimRankHipError
    df = synth.powerlaw_directed(700, 9, seed=5)
    _, csr = ingest.directed(df, False, "from", "to", "weight")
    rng = np.random.default_rng(2)
    A = rng.random((csr.n_rows, csr.n_rows))
    A = (A + A.T) / 2
    G = csr.dense()
    want_S, want_k = O.iterate_directed(G, C=0.7, iterations=100, eps=1e-4, E=O.evidence(G),
                                        apriori=A.astype(np.float32).astype(np.float64), lbd=0.3)
    plan = Plan(ops, csr, coef=0.7, evidence=True, apriori=A, lbd=0.3)
    done, conv = plan.run(100, 1e-4)
    assert conv == want_k
    assert_close(plan.result(), want_S)
    plan.free()
    A[3, 5] += 0.5
    A[10:40, 200:260] *= 0.25
    A32 = A.astype(np.float32).astype(np.float64)
    want_S, want_k = O.iterate_directed(G, C=0.7, iterations=100, eps=1e-4, E=O.evidence(G), apriori=A32, lbd=0.3)
    assert not np.array_equal(want_S, want_S.T)
    plan = Plan(ops, csr, coef=0.7, evidence=True, apriori=A, lbd=0.3)
    done, conv = plan.run(100, 1e-4)
    assert conv == want_k
    got = plan.result()
    assert_close(got, want_S)
    idx, val = plan.topk(5)
    for a in (0, 3, 11, 699):
        cand = np.array([c for c in range(csr.n_rows) if c != a])
        np.testing.assert_array_equal(val[a].astype(np.float64), got[a, cand[np.lexsort((cand, -got[a, cand]))][:5]])
    # step by step: the exact number of moved elements of every update (the epilogue pass counts every element)
    plan.reset()
    S = np.eye(csr.n_rows)
    for _ in range(3):
        new = O.update(G, S, 0.7, E=O.evidence(G), apriori=A32, lbd=0.3)
        moved = int((np.abs(new - S) > 1e-4).sum())
        got_moved = plan.step(1e-4, exact_count=True)
        assert abs(got_moved - moved) <= max(3, moved // 2000)       # (f32 values at the threshold)
        S = new
    plan.free()
    with pytest.raises(SimRankHipError, match="symmetric"):
        Plan(ops, csr, apriori=A, lbd=0.3, storage="fp16")


# ---------------------------------------------------------------------------------------
# the bipartite loop behind the C ABI (simrank_biplan_*)
# ---------------------------------------------------------------------------------------
def _csr_and_scales(G12, G21):
    """Pattern of the group-1 adjacency and the per-row values of both normalised adjacencies."""
    n1, n2 = G12.shape
    rows, cols = np.nonzero(G12)
    rowptr = np.zeros(n1 + 1, np.int32)
    np.cumsum(np.bincount(rows, minlength=n1), out=rowptr[1:])
    rs1 = np.zeros(n1)
    rs1[rows] = G12[rows, cols]
    r2, c2 = np.nonzero(G21)
    rs2 = np.zeros(n2)
    rs2[r2] = G21[r2, c2]
    return ingest.CSR(n1, n2, rowptr, cols.astype(np.int32), rs1), rs1, rs2


def _biplan_inputs(g):
    """CSR of the group-1 pattern and both groups' row values (normalisation x spread for the ++ classes,
    SimRank.py:396-397) from a golden case's own edge list, as a reference-side binding would pass them."""
    kw = g.kwargs
    _, _, _, _, g12, g21 = ingest.bipartite(g.frame, kw.get("weighted", False), kw.get("node_group1_column", "user"),
                                            kw.get("node_group2_column", "item"), kw.get("weight_column", "weight"))
    pp = g.cls != "BipartiteSimRank"
    w1 = ingest.spread(g12) * g12.rowscale if pp else g12.rowscale
    w2 = ingest.spread(g21) * g21.rowscale if pp else g21.rowscale
    return g12, w1, w2, pp


@pytest.mark.parametrize("name", golden_names("BipartiteSimRank", "BipartiteSimRankPP", "BipartitleAprioriSimRank"))
def test_biplan_reproduces_the_golden_vectors(ops, name):
    """create -> run -> result for EVERY bipartite vector generated from the reference — BipartiteSimRank,
    BipartiteSimRankPP and BipartitleAprioriSimRank with options.strict_reference = 1 (Evidence_N1 on both
    updates, SimRank.py:420-423, :488-491): S1 and S2 to 1e-5, the "Converged at iteration k" index exactly,
    iterations = 0 and eps >= 1 included; NumPy's broadcast error for n1 != n2 where the reference raises it
    (first group-2 update), the 1 x 1 broadcast; step by step the same.  Asymmetric priors (`*_asym`: both iterates
    asymmetric, un-fused epilogue inside the plan) included."""
    from simrank_amd._lib import SimRankHipError
    from simrank_amd.engine import BiPlan
    g = Golden(name)
    csr, w1, w2, pp = _biplan_inputs(g)
    kw = g.kwargs
    pri = dict(apriori1=g.args[0], apriori2=g.args[1], lbd1=kw.get("lbd1", 0.5), lbd2=kw.get("lbd2", 0.5)) if g.args else {}
    make = lambda: BiPlan(ops, csr, w1, w2, c1=kw.get("C1", 0.8), c2=kw.get("C2", 0.8), evidence=pp,
                          strict_reference=True, **pri)
    plan = make()
    its, eps = kw.get("iterations", 100), kw.get("eps", 1e-4)
    if g.raises:
        with pytest.raises(SimRankHipError) as e:
            plan.run(its, eps)
        assert g.meta["message"] in str(e.value)
        with pytest.raises(SimRankHipError, match="broadcast"):
            plan.step(eps)
        assert plan.run(0, eps) == (0, None) and plan.run(5, 1.0) == (0, 0)      # (as *_iter0 / *_eps1)
        plan.free()
        return
    done, conv = plan.run(its, eps)
    assert (conv if conv is not None else -1) == (g.k if g.k is not None else -1)
    s1, s2 = plan.result()
    assert_close(s1, g.out["S1"])
    assert_close(s2, g.out["S2"])
    plan.reset()
    counts = [plan.step(eps, exact_count=True) for _ in range(done)]
    if conv is not None and done:
        assert counts[-1] == (0, 0) and all(sum(c) > 0 for c in counts[:-1])
    t1, t2 = plan.result()
    assert_close(t1, g.out["S1"])
    assert_close(t2, g.out["S2"])
    plan.free()


def test_biplan_with_evidence_and_priors_against_the_oracle(ops):
    """BipartiteSimRankPP in its corrected form (E2 from the group-2 pattern, strict_reference = False) and
    BipartitleAprioriSimRank with symmetric priors and with one that is not, n1 != n2, through the C-level loop."""
    from simrank_amd.engine import BiPlan
    from simrank_amd._lib import SimRankHipError
    df = bipartite_random(700, 300, 0.03, seed=21)
    want = O.fit_bipartite_pp(df, verbose=False, strict_reference=False, C1=0.7, C2=0.85)
    csr, rs1, rs2 = _csr_and_scales(want["G12"], want["G21"])
    # SimRank++ row scales: spread x normalisation (W1 / W2 of the reference, one value per row)
    w1 = np.array([want["W1"][a][want["W1"][a] != 0][0] if (want["W1"][a] != 0).any() else 0.0 for a in range(700)])
    w2 = np.array([want["W2"][a][want["W2"][a] != 0][0] if (want["W2"][a] != 0).any() else 0.0 for a in range(300)])
    plan = BiPlan(ops, csr, w1, w2, c1=0.7, c2=0.85, evidence=True)
    done, conv = plan.run(100, 1e-4)
    s1, s2 = plan.result()
    assert conv == want["k"]
    assert_close(s1, want["S1"])
    assert_close(s2, want["S2"])
    plan.free()
    rng = np.random.default_rng(3)
    a1, a2 = rng.random((700, 700)), rng.random((300, 300))
    a1, a2 = (a1 + a1.T) / 2, (a2 + a2.T) / 2
    f32 = lambda a: a.astype(np.float32).astype(np.float64)
    want = O.fit_bipartite_pp(df, verbose=False, strict_reference=False, apriori1=f32(a1), apriori2=f32(a2),
                              lbd1=0.2, lbd2=0.4)
    plan = BiPlan(ops, csr, w1, w2, evidence=True, apriori1=a1, apriori2=a2, lbd1=0.2, lbd2=0.4)
    done, conv = plan.run(100, 1e-4)
    s1, s2 = plan.result()
    assert conv == want["k"]
    assert_close(s1, want["S1"])
    assert_close(s2, want["S2"])
    plan.free()
    # ONE prior that is not symmetric makes both iterates asymmetric (S1 feeds S2 and back): both updates un-fused
    a2[1, 2] += 0.25
    a2[40:90, 100:180] *= 0.5
    want = O.fit_bipartite_pp(df, verbose=False, strict_reference=False, apriori1=f32(a1), apriori2=f32(a2),
                              lbd1=0.2, lbd2=0.4)
    assert not np.array_equal(want["S1"], want["S1"].T) and not np.array_equal(want["S2"], want["S2"].T)
    plan = BiPlan(ops, csr, w1, w2, evidence=True, apriori1=a1, apriori2=a2, lbd1=0.2, lbd2=0.4)
    done, conv = plan.run(100, 1e-4)
    s1, s2 = plan.result()
    assert conv == want["k"]
    assert_close(s1, want["S1"])
    assert_close(s2, want["S2"])
    plan.free()
