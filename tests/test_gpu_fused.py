"""Leg 1 as one launch (csrc/fused.hip) on a real MI355X, through the C ABI: the matrix-core part of a
128-row block and its gathered remainder against float64 NumPy (1e-5, north_star) and against the
two-launch path of round 2 (rounding only), over the shapes that exercise every branch: blocks with and
without a dense set, sets of 1 .. many 64-column groups, rows and columns off the 32/128 grids,
rectangular (bipartite) patterns, rows sent whole to the matrix cores, 32-bit ids.
First `.dot` of SimRank.py:139 / :298 / :301 / :361 / :420 / :423."""
import contextlib

import numpy as np
import pytest

from simrank_amd.ingest import CSR
from tests.test_gpu_kernels import corner_csr, dense64, random_csr

pytestmark = pytest.mark.gpu
RTOL = 1e-5


@pytest.fixture(scope="module")
def ops():
    from simrank_amd.engine import HipOps
    o = HipOps(0)
    o.set_tuning(fuse_steps=1, fuse_min=2)   # small test graphs: a dense set however few steps it makes
    yield o
    o.set_tuning(fuse=1, fuse_min=0, fuse_steps=-1, fuse_unit=48, fuse_group=3, fuse_shards=1, fuse_rows=8192, fuse_order=0,
                 fuse_sym=-1)


@contextlib.contextmanager
def knobs(ops, **kw):
    ops.set_tuning(**kw)
    try:
        yield
    finally:
        ops.set_tuning(fuse=1, fuse_min=2, fuse_steps=1, fuse_unit=48, fuse_group=3, fuse_shards=1, fuse_rows=8192, fuse_order=0,
                       fuse_sym=-1)


def put_blocked(ops, a, dtype=np.float32):
    m = ops.matrix(a.shape[0], a.shape[1], dtype, blocked=True)
    ops.upload(m, a.astype(dtype))
    return m


def leg1(ops, g, X, M):
    yt = ops.matrix(X.shape[1], M, blocked=True)
    ops.spmm(g, put_blocked(ops, X), yt, transpose_out=True)
    return ops.download(yt)


@pytest.mark.parametrize("shape", [(520, 400, 333), (384, 384, 384), (1000, 300, 70), (130, 200, 2),
                                   (128, 128, 32), (129, 77, 33), (64, 1000, 96), (2100, 2100, 160)])
@pytest.mark.parametrize("fuse_min", [2, 4])
def test_fused_leg1_matches_numpy_and_the_two_launch_path(ops, shape, fuse_min):
    M, K, L = shape
    csr = corner_csr(M, K, seed=M + L, hubs=min(K, 150))
    X = (np.random.default_rng(5).random((K, L)) ** 3).astype(np.float32)
    want = (dense64(csr) @ X.astype(np.float64)).T
    with knobs(ops, fuse_min=fuse_min):
        g = ops.graph(csr)
        steps, cov, rem = ops.fused_stats(g)
        assert cov + rem == csr.nnz and (steps > 0) == (cov > 0)
        got = leg1(ops, g, X, M)
        assert np.array_equal(got, leg1(ops, g, X, M))          # reproducible
    np.testing.assert_allclose(got, want, rtol=RTOL, atol=1e-30)
    with knobs(ops, fuse=0):
        g0 = ops.graph(csr)
        assert ops.fused_stats(g0) == (0, 0, csr.nnz)
        old = leg1(ops, g0, X, M)
    np.testing.assert_allclose(got, old, rtol=2e-6, atol=1e-30)


@pytest.mark.parametrize("order", [1, 2, 3])
def test_fused_launch_order_with_split_blocks(ops, order):
    """fuse_order > 0 puts other units between the units of a split block: every unit still finds its own partial-sum
    slot and its block's ticket (round 4: the slots were taken from the neighbour in the list — a fault on the first
    graph large enough to separate them; tools/host/host_fuzz.cpp now builds such plans on the host)."""
    M, K, L = 2600, 3000, 96
    csr = corner_csr(M, K, seed=21 + order, hubs=900, p_hub=0.12, avg=10)
    X = (np.random.default_rng(8).random((K, L)) ** 2).astype(np.float32)
    want = (dense64(csr) @ X.astype(np.float64)).T
    with knobs(ops, fuse_min=3, fuse_steps=2, fuse_unit=4, fuse_rows=400, fuse_group=3, fuse_order=order):
        g = ops.graph(csr)
        got = leg1(ops, g, X, M)
        assert np.array_equal(got, leg1(ops, g, X, M))
    np.testing.assert_allclose(got, want, rtol=RTOL, atol=1e-30)
    with knobs(ops, fuse_min=3, fuse_steps=2, fuse_unit=4, fuse_rows=400, fuse_group=3, fuse_order=0):
        g = ops.graph(csr)
        assert np.array_equal(got, leg1(ops, g, X, M))          # the order of the launches does not touch the sums


def test_fused_split_is_exact(ops):
    """One entry per row, every column shared by two rows of a block: everything goes through the bf16
    hi + mid + lo split and must come back bit for bit."""
    M, K, L = 256, 64, 128
    rows = [np.array([a % 64], dtype=np.int32) for a in range(M)]
    csr = CSR(M, K, np.arange(M + 1, dtype=np.int32), np.concatenate(rows), np.ones(M))
    rng = np.random.default_rng(0)
    X = (rng.standard_normal((K, L)) * np.exp(rng.uniform(-60, 60, size=(K, L)))).astype(np.float32)
    X[0, :8] = [0.0, 1.0, -1.0, 2.0 ** -100, 1 + 2.0 ** -23, 16777215.0, -3.0000002, 1e-30]
    g = ops.graph(csr)
    assert ops.fused_stats(g) == (2 * 4, M, 0)
    assert np.array_equal(leg1(ops, g, X, M), X[np.arange(M) % 64].T)


def test_fused_without_any_dense_set(ops):
    """Rows that share nothing: every block is gather-only (the LDS tile starts undefined)."""
    M = K = 500
    rows = [np.array(sorted({(7 * a + 3) % K, (11 * a + 5) % K}), dtype=np.int32) for a in range(M)]
    rowptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int32)
    csr = CSR(M, K, rowptr, np.concatenate(rows), np.random.default_rng(1).random(M) + 0.5)
    X = np.random.default_rng(2).random((K, 200)).astype(np.float32)
    with knobs(ops, fuse_min=100):
        g = ops.graph(csr)
        assert ops.fused_stats(g)[0] == 0
        got = leg1(ops, g, X, M)
    np.testing.assert_allclose(got, (dense64(csr) @ X.astype(np.float64)).T, rtol=RTOL, atol=1e-30)


def test_fused_long_rows_go_to_the_matrix_cores_whole(ops):
    """A star: one row references every column and shares them with nobody in its block; its remainder
    would be K entries long, so all its columns join the block's dense set."""
    M = K = 1500
    rng = np.random.default_rng(4)
    rows = [np.sort(rng.choice(K, size=3, replace=False)).astype(np.int32) for _ in range(M)]
    rows[700] = np.arange(K, dtype=np.int32)
    rows[701] = np.arange(0, K, 3, dtype=np.int32)
    rowptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int32)
    csr = CSR(M, K, rowptr, np.concatenate(rows), rng.random(M) + 0.1)
    X = rng.random((K, 257)).astype(np.float32)
    g = ops.graph(csr)
    steps, cov, rem = ops.fused_stats(g)
    assert cov >= K + K // 3 and steps >= K // 16
    np.testing.assert_allclose(leg1(ops, g, X, M), (dense64(csr) @ X.astype(np.float64)).T, rtol=RTOL, atol=1e-30)


def test_fused_ids_beyond_16_bits(ops):
    """More than 65536 operand rows: 32-bit ids in the dense sets and the remainder."""
    M, K, L = 300, 70000, 64
    rng = np.random.default_rng(9)
    hubs = rng.choice(K, size=90, replace=False)
    rows = []
    for a in range(M):
        c = set(rng.choice(K, size=6, replace=False).tolist())
        c |= set(hubs[rng.random(90) < 0.3].tolist())
        rows.append(np.array(sorted(c), dtype=np.int32))
    rowptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int32)
    csr = CSR(M, K, rowptr, np.concatenate(rows), rng.random(M) + 0.1)
    X = rng.random((K, L)).astype(np.float32)
    g = ops.graph(csr)
    assert ops.fused_stats(g)[0] > 0
    np.testing.assert_allclose(leg1(ops, g, X, M), (dense64(csr) @ X.astype(np.float64)).T, rtol=RTOL, atol=1e-30)


@pytest.mark.parametrize("K", [65535, 65536, 65537])
def test_fused_ids_at_the_16_bit_edge(ops, K):
    """65535 operand rows stream 16-bit ids (0xFFFF marks an empty slot and is no column's id), 65536 and 65537 take 32-bit
    ids (the last column's id would be the marker); the last columns are referenced by rows of several blocks, with and
    without other shared columns."""
    M, L = 420, 64
    rng = np.random.default_rng(K)
    hubs = rng.choice(K - 8, size=60, replace=False)
    rows = []
    for a in range(M):
        c = set(rng.choice(K - 8, size=5, replace=False).tolist())
        if a < 300:
            c |= set(hubs[rng.random(60) < 0.3].tolist())
        if a % 7 == 3:
            c |= {K - 1}
        if a % 11 == 2:
            c |= {K - 2, K - 1}
        rows.append(np.array(sorted(c), dtype=np.int32))
    rowptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int32)
    csr = CSR(M, K, rowptr, np.concatenate(rows), rng.random(M) + 0.1)
    X = rng.random((K, L)).astype(np.float32)
    X[K - 1] = 7.0 + rng.random(L).astype(np.float32)          # (a marker misread as this row would show)
    want = (dense64(csr) @ X.astype(np.float64)).T
    for fuse_min in (2, 100):                                   # with shared columns on the matrix cores, and without any
        with knobs(ops, fuse_min=fuse_min):
            g = ops.graph(csr)
            np.testing.assert_allclose(leg1(ops, g, X, M), want, rtol=RTOL, atol=1e-30)


@pytest.mark.parametrize("seed", range(12))
def test_fused_randomized(ops, seed):
    rng = np.random.default_rng(3000 + seed)
    M = int(rng.integers(1, 1200))
    K = int(rng.integers(1, 1200))
    L = int(rng.integers(1, 500))
    if seed % 3 == 0:
        csr = random_csr(M, K, int(rng.integers(1, 30)), seed, heavy={0: min(K, 400)} if M > 1 else ())
    else:
        csr = corner_csr(M, K, seed=seed, hubs=int(rng.integers(1, max(2, K))), p_hub=float(rng.uniform(0.05, 0.9)),
                         avg=int(rng.integers(1, 12)))
    X = (rng.random((K, L)) ** 2).astype(np.float32)
    with knobs(ops, fuse_min=int(rng.integers(2, 6)), fuse_steps=int(rng.choice([0, 1, 3, 8])),
               fuse_unit=int(rng.choice([4, 8, 32, 1 << 20])), fuse_group=int(rng.integers(1, 5)),
               fuse_rows=int(rng.choice([64, 300, 2000, 8192])), fuse_order=int(rng.choice([0, 0, 1, 2, 3]))):
        g = ops.graph(csr)
        got = leg1(ops, g, X, M)
        steps, cov, rem = ops.fused_stats(g)
        assert cov + rem == csr.nnz
    np.testing.assert_allclose(got, (dense64(csr) @ X.astype(np.float64)).T, rtol=RTOL, atol=1e-30)


@pytest.mark.parametrize("rows", [64, 500, 8192])
def test_fused_gather_units(ops, rows):
    """Round 4: a block whose remainder exceeds fuse_rows entries is cut into gather units (every n-th row of its
    descending remainder order each) beside the matrix-core units of its set; all publish raw sums, the last arriver
    adds them in unit order and scales the rows: same values whatever the cut, same bits launch after launch."""
    M, K, L = 520, 3000, 130
    csr = corner_csr(M, K, seed=11, hubs=1500, p_hub=0.3, avg=30)
    X = np.random.default_rng(12).random((K, L)).astype(np.float32)
    want = (dense64(csr) @ X.astype(np.float64)).T
    with knobs(ops, fuse_rows=rows, fuse_unit=8, fuse_min=4):
        g = ops.graph(csr)
        first = leg1(ops, g, X, M)
        for _ in range(3):
            assert np.array_equal(first, leg1(ops, g, X, M))
    np.testing.assert_allclose(first, want, rtol=RTOL, atol=1e-30)
    with knobs(ops, fuse_unit=1 << 20, fuse_min=4):
        whole = leg1(ops, ops.graph(csr), X, M)
    np.testing.assert_allclose(first, whole, rtol=2e-6, atol=1e-30)
    # without any dense set: gather units only
    with knobs(ops, fuse_rows=rows, fuse_min=100):
        got = leg1(ops, ops.graph(csr), X, M)
    np.testing.assert_allclose(got, want, rtol=RTOL, atol=1e-30)


@pytest.mark.parametrize("unit", [4, 6, 32])
def test_fused_blocks_cut_into_units(ops, unit):
    """A dense set of many 64-column groups is cut into units (workgroups) whose partial sums meet in
    memory; whichever unit arrives last adds them in unit order: same bits whatever the cut, launch after
    launch (the tickets reset themselves), also for a second, narrower operand."""
    M, K, L = 300, 6000, 200
    csr = corner_csr(M, K, seed=3, hubs=3000, p_hub=0.6)
    X = np.random.default_rng(8).random((K, L)).astype(np.float32)
    want = (dense64(csr) @ X.astype(np.float64)).T
    with knobs(ops, fuse_unit=unit):
        g = ops.graph(csr)
        assert ops.fused_stats(g)[0] > 16 * 4
        first = leg1(ops, g, X, M)
        for _ in range(3):
            assert np.array_equal(first, leg1(ops, g, X, M))
        narrow = leg1(ops, g, X[:, :70], M)
        assert np.array_equal(narrow, first[:70])
        assert np.array_equal(first, leg1(ops, g, X, M))
    np.testing.assert_allclose(first, want, rtol=RTOL, atol=1e-30)
    with knobs(ops, fuse_unit=1 << 20):
        whole = leg1(ops, ops.graph(csr), X, M)
    np.testing.assert_allclose(first, whole, rtol=2e-6, atol=1e-30)


@pytest.mark.parametrize("shape,tb,pad", [((1024, 1024, 256), 256, 4), ((1024, 1024, 200), 128, 0), ((640, 900, 96), 0, 0),
                                          ((2048, 2048, 512), 256, 8), ((1000, 1000, 64), 384, 4)])
def test_fused_leg1_on_a_sharded_ranks_operand(ops, shape, tb, pad):
    """Round 4: the one-launch leg on what one rank of a column-sharded update holds — a ROW-MAJOR column block of
    S (here columns [c0, c0 + L) of a wider matrix) — with the transposed result in the chunks of the all-to-all
    (simrank_spmm's t_block layout): against float64 NumPy, against the gather kernels the shards ran until round
    3 (rounding only: fuse_shards = 0), and bit for bit against the same leg on a panel-blocked operand."""
    M, K, L = shape
    csr = corner_csr(M, K, seed=M + L, hubs=min(K, 200), p_hub=0.4)
    rng = np.random.default_rng(6)
    wide = (rng.random((K, L + 64)) ** 3).astype(np.float32)
    c0 = 32
    X = wide[:, c0:c0 + L]
    want = (dense64(csr) @ X.astype(np.float64)).T                 # L x M
    xw = ops.matrix(K, L + 64)
    ops.upload(xw, wide)
    tbe = tb if tb else M
    nblk = -(-M // tbe)
    size = nblk * L * (tbe + pad)

    def run(**kw):
        with knobs(ops, **kw):
            g = ops.graph(csr)
            if tb == 0:                   # one block, pitched rows: Y^T[c * ld + a]
                y = ops.matrix(L, M)
                ops.spmm(g, xw, y, n_cols=L, transpose_out=True, x_col0=c0)
                return ops.download(y)
            y = ops.matrix(1, size, ld=size)
            ops.spmm(g, xw, y, n_cols=L, transpose_out=True, t_block=tb, t_pad=pad, x_col0=c0)
            flat = ops.download(y).ravel()
        out = np.zeros((L, M), dtype=np.float32)
        for h in range(nblk):
            lo, hi = h * tbe, min(M, (h + 1) * tbe)
            w = hi - lo + pad
            out[:, lo:hi] = flat[h * L * (tbe + pad): h * L * (tbe + pad) + L * w].reshape(L, w)[:, :hi - lo]
        return out

    got = run()
    np.testing.assert_allclose(got, want, rtol=RTOL, atol=1e-30)
    old = run(fuse_shards=0)
    np.testing.assert_allclose(got, old, rtol=2e-6, atol=1e-30)
    if tb == 0 or tb % 128 == 0:
        g = ops.graph(csr)
        assert np.array_equal(got, leg1(ops, g, X, M))               # the same plan on a panel-blocked copy: same bits


# ---------------------------------------------------------------------------------------------------------------------
# leg 2 as one launch (round 6: fused_trans_kernel<…, SYM>): matrix cores + gathered remainder + epilogue + both stores of a
# tile; second `.dot` of SimRank.py:139 / :298 / :361 / :420, the element-wise lines :140, :316, :362, :453, the count of :74
# ---------------------------------------------------------------------------------------------------------------------
def _leg2_case(ops, n, variant, seed, **graph_knobs):
    """-> (result, exact count, count_any count, float64 reference, previous iterate): a symmetric product (Tt = (W S)^T of a
    symmetric S), symmetric counts / prior, through the epilogue with `symmetric=True` on panel-blocked matrices."""
    csr = corner_csr(n, n, seed=seed, hubs=min(n, 150))
    rng = np.random.default_rng(seed)
    W = dense64(csr)
    S = rng.random((n, n)) ** 3
    S = ((S + S.T) / 2).astype(np.float32)
    np.fill_diagonal(S, 1)
    Tt = (W @ S.astype(np.float64)).T.astype(np.float32)           # as leg 1 stores it (rounded to f32)
    cnt = prior = None
    lbd = 0.0
    if variant in ("evidence", "all"):
        c = rng.integers(0, 6, size=(n, n))
        cnt = (np.triu(c) + np.triu(c, 1).T).astype(np.uint8)
        cnt[0, :4] = cnt[:4, 0] = [0, 1, 30, 255]
    if variant in ("prior", "all"):
        pr = rng.random((n, n)).astype(np.float32)
        prior = ((pr + pr.T) / 2).astype(np.float32)
        lbd = 0.3
    want = 0.8 * (W @ Tt.astype(np.float64))
    if cnt is not None:
        want = want * (1 - 0.5 ** cnt.astype(np.float64))
    if prior is not None:
        want = (1 - np.float32(lbd)) * want + np.float32(lbd) * prior.astype(np.float64)
    np.fill_diagonal(want, 1.0)
    prev = S
    with knobs(ops, **graph_knobs):
        g = ops.graph(csr)
        steps, cov, rem = ops.fused_stats(g)
        y = ops.matrix(n, n, blocked=True)
        ep = dict(coef=0.8, previous=put_blocked(ops, prev), eps=0.05, diag_col0=0, symmetric=True, lbd=lbd,
                  evidence=None if cnt is None else put_blocked(ops, cnt, np.uint8),
                  apriori=None if prior is None else put_blocked(ops, prior))
        ops.spmm(g, put_blocked(ops, Tt), y, epilogue=ep)
        got, exact = ops.download(y), ops.read_changed()
        y2 = ops.matrix(n, n, blocked=True)
        ops.spmm(g, put_blocked(ops, Tt), y2, epilogue=dict(ep, count_any=True))
        assert np.array_equal(ops.download(y2), got)
        some = ops.read_changed()
    return got, exact, some, want, prev, (steps, cov, rem)


@pytest.mark.parametrize("n", [64, 129, 200, 520, 1000, 1031, 2100])
@pytest.mark.parametrize("variant", ["plain", "evidence", "prior", "all"])
def test_fused_leg2_epilogue_triangle_and_mirror(ops, n, variant):
    """The one-launch leg 2 (`fuse_sym=1`) against float64 and against the two-launch leg (`fuse_sym=0`: dense_tiles +
    gather3<kSym>): values within 1e-5 / to rounding, the result exactly symmetric outside the diagonal tiles and with a unit
    diagonal, the exact count = the count over the stored matrix, count_any non-zero exactly when the count is."""
    got, exact, some, want, prev, stats = _leg2_case(ops, n, variant, seed=n, fuse_sym=1)
    assert stats[1] > 0                                            # (the graph has dense sets: the matrix-core phase runs)
    np.testing.assert_allclose(got, want, rtol=RTOL, atol=1e-30)
    assert np.array_equal(np.diag(got), np.ones(n, dtype=np.float32))
    if n > 64:
        assert np.array_equal(got[:32, 32:], got[32:, :32].T)       # mirrored tiles: same bits
    np.testing.assert_allclose(got, got.T, rtol=2e-6, atol=1e-30)
    assert exact == int((np.abs(got.astype(np.float64) - prev.astype(np.float64)) > 0.05).sum())
    assert (some > 0) == (exact > 0) and 0 < some <= exact
    old, exact0, some0, _, _, _ = _leg2_case(ops, n, variant, seed=n, fuse_sym=0)
    np.testing.assert_allclose(got, old, rtol=2e-6, atol=1e-30)
    assert n < 500 or not np.array_equal(got, old)                  # (another order of the sums: it WAS the other launch)
    assert abs(exact - exact0) <= 4                                 # (elements within rounding of eps may fall either way)


@pytest.mark.parametrize("split", [dict(fuse_unit=4, fuse_rows=400), dict(fuse_group=1), dict(fuse_group=4, fuse_min=4)])
def test_fused_leg2_with_split_blocks_and_grouped_units(ops, split):
    """Blocks cut into matrix-core units and gather units (the last arriver combines, scales, runs the epilogue and stores),
    and units of several blocks without a set: the same values, the exact count."""
    got, exact, some, want, prev, stats = _leg2_case(ops, 2600, "all", seed=77, **{**dict(fuse_sym=1, fuse_min=3, fuse_steps=2), **split})
    np.testing.assert_allclose(got, want, rtol=RTOL, atol=1e-30)
    assert np.array_equal(got[:128, 128:], got[128:, :128].T)
    assert exact == int((np.abs(got.astype(np.float64) - prev.astype(np.float64)) > 0.05).sum()) and 0 < some <= exact


def test_fused_leg2_runs_the_bipartite_plan_where_the_dense_sets_dominate(ops):
    """`fuse_sym = -1` (the default): the two-matrix plan of a MovieLens-shaped graph takes the one-launch leg 2 (its dense sets
    hold > half of the entries), a sparse one keeps gather3<kSym>; both against the oracle."""
    from oracle import simrank_oracle as O
    from simrank_amd import ingest
    from simrank_amd.engine import BiPlan, HipOps
    from tests.graphs import bipartite_random
    fresh = HipOps(0)                                                # (the library's default knobs)
    fresh.set_tuning(fuse=1, fuse_min=0, fuse_pays=-1, fuse_steps=-1, fuse_unit=48, fuse_group=3, fuse_shards=1, fuse_rows=8192, fuse_order=0,
                     fuse_sym=-1)
    for n1, n2, dens in ((1400, 900, 0.12), (900, 700, 0.004)):
        df = bipartite_random(n1, n2, dens, seed=n1)
        _, _, _, _, g12, g21 = ingest.bipartite(df, False, "user", "item", "weight")
        want = O.fit_bipartite_pp(df, verbose=False, strict_reference=False)
        bp = BiPlan(fresh, g12, ingest.spread(g12) * g12.rowscale, ingest.spread(g21) * g21.rowscale, evidence=True)
        done, conv = bp.run(100, 1e-4)
        assert conv == want["k"]
        s1, s2 = bp.result()
        np.testing.assert_allclose(s1, want["S1"], rtol=RTOL, atol=1e-30)
        np.testing.assert_allclose(s2, want["S2"], rtol=RTOL, atol=1e-30)
        bp.free()
