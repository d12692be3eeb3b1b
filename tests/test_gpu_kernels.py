"""Kernel-level parity on a real MI355X, through the C ABI (ctypes -> libsimrank_hip.so).

Each kernel entry point is compared with a float64 NumPy/SciPy evaluation of the same
documented semantics on seeded inputs, over the shapes that exercise every code path:
vector and scalar variants, every panel width, ragged tails, empty and very long rows,
transposed/blocked stores, each epilogue feature, counter exactness.
Tolerance for float32 results: 1e-5 relative (north_star); integer results: exact.
"""
import numpy as np
import pytest
import scipy.sparse as sp

from simrank_amd.ingest import CSR

pytestmark = pytest.mark.gpu
RTOL = 1e-5


@pytest.fixture(scope="module")
def ops():
    from simrank_amd.engine import HipOps
    o = HipOps(0)
    yield o
    o.set_tuning(panel=0, xcd_map=1)


def random_csr(M, K, avg, seed, heavy=()):
    """Random pattern with some empty rows and optional very long rows."""
    rng = np.random.default_rng(seed)
    rows = []
    for a in range(M):
        if a in heavy:
            d = heavy[a]
        elif rng.random() < 0.1:
            d = 0
        else:
            d = min(K, rng.poisson(avg))
        rows.append(np.sort(rng.choice(K, size=d, replace=False)))
    rowptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int32)
    col = (np.concatenate(rows) if rowptr[-1] else np.empty(0)).astype(np.int32)
    rs = rng.random(M) + 0.1
    rs[rng.random(M) < 0.05] = 0.0
    return CSR(M, K, rowptr, col, rs)


def dense64(csr):
    pat = sp.csr_matrix((np.ones(csr.col.size), csr.col, csr.rowptr), shape=(csr.n_rows, csr.n_cols))
    return sp.diags(csr.rowscale.astype(np.float32).astype(np.float64)) @ pat


def put(ops, host, ld=None, dtype=np.float32):
    m = ops.matrix(host.shape[0], host.shape[1], dtype, ld=ld)
    ops.upload(m, host.astype(dtype))
    return m


@pytest.mark.parametrize("panel", [0, 16, 32, 64, 128, 256])
@pytest.mark.parametrize("shape", [(300, 257, 100), (64, 64, 64), (1000, 777, 36), (5, 5, 5)])
def test_spmm_plain(ops, panel, shape):
    M, K, L = shape
    ops.set_tuning(panel=panel)
    csr = random_csr(M, K, 9, seed=M + L, heavy={1: min(K, 200), 3: min(K, 70)} if M > 3 else {})
    rng = np.random.default_rng(1)
    X = rng.random((K, L)).astype(np.float32)
    g = ops.graph(csr)
    x = put(ops, X)
    y = ops.matrix(M, L)
    ops.spmm(g, x, y)
    want = dense64(csr) @ X.astype(np.float64)
    np.testing.assert_allclose(ops.download(y), want, rtol=RTOL, atol=1e-30)


@pytest.mark.parametrize("ld_extra", [1, 2, 3])
def test_spmm_scalar_variant_for_unaligned_ld(ops, ld_extra):
    M, K, L = 130, 90, 45
    csr = random_csr(M, K, 7, seed=5, heavy={0: 90})
    X = np.random.default_rng(2).random((K, L)).astype(np.float32)
    g = ops.graph(csr)
    x = put(ops, X, ld=L + ld_extra)
    y = ops.matrix(M, L, ld=L + ld_extra)
    ops.spmm(g, x, y)
    np.testing.assert_allclose(ops.download(y), dense64(csr) @ X.astype(np.float64), rtol=RTOL,
                               atol=1e-30)


@pytest.mark.parametrize("panel", [0, 16, 32, 64, 256])
@pytest.mark.parametrize("shape,tb", [((300, 200, 100), 0), ((300, 200, 100), 128),
                                      ((257, 64, 31), 100), ((64, 64, 64), 16), ((5, 7, 3), 2)])
def test_spmm_transposed_store(ops, panel, shape, tb):
    M, K, L = shape
    ops.set_tuning(panel=panel)
    csr = random_csr(M, K, 6, seed=L, heavy={2: min(K, 150)})
    X = np.random.default_rng(3).random((K, L)).astype(np.float32)
    g = ops.graph(csr)
    x = put(ops, X)
    want = dense64(csr) @ X.astype(np.float64)
    if tb == 0:                       # single block, pitched rows: Y[c*ld + a]
        y = ops.matrix(L, M)
        ops.spmm(g, x, y, transpose_out=True)
        np.testing.assert_allclose(ops.download(y), want.T, rtol=RTOL, atol=1e-30)
    else:                             # per-destination blocks, contiguous, rows padded
        for pad in (0, 5):
            nblk = -(-M // tb)
            y = ops.matrix(1, nblk * L * (tb + pad), ld=nblk * L * (tb + pad))
            ops.spmm(g, x, y, transpose_out=True, t_block=tb, t_pad=pad)
            flat = ops.download(y).ravel()
            for h in range(nblk):
                lo, hi = h * tb, min(M, (h + 1) * tb)
                w = hi - lo + pad
                blk = flat[h * L * (tb + pad): h * L * (tb + pad) + L * w].reshape(L, w)
                np.testing.assert_allclose(blk[:, :hi - lo], want[lo:hi].T, rtol=RTOL, atol=1e-30)


@pytest.mark.parametrize("aligned", [True, False])
@pytest.mark.parametrize("features", ["coef", "evidence", "apriori", "all"])
def test_spmm_epilogue(ops, features, aligned):
    """Column block [c0, c0+L) of an M x M update: scale, 1-2^-count, prior blend, diag <- 1,
    and the exact number of elements that moved by more than eps."""
    M, K, L, c0 = 200, 150, 64 if aligned else 61, 40
    ops.set_tuning(panel=0)
    csr = random_csr(M, K, 8, seed=9)
    rng = np.random.default_rng(4)
    X = rng.random((K, L)).astype(np.float32)
    cnt = rng.integers(0, 6, size=(M, L)).astype(np.uint8)
    cnt[0, :4] = [0, 1, 30, 255]
    prior = rng.random((M, L)).astype(np.float32)
    prev = (rng.random((M, L)) * 2).astype(np.float32)
    ld = None if aligned else L + 1
    g, x = ops.graph(csr), put(ops, X, ld=ld)
    y = ops.matrix(M, L, ld=ld)
    ep = dict(coef=0.8, diag_col0=c0, previous=put(ops, prev, ld=ld), eps=0.37)
    want = 0.8 * (dense64(csr) @ X.astype(np.float64))
    if features in ("evidence", "all"):
        ep["evidence"] = put(ops, cnt, dtype=np.uint8, ld=None if aligned else L + 3)
        want = want * (1 - 0.5 ** cnt.astype(np.float64))
    if features in ("apriori", "all"):
        ep.update(apriori=put(ops, prior, ld=ld), lbd=0.3)
        want = (1 - np.float32(0.3)) * want + np.float32(0.3) * prior.astype(np.float64)
    for c in range(L):
        if c0 + c < M:
            want[c0 + c, c] = 1.0
    ops.spmm(g, x, y, epilogue=ep)
    got = ops.download(y)
    np.testing.assert_allclose(got, want, rtol=RTOL, atol=1e-30)
    assert got[c0, 0] == 1.0 and got[c0 + L - 1, L - 1] == 1.0
    n = ops.read_changed()
    assert n == int((np.abs(got.astype(np.float64) - prev.astype(np.float64)) > 0.37).sum())
    assert 0 < n < M * L


def test_fill_identity(ops):
    for rows, cols, c0 in [(10, 10, 0), (100, 30, 50), (64, 64, 0), (33, 7, 30)]:
        m = ops.matrix(rows, cols)
        ops.fill_identity(m, c0)
        want = np.zeros((rows, cols), dtype=np.float32)
        for c in range(cols):
            if c0 + c < rows:
                want[c0 + c, c] = 1
        np.testing.assert_array_equal(ops.download(m), want)


def test_evidence_counts_exact_and_saturating(ops):
    M, K = 180, 400
    csr = random_csr(M, K, 12, seed=21, heavy={0: 400, 1: 300, 2: 256})
    g = ops.graph(csr)
    live = sp.diags((csr.rowscale.astype(np.float32) > 0).astype(np.float64)) @ sp.csr_matrix(
        (np.ones(csr.col.size), csr.col, csr.rowptr), shape=(M, K))
    want = np.minimum((live @ live.T).toarray(), 255).astype(np.uint8)
    assert want.max() == 255 and (want == 0).any()
    for c0, L in [(0, M), (37, 100), (M - 1, 1)]:
        out = ops.matrix(M, L, np.uint8)
        ops.evidence_counts(g, c0, out)
        np.testing.assert_array_equal(ops.download(out), want[:, c0:c0 + L])


@pytest.mark.parametrize("M,K,blocked", [(180, 400, False), (257, 90, True), (1000, 1000, True), (33, 500, True),
                                         (2048, 3000, True), (1, 5, False)])
def test_evidence_counts_upper_triangle_and_mirror(ops, M, K, blocked):
    """Round 4: a whole square block of counts is computed as its upper triangle (row a counts the paths to b >= a:
    half the LDS atomics) and mirrored by a second kernel — bit-equal to the exact counts, either layout, sizes off
    the 32-grid, rows without weight, saturation (`_cal_Evidence`, SimRank.py:311-320)."""
    csr = random_csr(M, K, 9, seed=M + K, heavy={0: min(K, 400), M // 2: min(K, 300)} if M > 2 else ())
    live = sp.diags((csr.rowscale.astype(np.float32) > 0).astype(np.float64)) @ sp.csr_matrix(
        (np.ones(csr.col.size), csr.col, csr.rowptr), shape=(M, K))
    want = np.minimum((live @ live.T).toarray(), 255).astype(np.uint8)
    got = {}
    for tri in (1, 0):
        ops.set_tuning(ev_tri=tri)
        try:
            g = ops.graph(csr)
            out = ops.matrix(M, M, np.uint8, blocked=True) if blocked else ops.matrix(M, M, np.uint8)
            ops.evidence_counts(g, 0, out)
            got[tri] = ops.download(out)
        finally:
            ops.set_tuning(ev_tri=1)
    np.testing.assert_array_equal(got[1], want)
    np.testing.assert_array_equal(got[0], want)


def hub_csr(M, K, n_hubs, p_hub, avg, seed):
    """Random pattern whose first n_hubs columns are referenced by a share p_hub of the rows each (some rows without weight)."""
    rng = np.random.default_rng(seed)
    rows = []
    for a in range(M):
        hubs = np.nonzero(rng.random(n_hubs) < p_hub)[0]
        rest = n_hubs + rng.choice(K - n_hubs, size=min(K - n_hubs, rng.poisson(avg)), replace=False) if K > n_hubs else []
        rows.append(np.unique(np.concatenate([hubs, rest]).astype(np.int64)))
    rowptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int32)
    col = np.concatenate(rows).astype(np.int32)
    rs = rng.random(M) + 0.1
    rs[rng.random(M) < 0.05] = 0.0
    return CSR(M, K, rowptr, col, rs)


@pytest.mark.parametrize("M,K,n_hubs,p_hub,blocked,block", [
    (600, 900, 40, 0.5, True, None), (600, 900, 40, 0.5, False, None), (1000, 1000, 300, 0.95, True, None),
    (257, 300, 33, 0.8, True, None), (2050, 2500, 70, 0.3, True, None), (1024, 1024, 64, 0.4, False, (200, 333)),
    (900, 1200, 50, 0.6, False, (128, 512)), (130, 260, 130, 0.7, False, (1, 129))])
def test_evidence_counts_with_hub_columns_on_the_matrix_cores(ops, M, K, n_hubs, p_hub, blocked, block):
    """Round 5: the pairs through the columns that very many rows share are counted as an i8 product of their 0/1 image
    (v_mfma_i32_32x32x32_i8, evidence_hub_kernel), everything else on the LDS counters starting from those counts —
    the same bytes as the LDS counters alone (tuning ev_hub = 0) and as the exact product (`_cal_Evidence`,
    SimRank.py:311-320): saturation beyond 255 common hubs, rows without weight, sizes off the 128-grid, both layouts,
    a column block that starts off the grid (a sharded rank's), the upper-triangle form and the full one."""
    csr = hub_csr(M, K, n_hubs, p_hub, 3, seed=M + n_hubs)
    live = sp.diags((csr.rowscale.astype(np.float32) > 0).astype(np.float64)) @ sp.csr_matrix(
        (np.ones(csr.col.size), csr.col, csr.rowptr), shape=(M, K))
    want = np.minimum((live @ live.T).toarray(), 255).astype(np.uint8)
    if p_hub > 0.9:
        assert want.max() == 255
    c0, L = block if block else (0, M)
    got = {}
    for hub, tri in ((1, 1), (1, 0), (0, 1)):
        ops.set_tuning(ev_hub=hub, ev_tri=tri)            # (1/1000 of the rows, at least 48: every planted column qualifies)
        try:
            g = ops.graph(csr)
            out = ops.matrix(M, L, np.uint8, blocked=True) if blocked else ops.matrix(M, L, np.uint8)
            ops.evidence_counts(g, c0, out)
            got[(hub, tri)] = ops.download(out)
            ops.evidence_counts(g, c0, out)                # (the image is kept with the graph: a second call reuses it)
            np.testing.assert_array_equal(ops.download(out), got[(hub, tri)])
        finally:
            ops.set_tuning(ev_hub=14, ev_tri=1)
    for key, val in got.items():
        np.testing.assert_array_equal(val, want[:, c0:c0 + L], err_msg=str(key))


@pytest.mark.parametrize("n,blocked,col0,L", [(700, True, 0, 700), (700, False, 0, 700), (1030, False, 256, 300), (25000, True, 0, 25000)])
def test_graph_creation_that_counts_on_the_side(ops, n, blocked, col0, L):
    """simrank_graph_create_counting: the evidence counts of the graph's own pattern are queued once the pattern is on the
    device and run beside the host threads that build the graph's plans (the last case is large enough for those
    threads) — the same bytes as counting afterwards; the graph is the ordinary one."""
    csr = random_csr(n, n, 7 if n < 5000 else 3, seed=n + L, heavy={0: min(n, 400), n // 2: min(n, 300)})
    cnt = ops.matrix(n, L, np.uint8, blocked=True) if blocked else ops.matrix(n, L, np.uint8)
    g = ops.graph(csr, counting=(cnt, col0))
    ref = ops.matrix(n, L, np.uint8, blocked=True) if blocked else ops.matrix(n, L, np.uint8)
    g2 = ops.graph(csr)
    ops.evidence_counts(g2, col0, ref)
    got, want = ops.download(cnt), ops.download(ref)
    assert np.array_equal(got, want) and want.any()
    assert ops.fused_stats(g) == ops.fused_stats(g2)
    x = ops.matrix(n, 64)
    ops.upload(x, np.random.default_rng(1).random((n, 64)).astype(np.float32))
    y, y2 = ops.matrix(n, 64), ops.matrix(n, 64)
    ops.spmm(g, x, y)
    ops.spmm(g2, x, y2)
    assert np.array_equal(ops.download(y), ops.download(y2))


def test_densify(ops):
    csr = random_csr(150, 90, 10, seed=3)
    g = ops.graph(csr)
    wd = ops.matrix(150, 90)
    ops.densify(g, wd)
    np.testing.assert_array_equal(ops.download(wd), dense64(csr).toarray().astype(np.float32))


@pytest.mark.parametrize("shape", [(128, 128, 32), (256, 384, 160), (200, 130, 77), (5, 3, 2),
                                   (1024, 1024, 1024)])
def test_gemm_nt_mfma(ops, shape):
    """C = A . B^T on v_mfma_f32_32x32x2_f32; asymmetric operands catch a transposed tile."""
    M, N, K = shape
    rng = np.random.default_rng(M + N)
    A = (rng.random((M, K)) - 0.3).astype(np.float32)
    B = (rng.random((N, K)) - 0.6).astype(np.float32)
    c = ops.matrix(M, N)
    ops.gemm_nt(put(ops, A), put(ops, B), c, M, N, K)
    want = A.astype(np.float64) @ B.astype(np.float64).T
    got = ops.download(c)
    scale = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64).T
    assert np.max(np.abs(got - want) / scale) < 2e-6


def test_gemm_nt_epilogue(ops):
    M, K = 200, 90
    rng = np.random.default_rng(8)
    A = rng.random((M, K)).astype(np.float32)
    B = rng.random((M, K)).astype(np.float32)
    cnt = rng.integers(0, 5, size=(M, M)).astype(np.uint8)
    prior = rng.random((M, M)).astype(np.float32)
    prev = (rng.random((M, M)) * 40).astype(np.float32)
    c = ops.matrix(M, M)
    ep = dict(coef=0.6, evidence=put(ops, cnt, dtype=np.uint8), apriori=put(ops, prior), lbd=0.25,
              previous=put(ops, prev), eps=3.0, diag_col0=0)
    ops.gemm_nt(put(ops, A), put(ops, B), c, M, M, K, epilogue=ep)
    want = 0.6 * (A.astype(np.float64) @ B.astype(np.float64).T) * (1 - 0.5 ** cnt.astype(np.float64))
    want = 0.75 * want + 0.25 * prior
    np.fill_diagonal(want, 1)
    got = ops.download(c)
    np.testing.assert_allclose(got, want, rtol=RTOL)
    assert ops.read_changed() == int((np.abs(got.astype(np.float64) - prev) > 3.0).sum())


def test_download_f64_matches_plain_download(ops):
    rng = np.random.default_rng(0)
    h = rng.random((700, 333)).astype(np.float32)
    m = put(ops, h)
    np.testing.assert_array_equal(ops.download_f64(m), h.astype(np.float64))


def test_epilogue_apply_standalone(ops):
    """The un-fused epilogue (asymmetric-prior path), in place and out of place."""
    M, L, c0 = 150, 77, 20
    rng = np.random.default_rng(6)
    Q = rng.random((M, L)).astype(np.float32)
    cnt = rng.integers(0, 4, size=(M, L)).astype(np.uint8)
    prior = rng.random((M, L)).astype(np.float32)
    prev = rng.random((M, L)).astype(np.float32)
    ep = dict(coef=0.8, evidence=put(ops, cnt, dtype=np.uint8), apriori=put(ops, prior), lbd=0.4,
              previous=put(ops, prev), eps=0.2, diag_col0=c0)
    want = 0.8 * Q.astype(np.float64) * (1 - 0.5 ** cnt.astype(np.float64))
    want = (1 - np.float32(0.4)) * want + np.float32(0.4) * prior.astype(np.float64)
    for c in range(L):
        want[c0 + c, c] = 1.0
    q = put(ops, Q)
    y = ops.matrix(M, L)
    ops.epilogue_apply(q, y, M, L, ep)
    got = ops.download(y)
    np.testing.assert_allclose(got, want, rtol=RTOL)
    assert ops.read_changed() == int((np.abs(got.astype(np.float64) - prev) > 0.2).sum())
    ops.epilogue_apply(q, q, M, L, ep)
    np.testing.assert_array_equal(ops.download(q), got)


@pytest.mark.parametrize("n", [64, 200, 1000, 1031])
def test_symmetric_leg2_upper_triangle_and_mirror(ops, n):
    """ep.symmetric: only tiles on/above the diagonal are computed, the rest is their mirror
    image.  Same values as the full computation (to rounding), exactly symmetric, and the
    convergence count equals the count over the full matrix."""
    csr = random_csr(n, n, 10, seed=n, heavy={3: min(n, 300), 40: min(n, 90)})
    rng = np.random.default_rng(n)
    S = rng.random((n, n)).astype(np.float32)
    S = ((S + S.T) / 2).astype(np.float32)
    np.fill_diagonal(S, 1)
    cnt = rng.integers(0, 5, size=(n, n))
    cnt = np.minimum(cnt, cnt.T).astype(np.uint8)
    g, s_in, tt = ops.graph(csr), put(ops, S), ops.matrix(n, n)
    ops.spmm(g, s_in, tt, transpose_out=True)                    # leg 1
    outs = {}
    for sym in (False, True):
        y = ops.matrix(n, n)
        ep = dict(coef=0.8, evidence=put(ops, cnt, dtype=np.uint8), previous=s_in, eps=0.05,
                  diag_col0=0, symmetric=sym)
        ops.spmm(g, tt, y, epilogue=ep)
        outs[sym] = (ops.download(y), ops.read_changed())
    full, mirrored = outs[False][0], outs[True][0]
    np.testing.assert_allclose(mirrored, full, rtol=RTOL, atol=1e-30)
    assert np.array_equal(mirrored[:32, 32:], mirrored[32:, :32].T)       # mirrored tiles: same bits
    assert outs[True][1] == int((np.abs(mirrored.astype(np.float64) - S) > 0.05).sum())
    assert outs[False][1] == int((np.abs(full.astype(np.float64) - S) > 0.05).sum())
    # the knobs are copied into a graph when it is created: an existing graph keeps its own
    ops.set_tuning(triangle=0)
    try:
        y = ops.matrix(n, n)
        ep = dict(coef=0.8, evidence=put(ops, cnt, dtype=np.uint8), previous=s_in, eps=0.05,
                  diag_col0=0, symmetric=True)
        ops.spmm(g, tt, y, epilogue=ep)
        assert np.array_equal(ops.download(y), mirrored)          # g was created with triangle = 1
        g0 = ops.graph(csr)
        ops.spmm(g0, tt, y, epilogue=ep)
    finally:
        ops.set_tuning(triangle=1)
    assert np.array_equal(ops.download(y), full)                  # knob off at creation -> plain path


@pytest.mark.parametrize("panel", [0, 32, 128])
def test_huge_rows_are_split_over_the_workgroup(ops, panel):
    """Rows of >= 1024 entries: four waves take a quarter each (phase A0); more huge rows in
    one workgroup than descriptor slots fall back to the owner wave."""
    M, K, L = 400, 3000, 70
    heavy = {0: 3000, 5: 1024, 6: 1025, 130: 2047, 131: 1500, 390: 1100}
    heavy.update({a: 1030 + a for a in range(256, 268)})        # 12 huge rows in one workgroup
    csr = random_csr(M, K, 20, seed=4, heavy=heavy)
    X = np.random.default_rng(9).random((K, L)).astype(np.float32)
    ops.set_tuning(panel=panel)
    g, x = ops.graph(csr), put(ops, X)
    want = dense64(csr) @ X.astype(np.float64)
    y = ops.matrix(M, L)
    ops.spmm(g, x, y)
    np.testing.assert_allclose(ops.download(y), want, rtol=RTOL, atol=1e-30)
    yt = ops.matrix(L, M)
    ops.spmm(g, x, yt, transpose_out=True)
    np.testing.assert_allclose(ops.download(yt), want.T, rtol=RTOL, atol=1e-30)
    ops.set_tuning(panel=0)


@pytest.mark.parametrize("shape,k,c0", [((300, 257), 7, 0), ((64, 64), 63, 0), ((100, 40), 5, 30),
                                        ((50, 3), 10, 0), ((1000, 1000), 16, 0)])
def test_topk_rows(ops, shape, k, c0):
    """k largest entries per row, largest first, ties by lower column, diagonal skipped,
    -1 padding when a row has fewer candidates — against a NumPy sort."""
    rows, cols = shape
    rng = np.random.default_rng(rows + k)
    h = rng.integers(0, 50, size=(rows, cols)).astype(np.float32) / 50      # many ties
    m = put(ops, h)
    idx, val = ops.topk_rows(m, k, col0=c0, exclude_diag=True)
    for r in range(rows):
        cand = [c for c in range(cols) if c != r - c0]
        cand.sort(key=lambda c: (-h[r, c], c))
        want = cand[:k]
        assert list(idx[r, :len(want)]) == [c0 + c for c in want]
        assert (idx[r, len(want):] == -1).all()
        np.testing.assert_array_equal(val[r, :len(want)], h[r, want])


@pytest.mark.parametrize("n,L", [(200, 200), (1031, 1031), (96, 64), (700, 333)])
def test_panel_blocked_operands_give_the_row_major_bits(ops, n, L):
    """The single-rank solver keeps its matrices panel-blocked (32-column panels of rows_pad rows);
    the lean kernel reads and writes either layout.  Same bits as row-major for the transposed
    leg, the plain leg with the whole epilogue, and (square case) the upper-triangle form."""
    csr = random_csr(n, n, 9, seed=n + L, heavy={3: min(n, 260), 50: min(n, 80)})
    rng = np.random.default_rng(n)
    X = rng.random((n, L)).astype(np.float32)
    cnt = rng.integers(0, 4, size=(n, L)).astype(np.uint8)
    prior = rng.random((n, L)).astype(np.float32)
    prev = rng.random((n, L)).astype(np.float32)
    if n == L:
        X = ((X + X.T) / 2).astype(np.float32)
        cnt = np.minimum(cnt, cnt.T)
        prior = ((prior + prior.T) / 2).astype(np.float32)
        prev = ((prev + prev.T) / 2).astype(np.float32)
    ops.set_tuning(fuse=0)               # the gather kernels of spmm.hip; the one-launch leg 1 is compared below
    g = ops.graph(csr)
    ops.set_tuning(fuse=1)

    def put_as(a, blocked, dtype=np.float32):
        m = ops.matrix(a.shape[0], a.shape[1], dtype, blocked=blocked)
        ops.upload(m, a)
        return m

    assert np.array_equal(ops.download(put_as(X, True)), X)            # layout round trip
    assert np.array_equal(ops.download(put_as(cnt, True, np.uint8)), cnt)
    out = {}
    for blocked in (False, True, "addr64"):
        if blocked == "addr64":              # the 64-bit addressing form (masked slots) on blocked operands
            ops.set_tuning(addr32=0, fuse=0)
            g = ops.graph(csr)
            ops.set_tuning(addr32=1, fuse=1)
        x = put_as(X, bool(blocked))
        yt = ops.matrix(L, n, blocked=bool(blocked))
        ops.spmm(g, x, yt, transpose_out=True)
        res = [ops.download(yt)]
        for sym in ((False, True) if n == L else (False,)):
            b = bool(blocked)
            y = ops.matrix(n, L, blocked=b)
            ep = dict(coef=0.7, evidence=put_as(cnt, b, np.uint8), apriori=put_as(prior, b),
                      lbd=0.25, previous=put_as(prev, b), eps=0.3, diag_col0=0, symmetric=sym)
            ops.spmm(g, x, y, epilogue=ep)
            res += [ops.download(y), ops.read_changed()]
        out[blocked] = res
    for other in (True, "addr64"):
        for a, b in zip(out[False], out[other]):
            assert np.array_equal(a, b)
    want = (dense64(csr) @ X.astype(np.float64)).T
    np.testing.assert_allclose(out[True][0], want, rtol=RTOL, atol=1e-30)
    # the one-launch leg 1 (fused.hip, what the solver runs on blocked operands): rows that go to the matrix
    # cores are summed in another order, so it agrees to rounding, not to the bit
    gf = ops.graph(csr)
    ytf = ops.matrix(L, n, blocked=True)
    ops.spmm(gf, put_as(X, True), ytf, transpose_out=True)
    np.testing.assert_allclose(ops.download(ytf), out[True][0], rtol=2e-6, atol=1e-30)
    # top-k and row hand-back straight from the blocked layout
    xb = put_as(X, True)
    idx, val = ops.topk_rows(xb, 5, exclude_diag=False)
    idx0, val0 = ops.topk_rows(put_as(X, False), 5, exclude_diag=False)
    assert np.array_equal(idx, idx0) and np.array_equal(val, val0)
    assert np.array_equal(ops.download_rows(xb, [0, n // 2, n - 1]), X[[0, n // 2, n - 1]])


@pytest.mark.parametrize("dtype", [np.float32, np.uint8])
def test_permute(ops, dtype):
    """dst[i, j] = src[row_idx[i], col_idx[j]], pitched operands, either index list optional."""
    rng = np.random.default_rng(8)
    src_h = (rng.random((70, 45)) * 200).astype(dtype)
    src = put(ops, src_h, ld=64 if dtype == np.float32 else 48, dtype=dtype)
    ri, ci = rng.integers(0, 70, size=33), rng.integers(0, 45, size=51)
    rv, cv = ops.index_vector(ri), ops.index_vector(ci)
    for rows, cols, want in ((rv, cv, src_h[ri][:, ci]), (rv, None, src_h[ri]), (None, cv, src_h[:, ci]),
                             (None, None, src_h)):
        dst = ops.matrix(want.shape[0], want.shape[1], dtype)
        ops.permute(src, dst, rows, cols)
        np.testing.assert_array_equal(ops.download(dst), want)


def test_topk_rows_with_column_ids(ops):
    """Ids given by the caller are what is reported and what breaks ties."""
    rng = np.random.default_rng(2)
    S = rng.integers(0, 4, size=(40, 30)).astype(np.float32)          # many ties
    ids = rng.permutation(1000)[:30].astype(np.int32)
    idx, val = ops.topk_rows(put(ops, S), 6, col0=5, exclude_diag=True, col_ids=ops.index_vector(ids))
    for a in range(40):
        cols = [c for c in range(30) if c != a - 5]
        cols.sort(key=lambda c: (-S[a, c], ids[c]))
        assert list(idx[a]) == [ids[c] for c in cols[:6]]
        assert list(val[a]) == [S[a, c] for c in cols[:6]]

@pytest.mark.parametrize("cols,k,with_ids", [(8192, 10, False), (9001, 10, True), (9001, 16, False), (8500, 20, True),
                                              (8200, 32, False), (9001, 40, True)])
def test_topk_rows_of_long_rows(ops, cols, k, with_ids):
    """Rows longer than 8192 columns take the one-pass selection for k <= 32 (the k best of every lane's
    columns in registers); same total order — value descending, id ascending — as the k-pass kernel (k = 40
    here), on rows full of ties, rows with fewer than k nonzeros, a width off the 4-column grid."""
    rows, c0 = 37, 3
    rng = np.random.default_rng(cols + k)
    h = (rng.integers(0, 6, size=(rows, cols)) * (rng.random((rows, cols)) < 0.01)).astype(np.float32) / 8   # sparse, ties
    h[1] = 0.0                                           # nothing but ties at zero
    h[2, :5] = [0.5, 0.5, 0.25, 0.5, 0.125]
    h[3] = rng.random(cols).astype(np.float32)           # dense, no ties
    ids = rng.permutation(3 * cols)[:cols].astype(np.int32) if with_ids else None
    key = ids if with_ids else c0 + np.arange(cols)
    idx, val = ops.topk_rows(put(ops, h), k, col0=c0, exclude_diag=True,
                             col_ids=ops.index_vector(ids) if with_ids else None)
    for r in range(rows):
        keep = np.ones(cols, bool)
        if 0 <= r - c0 < cols:
            keep[r - c0] = False
        cand = np.flatnonzero(keep)
        order = cand[np.lexsort((key[cand], -h[r, cand]))][:k]
        assert list(idx[r]) == [int(key[c]) for c in order]
        np.testing.assert_array_equal(val[r], h[r, order])


@pytest.mark.parametrize("rows,cols,k,with_ids", [(37, 8192, 10, False), (70, 9001, 10, True), (5, 33, 3, False), (64, 100, 16, True),
                                                  (129, 700, 20, True), (33, 2050, 32, False), (40, 300, 40, True), (1, 1, 1, False)])
def test_topk_rows_of_a_panel_blocked_matrix(ops, rows, cols, k, with_ids):
    """The one-pass selection on the panel-blocked layout itself (eight rows per wave, a lane group per row; k <= 32; k = 40
    takes the many-pass kernel): same total order as on a row-major copy — rows full of ties, rows with fewer than k
    candidates, row counts off the 8-row grid, widths off the 32-column grid, the diagonal at an offset."""
    c0 = 3
    rng = np.random.default_rng(rows * cols + k)
    h = (rng.integers(0, 6, size=(rows, cols)) * (rng.random((rows, cols)) < 0.05)).astype(np.float32) / 8
    h[0] = 0.0
    if rows > 3:
        h[3] = rng.random(cols).astype(np.float32)
    ids = rng.permutation(3 * cols + 5)[:cols].astype(np.int32) if with_ids else None
    key = ids if with_ids else c0 + np.arange(cols)
    m = ops.matrix(rows, cols, blocked=True)
    ops.upload(m, h)
    kk = min(k, cols)
    idx, val = ops.topk_rows(m, kk, col0=c0, exclude_diag=True, col_ids=ops.index_vector(ids) if with_ids else None)
    for r in range(rows):
        keep = np.ones(cols, bool)
        if 0 <= r - c0 < cols:
            keep[r - c0] = False
        cand = np.flatnonzero(keep)
        order = cand[np.lexsort((key[cand], -h[r, cand]))][:kk]
        assert list(idx[r, :len(order)]) == [int(key[c]) for c in order], r
        assert (idx[r, len(order):] == -1).all()
        np.testing.assert_array_equal(val[r, :len(order)], h[r, order])


@pytest.mark.parametrize("form", ["plain", "symmetric", "blocked"])
def test_count_any_short_circuits_the_comparison(ops, form):
    """epilogue.count_any: the kernel may stop comparing with the previous iterate once a difference
    is known.  Same result bits; the counter sum is zero exactly when the exact count is zero."""
    n = 1024
    csr = random_csr(n, n, 12, seed=77, heavy={5: 600})
    rng = np.random.default_rng(78)
    S = rng.random((n, n)).astype(np.float32)
    S = ((S + S.T) / 2).astype(np.float32)
    np.fill_diagonal(S, 1)
    blocked = form == "blocked"

    def mat(host=None):
        m = ops.matrix(n, n, blocked=True) if blocked else ops.matrix(n, n)
        if host is not None:
            ops.upload(m, host)
        return m
    g, s_in, tt = ops.graph(csr), mat(S), mat()
    ops.spmm(g, s_in, tt, transpose_out=True)
    ep = dict(coef=0.8, previous=s_in, eps=0.05, diag_col0=0, symmetric=form != "plain")
    y_exact, y_any = mat(), mat()
    ops.spmm(g, tt, y_exact, epilogue=dict(ep))
    exact = ops.read_changed()
    ops.spmm(g, tt, y_any, epilogue=dict(ep, count_any=True))
    some = ops.read_changed()
    assert exact > 1000 and 0 < some <= exact
    assert np.array_equal(ops.download(y_any), ops.download(y_exact))
    # nothing moves: compare the result with itself (eps far above any difference) -> exactly zero, and
    # with eps = 0 against itself -> still zero (strict >)
    for eps in (10.0, 0.0):
        ep0 = dict(ep, previous=y_exact, eps=eps, count_any=True)
        y2 = mat()
        ops.spmm(g, tt, y2, epilogue=ep0)
        assert ops.read_changed() == 0
        assert np.array_equal(ops.download(y2), ops.download(y_exact))


@pytest.mark.parametrize("world,mb,balance", [(2, 64, 2), (4, 128, 2), (3, 96, 0), (8, 32, 2)])
def test_half_form_shard_leg2(ops, world, mb, balance):
    """simrank_spmm_shard: for every shard h, row tile i and column tile j of the rank, i <= j is
    computed with the bits of the full form, i < j is also stored transposed — in place for the
    rank's own shard, else packed in the send chunk of rank h; i > j is left alone.  The counter
    counts a mirrored element twice.  simrank_shard_unpack puts received chunks in place."""
    n, T = world * mb, mb // 32
    rng = np.random.default_rng(world * 1000 + mb)
    lens = np.minimum(n, (rng.pareto(1.1, size=n) * 5).astype(int) + (rng.random(n) < 0.9))
    lens[rng.choice(n, 3, replace=False)] = [n, n // 2, 300 % n]
    # ascending inside every shard, like the solver's dealt order
    lens = np.concatenate([np.sort(lens[h * mb:(h + 1) * mb]) for h in range(world)])
    rows = [np.sort(rng.choice(n, size=d, replace=False)) for d in lens]
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    csr = CSR(n, n, rowptr, np.concatenate(rows).astype(np.int32), rng.random(n) + 0.1)
    ops.set_tuning(balance=balance, dense_min=0)
    try:
        g = ops.graph(csr)
    finally:
        ops.set_tuning(balance=2, dense_min=4)
    chunk = max(1, T * (T - 1) // 2 * 1024)
    for rank in (0, world - 1):
        X = rng.random((n, mb)).astype(np.float32)
        prev = rng.random((n, mb)).astype(np.float32)
        cnt = rng.integers(0, 4, size=(n, mb)).astype(np.uint8)
        x, pv, ev = put(ops, X), put(ops, prev), put(ops, cnt, dtype=np.uint8)
        ep = dict(coef=0.8, evidence=ev, previous=pv, eps=0.3, diag_col0=rank * mb)
        full_m = ops.matrix(n, mb)
        ops.spmm(g, x, full_m, epilogue=dict(ep))
        full, full_changed = ops.download(full_m), ops.read_changed()
        junk = np.full((n, mb), -7.0, np.float32)
        y, send = put(ops, junk), put(ops, np.full((world, chunk), -7.0, np.float32))
        ops.spmm_shard(g, x, y, dict(ep), rank, world, send, chunk)
        got, changed, sent = ops.download(y), ops.read_changed(), ops.download(send)
        moved = np.abs(full.astype(np.float64) - prev) > 0.3
        assert int(moved.sum()) == full_changed
        want, want_changed = junk.copy(), 0
        want_send = np.full((world, chunk), -7.0, np.float32)
        for h in range(world):
            for i in range(T):
                r = slice(h * mb + 32 * i, h * mb + 32 * i + 32)
                for j in range(i, T):
                    c = slice(32 * j, 32 * j + 32)
                    want[r, c] = full[r, c]
                    want_changed += int(moved[r, c].sum()) * (2 if i < j else 1)
                    if i < j and h == rank:
                        want[rank * mb + 32 * j:rank * mb + 32 * j + 32, 32 * i:32 * i + 32] = full[r, c].T
                    elif i < j:
                        slot = j * (j - 1) // 2 + i
                        want_send[h, slot * 1024:(slot + 1) * 1024] = full[r, c].T.reshape(-1)
        assert np.array_equal(got, want)
        assert np.array_equal(sent, want_send)
        assert changed == want_changed
        # unpack: chunk h of the receive buffer holds what rank h sent to this rank
        recv_h = rng.random((world, chunk)).astype(np.float32)
        y2 = put(ops, junk)
        ops.shard_unpack(y2, put(ops, recv_h), chunk, rank, world, n)
        want2 = junk.copy()
        for h in range(world):
            if h == rank:
                continue
            for j in range(T):
                for i in range(j):
                    slot = j * (j - 1) // 2 + i
                    want2[h * mb + 32 * j:h * mb + 32 * j + 32, 32 * i:32 * i + 32] = \
                        recv_h[h, slot * 1024:(slot + 1) * 1024].reshape(32, 32)
        assert np.array_equal(ops.download(y2), want2)


@pytest.mark.parametrize("balance", [0, 1, 4])
def test_balanced_tiles(ops, balance):
    """Rows sorted by length (all the long ones in the last 32-row blocks): heavy tiles are cut
    into halves down to single rows, workgroups run last-first, leg 2 takes its launch list
    from the host.  Every form of the leg against NumPy, for uniform and balanced tilings."""
    n = 700
    rng = np.random.default_rng(balance)
    lens = np.sort(np.minimum(n, (rng.pareto(1.2, size=n) * 6).astype(int) + (rng.random(n) < 0.9)))
    lens[-3:] = [520, 600, 700]
    rows = [np.sort(rng.choice(n, size=d, replace=False)) for d in lens]
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    csr = CSR(n, n, rowptr, np.concatenate(rows).astype(np.int32), rng.random(n) + 0.1)
    ops.set_tuning(balance=balance)
    try:
        g = ops.graph(csr)
    finally:
        ops.set_tuning(balance=2)
    S = rng.random((n, n)).astype(np.float32)
    S = ((S + S.T) / 2).astype(np.float32)
    np.fill_diagonal(S, 1)
    W = dense64(csr)
    s_in, tt, y = put(ops, S), ops.matrix(n, n), ops.matrix(n, n)
    ops.spmm(g, s_in, y)
    np.testing.assert_allclose(ops.download(y), W @ S.astype(np.float64), rtol=RTOL, atol=1e-30)
    ops.spmm(g, s_in, tt, transpose_out=True)
    np.testing.assert_allclose(ops.download(tt), (W @ S.astype(np.float64)).T, rtol=RTOL, atol=1e-30)
    cnt = rng.integers(0, 5, size=(n, n))
    cnt = np.minimum(cnt, cnt.T).astype(np.uint8)
    want = 0.8 * (W @ S.astype(np.float64) @ W.T) * (1 - 0.5 ** cnt.astype(np.float64))
    np.fill_diagonal(want, 1.0)
    for sym in (True, False):
        ops.spmm(g, tt, y, epilogue=dict(coef=0.8, evidence=put(ops, cnt, dtype=np.uint8), previous=s_in,
                                         eps=0.05, diag_col0=0, symmetric=sym))
        got = ops.download(y)
        np.testing.assert_allclose(got, want, rtol=RTOL, atol=1e-30)
        assert ops.read_changed() == int((np.abs(got.astype(np.float64) - S) > 0.05).sum())


# ---- block-dense part on the matrix cores (blockdense.hip) ----------------------------------
import contextlib


@contextlib.contextmanager
def dense_knobs(ops, dense_min=3, dense_cols=32, dense_sym=1):
    """Selection knobs of the dense part for small test graphs (copied into a graph when it is
    created); the tuned defaults are restored on exit."""
    ops.set_tuning(dense_min=dense_min, dense_cols=dense_cols, dense_sym=dense_sym)
    try:
        yield
    finally:
        ops.set_tuning(dense_min=4, dense_cols=128, dense_sym=-1)


def corner_csr(M, K, seed, hubs=120, p_hub=0.35, avg=5):
    """Sparse random pattern whose LAST rows reference the first `hubs` columns densely — the
    corner a power-law graph sorted by row length has."""
    rng = np.random.default_rng(seed)
    rows = []
    for a in range(M):
        c = set(rng.choice(K, size=min(K, rng.poisson(avg)), replace=False).tolist())
        if a >= M // 2:
            c |= set(np.flatnonzero(rng.random(min(hubs, K)) < p_hub * (a / M)).tolist())
        rows.append(np.array(sorted(c), dtype=np.int32))
    rowptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int32)
    return CSR(M, K, rowptr, np.concatenate(rows).astype(np.int32), rng.random(M) + 0.1)


@pytest.mark.parametrize("shape", [(520, 400, 333), (384, 384, 384), (1000, 300, 70), (130, 200, 2)])
def test_dense_part_matches_numpy_and_the_gather_path(ops, shape):
    """Entries of dense (row block, column) pairs go through bf16x3 MFMA, the rest is gathered:
    same result as NumPy in f64 (1e-5) and as the all-gather path (rounding only), in the plain
    and the transposed form, ragged widths included."""
    M, K, L = shape
    csr = corner_csr(M, K, seed=M + L)
    X = (np.random.default_rng(5).random((K, L)) ** 3).astype(np.float32)
    want = dense64(csr) @ X.astype(np.float64)
    with dense_knobs(ops):
        g = ops.graph(csr)
        nt, dk, cov = ops.dense_stats(g)
        assert nt >= 1 and dk >= 32 and 0 < cov < csr.nnz
        x, y, yt = put(ops, X), ops.matrix(M, L), ops.matrix(L, M)
        ops.spmm(g, x, y)
        ops.spmm(g, x, yt, transpose_out=True)
        got, got_t = ops.download(y), ops.download(yt)
    np.testing.assert_allclose(got, want, rtol=RTOL, atol=1e-30)
    np.testing.assert_allclose(got_t, want.T, rtol=RTOL, atol=1e-30)
    with dense_knobs(ops, dense_min=0):             # the same pattern without a dense part
        g0 = ops.graph(csr)
        assert ops.dense_stats(g0)[0] == 0
        ops.spmm(g0, x, y)
    np.testing.assert_allclose(got, ops.download(y), rtol=2e-6, atol=1e-30)


def test_dense_part_split_is_exact(ops):
    """One dense entry per row: the bf16 hi + mid + lo split must hand the f32 operand back
    bit for bit (24 mantissa bits, either sign, any exponent whose low-order term is still a
    normal bf16 number: |x| > ~1e-33)."""
    M, K, L = 256, 64, 128
    rows = [np.array([a % 64], dtype=np.int32) for a in range(M)]      # every column: 2 rows per block
    csr = CSR(M, K, np.arange(M + 1, dtype=np.int32), np.concatenate(rows), np.ones(M))
    rng = np.random.default_rng(0)
    X = (rng.standard_normal((K, L)) * np.exp(rng.uniform(-60, 60, size=(K, L)))).astype(np.float32)
    X[0, :8] = [0.0, 1.0, -1.0, 2.0 ** -100, 1 + 2.0 ** -23, 16777215.0, -3.0000002, 1e-30]
    with dense_knobs(ops, dense_min=2):
        g = ops.graph(csr)
        assert ops.dense_stats(g) == (2, 128, M)
        x, y = put(ops, X), ops.matrix(M, L)
        ops.spmm(g, x, y)
    assert np.array_equal(ops.download(y), X[np.arange(M) % 64])


@pytest.mark.parametrize("n", [256, 700])
def test_dense_part_in_a_whole_update(ops, n):
    """Both legs with dense sets, fused epilogue, upper-triangle and full forms, convergence count."""
    csr = corner_csr(n, n, seed=n, hubs=150)
    with dense_knobs(ops):
        _whole_update_with_dense_sets(ops, csr, n)


def _whole_update_with_dense_sets(ops, csr, n):
    g = ops.graph(csr)
    assert ops.dense_stats(g)[0] >= 1
    rng = np.random.default_rng(n)
    S = rng.random((n, n)).astype(np.float32)
    S = ((S + S.T) / 2).astype(np.float32)
    np.fill_diagonal(S, 1)
    W = dense64(csr)
    cnt = rng.integers(0, 5, size=(n, n))
    cnt = np.minimum(cnt, cnt.T).astype(np.uint8)
    want = 0.8 * (W @ S.astype(np.float64) @ W.T) * (1 - 0.5 ** cnt.astype(np.float64))
    np.fill_diagonal(want, 1.0)
    s_in, tt, y = put(ops, S), ops.matrix(n, n), ops.matrix(n, n)
    ops.spmm(g, s_in, tt, transpose_out=True)
    for sym in (True, False):
        ops.spmm(g, tt, y, epilogue=dict(coef=0.8, evidence=put(ops, cnt, dtype=np.uint8), previous=s_in,
                                         eps=0.05, diag_col0=0, symmetric=sym))
        got = ops.download(y)
        np.testing.assert_allclose(got, want, rtol=RTOL, atol=1e-30)
        assert ops.read_changed() == int((np.abs(got.astype(np.float64) - S) > 0.05).sum())
        if sym:                                   # mirrored tiles carry the same bits
            assert np.array_equal(got[:32, 32:], got[32:, :32].T)


def test_dense_part_with_a_set_cut_into_units(ops):
    """A dense set of more than 2048 columns is cut into units with one slab of partial sums
    each; the gather leg adds the slabs of a row block in a fixed order."""
    M, K, L = 300, 6000, 100
    csr = corner_csr(M, K, seed=3, hubs=5500, p_hub=0.6)
    g = ops.graph(csr)                                       # default knobs
    nt, dk, cov = ops.dense_stats(g)
    assert nt >= 1 and dk > 2048 + 2048
    X = np.random.default_rng(8).random((K, L)).astype(np.float32)
    want = dense64(csr) @ X.astype(np.float64)
    x, y, yt = put(ops, X), ops.matrix(M, L), ops.matrix(L, M)
    ops.spmm(g, x, y)
    ops.spmm(g, x, yt, transpose_out=True)
    np.testing.assert_allclose(ops.download(y), want, rtol=RTOL, atol=1e-30)
    np.testing.assert_allclose(ops.download(yt), want.T, rtol=RTOL, atol=1e-30)
    first = ops.download(y)
    ops.spmm(g, x, y)
    assert np.array_equal(first, ops.download(y))            # reproducible


@pytest.mark.parametrize("seed", range(16))
def test_dense_part_randomized(ops, seed):
    """Random shapes (row counts off the 128-row block grid, ragged widths), random corner
    densities and selection knobs, every form of the leg: the dense part + remainder must equal
    NumPy, and the upper-triangle form must count like the full one."""
    rng = np.random.default_rng(1000 + seed)
    square = seed % 2 == 0
    M = int(rng.integers(130, 900))
    K = M if square else int(rng.integers(60, 900))
    L = K if square else int(rng.integers(1, 700))
    csr = corner_csr(M, K, seed=seed, hubs=int(rng.integers(40, max(41, K))), p_hub=float(rng.uniform(0.1, 0.9)),
                     avg=int(rng.integers(1, 12)))
    W = dense64(csr)
    with dense_knobs(ops, dense_min=int(rng.integers(2, 5)), dense_cols=int(rng.choice([16, 32, 64]))):
        g = ops.graph(csr)
        if square:
            S = rng.random((M, M)).astype(np.float32)
            S = ((S + S.T) / 2).astype(np.float32)
            np.fill_diagonal(S, 1)
            cnt = rng.integers(0, 4, size=(M, M))
            cnt = np.minimum(cnt, cnt.T).astype(np.uint8)
            want = 0.8 * (W @ S.astype(np.float64) @ W.T) * (1 - 0.5 ** cnt.astype(np.float64))
            np.fill_diagonal(want, 1.0)
            s_in, tt, y = put(ops, S), ops.matrix(M, M), ops.matrix(M, M)
            ops.spmm(g, s_in, tt, transpose_out=True)
            np.testing.assert_allclose(ops.download(tt), (W @ S.astype(np.float64)).T, rtol=RTOL, atol=1e-30)
            for sym in (True, False):
                ops.spmm(g, tt, y, epilogue=dict(coef=0.8, evidence=put(ops, cnt, dtype=np.uint8),
                                                 previous=s_in, eps=0.05, diag_col0=0, symmetric=sym))
                got = ops.download(y)
                np.testing.assert_allclose(got, want, rtol=RTOL, atol=1e-30)
                assert ops.read_changed() == int((np.abs(got.astype(np.float64) - S) > 0.05).sum())
        else:
            X = (rng.random((K, L)) ** 2).astype(np.float32)
            want = W @ X.astype(np.float64)
            x, y, yt = put(ops, X), ops.matrix(M, L), ops.matrix(L, M)
            ops.spmm(g, x, y)
            ops.spmm(g, x, yt, transpose_out=True)
            np.testing.assert_allclose(ops.download(y), want, rtol=RTOL, atol=1e-30)
            np.testing.assert_allclose(ops.download(yt), want.T, rtol=RTOL, atol=1e-30)
