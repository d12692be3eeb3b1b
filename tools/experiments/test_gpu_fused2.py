"""Leg 1 as one PERSISTENT launch (csrc/fused2.hip, tuning fuse = 2) on a real MI355X, through the C ABI:
resident workgroups pulling pieces of blocks from per-XCD queues, operand rows through LDS rings, pieces of
split blocks meeting in memory.  Against float64 NumPy (1e-5, north_star) and against the one-launch leg of
round 3 (rounding only), over the shapes of tests/test_gpu_fused.py and over piece sizes from "every block
whole" down to "every block in many pieces".  First `.dot` of SimRank.py:139 / :298 / :301 / :361 / :420 / :423."""
import contextlib

import numpy as np
import pytest

from simrank_amd.ingest import CSR
from tests.test_gpu_kernels import corner_csr, dense64, random_csr

pytestmark = pytest.mark.gpu
RTOL = 1e-5
DEFAULTS = dict(fuse=1, fuse_min=3, fuse_steps=-1, fuse_unit=48, fuse_group=3, fuse_cap=40000, fuse_wgs=4)


@pytest.fixture(scope="module")
def ops():
    from simrank_amd.engine import HipOps
    o = HipOps(0)
    o.set_tuning(fuse=2, fuse_steps=1, fuse_min=2)
    yield o
    o.set_tuning(**DEFAULTS)


@contextlib.contextmanager
def knobs(ops, **kw):
    ops.set_tuning(**kw)
    try:
        yield
    finally:
        ops.set_tuning(**{**DEFAULTS, "fuse": 2, "fuse_min": 2, "fuse_steps": 1})


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
@pytest.mark.parametrize("cap", [1000, 4000, 1 << 30])
def test_persistent_leg1_matches_numpy_and_the_one_launch_leg(ops, shape, cap):
    M, K, L = shape
    csr = corner_csr(M, K, seed=M + L, hubs=min(K, 150))
    X = (np.random.default_rng(5).random((K, L)) ** 3).astype(np.float32)
    want = (dense64(csr) @ X.astype(np.float64)).T
    with knobs(ops, fuse_cap=cap):
        g = ops.graph(csr)
        got = leg1(ops, g, X, M)
        for _ in range(2):
            assert np.array_equal(got, leg1(ops, g, X, M))      # reproducible, tickets and queue heads reset
    np.testing.assert_allclose(got, want, rtol=RTOL, atol=1e-30)
    with knobs(ops, fuse=1):
        old = leg1(ops, ops.graph(csr), X, M)
    np.testing.assert_allclose(got, old, rtol=2e-6, atol=1e-30)


def test_persistent_split_is_exact(ops):
    """One entry per row, every column shared by two rows of a block: everything goes through the bf16
    hi + mid + lo split and must come back bit for bit."""
    M, K, L = 256, 64, 128
    rows = [np.array([a % 64], dtype=np.int32) for a in range(M)]
    csr = CSR(M, K, np.arange(M + 1, dtype=np.int32), np.concatenate(rows), np.ones(M))
    rng = np.random.default_rng(0)
    X = (rng.standard_normal((K, L)) * np.exp(rng.uniform(-60, 60, size=(K, L)))).astype(np.float32)
    X[0, :8] = [0.0, 1.0, -1.0, 2.0 ** -100, 1 + 2.0 ** -23, 16777215.0, -3.0000002, 1e-30]
    g = ops.graph(csr)
    assert np.array_equal(leg1(ops, g, X, M), X[np.arange(M) % 64].T)


def test_persistent_without_any_dense_set(ops):
    M = K = 500
    rows = [np.array(sorted({(7 * a + 3) % K, (11 * a + 5) % K}), dtype=np.int32) for a in range(M)]
    rowptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int32)
    csr = CSR(M, K, rowptr, np.concatenate(rows), np.random.default_rng(1).random(M) + 0.5)
    X = np.random.default_rng(2).random((K, 200)).astype(np.float32)
    for cap in (1000, 1 << 30):
        with knobs(ops, fuse_min=100, fuse_cap=cap):
            got = leg1(ops, ops.graph(csr), X, M)
        np.testing.assert_allclose(got, (dense64(csr) @ X.astype(np.float64)).T, rtol=RTOL, atol=1e-30)


def test_persistent_long_rows_and_many_pieces(ops):
    """A star (one row references every column) in pieces: the set is cut along its columns, the rows along the
    descending remainder order; many quads per wave exercise the ring's steady state and its tail."""
    M = K = 1500
    rng = np.random.default_rng(4)
    rows = [np.sort(rng.choice(K, size=3, replace=False)).astype(np.int32) for _ in range(M)]
    rows[700] = np.arange(K, dtype=np.int32)
    rows[701] = np.arange(0, K, 3, dtype=np.int32)
    rowptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int32)
    csr = CSR(M, K, rowptr, np.concatenate(rows), rng.random(M) + 0.1)
    X = rng.random((K, 257)).astype(np.float32)
    want = (dense64(csr) @ X.astype(np.float64)).T
    for cap in (1000, 3000, 1 << 30):
        with knobs(ops, fuse_cap=cap):
            np.testing.assert_allclose(leg1(ops, ops.graph(csr), X, M), want, rtol=RTOL, atol=1e-30)


def test_persistent_ids_beyond_16_bits(ops):
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
    for cap in (1500, 1 << 30):
        with knobs(ops, fuse_cap=cap):
            np.testing.assert_allclose(leg1(ops, ops.graph(csr), X, M), (dense64(csr) @ X.astype(np.float64)).T, rtol=RTOL,
                                       atol=1e-30)


@pytest.mark.parametrize("seed", range(16))
def test_persistent_randomized(ops, seed):
    rng = np.random.default_rng(7000 + seed)
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
               fuse_cap=int(rng.choice([1000, 2000, 8000, 1 << 30])), fuse_wgs=int(rng.integers(1, 5))):
        g = ops.graph(csr)
        got = leg1(ops, g, X, M)
        assert np.array_equal(got, leg1(ops, g, X, M))
    np.testing.assert_allclose(got, (dense64(csr) @ X.astype(np.float64)).T, rtol=RTOL, atol=1e-30)


def test_persistent_wide_sets_many_launches(ops):
    """A dense corner in many pieces, launch after launch (tickets and queue heads reset themselves), also for a
    second, narrower operand."""
    M, K, L = 300, 6000, 200
    csr = corner_csr(M, K, seed=3, hubs=3000, p_hub=0.6)
    X = np.random.default_rng(8).random((K, L)).astype(np.float32)
    want = (dense64(csr) @ X.astype(np.float64)).T
    with knobs(ops, fuse_cap=2000):
        g = ops.graph(csr)
        first = leg1(ops, g, X, M)
        for _ in range(3):
            assert np.array_equal(first, leg1(ops, g, X, M))
        narrow = leg1(ops, g, X[:, :70], M)
        assert np.array_equal(narrow, first[:70])
        assert np.array_equal(first, leg1(ops, g, X, M))
    np.testing.assert_allclose(first, want, rtol=RTOL, atol=1e-30)
    with knobs(ops, fuse_cap=1 << 30):
        whole = leg1(ops, ops.graph(csr), X, M)
    np.testing.assert_allclose(first, whole, rtol=2e-6, atol=1e-30)
