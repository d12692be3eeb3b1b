"""fp16-held matrices (csrc/half.hip, `fit(storage_precision="fp16")`) on a real MI355X, through the C ABI.
BASELINE.json config 5's reduced-precision mode: outside the 1e-5 parity bar by construction, so the tests
state the error it is allowed — ONE rounding to fp16 per stored value on top of f32 sums — and check it
against float64 NumPy on the same fp16 inputs (kernel level) and against the oracle (whole fits).
Both `.dot`s of SimRank.py:361, the element-wise lines :315-316 / :362 / :453, the test of :74."""
import numpy as np
import pytest

from simrank_amd.ingest import CSR
from tests.test_gpu_kernels import corner_csr, dense64, random_csr

pytestmark = pytest.mark.gpu
HALF_ULP = 2.0 ** -11          # relative half-ulp of fp16 for normal numbers


@pytest.fixture(scope="module")
def ops():
    from simrank_amd.engine import HipOps
    o = HipOps(0)
    # small test graphs: a dense set however few steps it makes; whole blocks (half.hip takes no units whose sums
    # meet in memory: fits on fp16-held matrices create their graphs that way themselves, driver.Side)
    o.set_tuning(fuse_steps=1, fuse_min=2, fuse_unit=1 << 20)
    yield o
    o.set_tuning(fuse=1, fuse_min=0, fuse_steps=-1, fuse_unit=48, fuse_group=3)


def put_half(ops, a, scale=1.0):
    """fp16 matrix holding ``a`` x ``scale`` (``a`` must be exactly representable after scaling)."""
    m = ops.matrix(a.shape[0], a.shape[1], np.float16, blocked=True)
    m.scale = scale
    ops.upload(m, a)
    return m


def put_blocked(ops, a, dtype=np.float32):
    m = ops.matrix(a.shape[0], a.shape[1], dtype, blocked=True)
    ops.upload(m, a.astype(dtype))
    return m


def close_to_rounded(got, want64, slack=1.02, scale=1.0):
    """``got`` (fp16 values / scale) is ``want64`` rounded once, up to the f32 summation error: within a bit
    more than half an fp16 ulp relative (normal range) or half the subnormal spacing absolute."""
    got = got.astype(np.float64)
    err = np.abs(got - want64)
    bound = slack * np.maximum(HALF_ULP * np.abs(want64), 2.0 ** -25 / scale) + 1e-6 * np.abs(want64)
    bad = err > bound
    assert not bad.any(), (int(bad.sum()), float(err[bad].max()), float(np.abs(want64[bad]).max()))


def test_layout_round_trip(ops):
    rng = np.random.default_rng(0)
    a = rng.random((77, 203)).astype(np.float16)
    m = put_half(ops, a)
    assert np.array_equal(ops.download(m), a)
    for scale in (1.0, 16384.0):
        s = ops.matrix(130, 130, np.float16, blocked=True)
        s.scale = scale
        ops.fill_identity(s, 0)
        assert np.array_equal(ops.download(s), np.eye(130, dtype=np.float32))
        b = put_half(ops, a / 4, scale)                 # (values x 2^14 stay below fp16's largest number)
        assert np.array_equal(ops.download(b), (a / 4).astype(np.float32))


@pytest.mark.parametrize("shape", [(520, 400, 333), (384, 384, 384), (1000, 300, 70), (130, 200, 2),
                                   (128, 128, 64), (129, 77, 65), (64, 1000, 96), (2100, 2100, 160)])
@pytest.mark.parametrize("fuse_min", [2, 4, 128])
def test_leg1_on_half_storage(ops, shape, fuse_min):
    M, K, L = shape
    csr = corner_csr(M, K, seed=M + L, hubs=min(K, 150))
    X = (np.random.default_rng(5).random((K, L)) ** 3).astype(np.float16)
    want = (dense64(csr) @ X.astype(np.float64)).T
    ops.set_tuning(fuse_min=fuse_min)
    try:
        g = ops.graph(csr)
        steps, cov, rem = ops.fused_stats(g)
        assert cov + rem == csr.nnz
        yt = ops.matrix(L, M, np.float16, blocked=True)
        ops.spmm(g, put_half(ops, X), yt, transpose_out=True)
        got = ops.download(yt)
        yt2 = ops.matrix(L, M, np.float16, blocked=True)
        ops.spmm(g, put_half(ops, X), yt2, transpose_out=True)
        assert np.array_equal(got, ops.download(yt2))            # reproducible
    finally:
        ops.set_tuning(fuse_min=2)
    close_to_rounded(got, want)


def test_matrix_core_part_is_exact_on_half_operands(ops):
    """One entry per row, every column shared: everything goes through v_mfma_f32_32x32x16_f16 with a 0/1
    pattern and must come back bit for bit (any fp16 value, subnormals and the largest included)."""
    M, K, L = 256, 64, 128
    csr = CSR(M, K, np.arange(M + 1, dtype=np.int32), (np.arange(M) % 64).astype(np.int32), np.ones(M))
    rng = np.random.default_rng(0)
    X = rng.integers(0, 0x7C00, size=(K, L)).astype(np.uint16).view(np.float16)     # every finite positive pattern
    X[1::2] = -X[1::2]
    g = ops.graph(csr)
    assert ops.fused_stats(g) == (2 * 4, M, 0)
    yt = ops.matrix(L, M, np.float16, blocked=True)
    ops.spmm(g, put_half(ops, X), yt, transpose_out=True)
    got, want = ops.download(yt), X[np.arange(M) % 64].T
    assert np.array_equal(got, want)                 # (values: -0 comes back as +0, 0 + (-0) in the accumulator)
    nz = want != 0
    assert np.array_equal(got.astype(np.float16).view(np.uint16)[nz], want.view(np.uint16)[nz])    # subnormals included


def reference_leg2(csr, Tt, coef, counts=None, prior=None, lbd=0.0):
    v = coef * (dense64(csr) @ Tt.astype(np.float64))
    if counts is not None:
        v = v * (1.0 - 0.5 ** counts.astype(np.float64))
    if prior is not None:
        v = (1.0 - lbd) * v + lbd * prior.astype(np.float64)
    np.fill_diagonal(v, 1.0)
    return v


@pytest.mark.parametrize("n", [64, 129, 200, 520, 1000, 2100])
@pytest.mark.parametrize("variant", ["plain", "evidence", "prior", "prior-scaled", "evidence-scaled"])
def test_leg2_on_half_storage(ops, n, variant):
    """Upper triangle + mirror with the fused epilogue: a symmetric product (Tt = W^T-ish built so that
    W . Tt is symmetric), symmetric counts and prior; the result must be exactly symmetric, within one
    rounding of float64, and the count must be the count of the stored values."""
    scale = 16384.0 if variant.endswith("-scaled") else 1.0      # stored = value x scale (the solver's 2^14)
    variant = variant.split("-")[0]
    csr = corner_csr(n, n, seed=n, hubs=min(n, 120))
    rng = np.random.default_rng(n)
    W = dense64(csr)
    Ssym = rng.random((n, n)) ** 4 / (4096.0 if scale > 1 else 1.0)
    Ssym = ((Ssym + Ssym.T) / 2).astype(np.float16)
    Tt = ((W @ Ssym.astype(np.float64)).T * scale).astype(np.float16).astype(np.float64) / scale   # as leg 1 stores it
    # W . Tt is symmetric only up to the rounding of Tt: the kernel computes the upper triangle and mirrors
    counts = prior = None
    lbd = 0.0
    if variant == "evidence":
        c = rng.integers(0, 6, size=(n, n))
        counts = np.triu(c) + np.triu(c, 1).T
        counts = counts.astype(np.uint8)
    if variant == "prior":
        pr = rng.random((n, n)).astype(np.float32)
        prior = ((pr + pr.T) / 2).astype(np.float32)
        lbd = 0.3
    want = reference_leg2(csr, Tt, 0.8, counts, prior, lbd)
    prev = (rng.random((n, n)) ** 4) / (4096.0 if scale > 1 else 1.0)
    prev = (((prev + prev.T) / 2) * scale).astype(np.float16).astype(np.float64) / scale
    g = ops.graph(csr)
    y = ops.matrix(n, n, np.float16, blocked=True)
    y.scale = scale
    put_half_s = lambda a: put_half(ops, a, scale)
    ep = dict(coef=0.8, previous=put_half_s(prev), eps=1e-3, set_diag=True, symmetric=True, lbd=lbd,
              evidence=None if counts is None else put_blocked(ops, counts, np.uint8),
              apriori=None if prior is None else put_blocked(ops, prior))
    ops.spmm(g, put_half_s(Tt), y, epilogue=ep)
    got = ops.download(y)
    changed = ops.read_changed()
    assert np.array_equal(got, got.T)
    iu = np.triu_indices(n)
    close_to_rounded(got[iu], want[iu], scale=scale)
    # the count: new value BEFORE rounding against the stored old one, eps widened by half the fp16 spacing at
    # the old value; bracketed, because the device's f32 sum is not the float64 one
    e16 = np.maximum(np.frexp(prev * scale)[1] + 14, 1)               # fp16 exponent field of the stored old value
    tol = 1e-3 + np.ldexp(1.0, e16 - 26) / scale
    upper = np.triu(want) + np.triu(want, 1).T       # (a < c is computed once and counted twice)
    move = np.abs(upper - prev)
    lo, hi = int((move > tol + 1e-5 * want + 1e-7).sum()), int((move > tol - 1e-5 * want - 1e-7).sum())
    assert lo <= changed <= hi, (lo, changed, hi)
    # the short-circuit form of the count: zero exactly when nothing moved
    ep["count_any"] = True
    ops.spmm(g, put_half_s(Tt), y, epilogue=ep)
    assert (ops.read_changed() > 0) == (changed > 0)
    ep["previous"] = put_half_s(got)
    ep["count_any"] = False
    ep["eps"] = 1e-6 / scale                 # own rounding never counts: |new - round(new)| <= half a spacing
    ops.spmm(g, put_half_s(Tt), y, epilogue=ep)
    assert ops.read_changed() == 0 and np.array_equal(ops.download(y), got)


def test_half_storage_needs_what_it_says(ops):
    from simrank_amd._lib import SimRankHipError
    csr = random_csr(200, 200, 0.05, seed=1)
    ops.set_tuning(fuse=0)
    try:
        g = ops.graph(csr)
    finally:
        ops.set_tuning(fuse=1)
    x = put_half(ops, np.zeros((200, 200)))
    y = ops.matrix(200, 200, np.float16, blocked=True)
    with pytest.raises(SimRankHipError, match="one-launch plan"):
        ops.spmm(g, x, y, transpose_out=True)
    g = ops.graph(csr)
    with pytest.raises(SimRankHipError, match="leg 1"):
        ops.spmm(g, x, y)                                       # neither transposed nor an epilogue
    with pytest.raises(SimRankHipError, match="column block of a sharded update"):
        ops.spmm(g, x, y, epilogue=dict(coef=0.8, symmetric=False, diag_col0=8))     # columns 8 .. 207 of 200
    with pytest.raises(SimRankHipError, match="no prior"):
        ops.spmm(g, x, y, epilogue=dict(coef=0.8, symmetric=False, apriori=put_blocked(ops, np.zeros((200, 200), np.float32))))


@pytest.mark.parametrize("n,col0,L", [(520, 128, 192), (1000, 0, 64), (1000, 936, 64), (2100, 640, 130), (200, 64, 136)])
@pytest.mark.parametrize("variant", ["plain", "evidence-scaled"])
def test_leg2_full_form_on_a_column_block(ops, n, col0, L, variant):
    """Leg 2 of one rank of a SHARDED update on fp16-held matrices (csrc/shardplan.hip, storage_fp16): the column block
    [col0, col0 + L) of S' — every element computed once (no triangle, no mirror image), the diagonal where
    row == col0 + column, each moved element counted once."""
    scale = 16384.0 if variant.endswith("-scaled") else 1.0
    csr = corner_csr(n, n, seed=n + L, hubs=min(n, 120))
    rng = np.random.default_rng(n + col0)
    W = dense64(csr)
    Tt = (rng.random((n, L)) ** 4 / (4096.0 if scale > 1 else 1.0) * scale).astype(np.float16).astype(np.float64) / scale
    counts = rng.integers(0, 6, size=(n, L)).astype(np.uint8) if variant.startswith("evidence") else None
    want = 0.8 * (W @ Tt)
    if counts is not None:
        want = want * (1.0 - 0.5 ** counts.astype(np.float64))
    want[np.arange(col0, col0 + L), np.arange(L)] = 1.0
    prev = (rng.random((n, L)) ** 4 / (4096.0 if scale > 1 else 1.0) * scale).astype(np.float16).astype(np.float64) / scale
    g = ops.graph(csr)
    y = ops.matrix(n, L, np.float16, blocked=True)
    y.scale = scale
    ep = dict(coef=0.8, previous=put_half(ops, prev, scale), eps=1e-3, set_diag=True, symmetric=False, diag_col0=col0,
              evidence=None if counts is None else put_blocked(ops, counts, np.uint8))
    ops.spmm(g, put_half(ops, Tt, scale), y, epilogue=ep)
    got = ops.download(y)
    changed = ops.read_changed()
    close_to_rounded(got, want, scale=scale)
    assert np.all(got[np.arange(col0, col0 + L), np.arange(L)] == 1.0)
    e16 = np.where(prev == 0, 1, np.maximum(np.frexp(prev * scale)[1] + 14, 1))      # (frexp(0) has exponent 0)
    tol = 1e-3 + np.ldexp(1.0, e16 - 26) / scale
    move = np.abs(want - prev)
    # (bracketed: the device's f32 sums of up to a few hundred terms are not the float64 ones)
    lo, hi = int((move > tol + 4e-5 * want + 1e-7).sum()), int((move > tol - 4e-5 * want - 1e-7).sum())
    assert lo <= changed <= hi, (lo, changed, hi)
    ep["previous"] = put_half(ops, got, scale)
    ep["eps"] = 1e-6 / scale
    ops.spmm(g, put_half(ops, Tt, scale), y, epilogue=ep)
    assert ops.read_changed() == 0 and np.array_equal(ops.download(y), got)


# ------------------------------------------------------------------------------------------------
# whole fits
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cls", ["SimRank", "SimRankPP", "AprioriSimRank"])
def test_fit_with_half_storage_against_the_oracle(cls):
    """N = 2048 power-law, whole fits against the float64 oracle.
    (1) Exactly 10 updates on both sides: the arithmetic error — two roundings to fp16 per update, damped by
    the contraction — measured 1.2e-3 max / 1.8e-4 median relative (profiles/r03_half_errors.log); bars at
    about twice that.  Exactly symmetric, diagonal exactly 1.
    (2) To eps = 1e-4: the loop may end LATER than the reference's (never earlier by more than one update):
    values above 1/8 are stored with a spacing above eps, and when one of them crosses a rounding boundary
    its dependents move by more than eps.  What comes back is then closer to the fixed point than the
    reference's iterate, within the distance the reference still had to go: eps . C / (1 - C)."""
    import simrank_amd.SimRank as SRA
    from oracle import simrank_oracle as O
    from simrank_amd import synth
    df = synth.powerlaw_directed(2048, 24, seed=12)
    args, okw = (), {}
    if cls == "AprioriSimRank":
        rng = np.random.default_rng(0)
        prior = rng.random((2048, 2048))
        prior = (prior + prior.T) / 2
        args, okw = (prior,), dict(apriori=prior, lbd=0.2)
    kw = dict(lbd=0.2) if cls == "AprioriSimRank" else {}
    oracle = O.fit_simrank if cls == "SimRank" else O.fit_simrank_pp
    want = oracle(df, verbose=False, iterations=10, eps=1e-30, **okw)
    got = getattr(SRA, cls)().fit(df, *args, verbose=False, storage_precision="fp16", iterations=10, eps=1e-30, **kw)
    assert list(got.index) == want["labels"]
    a, b = got.values, want["S"]
    assert np.array_equal(a, a.T) and np.all(np.diag(a) == 1.0)
    err = np.abs(a - b)
    rel = err[b > 0] / b[b > 0]
    assert err.max() < 6e-4 and rel.max() < 2.5e-3 and np.median(rel) < 4e-4, (err.max(), rel.max(), np.median(rel))
    want = oracle(df, verbose=False, **okw)
    est = getattr(SRA, cls)()
    got = est.fit(df, *args, verbose=False, storage_precision="fp16", **kw)
    assert est.converged_at is not None and est.converged_at >= want["k"] - 1
    assert np.abs(got.values - want["S"]).max() < 1e-4 * 0.8 / 0.2 + 6e-4


@pytest.mark.parametrize("cls", ["SimRank", "SimRankPP"])
def test_fit_with_half_storage_on_virtual_ranks(cls):
    """BASELINE config 5 in its stated form through the reference's class surface: `fit(storage_precision="fp16",
    world=LocalWorld(4))` runs the sharded loop behind the C ABI on fp16-held matrices (simrank_amd/cshard.py) — the
    bars of the single-rank mode against the float64 oracle, the dense and the top-k hand-back, the console text of
    the reference's loop."""
    import io
    from contextlib import redirect_stdout
    import simrank_amd.SimRank as SRA
    from oracle import simrank_oracle as O
    from simrank_amd import synth
    from tests.pydriver import LocalWorld
    df = synth.powerlaw_directed(2048, 24, seed=12)
    oracle = O.fit_simrank if cls == "SimRank" else O.fit_simrank_pp
    want = oracle(df, verbose=False, iterations=10, eps=1e-30)
    est = getattr(SRA, cls)()
    got = est.fit(df, verbose=False, storage_precision="fp16", iterations=10, eps=1e-30, world=LocalWorld(4))
    assert list(got.index) == want["labels"] and est.engine_mode == "sparse"
    a, b = got.values, want["S"]
    assert np.all(np.diag(a) == 1.0)
    err = np.abs(a - b)
    rel = err[b > 0] / b[b > 0]
    assert err.max() < 6e-4 and rel.max() < 2.5e-3 and np.median(rel) < 4e-4, (err.max(), rel.max(), np.median(rel))
    # to eps, with the reference's console text; top-k hand-back
    want = oracle(df, verbose=False)
    est = getattr(SRA, cls)()
    buf = io.StringIO()
    with redirect_stdout(buf):
        top = est.fit(df, verbose=True, storage_precision="fp16", world=LocalWorld(8), top_k=5)
    assert est.converged_at is not None and est.converged_at >= want["k"] - 1
    assert f"Converged at iteration {est.converged_at}" in buf.getvalue() and "Start iterating..." in buf.getvalue()
    assert len(top) == 5 * 2048 and set(top.columns) == {"node", "rank", "neighbor", "similarity"}
    full = getattr(SRA, cls)().fit(df, verbose=False, storage_precision="fp16", world=LocalWorld(8))
    first = top[top["rank"] == 1].set_index("node")
    for node in list(full.index[:50]):
        row = full.loc[node].drop(node)
        assert first.loc[node, "similarity"] == np.float32(row.max())
    if cls == "SimRankPP":
        assert est.Evidence.shape == (2048, 2048)          # (lazy attribute: from the CSR on the host in this mode)
    with pytest.raises(ValueError, match="multiple of"):
        getattr(SRA, cls)().fit(synth.powerlaw_directed(300, 5, seed=1), verbose=False, storage_precision="fp16",
                                world=LocalWorld(2))


def test_half_storage_is_refused_where_it_does_not_exist():
    import simrank_amd.SimRank as SRA
    from simrank_amd import synth
    df = synth.er_directed(300, 0.02, seed=5)
    with pytest.raises(ValueError, match="storage_precision"):
        SRA.SimRank().fit(df, verbose=False, storage_precision="bf16")
    with pytest.raises(ValueError, match="storage_precision='fp16' needs"):
        SRA.SimRank().fit(df, verbose=False, storage_precision="fp16", mode="dense")
    n = len(SRA.SimRank().fit(df, verbose=False, iterations=1))
    big = np.full((n, n), 5.0)
    with pytest.raises(ValueError, match="prior values below 4"):
        SRA.AprioriSimRank().fit(df, big, verbose=False, storage_precision="fp16")


def test_plan_api_with_half_storage(ops):
    """The C-level loop (simrank_plan_*) with `storage_fp16`: the same updates as the Python solver in that
    mode (same kernels; the two order equal-length nodes differently, so sums may round differently: a
    couple of fp16 spacings), prior and evidence included; hand-back through the f32 layout in the
    caller's order."""
    from simrank_amd import ingest, synth
    from tests.pydriver import LocalWorld, SideSpec, Solver
    from simrank_amd.engine import Plan
    df = synth.powerlaw_directed(1500, 12, seed=7)
    _, csr = ingest.directed(df, False, "from", "to", "weight")
    rng = np.random.default_rng(3)
    A = rng.random((csr.n_rows, csr.n_rows)).astype(np.float32)
    A = (A + A.T) / 2
    ops.set_tuning(fuse_steps=8, fuse_min=3)
    try:
        plan = Plan(ops, csr, coef=0.8, evidence=True, apriori=A, lbd=0.25, storage="fp16")
        plan.reset()
        counts = [plan.step(1e-4, exact_count=True) for _ in range(5)]
        got = plan.result()
        plan.free()
        s = Solver(lambda r: ops, LocalWorld(1),
                   [SideSpec(csr, csr.rowscale, 0.8, evidence_from=csr, apriori=A, lbd=0.25, storage="fp16")], "sparse")
        s.exact_count = True
        s.reset()
        want_counts = [s.step(1e-4) for _ in range(5)]
        want = s.result(0)
        s.release()
    finally:
        ops.set_tuning(fuse_steps=1, fuse_min=2)
    assert all(abs(c - w) <= 0.01 * w + 8 for c, w in zip(counts, want_counts)), (counts, want_counts)
    np.testing.assert_allclose(got, want, rtol=4 * HALF_ULP, atol=1e-7)
    assert np.array_equal(got, got.T) and np.all(np.diag(got) == 1.0)


def test_bipartite_fit_with_half_storage():
    """BipartiteSimRankPP (SimRank.py:410-424, Gauss-Seidel order): rectangular legs on fp16-held matrices,
    ten updates on both sides against the oracle; a tiny graph (where mode "auto" would pick the dense legs)
    runs the gather legs instead of being refused."""
    import simrank_amd.SimRank as SRA
    from oracle import simrank_oracle as O
    from tests.graphs import bipartite_random
    df = bipartite_random(900, 500, 0.03, seed=14)
    want = O.fit_bipartite_pp(df, verbose=False, strict_reference=False, iterations=10, eps=1e-30)
    # (round 6: the two-matrix plan behind the C ABI is f32; fp16-held rectangular legs run through the tests' Python
    # choreography — a world of tests/pydriver.py — and fit() on its own refuses the combination with the reason)
    from tests.pydriver import LocalWorld
    with pytest.raises(ValueError, match="no C-level plan"):
        SRA.BipartiteSimRankPP().fit(df, verbose=False, strict_reference=False, iterations=1, storage_precision="fp16")
    s1, s2 = SRA.BipartiteSimRankPP().fit(df, verbose=False, strict_reference=False, iterations=10, eps=1e-30,
                                          storage_precision="fp16", world=LocalWorld(1))
    for got, ref in ((s1.values, want["S1"]), (s2.values, want["S2"])):
        assert np.array_equal(got, got.T) and np.all(np.diag(got) == 1.0)
        err = np.abs(got - ref)
        rel = err[ref > 0] / ref[ref > 0]
        assert err.max() < 6e-4 and rel.max() < 3e-3 and np.median(rel) < 4e-4, (err.max(), rel.max(), np.median(rel))
    small = bipartite_random(40, 30, 0.2, seed=2)
    a, b = SRA.BipartiteSimRank().fit(small, verbose=False, storage_precision="fp16", world=LocalWorld(1))
    wa = O.fit_bipartite(small, verbose=False)
    assert np.abs(a.values - wa["S1"]).max() < 1e-3 and np.abs(b.values - wa["S2"]).max() < 1e-3


@pytest.mark.parametrize("n", [2, 3, 7, 33, 63, 64, 65, 127, 128, 129, 257, 700])
def test_odd_sizes_with_half_storage(n):
    """Node counts around every grid the kernel has (8-row pieces, 64-column panels, 128-row blocks): five
    updates of SimRank++ on fp16-held matrices against the f32 run of the same build."""
    import pandas as pd
    import simrank_amd.SimRank as SRA
    rng = np.random.default_rng(n)
    m = max(2, int(n * min(n - 1, 6)))
    src, dst = rng.integers(0, n, size=m), rng.integers(0, n, size=m)
    ring = np.arange(n)
    df = pd.DataFrame({"from": np.concatenate([src, ring]), "to": np.concatenate([dst, (ring + 1) % n]),
                       "weight": 1.0}).drop_duplicates(["from", "to"])
    df = df[df["from"] != df["to"]].reset_index(drop=True)
    kw = dict(verbose=False, iterations=5, eps=1e-30, weighted=False)
    want = SRA.SimRankPP().fit(df, **kw)
    got = SRA.SimRankPP().fit(df, storage_precision="fp16", **kw)
    assert list(got.index) == list(want.index)
    a, b = got.values, want.values
    assert np.array_equal(a, a.T) and np.all(np.diag(a) == 1.0)
    np.testing.assert_allclose(a, b, rtol=3e-3, atol=2e-8)
    assert np.array_equal(a == 0, b == 0) or np.abs(a - b).max() < 1e-7      # same support (above fp16's floor)


def test_graph_without_entries_on_half_storage(ops):
    """No edges at all (found by tools/soak_kernels.py): the one-launch plan exists for such a graph too, leg 1
    gives zeros and leg 2 the identity (and the prior, where there is one)."""
    n = 70
    csr = CSR(n, n, np.zeros(n + 1, np.int32), np.empty(0, np.int32), np.ones(n))
    g = ops.graph(csr)
    x = put_half(ops, np.random.default_rng(0).random((n, n)))
    yt = ops.matrix(n, n, np.float16, blocked=True)
    ops.spmm(g, x, yt, transpose_out=True)
    assert not ops.download(yt).any()
    y = ops.matrix(n, n, np.float16, blocked=True)
    ops.spmm(g, yt, y, epilogue=dict(coef=0.8, previous=x, eps=1e-4, set_diag=True, symmetric=True))
    assert np.array_equal(ops.download(y), np.eye(n, dtype=np.float32))
    assert ops.read_changed() > 0


# ---- the fp16 WIRE format of the sharded exchanges (f32 kernels, fp16 on the links) ----

def test_wire_narrow_widen_bits(ops):
    """simrank_narrow_h16 / simrank_widen_h16: value x 2^14, nearest even, saturating at the largest finite fp16;
    bit-equal to NumPy's float16 conversion of the same scaled floats — at aligned and unaligned offsets, with
    tails shorter than a vector.  (Raw device pointers: torch's own HIP runtime is not brought up inside this
    process — the torch-tensor wrappers HipOps.narrow_t / widen_t make the same two calls.)"""
    import ctypes as C
    from simrank_amd._lib import check
    lib, st = ops.lib, ops.stream
    rng = np.random.default_rng(5)
    n = 100_003
    a = (rng.random(n) ** 6).astype(np.float32)            # most values far below 1, as similarities are
    a[:8] = [0.0, 1.0, 3.99, 4.5, 2.0 ** -39, 2.0 ** -40, 65504.0 / 16384 + 1e-3, 1e30]
    scaled = np.clip(a.astype(np.float64) * 16384.0, -65504.0, 65504.0).astype(np.float32)
    want = scaled.astype(np.float16)
    assert np.isfinite(want.astype(np.float32)).all()       # saturated, never an infinity
    src, dst, back = ops._malloc(4 * n), ops._malloc(2 * n), ops._malloc(4 * n)
    scale = C.c_float(16384.0)
    try:
        check(lib.simrank_memcpy_h2d(src, a.ctypes.data, a.nbytes, st), "h2d")
        for off, cnt in ((0, n), (3, 77), (8, 4096), (5, n - 5), (16, 7)):
            check(lib.simrank_memset(C.c_void_p(dst), 0, 2 * n, st), "memset")
            check(lib.simrank_narrow_h16(C.c_void_p(src + 4 * off), C.c_void_p(dst + 2 * off), cnt, scale, st), "narrow")
            got = np.empty(n, dtype=np.uint16)
            check(lib.simrank_memcpy_d2h(got.ctypes.data, dst, got.nbytes, st), "d2h")
            assert np.array_equal(got[off:off + cnt], want[off:off + cnt].view(np.uint16)), (off, cnt)
            assert not got[:off].any() and not got[off + cnt:].any()
            check(lib.simrank_memset(C.c_void_p(back), 0xFF, 4 * n, st), "memset")
            check(lib.simrank_widen_h16(C.c_void_p(dst + 2 * off), C.c_void_p(back + 4 * off), cnt, scale, st), "widen")
            b = np.empty(n, dtype=np.float32)
            check(lib.simrank_memcpy_d2h(b.ctypes.data, back, b.nbytes, st), "d2h")
            assert np.array_equal(b[off:off + cnt], want[off:off + cnt].astype(np.float32) / np.float32(16384.0))
            assert (b[:off].view(np.uint32) == 0xFFFFFFFF).all() and (b[off + cnt:].view(np.uint32) == 0xFFFFFFFF).all()
    finally:
        for ptr in (src, dst, back):
            ops._free(ptr)


def test_wire_refuses_bad_scale_and_alignment(ops):
    import ctypes as C
    lib, st = ops.lib, ops.stream
    src, dst = ops._malloc(256), ops._malloc(128)
    try:
        rc = lib.simrank_narrow_h16(C.c_void_p(src), C.c_void_p(dst), 64, C.c_float(3.0), st)
        assert rc != 0 and b"power of two" in lib.simrank_last_error()
        rc = lib.simrank_widen_h16(C.c_void_p(dst + 1), C.c_void_p(src), 8, C.c_float(16384.0), st)
        assert rc != 0 and b"misaligned" in lib.simrank_last_error()
        assert lib.simrank_narrow_h16(None, None, 0, C.c_float(16384.0), st) == 0
    finally:
        ops._free(src)
        ops._free(dst)


@pytest.mark.parametrize("cls,half", [("SimRank", True), ("SimRank", False), ("SimRankPP", True)])
def test_fp16_wire_on_virtual_ranks(cls, half):
    """LocalWorld(P, exchange_precision="fp16") rounds what the ranks hand each other exactly as TorchWorld's fp16
    wire does (tests/test_distributed_gloo.py pins the two against each other on CPU): against the f32 wire the
    result moves by a few fp16 roundings per update and no more."""
    import simrank_amd.SimRank as SRA
    from simrank_amd import synth
    from tests.pydriver import LocalWorld
    df = synth.powerlaw_directed(1024, 8, seed=4)
    kw = dict(weighted=True) if cls.endswith("PP") else {}
    exact = getattr(SRA, cls)().fit(df, iterations=6, eps=0, verbose=False, mode="sparse",
                                    world=LocalWorld(4, symmetric_shards=half), **kw)
    wire = getattr(SRA, cls)().fit(df, iterations=6, eps=0, verbose=False, mode="sparse",
                                   world=LocalWorld(4, symmetric_shards=half, exchange_precision="fp16"), **kw)
    assert list(wire.index) == list(exact.index)
    big = exact.values > 1e-6
    rel = np.abs(wire.values - exact.values)[big] / exact.values[big]
    assert 1e-7 < rel.max() < 4e-3, rel.max()
    assert np.abs(wire.values - exact.values).max() < 1e-3
    assert np.array_equal(np.diag(wire.values), np.diag(exact.values))
