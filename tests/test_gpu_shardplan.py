"""K10 behind the C ABI on a real MI355X: `simrank_shardplan_*` (csrc/shardplan.hip) — the loop of SimRank.py:129-140,
:351-362, :443-454 with S split by column block over `world` ranks.  On the one GPU of the box the ranks are an
IN-PROCESS GROUP (exchanges = device copies) for every world size, and ONE rank of a real RCCL world (own process, no
torch) for the RCCL calls themselves; two GPUs, where visible, run a real two-rank world.  Checked against the golden
vectors of the reference, the single-rank plan, and — bit for bit in the full form — the Python driver's sharded path,
which makes the same launches over torch.distributed."""
import os
import subprocess
import sys

import numpy as np
import pytest

import simrank_amd.SimRank as SRA
from oracle import simrank_oracle as O
from simrank_amd import ingest, synth
from tests.pydriver import LocalWorld
from tests.conftest import Golden
from tests.helpers import RTOL, assert_close

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ops():
    from simrank_amd.engine import HipOps
    return HipOps(0)


def golden_csr(g):
    G = g.out["G"]
    n = len(G)
    rows, cols = np.nonzero(G)
    rowptr = np.zeros(n + 1, np.int32)
    np.cumsum(np.bincount(rows, minlength=n), out=rowptr[1:])
    scale = np.zeros(n)
    scale[rows] = G[rows, cols]
    return ingest.CSR(n, n, rowptr, cols.astype(np.int32), scale)


def assembled(sp):
    """The whole matrix from the ranks' blocks (what a host program of a multi-process world would do)."""
    out = np.full((sp.n, sp.n), np.nan)
    for i in range(len(sp.plans)):
        blk, ids = sp.block(i)
        out[:, ids] = blk
    return out


@pytest.mark.parametrize("world,form,stages", [(1, 0, 1), (2, 0, 1), (3, 0, 2), (4, 0, 1), (4, 1, 1), (2, 1, 3), (5, 0, 1)])
@pytest.mark.parametrize("name", ["SimRank_er128", "SimRankPP_er128", "SimRank_toy5", "SimRankPP_bts300"])
def test_shardplan_reproduces_the_golden_vectors(ops, name, world, form, stages):
    """create -> run -> result on `world` virtual ranks against the vectors generated from the reference: S to 1e-5,
    the "Converged at iteration k" index exactly; uneven and EMPTY blocks (toy5 on 3..5 ranks), both forms of leg 2,
    exchange 1 in stages."""
    from simrank_amd.engine import ShardPlans
    g = Golden(name)
    csr = golden_csr(g)
    if form == 1 and csr.n_rows % (32 * world):
        pytest.skip("the half form needs n % (32 x ranks) == 0")
    sp = ShardPlans(ops, csr, world=world, coef=g.kwargs.get("C", 0.8), evidence=name.startswith("SimRankPP"),
                    leg2_form=form, stages=stages)
    assert sp.info()["half_form"] == bool(form) and sp.info()["stages"] == stages
    done, conv = sp.run(g.kwargs.get("iterations", 100), g.kwargs.get("eps", 1e-4))
    assert (conv if conv is not None else -1) == (g.k if g.k is not None else -1)
    full = sp.result()
    assert_close(full, g.out["S"])
    assert np.array_equal(full, assembled(sp))                   # root's gather == the blocks put together
    # step by step with the exact count: the passing test is the first zero count
    sp.reset()
    eps = g.kwargs.get("eps", 1e-4)
    counts = [sp.step(eps, exact_count=True) for _ in range(done)]
    if conv is not None and done:
        assert counts[-1] == 0 and all(c > 0 for c in counts[:-1])
    assert_close(sp.result(), g.out["S"])
    assert sp.run(0, 1e-4) == (0, None)
    np.testing.assert_array_equal(sp.result(), np.eye(csr.n_rows))
    assert sp.run(5, 1.0) == (0, 0)
    sp.free()


@pytest.mark.parametrize("pp", [False, True])
def test_shardplan_full_form_is_the_python_drivers_bits(ops, pp):
    """The C choreography makes the launches driver.Side makes (same graph, node order, chunk layout, pads): in the
    full form, unstaged, the result is BIT-EQUAL to `fit(world=LocalWorld(P, symmetric_shards=False))` — and so to a
    single rank's full form; the half form agrees with the driver's half form bit for bit as well (same dealt order),
    and with one rank to rounding."""
    from simrank_amd.engine import ShardPlans
    df = synth.powerlaw_directed(2048, 12, seed=8)
    _, csr = ingest.directed(df, False, "from", "to", "weight")
    cls = SRA.SimRankPP if pp else SRA.SimRank
    scale = ingest.spread(csr) * csr.rowscale if pp else csr.rowscale
    for world in (2, 4, 8):
        want = cls().fit(df, iterations=5, eps=0, verbose=False, mode="sparse", world=LocalWorld(world, symmetric_shards=False))
        sp = ShardPlans(ops, csr, rowscale=scale, world=world, evidence=pp, leg2_form=0, stages=1)
        assert sp.run(5, 0.0) == (5, None)
        got = sp.result()
        sp.free()
        assert np.array_equal(got, want.values), world
        want_half = cls().fit(df, iterations=5, eps=0, verbose=False, mode="sparse", world=LocalWorld(world))
        sp = ShardPlans(ops, csr, rowscale=scale, world=world, evidence=pp, leg2_form=1, stages=1)
        assert sp.info()["half_form"]
        sp.run(5, 0.0)
        half = sp.result()
        sp.free()
        assert np.array_equal(half, want_half.values), world
        np.testing.assert_allclose(half, got, rtol=2e-6, atol=1e-30)
        # exchange 1 in stages: whole panels per slice, the same plan: same values
        sp = ShardPlans(ops, csr, rowscale=scale, world=world, evidence=pp, leg2_form=0, stages=3)
        sp.run(5, 0.0)
        staged = sp.result()
        sp.free()
        np.testing.assert_allclose(staged, got, rtol=2e-6, atol=1e-30)
        # the half form with leg 2 (and the exchange of the mirrored tiles) in stages of column tiles: the same tiles, the
        # same sums — the unstaged half form's bits whatever exchange 1's stages did to the operand's order of arrival
        for stages in (2, 4):
            sp = ShardPlans(ops, csr, rowscale=scale, world=world, evidence=pp, leg2_form=1, stages=stages)
            counts = [sp.step(0.0, exact_count=True) for _ in range(5)]
            staged_half = sp.result()
            sp.free()
            assert all(c > 0 for c in counts)
            if 2048 // (32 * world) >= 2 * 2:                       # (at least two column tiles per stage: really staged)
                assert np.array_equal(staged_half, half), (world, stages)
            else:
                np.testing.assert_allclose(staged_half, half, rtol=2e-6, atol=1e-30)


def test_shardplan_with_a_prior_against_the_oracle(ops):
    """AprioriSimRank's loop (SimRank.py:443-454) on four virtual ranks: symmetric prior, evidence, nodes dealt inside,
    results in the caller's order.  A prior that is NOT symmetric (asymmetric iterates: leg 2's product goes round a second
    all-to-all, the epilogue is a pass of its own) on 1 - 5 ranks, staged or not, uneven and empty blocks included, also on
    the fp16 wire; the half form refuses it."""
    from simrank_amd._lib import SimRankHipError
    from simrank_amd.engine import ShardPlans
    df = synth.powerlaw_directed(640, 9, seed=5)
    _, csr = ingest.directed(df, False, "from", "to", "weight")
    rng = np.random.default_rng(2)
    A = rng.random((csr.n_rows, csr.n_rows))
    A = (A + A.T) / 2
    G = csr.dense()
    want_S, want_k = O.iterate_directed(G, C=0.7, iterations=100, eps=1e-4, E=O.evidence(G),
                                        apriori=A.astype(np.float32).astype(np.float64), lbd=0.3)
    for form in (0, 1):
        sp = ShardPlans(ops, csr, world=4, coef=0.7, evidence=True, apriori=A, lbd=0.3, leg2_form=form)
        done, conv = sp.run(100, 1e-4)
        assert conv == want_k
        assert_close(sp.result(), want_S)
        sp.free()
    A[3, 5] += 0.5
    A[10:40, 200:260] *= 0.25
    want_S, want_k = O.iterate_directed(G, C=0.7, iterations=100, eps=1e-4, E=O.evidence(G),
                                        apriori=A.astype(np.float32).astype(np.float64), lbd=0.3)
    assert not np.array_equal(want_S, want_S.T)
    for world, stages in ((1, 1), (2, 1), (3, 2), (4, 1), (5, 3)):
        sp = ShardPlans(ops, csr, world=world, coef=0.7, evidence=True, apriori=A, lbd=0.3, leg2_form=-1, stages=stages)
        done, conv = sp.run(100, 1e-4)
        assert conv == want_k, (world, conv, want_k)
        assert_close(sp.result(), want_S)
        idx, val = sp.topk(4)
        got = sp.result()
        for a in (0, 5, 639):
            cand = np.array([c for c in range(csr.n_rows) if c != a])
            np.testing.assert_array_equal(val[a].astype(np.float64), got[a, cand[np.lexsort((cand, -got[a, cand]))][:4]])
        sp.free()
    # a graph smaller than the world: ranks without columns
    small = synth.er_directed(3, 0.9, seed=1)
    _, c3 = ingest.directed(small, False, "from", "to", "weight")
    A3 = np.random.default_rng(7).random((c3.n_rows, c3.n_rows))
    w3, k3 = O.iterate_directed(c3.dense(), C=0.8, iterations=20, eps=1e-4, E=O.evidence(c3.dense()),
                                apriori=A3.astype(np.float32).astype(np.float64), lbd=0.4)
    sp = ShardPlans(ops, c3, world=5, evidence=True, apriori=A3, lbd=0.4)
    done, conv = sp.run(20, 1e-4)
    assert conv == k3
    assert_close(sp.result(), w3)
    sp.free()
    with pytest.raises(SimRankHipError, match="not symmetric"):
        ShardPlans(ops, csr, world=4, apriori=A, lbd=0.3, leg2_form=1)
    # the fp16 wire under asymmetric iterates: both exchanges narrowed, one fp16 rounding of each product per update
    sp = ShardPlans(ops, csr, world=3, coef=0.7, evidence=True, apriori=A, lbd=0.3, wire_fp16=True, stages=2)
    sp.run(6, 0.0)
    wired = sp.result()
    sp.free()
    sp = ShardPlans(ops, csr, world=3, coef=0.7, evidence=True, apriori=A, lbd=0.3, stages=2)
    sp.run(6, 0.0)
    exact = sp.result()
    sp.free()
    assert not np.array_equal(wired, exact)
    big = exact > 1e-4
    assert (np.abs(wired - exact)[big] / exact[big]).max() < 5e-3 and np.abs(wired - exact).max() < 1e-3


def test_shardplan_fp16_wire(ops):
    """options.wire_fp16: the exchanges move fp16 x 2^14 — the same roundings as the Python driver's
    `exchange_precision="fp16"` (bit-equal to it in the full form), a few fp16 spacings from the f32 wire."""
    from simrank_amd.engine import ShardPlans
    df = synth.powerlaw_directed(1024, 8, seed=4)
    _, csr = ingest.directed(df, False, "from", "to", "weight")
    exact = SRA.SimRank().fit(df, iterations=6, eps=0, verbose=False, mode="sparse",
                              world=LocalWorld(4, symmetric_shards=False))
    emu = SRA.SimRank().fit(df, iterations=6, eps=0, verbose=False, mode="sparse",
                            world=LocalWorld(4, symmetric_shards=False, exchange_precision="fp16"))
    sp = ShardPlans(ops, csr, world=4, leg2_form=0, stages=1, wire_fp16=True)
    sp.run(6, 0.0)
    got = sp.result()
    sp.free()
    assert np.array_equal(got, emu.values)
    big = exact.values > 1e-6
    rel = np.abs(got - exact.values)[big] / exact.values[big]
    assert 1e-7 < rel.max() < 4e-3
    sp = ShardPlans(ops, csr, world=4, leg2_form=1, stages=2, wire_fp16=True)
    sp.run(6, 0.0)
    half = sp.result()
    sp.free()
    rel = np.abs(half - exact.values)[big] / exact.values[big]
    assert 1e-7 < rel.max() < 4e-3


@pytest.mark.parametrize("pp", [False, True])
@pytest.mark.parametrize("world,stages", [(1, 1), (2, 1), (4, 3), (8, 2)])
def test_shardplan_with_half_storage_against_the_oracle(ops, pp, world, stages):
    """options.storage_fp16: BASELINE config 5 in its stated form — reduced precision AND shards.  Every rank holds its
    block, the transposed product and the leg-2 operand in fp16 (half.hip on a column block: leg 2 in its full form), the
    exchange moves the fp16 panels themselves.  Against the float64 oracle with the bars of the single-rank fp16 tests
    (tests/test_gpu_half.py: two roundings per update, damped by the contraction), and against the single-rank fp16 plan
    to a few fp16 spacings (same kernels; the triangle + mirror form there, the full form here)."""
    from simrank_amd.engine import Plan, ShardPlans
    df = synth.powerlaw_directed(2048, 24, seed=12)
    _, csr = ingest.directed(df, False, "from", "to", "weight")
    scale = ingest.spread(csr) * csr.rowscale if pp else csr.rowscale
    want = (O.fit_simrank_pp if pp else O.fit_simrank)(df, verbose=False, iterations=10, eps=1e-30)
    sp = ShardPlans(ops, csr, rowscale=scale, world=world, evidence=pp, storage="fp16", stages=stages, leg2_form=0)
    assert sp.run(10, 1e-30) == (10, None)
    a, b = sp.result(), want["S"]
    assert np.array_equal(a, assembled(sp))
    sp.free()
    assert np.all(np.diag(a) == 1.0)
    err = np.abs(a - b)
    rel = err[b > 0] / b[b > 0]
    assert err.max() < 6e-4 and rel.max() < 2.5e-3 and np.median(rel) < 4e-4, (err.max(), rel.max(), np.median(rel))
    one = Plan(ops, csr, rowscale=scale, evidence=pp, storage="fp16")
    one.run(10, 1e-30)
    np.testing.assert_allclose(a, one.result(), rtol=8 * 2.0 ** -11, atol=1e-7)
    one.free()
    # to eps = 1e-4: may end later than the reference, never earlier by more than one update (tests/test_gpu_half.py)
    want = (O.fit_simrank_pp if pp else O.fit_simrank)(df, verbose=False)
    sp = ShardPlans(ops, csr, rowscale=scale, world=world, evidence=pp, storage="fp16", stages=stages)
    done, conv = sp.run(100, 1e-4)
    assert conv is not None and conv >= want["k"] - 1
    assert np.abs(sp.result() - want["S"]).max() < 1e-4 * 0.8 / 0.2 + 6e-4
    sp.free()


@pytest.mark.parametrize("storage,world,form", [("f32", 3, 0), ("f32", 4, 1), ("fp16", 2, 0), ("f32", 1, 0)])
def test_shardplan_topk_in_the_callers_ids(ops, storage, world, form):
    """simrank_shardplan_topk: every rank selects among its own columns, root merges — ids and tie order the caller's,
    against a sort of the dense result the same plans hand back."""
    from simrank_amd.engine import ShardPlans
    n_want = 640 if storage == "fp16" or form else 600
    df = synth.powerlaw_directed(n_want, 7, seed=9)
    _, csr = ingest.directed(df, False, "from", "to", "weight")
    if (storage == "fp16" and csr.n_rows % (64 * world)) or (form and csr.n_rows % (32 * world)):
        pytest.skip("size does not fit the mode")
    sp = ShardPlans(ops, csr, world=world, coef=0.8, evidence=True, leg2_form=form, storage=storage)
    sp.run(6, 1e-30)
    S = sp.result()
    idx, val = sp.topk(7)
    big, _ = sp.topk(5, exclude_diag=False)
    sp.free()
    n = csr.n_rows
    for a in range(n):
        cand = np.array([c for c in range(n) if c != a])
        order = cand[np.lexsort((cand, -S[a, cand]))][:7]
        assert list(idx[a]) == list(order), a
        np.testing.assert_array_equal(val[a].astype(np.float64), S[a, order])
        assert big[a][0] == a                       # the node itself (similarity 1) when the diagonal is not excluded


def test_shardplan_refuses_what_it_cannot_run(ops):
    from simrank_amd._lib import SimRankHipError
    from simrank_amd.engine import ShardPlans
    import ctypes as C
    df = synth.powerlaw_directed(100, 5, seed=1)
    _, csr = ingest.directed(df, False, "from", "to", "weight")
    with pytest.raises(SimRankHipError, match="multiple of 32"):
        ShardPlans(ops, csr, world=2, leg2_form=1)
    with pytest.raises(SimRankHipError, match="storage_fp16 on shards"):
        ShardPlans(ops, csr, world=2, storage="fp16")                       # 100 nodes: not whole 64-column panels per rank
    sp = ShardPlans(ops, csr, world=3)
    one = (C.c_void_p * 1)(sp.plans[0].value)
    assert ops.lib.simrank_shardplan_step(one, 1, 0.0, 1, None) != 0        # a group advances all its plans together
    assert b"in-process group" in ops.lib.simrank_last_error()
    sp.free()
    bad = ingest.CSR(csr.n_rows, csr.n_cols, csr.rowptr[::-1].copy(), csr.col, csr.rowscale)
    with pytest.raises(SimRankHipError):
        ShardPlans(ops, bad, world=2)


def _run_workers(n_ranks, tmp_path):
    idfile = tmp_path / "rccl_id.bin"
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, SHARD_RANK=str(r), SHARD_WORLD=str(n_ranks), SHARD_ID_FILE=str(idfile),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "shard_rccl_worker.py")], env=env,
                                      cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            out, err = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, out, err))
    return outs


def test_shardplan_over_rccl_one_rank(tmp_path):
    """The RCCL communicator itself (dlopen, ncclCommInitRank, the send / receive groups on their own stream, the
    all-reduce of the count) in a process of its own without torch: a world of one rank on the one GPU."""
    (rc, out, err), = _run_workers(1, tmp_path)
    assert rc == 0 and "SHARDPLAN RCCL ok" in out, out[-3000:] + err[-3000:]


def test_shardplan_over_rccl_two_ranks(tmp_path):
    """A real two-rank world (one process per GPU, the all-to-alls over xGMI); skipped on the one-GPU box."""
    from simrank_amd import _lib
    if _lib.device_count() < 2:
        pytest.skip("needs two GPUs")
    outs = _run_workers(2, tmp_path)
    assert all(rc == 0 and "SHARDPLAN RCCL ok" in out for rc, out, err in outs), [o[1][-2000:] + o[2][-2000:] for o in outs]


# ---------------------------------------------------------------------------------------------------------------------
# the two-matrix classes on shards (simrank_shardbiplan_*), and fits through the C loop on virtual ranks
# ---------------------------------------------------------------------------------------------------------------------
from tests.conftest import golden_names          # noqa: E402
from tests.helpers import check_against_golden, run_estimator          # noqa: E402


@pytest.mark.parametrize("world", [1, 2, 3, 4])
@pytest.mark.parametrize("name", golden_names("BipartiteSimRank", "BipartiteSimRankPP", "BipartitleAprioriSimRank"))
def test_bipartite_golden_vectors_through_the_sharded_c_loop(name, world):
    """Every bipartite vector of the reference through `fit(world=LocalWorld(P, loop="c"))`: cshard.CShardSolver ->
    simrank_shardbiplan_* on an in-process group of P virtual ranks (two exchanges per loop body, Gauss-Seidel order,
    Evidence_N1 on both updates and NumPy's broadcast error in strict mode, uneven and empty blocks) — labels, values,
    convergence index, console text.  Asymmetric priors (`*_asym`) included: a second exchange per update and group."""
    import simrank_amd.cshard as cshard
    g = Golden(name)
    made = []
    orig = cshard.CShardSolver.__init__

    def spy(self, *a, **k):
        orig(self, *a, **k)
        made.append(self)
    cshard.CShardSolver.__init__ = spy
    try:
        if g.raises:
            with pytest.raises(ValueError):
                run_estimator(g, world=LocalWorld(world, loop="c"), mode="sparse")
            return
        est, res, text = run_estimator(g, world=LocalWorld(world, loop="c"), mode="sparse")
    finally:
        cshard.CShardSolver.__init__ = orig
    if world > 1:
        assert len(made) == 1, (name, len(made))
    check_against_golden(g, est, res, text)


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("name", ["SimRank_er128", "SimRankPP_quirky", "AprioriSimRank_er64", "AprioriSimRank_er64_asym",
                                  "SimRank_toy5", "SimRankPP_bts300"])
def test_directed_golden_vectors_through_the_sharded_c_loop(name, world):
    g = Golden(name)
    est, res, text = run_estimator(g, world=LocalWorld(world, loop="c"), mode="sparse")
    check_against_golden(g, est, res, text)


@pytest.mark.parametrize("world,sym", [(2, False), (2, True), (4, True), (4, False), (8, True)])
def test_sharded_bipartite_c_loop_is_the_python_drivers_bits(world, sym):
    """The C loop of the two-matrix classes makes the launches of driver.Side on the same node orders: bit-equal to
    `fit(world=LocalWorld(P))` in the full form and in the half form (group sizes multiples of 32 x ranks), plain and
    SimRank++ (corrected evidence), with symmetric priors."""
    from tests.graphs import bipartite_random
    n1, n2 = 32 * world * 3, 32 * world * 2
    df = bipartite_random(n1, n2, 0.06, seed=world)
    rng = np.random.default_rng(world)
    p1, p2 = rng.random((n1, n1)), rng.random((n2, n2))
    p1, p2 = (p1 + p1.T) / 2, (p2 + p2.T) / 2
    cases = [(SRA.BipartiteSimRank, (), {}), (SRA.BipartiteSimRankPP, (), dict(strict_reference=False)),
             (SRA.BipartitleAprioriSimRank, (p1, p2), dict(strict_reference=False, lbd1=0.3, lbd2=0.2))]
    for cls, args, kw in cases:
        kw = dict(kw, iterations=5, eps=0, verbose=False, mode="sparse", C1=0.7, C2=0.85)
        a1, a2 = cls().fit(df, *args, world=LocalWorld(world, symmetric_shards=sym), **kw)
        b1, b2 = cls().fit(df, *args, world=LocalWorld(world, symmetric_shards=sym, loop="c"), **kw)
        assert np.array_equal(a1.values, b1.values) and np.array_equal(a2.values, b2.values), cls.__name__
        want = (O.fit_bipartite if cls is SRA.BipartiteSimRank else O.fit_bipartite_pp)(
            df, **{k: v for k, v in kw.items() if k not in ("mode",)},
            **(dict(apriori1=p1, apriori2=p2) if args else {}))
        assert_close(b1.values, want["S1"])
        assert_close(b2.values, want["S2"])


def test_sharded_bipartite_plan_step_by_step_and_top_k(ops):
    """simrank_shardbiplan_step / _run / hand-back entry points on 3 virtual ranks: counts of both groups, the convergence
    index, top-k per group against a sort of the dense result, timing stamps of a group's plan."""
    from simrank_amd.engine import ShardBiPlans
    from tests.graphs import bipartite_random
    df = bipartite_random(70, 45, 0.12, seed=2)
    _, _, _, _, g12, g21 = ingest.bipartite(df, False, "user", "item", "weight")
    want = O.fit_bipartite_pp(df, verbose=False, strict_reference=False)
    bp = ShardBiPlans(ops, g12, g12.rowscale, g21.rowscale, world=3, evidence=True, leg2_form=0, stages=2)
    done, conv = bp.run(100, 1e-4)
    assert conv == want["k"] and done == conv
    s1, s2 = bp.result(1), bp.result(2)
    assert_close(s1, want["S1"])
    assert_close(s2, want["S2"])
    bp.reset()
    counts = [bp.step(1e-4, exact_count=True) for _ in range(done)]
    assert counts[-1] == (0, 0) and all(c1 + c2 > 0 for c1, c2 in counts[:-1])
    for group, M in ((1, s1), (2, s2)):
        idx, val = bp.topk(group, 4)
        n = len(M)
        for a in range(n):
            cand = np.array([c for c in range(n) if c != a])
            order = cand[np.lexsort((cand, -M[a, cand]))][:4]
            assert list(idx[a]) == list(order)
            np.testing.assert_array_equal(val[a].astype(np.float64), M[a, order])
    assert bp.run(0, 1e-4) == (0, None) and bp.run(5, 1.0) == (0, 0)
    np.testing.assert_array_equal(bp.result(2), np.eye(45))
    bp.free()


# ---------------------------------------------------------------------------------------------------------------------
# concurrent ranks (round 6): one HOST THREAD per rank over the library's in-process transport (simrank_comm_thread_group) —
# each thread runs the code path of an RCCL rank (one plan, its own kernel and exchange streams, stage events, hops, grouped
# sends / receives that rendezvous with the peer's, the all-reduced count, speculative queueing of update k + 1) while the
# others run theirs.  What only a real RCCL world can add: RCCL's own progress engine and memory on several devices.
# ---------------------------------------------------------------------------------------------------------------------
def _thread_ranks(world, fn, timeout=420.0):
    from simrank_amd.engine import ThreadRanks
    tr = ThreadRanks(world)
    try:
        return tr.run(fn, timeout=timeout)
    finally:
        tr.close()


def _threads_vs_group(ops, csr, world, iterations, eps, scale=None, **kw):
    """One fit on `world` concurrent thread ranks and the same on the in-process group (ranks in turn): -> per-rank
    (done, conv, block, ids), root's full matrix, the group's (done, conv, full)."""
    from simrank_amd.engine import ShardPlans

    def rank(r, rops, comm):
        sp = ShardPlans(rops, csr, rowscale=scale, world=world, comm=comm, **kw)
        done, conv = sp.run(iterations, eps)
        full = sp.result(root=0, i_am_root=(r == 0))
        blk, ids = sp.block(0)
        inf = sp.info()
        sp.free()
        return done, conv, full, blk, ids, inf
    outs = _thread_ranks(world, rank)
    sp = ShardPlans(ops, csr, rowscale=scale, world=world, **kw)
    ref = sp.run(iterations, eps) + (sp.result(), sp.info())
    sp.free()
    return outs, ref


@pytest.mark.parametrize("world,form,stages", [(2, 0, 1), (2, 1, 3), (4, 0, 2), (4, 1, 1), (8, 0, 1), (8, 1, 2), (3, 0, 2), (5, 0, 1)])
@pytest.mark.parametrize("name", ["SimRank_er128", "SimRankPP_er128", "SimRank_toy5", "SimRankPP_bts300"])
def test_thread_ranks_reproduce_the_golden_vectors(ops, name, world, form, stages):
    """The golden cases of `test_shardplan_reproduces_the_golden_vectors` on 2 - 8 CONCURRENT ranks: S to 1e-5, the convergence
    index on every rank, root's gather == the blocks put together == bit for bit what the in-process group (ranks in turn,
    device copies) computes; uneven and empty blocks (toy5), both forms of leg 2, staged exchanges."""
    g = Golden(name)
    csr = golden_csr(g)
    if form == 1 and csr.n_rows % (32 * world):
        pytest.skip("the half form needs n % (32 x ranks) == 0")
    kw = dict(coef=g.kwargs.get("C", 0.8), evidence=name.startswith("SimRankPP"), leg2_form=form, stages=stages)
    outs, (rdone, rconv, rfull, rinf) = _threads_vs_group(ops, csr, world, g.kwargs.get("iterations", 100), g.kwargs.get("eps", 1e-4), **kw)
    put = np.full((csr.n_rows, csr.n_rows), np.nan)
    for r, (done, conv, full, blk, ids, inf) in enumerate(outs):
        assert (done, conv) == (rdone, rconv), (r, done, conv)
        assert (conv if conv is not None else -1) == (g.k if g.k is not None else -1)
        assert inf["half_form"] == bool(form) and inf["stages"] == stages
        put[:, ids] = blk
    assert_close(outs[0][2], g.out["S"])
    assert np.array_equal(outs[0][2], put)
    assert np.array_equal(outs[0][2], rfull)
    assert all(o[2] is None for o in outs[1:])


@pytest.mark.parametrize("world", [2, 3, 4])
def test_thread_ranks_with_priors_wires_and_half_storage(ops, world):
    """Concurrent ranks through the variants of the loop: a symmetric prior with evidence, a prior that is NOT symmetric (the
    second all-to-all, the epilogue as its own pass), the fp16 wire, fp16-held matrices — each bit-equal to the in-process
    group and within its bound of the float64 oracle."""
    df = synth.powerlaw_directed(64 * 2 * 3, 9, seed=5)            # n = 384: whole 64-column panels on 2 and 3 ranks
    _, csr = ingest.directed(df, False, "from", "to", "weight")
    n = csr.n_rows
    rng = np.random.default_rng(world)
    A = rng.random((n, n))
    sym = (A + A.T) / 2
    G = csr.dense()
    f32 = lambda x: x.astype(np.float32).astype(np.float64)       # noqa: E731
    want_sym, _ = O.iterate_directed(G, C=0.7, iterations=6, eps=0, E=O.evidence(G), apriori=f32(sym), lbd=0.3)
    want_asym, _ = O.iterate_directed(G, C=0.7, iterations=6, eps=0, E=O.evidence(G), apriori=f32(A), lbd=0.3)
    want_plain, _ = O.iterate_directed(G, C=0.7, iterations=6, eps=0)
    cases = [("prior+evidence", dict(evidence=True, apriori=sym, lbd=0.3, leg2_form=0, stages=2), want_sym, RTOL),
             ("asymmetric prior", dict(evidence=True, apriori=A, lbd=0.3, leg2_form=0, stages=1), want_asym, RTOL),
             ("fp16 wire", dict(leg2_form=0, stages=2, wire_fp16=True), want_plain, 5e-3)]
    if n % (64 * world) == 0:
        cases.append(("fp16-held", dict(leg2_form=0, stages=1, storage="fp16"), want_plain, 5e-3))
    for label, kw, want, tol in cases:
        outs, (rdone, rconv, rfull, _) = _threads_vs_group(ops, csr, world, 6, 0.0, coef=0.7, **kw)
        assert all(o[:2] == (6, None) for o in outs) and (rdone, rconv) == (6, None), label
        assert np.array_equal(outs[0][2], rfull), label
        if tol == RTOL:
            assert_close(outs[0][2], want)
        else:
            big = want > 1e-4
            assert (np.abs(outs[0][2] - want)[big] / want[big]).max() < tol and np.abs(outs[0][2] - want).max() < 1e-3, label


@pytest.mark.parametrize("world", [2, 3, 4])
def test_thread_ranks_on_the_two_matrix_loop(ops, world):
    """simrank_shardbiplan_* on concurrent ranks: two exchanges per loop body in Gauss-Seidel order (the group-2 update reads
    the NEW S1 of every peer), both counts all-reduced together — plain, SimRank++ (corrected evidence), strict quirk Q2 on
    equal group sizes, symmetric and asymmetric priors, full and half form: bit-equal to the in-process group; values and
    the convergence index against the float64 oracle."""
    from simrank_amd.engine import ShardBiPlans
    from tests.graphs import bipartite_random
    n1, n2 = 32 * world * 2, 32 * world
    df = bipartite_random(n1, n2, 0.08, seed=10 + world)
    _, _, _, _, g12, g21 = ingest.bipartite(df, False, "user", "item", "weight")
    rng = np.random.default_rng(world)
    p1, p2 = rng.random((n1, n1)), rng.random((n2, n2))
    s1, s2 = (p1 + p1.T) / 2, (p2 + p2.T) / 2
    pp1, pp2 = ingest.spread(g12) * g12.rowscale, ingest.spread(g21) * g21.rowscale
    cases = [("plain", (g12.rowscale, g21.rowscale), dict(leg2_form=0, stages=2), O.fit_bipartite(df, verbose=False)),
             ("plain, half form", (g12.rowscale, g21.rowscale), dict(leg2_form=1, stages=1), O.fit_bipartite(df, verbose=False)),
             ("pp", (pp1, pp2), dict(evidence=True, leg2_form=1, stages=2), O.fit_bipartite_pp(df, verbose=False, strict_reference=False)),
             ("priors", (pp1, pp2), dict(evidence=True, apriori1=s1, apriori2=s2, lbd1=0.3, lbd2=0.2, leg2_form=0, stages=1), None),
             ("asymmetric priors", (pp1, pp2), dict(evidence=True, apriori1=p1, apriori2=p2, lbd1=0.3, lbd2=0.2, leg2_form=0, stages=2), None)]
    for label, (rs1, rs2), opts, want in cases:
        def rank(r, rops, comm):
            bp = ShardBiPlans(rops, g12, rs1, rs2, world=world, comm=comm, **opts)
            res = bp.run(100, 1e-4)
            out = res, bp.result(1, root=0, i_am_root=(r == 0)), bp.result(2, root=0, i_am_root=(r == 0))
            bp.free()
            return out
        outs = _thread_ranks(world, rank)
        bp = ShardBiPlans(ops, g12, rs1, rs2, world=world, **opts)
        ref = bp.run(100, 1e-4)
        r1, r2 = bp.result(1), bp.result(2)
        bp.free()
        assert all(o[0] == ref for o in outs), (label, [o[0] for o in outs], ref)
        assert np.array_equal(outs[0][1], r1) and np.array_equal(outs[0][2], r2), label
        if want is not None:
            assert ref[1] == want["k"], label
            assert_close(outs[0][1], want["S1"])
            assert_close(outs[0][2], want["S2"])


def test_thread_ranks_at_config_4_to_eps(ops):
    """BASELINE config 4 (pl32768d32) to eps = 1e-4 on EIGHT concurrent ranks, half form, staged, with every device block
    poisoned before it is handed out: the loop ends at the single rank's iteration (16), every rank agrees, and the last
    rank's block is bit for bit the in-process group's (whose ranks run one after another)."""
    from simrank_amd.engine import Plan, ShardPlans
    df = synth.WORKLOADS["pl32768d32"][0]()
    _, csr = ingest.directed(df, False, "from", "to", "weight")
    P = 8
    os.environ["SIMRANK_POOL_POISON"] = "1"
    try:
        def rank(r, rops, comm):
            sp = ShardPlans(rops, csr, world=P, comm=comm, leg2_form=1, stages=2)
            res = sp.run(100, 1e-4)
            blk = sp.block(0) if r in (0, P - 1) else None
            sp.free()
            return res, blk
        outs = _thread_ranks(P, rank, timeout=600.0)
        sp = ShardPlans(ops, csr, world=P, leg2_form=1, stages=2)
        ref = sp.run(100, 1e-4)
        first, last = sp.block(0), sp.block(P - 1)
        sp.free()
        plan = Plan(ops, csr, coef=0.8)
        single = plan.run(100, 1e-4)
        plan.free()
    finally:
        os.environ.pop("SIMRANK_POOL_POISON", None)
    assert all(o[0] == ref for o in outs), [o[0] for o in outs]
    assert ref == single == (16, 16)
    assert np.array_equal(outs[0][1][0], first[0]) and np.array_equal(outs[0][1][1], first[1])
    assert np.array_equal(outs[P - 1][1][0], last[0]) and np.array_equal(outs[P - 1][1][1], last[1])


def test_a_thread_rank_that_never_arrives_is_an_error_not_a_hang(ops):
    """The transport's watchdog: three of four ranks enter the loop, the fourth never does — every waiting rank gets
    SimRankHipError after SIMRANK_THREAD_COMM_TIMEOUT seconds instead of waiting for ever."""
    from simrank_amd._lib import SimRankHipError
    from simrank_amd.engine import ShardPlans, ThreadRanks
    g = Golden("SimRank_er128")
    csr = golden_csr(g)
    os.environ["SIMRANK_THREAD_COMM_TIMEOUT"] = "3"
    try:
        tr = ThreadRanks(4)
    finally:
        os.environ.pop("SIMRANK_THREAD_COMM_TIMEOUT", None)

    def rank(r, rops, comm):
        sp = ShardPlans(rops, csr, world=4, comm=comm, leg2_form=0, stages=1)
        try:
            if r == 3:
                return "absent"
            with pytest.raises(SimRankHipError):
                sp.run(5, 0.0)
            return "error"
        finally:
            sp.free()
    try:
        assert tr.run(rank, timeout=120.0) == ["error", "error", "error", "absent"]
    finally:
        tr.close()
