"""The path a user's ``fit()`` takes on one MI355X since round 5: the loops behind the C ABI
(``cplan.PlanSolver`` over simrank_plan_* / simrank_biplan_*, SimRank.py:129-141, :288-303), their console hooks, the
symmetric hand-back (csrc/handback.hip: upper triangle over PCIe, mirrored on the host), per-engine counter slots, and
un-zeroed pooled memory at ragged sizes."""
import contextlib
import io
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

import simrank_amd.SimRank as SRA
from oracle import simrank_oracle as O
from simrank_amd import ingest, synth
from tests.graphs import bipartite_random
from tests.helpers import assert_close

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ops():
    from simrank_amd.engine import HipOps
    return HipOps(0)


# ---------------------------------------------------------------------------------------------------------------
# the symmetric hand-back against the full one, bit for bit
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("blocked", [False, True])
@pytest.mark.parametrize("n", [1, 5, 64, 65, 300, 1000, 2049, 4096])
def test_handback_f64_in_both_forms(ops, n, blocked):
    """dst[i][j] = src[idx[i]][idx[j]] as float64 from a matrix in either layout, with and without a node order, in the
    full form and — on a matrix that is mirror-equal outside its diagonal blocks — the symmetric one: exactly the values
    of round 4's hand-back (one permute of the whole matrix + simrank_download_f64)."""
    rng = np.random.default_rng(n)
    a = rng.random((n, n)).astype(np.float32)
    a = np.maximum(a, a.T)                                   # bitwise symmetric
    a[rng.random((n, n)) < 0.3] = 0.0
    a = np.maximum(a, a.T)
    for k in range(0, n, 32):                                # ... except inside the 32 x 32 diagonal blocks
        a[k:k + 32, k:k + 32] = rng.random(a[k:k + 32, k:k + 32].shape).astype(np.float32)
    m = ops.matrix(n, n, blocked=blocked)
    ops.upload(m, a)
    for order in (None, rng.permutation(n)):
        idx = None if order is None else ops.index_vector(order)
        want = a.astype(np.float64) if order is None else a[order][:, order].astype(np.float64)
        for symmetric in (False, True):
            got = ops.handback_f64(m, idx, symmetric=symmetric)
            np.testing.assert_array_equal(got, want)
        # the full path of round 4: one permute into a row-major copy, every element over PCIe
        tmp = ops.matrix(n, n)
        ops.permute(m, tmp, idx, idx)
        np.testing.assert_array_equal(ops.download_f64(tmp), got)
        tmp.free()
        if idx is not None:
            idx.free()
    m.free()


def test_handback_f64_into_a_wider_frame_and_of_an_asymmetric_matrix(ops):
    n, ld = 200, 333
    rng = np.random.default_rng(3)
    a = rng.random((n, n)).astype(np.float32)
    from simrank_amd._lib import check
    for sym_input in (True, False):
        b = np.minimum(a, a.T) if sym_input else a            # (asymmetric: the symmetric form must notice and fall back)
        m = ops.matrix(n, n, blocked=True)
        ops.upload(m, b)
        for mode in (0, 1):
            out = np.full((n, ld), -1.0)
            check(ops.lib.simrank_handback_f64(out.ctypes.data, ld, m.ptr, m.ld, m.rows_pad, n, None, mode, ops.stream))
            np.testing.assert_array_equal(out[:, :n], b.astype(np.float64))
            assert (out[:, n:] == -1.0).all()                 # nothing beyond the n columns is touched
        m.free()


def test_half_and_full_hand_back_of_config_4_are_the_same_bits(ops, monkeypatch):
    """BASELINE config 4 (N = 32768, nnz 1 048 576): four updates through the C plan, then the hand-back fit() returns
    (pipelined, full form), the symmetric form (upper triangle over PCIe) and round 4's path (simrank_plan_result +
    simrank_download_f64), every element."""
    from simrank_amd.engine import Plan
    df = synth.WORKLOADS["pl32768d32"][0]()
    _, csr = ingest.directed(df, False, "from", "to", "weight")
    n = csr.n_rows
    plan = Plan(ops, csr, coef=0.8)
    plan.reset()
    for _ in range(3):
        plan.step(0.0)
    before = plan.result()
    plan.step(0.0)
    pipelined = plan.result()
    # 24 sampled rows of the fourth update (the C loop fit() runs) within 1e-5 of a float64 recomputation from the third
    import scipy.sparse as sp
    rs = csr.rowscale.astype(np.float32).astype(np.float64)
    W = sp.diags(rs) @ sp.csr_matrix((np.ones(csr.col.size), csr.col, csr.rowptr), shape=(n, n))
    rows = sorted({0, n - 1, int(np.argmax(np.diff(csr.rowptr)))} |
                  set(int(r) for r in np.random.default_rng(4).choice(n, size=21, replace=False)))
    for a in rows:
        t = rs[a] * before[csr.col[csr.rowptr[a]:csr.rowptr[a + 1]]].sum(axis=0)
        want = 0.8 * (W @ t)
        want[a] = 1.0
        np.testing.assert_allclose(pipelined[a], want, rtol=1e-5, atol=1e-30)
    del before
    monkeypatch.setenv("SIMRANK_SYM_HANDBACK", "1")
    half = plan.result()
    monkeypatch.delenv("SIMRANK_SYM_HANDBACK")
    for r0 in range(0, n, 4096):
        assert np.array_equal(half[r0:r0 + 4096], pipelined[r0:r0 + 4096])
    del pipelined
    tmp = ops.matrix(n, n)
    from simrank_amd._lib import check
    check(ops.lib.simrank_plan_result(plan.handle, tmp.ptr, tmp.ld), "simrank_plan_result")
    full = ops.download_f64(tmp)
    tmp.free()
    plan.free()
    assert half.shape == full.shape == (n, n)
    for r0 in range(0, n, 4096):                           # (slab by slab: no third 8.6 GB temporary)
        assert np.array_equal(half[r0:r0 + 4096], full[r0:r0 + 4096])
    rows = np.random.default_rng(0).integers(0, n, 64)
    # (mirror-equal except inside the 32 x 32 diagonal blocks of the solver's order, where both triangles are computed)
    assert (half[rows] == half[:, rows].T).mean() > 0.998
    assert (np.diag(half) == 1.0).all()
    del half, full
    type(ops).trim_pool()


# ---------------------------------------------------------------------------------------------------------------
# which loop fit() runs, and its console hooks
# ---------------------------------------------------------------------------------------------------------------
def test_fit_runs_the_c_level_loops(monkeypatch):
    """The product has TWO choreographies, both behind the C ABI: every class on one GPU goes through cplan.PlanSolver
    (simrank_plan_run_cb / simrank_biplan_run_cb) — asymmetric priors included unless the matrices are fp16-held —, several
    ranks through cshard.CShardSolver (simrank_shardplan_* / simrank_shardbiplan_*).  With the tests' double unplugged
    (estimators.PYTHON_SOLVER = None: the state of a user's process) nothing else exists: the GEMM modes and an injected
    engine are refused with the reason; with it plugged in they — and only they — run tests/pydriver.Solver."""
    import simrank_amd.cplan as cplan
    import simrank_amd.cshard as cshard
    import simrank_amd.estimators as est
    import tests.pydriver as drv
    made = []
    for cls, tag in ((cplan.PlanSolver, "plan"), (cshard.CShardSolver, "shards"), (drv.Solver, "double")):
        orig = cls.__init__

        def spy(self, *a, _orig=orig, _tag=tag, **k):
            _orig(self, *a, **k)
            made.append(_tag)
        monkeypatch.setattr(cls, "__init__", spy)
    from simrank_amd.driver import LocalWorld as ProductWorld
    df = synth.er_directed(96, 0.08, seed=2)
    dfb = bipartite_random(40, 40, 0.15, seed=5)
    rng = np.random.default_rng(1)
    sym = rng.random((96, 96))
    sym = (sym + sym.T) / 2
    asym = rng.random((96, 96))
    b1 = rng.random((40, 40))
    b1 = (b1 + b1.T) / 2
    cases = [
        (lambda: SRA.SimRank().fit(df, verbose=False), "plan"),
        (lambda: SRA.SimRank().fit(df, verbose=False, mode="sparse"), "plan"),
        (lambda: SRA.SimRankPP().fit(df, verbose=False), "plan"),
        (lambda: SRA.SimRankPP().fit(df, verbose=False, storage_precision="fp16"), "plan"),
        (lambda: SRA.SimRankPP().fit(df, verbose=False, dense_precision="fp16"), "plan"),
        (lambda: SRA.AprioriSimRank().fit(df, sym, verbose=False), "plan"),
        (lambda: SRA.AprioriSimRank().fit(df, asym, verbose=False), "plan"),
        (lambda: SRA.BipartitleAprioriSimRank().fit(dfb, b1, rng.random((40, 40)), verbose=False), "plan"),
        (lambda: SRA.BipartiteSimRank().fit(dfb, verbose=False), "plan"),
        (lambda: SRA.BipartiteSimRankPP().fit(dfb, verbose=False), "plan"),
        (lambda: SRA.BipartiteSimRankPP().fit(dfb, verbose=False, strict_reference=False, top_k=3), "plan"),
        (lambda: SRA.BipartitleAprioriSimRank().fit(dfb, b1, b1.copy(), verbose=False), "plan"),
        (lambda: SRA.SimRank().fit(df, verbose=False, world=ProductWorld(1)), "plan"),
        (lambda: SRA.SimRank().fit(df, verbose=False, world=ProductWorld(2)), "shards"),
        (lambda: SRA.SimRankPP().fit(df, verbose=False, world=ProductWorld(3, symmetric_shards=False)), "shards"),
        (lambda: SRA.AprioriSimRank().fit(df, asym, verbose=False, world=ProductWorld(2)), "shards"),
        (lambda: SRA.BipartiteSimRankPP().fit(dfb, verbose=False, world=ProductWorld(4)), "shards"),
    ]
    monkeypatch.setattr(est, "PYTHON_SOLVER", None)
    for run, want in cases:
        made.clear()
        run()
        assert made == [want], (made, want)
    for kw, why in ((dict(mode="dense"), "mode='dense'"), (dict(mode="hybrid"), "mode='hybrid'"),
                    (dict(_ops_factory=lambda r: None), "an injected engine"), (dict(world=drv.LocalWorld(2)), "loop='python'")):
        with pytest.raises(ValueError, match=why):
            SRA.SimRank().fit(df, verbose=False, **kw)
    with pytest.raises(ValueError, match="mode must be"):
        SRA.SimRank().fit(df, verbose=False, mode="gemm")
    monkeypatch.setattr(est, "PYTHON_SOLVER", drv.make_solver)
    for kw in (dict(mode="dense"), dict(mode="hybrid"), dict(world=drv.LocalWorld(2))):
        made.clear()
        SRA.SimRank().fit(df, verbose=False, **kw)
        assert made == ["double"], (kw, made)
    made.clear()
    SRA.SimRank().fit(df, verbose=False)                       # (the double plugged in changes nothing for an ordinary fit)
    assert made == ["plan"]


def test_progress_hooks_of_the_c_loop(ops):
    """simrank_plan_run_cb: on_iteration(k) for every loop index that goes on to an update, on_converged(k) once, in the
    reference's order (SimRank.py:131-135) — with and without the speculative update (N below / above 16384 is decided in
    C; here: below) — and an exception raised in a hook ends the loop and reaches the caller."""
    from simrank_amd.engine import BiPlan, Plan
    df = synth.er_directed(300, 0.03, seed=11)
    _, csr = ingest.directed(df, False, "from", "to", "weight")
    want = O.fit_simrank(df, verbose=False)
    plan = Plan(ops, csr, coef=0.8)
    seen = []
    done, conv = plan.run(100, 1e-4, on_iteration=lambda k: seen.append(("it", k)), on_converged=lambda k: seen.append(("conv", k)))
    assert conv == want["k"] and done == conv
    assert seen == [("it", k) for k in range(conv)] + [("conv", conv)]
    assert_close(plan.result(), want["S"])
    seen.clear()
    assert plan.run(3, 0.0, on_iteration=lambda k: seen.append(k)) == (3, None)
    assert seen == [0, 1, 2]
    seen.clear()
    assert plan.run(5, 1.0, on_iteration=lambda k: seen.append(k), on_converged=lambda k: seen.append(("conv", k))) == (0, 0)
    assert seen == [("conv", 0)]

    def boom(k):
        if k == 2:
            raise KeyError("stop here")
    with pytest.raises(KeyError):
        plan.run(50, 0.0, on_iteration=boom)
    assert plan.run(2, 0.0) == (2, None)                   # the plan is still usable
    plan.free()
    dfb = bipartite_random(50, 30, 0.2, seed=8)
    _, _, _, _, g12, g21 = ingest.bipartite(dfb, False, "user", "item", "weight")
    wantb = O.fit_bipartite(dfb, verbose=False)
    bp = BiPlan(ops, g12, g12.rowscale, g21.rowscale)
    seen.clear()
    done, conv = bp.run(100, 1e-4, on_iteration=lambda k: seen.append(("it", k)), on_converged=lambda k: seen.append(("conv", k)))
    assert conv == wantb["k"]
    assert seen == [("it", k) for k in range(conv)] + [("conv", conv)]
    with pytest.raises(KeyError):
        bp.run(50, 0.0, on_iteration=boom)
    s1, s2 = bp.result()
    bp.free()


def test_verbose_fit_prints_the_references_text():
    """The console text of a verbose fit through the C loop, character for character what the oracle prints."""
    df = synth.er_directed(200, 0.04, seed=4)
    import re
    strip = lambda t: re.sub(r"Finished in [0-9.e+-]+s!", "Finished in <t>s!", t)
    for cls, ref in ((SRA.SimRank, O.fit_simrank), (SRA.SimRankPP, O.fit_simrank_pp)):
        a = io.StringIO()
        with contextlib.redirect_stdout(a):
            got = cls().fit(df)
        want = ref(df)
        assert strip(a.getvalue()) == strip(want["stdout"]) and "Converged at iteration" in a.getvalue()
        assert_close(got.values, want["S"])


def test_plan_evidence_counts_trim_and_biplan_topk(ops):
    from simrank_amd._lib import SimRankHipError
    from simrank_amd.engine import BiPlan, Plan
    df = synth.powerlaw_directed(333, 6, seed=3)
    _, csr = ingest.directed(df, False, "from", "to", "weight")
    want = O.fit_simrank_pp(df, verbose=False)
    plan = Plan(ops, csr, ingest.spread(csr) * csr.rowscale, coef=0.8, evidence=True)
    plan.run(100, 1e-4)
    np.testing.assert_array_equal(1 - 0.5 ** plan.evidence_counts().astype(np.float64), want["E"])
    plan.trim()
    np.testing.assert_array_equal(1 - 0.5 ** plan.evidence_counts().astype(np.float64), want["E"])
    with pytest.raises(SimRankHipError):
        plan.result()
    with pytest.raises(SimRankHipError):
        plan.run(1, 0.0)
    plan.free()
    dfb = bipartite_random(70, 45, 0.12, seed=2)
    _, _, _, _, g12, g21 = ingest.bipartite(dfb, False, "user", "item", "weight")
    bp = BiPlan(ops, g12, g12.rowscale, g21.rowscale, evidence=True)
    bp.run(5, 1e-30)
    S = bp.result()
    wantb = O.fit_bipartite_pp(dfb, verbose=False, iterations=5, eps=1e-30, strict_reference=False)
    for group in (1, 2):
        np.testing.assert_array_equal(1 - 0.5 ** bp.evidence_counts(group).astype(np.float64), wantb[f"E{group}"])
        idx, val = bp.topk(group, 4)
        M = S[group - 1]
        n = len(M)
        for a in range(n):
            cand = np.array([c for c in range(n) if c != a])
            order = cand[np.lexsort((cand, -M[a, cand]))][:4]
            assert list(idx[a]) == list(order)
            np.testing.assert_array_equal(val[a].astype(np.float64), M[a, order])
    bp.free()


# ---------------------------------------------------------------------------------------------------------------
# two fits side by side on one device (advisor, round 4: the pinned counter slots were a per-device table)
# ---------------------------------------------------------------------------------------------------------------
def test_two_threads_fit_side_by_side():
    from tests.pydriver import LocalWorld, SideSpec, Solver
    from simrank_amd.engine import HipOps
    dfs = [synth.er_directed(500, 0.02, seed=21), synth.powerlaw_directed(700, 5, seed=22)]
    wants = [O.fit_simrank(d, verbose=False) for d in dfs]
    assert wants[0]["k"] != wants[1]["k"]                  # (different loop lengths: a shared slot would show)
    errors = []

    def through_fit(i):
        try:
            for _ in range(4):
                est = SRA.SimRank()
                got = est.fit(dfs[i], verbose=False)
                assert est.converged_at == wants[i]["k"], (i, est.converged_at, wants[i]["k"])
                assert_close(got.values, wants[i]["S"])
        except BaseException as e:
            errors.append(e)

    def through_the_python_driver(i):
        # driver.Solver.run reads its counts one update late through the engine's own pinned slots
        try:
            ops = HipOps(0)
            _, csr = ingest.directed(dfs[i], False, "from", "to", "weight")
            for _ in range(4):
                s = Solver(lambda r: ops, LocalWorld(1), [SideSpec(csr, csr.rowscale, 0.8)], "sparse")
                assert s._can_defer()
                k = s.run(100, 1e-4)
                assert k == wants[i]["k"], (i, k, wants[i]["k"])
                S = s.result(0)
                s.release()
                # (labels: the solver hands back in the caller's CSR order = the oracle's label order)
                assert_close(S, wants[i]["S"])
        except BaseException as e:
            errors.append(e)

    for body in (through_fit, through_the_python_driver):
        threads = [threading.Thread(target=body, args=(i,)) for i in range(2)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors


# ---------------------------------------------------------------------------------------------------------------
# un-zeroed memory at ragged sizes (advisor, round 4)
# ---------------------------------------------------------------------------------------------------------------
POISON_SCRIPT = r'''
import sys
sys.path.insert(0, sys.argv[1])
import numpy as np
import simrank_amd.SimRank as SRA
from oracle import simrank_oracle as O
from simrank_amd import synth
from tests.pydriver import LocalWorld
for n, deg in ((77, 4), (333, 6), (1000, 5), (2113, 7)):          # not multiples of 32 / 64
    df = synth.powerlaw_directed(n, deg, seed=n)
    for cls, ref in ((SRA.SimRank, O.fit_simrank), (SRA.SimRankPP, O.fit_simrank_pp)):
        want = ref(df, verbose=False)
        est = cls()
        got = est.fit(df, verbose=False)
        assert est.converged_at == want["k"], (n, cls.__name__, est.converged_at, want["k"])
        np.testing.assert_allclose(got.values, want["S"], rtol=1e-5, atol=1e-30)
        # the Python driver on the same poisoned allocator (virtual ranks; asymmetric iterates are covered by the goldens)
        got2 = cls().fit(df, verbose=False, world=LocalWorld(2))
        np.testing.assert_allclose(got2.values, want["S"], rtol=1e-5, atol=1e-30)
        # top-k from the panel-blocked matrix: the last, partial panel must not offer its padding
        top = cls().fit(df, verbose=False, top_k=5)
        S = want["S"].copy()
        np.fill_diagonal(S, -1.0)
        labels = np.array(want["labels"])
        pos = {l: i for i, l in enumerate(labels)}
        for node, grp in top.groupby("node"):
            a = pos[node]
            best = np.sort(S[a])[::-1][:len(grp)]
            np.testing.assert_allclose(np.sort(grp["similarity"].values)[::-1], best, rtol=1e-5, atol=1e-30)
            assert not grp["neighbor"].isin([node]).any()
        # fp16-held matrices (64-column panels): five updates on both sides, a few fp16 roundings apart
        w5 = ref(df, verbose=False, iterations=5, eps=1e-30)
        g5 = cls().fit(df, verbose=False, iterations=5, eps=1e-30, storage_precision="fp16")
        assert np.isfinite(g5.values).all()
        np.testing.assert_allclose(g5.values, w5["S"], rtol=3e-3, atol=2e-7)
        t5 = cls().fit(df, verbose=False, iterations=5, eps=1e-30, storage_precision="fp16", top_k=4)
        assert np.isfinite(t5["similarity"].values).all() and (t5["similarity"].values <= 1.0).all()
print("poisoned pool ok")
'''


def test_fits_on_poisoned_memory_at_ragged_sizes(tmp_path):
    """SIMRANK_POOL_POISON=1 fills every device block the library hands out with 0xFF bytes (NaN as f32 / fp16): no
    kernel may let the padding rows or columns of an un-zeroed matrix reach a sum, a count, a top-k or a store —
    f32 and fp16-held matrices, N not a multiple of 32 / 64, SimRank and SimRank++, dense and top-k hand-back."""
    script = tmp_path / "poison.py"
    script.write_text(POISON_SCRIPT)
    env = dict(os.environ, SIMRANK_POOL_POISON="1")
    run = subprocess.run([sys.executable, str(script), ROOT], env=env, capture_output=True, text=True, timeout=900)
    assert run.returncode == 0 and "poisoned pool ok" in run.stdout, (run.stdout[-2000:], run.stderr[-4000:])


def test_first_update_from_the_identity_without_gathers(ops, monkeypatch):
    """S_0 = I (SimRank.py:124-126): the first leg 1 of a plan writes W^T directly (a zero fill + one value per entry) instead
    of gathering — the SAME BITS as the gather leg, update by update, for every class a plan runs (plain, evidence,
    symmetric and asymmetric priors, the two-matrix plan whose group 1 reads S2 = I), also after a reset and in run()."""
    from simrank_amd.engine import BiPlan, Plan
    from simrank_amd import ingest
    df = synth.powerlaw_directed(2500, 14, seed=3)
    _, csr = ingest.directed(df, False, "from", "to", "weight")
    rng = np.random.default_rng(5)
    sym = rng.random((csr.n_rows, csr.n_rows))
    sym = (sym + sym.T) / 2
    asym = rng.random((csr.n_rows, csr.n_rows))

    def directed(**kw):
        p = Plan(ops, csr, coef=0.75, **kw)
        outs = []
        for _ in range(3):
            p.step(0.0, exact_count=True)
            outs.append(p.result())
        p.reset()
        p.step(0.0)
        outs.append(p.result())                      # after a reset: from the identity again
        outs.append(p.run(6, 1e-4))
        outs.append(p.result())
        p.free()
        return outs
    dfb = bipartite_random(900, 400, 0.03, seed=4)
    _, _, _, _, g12, g21 = ingest.bipartite(dfb, False, "user", "item", "weight")

    def two_matrix():
        p = BiPlan(ops, g12, g12.rowscale, g21.rowscale, evidence=True)
        outs = []
        for _ in range(2):
            p.step(0.0, True)
            outs.extend(p.result())
        outs.append(p.run(5, 1e-4))
        outs.extend(p.result())
        p.free()
        return outs
    cases = [lambda: directed(), lambda: directed(evidence=True), lambda: directed(evidence=True, apriori=sym, lbd=0.3),
             lambda: directed(evidence=True, apriori=asym, lbd=0.3), two_matrix,
             lambda: directed(storage="fp16"), lambda: directed(evidence=True, storage="fp16")]      # fp16-held: the same rounding
    for case in cases:
        monkeypatch.setenv("SIMRANK_IDENTITY_LEG1", "0")
        want = case()
        monkeypatch.delenv("SIMRANK_IDENTITY_LEG1")
        got = case()
        for a, b in zip(got, want):
            if isinstance(a, tuple):
                assert a == b
            else:
                assert np.array_equal(a, b)
