#!/usr/bin/env python3
"""Randomised soak of the one-launch legs (fused.hip: leg 1, and leg 2 of a symmetric update) and of both legs on fp16-held matrices (half.hip) on a
real GPU: random shapes, densities, hub corners, knobs; every case against float64 NumPy.  Not part of the
test suite (minutes, not seconds): `python3 tools/soak_kernels.py [cases] [seed]`, log in profiles/.
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simrank_amd.engine import HipOps                               # noqa: E402
from tests.test_gpu_kernels import corner_csr, dense64, random_csr  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ops = HipOps(0)
rng = np.random.default_rng(seed0)
HALF_ULP = 2.0 ** -11
t0 = time.time()
worst32 = worst16 = 0.0
for case in range(cases):
    M = int(rng.choice([rng.integers(1, 140), rng.integers(100, 700), rng.integers(600, 2600)]))
    square = rng.random() < 0.6
    K = M if square else int(rng.integers(1, 2600))
    L = K if square else int(rng.integers(1, 400))
    kind = rng.integers(0, 3)
    if kind == 0:
        csr = corner_csr(M, K, seed=int(rng.integers(1 << 30)), hubs=int(min(K, rng.integers(1, 400))),
                         p_hub=float(rng.uniform(0.05, 0.9)), avg=float(rng.uniform(0.2, 12)))
    elif kind == 1:
        heavy = {int(rng.integers(0, M)): int(min(K, rng.integers(1, 1200))) for _ in range(int(rng.integers(0, 4)))}
        csr = random_csr(M, K, float(rng.uniform(0.2, 20)), seed=int(rng.integers(1 << 30)), heavy=heavy)
    else:
        csr = random_csr(M, K, float(rng.uniform(0, 3)), seed=int(rng.integers(1 << 30)))
    # row scales as a normalised adjacency has them (rows of W sum to at most 1: products stay inside fp16's range)
    from simrank_amd.ingest import CSR
    csr = CSR(csr.n_rows, csr.n_cols, csr.rowptr, csr.col,
              csr.rowscale / np.maximum(1, np.diff(csr.rowptr)))
    knobs = dict(fuse_min=int(rng.choice([0, 0, 2, 3, 4, 8])), fuse_pays=int(rng.choice([-1, 64, 192, 320])),
                 fuse_steps=int(rng.choice([1, 2, 8])),
                 fuse_group=int(rng.choice([1, 2, 4])), fuse_unit=int(rng.choice([4, 8, 48, 1 << 20])),
                 fuse_rows=int(rng.choice([64, 500, 8192])), fuse_order=int(rng.choice([0, 0, 1, 2, 3])))
    g = ops.graph(csr, knobs=knobs)
    # (fp16-held matrices take whole blocks only: the graph a fit on them creates, driver.Side)
    gh = ops.graph(csr, knobs=dict(knobs, fuse_unit=1 << 20))
    W = dense64(csr)
    X = (rng.random((K, L)) ** 3).astype(np.float32)
    want = (W @ X.astype(np.float64)).T
    # f32, one-launch leg 1
    xb = ops.matrix(K, L, blocked=True)
    ops.upload(xb, X)
    yt = ops.matrix(L, M, blocked=True)
    ops.spmm(g, xb, yt, transpose_out=True)
    got = ops.download(yt)
    ops.spmm(g, xb, yt, transpose_out=True)
    assert np.array_equal(got, ops.download(yt)), ("f32 leg 1 not reproducible", case, M, K, L, knobs)
    err = np.abs(got - want) / np.maximum(np.abs(want), 1e-30)
    bad = err.max() if want.size else 0.0
    assert bad < 1e-5, ("f32 leg 1", case, M, K, L, knobs, bad)
    worst32 = max(worst32, float(bad))
    # f32, leg 2 of a symmetric update as ONE launch (fused.hip, SYM; square patterns): every variant of the epilogue against
    # float64, the mirrored tiles bit for bit, the exact count = the count over the stored matrix
    if square and M >= 64:
        gs = ops.graph(csr, knobs=dict(knobs, fuse_sym=1))
        S = rng.random((M, M)) ** 3
        S = ((S + S.T) / 2).astype(np.float32)
        Tt = (W @ S.astype(np.float64)).T.astype(np.float32)
        c = rng.integers(0, 5, size=(M, M))
        counts = (np.triu(c) + np.triu(c, 1).T).astype(np.uint8) if rng.random() < 0.5 else None
        prior = None
        if rng.random() < 0.5:
            prior = rng.random((M, M)).astype(np.float32)
            prior = ((prior + prior.T) / 2).astype(np.float32)
        lbd = 0.3 if prior is not None else 0.0
        v = 0.8 * (W @ Tt.astype(np.float64))
        if counts is not None:
            v = v * (1.0 - 0.5 ** counts.astype(np.float64))
        if prior is not None:
            v = (1 - np.float32(lbd)) * v + np.float32(lbd) * prior.astype(np.float64)
        np.fill_diagonal(v, 1.0)
        mats = [ops.matrix(M, M, blocked=True) for _ in range(3)]
        ops.upload(mats[0], Tt)
        ops.upload(mats[1], S)
        ev = ap = None
        if counts is not None:
            ev = ops.matrix(M, M, np.uint8, blocked=True)
            ops.upload(ev, counts)
            mats.append(ev)
        if prior is not None:
            ap = ops.matrix(M, M, blocked=True)
            ops.upload(ap, prior)
            mats.append(ap)
        ops.spmm(gs, mats[0], mats[2], epilogue=dict(coef=0.8, previous=mats[1], eps=0.02, diag_col0=0, symmetric=True,
                                                     evidence=ev, apriori=ap, lbd=lbd))
        got2, moved = ops.download(mats[2]), ops.read_changed()
        e2 = np.abs(got2 - v) / np.maximum(np.abs(v), 1e-30)
        assert e2.max() < 1e-5, ("f32 one-launch leg 2", case, M, knobs, float(e2.max()))
        iu = np.triu_indices(M, 32)                              # (tiles off the diagonal: the same bits both sides)
        tl = (iu[0] // 32) != (iu[1] // 32)
        assert np.array_equal(got2[iu][tl], got2.T[iu][tl]), ("f32 one-launch leg 2: mirrored tiles", case, M, knobs)
        assert moved == int((np.abs(got2.astype(np.float64) - S) > 0.02).sum()), ("f32 one-launch leg 2: count", case, M, knobs)
        worst32 = max(worst32, float(e2.max()))
        for m in mats:
            m.free()
        gs.free()
    # fp16-held, leg 1
    Xh = X.astype(np.float16)
    want_h = (W @ Xh.astype(np.float64)).T
    xh = ops.matrix(K, L, np.float16, blocked=True)
    ops.upload(xh, Xh)
    yh = ops.matrix(L, M, np.float16, blocked=True)
    ops.spmm(gh, xh, yh, transpose_out=True)
    got_h = ops.download(yh).astype(np.float64)
    bound = 1.02 * np.maximum(HALF_ULP * np.abs(want_h), 2.0 ** -25) + 1e-6 * np.abs(want_h)
    over = np.abs(got_h - want_h) - bound
    assert not (over > 0).any(), ("fp16 leg 1", case, M, K, L, knobs, float(over.max()))
    # fp16-held, leg 2 (square patterns): symmetric result, within one rounding, count of what moved
    if square and M >= 2:
        S = rng.random((M, M)) ** 4
        S = ((S + S.T) / 2).astype(np.float16)
        Tt = (W @ S.astype(np.float64)).T.astype(np.float16)
        prev = rng.random((M, M)) ** 4
        prev = ((prev + prev.T) / 2).astype(np.float16)
        c = rng.integers(0, 5, size=(M, M))
        counts = (np.triu(c) + np.triu(c, 1).T).astype(np.uint8) if rng.random() < 0.5 else None
        v = 0.8 * (W @ Tt.astype(np.float64))
        if counts is not None:
            v = v * (1.0 - 0.5 ** counts.astype(np.float64))
        np.fill_diagonal(v, 1.0)
        th = ops.matrix(M, M, np.float16, blocked=True)
        ops.upload(th, Tt)
        ph = ops.matrix(M, M, np.float16, blocked=True)
        ops.upload(ph, prev)
        ev = None
        if counts is not None:
            ev = ops.matrix(M, M, np.uint8, blocked=True)
            ops.upload(ev, counts)
        y2 = ops.matrix(M, M, np.float16, blocked=True)
        ops.spmm(gh, th, y2, epilogue=dict(coef=0.8, previous=ph, eps=1e-3, set_diag=True, symmetric=True, evidence=ev))
        got2 = ops.download(y2).astype(np.float64)
        assert np.array_equal(got2, got2.T), ("fp16 leg 2 symmetry", case, M, knobs)
        iu = np.triu_indices(M)
        b2 = 1.02 * np.maximum(HALF_ULP * np.abs(v[iu]), 2.0 ** -25) + 1e-6 * np.abs(v[iu])
        o2 = np.abs(got2[iu] - v[iu]) - b2
        assert not (o2 > 0).any(), ("fp16 leg 2", case, M, knobs, float(o2.max()))
        big = np.abs(v[iu]) > 1e-3                                  # (smaller values: the absolute floor of the bound)
        if big.any():
            worst16 = max(worst16, float((np.abs(got2[iu] - v[iu])[big] / np.abs(v[iu])[big]).max()))
        for m in (th, ph, y2) + ((ev,) if ev is not None else ()):
            m.free()
    for m in (xb, yt, xh, yh):
        m.free()
    g.free()
    gh.free()
    if case % 25 == 24:
        print(f"{case + 1} cases, {time.time() - t0:.0f} s, worst f32 rel {worst32:.2e}, worst fp16 leg-2 rel (values > 1e-3) {worst16:.2e}",
              flush=True)
# the hand-off of split blocks under load: the full-size launch (about 40 split units x 1024 panels, every CU busy)
# ten times over — one stale partial sum anywhere changes a bit
if os.environ.get("SOAK_BIG", "1") != "0":
    from simrank_amd import ingest, synth
    from tests.pydriver import SideSpec, reorder_specs
    df = synth.WORKLOADS["pl32768d32"][0]()
    _, csr = ingest.directed(df, False, "from", "to", "weight")
    specs, _ = reorder_specs([SideSpec(csr, csr.rowscale, 0.8)])
    c = specs[0].csr
    g = ops.graph(c)
    n = c.n_rows
    Xb = ops.matrix(n, n, blocked=True)
    ops.upload(Xb, (rng.random((n, n), dtype=np.float32) ** 3))
    Yb = ops.matrix(n, n, blocked=True)
    ops.spmm(g, Xb, Yb, transpose_out=True)
    first = ops.download(Yb)
    for rep in range(9):
        ops.spmm(g, Xb, Yb, transpose_out=True)
        assert np.array_equal(first, ops.download(Yb)), ("full-size leg 1 not reproducible", rep)
    rows = rng.integers(0, n, size=24)
    Xh = ops.download(Xb).astype(np.float64)
    for a in rows:
        cols = c.col[c.rowptr[a]:c.rowptr[a + 1]]
        want_row = c.rowscale[a] * Xh[cols].sum(axis=0)
        np.testing.assert_allclose(first[:, a], want_row, rtol=1e-5, atol=1e-30)
    print(f"full-size pl32768d32 leg 1: ten launches bit-equal, {len(rows)} sampled rows within 1e-5 of float64", flush=True)
print(f"soak: {cases} random cases passed (seed {seed0}); worst f32 leg-1 relative error {worst32:.2e}, "
      f"worst fp16 leg-2 relative error on values > 1e-3 {worst16:.2e} (half an fp16 spacing = {HALF_ULP:.2e})")
