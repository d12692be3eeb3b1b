"""NumPy stand-in for ``simrank_amd.engine.HipOps`` — TEST INFRASTRUCTURE ONLY.

It implements the documented semantics of the C ABI entry points (include/simrank_hip.h)
on host arrays, so that the host logic above the ABI — ingest, the iteration driver, the
sharded exchange layout, the estimators' console text and quirk handling — can be tested
without a GPU (`-m "not gpu"`), including a world_size-2 ``gloo`` run.  It is never
imported by the product package; the product path has no CPU fallback.
"""
import numpy as np
import scipy.sparse as sp


class NpMatrix:
    _next_base = 1 << 40

    def __init__(self, ops, rows, cols, dtype, ld=None, external=None):
        self.ops = ops
        self.rows, self.cols = int(rows), int(cols)
        self.dtype = np.dtype(dtype)
        self.ld = int(ld if ld is not None else ops.pitch(cols, self.dtype))
        n = max(1, self.rows) * self.ld
        if external is not None:
            self.flat = external.numpy()[:n]          # torch CPU tensor: shared memory
        else:
            self.flat = np.full(n, 7, dtype=self.dtype)   # junk, like fresh device memory
        self.external = external
        self.ptr = NpMatrix._next_base
        NpMatrix._next_base += 1 << 40
        ops._buffers[self.ptr] = self
        self.nbytes = n * self.dtype.itemsize

    @property
    def a(self):
        return self.flat[: self.rows * self.ld].reshape(self.rows, self.ld)

    def free(self):
        self.ops._buffers.pop(self.ptr, None)


class NpGraph:
    def __init__(self, csr, rowscale):
        self.n_rows, self.n_cols = csr.n_rows, csr.n_cols
        self.rowscale = np.asarray(csr.rowscale if rowscale is None else rowscale,
                                   dtype=np.float32)
        self.pattern = sp.csr_matrix((np.ones(csr.col.size, dtype=np.float32), csr.col,
                                      csr.rowptr), shape=(csr.n_rows, csr.n_cols))


class NumpyOps:
    name = "numpy-test-double"
    supports_shard_symmetric = True

    def __init__(self):
        self._buffers = {}
        self._changed = 0
        self.calls = []

    def pitch(self, cols, dtype):
        unit = 16 // np.dtype(dtype).itemsize
        return -(-cols // unit) * unit

    def matrix(self, rows, cols, dtype=np.float32, ld=None, external=None):
        return NpMatrix(self, rows, cols, dtype, ld, external)

    def exchange_buffer(self, n_floats):
        import torch
        return torch.full((max(1, n_floats),), 7.0, dtype=torch.float32)

    WIRE_SCALE = 16384.0

    def exchange_buffer_h(self, n):
        import torch
        return torch.full((max(8, n),), 3.0, dtype=torch.float16)

    def narrow_t(self, src_t, dst_t, off, n):
        v = np.clip(src_t.numpy()[off:off + n].astype(np.float64) * self.WIRE_SCALE, -65504.0, 65504.0)
        dst_t.numpy()[off:off + n] = v.astype(np.float16)

    def widen_t(self, src_t, dst_t, off, n):
        dst_t.numpy()[off:off + n] = (src_t.numpy()[off:off + n].astype(np.float32) / np.float32(self.WIRE_SCALE))

    def round_trip_h16(self, m, n):
        v = np.clip(m.flat[:n].astype(np.float64) * self.WIRE_SCALE, -65504.0, 65504.0).astype(np.float16)
        m.flat[:n] = v.astype(np.float32) / np.float32(self.WIRE_SCALE)

    def _locate(self, ptr):
        base = ptr & ~((1 << 40) - 1)
        return self._buffers[base], ptr - base

    def copy_bytes(self, dst_ptr, src_ptr, nbytes):
        if not nbytes:
            return
        d, do = self._locate(dst_ptr)
        s, so = self._locate(src_ptr)
        d.flat.view(np.uint8)[do:do + nbytes] = s.flat.view(np.uint8)[so:so + nbytes]

    def upload(self, m, host):
        m.a[:, :m.cols] = host

    def download(self, m):
        return m.a[:, :m.cols].copy()

    def download_f64(self, m, out=None):
        r = m.a[:, :m.cols].astype(np.float64)
        if out is not None:
            out[...] = r
            return out
        return r

    def topk_rows(self, m, k, col0=0, exclude_diag=True, col_ids=None):
        a = m.a[:, :m.cols].astype(np.float32)
        ids = np.arange(col0, col0 + m.cols) if col_ids is None else col_ids.a[0, :m.cols]
        idx = np.full((m.rows, k), -1, dtype=np.int32)
        val = np.zeros((m.rows, k), dtype=np.float32)
        for r in range(m.rows):
            cols = [c for c in range(m.cols) if not (exclude_diag and c == r - col0)]
            cols.sort(key=lambda c: (-a[r, c], ids[c]))
            for j, c in enumerate(cols[:k]):
                idx[r, j], val[r, j] = ids[c], a[r, c]
        return idx, val

    def index_vector(self, values):
        v = np.ascontiguousarray(values, dtype=np.int32).reshape(1, -1)
        m = self.matrix(1, v.shape[1], np.int32, ld=v.shape[1])
        self.upload(m, v)
        return m

    def permute(self, src, dst, row_idx=None, col_idx=None):
        rows = np.arange(dst.rows) if row_idx is None else row_idx.a[0, :dst.rows]
        cols = np.arange(dst.cols) if col_idx is None else col_idx.a[0, :dst.cols]
        dst.a[:, :dst.cols] = src.a[rows][:, cols]

    def synchronize(self):
        pass

    def collective_done(self):
        pass

    def graph(self, csr, rowscale=None, dense_terms=3):
        return NpGraph(csr, rowscale)

    def fill_identity(self, S, col0):
        S.a[:, :S.cols] = 0
        for c in range(S.cols):
            if col0 + c < S.rows:
                S.a[col0 + c, c] = 1

    def spmm(self, g, X, Y, n_cols=None, transpose_out=False, t_block=0, t_pad=0, epilogue=None,
             x_col0=0, y_offset=0):
        self.calls.append(("spmm", transpose_out, bool(epilogue)))
        L = X.cols if n_cols is None else n_cols
        M = g.n_rows
        x = X.a[:g.n_cols, x_col0:x_col0 + L]
        yflat = Y.flat[y_offset:]
        scale = g.rowscale * (np.float32(epilogue["coef"]) if epilogue else np.float32(1))
        v = (g.pattern @ x) * scale[:, None]
        if transpose_out:
            assert epilogue is None
            tb = M if (t_block <= 0 or t_block > M) else t_block
            if tb == M and Y.ld >= M and Y.rows > 1 and not y_offset:
                Y.a[:L, :M] = v.T
                return
            for h in range(-(-M // tb)):
                lo, hi = h * tb, min(M, (h + 1) * tb)
                w = hi - lo + t_pad
                blk = yflat[h * L * (tb + t_pad): h * L * (tb + t_pad) + L * w].reshape(L, w)
                blk[:, :hi - lo] = v[lo:hi].T
            return
        if epilogue:
            v = self._epilogue(v, epilogue, M, L)
        Y.a[:M, :L] = v

    def spmm_shard(self, g, X, Y, epilogue, rank, world, send, chunk_floats):
        T = g.n_rows // world // 32
        self.spmm_shard_stage(g, X, Y, epilogue, rank, world, send, 0, chunk_floats, 0, T, True)

    def spmm_shard_stage(self, g, X, Y, epilogue, rank, world, send, send_off, chunk_floats, tile_lo, tile_hi,
                         zero_counters):
        """simrank_spmm_shard(_stage): only tiles i <= j (of the stage's column tiles) are computed; what is
        not written keeps its junk; the counters accumulate over the stages of an update."""
        self.calls.append(("spmm_shard",))
        M = g.n_rows
        mb = M // world
        T = mb // 32
        assert mb * world == M and T * 32 == mb and epilogue["diag_col0"] == rank * mb
        slot0 = tile_lo * (tile_lo - 1) // 2
        assert 0 <= tile_lo < tile_hi <= T and chunk_floats >= (tile_hi * (tile_hi - 1) // 2 - slot0) * 1024
        x = X.a[:g.n_cols, :mb]
        v = (g.pattern @ x) * (g.rowscale * np.float32(epilogue["coef"]))[:, None]
        old_changed = self._changed
        v = self._epilogue(v, dict(epilogue, previous=None), M, mb)
        prev, eps = epilogue.get("previous"), epilogue.get("eps", 0.0)
        moved = (np.abs(v.astype(np.float64) - prev.a[:M, :mb].astype(np.float64)) > eps
                 if prev is not None else np.zeros((M, mb), bool))
        changed = 0 if zero_counters else old_changed
        for h in range(world):
            chunk = send.flat[send_off + h * chunk_floats:send_off + (h + 1) * chunk_floats]
            for i in range(T):
                rows = slice(h * mb + 32 * i, h * mb + 32 * i + 32)
                for j in range(max(i, tile_lo), tile_hi):
                    cols = slice(32 * j, 32 * j + 32)
                    Y.a[rows, cols] = v[rows, cols]
                    changed += int(moved[rows, cols].sum()) * (2 if i < j else 1)
                    if i == j:
                        continue
                    if h == rank:
                        Y.a[rank * mb + 32 * j: rank * mb + 32 * j + 32, 32 * i:32 * i + 32] = v[rows, cols].T
                    else:
                        slot = j * (j - 1) // 2 + i - slot0
                        chunk[slot * 1024:(slot + 1) * 1024] = v[rows, cols].T.reshape(-1)
        self._changed = changed if prev is not None else old_changed

    def shard_unpack(self, Y, recv, chunk_floats, rank, world, n_rows):
        self.shard_unpack_stage(Y, recv, 0, chunk_floats, rank, world, n_rows, 0, n_rows // world // 32)

    def shard_unpack_stage(self, Y, recv, recv_off, chunk_floats, rank, world, n_rows, tile_lo, tile_hi):
        self.calls.append(("shard_unpack",))
        mb = n_rows // world
        slot0 = tile_lo * (tile_lo - 1) // 2
        for h in range(world):
            if h == rank:
                continue
            chunk = recv.flat[recv_off + h * chunk_floats:recv_off + (h + 1) * chunk_floats]
            for j in range(tile_lo, tile_hi):
                for i in range(j):
                    slot = j * (j - 1) // 2 + i - slot0
                    Y.a[h * mb + 32 * j: h * mb + 32 * j + 32, 32 * i:32 * i + 32] = \
                        chunk[slot * 1024:(slot + 1) * 1024].reshape(32, 32)

    def _epilogue(self, v, ep, M, L):
        v = v.astype(np.float32)
        if ep.get("evidence") is not None:
            cnt = ep["evidence"].a[:M, :L].astype(np.int32)
            v = v * (np.float32(1) - np.ldexp(np.float32(1), -cnt)).astype(np.float32)
        if ep.get("apriori") is not None:
            lbd = np.float32(ep["lbd"])
            v = (np.float32(1) - lbd) * v + lbd * ep["apriori"].a[:M, :L]
        if ep.get("set_diag", True):
            d0 = ep.get("diag_col0", 0)
            for c in range(L):
                if 0 <= d0 + c < M:
                    v[d0 + c, c] = 1
        if ep.get("previous") is not None:
            old = ep["previous"].a[:M, :L]
            self._changed = int((np.abs(v.astype(np.float64) - old.astype(np.float64))
                                 > ep["eps"]).sum())
        return v

    def epilogue_apply(self, Q, Y, n_rows, n_cols, epilogue):
        self.calls.append(("epilogue_apply",))
        v = Q.a[:n_rows, :n_cols] * np.float32(epilogue["coef"])
        Y.a[:n_rows, :n_cols] = self._epilogue(v, epilogue, n_rows, n_cols)

    def gemm_nt(self, A, B, Cm, M, N, K, epilogue=None):
        self.calls.append(("gemm_nt", bool(epilogue)))
        v = A.a[:M, :K] @ B.a[:N, :K].T
        if epilogue:
            v = self._epilogue(v * np.float32(epilogue["coef"]), epilogue, M, N)
        Cm.a[:M, :N] = v

    def densify(self, g, Wd):
        Wd.a[:, :] = 0
        Wd.a[:g.n_rows, :g.n_cols] = (sp.diags(g.rowscale) @ g.pattern).toarray()

    def evidence_counts(self, g, col0, out):
        live = sp.diags((g.rowscale > 0).astype(np.float32)) @ g.pattern
        cnt = (live @ live.T).toarray()
        out.a[:, :out.cols] = np.minimum(cnt[:, col0:col0 + out.cols], 255).astype(np.uint8)

    def evidence_live_fraction(self, counts):
        a = counts.a[:, :counts.cols]
        segs = -(-counts.cols // 32)
        live = sum(int((a[:, 32 * k:32 * k + 32] != 0).any(axis=1).sum()) for k in range(segs))
        return live / max(1, counts.rows * segs)

    def read_changed(self):
        return self._changed

    def event(self):
        return 0

    def event_synchronize(self, ev):
        pass

    def event_destroy(self, ev):
        pass

    def record(self, ev):
        pass

    def elapsed_ms(self, a, b):
        return 0.0
