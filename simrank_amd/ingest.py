"""Edge list -> node order + CSR (host side of the drop-in boundary).

Replaces ``_create_graph`` (SimRank.py:24-52 directed, :168-200 / :376-391 bipartite),
``_cal_Weight`` (:322-337) and the pattern half of ``_cal_Evidence`` (:311-320), without
ever forming an N x N array: the normalised adjacency the reference stores densely is
``diag(rowscale) . A`` with A the 0/1 edge pattern, because every value it writes is
1/in-degree (or 1/sum of weights) of the ROW node (quirks Q3, Q9 of SURVEY.md).

Everything here is vectorised pandas/NumPy; no Python loop over edges or nodes.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import pandas as pd


@dataclass
class CSR:
    """W = diag(rowscale) . pattern; rows = target/own-group nodes."""
    n_rows: int
    n_cols: int
    rowptr: np.ndarray      # int32 [n_rows + 1]
    col: np.ndarray         # int32 [nnz], ascending inside a row
    rowscale: np.ndarray    # float64 [n_rows]; 0 for rows without entries

    @property
    def nnz(self) -> int:
        return int(self.col.size)

    @property
    def density(self) -> float:
        return self.nnz / float(self.n_rows * self.n_cols)

    def dense(self, scale: np.ndarray | None = None) -> np.ndarray:
        """The float64 matrix the reference keeps (``Graph`` / ``Weight``)."""
        rs = self.rowscale if scale is None else scale
        out = np.zeros((self.n_rows, self.n_cols))
        rows = np.repeat(np.arange(self.n_rows), np.diff(self.rowptr))
        out[rows, self.col] = rs[rows]
        return out

    def with_scale(self, rowscale: np.ndarray) -> "CSR":
        return CSR(self.n_rows, self.n_cols, self.rowptr, self.col, rowscale)


def relabel(csr: CSR, row_order: np.ndarray | None, col_order: np.ndarray | None) -> CSR:
    """The same matrix with its rows and columns taken in another order: row i of the result is
    row ``row_order[i]`` of ``csr`` and its column j is column ``col_order[j]`` (None = as they
    are).  Column ids stay ascending inside a row."""
    M = csr.n_rows
    rows = np.arange(M) if row_order is None else np.asarray(row_order, dtype=np.int64)
    lens = np.diff(csr.rowptr)[rows]
    rowptr = np.zeros(M + 1, dtype=np.int64)
    np.cumsum(lens, out=rowptr[1:])
    # positions of the entries of the selected rows, in the new row order
    src = np.repeat(csr.rowptr[:-1][rows].astype(np.int64) - rowptr[:-1], lens) + np.arange(rowptr[-1])
    col = csr.col[src].astype(np.int64)
    if col_order is not None:
        new_id = np.empty(csr.n_cols, dtype=np.int64)
        new_id[np.asarray(col_order, dtype=np.int64)] = np.arange(csr.n_cols)
        col = new_id[col]
        bits = max(1, int(csr.n_cols - 1).bit_length())          # (row, column) keys, the column in the low bits
        key = (np.repeat(np.arange(M, dtype=np.int64), lens) << bits) | col
        key.sort()
        col = key & ((1 << bits) - 1)
    return CSR(M, csr.n_cols, rowptr.astype(np.int32), col.astype(np.int32), csr.rowscale[rows])


def _reciprocal(x) -> np.ndarray:
    """1/x with +-inf replaced by 0 (SimRank.py:49, :197-198)."""
    with np.errstate(divide="ignore"):
        r = 1.0 / np.asarray(x, dtype=np.float64)
    r[np.isinf(r)] = 0.0
    return np.nan_to_num(r, nan=0.0, posinf=0.0, neginf=0.0)


def _csr(rows: np.ndarray, cols: np.ndarray, n_rows: int, n_cols: int,
         row_value: np.ndarray) -> CSR:
    # one sort of (row, column) keys; the column sits in the low bits (shifts and masks instead of divisions)
    bits = max(1, int(n_cols - 1).bit_length())
    # (32-bit keys whenever row and column fit together — up to 65536 x 65536, every BASELINE configuration —: the sort is
    # half of this function and takes half the time on 4-byte keys)
    kt = np.uint32 if bits + max(1, int(n_rows - 1).bit_length()) <= 32 else np.int64
    key = (rows.astype(kt) << kt(bits)) | cols.astype(kt)
    key.sort()
    if key.size > 1 and (key[1:] == key[:-1]).any():
        # what DataFrame.pivot raises on a repeated (index, column) pair (SimRank.py:50)
        raise ValueError("Index contains duplicate entries, cannot reshape")
    rowptr = np.zeros(n_rows + 1, dtype=np.int64)
    np.cumsum(np.bincount(key >> kt(bits), minlength=n_rows), out=rowptr[1:])
    if rowptr[-1] >= 2**31:
        raise ValueError("more than 2^31 edges")
    return CSR(n_rows, n_cols, rowptr.astype(np.int32), (key & kt((1 << bits) - 1)).astype(np.int32),
               row_value)


def _int_codes(nodes, src, dst):
    """Positions of the edge endpoints in ``nodes`` through a lookup table, when both columns hold integers without
    gaps worth mentioning; None otherwise (any other labels go through a pandas Index)."""
    a, b = src.to_numpy(), dst.to_numpy()
    if not (a.dtype.kind in "iu" and b.dtype.kind in "iu" and len(nodes)):
        return None
    arr = np.asarray(nodes)
    if arr.dtype.kind not in "iu":
        return None
    lo, hi = int(arr.min()), int(arr.max())
    if hi - lo > 16 * len(nodes) + 4096:
        return None
    lut = np.full(hi - lo + 1, -1, dtype=np.int32)           # (positions fit 32 bits: half the table, half the gathers' bytes)
    lut[arr.astype(np.int64, copy=False) - lo] = np.arange(len(nodes), dtype=np.int32)

    def place(x):                                             # (no copies where none is needed: int64 labels from 0)
        x = x.astype(np.int64, copy=False)
        return lut[x if lo == 0 else x - lo]
    ca = place(a)
    return ca, (ca if dst is src else place(b))


def directed(data: pd.DataFrame, weighted: bool, from_node_column: str, to_node_column: str,
             weight_column: str):
    """-> (nodes, CSR).  ``nodes`` is the list in the iteration order of the Python set
    the reference builds at SimRank.py:42 (quirk Q0: that order *is* the row/column order
    of every matrix and of the returned DataFrame)."""
    src = data[from_node_column]
    dst = data[to_node_column]
    nodes = list(set(src.unique()) | set(dst.unique()))
    n = len(nodes)
    codes = _int_codes(nodes, src, dst)
    if codes is not None:
        # integer labels in a compact range (the usual edge list): positions through a lookup table instead of two hash
        # joins, in-degrees by counting — the same numbers (1/count is exact either way)
        src_c, dst_c = codes
        index = None
    else:
        index = pd.Index(nodes)
        src_c, dst_c = index.get_indexer(src), index.get_indexer(dst)
    if weighted or index is not None:
        if index is None:
            index = pd.Index(nodes)
        if weighted:
            per_target = data.groupby(to_node_column)[weight_column].sum()
        else:
            per_target = data.groupby(to_node_column)[from_node_column].count()
        rowscale = np.zeros(n)
        rowscale[index.get_indexer(per_target.index)] = _reciprocal(per_target.to_numpy())
    else:
        rowscale = _reciprocal(np.bincount(dst_c, minlength=n)) if n else np.zeros(0)
    return nodes, _csr(dst_c, src_c, n, n, rowscale)


def bipartite(data: pd.DataFrame, weighted: bool, node_group1_column: str,
              node_group2_column: str, weight_column: str):
    """-> (set_order1, set_order2, sorted1, sorted2, CSR12, CSR21).

    The reference's matrices come out of ``pivot`` (SimRank.py:199-200), i.e. in sorted
    label order, while the labels it later attaches are in Python-set order (quirk Q1)."""
    g1 = data[node_group1_column]
    g2 = data[node_group2_column]
    u1, u2 = g1.unique(), g2.unique()
    set1 = list(set(u1))
    set2 = list(set(u2))
    sorted1 = pd.Index(np.sort(u1), name=node_group1_column)
    sorted2 = pd.Index(np.sort(u2), name=node_group2_column)
    n1, n2 = len(sorted1), len(sorted2)
    c1 = _int_codes(sorted1.to_numpy(), g1, g1)
    c2 = _int_codes(sorted2.to_numpy(), g2, g2)
    w = data[weight_column].to_numpy() if weighted else None
    if c1 is not None and c2 is not None and (w is None or w.dtype.kind in "iu"):
        # integer labels in compact ranges and counts / integer weights: lookup tables and exact integer sums instead of
        # hash joins and group-bys (the same numbers: every sum is exact)
        i1, i2 = c1[0], c2[0]
        if w is None:
            rs1 = _reciprocal(np.bincount(i1, minlength=n1))
            rs2 = _reciprocal(np.bincount(i2, minlength=n2))
        else:
            rs1 = _reciprocal(np.bincount(i1, weights=w, minlength=n1))
            rs2 = _reciprocal(np.bincount(i2, weights=w, minlength=n2))
    else:
        if weighted:
            d1 = data.groupby(node_group1_column)[weight_column].sum()
            d2 = data.groupby(node_group2_column)[weight_column].sum()
        else:
            d1 = data.groupby(node_group1_column)[node_group2_column].count()
            d2 = data.groupby(node_group2_column)[node_group1_column].count()
        rs1 = np.zeros(n1)
        rs1[sorted1.get_indexer(d1.index)] = _reciprocal(d1.to_numpy())
        rs2 = np.zeros(n2)
        rs2[sorted2.get_indexer(d2.index)] = _reciprocal(d2.to_numpy())
        i1 = sorted1.get_indexer(g1)
        i2 = sorted2.get_indexer(g2)
    return (set1, set2, sorted1, sorted2,
            _csr(i1, i2, n1, n2, rs1), _csr(i2, i1, n2, n1, rs2))


def spread(csr: CSR) -> np.ndarray:
    """exp(-sample variance (ddof=1) of the non-zero entries of each row), NaN -> 0
    (SimRank.py:326-332, quirk Q4).  Entries that are exactly 0 (rowscale 0) are not
    "non-zero entries"; a row with fewer than two has variance NaN -> spread 1."""
    deg = np.diff(csr.rowptr).astype(np.float64)
    live = np.where(csr.rowscale != 0, deg, 0.0)
    with np.errstate(invalid="ignore", divide="ignore"):
        total = csr.rowscale * live
        mean = np.where(live > 0, total / np.maximum(live, 1), 0.0)
        var = live * (csr.rowscale - mean) ** 2 / (live - 1)
    var = np.where(live > 1, var, 0.0)
    return np.exp(-var)


def partition(n: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous block [lo, hi) of rank ``rank``; every block but the last has
    ceil(n/world) elements (the layout simrank_spmm's transposed store assumes)."""
    b = -(-n // world)
    lo = min(n, rank * b)
    return lo, min(n, lo + b)
