// Block-dense part of the sparse legs on the matrix cores (gfx950, v_mfma_f32_32x32x16_bf16).
//
// A power-law pattern sorted by row length has a corner that is dense enough for MFMA: in the
// bench graph (N = 32768, 783 k entries) the column sets "referenced by >= 4 of a block's 128
// rows", kept for the blocks that have 128 such columns, hold 39 % of all entries in 0.3 % of
// the matrix.  Gathering those entries costs one 128-byte L2 request each per 32-column panel;
// multiplying the block by the operand rows of its dense set reads every such row ONCE per block.  The products must stay exact f32 (parity bar 1e-5 on f32
// results, SimRank.py:139), and gfx950's f32 MFMA peaks at 157 TFLOP/s, so the operand is
// split on the fly into three bf16 terms, x = hi + mid + lo EXACTLY (8 + 8 + 8 mantissa bits,
// truncation split), and the 0/1 pattern is exact in bf16: three bf16 MFMAs accumulate in f32
// what one f32 MFMA would, at 2.5 PFLOP/s / 3 instead of 157 TFLOP/s.
//
//   P[t*128 + r][c] = sum_{k in dense set of tile t} A[row0(t) + r][k] * X[k][c]      (raw sums)
//
// The gather kernel (spmm.hip) then runs on the remainder pattern and adds P before its
// epilogue.  A dense set is cut into units of <= 2048 columns (one slab of P each, added in a
// fixed order).  One workgroup = one unit x 256 columns, one wave = 128 rows x 64 columns:
// accumulators 4 x 2 MFMA tiles (128 registers); the B operand goes global -> registers ->
// three bf16 fragments without LDS (every wave owns its columns), the A operand is stored on
// the device already in MFMA fragment order.  Deterministic: fixed k order, no atomics.
// Measured: 0.93 ms / 760 TFLOP/s on the bench graph, where it is bound by
// the HBM stream of the operand rows (192 flop per byte at 128-row blocks), 1.23 PFLOP/s when
// every entry of the graph is sent to it.
#include <algorithm>
#include <numeric>
#include <vector>

#include <thread>

#include "common.h"

namespace simrank {

constexpr int kDM = 128;   // rows per dense tile
constexpr int kDN = 256;   // columns per workgroup (64 per wave)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct DenseArgs {
    const float* X;
    int64_t ldx, L;
    int64_t xpad;             // > 0: X is panel-blocked with this many rows per panel (ldx = 32)
    float* P;
    int64_t ldp;
    const int32_t* unit_row0;
    const int32_t* unit_slab;
    const int32_t* unit_kofs;
    const int32_t* dcols;
    const uint4* afrag;
    int32_t n_units, n_cblocks, tri;
    int32_t wg_cols;          // columns per workgroup: 64 per wave, 4 waves or 1
};

// (x0, x1) -> packed bf16 pairs of the three terms of the truncation split
__device__ __forceinline__ void split3(float x0, float x1, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
    const uint32_t u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
    hi = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
    const float r0 = x0 - __uint_as_float(u0 & 0xFFFF0000u);
    const float r1 = x1 - __uint_as_float(u1 & 0xFFFF0000u);
    const uint32_t v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
    mid = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
    const float q0 = r0 - __uint_as_float(v0 & 0xFFFF0000u);
    const float q1 = r1 - __uint_as_float(v1 & 0xFFFF0000u);
    lo = __builtin_amdgcn_perm(__float_as_uint(q1), __float_as_uint(q0), 0x07060302u);
}

__device__ __forceinline__ bf16x8 as_bf16x8(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    const uint4 v = make_uint4(a, b, c, d);
    return __builtin_bit_cast(bf16x8, v);
}

// (x0, x1) -> packed fp16 pair, round to nearest even (the reduced-precision form, TERMS = 1)
__device__ __forceinline__ uint32_t pack_f16(float x0, float x1) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 v = {(_Float16)x0, (_Float16)x1};
    return __builtin_bit_cast(uint32_t, v);
}

#ifndef SIMRANK_HOST_ONLY          // (the sanitizer build of the host logic has no device code: common.h)
// TERMS = 3: exact f32 products (operand = hi + mid + lo in bf16).  TERMS = 1: the operand rounded
// to ONE fp16 term (11 significant bits; S lies in [0, 1], fp16 subnormals reach 6e-8), one MFMA
// instead of three and no split arithmetic — BASELINE.json's "fp16 MFMA dense leg" (config 5).
// Not within the 1e-5 parity bar: selected only on request (tuning "dense_terms").
template <int TERMS>
__global__ __launch_bounds__(256, 2) void dense_tiles_kernel(const DenseArgs p) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    // blocks equal mod 8 share an XCD: a column block stays on one XCD, whose L2 then serves the
    // operand rows that the dense sets of different tiles have in common
    const int x = blockIdx.x & 7;
    const int local = blockIdx.x >> 3;
    const int cb = (local / p.n_units) * 8 + x;
    const int t = local % p.n_units;
    if (cb >= p.n_cblocks) return;
    const int row0 = p.unit_row0[t];
    // upper-triangle form of leg 2: row block rb needs the columns >= rb only
    if (p.tri && (int64_t(cb) + 1) * p.wg_cols <= row0) return;
    const int64_t wcol = int64_t(cb) * p.wg_cols + wave * 64;
    if (wcol >= p.L) return;                       // no barrier in this kernel: a wave may leave
    const int64_t col = wcol + 2 * r;
    const bool col_ok = col < p.L;                 // (col + 1 may be L: inside the padded row)
    // row-major X: element (i, c) at i * ldx + c; panel-blocked (xpad > 0): ((c >> 5) * xpad + i) * 32 + (c & 31)
    const int64_t cc = col_ok ? col : 0;
    const float* __restrict__ Xc = p.xpad ? p.X + ((cc >> 5) * p.xpad) * 32 + (cc & 31) : p.X + cc;
    const int k0 = p.unit_kofs[t], k1 = p.unit_kofs[t + 1];

    f32x16 acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;

    const uint4* __restrict__ ap = p.afrag + size_t(k0 >> 4) * 4 * 64 + lane;
    float2 raw[8];
    uint4 af[4];
    int ids[16];                                   // operand rows of the step issued next (SGPRs)
    auto fetch_ids = [&](int k) {
#pragma unroll
        for (int j = 0; j < 16; ++j) ids[j] = __builtin_amdgcn_readfirstlane(p.dcols[k + j]);
    };
    auto issue = [&](int k) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int id = h ? ids[8 + j] : ids[j];
            raw[j] = *reinterpret_cast<const float2*>(Xc + int64_t(id) * p.ldx);
        }
        const uint4* a = ap + size_t((k - k0) >> 4) * 4 * 64;
#pragma unroll
        for (int m = 0; m < 4; ++m) af[m] = a[m * 64];
    };
    fetch_ids(k0);
    issue(k0);
    if (k0 + 16 < k1) fetch_ids(k0 + 16);
    for (int k = k0; k < k1; k += 16) {
        float2 cur[8];
        uint4 ac[4];
#pragma unroll
        for (int j = 0; j < 8; ++j) cur[j] = raw[j];
#pragma unroll
        for (int m = 0; m < 4; ++m) ac[m] = af[m];
        if (k + 16 < k1) {                         // next step's operands fly during the MFMAs
            issue(k + 16);
            if (k + 32 < k1) fetch_ids(k + 32);
        }
        if constexpr (TERMS == 3) {
            uint32_t hi[2][4], mid[2][4], lo[2][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                split3(cur[2 * j].x, cur[2 * j + 1].x, hi[0][j], mid[0][j], lo[0][j]);
                split3(cur[2 * j].y, cur[2 * j + 1].y, hi[1][j], mid[1][j], lo[1][j]);
            }
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const bf16x8 bh = as_bf16x8(hi[n][0], hi[n][1], hi[n][2], hi[n][3]);
                const bf16x8 bm = as_bf16x8(mid[n][0], mid[n][1], mid[n][2], mid[n][3]);
                const bf16x8 bl = as_bf16x8(lo[n][0], lo[n][1], lo[n][2], lo[n][3]);
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const bf16x8 a = __builtin_bit_cast(bf16x8, ac[m]);
                    // smallest term first: the accumulator sees the low-order parts before the
                    // high-order ones of the same step
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bl, acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bm, acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bh, acc[m][n], 0, 0, 0);
                }
            }
        } else {
            // the pattern image holds bf16 1.0 (0x3F80); fp16 1.0 is 0x3C00 = 0x3F80 & 0x3C00
            f16x8 a16[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const uint4 v = make_uint4(ac[m].x & 0x3C003C00u, ac[m].y & 0x3C003C00u,
                                           ac[m].z & 0x3C003C00u, ac[m].w & 0x3C003C00u);
                a16[m] = __builtin_bit_cast(f16x8, v);
            }
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                uint4 b;
                if (n == 0)
                    b = make_uint4(pack_f16(cur[0].x, cur[1].x), pack_f16(cur[2].x, cur[3].x),
                                   pack_f16(cur[4].x, cur[5].x), pack_f16(cur[6].x, cur[7].x));
                else
                    b = make_uint4(pack_f16(cur[0].y, cur[1].y), pack_f16(cur[2].y, cur[3].y),
                                   pack_f16(cur[4].y, cur[5].y), pack_f16(cur[6].y, cur[7].y));
                const f16x8 b16 = __builtin_bit_cast(f16x8, b);
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a16[m], b16, acc[m][n], 0, 0, 0);
            }
        }
    }
    if (!col_ok) return;
    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    float* Pw = p.P + (int64_t(p.unit_slab[t]) * kDM) * p.ldp + col;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = 32 * m + (i & 3) + 8 * (i >> 2) + 4 * h;
            *reinterpret_cast<float2*>(Pw + int64_t(row) * p.ldp) = make_float2(acc[m][0][i], acc[m][1][i]);
        }
}

#endif  // SIMRANK_HOST_ONLY

template <typename T>
static int upload(T** d, const std::vector<T>& h) {
    const size_t bytes = std::max<size_t>(16, h.size() * sizeof(T));
    SR_HIP(plan_alloc((void**)d, bytes));
    if (!h.empty()) SR_HIP(plan_upload(*d, h.data(), h.size() * sizeof(T)));
    return SIMRANK_OK;
}

void free_dense_plan(simrank_dense_plan* p) {
    if (!p) return;
    plan_free(p->unit_row0); plan_free(p->unit_slab); plan_free(p->unit_kofs);
    plan_free(p->dcols); plan_free(p->afrag); plan_free(p->block_slab0);
    plan_free(p->block_nslab); plan_free(p->r_rowptr);
    plan_free(p->r_col); plan_free(p->r_col16); plan_free(p->r_tile_row0); plan_free(p->r_sym_map);
    (void)hipFree(p->part);
    delete p;
}

constexpr int kUnitCols = 2048;   // a dense set is cut into units of about this many columns, or more
                                  // when there is enough work for every CU without cutting

// Host side: pick the dense sets, cut them into units, lay the pattern out in A-fragment order,
// build the remainder.
int build_dense_plan(simrank_graph* g, const int32_t* rowptr, const int32_t* col) {
    const int64_t M = g->n_rows, K = g->n_cols;
    const int64_t min_rows = g->tun.dense_min, min_cols = g->tun.dense_cols;
    const int64_t nblk = (M + kDM - 1) / kDM;
    struct Unit { int32_t block, slab, first, count; };        // columns [first, first+count) of the block's set
    std::vector<std::vector<int32_t>> sets(static_cast<size_t>(nblk));
    std::vector<Unit> units;
    std::vector<int32_t> block_slab0(size_t(nblk), 0), block_nslab(size_t(nblk), 0);
    std::vector<uint16_t> cnt(size_t(K), 0);
    std::vector<int32_t> touched;
    int64_t covered = 0, total_k = 0, set_cols = 0;
    int32_t n_slabs = 0, n_blocks_dense = 0;
    for (int64_t b = 0; b < nblk; ++b) {
        const int64_t lo = b * kDM, hi = std::min<int64_t>(M, lo + kDM);
        touched.clear();
        for (int32_t j = rowptr[lo]; j < rowptr[hi]; ++j)
            if (cnt[col[j]]++ == 0) touched.push_back(col[j]);
        std::vector<int32_t>& set = sets[size_t(b)];
        int64_t cov = 0;
        for (int32_t c : touched)
            if (cnt[c] >= min_rows) { set.push_back(c); cov += cnt[c]; }
        for (int32_t c : touched) cnt[c] = 0;
        if ((int64_t)set.size() < min_cols) { set.clear(); continue; }
        std::sort(set.begin(), set.end());
        covered += cov;
        ++n_blocks_dense;
        set_cols += (int64_t)set.size();
    }
    // (a single-GPU plan whose leg 1 is the one-launch kernel: the only taker would be the upper-triangle leg 2, and it takes
    // the plan only under this rule — spmm.hip, `dp && want_sym`)
    if (g->tun.dense_lazy && !(g->tun.dense_sym > 0 || (g->tun.dense_sym < 0 && 2 * covered >= g->nnz))) return SIMRANK_OK;
    // ... and (round 6) not where that leg is the one-launch kernel's (fused.hip, SYM): the one-launch plan, built on a thread
    // beside this one, says so once it is there (spmm.hip dispatches on the same rule; a launch the rule's shapes exclude
    // falls back to the plain gather, which needs no plan)
    if (g->tun.dense_lazy && g->tun.fuse && g->tun.fuse_sym != 0 && g->tun.dense_terms == 3 && g->n_rows >= 64) {
        while (g->fused_build.load(std::memory_order_acquire) == 1) std::this_thread::yield();
        if (g->fused_build.load(std::memory_order_acquire) == 2 && g->fused &&
            (g->tun.fuse_sym > 0 || 2 * g->fused->nnz_covered >= g->nnz))
            return SIMRANK_OK;
    }
    // Units: a workgroup per (unit, 256 output columns); an XCD works through a column block with
    // 64 resident workgroups, so no unit should be longer than 1/64 of the column block's work —
    // and none shorter than kUnitCols, because every unit costs a slab of partial sums.
    const int64_t unit_cols = std::max<int64_t>(kUnitCols, (set_cols / 64 + 15) / 16 * 16);
    for (int64_t b = 0; b < nblk; ++b) {
        const std::vector<int32_t>& set = sets[size_t(b)];
        if (set.empty()) continue;
        const int32_t n = (int32_t)set.size();
        const int32_t nu = (int32_t)((n + unit_cols - 1) / unit_cols);
        block_slab0[size_t(b)] = n_slabs;
        block_nslab[size_t(b)] = nu;
        for (int32_t u = 0; u < nu; ++u) {
            const int32_t first = int32_t(int64_t(n) * u / nu), last = int32_t(int64_t(n) * (u + 1) / nu);
            units.push_back({(int32_t)b, n_slabs + u, first, last - first});
            total_k += (last - first + 15) / 16 * 16;
        }
        n_slabs += nu;
    }
    // budget: the fragment image takes 256 bytes per dense column (8 GiB ~ a fully dense 32768 x 32768)
    if (units.empty() || total_k * 256 > (int64_t(8) << 30)) return SIMRANK_OK;
    std::stable_sort(units.begin(), units.end(),
                     [](const Unit& a, const Unit& b) { return a.count > b.count; });

    const size_t nu = units.size();
    std::vector<int32_t> unit_row0(nu), unit_slab(nu), unit_kofs(nu + 1, 0), dcols(size_t(total_k), 0);
    std::vector<uint16_t> afrag(size_t(total_k) / 16 * 4 * 64 * 8, 0);
    // position of (block, column of its set) in the dcols / afrag images
    std::vector<std::vector<int32_t>> set_pos(static_cast<size_t>(nblk));
    for (size_t b = 0; b < size_t(nblk); ++b) set_pos[b].assign(sets[b].size(), -1);
    for (size_t u = 0; u < nu; ++u) {
        const Unit& un = units[u];
        unit_row0[u] = un.block * kDM;
        unit_slab[u] = un.slab;
        unit_kofs[u + 1] = unit_kofs[u] + (un.count + 15) / 16 * 16;
        for (int32_t q = 0; q < un.count; ++q) {
            dcols[size_t(unit_kofs[u] + q)] = sets[size_t(un.block)][size_t(un.first + q)];
            set_pos[size_t(un.block)][size_t(un.first + q)] = unit_kofs[u] + q;
        }
    }
    std::vector<int32_t> r_rowptr(size_t(M) + 1, 0), r_col;
    r_col.reserve(size_t(g->nnz - covered));
    std::vector<int32_t> kpos(size_t(K), -1);
    int32_t r_max = 0;
    for (int64_t b = 0; b < nblk; ++b) {
        const int64_t lo = b * kDM, hi = std::min<int64_t>(M, lo + kDM);
        const std::vector<int32_t>& set = sets[size_t(b)];
        for (size_t q = 0; q < set.size(); ++q) kpos[set[q]] = set_pos[size_t(b)][q];
        for (int64_t a = lo; a < hi; ++a) {
            for (int32_t j = rowptr[a]; j < rowptr[a + 1]; ++j) {
                const int32_t q = kpos[col[j]];
                if (q < 0) {
                    r_col.push_back(col[j]);
                } else {
                    // element j of lane (h, r): A[row 32 mb + r][k = 16 s + 8 h + j]
                    const int rr = int(a - lo);
                    const int kk = q & 15;
                    const int lane = (kk >> 3) * 32 + (rr & 31);
                    afrag[((size_t(q >> 4) * 4 + size_t(rr >> 5)) * 64 + lane) * 8 + (kk & 7)] = 0x3F80;  // bf16 1.0
                }
            }
            r_rowptr[size_t(a) + 1] = (int32_t)r_col.size();
            r_max = std::max(r_max, r_rowptr[size_t(a) + 1] - r_rowptr[size_t(a)]);
        }
        for (int32_t c : set) kpos[c] = -1;
    }
    std::vector<int32_t> r_tile_row0, r_sym_map;
    const int64_t r_n_tiles = build_tiles(r_rowptr.data(), M, (int64_t)r_col.size(), g->tun.balance,
                                          r_tile_row0, r_sym_map, g->tun.sym_desc != 0);

    simrank_dense_plan* pl = new simrank_dense_plan;
    pl->n_units = (int32_t)nu;
    pl->n_slabs = n_slabs;
    pl->n_blocks_dense = n_blocks_dense;
    pl->total_k = total_k;
    pl->nnz_covered = covered;
    pl->r_nnz = (int64_t)r_col.size();
    pl->r_max_row = r_max;
    pl->r_n_tiles = (int32_t)r_n_tiles;
    pl->r_sym_blocks = (int32_t)(r_sym_map.size() / 2);
    int rc = upload(&pl->unit_row0, unit_row0);
    if (!rc) rc = upload(&pl->unit_slab, unit_slab);
    if (!rc) rc = upload(&pl->unit_kofs, unit_kofs);
    if (!rc) rc = upload(&pl->dcols, dcols);
    if (!rc) rc = upload(reinterpret_cast<uint16_t**>(&pl->afrag), afrag);
    if (!rc) rc = upload(&pl->block_slab0, block_slab0);
    if (!rc) rc = upload(&pl->block_nslab, block_nslab);
    if (!rc) rc = upload(&pl->r_rowptr, r_rowptr);
    if (!rc) rc = upload(&pl->r_col, r_col);
    if (!rc && K <= 65536 && !r_col.empty()) {
        std::vector<uint16_t> c16(r_col.begin(), r_col.end());
        rc = upload(&pl->r_col16, c16);
    }
    if (!rc && pl->r_n_tiles) rc = upload(&pl->r_tile_row0, r_tile_row0);
    if (!rc && pl->r_sym_blocks) rc = upload(&pl->r_sym_map, r_sym_map);
    if (rc) {
        free_dense_plan(pl);
        return rc;
    }
    g->dense = pl;
    return SIMRANK_OK;
}

int launch_dense_tiles(const simrank_graph* g, const float* X, int64_t ldx, int64_t L, bool tri,
                       hipStream_t st, DenseUse* use) {
    simrank_dense_plan* pl = g->dense;
    SR_REQUIRE(pl && pl->n_units > 0, "graph has no dense plan");
    // (ldx < 0: X is panel-blocked with -ldx rows per panel)
    SR_REQUIRE((reinterpret_cast<uintptr_t>(X) & 7u) == 0 && ldx % 2 == 0,
               "dense tiles need an 8-byte aligned operand");
    const int64_t ldp = (L + 63) / 64 * 64;
    const size_t need = size_t(pl->n_slabs) * kDM * size_t(ldp);
    if (need > pl->part_cap) {
        if (pl->part) {
            SR_HIP(hipStreamSynchronize(st));      // an earlier launch may still read it
            SR_HIP(hipFree(pl->part));
            pl->part = nullptr;
            pl->part_cap = 0;
        }
        SR_HIP(hipMalloc((void**)&pl->part, need * sizeof(float)));
        pl->part_cap = need;
    }
    DenseArgs a{};
    a.X = X; a.ldx = ldx < 0 ? 32 : ldx; a.L = L;
    a.xpad = ldx < 0 ? -ldx : 0;
    a.P = pl->part; a.ldp = ldp;
    a.unit_row0 = pl->unit_row0; a.unit_slab = pl->unit_slab; a.unit_kofs = pl->unit_kofs;
    a.dcols = pl->dcols; a.afrag = pl->afrag;
    a.n_units = pl->n_units;
    // narrow launches (a stage of a sharded leg): one wave per workgroup, four times the workgroups
    a.wg_cols = ((L + kDN - 1) / kDN) * pl->n_units >= 2048 ? kDN : 64;
    a.n_cblocks = (int32_t)((L + a.wg_cols - 1) / a.wg_cols);
    a.tri = tri ? 1 : 0;
    const int64_t grid = int64_t((a.n_cblocks + 7) / 8) * 8 * a.n_units;
    SR_REQUIRE(grid > 0 && grid < (int64_t(1) << 31), "grid of %lld blocks", (long long)grid);
#ifdef SIMRANK_HOST_ONLY
    SR_REQUIRE(false, "host-only build: no kernels");
#else
    if (g->tun.dense_terms == 1)
        hipLaunchKernelGGL(dense_tiles_kernel<1>, dim3((unsigned)grid), dim3((unsigned)a.wg_cols), 0, st, a);
    else
        hipLaunchKernelGGL(dense_tiles_kernel<3>, dim3((unsigned)grid), dim3((unsigned)a.wg_cols), 0, st, a);
#endif
    SR_HIP(hipGetLastError());
    use->part = pl->part;
    use->ldp = ldp;
    use->block_slab0 = pl->block_slab0;
    use->block_nslab = pl->block_nslab;
    return SIMRANK_OK;
}

}  // namespace simrank

using namespace simrank;

extern "C" {

int simrank_dense_part(const simrank_graph* g, const float* X, int64_t ldx, int64_t n_cols_x,
                       void* stream) {
    SR_REQUIRE(g && X, "NULL argument");
    SR_REQUIRE(g->dense, "the graph has no dense sets");
    // (ldx < 0: X is panel-blocked with -ldx rows per panel)
    SR_REQUIRE(n_cols_x > 0 && (ldx >= n_cols_x || -ldx >= g->n_cols), "X: %lld columns, ld %lld",
               (long long)n_cols_x, (long long)ldx);
    DenseUse use;
    return launch_dense_tiles(g, X, ldx, n_cols_x, false, as_stream(stream), &use);
}

int simrank_graph_dense_stats(const simrank_graph* g, int64_t* n_tiles, int64_t* dense_cols,
                              int64_t* nnz_covered) {
    SR_REQUIRE(g, "graph is NULL");
    const simrank_dense_plan* pl = g->dense;
    if (n_tiles) *n_tiles = pl ? pl->n_blocks_dense : 0;
    if (dense_cols) *dense_cols = pl ? pl->total_k : 0;
    if (nnz_covered) *nnz_covered = pl ? pl->nnz_covered : 0;
    return SIMRANK_OK;
}

}  // extern "C"
