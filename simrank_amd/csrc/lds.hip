// LDS-tiled gather legs for graphs whose source range fits one tile (K <= 8192 rows).
//
// The L2-gather kernel of spmm.hip moves every referenced 128-byte row segment through the
// L2 -> L1 path (deg times the algorithmic bytes).  Here the reuse is served from LDS instead:
//
//   * matrices live in the panel-blocked layout "B4": element (r, c) of an Rp x Cp matrix
//     (both padded to multiples of 4) is at ((c >> 2) * Rp + r) * 4 + (c & 3) — the 4 columns
//     of a panel are one 16-byte entry per row, and a whole K x 4 panel is CONTIGUOUS;
//   * a 1024-thread workgroup stages one panel (K x 16 B <= 128 KiB) into LDS with fully
//     coalesced loads and computes that panel for a block of up to 8192 output rows;
//   * ONE OUTPUT ROW PER LANE: rows of at most kShortMax entries are sorted by length on the
//     host and packed 64 at a time in SELL order (entry t of the 64 rows = 64 consecutive
//     u16 ids, one coalesced load), each lane gathers its row's 16-byte entries from LDS
//     (ds_read_b128, ~45 TB/s aggregate with random banks) into a register accumulator;
//   * longer rows are taken one per wave: lanes stride over the neighbour list, partial sums
//     combined by a fixed-order shuffle tree;
//   * results are staged through LDS (the tile is dead by then) and leave TRANSPOSED, again in
//     B4, as 64-byte runs: Zt(c, a) = rowscale[a] * sum_i X(i, c).  Because the iterates are
//     symmetric the transposed result of leg 2 IS the next iterate, so both legs use this one
//     kernel and no data ever changes layout between legs.  The fused epilogue (coef,
//     evidence, prior, diagonal, convergence count) indexes its operands like the output.
//
// Deterministic: fixed summation order, integer atomics only.
#include <algorithm>
#include <mutex>
#include <numeric>
#include <vector>

#include "common.h"

namespace simrank {

constexpr int kLdsThreads = 1024;
constexpr int kLdsWaves = 16;
constexpr int kShortMax = 1024;     // rows up to this many entries go one-per-lane (SELL)
constexpr int kRplMax = 8;          // 64-row groups a wave may own per block
constexpr int kBlockRowsMax = kLdsThreads * kRplMax;   // 8192
constexpr int kLdsBudget = 160 * 1024 - 1024;
constexpr unsigned short kPad = 0xFFFF;

struct LdsBlock {            // one block of consecutive output rows
    int32_t row_lo, n_rows;  // row_lo is a multiple of 4
    int32_t n_short, n_long;
    int32_t slot_base;       // into slot_row
    int32_t grp_base;        // into grp_off / grp_len, ceil(n_short / 64) groups
    int32_t long_base;       // into long_row / long_off / long_len
    int32_t pad;
};

}  // namespace simrank

// SELL / long-row packing of a graph for the LDS kernel (built on first use, cached).
struct simrank_lds_plan {
    int32_t n_blocks = 0;
    int32_t max_long = 0;          // largest n_long of a block
    int32_t max_rows = 0;          // largest n_rows of a block
    simrank::LdsBlock* blocks = nullptr;
    uint16_t* slot_row = nullptr;  // sorted slot -> block-local row
    uint32_t* grp_off = nullptr;   // offset (in u16) of a group's ids
    uint16_t* grp_len = nullptr;   // longest row of the group
    uint16_t* long_row = nullptr;
    uint32_t* long_off = nullptr;
    uint32_t* long_len = nullptr;
    uint16_t* ids = nullptr;
};

namespace simrank {

template <typename T>
static int upload_vec(T** d, const std::vector<T>& h) {
    const size_t bytes = std::max<size_t>(16, h.size() * sizeof(T));
    SR_HIP(hipMalloc((void**)d, bytes));
    if (!h.empty()) SR_HIP(hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return SIMRANK_OK;
}

// Builds the plan from host copies of the CSR (kept by the graph for this purpose).
static int build_plan(const simrank_graph* g, simrank_lds_plan** out) {
    const int64_t M = g->n_rows, K = g->n_cols;
    const std::vector<int32_t>& rowptr = g->h_rowptr;
    const std::vector<int32_t>& col = g->h_col;
    std::vector<LdsBlock> blocks;
    std::vector<uint16_t> slot_row, grp_len, long_row, ids;
    std::vector<uint32_t> grp_off, long_off, long_len;
    const int64_t tile_bytes = ((K + 3) / 4 * 4) * 16;
    int max_long = 0, max_rows = 0;
    int64_t a = 0;
    while (a < M) {
        // rows of this block: up to 8192, long-row results + max(tile, row buffer) must fit LDS
        int64_t n = 0, n_long = 0;
        while (a + n < M && n < kBlockRowsMax) {
            const int len = rowptr[a + n + 1] - rowptr[a + n];
            const int64_t nl = n_long + (len > kShortMax ? 1 : 0);
            const int64_t need = std::max<int64_t>(tile_bytes, (n + 4) / 4 * 4 * 16) + nl * 16;
            if (need > kLdsBudget) break;
            n_long = nl;
            ++n;
        }
        if (n == 0) return SIMRANK_ERR_INVALID;             // a single row does not fit
        if (a + n < M) n = n / 4 * 4;                        // blocks start at multiples of 4
        if (n == 0) return SIMRANK_ERR_INVALID;
        LdsBlock b{};
        b.row_lo = (int32_t)a;
        b.n_rows = (int32_t)n;
        b.slot_base = (int32_t)slot_row.size();
        b.grp_base = (int32_t)grp_off.size();
        b.long_base = (int32_t)long_row.size();
        std::vector<int32_t> shorts;
        for (int64_t r = 0; r < n; ++r) {
            const int len = rowptr[a + r + 1] - rowptr[a + r];
            if (len > kShortMax) {
                long_row.push_back((uint16_t)r);
                long_off.push_back((uint32_t)ids.size());
                long_len.push_back((uint32_t)len);
                for (int j = rowptr[a + r]; j < rowptr[a + r + 1]; ++j) ids.push_back((uint16_t)col[j]);
                ++b.n_long;
            } else {
                shorts.push_back((int32_t)r);
            }
        }
        std::stable_sort(shorts.begin(), shorts.end(), [&](int32_t x, int32_t y) {
            return rowptr[a + x + 1] - rowptr[a + x] > rowptr[a + y + 1] - rowptr[a + y];
        });
        b.n_short = (int32_t)shorts.size();
        for (size_t s0 = 0; s0 < shorts.size(); s0 += 64) {
            const size_t cnt = std::min<size_t>(64, shorts.size() - s0);
            const int maxlen = rowptr[a + shorts[s0] + 1] - rowptr[a + shorts[s0]];
            grp_off.push_back((uint32_t)ids.size());
            grp_len.push_back((uint16_t)maxlen);
            for (int t = 0; t < maxlen; ++t)
                for (size_t l = 0; l < 64; ++l) {
                    uint16_t v = kPad;
                    if (l < cnt) {
                        const int64_t row = a + shorts[s0 + l];
                        if (t < rowptr[row + 1] - rowptr[row]) v = (uint16_t)col[rowptr[row] + t];
                    }
                    ids.push_back(v);
                }
        }
        for (int32_t r : shorts) slot_row.push_back((uint16_t)r);
        while (slot_row.size() % 64) slot_row.push_back(kPad);
        max_long = std::max(max_long, b.n_long);
        max_rows = std::max(max_rows, b.n_rows);
        blocks.push_back(b);
        a += n;
    }
    simrank_lds_plan* p = new simrank_lds_plan;
    p->n_blocks = (int32_t)blocks.size();
    p->max_long = max_long;
    p->max_rows = max_rows;
    int rc = upload_vec(&p->blocks, blocks);
    if (!rc) rc = upload_vec(&p->slot_row, slot_row);
    if (!rc) rc = upload_vec(&p->grp_off, grp_off);
    if (!rc) rc = upload_vec(&p->grp_len, grp_len);
    if (!rc) rc = upload_vec(&p->long_row, long_row);
    if (!rc) rc = upload_vec(&p->long_off, long_off);
    if (!rc) rc = upload_vec(&p->long_len, long_len);
    if (!rc) rc = upload_vec(&p->ids, ids);
    if (rc) return rc;
    *out = p;
    return SIMRANK_OK;
}

struct LdsArgs {
    const LdsBlock* blocks;
    const uint16_t* slot_row;
    const uint32_t* grp_off;
    const uint16_t* grp_len;
    const uint16_t* long_row;
    const uint32_t* long_off;
    const uint32_t* long_len;
    const uint16_t* ids;
    const float* rowscale;
    const float4* X;     // B4, Kp rows
    float4* Zt;          // B4, Lp rows (transposed result)
    int32_t K, Kp, M, Mp, L, Lp;
    int32_t n_panels, n_blocks;
    int32_t lds_long_off;   // float4 index of the long-row result area
    int32_t has_ep;
    float coef, lbd;
    const uint32_t* ev;     // B4 of u8: one u32 per (row, panel)
    const float4* ap;
    const float4* prev;
    double eps;
    unsigned long long* n_changed;
    int32_t set_diag;
};

__device__ __forceinline__ void add4(float4& a, const float4 b) {
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
}

__global__ __launch_bounds__(kLdsThreads) void spmm_lds_kernel(const LdsArgs p) {
    extern __shared__ __attribute__((aligned(16))) float4 lds[];   // tile | long results; row buffer aliases the tile
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // block -> (panel, row block).  Adjacent panel PAIRS share an XCD (blockIdx % 8) so the two
    // 64-byte halves of every output line meet in one L2.
    int panel, blk;
    {
        const int x = blockIdx.x & 7;
        const int local = blockIdx.x >> 3;
        const int pair = (local / (2 * p.n_blocks)) * 8 + x;
        const int in = local % (2 * p.n_blocks);
        panel = pair * 2 + (in & 1);
        blk = in >> 1;
    }
    if (panel >= p.n_panels) return;
    const LdsBlock b = p.blocks[blk];

    // ---- stage the panel: K contiguous 16-byte entries
    const float4* src = p.X + size_t(panel) * p.Kp;
    for (int k = tid; k < p.K; k += kLdsThreads) lds[k] = src[k];
    __syncthreads();

    // ---- phase 1: short rows, one per lane, 64 rows per group in SELL order.  With 8 waves
    // per CU the dependent id loads are the critical path, so four groups are walked together:
    // 32 coalesced id loads in flight per wave, then 32 LDS gathers.
    float4 acc[kRplMax];
#pragma unroll
    for (int j = 0; j < kRplMax; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int n_groups = (b.n_short + 63) >> 6;
#pragma unroll
    for (int jb = 0; jb < kRplMax; jb += 4) {
        if (jb * kLdsWaves + wave < n_groups) {
            const uint16_t* id[4];
            int len[4];
            int maxlen = 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int g = (jb + u) * kLdsWaves + wave;
                const bool on = g < n_groups;
                id[u] = p.ids + (on ? p.grp_off[b.grp_base + g] : 0) + lane;
                len[u] = on ? int(p.grp_len[b.grp_base + g]) : 0;
                maxlen = max(maxlen, len[u]);
            }
            for (int t0 = 0; t0 < maxlen; t0 += 8) {
                unsigned ids[4][8];
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        ids[u][k] = (t0 + k < len[u]) ? unsigned(id[u][(t0 + k) * 64]) : unsigned(kPad);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        if (ids[u][k] != kPad) add4(acc[jb + u], lds[ids[u][k]]);
            }
        }
    }

    // ---- phase 2: long rows, one per wave, two rows in flight: lanes stride over the id list
    // (coalesced), partial sums combined by a fixed-order shuffle tree
    float4* longres = lds + p.lds_long_off;
    for (int q = wave; q < b.n_long; q += 2 * kLdsWaves) {
        const int qb = q + kLdsWaves;
        const bool two = qb < b.n_long;
        const uint16_t* ida = p.ids + p.long_off[b.long_base + q] + lane;
        const int lena = int(p.long_len[b.long_base + q]);
        const uint16_t* idb = p.ids + (two ? p.long_off[b.long_base + qb] : 0) + lane;
        const int lenb = two ? int(p.long_len[b.long_base + qb]) : 0;
        float4 pa = make_float4(0.f, 0.f, 0.f, 0.f), pb = pa;
        const int maxlen = max(lena, lenb);
        for (int t0 = 0; t0 < maxlen; t0 += 4 * 64) {
            unsigned ia[4], ib[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int t = t0 + k * 64 + lane;
                ia[k] = t < lena ? unsigned(ida[t0 + k * 64]) : unsigned(kPad);
                ib[k] = t < lenb ? unsigned(idb[t0 + k * 64]) : unsigned(kPad);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (ia[k] != kPad) add4(pa, lds[ia[k]]);
                if (ib[k] != kPad) add4(pb, lds[ib[k]]);
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            pa.x += __shfl_xor(pa.x, off); pa.y += __shfl_xor(pa.y, off);
            pa.z += __shfl_xor(pa.z, off); pa.w += __shfl_xor(pa.w, off);
            pb.x += __shfl_xor(pb.x, off); pb.y += __shfl_xor(pb.y, off);
            pb.z += __shfl_xor(pb.z, off); pb.w += __shfl_xor(pb.w, off);
        }
        if (lane == 0) {
            longres[q] = pa;
            if (two) longres[qb] = pb;
        }
    }
    __syncthreads();                       // the tile is dead: its space becomes the row buffer

    // ---- scaled row results into the row buffer (block-local row order)
    const float cf = p.has_ep ? p.coef : 1.0f;
#pragma unroll
    for (int j = 0; j < kRplMax; ++j) {
        const int slot = (j * kLdsWaves + wave) * 64 + lane;
        if (slot < b.n_short) {
            const int r = p.slot_row[b.slot_base + slot];
            const float sc = p.rowscale[b.row_lo + r] * cf;
            lds[r] = make_float4(acc[j].x * sc, acc[j].y * sc, acc[j].z * sc, acc[j].w * sc);
        }
    }
    __syncthreads();                       // (long results live outside the row buffer)
    for (int q = tid; q < b.n_long; q += kLdsThreads) {
        const int r = p.long_row[b.long_base + q];
        const float sc = p.rowscale[b.row_lo + r] * cf;
        const float4 v = longres[q];
        lds[r] = make_float4(v.x * sc, v.y * sc, v.z * sc, v.w * sc);
    }
    __syncthreads();

    // ---- transposed store, B4: for a quad of rows (4aq .. 4aq+3) and column c0+i one 16-byte
    // entry at ((aq) * Lp + c0 + i); epilogue operands are indexed the same way
    const int n_quads = (b.n_rows + 3) >> 2;
    const int c0 = panel * 4;
    const float* rb = reinterpret_cast<const float*>(lds);
    unsigned changed = 0;
    for (int e = tid; e < n_quads * 4; e += kLdsThreads) {
        const int aq = e >> 2, i = e & 3;
        const int r0 = aq * 4;
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (r0 + j < b.n_rows) ? rb[(r0 + j) * 4 + i] : 0.f;
        const int c = c0 + i;
        const size_t idx = size_t((b.row_lo >> 2) + aq) * p.Lp + c;
        if (p.has_ep) {
            if (p.ev) {
                const unsigned w = p.ev[idx];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    o[j] *= 1.0f - __builtin_ldexpf(1.0f, -int((w >> (8 * j)) & 255u));
            }
            if (p.ap) {
                const float4 pr = p.ap[idx];
                const float keep = 1.0f - p.lbd;
                o[0] = keep * o[0] + p.lbd * pr.x; o[1] = keep * o[1] + p.lbd * pr.y;
                o[2] = keep * o[2] + p.lbd * pr.z; o[3] = keep * o[3] + p.lbd * pr.w;
            }
            if (p.set_diag) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (b.row_lo + r0 + j == c && c < p.L) o[j] = 1.0f;
            }
            if (p.prev) {
                const float4 old = p.prev[idx];
                const float ov[4] = {old.x, old.y, old.z, old.w};
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    changed += (b.row_lo + r0 + j < p.M && c < p.L &&
                                fabs(double(o[j]) - double(ov[j])) > p.eps) ? 1u : 0u;
            }
        }
        p.Zt[idx] = make_float4(o[0], o[1], o[2], o[3]);
    }
    if (p.has_ep && p.prev) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) changed += __shfl_down(changed, off);
        if (lane == 0 && changed)
            atomicAdd(p.n_changed + ((blockIdx.x * 16u + (threadIdx.x >> 6)) * 7u) % SIMRANK_CHANGED_SLOTS,
                      (unsigned long long)changed);
    }
}

// ---- layout helpers -------------------------------------------------------------------
__global__ __launch_bounds__(256) void b4_identity_kernel(float4* S, int64_t np4, int64_t rows_p) {
    // S is B4 with rows_p rows and np4 panels; entry (r, panel) holds columns 4*panel .. +3
    const int64_t total = np4 * rows_p;
    for (int64_t t = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; t < total;
         t += int64_t(gridDim.x) * blockDim.x) {
        const int64_t panel = t / rows_p, r = t - panel * rows_p;
        const int64_t d = r - panel * 4;
        S[t] = make_float4(d == 0 ? 1.f : 0.f, d == 1 ? 1.f : 0.f, d == 2 ? 1.f : 0.f, d == 3 ? 1.f : 0.f);
    }
}

__global__ __launch_bounds__(256) void b4_unpack_kernel(const float4* S, int64_t rows_p, float* out,
                                                        int64_t ld, int64_t n_rows, int64_t n_cols) {
    const int64_t np4 = (n_cols + 3) / 4;
    const int64_t total = np4 * n_rows;
    for (int64_t t = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; t < total;
         t += int64_t(gridDim.x) * blockDim.x) {
        const int64_t r = t / np4, panel = t - r * np4;      // consecutive threads: one row
        const float4 v = S[panel * rows_p + r];
        float* o = out + r * ld + panel * 4;
        const float vv[4] = {v.x, v.y, v.z, v.w};
        for (int j = 0; j < 4; ++j)
            if (panel * 4 + j < n_cols) o[j] = vv[j];
    }
}

template <typename T, typename T4>
__global__ __launch_bounds__(256) void b4_pack_kernel(const T* in, int64_t ld, int64_t n_rows,
                                                      int64_t n_cols, T4* out, int64_t rows_p,
                                                      int64_t np4) {
    const int64_t total = np4 * rows_p;
    for (int64_t t = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; t < total;
         t += int64_t(gridDim.x) * blockDim.x) {
        const int64_t panel = t / rows_p, r = t - panel * rows_p;
        T v[4];
        for (int j = 0; j < 4; ++j)
            v[j] = (r < n_rows && panel * 4 + j < n_cols) ? in[r * ld + panel * 4 + j] : T(0);
        if constexpr (sizeof(T) == 1) {
            out[t] = T4(unsigned(v[0]) | (unsigned(v[1]) << 8) | (unsigned(v[2]) << 16) | (unsigned(v[3]) << 24));
        } else {
            out[t] = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

static std::mutex g_plan_mutex;

}  // namespace simrank

using namespace simrank;

extern "C" {

int simrank_lds_supported(const simrank_graph* g, int32_t* ok) {
    SR_REQUIRE(g && ok, "NULL argument");
    *ok = 0;
    if (g->n_cols > 8192 || g->n_rows >= (int64_t(1) << 30)) return SIMRANK_OK;
    std::lock_guard<std::mutex> lock(g_plan_mutex);
    simrank_graph* gg = const_cast<simrank_graph*>(g);
    if (!gg->lds_plan && !gg->lds_plan_failed) {
        simrank_lds_plan* plan = nullptr;
        const int rc = build_plan(g, &plan);
        if (rc == SIMRANK_OK) gg->lds_plan = plan;
        else gg->lds_plan_failed = true;
    }
    *ok = gg->lds_plan ? 1 : 0;
    return SIMRANK_OK;
}

int simrank_spmm_lds(const simrank_graph* g, const float* X_b4, int64_t n_cols_x, float* Zt_b4,
                     const simrank_epilogue* ep, void* stream) {
    SR_REQUIRE(g && X_b4 && Zt_b4, "NULL argument");
    int32_t ok = 0;
    int rc = simrank_lds_supported(g, &ok);
    if (rc) return rc;
    SR_REQUIRE(ok, "graph not eligible for the LDS-tiled kernel (more than 8192 source rows)");
    SR_REQUIRE(n_cols_x > 0 && aligned16(X_b4) && aligned16(Zt_b4), "bad operands");
    const simrank_lds_plan* pl = g->lds_plan;
    LdsArgs a{};
    a.blocks = pl->blocks; a.slot_row = pl->slot_row; a.grp_off = pl->grp_off; a.grp_len = pl->grp_len;
    a.long_row = pl->long_row; a.long_off = pl->long_off; a.long_len = pl->long_len; a.ids = pl->ids;
    a.rowscale = g->rowscale;
    a.X = reinterpret_cast<const float4*>(X_b4);
    a.Zt = reinterpret_cast<float4*>(Zt_b4);
    a.K = (int32_t)g->n_cols; a.Kp = (a.K + 3) / 4 * 4;
    a.M = (int32_t)g->n_rows; a.Mp = (a.M + 3) / 4 * 4;
    a.L = (int32_t)n_cols_x;  a.Lp = (a.L + 3) / 4 * 4;
    a.n_panels = a.Lp / 4;
    a.n_blocks = pl->n_blocks;
    const int64_t buf_entries = std::max<int64_t>(a.Kp, (pl->max_rows + 3) / 4 * 4);
    a.lds_long_off = (int32_t)buf_entries;
    const size_t lds = size_t(buf_entries + pl->max_long) * 16;
    SR_REQUIRE(lds <= 160 * 1024, "LDS plan needs %zu bytes", lds);
    hipStream_t st = as_stream(stream);
    if (ep) {
        a.has_ep = 1;
        a.coef = ep->coef; a.lbd = ep->lbd;
        a.ev = reinterpret_cast<const uint32_t*>(ep->evidence);
        a.ap = reinterpret_cast<const float4*>(ep->apriori);
        a.prev = reinterpret_cast<const float4*>(ep->previous);
        a.eps = ep->eps; a.n_changed = ep->n_changed; a.set_diag = ep->set_diag;
        SR_REQUIRE(ep->diag_col0 == 0, "the LDS kernel works on whole matrices");
        SR_REQUIRE(!a.prev || a.n_changed, "previous needs a counter");
        if (a.prev)
            SR_HIP(hipMemsetAsync(a.n_changed, 0, sizeof(unsigned long long) * SIMRANK_CHANGED_SLOTS, st));
    }
    auto kern = spmm_lds_kernel;
    SR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int64_t pairs = (a.n_panels + 1) / 2;
    const int64_t grid = ((pairs + 7) / 8) * 8 * 2 * a.n_blocks;
    SR_REQUIRE(grid > 0 && grid < (int64_t(1) << 31), "grid too large");
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kLdsThreads), lds, st, a);
    SR_HIP(hipGetLastError());
    return SIMRANK_OK;
}

int simrank_b4_identity(float* S_b4, int64_t n, void* stream) {
    SR_REQUIRE(S_b4 && n > 0, "bad argument");
    const int64_t rows_p = (n + 3) / 4 * 4, np4 = rows_p / 4;
    const int grid = (int)std::min<int64_t>((np4 * rows_p + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(b4_identity_kernel, dim3(grid), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<float4*>(S_b4), np4, rows_p);
    SR_HIP(hipGetLastError());
    return SIMRANK_OK;
}

int simrank_b4_unpack(const float* S_b4, int64_t n_rows, int64_t n_cols, float* out, int64_t ld,
                      void* stream) {
    SR_REQUIRE(S_b4 && out && n_rows > 0 && n_cols > 0 && ld >= n_cols, "bad argument");
    const int64_t rows_p = (n_rows + 3) / 4 * 4;
    const int grid = (int)std::min<int64_t>(((n_cols + 3) / 4 * n_rows + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(b4_unpack_kernel, dim3(grid), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float4*>(S_b4), rows_p, out, ld, n_rows, n_cols);
    SR_HIP(hipGetLastError());
    return SIMRANK_OK;
}

int simrank_b4_pack(const void* in, int64_t ld, int64_t n_rows, int64_t n_cols, int32_t elem_bytes,
                    void* out_b4, void* stream) {
    SR_REQUIRE(in && out_b4 && n_rows > 0 && n_cols > 0 && ld >= n_cols, "bad argument");
    SR_REQUIRE(elem_bytes == 1 || elem_bytes == 4, "elem_bytes must be 1 (u8) or 4 (f32)");
    const int64_t rows_p = (n_rows + 3) / 4 * 4, np4 = (n_cols + 3) / 4;
    const int grid = (int)std::min<int64_t>((np4 * rows_p + 255) / 256, 256 * 16);
    if (elem_bytes == 1)
        hipLaunchKernelGGL((b4_pack_kernel<uint8_t, uint32_t>), dim3(grid), dim3(256), 0,
                           as_stream(stream), (const uint8_t*)in, ld, n_rows, n_cols,
                           (uint32_t*)out_b4, rows_p, np4);
    else
        hipLaunchKernelGGL((b4_pack_kernel<float, float4>), dim3(grid), dim3(256), 0,
                           as_stream(stream), (const float*)in, ld, n_rows, n_cols,
                           (float4*)out_b4, rows_p, np4);
    SR_HIP(hipGetLastError());
    return SIMRANK_OK;
}

}  // extern "C"

namespace simrank {
void free_lds_plan(simrank_lds_plan* p) {
    if (!p) return;
    (void)hipFree(p->blocks); (void)hipFree(p->slot_row); (void)hipFree(p->grp_off);
    (void)hipFree(p->grp_len); (void)hipFree(p->long_row); (void)hipFree(p->long_off);
    (void)hipFree(p->long_len); (void)hipFree(p->ids);
    delete p;
}
}  // namespace simrank
