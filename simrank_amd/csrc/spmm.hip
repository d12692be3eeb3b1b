// Sparse legs of the SimRank update for gfx950 (MI355X), plus identity init, the stand-alone
// epilogue and the SimRank++ evidence counts.
//
//   Y = diag(rowscale) . A . X          A = 0/1 CSR pattern, X dense row-major
//
// Row-gather form: output row a is rowscale[a] times the sum of the rows of X listed in CSR
// row a.  This is bandwidth work, not arithmetic:
//   * a launch is tiled into column PANELS of PW = VEC*LPR floats; inside a panel one 64-lane
//     wave owns a tile of RT consecutive output rows.  LPR lanes cover one neighbour-row
//     segment (16 B per lane), so one load instruction gathers G = 64/LPR different rows;
//   * the wave sorts its tile's rows by length (bitonic network in registers).  Long rows
//     (>= kHeavy entries) are gathered one at a time, the G lane groups splitting the
//     neighbours, partial sums combined by shuffles in a fixed order.  All other rows go G at
//     a time, ONE ROW PER LANE GROUP: no cross-lane reduction, G independent gather streams,
//     8 loads in flight in each, neighbour ids requested one chunk ahead;
//   * results are bitwise reproducible: fixed summation order, integer atomics only;
//   * blocks equal mod 8 share an XCD and its L2 (observed dispatch order; speed only):
//     panel p is processed by the blocks with blockIdx % 8 == p % 8, in row-tile order, so
//     the K x PW slice of X that a panel re-reads deg times stays hot in that L2;
//   * leg 1 stores its tile TRANSPOSED through a per-wave LDS tile (RT-float row segments,
//     optionally in per-destination blocks for the all-to-all), so leg 2 is the same gather;
//   * leg 2 fuses the whole reference epilogue: coef, evidence 1-2^-count, prior blend,
//     diag <- 1 and the |new-old| > eps count of `_converged` (SimRank.py:74,139-140); on a
//     single rank it computes only the tiles on/above the diagonal of the symmetric result
//     and stores their mirror image (kSym).
#include <algorithm>
#include <type_traits>

#include <mutex>

#include "common.h"

namespace simrank {

__device__ __forceinline__ int64_t imin(int64_t a, int64_t b) { return a < b ? a : b; }

// element (r, c) of a matrix: row-major with leading dimension ld (rows_pad == 0), or panel-blocked
// (32-column panels of rows_pad rows: see SpmmArgs::blocked)
__device__ __forceinline__ int64_t elem_at(int64_t r, int64_t c, int64_t ld, int64_t rows_pad) {
    return rows_pad ? ((c >> 5) * rows_pad + r) * 32 + (c & 31) : r * ld + c;
}

struct SpmmArgs {
    const int32_t* rowptr;
    const int32_t* col;       // neighbour ids, ascending per row
    const uint16_t* col16;    // the same in 16 bits (NULL when the graph has more than 65536 columns)
    // panel-blocked operands (lean kernel only): a matrix of R rows x C columns is stored as
    // ceil(C/32) panels of rows_pad x 32 floats (u8 counts: 32 bytes), element (r, c) at
    // ((c >> 5) * rows_pad + r) * 32 + (c & 31).  A panel's slice is then contiguous (4 MiB at
    // N = 32768) instead of 32768 segments spread over 4 GB: the gathers stop missing the TLB.
    int32_t blocked;
    int64_t x_rows_pad, y_rows_pad;   // X; Y and the epilogue operands (and Y^T of a transposed store)
    int64_t x_span_bytes;     // A32: bytes from X (blocked: from a panel's start) to the end of what a gather may touch
    int32_t x_sentinel;       // A32: a row id whose offset is past that span (loads zeros)
    int32_t addr32;           // tuning "addr32": allow the 32-bit buffer addressing
    int32_t sh_rank, sh_mb;   // kShard: this rank's shard, rows (= columns) per shard (multiple of 32)
    float* sh_send;           // kShard: packed mirrored tiles for the other ranks
    int64_t sh_chunk;         // kShard: floats per destination rank in sh_send
    int32_t sh_tile0, sh_ntiles; // kShard, one STAGE of the leg: column tiles [sh_tile0, sh_tile0 + sh_ntiles) (0 tiles = all)
    int64_t sh_slot0;         // ... whose packed mirrored tiles start at slot sh_slot0 = sh_tile0 (sh_tile0 - 1) / 2 of a chunk
    int32_t idx_mask;         // diagnostic (tuning "probe_mask"): neighbour ids are ANDed with it; -1 = off
    int32_t probe;            // diagnostic (tuning "probe_flags", lean kernel): 1 no gathers, 2 no stores of Y,
                              // 4 no partial sums of the dense part, 8 no neighbour-id loads
    int32_t nt;               // non-temporal output stores
    int32_t has_huge;         // the graph has rows of >= huge_len entries
    int32_t huge_len;         // rows this long are split over the waves of a workgroup
    const float* rowscale;
    const float* X;
    int64_t ldx;
    int64_t L;  // columns of X and of Y
    float* Y;
    int64_t ldy;
    int64_t M;       // rows of the graph = rows of Y
    int64_t K;       // columns of the graph = rows of X
    int64_t tblock;  // rows per transposed block (TRANS only)
    int64_t tstride; // TRANS, single block: row stride of Y^T (0 = rows in block)
    int64_t tpad;    // TRANS, blocked: padding floats per row of a block
    int32_t tvec;    // transposed stores may use 16-byte pieces (alignment checked on the host)
    int32_t n_panels;
    int32_t row_tiles;        // workgroups per panel (kWaves tiles each)
    // balanced tiling (RT = 32 only): tile t covers rows [tile_row0[t], tile_row0[t+1]), at
    // most 32 of them and never across a multiple of 32; NULL = uniform RT-row tiles
    const int32_t* tile_row0;
    int32_t n_tiles;
    const int32_t* sym_map;   // kSym + tile list: blockIdx -> (panel, workgroup of the panel) pairs
    int32_t sym_blocks;
    int32_t xcd_map;
    int32_t has_ep;
    // epilogue (has_ep)
    float coef;
    float lbd;
    const uint8_t* ev;
    int64_t ld_ev;
    const float* ap;
    int64_t ld_ap;
    const float* prev;
    int64_t ld_prev;
    double eps;
    unsigned long long* n_changed;
    int64_t diag_col0;
    int32_t set_diag;
    int32_t count_any;        // lean kernel: stop comparing with `prev` once counter 0 is non-zero (see simrank_epilogue)
    int32_t restrict_support; // lean kernel, evidence given: skip the gathers of 32-column segments whose
                              // evidence counts are all zero (their result is exactly 0: S inside supp(E))
    // block-dense part (blockdense.hip): raw partial sums of the entries that went to the matrix
    // cores; row a of 128-row block b has one row in each of the block's slabs,
    // dpart[((dslab0[b] + s) * 128 + a % 128) * ldp], s < dnslab[b]; NULL = none
    const float* dpart;
    int64_t ldp;
    const int32_t* dslab0;
    const int32_t* dnslab;
};

template <int VEC>
__device__ __forceinline__ void vload(float (&d)[VEC], const float* p) {
    if constexpr (VEC == 4) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w;
    } else {
        d[0] = *p;
    }
}

template <int VEC>
__device__ __forceinline__ void vstore(float* p, const float (&d)[VEC]) {
    if constexpr (VEC == 4) {
        *reinterpret_cast<float4*>(p) = make_float4(d[0], d[1], d[2], d[3]);
    } else {
        *p = d[0];
    }
}

// Streamed-once traffic (previous iterate, prior, evidence, the output rows) is loaded /
// stored non-temporally so it does not push the panel of X out of the XCD's L2.
template <int VEC>
__device__ __forceinline__ void vload_nt(float (&d)[VEC], const float* p) {
    if constexpr (VEC == 4) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4 t = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p));
        d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w;
    } else {
        d[0] = __builtin_nontemporal_load(p);
    }
}

template <int VEC>
__device__ __forceinline__ void vstore_nt(float* p, const float (&d)[VEC]) {
    if constexpr (VEC == 4) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4 t;
        t.x = d[0]; t.y = d[1]; t.z = d[2]; t.w = d[3];
        __builtin_nontemporal_store(t, reinterpret_cast<f4*>(p));
    } else {
        __builtin_nontemporal_store(d[0], p);
    }
}

// neighbour ids are streamed once per panel pass; keep them from displacing X in L2
__device__ __forceinline__ int ldidx(const SpmmArgs& p, int j) {
    if (p.col16) return p.nt ? int(__builtin_nontemporal_load(p.col16 + j)) : int(p.col16[j]);
    return p.nt ? __builtin_nontemporal_load(p.col + j) : p.col[j];
}

constexpr int kWaves = 4;   // waves per workgroup
constexpr int kHeavy = 64;
constexpr int kSkip = INT32_MIN;  // "no neighbour in this slot"  // rows with at least this many entries are gathered cooperatively

// One output row segment (VEC floats per lane, LPR lanes) -> epilogue -> memory / LDS tile.
// MODE: how a tile leaves the wave.
//   kPlain  Y[a][c]                       (+ fused epilogue)
//   kTrans  Y^T, through the wave's LDS tile (leg 1)
//   kSym    kPlain for the tiles on or above the diagonal of a symmetric result, which are
//           ALSO stored mirrored (through the LDS tile); tiles below the diagonal are
//           never computed.  Halves the gathers of leg 2 on a single rank.
constexpr int kPlain = 0, kTrans = 1, kSym = 2;
// kShard (lean kernel only): leg 2 of one rank of a SHARDED symmetric update.  The rank owns the
// columns of shard g (sh_mb columns); node positions are dealt so that every shard holds an equal mix
// of short and long rows, ascending inside the shard.  For the row tile i of shard h and its own column
// tile j the rank computes S'[rows (h,i), cols (g,j)] only when i <= j; for i < j it also stores the
// transposed tile — into its own block when h == g, else into a send buffer (packed 32 x 32 tiles,
// slot j(j-1)/2 + i of the chunk for rank h), which a second all-to-all delivers to the rank that
// owns columns (h,i) (simrank_shard_unpack puts it at rows (g,j)).  Halves the gathers of a sharded
// leg 2 the way kSym does on one rank.
constexpr int kShard = 3;

template <int VEC, int LPR, int MODE, int RT>
__device__ __forceinline__ void emit_row(const SpmmArgs& p, float* tbuf_wave, int r_local,
                                         int64_t a, int q, int64_t mycol, const float (&acc)[VEC],
                                         unsigned& changed, bool mirror) {
    constexpr bool TRANS = MODE == kTrans;
    const float sc = p.rowscale[a] * (p.has_ep ? p.coef : 1.0f);
    float o[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) o[i] = acc[i];
    if (p.dpart) {
        // the slabs are summed first (in slab order) and added as ONE term, as the lean kernel does:
        // both kernels then give the same bits
        const int ns = p.dnslab[a >> 7];
        const float* dp = p.dpart + (int64_t(p.dslab0[a >> 7]) * 128 + (a & 127)) * p.ldp + mycol;
        float dsum[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) dsum[i] = 0.f;
        for (int s = 0; s < ns; ++s, dp += 128 * p.ldp) {        // fixed order
            float d[VEC];
            vload_nt<VEC>(d, dp);
#pragma unroll
            for (int i = 0; i < VEC; ++i) dsum[i] += d[i];
        }
#pragma unroll
        for (int i = 0; i < VEC; ++i) o[i] += dsum[i];
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) o[i] *= sc;
    if constexpr (TRANS) {
        float* t = tbuf_wave + (q * VEC) * (RT + 1) + r_local;
#pragma unroll
        for (int i = 0; i < VEC; ++i) t[i * (RT + 1)] = o[i];
    } else {
        const int nvalid = int(imin(VEC, p.L - mycol));
        if (p.has_ep) {
            if (p.ev) {
                const uint8_t* ep = p.ev + a * p.ld_ev + mycol;
                unsigned cnt[VEC];
                if constexpr (VEC == 4) {
                    const unsigned w = __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(ep));
                    cnt[0] = w & 255u; cnt[1] = (w >> 8) & 255u;
                    cnt[2] = (w >> 16) & 255u; cnt[3] = w >> 24;
                } else {
                    cnt[0] = ep[0];
                }
#pragma unroll
                for (int i = 0; i < VEC; ++i) o[i] *= 1.0f - __builtin_ldexpf(1.0f, -int(cnt[i]));
            }
            if (p.ap) {
                float pr[VEC];
                vload_nt<VEC>(pr, p.ap + a * p.ld_ap + mycol);
                const float keep = 1.0f - p.lbd;
#pragma unroll
                for (int i = 0; i < VEC; ++i) o[i] = keep * o[i] + p.lbd * pr[i];
            }
            if (p.set_diag) {
                const int64_t d = a - (p.diag_col0 + mycol);
#pragma unroll
                for (int i = 0; i < VEC; ++i)
                    if (d == i) o[i] = 1.0f;
            }
            if (p.prev) {
                float old[VEC];
                vload_nt<VEC>(old, p.prev + a * p.ld_prev + mycol);
#pragma unroll
                for (int i = 0; i < VEC; ++i)
                    changed += (i < nvalid && fabs(double(o[i]) - double(old[i])) > p.eps)
                                   ? (mirror ? 2u : 1u) : 0u;
            }
        }
        if constexpr (MODE == kSym) {
            if (mirror) {
                float* t = tbuf_wave + (q * VEC) * (RT + 1) + r_local;
#pragma unroll
                for (int i = 0; i < VEC; ++i) t[i * (RT + 1)] = o[i];
            }
        }
        float* y = p.Y + a * p.ldy + mycol;
        if (nvalid == VEC) {
            if (p.nt) vstore_nt<VEC>(y, o); else vstore<VEC>(y, o);
        } else {
#pragma unroll
            for (int i = 0; i < VEC; ++i)
                if (i < nvalid) y[i] = o[i];
        }
    }
}

constexpr int kMaxHuge = 8;    // ... at most this many per workgroup and tile round

// Sum of the X segments of the neighbours at CSR positions [s, e): the G lane groups take
// them round-robin, UNROLL loads in flight each; partial sums combined by shuffles in a
// fixed order.  Every lane ends with the total of its VEC columns.
template <int VEC, int LPR>
__device__ __forceinline__ void gather_range(const SpmmArgs& p, const float* __restrict__ Xc,
                                             int s, int e, int lane, int g, bool col_active,
                                             float (&acc)[VEC]) {
    constexpr int G = 64 / LPR, UNROLL = 4;
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
    for (int base = s; base < e; base += 64) {
        const int n = min(64, e - base);
        const int myidx = lane < n ? ldidx(p, base + lane) : 0;
        for (int k0 = 0; k0 < n; k0 += G * UNROLL) {
            float v[UNROLL][VEC];
            int idx[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const int k = k0 + u * G + g;
                idx[u] = __shfl(myidx, k & 63) & p.idx_mask;
                if (!(col_active && k < n)) idx[u] = kSkip;
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                if (idx[u] >= 0) {
                    vload<VEC>(v[u], Xc + int64_t(idx[u]) * p.ldx);
                } else {
#pragma unroll
                    for (int i = 0; i < VEC; ++i) v[u][i] = 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
#pragma unroll
                for (int i = 0; i < VEC; ++i) acc[i] += v[u][i];
        }
    }
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1)
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] += __shfl_xor(acc[i], off);
}

template <int VEC, int LPR, int MODE, int RT>
__global__ __launch_bounds__(256) void spmm_gather_kernel(const SpmmArgs p) {
    static_assert(RT <= 64, "one lane per tile row");
    static_assert(MODE != kSym || VEC * LPR == RT, "mirrored tiles are square");
    constexpr bool TRANS = MODE == kTrans;
    constexpr bool TILE = MODE != kPlain;    // the wave owns an LDS tile
    constexpr int PW = VEC * LPR;            // panel width in floats
    constexpr int G = 64 / LPR;              // lane groups = rows (or neighbours) in flight
    constexpr int JU = LPR < 8 ? LPR : 8;    // gathers a group keeps in flight
    extern __shared__ __attribute__((aligned(16))) float smem[];  // [wave tiles][huge-row area]

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    // block -> (panel, row-tile group); blocks equal mod 8 share an XCD (speed only)
    int panel, rt;
    if (MODE == kSym && p.sym_map) {
        // balanced tiles: the host lists the (panel, workgroup) pairs that touch the upper
        // triangle, per XCD (blockIdx % 8), heavy workgroups of a panel first
        panel = p.sym_map[2 * blockIdx.x];
        rt = p.sym_map[2 * blockIdx.x + 1];
        if (panel < 0) return;                           // padding of the shorter XCD lists
    } else if constexpr (MODE == kSym) {
        // only the (panel, row block) pairs that touch the upper triangle are launched.
        // XCD x owns panels x + 8j; panel p needs row blocks 0 .. p/4 (128 rows per block,
        // 32 columns per panel), i.e. 2j + b of them with b = x/4 + 1; prefix j^2 + (b-1)j.
        const int x = int(blockIdx.x & 7);
        const int t = int(blockIdx.x >> 3);
        const int b1 = x >> 2;                           // b - 1
        int j = int((sqrtf(float(b1 * b1 + 4 * t)) - float(b1)) * 0.5f);
        while ((j + 1) * (j + 1) + b1 * (j + 1) <= t) ++j;
        while (j * j + b1 * j > t) --j;
        panel = x + 8 * j;
        rt = t - (j * j + b1 * j);
        if (rt > panel / 4) return;                      // padding of the shorter XCD lists
    } else {
        const int64_t bid = blockIdx.x;
        if (p.xcd_map) {
            const int x = int(bid & 7);
            const int64_t local = bid >> 3;
            panel = int(local / p.row_tiles) * 8 + x;
            rt = int(local % p.row_tiles);
        } else {
            panel = int(bid / p.row_tiles);
            rt = int(bid % p.row_tiles);
        }
        // last workgroups first: with rows in ascending length order those are the long ones
        if (p.tile_row0) rt = p.row_tiles - 1 - rt;
    }
    if (panel >= p.n_panels) return;  // uniform over the workgroup

    const int64_t c0 = int64_t(panel) * PW;
    const int g = lane / LPR;
    const int q = lane % LPR;
    const int gbase = lane - q;              // first lane of this lane's group
    const int64_t mycol = c0 + int64_t(q) * VEC;
    const bool col_active = mycol < p.L;  // VEC=4: mycol+3 < ldx because ldx % 4 == 0
    const float* __restrict__ Xc = p.X + mycol;
    float* tbuf_wave = smem + (TILE ? wave * PW * (RT + 1) : 0);
    unsigned changed = 0;


    {
    int64_t row0 = (int64_t(rt) * kWaves + wave) * RT;
    int nrows = int(imin(RT, p.M - row0));  // rows of this wave's tile (may be <= 0)
    if (RT == 32 && p.tile_row0) {          // balanced tiling: heavy 32-row blocks are cut up
        const int t = rt * kWaves + wave;
        row0 = t < p.n_tiles ? p.tile_row0[t] : p.M;
        nrows = t < p.n_tiles ? p.tile_row0[t + 1] - int(row0) : 0;
    }
    bool mirror = false;
    if constexpr (MODE == kSym) {
        const int64_t rb = row0 & ~int64_t(RT - 1);   // the tile's 32-row block
        if (rb > c0) nrows = 0;     // below the diagonal: some other tile's mirror image
        mirror = rb < c0;           // strictly above: store the mirror image too
    }

    // ---- rows of the tile sorted by length, longest first (bitonic over the 64 lanes).
    // key = length * 64 + tile row; lanes without a row get a negative key and sort last.
    int my_start = 0, key = -64 + lane;
    if (lane < nrows) {
        my_start = p.rowptr[row0 + lane];
        key = ((p.rowptr[row0 + lane + 1] - my_start) << 6) | lane;
    }
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            const int other = __shfl_xor(key, j);
            const bool desc = (lane & k) == 0;
            const bool lower = (lane & j) == 0;
            const bool take = (lower == desc) ? (other > key) : (other < key);
            key = take ? other : key;
        }
    }
    const int s_row = key & 63;              // tile row held by this sorted position
    const int s_len = key >> 6;              // its length (-1: none)
    const int s_start = __shfl(my_start, s_row);
    const int n_heavy = __popcll(__ballot(s_len >= kHeavy));

    // ---- phase A0: huge rows (>= huge_len entries, default 512).  One wave would need ~0.7 ms for the
    // 15 336-entry row of the bench graph — longer than a whole sharded launch — so the owner
    // posts them in LDS and all four waves of the workgroup take a quarter of the neighbours
    // each; the owner adds the four partial sums in wave order and emits the row.
    int posted = 0;
    if (p.has_huge) {
        int* hmeta = reinterpret_cast<int*>(smem + (TILE ? kWaves * PW * (RT + 1) : 0));
        float* hpart = reinterpret_cast<float*>(hmeta + 64);
        const int n_huge = __popcll(__ballot(s_len >= p.huge_len));
        // slots are handed out in wave order: which rows get one never depends on timing
        if (lane == 0) hmeta[60 + wave] = n_huge;
        __syncthreads();
        int slot0 = 0, total = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {
            const int c = hmeta[60 + w];
            slot0 += w < wave ? c : 0;
            total += c;
        }
        posted = max(0, min(n_huge, kMaxHuge - slot0));
        if (lane < posted) {
            int* d = hmeta + 4 + 4 * (slot0 + lane);
            d[0] = s_row; d[1] = s_start; d[2] = s_len; d[3] = wave;
        }
        __syncthreads();
        const int nh = min(total, kMaxHuge);
        for (int i = 0; i < nh; ++i) {
            const int* d = hmeta + 4 + 4 * i;
            const int hs = d[1], hl = d[2];
            const int chunk = ((hl + kWaves - 1) / kWaves + 63) & ~63;
            const int s = hs + wave * chunk;
            const int e = min(hs + hl, s + chunk);
            float part[VEC];
            gather_range<VEC, LPR>(p, Xc, s, e, lane, g, col_active, part);
            if (g == 0) vstore<VEC>(hpart + (i * kWaves + wave) * PW + q * VEC, part);
        }
        __syncthreads();
        for (int i = 0; i < nh; ++i) {
            const int* d = hmeta + 4 + 4 * i;
            if (d[3] == wave && g == 0 && col_active) {
                float acc[VEC], t[VEC];
                vload<VEC>(acc, hpart + (i * kWaves + 0) * PW + q * VEC);
#pragma unroll
                for (int w = 1; w < kWaves; ++w) {
                    vload<VEC>(t, hpart + (i * kWaves + w) * PW + q * VEC);
#pragma unroll
                    for (int k = 0; k < VEC; ++k) acc[k] += t[k];
                }
                const int r = d[0];
                emit_row<VEC, LPR, MODE, RT>(p, tbuf_wave, r, row0 + r, q, mycol, acc, changed, mirror);
            }
        }
        __syncthreads();   // descriptors may be rewritten by the next tile round
    }

    // ---- phase A: the other long rows, one at a time per wave; the G lane groups split the
    // neighbours and their partial sums are combined by shuffles in a fixed order
    for (int h = posted; h < n_heavy; ++h) {
        const int r = __builtin_amdgcn_readfirstlane(__shfl(s_row, h));
        const int s = __builtin_amdgcn_readfirstlane(__shfl(s_start, h));
        const int e = s + __builtin_amdgcn_readfirstlane(__shfl(s_len, h));
        float acc[VEC];
        gather_range<VEC, LPR>(p, Xc, s, e, lane, g, col_active, acc);
        if (g == 0 && col_active)
            emit_row<VEC, LPR, MODE, RT>(p, tbuf_wave, r, row0 + r, q, mycol, acc, changed, mirror);
    }

    // ---- phase B: the other rows, G at a time, one row per lane group: no cross-lane
    // reduction, G independent gather streams per wave, JU loads in flight in each.
    // The walk over (pass, chunk of LPR neighbours) is software-pipelined: the ids of the
    // NEXT chunk are requested before the current chunk's gathers, so a wave pays one
    // memory latency per chunk instead of two.
    if (n_heavy < nrows) {
        int pos = n_heavy, t0 = 0;
        int src = pos + g;
        int r = __shfl(s_row, src & 63);
        int st = __shfl(s_start, src & 63);
        int len = __shfl(s_len, src & 63);
        if (src >= nrows) len = 0;
        int maxlen = __builtin_amdgcn_readfirstlane(__shfl(s_len, pos));  // sorted: longest of the pass
        int iv = (q < len) ? ldidx(p, st + q) : 0;
        float acc[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
        while (true) {
            // where the walk goes next (wave-uniform)
            const bool same_pass = t0 + LPR < maxlen;
            const int npos = same_pass ? pos : pos + G;
            const int nt0 = same_pass ? t0 + LPR : 0;
            const bool more = npos < nrows;
            int nr = r, nst = st, nlen = len, nmax = maxlen, niv = 0;
            if (!same_pass) {
                const int nsrc = npos + g;
                nr = __shfl(s_row, nsrc & 63);
                nst = __shfl(s_start, nsrc & 63);
                nlen = __shfl(s_len, nsrc & 63);
                if (nsrc >= nrows) nlen = 0;
                nmax = __builtin_amdgcn_readfirstlane(__shfl(s_len, npos & 63));
            }
            if (more && nt0 + q < nlen) niv = ldidx(p, nst + nt0 + q);

            // the current chunk: up to LPR neighbours of each of the G rows
#pragma unroll
            for (int jb = 0; jb < LPR; jb += JU) {
                if (t0 + jb < maxlen) {
                    float v[JU][VEC];
                    int idx[JU];
#pragma unroll
                    for (int j = 0; j < JU; ++j) {
                        idx[j] = __shfl(iv, gbase + jb + j) & p.idx_mask;
                        if (!(col_active && t0 + jb + j < len)) idx[j] = kSkip;
                    }
#pragma unroll
                    for (int j = 0; j < JU; ++j) {
                        if (idx[j] >= 0) {
                            vload<VEC>(v[j], Xc + int64_t(idx[j]) * p.ldx);
                        } else {
#pragma unroll
                            for (int i = 0; i < VEC; ++i) v[j][i] = 0.f;
                        }
                    }
#pragma unroll
                    for (int j = 0; j < JU; ++j)
#pragma unroll
                        for (int i = 0; i < VEC; ++i) acc[i] += v[j][i];
                }
            }
            if (!same_pass) {
                if (pos + g < nrows && col_active)
                    emit_row<VEC, LPR, MODE, RT>(p, tbuf_wave, r, row0 + r, q, mycol, acc, changed, mirror);
#pragma unroll
                for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
            }
            if (!more) break;
            pos = npos; t0 = nt0; r = nr; st = nst; len = nlen; maxlen = nmax; iv = niv;
        }
    }

    if constexpr (TILE) {
        __syncthreads();
        const int cols_here = int(imin(PW, p.L - c0));
        const int64_t tb = p.tblock;
        const int rows_out = (TRANS || mirror) ? nrows : 0;
        if (MODE == kTrans && p.tvec && rows_out == RT && cols_here == PW) {
            // whole tile, aligned destination: 16-byte stores of 4 consecutive rows
            for (int x = lane; x < PW * (RT / 4); x += 64) {
                const int c = x / (RT / 4);
                const int r = (x % (RT / 4)) * 4;
                const float* t = tbuf_wave + c * (RT + 1) + r;
                const float v4[4] = {t[0], t[1], t[2], t[3]};
                const int64_t a = row0 + r;
                const int64_t blk = a / tb;
                const int64_t a_in = a - blk * tb;
                const int64_t stride = p.tstride ? p.tstride : imin(tb, p.M - blk * tb) + p.tpad;
                float* dst = p.Y + blk * (p.L * (tb + p.tpad)) + (c0 + c) * stride + a_in;
                if (p.nt) vstore_nt<4>(dst, v4); else vstore<4>(dst, v4);
            }
        } else {
        for (int x = lane; x < PW * RT; x += 64) {
            const int c = x / RT;
            const int r = x % RT;
            if (c < cols_here && r < rows_out) {
                const int64_t a = row0 + r;
                const int64_t blk = a / tb;
                const int64_t a_in = a - blk * tb;
                const int64_t stride = p.tstride ? p.tstride : imin(tb, p.M - blk * tb) + p.tpad;
                float* dst = p.Y + blk * (p.L * (tb + p.tpad)) + (c0 + c) * stride + a_in;
                if (p.nt) __builtin_nontemporal_store(tbuf_wave[c * (RT + 1) + r], dst);
                else *dst = tbuf_wave[c * (RT + 1) + r];
            }
        }
        }
    }
    }

    if constexpr (!TRANS) {
        if (p.has_ep && p.prev) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) changed += __shfl_down(changed, off);
            if (lane == 0 && changed)
                atomicAdd(p.n_changed + ((blockIdx.x * 4u + (threadIdx.x >> 6)) * 7u) % SIMRANK_CHANGED_SLOTS,
                          (unsigned long long)changed);
        }
    }
}

template <int VEC, int LPR, int MODE, int RT>
static int launch_spmm(SpmmArgs a, hipStream_t st) {
    constexpr int PW = VEC * LPR;
    a.n_panels = int((a.L + PW - 1) / PW);
    if (MODE == kShard && a.sh_ntiles > 0) a.n_panels = a.sh_ntiles;     // one stage of the leg
    const int64_t rows_per_block = int64_t(kWaves) * RT;
    a.row_tiles = int((a.M + rows_per_block - 1) / rows_per_block);
    if (RT != 32 || (MODE == kSym && !a.sym_map)) a.tile_row0 = nullptr;
    if (a.tile_row0) a.row_tiles = (a.n_tiles + kWaves - 1) / kWaves;
    const int64_t panels_padded = a.xcd_map ? int64_t((a.n_panels + 7) / 8) * 8 : a.n_panels;
    int64_t grid = panels_padded * a.row_tiles;
    if constexpr (MODE == kSym) {
        static_assert(PW == 32 && RT == 32, "triangular block map assumes 128-row blocks, 32-col panels");
        const int64_t J = (a.n_panels + 7) / 8;      // panels per XCD
        grid = 8 * (J * J + J);                       // longest list (b = 2), others padded
        if (a.tile_row0) grid = a.sym_blocks;
    }
    SR_REQUIRE(grid > 0 && grid < (int64_t(1) << 31), "grid of %lld blocks", (long long)grid);
    const size_t lds = sizeof(float) * ((MODE != kPlain ? size_t(kWaves) * PW * (RT + 1) : 0) +
                                        (a.has_huge ? 64 + size_t(kMaxHuge) * kWaves * PW : 0));
    auto kern = spmm_gather_kernel<VEC, LPR, MODE, RT>;
    if (lds > 48 * 1024)
        SR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, st, a);
    SR_HIP(hipGetLastError());
    return SIMRANK_OK;
}

// ---------------------------------------------------------------------------------------
// The lean form of the same kernel (round 2).  Counters and an ablation (every neighbour id
// forced to one row: the gathers become L1 hits, nothing else changes) showed that the generic
// kernel above is bound by instruction issue, not by the memory path: its ~18 instructions per
// gather (two 64-bit multiply-adds, zero fills, an exec-masked branch per gather) and a 64-bit
// division per piece of the transposed store cost more than the L2.  This version fixes
// VEC = 4, LPR = 8 (32-float panels, 32-row tiles) and
//   * addresses a neighbour row with one 24-bit multiply (row id x pitch in 16-byte units) and
//     one 64-bit shift-add;
//   * issues the gathers of a chunk unconditionally: slots that every row of the pass has are
//     plain adds, the ragged rest is multiplied by a 0/1 mask (fma(v, 1, acc) == v + acc
//     bit for bit; masked slots read row 0, which is valid memory);
//   * skips the bitonic sort when the tile's rows already come in ascending length (the solver's
//     node order), preloads the row scales, and computes block offsets of the transposed store
//     per tile, not per element.
//   * requests a row's partial sums of the matrix-core part before its gathers, four slabs at a time.
// Same phases and the same summation order of a row's neighbours as the generic kernel (A0: huge rows
// over the four waves; A: long rows over the 8 lane groups; B: one row per lane group); the dense
// partial sums are added as one term (slabs summed first) in both kernels: same bits from either.
// Deterministic, and identical for 1 or P shards.
// ---------------------------------------------------------------------------------------
template <bool IDS16>
__device__ __forceinline__ int ld_id(const SpmmArgs& p, int j) {
    if constexpr (IDS16) return int(p.col16[j]);
    else return p.col[j];
}

// Where a lane gathers from.  Two addressing forms:
//  * Src64: a 64-bit per-lane base + row id x pitch (24-bit multiply, 64-bit shift-add).  Slots a row
//    does not have read row 0 and are multiplied by a 0/1 mask.
//  * Src32 (whenever the operand rows a panel can touch span < 2 GiB: always for panel-blocked
//    operands, and for the N x N/P blocks of sharded ranks): a buffer descriptor over the panel's
//    slice + a 32-bit byte offset = ONE VALU instruction per gather (v_mad_u32_u24), and the
//    hardware range check does the masking: slots a row does not have carry the id `sent`, whose
//    offset lies past the descriptor's end, so they load zeros and every add is unmasked.
struct Src64 {
    const float* Xc;
    uint32_t pitch16;
    static constexpr bool kRangeChecked = false;
    __device__ int sent() const { return 0; }
};
struct Src32 {
    __amdgpu_buffer_rsrc_t srd;
    uint32_t pitch;          // bytes between operand rows
    uint32_t qoff;           // this lane's byte offset inside a row segment
    int sentinel;            // row id whose offset is out of range
    static constexpr bool kRangeChecked = true;
    __device__ int sent() const { return sentinel; }
};

__device__ __forceinline__ float4 ld_row(const Src64& s, int idx) {
    return reinterpret_cast<const float4*>(s.Xc)[size_t(uint32_t(__umul24(uint32_t(idx), s.pitch16)))];
}
__device__ __forceinline__ float4 ld_row(const Src32& s, int idx) {
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    const v4u v = __builtin_amdgcn_raw_buffer_load_b128(s.srd, int(__umul24(uint32_t(idx), s.pitch) + s.qoff), 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// N slots of a chunk: slot j reads the row whose id sits in lane lane0 + j * LS of `ids`.
// MASKED (Src64 only): slot j counts for this lane only when j < rem.
template <int N, int LS, bool MASKED, typename SRC>
__device__ __forceinline__ void gather_slots(const SRC& src, int ids, int lane0, int rem, float (&acc)[4]) {
    float4 v[N > 0 ? N : 1];
#pragma unroll
    for (int j = 0; j < N; ++j) v[j] = ld_row(src, __shfl(ids, lane0 + j * LS));
#pragma unroll
    for (int j = 0; j < N; ++j) {
        if constexpr (MASKED && !SRC::kRangeChecked) {
            const float m = j < rem ? 1.0f : 0.0f;
            acc[0] = fmaf(v[j].x, m, acc[0]);
            acc[1] = fmaf(v[j].y, m, acc[1]);
            acc[2] = fmaf(v[j].z, m, acc[2]);
            acc[3] = fmaf(v[j].w, m, acc[3]);
        } else {
            acc[0] += v[j].x;
            acc[1] += v[j].y;
            acc[2] += v[j].z;
            acc[3] += v[j].w;
        }
    }
}

// n_max (wave-uniform, 0..8) slots are needed by some lane group, n_full by all of them.  At most
// four gathers are in flight per call (16 registers of data): the kernel's latency hiding comes
// from 7-8 waves per SIMD, not from long per-wave queues.
template <int LS, typename SRC>
__device__ __forceinline__ void gather_half(const SRC& src, int ids, int lane0, int n_max, int n_full,
                                            int rem, float (&acc)[4]) {
    if (SRC::kRangeChecked || n_full >= n_max) {
        switch (n_max) {
            case 4: gather_slots<4, LS, false>(src, ids, lane0, rem, acc); break;
            case 3: gather_slots<3, LS, false>(src, ids, lane0, rem, acc); break;
            case 2: gather_slots<2, LS, false>(src, ids, lane0, rem, acc); break;
            case 1: gather_slots<1, LS, false>(src, ids, lane0, rem, acc); break;
            default: break;
        }
    } else {
        switch (n_max) {
            case 4: gather_slots<4, LS, true>(src, ids, lane0, rem, acc); break;
            case 3: gather_slots<3, LS, true>(src, ids, lane0, rem, acc); break;
            case 2: gather_slots<2, LS, true>(src, ids, lane0, rem, acc); break;
            case 1: gather_slots<1, LS, true>(src, ids, lane0, rem, acc); break;
            default: break;
        }
    }
}

template <int LS, typename SRC>
__device__ __forceinline__ void gather_chunk(const SRC& src, int ids, int lane0, int n_max, int n_full,
                                             int rem, float (&acc)[4]) {
#ifdef SIMRANK_DEPTH8
    // experiment build (tools/build_variant.sh): chunks of 5..8 slots issue all their gathers together
    if (SRC::kRangeChecked && n_max > 4) {
        switch (n_max) {
            case 8: gather_slots<8, LS, false>(src, ids, lane0, rem, acc); break;
            case 7: gather_slots<7, LS, false>(src, ids, lane0, rem, acc); break;
            case 6: gather_slots<6, LS, false>(src, ids, lane0, rem, acc); break;
            default: gather_slots<5, LS, false>(src, ids, lane0, rem, acc); break;
        }
        return;
    }
#endif
    gather_half<LS>(src, ids, lane0, min(n_max, 4), min(n_full, 4), rem, acc);
    if (n_max > 4) gather_half<LS>(src, ids, lane0 + 4 * LS, n_max - 4, n_full - 4, rem - 4, acc);
}

// Sum of the X segments of the neighbours at CSR positions [s, e) (s, e wave-uniform): blocks of
// 64 ids, one per lane; slot u of lane group g is neighbour 8 u + g of the block; partial sums of
// the 8 groups combined by shuffles in a fixed order.  Every lane ends with the total.
template <bool IDS16, typename SRC>
__device__ __forceinline__ void gather_range3(const SpmmArgs& p, const SRC& src, int s, int e, int lane,
                                              int g, float (&acc)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = 0.f;
    for (int base = s; base < e; base += 64) {
        const int n = min(64, e - base);
        const int myid = (lane < n && !(p.probe & 8)) ? (ld_id<IDS16>(p, base + lane) & p.idx_mask) : src.sent();
        if (!(p.probe & 1))
            gather_chunk<8>(src, myid, g, (n + 7) >> 3, n >> 3, (n - g + 7) >> 3, acc);
    }
#pragma unroll
    for (int off = 8; off < 64; off <<= 1)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] += __shfl_xor(acc[i], off);
}

// Partial sums of the matrix-core part for row a (blockdense.hip), its slabs added in slab order.
// Called BEFORE the row's gathers are issued, so these loads are in flight beside them instead
// of forming a dependent chain in the epilogue (0.6 ms of leg 1 at pl32768 when they did).
// (ns, slab0): number and first index of the slabs of the row's 128-row block — the same for every row of
// a tile, so the caller looks them up ONCE per tile (wave-uniform): a tile outside the dense blocks (most
// are) then pays no dependent load per pass.
template <bool DENSE>
__device__ __forceinline__ void dense_partial(const SpmmArgs& p, int64_t a, int64_t mycol, bool on,
                                              int ns, int slab0, float (&dsum)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) dsum[i] = 0.f;
    if constexpr (!DENSE) return;
    if (ns == 0 || !on) return;
    const float* dp = p.dpart + (int64_t(slab0) * 128 + (a & 127)) * p.ldp + mycol;
    int s = 0;
    for (; s + 4 <= ns; s += 4, dp += 4 * 128 * p.ldp) {       // four independent loads at a time
        float d0[4], d1[4], d2[4], d3[4];
        vload_nt<4>(d0, dp);
        vload_nt<4>(d1, dp + 128 * p.ldp);
        vload_nt<4>(d2, dp + 2 * 128 * p.ldp);
        vload_nt<4>(d3, dp + 3 * 128 * p.ldp);
#pragma unroll
        for (int i = 0; i < 4; ++i) dsum[i] = (((dsum[i] + d0[i]) + d1[i]) + d2[i]) + d3[i];
    }
    for (; s < ns; ++s, dp += 128 * p.ldp) {
        float d[4];
        vload_nt<4>(d, dp);
#pragma unroll
        for (int i = 0; i < 4; ++i) dsum[i] += d[i];
    }
}

// emit_row of the generic kernel with the row scale and the dense partial sums handed in
template <int MODE>
__device__ __forceinline__ void emit_row3(const SpmmArgs& p, float* tbuf_wave, int r_local, int64_t a,
                                          int q, int64_t mycol, int64_t colofs, float rowscale,
                                          const float (&acc)[4], const float (&dsum)[4], unsigned& changed,
                                          bool mirror, bool check_prev, const float* old_pre = nullptr) {
    // colofs: where this lane's 4 columns start inside a row of Y / prev / prior / evidence
    // (row-major: the column; panel-blocked: the panel's base + 4 q, rows then 32 apart)
    constexpr int RT = 32;
    const float sc = rowscale * (p.has_ep ? p.coef : 1.0f);
    float o[4] = {acc[0] + dsum[0], acc[1] + dsum[1], acc[2] + dsum[2], acc[3] + dsum[3]};
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] *= sc;
    if constexpr (MODE == kTrans) {
        float* t = tbuf_wave + (q * 4) * (RT + 1) + r_local;
#pragma unroll
        for (int i = 0; i < 4; ++i) t[i * (RT + 1)] = o[i];
    } else {
        const int nvalid = int(imin(4, p.L - mycol));
        if (p.has_ep) {
            if (p.ev) {
                const unsigned w = __builtin_nontemporal_load(
                    reinterpret_cast<const unsigned*>(p.ev + a * p.ld_ev + colofs));
                o[0] *= 1.0f - __builtin_ldexpf(1.0f, -int(w & 255u));
                o[1] *= 1.0f - __builtin_ldexpf(1.0f, -int((w >> 8) & 255u));
                o[2] *= 1.0f - __builtin_ldexpf(1.0f, -int((w >> 16) & 255u));
                o[3] *= 1.0f - __builtin_ldexpf(1.0f, -int(w >> 24));
            }
            if (p.ap) {
                float pr[4];
                vload_nt<4>(pr, p.ap + a * p.ld_ap + colofs);
                const float keep = 1.0f - p.lbd;
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = keep * o[i] + p.lbd * pr[i];
            }
            if (p.set_diag) {
                const int64_t d = a - (p.diag_col0 + mycol);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (d == i) o[i] = 1.0f;
            }
            if (check_prev) {       // (false: no previous iterate, or count_any and a difference is known)
                float old[4];
                if (old_pre) {      // (phase B: requested before the pass's gathers, round 5)
#pragma unroll
                    for (int i = 0; i < 4; ++i) old[i] = old_pre[i];
                } else {
                    vload_nt<4>(old, p.prev + a * p.ld_prev + colofs);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    changed += (i < nvalid && fabs(double(o[i]) - double(old[i])) > p.eps)
                                   ? (mirror ? 2u : 1u) : 0u;
            }
        }
        if constexpr (MODE == kSym || MODE == kShard) {
            if (mirror) {
                float* t = tbuf_wave + (q * 4) * (RT + 1) + r_local;
#pragma unroll
                for (int i = 0; i < 4; ++i) t[i * (RT + 1)] = o[i];
            }
        }
        float* y = p.Y + a * p.ldy + colofs;
        if (p.probe & 2) {
        } else if (nvalid == 4) {
            if (p.nt) vstore_nt<4>(y, o); else vstore<4>(y, o);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i < nvalid) y[i] = o[i];
        }
    }
}

// RESTRICT (leg 2 of SimRank++, SimRank.py:315-316, :361): a separate instantiation, so the plain
// legs keep their register budget; chosen by the driver when few segments of E are live.
#ifdef SIMRANK_STAMPS
// Diagnostic build only (bash tools/build_variant.sh stamps -DSIMRANK_STAMPS; tools/stamps.py): where
// a wave's cycles go.  s_memtime at the phase boundaries; lane 0 of every wave adds the differences
// to g_stamps[phase] (a __device__ array nothing else reads; simrank_read_stamps copies it out).
// Never in the product build.
__device__ unsigned long long g_stamps[16];
__device__ __forceinline__ unsigned long long stamp_now() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define STAMP(i)                                                          \
    do {                                                                  \
        const unsigned long long now_ = stamp_now();                      \
        if (lane == 0) atomicAdd(&g_stamps[i], now_ - stamp_prev);        \
        stamp_prev = now_;                                                \
    } while (0)
#else
#define STAMP(i) do {} while (0)
#endif

// DENSE: the launch adds partial sums of the matrix-core part; launches without them (no dense plan,
// or the upper-triangle leg of a power-law graph) run an instantiation that does not carry the four
// registers of those sums: 60 instead of 68 VGPRs for the transposed leg (8 waves per SIMD), 79
// instead of 89 for the others (6 instead of 5): leg 2 -5...7 %.
// A32: Src32 addressing (see above).
template <int MODE, bool IDS16, bool RESTRICT, bool DENSE, bool A32>
#ifndef SIMRANK_LB_DELTA
#define SIMRANK_LB_DELTA 0      // experiment builds lower the occupancy bounds by this much
#endif
__global__ __launch_bounds__(256, (MODE == kTrans ? ((DENSE && !A32) ? 7 : 8) : (DENSE ? 5 : 6)) - SIMRANK_LB_DELTA)
void gather3_kernel(const SpmmArgs p) {
    constexpr bool TRANS = MODE == kTrans;
    constexpr bool TILE = MODE != kPlain;
    constexpr int PW = 32, RT = 32, LPR = 8;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if (p.probe & 32) return;                                    // diagnostic: the price of the bare launch

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#ifdef SIMRANK_STAMPS
    unsigned long long stamp_prev = stamp_now();
#endif

    int panel, rt;
    if (MODE == kSym && p.sym_map) {
        panel = p.sym_map[2 * blockIdx.x];
        rt = p.sym_map[2 * blockIdx.x + 1];
        if (panel < 0) return;
    } else if constexpr (MODE == kSym) {
        const int x = int(blockIdx.x & 7);
        const int t = int(blockIdx.x >> 3);
        const int b1 = x >> 2;
        int j = int((sqrtf(float(b1 * b1 + 4 * t)) - float(b1)) * 0.5f);
        while ((j + 1) * (j + 1) + b1 * (j + 1) <= t) ++j;
        while (j * j + b1 * j > t) --j;
        panel = x + 8 * j;
        rt = t - (j * j + b1 * j);
        if (rt > panel / 4) return;
    } else {
        const uint32_t bid = blockIdx.x;
        if (p.xcd_map) {
            const uint32_t local = bid >> 3;
            panel = int(local / uint32_t(p.row_tiles)) * 8 + int(bid & 7);
            rt = int(local % uint32_t(p.row_tiles));
        } else {
            panel = int(bid / uint32_t(p.row_tiles));
            rt = int(bid % uint32_t(p.row_tiles));
        }
        if (p.tile_row0) rt = p.tile_row0[p.n_tiles + 1 + rt];     // groups of tiles, most entries first
    }
    if (panel >= p.n_panels) return;
    if ((p.probe & 16) && (blockIdx.x & 7) != 0) return;      // diagnostic: one XCD's share of the launch only
    if constexpr (MODE == kShard) {
        // column tile j works on the row tiles i <= j of every shard: the late panels are the heavy
        // ones and go first (a panel stays on one XCD: n_panels is a multiple of 8 or the tail is short)
        panel = p.sh_tile0 + p.n_panels - 1 - panel;
        // the workgroup's first tile has its smallest row: if even that lies in a tile i > j, nothing to do
        const int t0 = rt * kWaves;
        const int first = p.tile_row0 ? (t0 < p.n_tiles ? p.tile_row0[t0] : int(p.M)) : t0 * RT;
        if (first >= p.M) return;
        const int h0 = int(uint32_t(first) / uint32_t(p.sh_mb));
        // (a workgroup's tiles may straddle two shards: then the later tiles start again at i = 0)
        const int last = p.tile_row0 ? (t0 + kWaves < p.n_tiles ? p.tile_row0[t0 + kWaves] : int(p.M))
                                     : min(int(p.M), (t0 + kWaves) * RT);
        if (int(uint32_t(last - 1) / uint32_t(p.sh_mb)) == h0 && ((first - h0 * p.sh_mb) >> 5) > panel) return;
    }

    const int64_t c0 = int64_t(panel) * PW;
    const int g = lane >> 3;
    const int q = lane & 7;
    const int gbase = lane & ~7;
    const int64_t mycol = c0 + int64_t(q) * 4;
    const bool col_active = mycol < p.L;
    // lanes past the last column gather from column 0 (valid memory); nothing of theirs is stored
    using SRC = typename std::conditional<A32, Src32, Src64>::type;
    SRC xs;
    if constexpr (A32) {
        // descriptor over everything this panel can touch: the panel's slice (blocked), or from its
        // first column to the end of the matrix (row-major; the padding of a row is readable)
        const float* base = p.blocked ? p.X + (int64_t(panel) * p.x_rows_pad) * 32 : p.X + c0;
        xs.srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, int(p.x_span_bytes - (p.blocked ? 0 : c0 * 4)),
                                                    0x00020000);
        xs.pitch = uint32_t(p.ldx) * 4u;
        xs.qoff = uint32_t(q) * 16u;
        xs.sentinel = p.x_sentinel;
    } else {
        xs.Xc = p.blocked ? p.X + (int64_t(panel) * p.x_rows_pad) * 32 + q * 4
                           : p.X + (col_active ? mycol : 0);
        xs.pitch16 = uint32_t(p.ldx >> 2);                       // (blocked: ldx = 32)
    }
    const int64_t colofs = p.blocked ? (int64_t(panel) * p.y_rows_pad) * 32 + q * 4 : mycol;
    float* tbuf_wave = smem + (TILE ? wave * PW * (RT + 1) : 0);
    unsigned changed = 0;
    // count_any: has a wave that shares this wave's counter already found an element that moved?  (vector load at agent scope — a
    // scalar load could be served a stale zero by the constant cache for the rest of the launch; a stale
    // value only means this wave still compares.)  Issued here, needed at the first emitted row.
    unsigned long long seen = 0;
    const unsigned slot = ((blockIdx.x * 4u + unsigned(wave)) * 7u) % SIMRANK_CHANGED_SLOTS;
    if constexpr (!TRANS) {
        {
            // (the wave's own striped counter: one shared flag word would be a hot spot in one L2 channel.)  Round 5: an
            // UNCONDITIONAL buffer load past the L1 (sc1; a descriptor of zero bytes when there is nothing to watch: no
            // access, the value 0) — the generic-pointer atomic load inside a branch compiled to a FLAT load (which counts
            // against lgkmcnt as well) with its wait right behind it: a trip to the L2 at the head of the prologue of
            // every one of the launch's 275 k workgroups
            typedef unsigned v2u __attribute__((ext_vector_type(2)));
            const bool watch = p.has_ep && p.prev && p.count_any;
            const __amdgpu_buffer_rsrc_t csrd = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<unsigned long long*>(p.n_changed), 0, watch ? SIMRANK_CHANGED_SLOTS * 8 : 0, 0x00020000);
            const v2u w = __builtin_amdgcn_raw_buffer_load_b64(csrd, int(slot * 8u), 0, 16);
            seen = (unsigned long long)w.x | ((unsigned long long)w.y << 32);
        }
    }

    int row0 = (rt * kWaves + wave) * RT;
    int nrows = int(imin(RT, p.M - row0));
    if (p.tile_row0) {
        // (wave-uniform: both ends of the tile through the scalar cache, requested together — they were two vector loads
        // with a full wait between them)
        const int t = rt * kWaves + wave;
        const int tc = __builtin_amdgcn_readfirstlane(max(0, min(t, p.n_tiles - 1)));
        const int r0 = p.tile_row0[tc], r1 = p.tile_row0[tc + 1];
        row0 = t < p.n_tiles ? r0 : int(p.M);
        nrows = t < p.n_tiles ? r1 - r0 : 0;
    }
    bool mirror = false;
    if constexpr (MODE == kSym) {
        const int rb = row0 & ~(RT - 1);
        if (rb > c0) nrows = 0;
        mirror = rb < c0;
    }
    int shard_h = 0, tile_i = 0;
    if constexpr (MODE == kShard) {
        shard_h = int(uint32_t(row0) / uint32_t(p.sh_mb));
        tile_i = (row0 - shard_h * p.sh_mb) >> 5;
        if (tile_i > panel) nrows = 0;          // the rank that owns columns (h, i) computes the mirror image
        mirror = tile_i < panel;
    }

    // ---- rows of the tile by length, longest first
    int my_start = 0, my_len = -1;
    float my_scale = 0.f;
    if (lane < nrows) {
        my_start = p.rowptr[row0 + lane];
        my_len = p.rowptr[row0 + lane + 1] - my_start;
        my_scale = p.rowscale[row0 + lane];
    }
    // slabs of the matrix-core part: a tile lies inside one 128-row block
    int tile_ns = 0, tile_slab0 = 0;
    if constexpr (DENSE) {
        if (p.dpart && !(p.probe & 4) && nrows > 0) {
            tile_ns = p.dnslab[row0 >> 7];            // (uniform addresses: scalar loads)
            tile_slab0 = p.dslab0[row0 >> 7];
        }
    }
    int s_row, s_len, s_start;
    {
        const int nxt = __shfl_down(my_len, 1);
        const bool out_of_order = lane + 1 < nrows && nxt < my_len;
        if (__ballot(out_of_order) == 0) {
            // already ascending (the solver's node order): longest first = reversed
            s_row = lane < nrows ? nrows - 1 - lane : lane;
            s_len = __shfl(my_len, s_row);
            s_start = __shfl(my_start, s_row);
            if (lane >= nrows) s_len = -1;
        } else {
            int key = lane < nrows ? ((my_len << 6) | lane) : -64 + lane;
#pragma unroll
            for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
                for (int j = k >> 1; j > 0; j >>= 1) {
                    const int other = __shfl_xor(key, j);
                    const bool desc = (lane & k) == 0;
                    const bool lower = (lane & j) == 0;
                    const bool take = (lower == desc) ? (other > key) : (other < key);
                    key = take ? other : key;
                }
            }
            s_row = key & 63;
            s_len = key >> 6;
            s_start = __shfl(my_start, s_row);
        }
    }
    const int n_heavy = __popcll(__ballot(s_len >= kHeavy));
    const bool check_prev = !TRANS && p.has_ep && p.prev && __builtin_amdgcn_readfirstlane(int(seen != 0)) == 0;
    STAMP(0);      // prologue: arguments, tile lookup, row pointers, order

    // ---- phase A0: huge rows, split over the four waves of the workgroup (see the generic kernel)
    int posted = 0;
    if (p.has_huge) {
        int* hmeta = reinterpret_cast<int*>(smem + (TILE ? kWaves * PW * (RT + 1) : 0));
        float* hpart = reinterpret_cast<float*>(hmeta + 64);
        const int n_huge = __popcll(__ballot(s_len >= p.huge_len));
        if (lane == 0) hmeta[60 + wave] = n_huge;
        __syncthreads();
        int slot0 = 0, total = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {
            const int c = hmeta[60 + w];
            slot0 += w < wave ? c : 0;
            total += c;
        }
        posted = max(0, min(n_huge, kMaxHuge - slot0));
        if (lane < posted) {
            int* d = hmeta + 4 + 4 * (slot0 + lane);
            d[0] = s_row; d[1] = s_start; d[2] = s_len; d[3] = wave;
        }
        __syncthreads();
        const int nh = min(total, kMaxHuge);
        for (int i = 0; i < nh; ++i) {
            const int* d = hmeta + 4 + 4 * i;
            const int hs = d[1], hl = d[2];
            const int chunk = ((hl + kWaves - 1) / kWaves + 63) & ~63;
            const int s = hs + wave * chunk;
            const int e = min(hs + hl, s + chunk);
            float part[4];
            gather_range3<IDS16>(p, xs, s, e, lane, g, part);
            if (g == 0) vstore<4>(hpart + (i * kWaves + wave) * PW + q * 4, part);
        }
        __syncthreads();
        for (int i = 0; i < nh; ++i) {
            const int* d = hmeta + 4 + 4 * i;
            if (d[3] == wave) {
                const int r = d[0];
                const float sc = __shfl(my_scale, r);
                if (g == 0 && col_active) {
                    float acc[4], t[4], dsum[4];
                    dense_partial<DENSE>(p, int64_t(row0) + r, mycol, true, tile_ns, tile_slab0, dsum);
                    vload<4>(acc, hpart + (i * kWaves + 0) * PW + q * 4);
#pragma unroll
                    for (int w = 1; w < kWaves; ++w) {
                        vload<4>(t, hpart + (i * kWaves + w) * PW + q * 4);
#pragma unroll
                        for (int k = 0; k < 4; ++k) acc[k] += t[k];
                    }
                    emit_row3<MODE>(p, tbuf_wave, r, int64_t(row0) + r, q, mycol, colofs, sc, acc, dsum, changed, mirror, check_prev);
                }
            }
        }
        __syncthreads();
    }

    STAMP(1);      // phase A0
    // ---- phase A: the other long rows, one at a time, the 8 lane groups splitting the neighbours
    for (int h = posted; h < n_heavy; ++h) {
        const int r = __builtin_amdgcn_readfirstlane(__shfl(s_row, h));
        const int s = __builtin_amdgcn_readfirstlane(__shfl(s_start, h));
        const int e = s + __builtin_amdgcn_readfirstlane(__shfl(s_len, h));
        const float sc = __shfl(my_scale, r);
        float acc[4], dsum[4];
        dense_partial<DENSE>(p, int64_t(row0) + r, mycol, g == 0 && col_active, tile_ns, tile_slab0, dsum);
        bool live = true;
        if constexpr (RESTRICT) {
            const unsigned w = col_active ? *reinterpret_cast<const unsigned*>(
                                                p.ev + (int64_t(row0) + r) * p.ld_ev + colofs) : 0u;
            live = __ballot(w != 0u) != 0;                           // uniform: the row's 32 columns
        }
        if (live) {
            gather_range3<IDS16>(p, xs, s, e, lane, g, acc);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = 0.f;
        }
        if (g == 0 && col_active)
            emit_row3<MODE>(p, tbuf_wave, r, int64_t(row0) + r, q, mycol, colofs, sc, acc, dsum, changed, mirror, check_prev);
    }

    STAMP(2);      // phase A
    // ---- phase B: the other rows, 8 at a time, one row per lane group; the ids of the next chunk
    // are requested before the gathers of the current one
    if (n_heavy < nrows) {
        int pos = n_heavy, t0 = 0;
        int src = pos + g;
        int r = __shfl(s_row, src & 63);
        int st = __shfl(s_start, src & 63);
        int len = __shfl(s_len, src & 63);
        if (src >= nrows) len = 0;
        float sc = __shfl(my_scale, r & 63);
        int maxlen = __builtin_amdgcn_readfirstlane(len);           // sorted: group 0 has the longest
        int minlen = __builtin_amdgcn_readlane(len, 63);            // ... group 7 the shortest (0: no row)
        int iv = (q < len && !(p.probe & 8)) ? (ld_id<IDS16>(p, st + q) & p.idx_mask) : xs.sent();
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        float dsum[4];
        dense_partial<DENSE>(p, int64_t(row0) + r, mycol, src < nrows && col_active, tile_ns, tile_slab0, dsum);
        // RESTRICT: a lane group whose 32 evidence counts are all zero issues no gathers
        auto group_live = [&](int row, bool on) -> bool {
            if constexpr (!RESTRICT) return true;
            const unsigned w = on ? *reinterpret_cast<const unsigned*>(
                                        p.ev + (int64_t(row0) + row) * p.ld_ev + colofs) : 0u;
            return ((__ballot(w != 0u) >> gbase) & 0xFFull) != 0;
        };
        bool glive = group_live(r, src < nrows && col_active);
        // (round 5) the previous iterate's values of the pass's rows — what the exact convergence count compares with — are
        // requested BEFORE the pass's gathers (they depend on nothing the pass computes); they used to be loaded when a row
        // was emitted: a trip to memory behind every pass of eight rows
        float old[4] = {0.f, 0.f, 0.f, 0.f};
        auto load_prev = [&](int row, bool on) {
            // (32-bit operand addressing only: with 64-bit gather addresses the four registers no longer fit 6 waves per SIMD)
            if constexpr (!TRANS && A32) {
                if (check_prev && on) vload_nt<4>(old, p.prev + (int64_t(row0) + row) * p.ld_prev + colofs);
            }
        };
        load_prev(r, src < nrows && col_active);
        while (true) {
            const bool same_pass = t0 + LPR < maxlen;
            const int npos = same_pass ? pos : pos + 8;
            const int nt0 = same_pass ? t0 + LPR : 0;
            const bool more = npos < nrows;
            int nr = r, nst = st, nlen = len, nmax = maxlen, nmin = minlen, niv = xs.sent();
            float nsc = sc;
            if (!same_pass && more) {
                const int nsrc = npos + g;
                nr = __shfl(s_row, nsrc & 63);
                nst = __shfl(s_start, nsrc & 63);
                nlen = __shfl(s_len, nsrc & 63);
                if (nsrc >= nrows) nlen = 0;
                nsc = __shfl(my_scale, nr & 63);
                nmax = __builtin_amdgcn_readfirstlane(nlen);
                nmin = __builtin_amdgcn_readlane(nlen, 63);
            }
            if (more && nt0 + q < nlen && !(p.probe & 8)) niv = ld_id<IDS16>(p, nst + nt0 + q) & p.idx_mask;

            if (!(p.probe & 1) && glive)
                gather_chunk<1>(xs, iv, gbase, min(LPR, maxlen - t0), max(0, min(LPR, minlen - t0)),
                                len - t0, acc);

            if (!same_pass) {
                if (pos + g < nrows && col_active)
                    emit_row3<MODE>(p, tbuf_wave, r, int64_t(row0) + r, q, mycol, colofs, sc, acc, dsum, changed, mirror, check_prev,
                                    (!TRANS && A32) ? old : nullptr);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = 0.f;
                if (more) {
                    dense_partial<DENSE>(p, int64_t(row0) + nr, mycol, npos + g < nrows && col_active, tile_ns, tile_slab0, dsum);
                    glive = group_live(nr, npos + g < nrows && col_active);
                    load_prev(nr, npos + g < nrows && col_active);
                }
            }
            if (!more) break;
            pos = npos; t0 = nt0; r = nr; st = nst; len = nlen; maxlen = nmax; minlen = nmin; iv = niv; sc = nsc;
        }
    }

    STAMP(3);          // phase B
    if constexpr (TILE) {
        __syncthreads();
        STAMP(4);      // waiting for the other waves of the workgroup
        const int cols_here = int(imin(PW, p.L - c0));
        const int rows_out = ((TRANS || mirror) && !(p.probe & 2)) ? nrows : 0;
        if (MODE == kShard && rows_out > 0) {
            // the transposed tile: element (c, r) is S'[global row of my column c0 + c][column (h, i) * 32 + r]
            const int in32 = row0 & 31;
            float* base;
            int64_t rstride;
            if (shard_h == p.sh_rank) {
                base = p.Y + (p.diag_col0 + c0) * p.ldy + (tile_i * 32 + in32);
                rstride = p.ldy;
            } else {
                const int64_t slot = int64_t(panel) * (panel - 1) / 2 + tile_i - p.sh_slot0;
                base = p.sh_send + int64_t(shard_h) * p.sh_chunk + slot * 1024 + in32;
                rstride = 32;
            }
            if ((rows_out & 3) == 0 && (row0 & 3) == 0) {
#pragma unroll
                for (int it = 0; it < PW * (RT / 4) / 64; ++it) {
                    const int x = lane + it * 64;
                    const int c = x >> 3;
                    const int r4 = (x & 7) * 4;
                    if (r4 < rows_out) {
                        const float* t = tbuf_wave + c * (RT + 1) + r4;
                        const float v4[4] = {t[0], t[1], t[2], t[3]};
                        vstore<4>(base + c * rstride + r4, v4);
                    }
                }
            } else {
                for (int x = lane; x < PW * RT; x += 64) {
                    const int c = x >> 5;
                    const int r = x & 31;
                    if (r < rows_out) base[c * rstride + r] = tbuf_wave[c * (RT + 1) + r];
                }
            }
        } else if (rows_out > 0 && p.blocked) {
            // panel-blocked Y^T: the tile's 32 c-rows are consecutive 128-byte lines of panel row0 / 32
            float* base = p.Y + ((int64_t(row0 >> 5) * p.y_rows_pad) + c0) * 32 + (row0 & 31);
            if (cols_here == PW && (rows_out & 3) == 0 && (row0 & 3) == 0) {
                // whole tile, or an aligned piece of a cut tile (16, 8, 4 rows): 16-byte stores
#pragma unroll
                for (int it = 0; it < PW * (RT / 4) / 64; ++it) {
                    const int x = lane + it * 64;
                    const int c = x >> 3;
                    const int r4 = (x & 7) * 4;
                    if (r4 < rows_out) {
                        const float* t = tbuf_wave + c * (RT + 1) + r4;
                        const float v4[4] = {t[0], t[1], t[2], t[3]};
                        float* dst = base + c * 32 + r4;
                        if (p.nt) vstore_nt<4>(dst, v4); else vstore<4>(dst, v4);
                    }
                }
            } else {
                for (int x = lane; x < PW * RT; x += 64) {
                    const int c = x >> 5;
                    const int r = x & 31;
                    if (c < cols_here && r < rows_out) base[c * 32 + r] = tbuf_wave[c * (RT + 1) + r];
                }
            }
        } else if (rows_out > 0) {
            // where the tile lands: Y^T rows are the tile's columns; block h of t_block rows is
            // contiguous.  A tile never crosses more than one block boundary when t_block >= 32.
            const int64_t tb = p.tblock;
            const bool one_block = p.tstride != 0;
            const uint32_t blk0 = one_block ? 0u : uint32_t(row0) / uint32_t(tb);
            const int in0 = row0 - int(blk0 * uint32_t(tb));          // row0's index inside its block
            if (MODE == kTrans && p.tvec && rows_out == RT && cols_here == PW && (one_block || tb >= RT)) {
                // whole tile, aligned destination: 16-byte stores of 4 consecutive rows
                const int64_t rows0 = one_block ? p.tstride : imin(tb, p.M - int64_t(blk0) * tb) + p.tpad;
                const int64_t rows1 = one_block ? p.tstride : imin(tb, p.M - int64_t(blk0 + 1) * tb) + p.tpad;
                float* base0 = p.Y + int64_t(blk0) * (p.L * (tb + p.tpad)) + c0 * rows0 + in0;
                float* base1 = p.Y + int64_t(blk0 + 1) * (p.L * (tb + p.tpad)) + c0 * rows1 + (in0 - int(tb));
#pragma unroll
                for (int it = 0; it < PW * (RT / 4) / 64; ++it) {
                    const int x = lane + it * 64;
                    const int c = x >> 3;
                    const int r4 = (x & 7) * 4;
                    const float* t = tbuf_wave + c * (RT + 1) + r4;
                    const float v4[4] = {t[0], t[1], t[2], t[3]};
                    const bool second = !one_block && in0 + r4 >= tb;
                    float* dst = second ? base1 + int64_t(c) * rows1 + r4 : base0 + int64_t(c) * rows0 + r4;
                    if (p.nt) vstore_nt<4>(dst, v4); else vstore<4>(dst, v4);
                }
            } else {
                for (int x = lane; x < PW * RT; x += 64) {
                    const int c = x >> 5;
                    const int r = x & 31;
                    if (c < cols_here && r < rows_out) {
                        const int64_t a = int64_t(row0) + r;
                        const int64_t blk = one_block ? 0 : a / tb;
                        const int64_t a_in = a - blk * tb;
                        const int64_t stride = one_block ? p.tstride : imin(tb, p.M - blk * tb) + p.tpad;
                        float* dst = p.Y + blk * (p.L * (tb + p.tpad)) + (c0 + c) * stride + a_in;
                        if (p.nt) __builtin_nontemporal_store(tbuf_wave[c * (RT + 1) + r], dst);
                        else *dst = tbuf_wave[c * (RT + 1) + r];
                    }
                }
            }
        }
    }

    STAMP(5);          // tile store issued
    if constexpr (!TRANS) {
        if (p.has_ep && p.prev) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) changed += __shfl_down(changed, off);
            if (lane == 0 && changed) atomicAdd(p.n_changed + slot, (unsigned long long)changed);
        }
    }
}

template <int MODE>
static int launch_gather3(SpmmArgs a, hipStream_t st) {
    constexpr int PW = 32, RT = 32;
    a.n_panels = int((a.L + PW - 1) / PW);
    if (MODE == kShard && a.sh_ntiles > 0) a.n_panels = a.sh_ntiles;     // one stage of the leg
    const int64_t rows_per_block = int64_t(kWaves) * RT;
    a.row_tiles = int((a.M + rows_per_block - 1) / rows_per_block);
    if (MODE == kSym && !a.sym_map) a.tile_row0 = nullptr;
    if (a.tile_row0) a.row_tiles = (a.n_tiles + kWaves - 1) / kWaves;
    const int64_t panels_padded = a.xcd_map ? int64_t((a.n_panels + 7) / 8) * 8 : a.n_panels;
    int64_t grid = panels_padded * a.row_tiles;
    if constexpr (MODE == kSym) {
        const int64_t J = (a.n_panels + 7) / 8;
        grid = 8 * (J * J + J);
        if (a.tile_row0) grid = a.sym_blocks;
    }
    SR_REQUIRE(grid > 0 && grid < (int64_t(1) << 31), "grid of %lld blocks", (long long)grid);
    const size_t lds = sizeof(float) * ((MODE != kPlain ? size_t(kWaves) * PW * (RT + 1) : 0) +
                                        (a.has_huge ? 64 + size_t(kMaxHuge) * kWaves * PW : 0));
    const bool restricted = MODE != kTrans && a.has_ep && a.ev && a.restrict_support;
    const bool dense = a.dpart != nullptr;
    const dim3 gr((unsigned)grid), bl(256);
    // 32-bit buffer addressing when one more row than the operand has still fits under 2 GiB
    const int64_t x_rows = a.blocked ? a.x_rows_pad : a.K;
    const int64_t span = a.blocked ? a.x_rows_pad * 128 : ((x_rows - 1) * a.ldx + a.ldx) * 4;
    const bool a32 = a.addr32 && (x_rows + 1) * a.ldx * 4 + 256 < (int64_t(1) << 31) &&
                     a.ldx * 4 < (int64_t(1) << 24) && x_rows < (int64_t(1) << 24) - 1;
    a.x_span_bytes = span;
    a.x_sentinel = (int32_t)x_rows;
#define SR_LAUNCH3(IDS, RES, DEN) \
    do { if (a32) hipLaunchKernelGGL((gather3_kernel<MODE, IDS, RES, DEN, true>), gr, bl, lds, st, a); \
         else hipLaunchKernelGGL((gather3_kernel<MODE, IDS, RES, DEN, false>), gr, bl, lds, st, a); } while (0)
#define SR_LAUNCH3_IDS(RES, DEN) \
    do { if (a.col16) SR_LAUNCH3(true, RES, DEN); else SR_LAUNCH3(false, RES, DEN); } while (0)
    if constexpr (MODE != kTrans) {
        if (restricted) {
            if (dense) SR_LAUNCH3_IDS(true, true); else SR_LAUNCH3_IDS(true, false);
            SR_HIP(hipGetLastError());
            return SIMRANK_OK;
        }
    }
    if (dense) SR_LAUNCH3_IDS(false, true); else SR_LAUNCH3_IDS(false, false);
#undef SR_LAUNCH3_IDS
#undef SR_LAUNCH3
    SR_HIP(hipGetLastError());
    return SIMRANK_OK;
}

// ---------------------------------------------------------------------------------------
// K0: identity columns
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fill_identity_kernel(float* S, int64_t n_rows,
                                                            int64_t n_cols, int64_t ld,
                                                            int64_t col0) {
    const int64_t total = n_rows * n_cols;
    for (int64_t t = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; t < total;
         t += int64_t(gridDim.x) * blockDim.x) {
        const int64_t a = t / n_cols;
        const int64_t c = t - a * n_cols;
        S[a * ld + c] = (a == col0 + c) ? 1.0f : 0.0f;
    }
}

// panel-blocked: the buffer is zeroed by a memset, this sets the ones
__global__ __launch_bounds__(256) void set_diagonal_blocked_kernel(float* S, int64_t n_rows, int64_t n_cols,
                                                                   int64_t rows_pad, int64_t col0) {
    for (int64_t c = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; c < n_cols;
         c += int64_t(gridDim.x) * blockDim.x)
        if (col0 + c < n_rows) S[elem_at(col0 + c, c, 0, rows_pad)] = 1.0f;
}

// ---------------------------------------------------------------------------------------
// K4/K5 stand-alone: element-wise epilogue over a block (asymmetric-prior path)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void epilogue_kernel(const float* __restrict__ Q, int64_t ldq,
                                                       float* Y, int64_t ldy, int64_t n_rows,
                                                       int64_t n_cols, const SpmmArgs p) {
    // (p.y_rows_pad > 0: every operand is panel-blocked with that many rows per panel; a thread
    // then walks (panel, row, column in panel) so that consecutive threads touch consecutive floats)
    const int64_t rp = p.y_rows_pad;
    const int64_t width = rp ? ((n_cols + 31) >> 5) * 32 : n_cols;
    const int64_t total = n_rows * width;
    unsigned changed = 0;
    for (int64_t t = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; t < total;
         t += int64_t(gridDim.x) * blockDim.x) {
        int64_t a, c;
        if (rp) {
            const int64_t pn = t / (n_rows * 32), rest = t - pn * (n_rows * 32);
            a = rest >> 5;
            c = pn * 32 + (rest & 31);
            if (c >= n_cols) continue;
        } else {
            a = t / n_cols;
            c = t - a * n_cols;
        }
        float v = Q[elem_at(a, c, ldq, rp)] * p.coef;
        if (p.ev) v *= 1.0f - __builtin_ldexpf(1.0f, -int(p.ev[elem_at(a, c, p.ld_ev, rp)]));
        if (p.ap) v = (1.0f - p.lbd) * v + p.lbd * p.ap[elem_at(a, c, p.ld_ap, rp)];
        if (p.set_diag && a == p.diag_col0 + c) v = 1.0f;
        if (p.prev) changed += fabs(double(v) - double(p.prev[elem_at(a, c, p.ld_prev, rp)])) > p.eps ? 1u : 0u;
        Y[elem_at(a, c, ldy, rp)] = v;
    }
    if (p.prev) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) changed += __shfl_down(changed, off);
        if ((threadIdx.x & 63) == 0 && changed)
            atomicAdd(p.n_changed + ((blockIdx.x * 4u + (threadIdx.x >> 6)) * 7u) % SIMRANK_CHANGED_SLOTS,
                      (unsigned long long)changed);
    }
}

// ---------------------------------------------------------------------------------------
// top-k per row: one wave per row, k rounds of "largest element after the previous pick" in
// the total order (value descending, column id ascending; the id of block column c is
// col_ids[c] when the caller works in a permuted node order, col0 + c otherwise).  Exact and deterministic; the row is
// re-read k times from L2 (a 128 KiB row stays resident).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void topk_rows_kernel(const float* __restrict__ S, int64_t ld,
                                                        int64_t rows_pad, int64_t n_rows, int64_t n_cols,
                                                        int64_t col0, int k, int exclude_diag,
                                                        const int32_t* __restrict__ col_ids,
                                                        int32_t* idx_out, float* val_out) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (blockIdx.x * int64_t(blockDim.x) + threadIdx.x) >> 6;
    const int64_t nwaves = (int64_t(gridDim.x) * blockDim.x) >> 6;
    for (int64_t a = wave; a < n_rows; a += nwaves) {
        const int64_t skip = exclude_diag ? a - col0 : -1;
        float pv = __builtin_inff();   // previous pick: everything is "after" (+inf, -1)
        int pi = -1;
        for (int j = 0; j < k; ++j) {
            float bv = -__builtin_inff();
            int bi = 0x7fffffff;
            for (int64_t c = lane; c < n_cols; c += 64) {
                const float v = S[elem_at(a, c, ld, rows_pad)];
                const int id = col_ids ? col_ids[c] : int(col0 + c);
                const bool after = (v < pv) || (v == pv && id > pi);
                const bool better = (v > bv) || (v == bv && id < bi);
                if (c != skip && after && better) { bv = v; bi = id; }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const float ov = __shfl_xor(bv, off);
                const int oi = __shfl_xor(bi, off);
                if ((ov > bv) || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
            }
            const bool found = bi != 0x7fffffff;
            if (lane == 0) {
                idx_out[a * k + j] = found ? bi : -1;
                val_out[a * k + j] = found ? bv : 0.f;
            }
            if (!found) {
                for (int jj = j + 1; jj < k; ++jj)
                    if (lane == 0) { idx_out[a * k + jj] = -1; val_out[a * k + jj] = 0.f; }
                break;
            }
            pv = bv;
            pi = bi;
        }
    }
}

// The same selection in ONE pass over the row for k <= K (16 or 32): every lane keeps the K best of its own
// columns in registers, sorted by the same total order (an element that does not beat the lane's K-th is one
// compare), then the wave takes the best head k times.  At N = 65536 the k + 1 passes of the kernel above move
// 190 GB for k = 10; this one 17 GB.
template <int K>
__global__ __launch_bounds__(256) void topk_rows_onepass_kernel(const float* __restrict__ S, int64_t ld,
                                                                int64_t n_rows, int64_t n_cols, int64_t col0,
                                                                int k, int exclude_diag,
                                                                const int32_t* __restrict__ col_ids,
                                                                int32_t* idx_out, float* val_out) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (blockIdx.x * int64_t(blockDim.x) + threadIdx.x) >> 6;
    const int64_t nwaves = (int64_t(gridDim.x) * blockDim.x) >> 6;
    const bool vec = (ld & 3) == 0 && (reinterpret_cast<uintptr_t>(S) & 15) == 0 &&
                     (!col_ids || (reinterpret_cast<uintptr_t>(col_ids) & 15) == 0);
    for (int64_t a = wave; a < n_rows; a += nwaves) {
        const int64_t skip = exclude_diag ? a - col0 : -1;
        float tv[K];
        int ti[K];
#pragma unroll
        for (int i = 0; i < K; ++i) { tv[i] = -__builtin_inff(); ti[i] = 0x7fffffff; }
        const float* row = S + a * ld;
        auto offer = [&](float v, int id, int64_t c) __attribute__((always_inline)) {
            if (c != skip && ((v > tv[K - 1]) || (v == tv[K - 1] && id < ti[K - 1]))) {
                // insert: carry the element down the sorted list, swapping where it is better
#pragma unroll
                for (int i = 0; i < K; ++i) {
                    const bool better = (v > tv[i]) || (v == tv[i] && id < ti[i]);
                    const float nv = better ? tv[i] : v;
                    const int ni = better ? ti[i] : id;
                    tv[i] = better ? v : tv[i];
                    ti[i] = better ? id : ti[i];
                    v = nv;
                    id = ni;
                }
            }
        };
        // 16 bytes per lane and load (rows are 16-byte aligned: ld is a multiple of 4), two loads in flight
        typedef float v4f32 __attribute__((ext_vector_type(4)));
        typedef int v4i32 __attribute__((ext_vector_type(4)));
        const int64_t n4 = vec ? (n_cols & ~int64_t(3)) : 0;
        for (int64_t c = 4 * lane; c < n4; c += 512) {
            const int64_t c2 = c + 256;
            const bool two = c2 < n4;
            const v4f32 x0 = __builtin_nontemporal_load(reinterpret_cast<const v4f32*>(row + c));
            const v4f32 x1 = two ? __builtin_nontemporal_load(reinterpret_cast<const v4f32*>(row + c2)) : v4f32{0, 0, 0, 0};
            v4i32 i0 = v4i32{int(col0 + c), int(col0 + c + 1), int(col0 + c + 2), int(col0 + c + 3)};
            v4i32 i1 = v4i32{int(col0 + c2), int(col0 + c2 + 1), int(col0 + c2 + 2), int(col0 + c2 + 3)};
            if (col_ids) {
                i0 = *reinterpret_cast<const v4i32*>(col_ids + c);
                if (two) i1 = *reinterpret_cast<const v4i32*>(col_ids + c2);
            }
            offer(x0.x, i0.x, c); offer(x0.y, i0.y, c + 1); offer(x0.z, i0.z, c + 2); offer(x0.w, i0.w, c + 3);
            if (two) {
                offer(x1.x, i1.x, c2); offer(x1.y, i1.y, c2 + 1); offer(x1.z, i1.z, c2 + 2); offer(x1.w, i1.w, c2 + 3);
            }
        }
        for (int64_t c = n4 + lane; c < n_cols; c += 64)
            offer(row[c], col_ids ? col_ids[c] : int(col0 + c), c);
        for (int j = 0; j < k; ++j) {
            float bv = tv[0];
            int bi = ti[0];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const float ov = __shfl_xor(bv, off);
                const int oi = __shfl_xor(bi, off);
                if ((ov > bv) || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
            }
            const bool found = bi != 0x7fffffff;
            if (lane == 0) {
                idx_out[a * k + j] = found ? bi : -1;
                val_out[a * k + j] = found ? bv : 0.f;
            }
            if (found && ti[0] == bi) {          // ids are distinct: exactly one lane owns the pick; it pops its head
#pragma unroll
                for (int i = 0; i + 1 < K; ++i) { tv[i] = tv[i + 1]; ti[i] = ti[i + 1]; }
                tv[K - 1] = -__builtin_inff();
                ti[K - 1] = 0x7fffffff;
            }
        }
    }
}

// One pass over a PANEL-BLOCKED matrix (32-column panels of rows_pad rows): a wave takes EIGHT consecutive rows, lane
// group g = lane >> 3 owns row 8 w + g, lane q = lane & 7 the columns 4 q .. 4 q + 3 of every panel — one load
// instruction reads the eight rows' segments of a panel, 1 KiB contiguous (a single row of this layout is 128-byte
// pieces rows_pad x 128 bytes apart: what made the callers copy the matrix to row-major first, 2 x N^2 x 4 bytes of
// traffic the selection itself does not need).  Each lane keeps the K best of its columns, the eight lanes of a
// group merge k times.  Same total order as the kernels above.
template <int K>
__global__ __launch_bounds__(256) void topk_rows_blocked_onepass_kernel(const float* __restrict__ S, int64_t rows_pad,
                                                                        int64_t n_rows, int64_t n_cols, int64_t col0,
                                                                        int k, int exclude_diag,
                                                                        const int32_t* __restrict__ col_ids,
                                                                        int32_t* idx_out, float* val_out) {
    const int lane = threadIdx.x & 63, g = lane >> 3, q = lane & 7;
    const int64_t wave = (blockIdx.x * int64_t(blockDim.x) + threadIdx.x) >> 6;
    const int64_t nwaves = (int64_t(gridDim.x) * blockDim.x) >> 6;
    const int64_t n_panels = (n_cols + 31) / 32;
    typedef float v4f32 __attribute__((ext_vector_type(4)));
    for (int64_t a8 = wave * 8; a8 < n_rows; a8 += nwaves * 8) {
        const int64_t a = a8 + g;
        const bool live = a < n_rows;
        const int64_t skip = exclude_diag ? a - col0 : -1;
        float tv[K];
        int ti[K];
#pragma unroll
        for (int i = 0; i < K; ++i) { tv[i] = -__builtin_inff(); ti[i] = 0x7fffffff; }
        auto offer = [&](float v, int id, int64_t c) __attribute__((always_inline)) {
            if (c < n_cols && c != skip && ((v > tv[K - 1]) || (v == tv[K - 1] && id < ti[K - 1]))) {
#pragma unroll
                for (int i = 0; i < K; ++i) {
                    const bool better = (v > tv[i]) || (v == tv[i] && id < ti[i]);
                    const float nv = better ? tv[i] : v;
                    const int ni = better ? ti[i] : id;
                    tv[i] = better ? v : tv[i];
                    ti[i] = better ? id : ti[i];
                    v = nv;
                    id = ni;
                }
            }
        };
        if (live) {
            const float* base = S + (a * 32 + 4 * q);
            // two panels in flight
            for (int64_t pn = 0; pn < n_panels; pn += 2) {
                const bool two = pn + 1 < n_panels;
                const v4f32 x0 = __builtin_nontemporal_load(reinterpret_cast<const v4f32*>(base + pn * rows_pad * 32));
                const v4f32 x1 = two ? __builtin_nontemporal_load(reinterpret_cast<const v4f32*>(base + (pn + 1) * rows_pad * 32))
                                     : v4f32{0, 0, 0, 0};
                const int64_t c = pn * 32 + 4 * q, c2 = c + 32;
                int i0[4], i1[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    i0[i] = col_ids ? (c + i < n_cols ? col_ids[c + i] : 0) : int(col0 + c + i);
                    i1[i] = col_ids ? (two && c2 + i < n_cols ? col_ids[c2 + i] : 0) : int(col0 + c2 + i);
                }
                offer(x0.x, i0[0], c); offer(x0.y, i0[1], c + 1); offer(x0.z, i0[2], c + 2); offer(x0.w, i0[3], c + 3);
                if (two) {
                    offer(x1.x, i1[0], c2); offer(x1.y, i1[1], c2 + 1); offer(x1.z, i1[2], c2 + 2); offer(x1.w, i1[3], c2 + 3);
                }
            }
        }
        for (int j = 0; j < k; ++j) {
            float bv = tv[0];
            int bi = ti[0];
#pragma unroll
            for (int off = 4; off > 0; off >>= 1) {          // the eight lanes of the row's group
                const float ov = __shfl_xor(bv, off);
                const int oi = __shfl_xor(bi, off);
                if ((ov > bv) || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
            }
            const bool found = bi != 0x7fffffff;
            if (live && q == 0) {
                idx_out[a * k + j] = found ? bi : -1;
                val_out[a * k + j] = found ? bv : 0.f;
            }
            if (found && ti[0] == bi) {          // ids are distinct: exactly one lane of the group owns the pick
#pragma unroll
                for (int i = 0; i + 1 < K; ++i) { tv[i] = tv[i + 1]; ti[i] = ti[i + 1]; }
                tv[K - 1] = -__builtin_inff();
                ti[K - 1] = 0x7fffffff;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// dst[i][j] = src[row_idx[i]][col_idx[j]] (a NULL index list = identity): moves a matrix
// between the solver's node order (rows sorted by length) and the caller's.  One workgroup
// per destination row: the source row is read scattered (it sits in L2), written coalesced.
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void permute_kernel(const T* __restrict__ src, int64_t ld_src,
                                                      int64_t pad_src, T* __restrict__ dst,
                                                      int64_t ld_dst, int64_t pad_dst, int64_t n_rows,
                                                      int64_t n_cols,
                                                      const int32_t* __restrict__ row_idx,
                                                      const int32_t* __restrict__ col_idx) {
    // pad_* > 0: that side is panel-blocked with so many rows per panel (this is also how a
    // matrix moves between the two layouts)
    for (int64_t i = blockIdx.x; i < n_rows; i += gridDim.x) {
        const int64_t ri = row_idx ? row_idx[i] : i;
        for (int64_t j = threadIdx.x; j < n_cols; j += blockDim.x)
            dst[elem_at(i, j, ld_dst, pad_dst)] = src[elem_at(ri, col_idx ? col_idx[j] : j, ld_src, pad_src)];
    }
}

// ---------------------------------------------------------------------------------------
// K7: evidence counts.  One workgroup per row a: LDS counters for a chunk of columns,
// incremented along every 2-hop path a <- i -> b; saturated to u8 on the way out.
// ---------------------------------------------------------------------------------------
// Round 3: one workgroup of 16 waves per row, ALL columns of the block in one pass — two 16-bit counters
// per LDS word, 65536 columns = 128 KiB (counters saturate far above the 255 that is stored).
// Round 2 walked every path once per 16384-column pass (4 x at N = 65536) with 4 waves per workgroup and a
// rowscale load per path; the transposed pattern now lists live rows only (simrank_graph_create).
constexpr int kEvChunk = 65536;  // columns per pass
// (round 5, measured and dropped: with the hub columns on the matrix cores what is left per row is a short walk behind a chain
// of dependent loads; four passes of 16384 columns with 512 threads — four workgroups per CU instead of one — paid that chain
// four times: 4 x 3.5 ms against 6.3 ms in one pass at N = 65536)

template <int kEvThreads>
__global__ __launch_bounds__(kEvThreads) void evidence_counts_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
    const float* __restrict__ rowscale, const int32_t* __restrict__ t_rowptr,
    const int32_t* __restrict__ t_col, const int32_t* __restrict__ t_pos, int64_t M, int64_t col0, int n_cols,
    uint8_t* out, int64_t ld, int64_t rows_pad, int64_t out_col0, int vec4, int tri,
    const int32_t* __restrict__ hubidx, int regroup_rows) {
    extern __shared__ unsigned cnt[];                 // counters of columns 2 w and 2 w + 1 in word w
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int words = (n_cols + 1) >> 1;
    // tri (round 4; the whole square block in one pass): the counts are symmetric — common in-neighbours of (a, b)
    // = of (b, a) — so row a counts the paths to b >= a only (half the LDS atomics, which bound this kernel) and
    // writes its columns from the 32-column panel of the diagonal on; evidence_mirror_kernel fills the rest.
    // Rows are taken LONGEST FIRST there (the long rows, late in the solver's order, have the short ranges).
    // Blocked layout: a row's 32 bytes of a panel share a 128-byte line with its three neighbours' — rows 4 k .. 4 k + 3 are
    // given to workgroups of ONE XCD that run at the same time (blocks b, b + 8, b + 16, b + 24), so that its L2 reads and
    // writes the line once instead of four L2s a quarter each (round 5: 11.3 GB through the fabric for 4.3 GB of counts).
    const bool regroup = regroup_rows && rows_pad > 0 && (gridDim.x & 7) == 0;
    const int64_t M_it = regroup ? (M + 31) / 32 * 32 : M;
    for (int64_t it0 = blockIdx.x; it0 < M_it; it0 += gridDim.x) {
        int64_t it = it0;
        if (regroup) {
            const int64_t q = it0 >> 3, x = it0 & 7;
            it = (((q >> 2) << 3) + x) * 4 + (q & 3);
            if (it >= M) continue;
        }
        const int64_t a = tri ? M - 1 - it : it;
        // first column (of this pass's n_cols) this row zeroes, counts from and writes; tri in several passes (out_col0 =
        // where the pass starts in the square): a row below the pass's columns has nothing to do in it
        const int64_t first64 = tri ? (a & ~int64_t(31)) - out_col0 : 0;
        if (first64 >= n_cols) continue;
        const int first = first64 > 0 ? int(first64) : 0;
        if (hubidx) {
            // (round 5) the pairs through the HUB columns were counted on the matrix cores (evidence_hub_kernel) and sit in
            // `out` already: the counters start from them, and the walk below leaves those columns out
            if (vec4) {
                for (int c4 = first + tid * 4; c4 < n_cols; c4 += kEvThreads * 4) {
                    const uint8_t* src = out + elem_at(a, out_col0 + c4, ld, rows_pad);
                    unsigned v = 0;
                    if (c4 + 4 <= n_cols) v = *reinterpret_cast<const unsigned*>(src);
                    else for (int k = 0; c4 + k < n_cols; ++k) v |= unsigned(src[k]) << (8 * k);
                    cnt[c4 >> 1] = (v & 0xFFu) | ((v >> 8) & 0xFFu) << 16;
                    if (c4 + 2 < n_cols) cnt[(c4 >> 1) + 1] = ((v >> 16) & 0xFFu) | (v >> 24) << 16;
                }
            } else {
                for (int w = (first >> 1) + tid; w < words; w += kEvThreads) {
                    const int c = 2 * w;
                    const unsigned lo = out[elem_at(a, out_col0 + c, ld, rows_pad)];
                    const unsigned hi = c + 1 < n_cols ? out[elem_at(a, out_col0 + c + 1, ld, rows_pad)] : 0u;
                    cnt[w] = lo | hi << 16;
                }
            }
        } else {
            for (int w = (first >> 1) + tid; w < words; w += kEvThreads) cnt[w] = 0;
        }
        __syncthreads();
        if (rowscale[a] > 0.f) {
            const int s = rowptr[a], e = rowptr[a + 1];
            for (int j = s + wave; j < e; j += kEvThreads / 64) {
                const int i = col[j];
                if (hubidx && hubidx[i] >= 0) continue;
                // (tri: column i's list is ascending and holds a itself at t_pos[j]: from there on it is b >= a)
                const int ts = tri ? t_pos[j] : t_rowptr[i], te = t_rowptr[i + 1];
                for (int t = ts + lane; t < te; t += 64) {
                    const int64_t c = int64_t(t_col[t]) - col0;
                    if (c >= 0 && c < n_cols) {
                        // a counter stops growing at 0x4000 (only min(count, 255) is stored); the threads that
                        // read it just below can overshoot by at most one each (1024): never a carry into
                        // the neighbouring counter
                        const unsigned sh = (unsigned(c) & 1u) * 16u;
                        unsigned* wp = &cnt[c >> 1];
                        if (((*(volatile unsigned*)wp >> sh) & 0xFFFFu) < 0x4000u) atomicAdd(wp, 1u << sh);
                    }
                }
            }
        }
        __syncthreads();
        if (vec4) {
            // four columns per thread: two words -> four saturated bytes, one 4-byte store (a panel of the
            // blocked layout and an aligned row-major row both keep 4 consecutive columns together)
            for (int c4 = first + tid * 4; c4 < n_cols; c4 += kEvThreads * 4) {
                const unsigned w0 = cnt[c4 >> 1], w1 = (c4 + 2 < n_cols) ? cnt[(c4 >> 1) + 1] : 0u;
                const unsigned b0 = min(w0 & 0xFFFFu, 255u), b1 = min(w0 >> 16, 255u);
                const unsigned b2 = min(w1 & 0xFFFFu, 255u), b3 = min(w1 >> 16, 255u);
                uint8_t* dst = out + elem_at(a, out_col0 + c4, ld, rows_pad);
                if (c4 + 4 <= n_cols) {
                    *reinterpret_cast<unsigned*>(dst) = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
                } else {
                    dst[0] = (uint8_t)b0;
                    if (c4 + 1 < n_cols) dst[1] = (uint8_t)b1;
                    if (c4 + 2 < n_cols) dst[2] = (uint8_t)b2;
                }
            }
        } else {
            for (int c = first + tid; c < n_cols; c += kEvThreads)
                out[elem_at(a, out_col0 + c, ld, rows_pad)] = (uint8_t)min((cnt[c >> 1] >> ((c & 1) * 16)) & 0xFFFFu, 255u);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------
// K7h (round 5): the pairs through the HUB columns on the matrix cores.  counts(a, b) = sum over common in-neighbours i
// (SimRank.py:315 `Graph.dot(Graph.T)` on the 0/1 pattern); for the H columns i that very many rows share, the 0/1
// image P (live rows x H, one byte each) is built once and counts_hub = P . P^T is an integer GEMM:
// v_mfma_i32_32x32x32_i8, exact.  A workgroup owns a 128 x 128 tile of the counts, a wave 64 x 64 of it (2 x 2 MFMA
// tiles); A and B fragments come from the same matrix by the same rule (lane (m, h): 16 bytes of row m from byte
// k0 + 16 h), so whatever order the instruction takes k in, both operands agree on it.  Saturated to u8 on the way out
// through a 1 KiB LDS tile per wave (16-byte stores along the count rows, either layout).  tri: the tiles right of or on
// the diagonal only (the LDS-counter kernel and the mirror pass do the rest, as without hubs).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void hub_image_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                        const float* __restrict__ rowscale,
                                                        const int32_t* __restrict__ hubidx, int64_t M, int Hp,
                                                        uint8_t* __restrict__ P) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t(blockIdx.x) * 256 + threadIdx.x) >> 6, n_waves = (int64_t(gridDim.x) * 256) >> 6;
    for (int64_t a = wave0; a < M; a += n_waves) {
        if (!(rowscale[a] > 0.f)) continue;                     // (`G > 0`: rows without weight take no part)
        for (int j = rowptr[a] + lane; j < rowptr[a + 1]; j += 64) {
            const int h = hubidx[col[j]];
            if (h >= 0) P[a * Hp + h] = 1;
        }
    }
}

__global__ __launch_bounds__(256) void evidence_hub_kernel(const uint8_t* __restrict__ P, int Hp, int64_t M, int64_t col0,
                                                           int64_t n_cols, uint8_t* __restrict__ out, int64_t ld, int64_t rows_pad,
                                                           int tiles_j, int tri, int vec16) {
    typedef int v4i __attribute__((ext_vector_type(4)));
    typedef int v16i __attribute__((ext_vector_type(16)));
    __shared__ __attribute__((aligned(16))) unsigned char tbuf[4][32][48];
    const int ti = int(blockIdx.x / unsigned(tiles_j)), tj = int(blockIdx.x % unsigned(tiles_j));
    if (tri && tj < ti) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 31, h = lane >> 5;
    const int64_t a0 = int64_t(ti) * 128 + (wave >> 1) * 64;          // rows of the counts
    const int64_t c0 = int64_t(tj) * 128 + (wave & 1) * 64;           // columns of the block; node = col0 + column
    const uint8_t* pa = P + (a0 + m) * Hp + 16 * h;
    const uint8_t* pb = P + (col0 + c0 + m) * Hp + 16 * h;
    v16i acc[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[x][y][r] = 0;
    for (int k0 = 0; k0 < Hp; k0 += 32) {
        const v4i fa0 = *reinterpret_cast<const v4i*>(pa + k0), fa1 = *reinterpret_cast<const v4i*>(pa + 32 * Hp + k0);
        const v4i fb0 = *reinterpret_cast<const v4i*>(pb + k0), fb1 = *reinterpret_cast<const v4i*>(pb + 32 * Hp + k0);
        acc[0][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa0, fb0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa0, fb1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa1, fb0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa1, fb1, acc[1][1], 0, 0, 0);
    }
    // C/D layout of the 32 x 32 MFMA: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                tbuf[wave][(r & 3) + 8 * (r >> 2) + 4 * h][m] = (unsigned char)min(acc[x][y][r], 255);
            __builtin_amdgcn_s_waitcnt(0xc07f);                   // (lgkmcnt(0): the wave's own bytes are in its tile)
            __builtin_amdgcn_wave_barrier();
            const int row = lane >> 1, half = lane & 1;
            const int64_t a = a0 + 32 * x + row, c = c0 + 32 * y + 16 * half;
            if (a < M && c < n_cols) {
                const unsigned char* src = &tbuf[wave][row][16 * half];
                uint8_t* dst = out + elem_at(a, c, ld, rows_pad);
                if (vec16 && c + 16 <= n_cols) {
                    *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(src);
                } else {
                    for (int k = 0; k < 16 && c + k < n_cols; ++k) dst[k] = src[k];
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
}

// The lower triangle of a square u8 count block from its upper one: 64 x 64 tiles, tile (I, J) with I > J is the
// transpose of tile (J, I); a tile on the diagonal mirrors itself.  One workgroup of 256 threads per tile pair: thread
// (row, piece) moves 16 bytes on both sides (a piece never straddles a 32-column panel of the blocked layout), the
// transposition happens in LDS.  (Round 4: 32 x 32 tiles, 4 bytes per thread: 1.87 ms at N = 65536.)
__global__ __launch_bounds__(256) void evidence_mirror_kernel(uint8_t* cnt, int64_t n, int64_t ld, int64_t rows_pad, int vec16) {
    __shared__ __attribute__((aligned(16))) unsigned char t[64][80];
    const int64_t w = blockIdx.x;
    int64_t I = (int64_t)((sqrt(8.0 * double(w) + 1.0) - 1.0) * 0.5);
    while (I * (I + 1) / 2 > w) --I;
    while ((I + 1) * (I + 2) / 2 <= w) ++I;
    const int64_t J = w - I * (I + 1) / 2;                        // I >= J
    const int tid = threadIdx.x, r = tid >> 2, p = tid & 3;       // row r of the tile, bytes 16 p .. 16 p + 15
    // source tile (J, I): rows 64 J + r, columns 64 I + 16 p ..
    {
        const int64_t row = 64 * J + r, c = 64 * I + 16 * p;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (row < n && c < n) {
            const uint8_t* src = cnt + elem_at(row, c, ld, rows_pad);
            if (vec16 && c + 16 <= n) {
                v = *reinterpret_cast<const uint4*>(src);
            } else {
                unsigned char b[16];
                for (int k = 0; k < 16; ++k) b[k] = c + k < n ? src[k] : 0;
                memcpy(&v, b, 16);
            }
        }
        *reinterpret_cast<uint4*>(&t[r][16 * p]) = v;
    }
    __syncthreads();
    // destination tile (I, J): row 64 I + r, columns 64 J + 16 p .. = source (column r, rows 16 p ..)
    const int64_t row = 64 * I + r, c = 64 * J + 16 * p;
    if (row >= n || c >= n) return;
    unsigned char b[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) b[k] = t[16 * p + k][r];
    if (I == J)                                                   // on the diagonal: only what lies left of it
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (16 * p + k >= r) b[k] = t[r][16 * p + k];
    uint8_t* dst = cnt + elem_at(row, c, ld, rows_pad);
    if (vec16 && c + 16 <= n) {
        uint4 v;
        memcpy(&v, b, 16);
        *reinterpret_cast<uint4*>(dst) = v;
    } else {
        for (int k = 0; k < 16; ++k)
            if (c + k < n) dst[k] = b[k];
    }
}

// 32-column segments (aligned to 32) of a u8 count block that hold a nonzero count
__global__ __launch_bounds__(256) void live_segments_kernel(const uint8_t* __restrict__ cnt, int64_t ld,
                                                            int64_t rows_pad, int64_t n_rows, int64_t n_cols,
                                                            unsigned long long* live) {
    const int64_t segs = (n_cols + 31) / 32;
    const int64_t total = n_rows * segs;
    unsigned mine = 0;
    if (rows_pad > 0 && (reinterpret_cast<uintptr_t>(cnt) & 15) == 0) {
        // panel-blocked: segment (row a, panel s) is 32 aligned bytes, the rows of a panel one after the other —
        // consecutive threads take consecutive rows (two 16-byte loads each; 11 -> 1 ms at N = 65536)
        for (int64_t t = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; t < total;
             t += int64_t(gridDim.x) * blockDim.x) {
            const int64_t sgm = t / n_rows, a = t - sgm * n_rows;
            const uint4* row = reinterpret_cast<const uint4*>(cnt + (sgm * rows_pad + a) * 32);
            const int n = int(imin(32, n_cols - sgm * 32));
            typedef unsigned v4u32 __attribute__((ext_vector_type(4)));
            const v4u32 lo = __builtin_nontemporal_load(reinterpret_cast<const v4u32*>(row));
            const v4u32 hi = __builtin_nontemporal_load(reinterpret_cast<const v4u32*>(row + 1));
            const unsigned w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            unsigned any = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int left = n - 4 * k;                      // valid bytes of this word
                const unsigned mask = left >= 4 ? 0xFFFFFFFFu : left <= 0 ? 0u : (1u << (8 * left)) - 1u;
                any |= w[k] & mask;
            }
            mine += any ? 1u : 0u;
        }
    } else {
        for (int64_t t = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; t < total;
             t += int64_t(gridDim.x) * blockDim.x) {
            const int64_t a = t / segs, sgm = t - a * segs;
            const uint8_t* row = cnt + elem_at(a, sgm * 32, ld, rows_pad);
            const int n = int(imin(32, n_cols - sgm * 32));
            unsigned any = 0;
            for (int c = 0; c < n; ++c) any |= row[c];
            mine += any ? 1u : 0u;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(live, (unsigned long long)mine);
}

// kShard: the mirrored tiles a rank received (packed 32 x 32 tiles, slot j(j-1)/2 + i of the chunk of
// source rank h) go to rows (h, j) x columns (mine, i) of its block.  One workgroup per tile.
__global__ __launch_bounds__(256) void shard_unpack_kernel(float* Y, int64_t ldy, const float* recv,
                                                           int64_t chunk, int rank, int mb, int64_t slot0) {
    const int h = blockIdx.y;                                   // source rank
    if (h == rank) return;
    const int64_t slot = slot0 + blockIdx.x;                    // = j (j - 1) / 2 + i, i < j
    int j = int((1.0 + sqrt(1.0 + 8.0 * double(slot))) * 0.5);
    while (int64_t(j) * (j - 1) / 2 > slot) --j;
    while (int64_t(j + 1) * j / 2 <= slot) ++j;
    const int i = int(slot - int64_t(j) * (j - 1) / 2);
    const float* src = recv + int64_t(h) * chunk + int64_t(blockIdx.x) * 1024;
    float* dst = Y + (int64_t(h) * mb + 32 * j) * ldy + 32 * i;
    const int c = threadIdx.x >> 3, r4 = (threadIdx.x & 7) * 4;
    *reinterpret_cast<float4*>(dst + c * ldy + r4) = *reinterpret_cast<const float4*>(src + c * 32 + r4);
}

}  // namespace simrank

using namespace simrank;

extern "C" {

// simrank_graph_create + simrank_evidence_counts(_blocked) of the same pattern, the counting kernel queued as soon as the CSR /
// CSC arrays are on the device — it runs beside the host threads that build the graph's plans (SimRank.py:311-320 while
// :24-52 is still being digested).  rows_pad > 0: panel-blocked counts (ld ignored).
int simrank_graph_create_counting(int64_t n_rows, int64_t n_cols, int64_t nnz, const int32_t* rowptr, const int32_t* col,
                                  const float* rowscale, int64_t col0, int64_t n_cols_ev, uint8_t* counts, int64_t ld,
                                  int64_t rows_pad, void* stream, simrank_graph** out) {
    SR_REQUIRE(counts && n_rows == n_cols, "evidence counts need a square pattern and a counts block");
    std::function<int(simrank_graph*)> hook = [&](simrank_graph* g) -> int {
        return rows_pad > 0 ? simrank_evidence_counts_blocked(g, col0, n_cols_ev, counts, rows_pad, stream)
                            : simrank_evidence_counts(g, col0, n_cols_ev, counts, ld, stream);
    };
    return graph_create_with(tuning_snapshot(), n_rows, n_cols, nnz, rowptr, col, rowscale, out, &hook);
}

int simrank_evidence_live_segments(const uint8_t* counts, int64_t ld, int64_t rows_pad, int64_t n_rows,
                                   int64_t n_cols, int64_t* live, int64_t* total, void* stream) {
    SR_REQUIRE(counts && live && total && n_rows > 0 && n_cols > 0 &&
                   (rows_pad ? rows_pad >= n_rows : ld >= n_cols), "bad evidence block");
    unsigned long long* d = nullptr;
    SR_HIP(hipMalloc((void**)&d, sizeof(unsigned long long)));
    hipStream_t st = as_stream(stream);
    hipError_t e = hipMemsetAsync(d, 0, sizeof(unsigned long long), st);
    unsigned long long h = 0;
    if (e == hipSuccess) {
        hipLaunchKernelGGL(live_segments_kernel, dim3(256 * 8), dim3(256), 0, st, counts, ld, rows_pad, n_rows,
                           n_cols, d);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&h, d, sizeof(h), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(d);
    if (e != hipSuccess) {
        set_error("simrank_evidence_live_segments: %s", hipGetErrorString(e));
        return SIMRANK_ERR_HIP;
    }
    *live = (int64_t)h;
    *total = n_rows * ((n_cols + 31) / 32);
    return SIMRANK_OK;
}

int simrank_fill_identity(float* S, int64_t n_rows, int64_t n_cols, int64_t ld, int64_t col0,
                          void* stream) {
    SR_REQUIRE(S && n_rows > 0 && n_cols > 0 && ld >= n_cols, "bad identity block");
    const int64_t total = n_rows * n_cols;
    const int grid = (int)std::min<int64_t>((total + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(fill_identity_kernel, dim3(grid), dim3(256), 0, as_stream(stream), S,
                       n_rows, n_cols, ld, col0);
    SR_HIP(hipGetLastError());
    return SIMRANK_OK;
}

static int spmm_impl(const simrank_graph* g, const float* X, int64_t ldx, int64_t n_cols_x,
                     float* Y, int64_t ldy, int32_t transpose_out, int64_t t_block, int64_t t_pad,
                     const simrank_epilogue* ep, void* stream, bool blocked, int64_t x_rows_pad,
                     int64_t y_rows_pad) {
    SR_REQUIRE(t_pad >= 0 && t_pad < 4096, "t_pad out of range");
    SR_REQUIRE(g && X && Y, "NULL argument");
    const Tuning& T = g->tun;        // the knobs as they were when the graph was created
    SR_REQUIRE(!(transpose_out && ep), "an epilogue needs transpose_out = 0");
    SpmmArgs a{};
    a.blocked = blocked ? 1 : 0;
    a.x_rows_pad = x_rows_pad;
    a.y_rows_pad = y_rows_pad;
    a.rowptr = g->rowptr;
    a.col = g->col;
    a.col16 = T.ids16 ? g->col16 : nullptr;
    a.rowscale = g->rowscale;
    a.huge_len = (int32_t)std::max<int64_t>(kHeavy, T.huge);
    a.has_huge = g->max_row_nnz >= a.huge_len ? 1 : 0;
    a.X = X;
    a.ldx = ldx;
    a.L = n_cols_x;
    a.Y = Y;
    a.ldy = ldy;
    a.M = g->n_rows;
    a.K = g->n_cols;
    a.tblock = (t_block <= 0 || t_block > g->n_rows) ? g->n_rows : t_block;
    a.tstride = (transpose_out && a.tblock == g->n_rows && ldy >= g->n_rows) ? ldy : 0;
    a.tpad = a.tstride ? 0 : t_pad;
    // 16-byte transposed stores: every 4-row piece of a full tile must be 16-byte aligned
    // and inside one block (tiles are 16/32/64 rows, blocks start at multiples of t_block)
    a.tvec = aligned16(Y) && (a.tstride ? a.tstride % 4 == 0
                                        : (a.tblock % 4 == 0 && a.tpad % 4 == 0 && g->n_rows % 4 == 0));
    a.xcd_map = (int)T.xcd_map;
    a.idx_mask = (int32_t)T.probe_mask;
    a.addr32 = (int32_t)T.addr32;
    a.probe = (int32_t)T.probe_flags;
    bool vec_ok = aligned16(X) && ldx % 4 == 0;
    if (!transpose_out) vec_ok = vec_ok && aligned16(Y) && ldy % 4 == 0;
    hipStream_t st = as_stream(stream);
    if (ep) {
        a.has_ep = 1;
        a.coef = ep->coef;
        a.lbd = ep->lbd;
        a.ev = ep->evidence;
        a.ld_ev = ep->ld_evidence;
        a.ap = ep->apriori;
        a.ld_ap = ep->ld_apriori;
        a.prev = ep->previous;
        a.ld_prev = ep->ld_previous;
        a.eps = ep->eps;
        a.n_changed = ep->n_changed;
        a.diag_col0 = ep->diag_col0;
        a.set_diag = ep->set_diag;
        a.restrict_support = ep->restrict_support;
        a.count_any = ep->count_any;
        if (blocked) a.ld_ev = a.ld_ap = a.ld_prev = 32;      // rows of a panel are 32 elements apart
        SR_REQUIRE(!a.ev || a.ld_ev >= n_cols_x || blocked, "evidence ld too small");
        SR_REQUIRE(!a.ap || a.ld_ap >= n_cols_x || blocked, "apriori ld too small");
        SR_REQUIRE(!a.prev || ((a.ld_prev >= n_cols_x || blocked) && a.n_changed),
                   "previous needs ld >= columns and a counter");
        if (a.ev) vec_ok = vec_ok && (reinterpret_cast<uintptr_t>(a.ev) % 4 == 0) && a.ld_ev % 4 == 0;
        if (a.ap) vec_ok = vec_ok && aligned16(a.ap) && a.ld_ap % 4 == 0;
        if (a.prev) vec_ok = vec_ok && aligned16(a.prev) && a.ld_prev % 4 == 0;
        if (a.prev) SR_HIP(hipMemsetAsync(a.n_changed, 0, sizeof(unsigned long long) * SIMRANK_CHANGED_SLOTS, st));
    }
    // automatic choice (profiles/ sweep_r01.log): 32 rows per wave tile; 32-float panels for
    // the transposed leg (keeps its LDS tile at 17 KiB -> 7 workgroups per CU), 64 otherwise
    int64_t panel = T.panel;
    if (panel == 0) panel = transpose_out ? 32 : 64;
    int64_t tile = T.tile;
    if (tile == 0) tile = 32;
    a.nt = (int32_t)T.stream_nt;
    const bool want_sym = ep && ep->symmetric && T.triangle && vec_ok && !transpose_out &&
                          n_cols_x == g->n_rows && ep->diag_col0 == 0 && g->n_rows >= 64;
    // leg 1 of a panel-blocked update: one launch, matrix cores + gathers on the same panel slice (fused.hip)
    // (measured against the two-launch leg: -13 % at K = 32768 power-law, -10 % Erdos-Renyi, -6 % at K = 65536)
    if (blocked && transpose_out && g->fused && T.fuse && vec_ok && T.dense_terms == 3 && g->n_cols <= T.fuse_max_rows &&
        (x_rows_pad + 1) * 128 < (int64_t(1) << 31)) {
#ifdef SIMRANK_EXPERIMENT_FUSED2
        if (T.fuse == 2 && g->fused2 && (g->fused2->n_pslots == 0 || (n_cols_x + 31) / 32 <= g->fused2->cap_panels))
            return launch_fused2_trans(g, X, x_rows_pad, n_cols_x, Y, y_rows_pad, st);
#endif
        return launch_fused_trans(g, X, x_rows_pad, n_cols_x, Y, y_rows_pad, st);
    }
    // ... and of a rank of a sharded update: the same launch on its row-major column block, the result in the
    // chunks of the all-to-all (round 4; tuning "fuse_shards")
    if (!blocked && transpose_out && g->fused && T.fuse && T.fuse_shards && vec_ok && T.dense_terms == 3 &&
        g->n_cols <= T.fuse_max_rows) {
        // (one block with pitched rows, Y^T[c * ldy + a], is the chunked layout with t_pad = ldy - n_rows)
        const int64_t tb = a.tstride ? g->n_rows : t_block, tp = a.tstride ? a.tstride - g->n_rows : t_pad;
        if (fused_rowmajor_fits(g, X, ldx, n_cols_x, Y, tb, tp))
            return launch_fused_trans_rowmajor(g, X, ldx, n_cols_x, Y, tb, tp, st);
    }
    // leg 2 of a symmetric panel-blocked update in ONE launch where the dense sets carry it (round 6; tuning "fuse_sym"): the
    // matrix-core phase, the gathered remainder, the epilogue and both stores of a tile by the same workgroup — the two-launch
    // leg below hands the dense part's partial sums over through memory
    if (blocked && want_sym && !transpose_out && ep && (!ep->restrict_support || T.fuse_sym > 0) &&
        fused_sym_applies(g, x_rows_pad, n_cols_x, y_rows_pad))
        return launch_fused_sym(g, X, x_rows_pad, n_cols_x, Y, y_rows_pad, a.coef, a.lbd, a.eps, a.ev, a.ap, a.prev, a.n_changed,
                                a.set_diag, a.count_any, st);
    // the block-dense part goes to the matrix cores first; the gather then runs on the remainder
    // In the upper-triangle form only when the pattern is dense throughout (MovieLens-like: 87 % of
    // the entries in dense sets, leg 2 0.9 -> 0.4 ms): on a power-law pattern the long rows, which
    // own the dense sets, compute only the few columns right of the diagonal there, and the
    // partial sums cost as much as they save.
    const simrank_dense_plan* dp = (vec_ok && T.dense_min > 0 && tile == 32) ? g->dense : nullptr;
    if (dp && want_sym) {
        const int64_t mode = T.dense_sym;
        if (mode == 0 || (mode < 0 && 2 * dp->nnz_covered < g->nnz)) dp = nullptr;
    }
    if (dp) {
        DenseUse use;
        const int rc = launch_dense_tiles(g, X, blocked ? -x_rows_pad : ldx, n_cols_x, want_sym, st, &use);
        if (rc) return rc;
        a.dpart = use.part;
        a.ldp = use.ldp;
        a.dslab0 = use.block_slab0;
        a.dnslab = use.block_nslab;
        a.rowptr = dp->r_rowptr;
        a.col = dp->r_col;
        a.col16 = T.ids16 ? dp->r_col16 : nullptr;
        a.has_huge = dp->r_max_row >= a.huge_len ? 1 : 0;
    }
    if (T.balance) {
        const int32_t* tr0 = dp ? dp->r_tile_row0 : g->tile_row0;
        if (tr0) {
            a.tile_row0 = tr0;
            a.n_tiles = dp ? dp->r_n_tiles : g->n_tiles;
            a.sym_map = dp ? dp->r_sym_map : g->sym_map;
            a.sym_blocks = dp ? dp->r_sym_blocks : g->sym_blocks;
        }
    }
    if (!vec_ok) {
        SR_REQUIRE(!blocked, "panel-blocked operands must be 16-byte aligned");
        return transpose_out ? launch_spmm<1, 32, kTrans, 32>(a, st)
                             : launch_spmm<1, 32, kPlain, 32>(a, st);
    }
    // the lean kernel: 32-float panels, 32-row tiles, row offsets (in 16-byte units) in 32 bits
    const bool lean_ok = T.lean && tile == 32 && ldx / 4 < (int64_t(1) << 24) &&
                         g->n_cols < (int64_t(1) << 24) && g->n_cols * (ldx / 4) < (int64_t(1) << 32) &&
                         g->n_rows < (int64_t(1) << 30);
    if (blocked) {
        SR_REQUIRE(lean_ok && (T.panel == 0 || T.panel == 32),
                   "panel-blocked operands need the lean kernel (tile 32, panel 32, < 2^24 rows)");
        if (want_sym) return launch_gather3<kSym>(a, st);
        return transpose_out ? launch_gather3<kTrans>(a, st) : launch_gather3<kPlain>(a, st);
    }
    if (lean_ok && want_sym) {
        a.tblock = g->n_rows;
        a.tstride = ldy;
        return launch_gather3<kSym>(a, st);
    }
    if (lean_ok && transpose_out && (T.panel == 0 || T.panel == 32))
        return launch_gather3<kTrans>(a, st);
    if (lean_ok && !transpose_out && (T.panel == 0 || T.panel == 32))
        return launch_gather3<kPlain>(a, st);
#define SR_TILE_SWITCH(LPR, TR)                                             \
    switch (tile) {                                                         \
        case 16: return launch_spmm<4, LPR, TR, 16>(a, st);                 \
        case 32: return launch_spmm<4, LPR, TR, 32>(a, st);                 \
        default: return launch_spmm<4, LPR, TR, 64>(a, st);                 \
    }
    if (want_sym) {
        // upper triangle + mirror: square 32 x 32 wave tiles
        a.tblock = g->n_rows;
        a.tstride = ldy;
        return launch_spmm<4, 8, kSym, 32>(a, st);
    }
    if (transpose_out) {
        if (panel > 64) panel = 64;  // wider panels would not leave LDS for the transpose tile
        if (panel == 64 && tile > 32) tile = 32;
        switch (panel) {
            case 16: SR_TILE_SWITCH(4, kTrans)
            case 32: SR_TILE_SWITCH(8, kTrans)
            default: SR_TILE_SWITCH(16, kTrans)
        }
    }
    switch (panel) {
        case 16: SR_TILE_SWITCH(4, kPlain)
        case 32: SR_TILE_SWITCH(8, kPlain)
        case 64: SR_TILE_SWITCH(16, kPlain)
        case 128: SR_TILE_SWITCH(32, kPlain)
        default: SR_TILE_SWITCH(64, kPlain)
    }
#undef SR_TILE_SWITCH
}

#ifdef SIMRANK_STAMPS
__attribute__((visibility("default"))) int simrank_read_stamps(unsigned long long* out16, int32_t reset) {
    SR_HIP(hipDeviceSynchronize());
    SR_HIP(hipMemcpyFromSymbol(out16, HIP_SYMBOL(simrank::g_stamps), 16 * sizeof(unsigned long long)));
    if (reset) {
        unsigned long long z[16] = {0};
        SR_HIP(hipMemcpyToSymbol(HIP_SYMBOL(simrank::g_stamps), z, sizeof(z)));
    }
    return SIMRANK_OK;
}
#endif

int simrank_fill_identity_blocked(float* S, int64_t n_rows, int64_t n_cols, int64_t rows_pad,
                                  int64_t col0, void* stream) {
    SR_REQUIRE(S && n_rows > 0 && n_cols > 0 && rows_pad >= n_rows, "bad identity block");
    const size_t bytes = size_t((n_cols + 31) / 32) * size_t(rows_pad) * 32 * sizeof(float);
    SR_HIP(hipMemsetAsync(S, 0, bytes, as_stream(stream)));
    const int grid = (int)std::min<int64_t>((n_cols + 255) / 256, 256 * 4);
    hipLaunchKernelGGL(set_diagonal_blocked_kernel, dim3(grid), dim3(256), 0, as_stream(stream), S, n_rows,
                       n_cols, rows_pad, col0);
    SR_HIP(hipGetLastError());
    return SIMRANK_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------
// Leg 1 of the FIRST update of every fit: S_0 = I (SimRank.py:124-126), so (W . S_0)^T = W^T — a zero fill and one
// value per entry instead of a leg of gathers (17.6 ms of a 4-update fit at N = 65536).  The same bits: the legs
// compute rowscale[a] * (1.0 + zeros) for an entry, rowscale[a] * 0 elsewhere.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void identity_leg1_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                            const float* __restrict__ rowscale, int64_t n_rows,
                                                            float* __restrict__ Tt, int64_t t_rows_pad) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (blockIdx.x * int64_t(blockDim.x) + threadIdx.x) >> 6;
    const int64_t nwaves = (int64_t(gridDim.x) * blockDim.x) >> 6;
    for (int64_t a = wave; a < n_rows; a += nwaves) {
        const float v = rowscale[a];
        float* base = Tt + ((a >> 5) * t_rows_pad) * 32 + (a & 31);         // column a of Tt: element (i, a) at base + 32 i
        for (int j = rowptr[a] + lane; j < rowptr[a + 1]; j += 64) base[int64_t(col[j]) * 32] = v;
    }
}

// the same for fp16-held matrices (half.hip: value x scale on 64-column panels): what its leg 1 stores for S_0 = I is the fp16
// nearest to scale * rowscale[a] (f32 sum of one stored 1.0 x scale, times the row scale, one rounding)
__global__ __launch_bounds__(256) void identity_leg1_h16_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                                const float* __restrict__ rowscale, int64_t n_rows,
                                                                uint16_t* __restrict__ Tt, int64_t t_rows_pad, float scale) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (blockIdx.x * int64_t(blockDim.x) + threadIdx.x) >> 6;
    const int64_t nwaves = (int64_t(gridDim.x) * blockDim.x) >> 6;
    for (int64_t a = wave; a < n_rows; a += nwaves) {
        const uint16_t v = __builtin_bit_cast(uint16_t, _Float16(scale * rowscale[a]));
        uint16_t* base = Tt + ((a >> 6) * t_rows_pad) * 64 + (a & 63);       // column a of Tt: element (i, a) at base + 64 i
        for (int j = rowptr[a] + lane; j < rowptr[a + 1]; j += 64) base[int64_t(col[j]) * 64] = v;
    }
}

namespace simrank {
int identity_leg1_blocked_h16(const simrank_graph* g, uint16_t* Tt, int64_t t_rows_pad, float scale, void* stream) {
    SR_REQUIRE(g && Tt && t_rows_pad >= g->n_cols, "bad identity product");
    const size_t bytes = size_t((g->n_rows + 63) / 64) * size_t(t_rows_pad) * 64 * sizeof(uint16_t);
    SR_HIP(hipMemsetAsync(Tt, 0, bytes, as_stream(stream)));
    const int grid = (int)std::min<int64_t>((g->n_rows + 3) / 4, 256 * 8);
    hipLaunchKernelGGL(identity_leg1_h16_kernel, dim3(grid), dim3(256), 0, as_stream(stream), g->rowptr, g->col, g->rowscale,
                       g->n_rows, Tt, t_rows_pad, scale);
    SR_HIP(hipGetLastError());
    return SIMRANK_OK;
}

int identity_leg1_blocked(const simrank_graph* g, float* Tt, int64_t t_rows_pad, void* stream) {
    SR_REQUIRE(g && Tt && t_rows_pad >= g->n_cols, "bad identity product");
    const size_t bytes = size_t((g->n_rows + 31) / 32) * size_t(t_rows_pad) * 32 * sizeof(float);
    SR_HIP(hipMemsetAsync(Tt, 0, bytes, as_stream(stream)));
    const int grid = (int)std::min<int64_t>((g->n_rows + 3) / 4, 256 * 8);
    hipLaunchKernelGGL(identity_leg1_kernel, dim3(grid), dim3(256), 0, as_stream(stream), g->rowptr, g->col, g->rowscale,
                       g->n_rows, Tt, t_rows_pad);
    SR_HIP(hipGetLastError());
    return SIMRANK_OK;
}
}  // namespace simrank

extern "C" {

int simrank_spmm(const simrank_graph* g, const float* X, int64_t ldx, int64_t n_cols_x,
                 float* Y, int64_t ldy, int32_t transpose_out, int64_t t_block, int64_t t_pad,
                 const simrank_epilogue* ep, void* stream) {
    SR_REQUIRE(n_cols_x > 0 && ldx >= n_cols_x, "X: %lld columns, ld %lld", (long long)n_cols_x,
               (long long)ldx);
    SR_REQUIRE(transpose_out || ldy >= n_cols_x, "Y: ld %lld < %lld columns", (long long)ldy,
               (long long)n_cols_x);
    return spmm_impl(g, X, ldx, n_cols_x, Y, ldy, transpose_out, t_block, t_pad, ep, stream, false, 0, 0);
}

int simrank_spmm_blocked(const simrank_graph* g, const float* X, int64_t x_rows_pad, int64_t n_cols_x,
                         float* Y, int64_t y_rows_pad, int32_t transpose_out,
                         const simrank_epilogue* ep, void* stream) {
    SR_REQUIRE(g && n_cols_x > 0, "bad arguments");
    SR_REQUIRE(x_rows_pad >= g->n_cols && y_rows_pad >= (transpose_out ? n_cols_x : g->n_rows),
               "padded row counts %lld / %lld too small", (long long)x_rows_pad, (long long)y_rows_pad);
    return spmm_impl(g, X, 32, n_cols_x, Y, 32, transpose_out, 0, 0, ep, stream, true, x_rows_pad, y_rows_pad);
}

static int epilogue_apply_impl(const float* Q, int64_t ldq, float* Y, int64_t ldy, int64_t n_rows,
                               int64_t n_cols, const simrank_epilogue* ep, void* stream, int64_t rows_pad) {
    SR_REQUIRE(Q && Y && ep, "NULL argument");
    SR_REQUIRE(n_rows > 0 && n_cols > 0 && (rows_pad ? rows_pad >= n_rows : (ldq >= n_cols && ldy >= n_cols)),
               "bad block shape");
    SpmmArgs a{};
    a.y_rows_pad = rows_pad;
    a.has_ep = 1;
    a.coef = ep->coef;
    a.lbd = ep->lbd;
    a.ev = ep->evidence;
    a.ld_ev = ep->ld_evidence;
    a.ap = ep->apriori;
    a.ld_ap = ep->ld_apriori;
    a.prev = ep->previous;
    a.ld_prev = ep->ld_previous;
    a.eps = ep->eps;
    a.n_changed = ep->n_changed;
    a.diag_col0 = ep->diag_col0;
    a.set_diag = ep->set_diag;
    SR_REQUIRE(!a.prev || a.n_changed, "previous needs a counter");
    hipStream_t st = as_stream(stream);
    if (a.prev)
        SR_HIP(hipMemsetAsync(a.n_changed, 0, sizeof(unsigned long long) * SIMRANK_CHANGED_SLOTS, st));
    const int grid = (int)std::min<int64_t>((n_rows * n_cols + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(epilogue_kernel, dim3(grid), dim3(256), 0, st, Q, ldq, Y, ldy, n_rows, n_cols, a);
    SR_HIP(hipGetLastError());
    return SIMRANK_OK;
}

static int spmm_shard_impl(const simrank_graph* g, const float* X, int64_t ldx, float* Y, int64_t ldy,
                           const simrank_epilogue* ep, int32_t rank, int32_t world, float* send,
                           int64_t chunk_floats, int32_t tile_lo, int32_t tile_hi, int32_t zero_counters,
                           void* stream) {
    SR_REQUIRE(g && X && Y && ep && send, "NULL argument");
    SR_REQUIRE(world >= 1 && rank >= 0 && rank < world && g->n_rows % (int64_t(world) * 32) == 0,
               "a symmetric sharded leg needs n_rows (%lld) divisible by 32 x world (%d)",
               (long long)g->n_rows, world);
    const int64_t mb = g->n_rows / world, tiles = mb / 32;
    SR_REQUIRE(tile_lo >= 0 && tile_lo < tile_hi && tile_hi <= tiles, "stage tiles [%d, %d) of %lld", tile_lo, tile_hi,
               (long long)tiles);
    const int64_t slot_lo = int64_t(tile_lo) * (tile_lo - 1) / 2, slot_hi = int64_t(tile_hi) * (tile_hi - 1) / 2;
    SR_REQUIRE(chunk_floats >= (slot_hi - slot_lo) * 1024 && (slot_hi == slot_lo || chunk_floats % 4 == 0),
               "send chunk too small or not a multiple of 4 floats");
    SR_REQUIRE(ep->diag_col0 == int64_t(rank) * mb && ldx >= mb && ldy >= mb && aligned16(X) && aligned16(Y) &&
                   aligned16(send) && ldx % 4 == 0 && ldy % 4 == 0, "bad shard operands");
    const Tuning& T = g->tun;
    SR_REQUIRE(T.lean && (T.tile == 0 || T.tile == 32) && (T.panel == 0 || T.panel == 32) &&
                   g->n_cols < (int64_t(1) << 24) && ldx / 4 < (int64_t(1) << 24) &&
                   g->n_cols * (ldx / 4) < (int64_t(1) << 32), "the symmetric sharded leg needs the lean kernel");
    SpmmArgs a{};
    a.rowptr = g->rowptr;
    a.col = g->col;
    a.col16 = T.ids16 ? g->col16 : nullptr;
    a.rowscale = g->rowscale;
    a.huge_len = (int32_t)std::max<int64_t>(kHeavy, T.huge);
    a.has_huge = g->max_row_nnz >= a.huge_len ? 1 : 0;
    a.X = X; a.ldx = ldx; a.L = mb; a.Y = Y; a.ldy = ldy;
    a.M = g->n_rows; a.K = g->n_cols;
    a.tblock = g->n_rows;
    a.xcd_map = (int)T.xcd_map;
    a.idx_mask = (int32_t)T.probe_mask;
    a.addr32 = (int32_t)T.addr32;
    a.probe = (int32_t)T.probe_flags;
    a.nt = 0;
    a.has_ep = 1;
    a.coef = ep->coef; a.lbd = ep->lbd;
    a.ev = ep->evidence; a.ld_ev = ep->ld_evidence;
    a.ap = ep->apriori; a.ld_ap = ep->ld_apriori;
    a.prev = ep->previous; a.ld_prev = ep->ld_previous;
    a.eps = ep->eps; a.n_changed = ep->n_changed;
    a.diag_col0 = ep->diag_col0; a.set_diag = ep->set_diag;
    a.restrict_support = ep->restrict_support;
    a.count_any = ep->count_any;
    SR_REQUIRE(!a.prev || a.n_changed, "previous needs a counter");
    SR_REQUIRE((!a.ev || (a.ld_ev % 4 == 0 && reinterpret_cast<uintptr_t>(a.ev) % 4 == 0)) &&
                   (!a.ap || (aligned16(a.ap) && a.ld_ap % 4 == 0)) &&
                   (!a.prev || (aligned16(a.prev) && a.ld_prev % 4 == 0)), "unaligned epilogue operand");
    a.sh_rank = rank; a.sh_mb = (int32_t)mb; a.sh_send = send; a.sh_chunk = chunk_floats;
    a.sh_tile0 = tile_lo; a.sh_ntiles = tile_hi - tile_lo; a.sh_slot0 = slot_lo;
    hipStream_t st = as_stream(stream);
    if (a.prev && zero_counters)
        SR_HIP(hipMemsetAsync(a.n_changed, 0, sizeof(unsigned long long) * SIMRANK_CHANGED_SLOTS, st));
    if (T.balance && g->tile_row0) {
        a.tile_row0 = g->tile_row0;
        a.n_tiles = g->n_tiles;
    }
    return launch_gather3<kShard>(a, st);
}

int simrank_spmm_shard(const simrank_graph* g, const float* X, int64_t ldx, float* Y, int64_t ldy,
                       const simrank_epilogue* ep, int32_t rank, int32_t world, float* send,
                       int64_t chunk_floats, void* stream) {
    SR_REQUIRE(g && world >= 1 && g->n_rows % (int64_t(world) * 32) == 0, "bad shard arguments");
    const int32_t tiles = (int32_t)(g->n_rows / world / 32);
    return spmm_shard_impl(g, X, ldx, Y, ldy, ep, rank, world, send, chunk_floats, 0, tiles, 1, stream);
}

int simrank_spmm_shard_stage(const simrank_graph* g, const float* X, int64_t ldx, float* Y, int64_t ldy,
                             const simrank_epilogue* ep, int32_t rank, int32_t world, float* send_stage,
                             int64_t stage_chunk_floats, int32_t tile_lo, int32_t tile_hi, int32_t zero_counters,
                             void* stream) {
    return spmm_shard_impl(g, X, ldx, Y, ldy, ep, rank, world, send_stage, stage_chunk_floats, tile_lo, tile_hi,
                           zero_counters, stream);
}

static int shard_unpack_impl(float* Y, int64_t ldy, const float* recv, int64_t chunk_floats, int32_t rank,
                             int32_t world, int64_t n_rows, int32_t tile_lo, int32_t tile_hi, void* stream) {
    SR_REQUIRE(Y && recv && world >= 1 && rank >= 0 && rank < world && n_rows % (int64_t(world) * 32) == 0,
               "bad shard unpack");
    const int mb = int(n_rows / world), tiles = mb / 32;
    SR_REQUIRE(tile_lo >= 0 && tile_lo < tile_hi && tile_hi <= tiles, "stage tiles [%d, %d) of %d", tile_lo, tile_hi, tiles);
    const int64_t slot_lo = int64_t(tile_lo) * (tile_lo - 1) / 2, slot_hi = int64_t(tile_hi) * (tile_hi - 1) / 2;
    const int64_t per_src = slot_hi - slot_lo;
    if (per_src == 0 || world == 1) return SIMRANK_OK;
    SR_REQUIRE(per_src < (int64_t(1) << 31), "too many tiles");
    SR_REQUIRE(ldy >= mb && ldy % 4 == 0 && aligned16(Y) && aligned16(recv) && chunk_floats % 4 == 0 &&
                   chunk_floats >= per_src * 1024, "bad shard unpack operands");
    hipLaunchKernelGGL(shard_unpack_kernel, dim3((unsigned)per_src, (unsigned)world), dim3(256), 0,
                       as_stream(stream), Y, ldy, recv, chunk_floats, (int)rank, mb, slot_lo);
    SR_HIP(hipGetLastError());
    return SIMRANK_OK;
}

int simrank_shard_unpack(float* Y, int64_t ldy, const float* recv, int64_t chunk_floats, int32_t rank,
                         int32_t world, int64_t n_rows, void* stream) {
    SR_REQUIRE(world >= 1 && n_rows % (int64_t(world) * 32) == 0, "bad shard unpack");
    return shard_unpack_impl(Y, ldy, recv, chunk_floats, rank, world, n_rows, 0, (int32_t)(n_rows / world / 32), stream);
}

int simrank_shard_unpack_stage(float* Y, int64_t ldy, const float* recv_stage, int64_t stage_chunk_floats,
                               int32_t rank, int32_t world, int64_t n_rows, int32_t tile_lo, int32_t tile_hi,
                               void* stream) {
    return shard_unpack_impl(Y, ldy, recv_stage, stage_chunk_floats, rank, world, n_rows, tile_lo, tile_hi, stream);
}

int simrank_epilogue_apply(const float* Q, int64_t ldq, float* Y, int64_t ldy, int64_t n_rows,
                           int64_t n_cols, const simrank_epilogue* ep, void* stream) {
    return epilogue_apply_impl(Q, ldq, Y, ldy, n_rows, n_cols, ep, stream, 0);
}

int simrank_epilogue_apply_blocked(const float* Q, float* Y, int64_t n_rows, int64_t n_cols,
                                   int64_t rows_pad, const simrank_epilogue* ep, void* stream) {
    return epilogue_apply_impl(Q, 32, Y, 32, n_rows, n_cols, ep, stream, rows_pad);
}

static int topk_impl(const float* S, int64_t ld, int64_t rows_pad, int64_t n_rows, int64_t n_cols,
                     int64_t col0, const int32_t* col_ids, int32_t k, int32_t exclude_diag,
                     int32_t* idx_out, float* val_out, void* stream) {
    SR_REQUIRE(S && idx_out && val_out, "NULL argument");
    SR_REQUIRE(n_rows > 0 && n_cols > 0 && (rows_pad ? rows_pad >= n_rows : ld >= n_cols) &&
                   n_cols < (int64_t(1) << 31) && k > 0 && k <= 1024, "bad top-k request");
    const int grid = (int)std::min<int64_t>((n_rows + 3) / 4, 256 * 8);
    // row-major rows longer than the L2 keeps: one pass with the k best per lane in registers
    if (!rows_pad && k <= 32 && n_cols >= 8192) {
        if (k <= 16)
            hipLaunchKernelGGL(topk_rows_onepass_kernel<16>, dim3(grid), dim3(256), 0, as_stream(stream), S, ld, n_rows,
                               n_cols, col0, k, exclude_diag, col_ids, idx_out, val_out);
        else
            hipLaunchKernelGGL(topk_rows_onepass_kernel<32>, dim3(grid), dim3(256), 0, as_stream(stream), S, ld, n_rows,
                               n_cols, col0, k, exclude_diag, col_ids, idx_out, val_out);
        SR_HIP(hipGetLastError());
        return SIMRANK_OK;
    }
    // panel-blocked, any width: one pass, eight rows per wave (aligned panels: every panel starts 16-byte aligned)
    if (rows_pad && k <= 32 && (reinterpret_cast<uintptr_t>(S) & 15) == 0) {
        const int grid8 = (int)std::min<int64_t>((n_rows + 31) / 32, 256 * 8);
        if (k <= 16)
            hipLaunchKernelGGL(topk_rows_blocked_onepass_kernel<16>, dim3(grid8), dim3(256), 0, as_stream(stream), S, rows_pad,
                               n_rows, n_cols, col0, k, exclude_diag, col_ids, idx_out, val_out);
        else
            hipLaunchKernelGGL(topk_rows_blocked_onepass_kernel<32>, dim3(grid8), dim3(256), 0, as_stream(stream), S, rows_pad,
                               n_rows, n_cols, col0, k, exclude_diag, col_ids, idx_out, val_out);
        SR_HIP(hipGetLastError());
        return SIMRANK_OK;
    }
    hipLaunchKernelGGL(topk_rows_kernel, dim3(grid), dim3(256), 0, as_stream(stream), S, ld, rows_pad,
                       n_rows, n_cols, col0, k, exclude_diag, col_ids, idx_out, val_out);
    SR_HIP(hipGetLastError());
    return SIMRANK_OK;
}

int simrank_topk_rows_ids(const float* S, int64_t ld, int64_t n_rows, int64_t n_cols, int64_t col0,
                          const int32_t* col_ids, int32_t k, int32_t exclude_diag,
                          int32_t* idx_out, float* val_out, void* stream) {
    return topk_impl(S, ld, 0, n_rows, n_cols, col0, col_ids, k, exclude_diag, idx_out, val_out, stream);
}

int simrank_topk_rows_blocked(const float* S, int64_t rows_pad, int64_t n_rows, int64_t n_cols,
                              int64_t col0, const int32_t* col_ids, int32_t k, int32_t exclude_diag,
                              int32_t* idx_out, float* val_out, void* stream) {
    return topk_impl(S, 32, rows_pad, n_rows, n_cols, col0, col_ids, k, exclude_diag, idx_out, val_out,
                     stream);
}

int simrank_topk_rows(const float* S, int64_t ld, int64_t n_rows, int64_t n_cols, int64_t col0,
                      int32_t k, int32_t exclude_diag, int32_t* idx_out, float* val_out,
                      void* stream) {
    return simrank_topk_rows_ids(S, ld, n_rows, n_cols, col0, nullptr, k, exclude_diag, idx_out,
                                 val_out, stream);
}

int simrank_permute_layout(const void* src, int64_t ld_src, int64_t src_rows_pad, void* dst,
                           int64_t ld_dst, int64_t dst_rows_pad, int64_t n_rows, int64_t n_cols,
                           const int32_t* row_idx, const int32_t* col_idx, int32_t elem_bytes,
                           void* stream) {
    SR_REQUIRE(src && dst && src != dst, "permute needs two distinct matrices");
    SR_REQUIRE(n_rows > 0 && n_cols > 0 && (dst_rows_pad ? dst_rows_pad >= n_rows : ld_dst >= n_cols) &&
                   (src_rows_pad > 0 || ld_src > 0), "bad block shape");
    SR_REQUIRE(elem_bytes == 1 || elem_bytes == 4, "elem_bytes must be 1 (u8) or 4 (f32)");
    const int grid = (int)std::min<int64_t>(n_rows, 256 * 16);
    if (elem_bytes == 4)
        hipLaunchKernelGGL(permute_kernel<float>, dim3(grid), dim3(256), 0, as_stream(stream),
                           (const float*)src, ld_src, src_rows_pad, (float*)dst, ld_dst, dst_rows_pad,
                           n_rows, n_cols, row_idx, col_idx);
    else
        hipLaunchKernelGGL(permute_kernel<uint8_t>, dim3(grid), dim3(256), 0, as_stream(stream),
                           (const uint8_t*)src, ld_src, src_rows_pad, (uint8_t*)dst, ld_dst, dst_rows_pad,
                           n_rows, n_cols, row_idx, col_idx);
    SR_HIP(hipGetLastError());
    return SIMRANK_OK;
}

int simrank_permute(const void* src, int64_t ld_src, void* dst, int64_t ld_dst, int64_t n_rows,
                    int64_t n_cols, const int32_t* row_idx, const int32_t* col_idx,
                    int32_t elem_bytes, void* stream) {
    return simrank_permute_layout(src, ld_src, 0, dst, ld_dst, 0, n_rows, n_cols, row_idx, col_idx,
                                  elem_bytes, stream);
}

static int evidence_counts_impl(const simrank_graph* g, int64_t col0, int64_t n_cols, uint8_t* counts,
                                int64_t ld, int64_t rows_pad, void* stream) {
    SR_REQUIRE(g && counts, "NULL argument");
    SR_REQUIRE(col0 >= 0 && n_cols > 0 && col0 + n_cols <= g->n_rows &&
                   (rows_pad ? rows_pad >= g->n_rows : ld >= n_cols),
               "evidence block [%lld, %lld) of %lld, ld %lld", (long long)col0,
               (long long)(col0 + n_cols), (long long)g->n_rows, (long long)ld);
    const int grid = (int)std::min<int64_t>(g->n_rows, 256 * 4);
    static const int regroup_rows = [] { const char* e = std::getenv("SIMRANK_EV_REGROUP"); return e && *e == '0' ? 0 : 1; }();
    // 4-byte stores need the block's first column, the pass boundaries (multiples of 65536) and the row
    // pitch to be multiples of 4 (the blocked layout: always, out_col0 being a multiple of 4)
    const int vec4 = (reinterpret_cast<uintptr_t>(counts) % 4 == 0) && (rows_pad ? true : ld % 4 == 0);
    // the whole square in one pass: upper triangle + mirror (tuning "ev_tri"; column blocks of sharded ranks and
    // blocks of more than 65536 columns take every path)
    const bool hubs_on = g->ev_hubs > 0 && g->ev_hubidx;
    const int tri = (g->tun.ev_tri && col0 == 0 && n_cols == g->n_rows && (hubs_on || n_cols <= kEvChunk)) ? 1 : 0;
    // (round 5) the pairs through the hub columns first, on the matrix cores: the 0/1 image of the live rows over those
    // columns (built once per graph, kept with it), then P . P^T in i8, saturated into `counts`; the LDS-counter kernel
    // below starts from what is there and leaves the hub columns out of its walk
    const int32_t* hubidx = nullptr;
    if (hubs_on) {
        simrank_graph* gm = const_cast<simrank_graph*>(g);
        const int Hp = g->ev_hubs;
        const int64_t Mp = (g->n_rows + 127) / 128 * 128;
        hipStream_t st = as_stream(stream);
        {
            // (the image belongs to the graph and is built by whoever asks first: under a lock, with an event behind the build
            // that every later call makes ITS stream wait for — two evidence calls on one graph from different streams or
            // threads used to race on the pointer, or read an image still being written; advisor, round 5)
            static std::mutex image_mutex;
            std::lock_guard<std::mutex> image_lock(image_mutex);
            if (!gm->ev_hub_image) {
                uint8_t* image = nullptr;
                hipEvent_t ready = nullptr;
                // (+ 128 rows: the last column tile of a block that does not start at a multiple of 128 reads past row Mp)
                SR_HIP(plan_alloc((void**)&image, size_t(Mp + 128) * size_t(Hp)));
                hipError_t e = hipMemsetAsync(image, 0, size_t(Mp + 128) * size_t(Hp), st);
                if (e == hipSuccess) {
                    hipLaunchKernelGGL(hub_image_kernel, dim3((unsigned)std::min<int64_t>((g->n_rows + 3) / 4, 4096)), dim3(256), 0,
                                       st, g->rowptr, g->col, g->rowscale, g->ev_hubidx, g->n_rows, Hp, image);
                    e = hipGetLastError();
                }
                if (e == hipSuccess) e = hipEventCreateWithFlags(&ready, hipEventDisableTiming);
                if (e == hipSuccess) e = hipEventRecord(ready, st);
                if (e != hipSuccess) {
                    (void)hipStreamSynchronize(st);
                    plan_free(image);
                    if (ready) (void)hipEventDestroy(ready);
                    SR_HIP(e);
                }
                gm->ev_hub_ready = ready;
                gm->ev_hub_image = image;
            } else if (gm->ev_hub_ready) {
                SR_HIP(hipStreamWaitEvent(st, (hipEvent_t)gm->ev_hub_ready, 0));
            }
        }
        const int tiles_i = int(Mp / 128), tiles_j = int((n_cols + 127) / 128);
        const int vec16 = (reinterpret_cast<uintptr_t>(counts) % 16 == 0) && (rows_pad ? true : ld % 16 == 0);
        hipLaunchKernelGGL(evidence_hub_kernel, dim3((unsigned)(tiles_i * tiles_j)), dim3(256), 0, st, gm->ev_hub_image, Hp,
                           g->n_rows, col0, n_cols, counts, ld, rows_pad, tiles_j, tri, vec16);
        SR_HIP(hipGetLastError());
        hubidx = g->ev_hubidx;
    }
    for (int64_t c = 0; c < n_cols; c += kEvChunk) {
        const int nc = (int)std::min<int64_t>(kEvChunk, n_cols - c);
        const size_t lds = size_t((nc + 1) / 2) * 4;
        if (lds > 64 * 1024)
            SR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(evidence_counts_kernel<1024>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(evidence_counts_kernel<1024>, dim3(grid), dim3(1024), lds, as_stream(stream), g->rowptr, g->col,
                           g->rowscale, g->t_rowptr, g->t_col, g->t_pos, g->n_rows, col0 + c, nc, counts, ld, rows_pad, c,
                           vec4, tri, hubidx, regroup_rows);
        SR_HIP(hipGetLastError());
    }
    if (tri) {
        const int64_t T = (g->n_rows + 63) / 64, tiles = T * (T + 1) / 2;
        SR_REQUIRE(tiles < (int64_t(1) << 31), "evidence mirror: %lld tiles", (long long)tiles);
        const int v16 = (reinterpret_cast<uintptr_t>(counts) % 16 == 0) && (rows_pad ? true : ld % 16 == 0);
        hipLaunchKernelGGL(evidence_mirror_kernel, dim3((unsigned)tiles), dim3(256), 0, as_stream(stream), counts,
                           g->n_rows, ld, rows_pad, v16);
        SR_HIP(hipGetLastError());
    }
    return SIMRANK_OK;
}

int simrank_evidence_counts(const simrank_graph* g, int64_t col0, int64_t n_cols,
                            uint8_t* counts, int64_t ld, void* stream) {
    return evidence_counts_impl(g, col0, n_cols, counts, ld, 0, stream);
}

int simrank_evidence_counts_blocked(const simrank_graph* g, int64_t col0, int64_t n_cols,
                                    uint8_t* counts, int64_t rows_pad, void* stream) {
    return evidence_counts_impl(g, col0, n_cols, counts, 32, rows_pad, stream);
}

}  // extern "C"
