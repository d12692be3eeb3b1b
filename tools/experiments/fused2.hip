// Leg 1 of the single-rank update as ONE PERSISTENT launch (gfx950, round 4):
//
//   Tt = (diag(rowscale) . A . X)^T        A = 0/1 CSR pattern, X and Tt panel-blocked      (SimRank.py:139, first .dot)
//
// Same arithmetic as fused.hip — per 128-row block the columns shared by >= fuse_min of its rows are multiplied
// on the matrix cores (0/1 pattern x the operand split into three bf16 terms = exact f32 products), the
// remainder is gathered from balanced per-lane-group id streams — restructured around what the timeline of
// fused.hip's workgroups showed (profiles/r04_s1.log):
//
//  * 18 % of a workgroup's life was its prologue (unit record -> row records / ids -> first gathers: three
//    dependent round trips through a saturated texture path) and a panel's ~130 workgroups were ONE round of
//    the XCD's 128 slots, so 7-12 panels' slices shared its 4 MiB L2 (hit rate 66 %).
//    Here 4 workgroups per CU stay resident and PULL work items from a per-XCD queue (one atomic per item, in
//    panel-major order; XCC id from the hardware register, so placement is never assumed), and everything an
//    item needs before its first gather is fetched while the item before it runs: the queue index two items
//    ahead, the record one item ahead, row records + first ids + first matrix-core ids with the record.
//  * The heaviest block ran 7.5 panel periods (its set holds 25 k of the 32 k columns) and tied its slot and
//    its panel's slice down for that long.  Here a block is cut into PIECES of bounded cost — each piece a
//    share of the set's columns (all 128 rows) plus the whole remainder of a share of the rows — whose raw
//    sums meet in memory: written through (sc1), a ticket per (block, panel), the LAST arriver adds them in
//    piece order (deterministic), scales and stores.  No workgroup ever waits for another.
//  * The matrix-core phase ran at 1 600 cycles per 16-column step and wave (384 of MFMA): two steps of operand
//    rows in flight against a ~3 000-cycle load latency.  Here the operand rows go global -> LDS directly
//    (buffer_load ... lds, 2 KiB per step, the layout the fragment reads want anyway) into a 4-slot ring per
//    wave that lives in the tile's memory: 4-5 steps in flight, no staging registers, counted vmcnt waits.
//  * Pattern bits -> A fragments through a NIBBLE table laid out [nibble][lane & 31] (8-byte entries): a lane
//    always reads its own two banks, no conflicts (the 256-entry table of fused.hip cost a third of the LDS
//    cycles in conflicts).
// Bitwise reproducible (fixed summation order everywhere; which workgroup handles which item does not matter).
#include <algorithm>
#include <cstring>
#include <numeric>
#include <type_traits>
#include <vector>

#include "common.h"
#include "fused_dev.h"

namespace simrank {

constexpr int kTS2 = 132;         // floats per column of the LDS tile (32 columns x 128 rows, transposed)
constexpr int kRec = 16;          // ints per item record
#ifndef SIMRANK_F2_MAXREM
#define SIMRANK_F2_MAXREM 256
#endif

struct Fused2Args {
    const float* X;
    float* Y;
    int64_t x_rows_pad, y_rows_pad;
    int64_t L, M;
    int32_t n_panels, n_items, nt;
    int32_t x_sentinel;
    int32_t idx_mask;         // DIAGNOSTIC (tuning "probe_mask")
    int32_t probe;            // DIAGNOSTIC (tuning "probe_flags"): 1 no gather phase, 2 no stores, 4 no MFMA phase
    int32_t cap_panels;       // panels the partial-sum slots were sized for
    const int32_t* items;     // [n_items + 1][kRec]: block, first quad, quads, row-record index, pieces of the block,
                              // partial slot (-1), ticket slot (-1), piece index, then per wave: first round of its id
                              // stream (8..11) and its rounds (12..15); the extra record is empty
    const float* rowscale;    // [M]
    float* partials;          // [slot][cap_panels][32 x 128] raw sums of the pieces of split blocks
    int32_t* tickets;         // [ticket slot][cap_panels] arrivals (the last arriver resets it)
    uint32_t* heads;          // [8][32] queue head per XCD (one 128-byte line each), zeroed before every launch
    const uint16_t* dcols16;
    const int32_t* dcols32;
    const uint4* abits;
    const int2* gmeta;        // [row-record index][wave][lane group][4 rows]: (end << 8 | row; 255: none, end 0xFFFFFF: no
                              // remainder), rowscale bits
    const uint16_t* sids16;
    const int32_t* sids32;
};

#ifndef SIMRANK_HOST_ONLY

#define F2_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define F2_COMPILER_FENCE() asm volatile("" ::: "memory")

typedef __attribute__((address_space(3))) void* f2_lds_ptr;

// a copy of v the compiler must treat as new: what is derived from it is recomputed where it is used instead of
// being kept in registers (or spilled) across the phases of an item
__device__ __forceinline__ int f2_opaque(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

#ifdef SIMRANK_F2_STAMPS
// Diagnostic build only (bash tools/build_variant.sh f2st -DSIMRANK_F2_STAMPS; tools/fused2_stamps.py): the timeline
// of the items wave 0 of every workgroup works through — s_memtime at the phase boundaries — into a __device__
// array nothing else reads (the first kF2Cap items to finish).
constexpr unsigned kF2Cap = 1u << 17;
__device__ unsigned long long g_f2st[size_t(kF2Cap) * 8];
__device__ unsigned int g_f2n;
__device__ __forceinline__ unsigned long long f2_now() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define F2ST(i) do { f2t[i] = f2_now(); } while (0)
#else
#define F2ST(i) do {} while (0)
#endif

#ifndef SIMRANK_F2_EARLY
#define SIMRANK_F2_EARLY 0     // 1: the first gather round of the next item is requested while this item's last one is
                               // summed (32 registers carried through the item boundary: spills at 128)
#endif
#ifndef SIMRANK_F2_LB
#define SIMRANK_F2_LB 4        // waves per SIMD the register allocation aims at (= resident workgroups per CU)
#endif
template <bool IDS16>
__global__ __launch_bounds__(256, SIMRANK_F2_LB) void fused2_kernel(const Fused2Args p) {
    // rings of the matrix-core phase (4 waves x 4 slots x 16 rows x 32 floats) | the block's result tile [column][row]
    __shared__ __attribute__((aligned(16))) float ring_tile[4 * 4 * 512];
    __shared__ __attribute__((aligned(16))) uint2 lut[16 * 32];              // [nibble][lane & 31] -> 4 bf16 (0 / 1.0)
    __shared__ __attribute__((aligned(16))) int2 gm_lds[4 * 8 * 4];          // per wave, lane group: four rows
    __shared__ int ids_lds[4 * 2 * 64];                                      // per wave: set columns of two quads
    __shared__ unsigned sh_k[4];
    __shared__ int sh_flag;
    float* const tile = ring_tile;

    const int lane0 = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned xcd = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 7u;     // HW_REG_XCC_ID
    uint32_t* const head = p.heads + size_t(xcd) * 32;
    const unsigned n_items = unsigned(p.n_items);
    const unsigned my_panels = unsigned(p.n_panels) > xcd ? (unsigned(p.n_panels) - xcd + 7u) / 8u : 0u;
    const unsigned total = my_panels * n_items;
    const int sent = p.x_sentinel;

    for (unsigned e = threadIdx.x; e < 512u; e += 256u) {
        const unsigned nib = e >> 5;
        uint2 v;
        v.x = ((nib >> 0) & 1u) * 0x3F80u | ((nib >> 1) & 1u) * 0x3F800000u;
        v.y = ((nib >> 2) & 1u) * 0x3F80u | ((nib >> 3) & 1u) * 0x3F800000u;
        lut[e] = v;
    }
    if (threadIdx.x == 0) {
        for (int i = 0; i < 3; ++i) sh_k[i] = __hip_atomic_fetch_add(head, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();

    struct Rec { int b0, quad0, nq, gmi, np, pslot, cslot, kidx, round0, nr, panel; bool valid; };
    auto load_rec = [&](unsigned k) -> Rec {
        Rec r;
        k = __builtin_amdgcn_readfirstlane(k);
        const bool ok = k < total;
        const unsigned pi = ok ? k / n_items : 0u;
        const unsigned u = ok ? k - pi * n_items : n_items;            // (the record behind the last one is empty)
        const int32_t* it = p.items + size_t(u) * kRec;
        r.b0 = it[0]; r.quad0 = it[1]; r.nq = (p.probe & 4) ? 0 : it[2]; r.gmi = it[3]; r.np = it[4];
        r.pslot = it[5]; r.cslot = it[6]; r.kidx = it[7];
        r.round0 = it[8 + wave];
        r.nr = (p.probe & 1) ? 0 : it[12 + wave];
        r.panel = int(xcd + 8u * pi);
        r.valid = ok;
        return r;
    };
    // round rr of the wave's id stream: this lane's id (uniform base, 32-bit lane offset)
    auto ld_sid = [&](const Rec& r, int rr, int ln) -> int {
        const size_t rnd = size_t(r.round0) + size_t(min(rr, max(r.nr - 1, 0)));
        if constexpr (IDS16) {
            const uint16_t* base = p.sids16 + rnd * 64;
            const int v = int(base[unsigned(ln)]);
            return v == 0xFFFF ? sent : (v & p.idx_mask);
        } else {
            const int32_t* base = p.sids32 + rnd * 64;
            const int v = base[unsigned(ln)];
            return v < 0 ? sent : (v & p.idx_mask);
        }
    };
    auto ld_ids = [&](int quad, int ln) -> int {
        if constexpr (IDS16) {
            const uint16_t* base = p.dcols16 + size_t(quad) * 64;
            return int(base[unsigned(ln)]) & p.idx_mask;
        } else {
            const int32_t* base = p.dcols32 + size_t(quad) * 64;
            return base[unsigned(ln)] & p.idx_mask;
        }
    };
    // what an item needs before its first gather / first matrix-core step, fetched one item ahead
    struct Pre { int2 gm; int iv0, iv1, ida, idb; };
    auto issue_pre = [&](const Rec& r) -> Pre {
        const int ln = f2_opaque(lane0);
        Pre s;
        const int2* gbase_ = p.gmeta + (size_t(r.gmi) * 4 + wave) * 32;
        s.gm = gbase_[unsigned((ln >> 3) * 4 + (ln & 3))];
        s.iv0 = ld_sid(r, 0, ln);
        s.iv1 = ld_sid(r, 1, ln);
        const int per = (r.nq + 3) >> 2;
        const int last = max(r.nq - 1, 0);
        s.ida = ld_ids(r.quad0 + min(wave * per, last), ln);
        s.idb = ld_ids(r.quad0 + min(wave * per + 1, last), ln);
        return s;
    };

    Rec cur = load_rec(sh_k[0]);
    Rec nx1 = load_rec(sh_k[1]);
    Pre pc = issue_pre(cur);
    unsigned u = 0;
    // the gather stream runs THROUGH the items: when the next item has no matrix-core phase its first round is
    // requested while this item's last one is summed (pre = 1: it is in vA when that item's turn comes)
    float4 vA[8];
    int pre = 0;

    while (cur.valid) {
#ifdef SIMRANK_F2_STAMPS
        unsigned long long f2t[6];
#endif
        F2ST(0);
        // ---- prefetch for the items behind this one (consumed at the top of the next iteration)
        const Pre pn = issue_pre(nx1);
        const Rec nx2 = load_rec(sh_k[(u + 2) & 3]);
        unsigned kf = 0;
        if (threadIdx.x == 0) kf = __hip_atomic_fetch_add(head, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

        const int panel = cur.panel;
        const int64_t c0 = int64_t(panel) * 32;
        const float* xbase = p.X + (int64_t(panel) * p.x_rows_pad) * 32;
        const __amdgpu_buffer_rsrc_t srd =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xbase), 0, int(p.x_rows_pad * 128), 0x00020000);
        const int nq = cur.nq;
        const bool split = cur.np > 1;
        const bool has_d = nq > 0 || split;                     // the tile holds something before the rows are emitted
        const int n_rounds = cur.nr;
        {
            const int ln = f2_opaque(lane0);
            gm_lds[wave * 32 + (ln >> 3) * 4 + (ln & 3)] = pc.gm;       // (lanes q and q + 4 write the same value)
        }
        const int per = (nq + 3) >> 2;                          // quads per wave
        const int n_active = per ? (nq + per - 1) / per : 0;    // waves that have matrix-core work

        // ---------------------------------------------------------------- 3. gather phase (the remainder)
        // (FRESH: nothing of this item was requested ahead — always so behind a matrix-core phase; a separate
        // instantiation, so that the compiler sees vA / vB dead across that phase)
        auto gather_phase = [&](auto fresh_tag) {
            constexpr bool FRESH = decltype(fresh_tag)::value;
            __builtin_amdgcn_sched_barrier(0);                   // (nothing of this phase moves up into the one before it)
            F2_COMPILER_FENCE();
            const int ln = f2_opaque(lane0);
            const int g = ln >> 3, q = ln & 7, gbase = ln & ~7;
            const uint32_t qoff = uint32_t(q) * 16u;
            const int2* gmp = gm_lds + (wave * 8 + g) * 4;
            wave_lds_order();
            auto unpack = [](const int2& m) -> int3 {
                const unsigned v = unsigned(m.x);
                const int row = int(v & 255u), end = int(v >> 8);
                return make_int3(row == 255 ? -1 : row, m.y, end == 0xFFFFFF ? -1 : end);
            };
            int3 m_cur = unpack(gmp[0]);
            float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
            int krow = 0;
            auto emit = [&](const int3& m, const float4& sv) {
                if (m.x >= 0) {
                    const float sc = split ? 1.0f : __int_as_float(m.y);
                    float* tp = tile + (4 * q) * kTS2 + m.x;
                    const float d0 = has_d ? tp[0] : 0.f, d1 = has_d ? tp[kTS2] : 0.f;
                    const float d2 = has_d ? tp[2 * kTS2] : 0.f, d3 = has_d ? tp[3 * kTS2] : 0.f;
                    tp[0] = (sv.x + d0) * sc;
                    tp[kTS2] = (sv.y + d1) * sc;
                    tp[2 * kTS2] = (sv.z + d2) * sc;
                    tp[3 * kTS2] = (sv.w + d3) * sc;
                }
            };
            auto row_end = [&](int f) {
                if (f + 1 == m_cur.z) {
                    emit(m_cur, sum);
                    sum = make_float4(0.f, 0.f, 0.f, 0.f);
                    ++krow;
                    m_cur = unpack(gmp[min(krow, 3)]);
                    if (krow > 3) m_cur.z = -1;
                }
            };
            auto issue8 = [&](int iv, float4 (&v)[8]) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = ld_seg(srd, __shfl(iv, gbase + j), qoff);
            };
            auto consume = [&](const float4 (&v)[8], int r) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    sum.x += v[j].x; sum.y += v[j].y; sum.z += v[j].z; sum.w += v[j].w;
                    row_end(8 * r + j);
                }
            };
            // (every wave's stream has an even number of rounds: rounds r and r + 1 are in flight at the top of the loop)
            const bool early = SIMRANK_F2_EARLY && nx1.valid && nx1.nq == 0 && nx1.nr > 0;
            const float* xbase_n = p.X + (int64_t(nx1.panel) * p.x_rows_pad) * 32;
            const __amdgpu_buffer_rsrc_t srd_n =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xbase_n), 0, int(p.x_rows_pad * 128), 0x00020000);
            auto issue8n = [&](int iv, float4 (&v)[8]) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = ld_seg(srd_n, __shfl(iv, gbase + j), qoff);
            };
            float4 vB[8];
            if (n_rounds > 0) {
                // ids are requested two rounds before their gathers, ahead of the gathers issued in between: a
                // wait for ids then never waits for those gathers (vmcnt counts in order)
                int iv2 = ld_sid(cur, 2, ln), iv3 = ld_sid(cur, 3, ln);
                if (FRESH || !pre) issue8(pc.iv0, vA);
                issue8(pc.iv1, vB);
                int r = 0;
                while (r + 2 < n_rounds) {
                    consume(vA, r);
                    const int iv4 = ld_sid(cur, r + 4, ln);
                    issue8(iv2, vA);
                    consume(vB, r + 1);
                    const int iv5 = ld_sid(cur, r + 5, ln);
                    issue8(iv3, vB);
                    iv2 = iv4;
                    iv3 = iv5;
                    r += 2;
                }
                consume(vA, r);
                if (early) issue8n(pn.iv0, vA);
                consume(vB, r + 1);
            } else if (early) {
                issue8n(pn.iv0, vA);
            }
            pre = early ? 1 : 0;
            for (int k = krow; k < 4; ++k) emit(unpack(gmp[k]), make_float4(0.f, 0.f, 0.f, 0.f));    // rows without a remainder
        };

        F2ST(1);
        // ---------------------------------------------------------------- 1. matrix-core phase
        if (nq > 0) {
            f32x16 acc[4];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
            const int q_lo = wave * per, q_hi = min(nq, q_lo + per);
            if (q_lo < q_hi) {
                const int ln = f2_opaque(lane0);
                const int g = ln >> 3;
                const uint32_t qoff = uint32_t(ln & 7) * 16u;
                float* const ring = ring_tile + wave * (4 * 512);
                int* const idr = ids_lds + wave * 128;            // ids of two quads: quad k of the wave at (k & 1) * 64
                const int T = 4 * (q_hi - q_lo);                 // steps of this wave
                // step j of the wave -> ring slot j & 3: 16 operand rows x 128 bytes, two wave instructions of 8 rows
                // (row ids from the wave's id ring: quad j >> 2, positions (j & 3) * 16 + g and + 8)
                auto dma = [&](int j) {
                    float* slot = ring + (j & 3) * 512;
                    const int* ip = idr + ((j >> 2) & 1) * 64 + (j & 3) * 16 + g;
                    const int r0 = ip[0], r1 = ip[8];
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (f2_lds_ptr)slot, 16, int(__umul24(uint32_t(r0), 128u) + qoff), 0, 0, 0);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (f2_lds_ptr)(slot + 256), 16, int(__umul24(uint32_t(r1), 128u) + qoff), 0, 0, 0);
                };
                struct Terms { uint32_t lo[4], mid[4], hi[4]; };
                // B fragment from the slot: [k][n] -> lane (n, h) holds k = 8h .. 8h + 7
                auto fetch = [&](int j, float (&x)[8]) {
                    const float* slot = ring + (j & 3) * 512 + (ln >> 5) * 256 + (ln & 31);
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) x[jj] = slot[jj * 32];
                };
                auto split3 = [&](const float (&x)[8], Terms& t) {
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) split3f(x[2 * jj], x[2 * jj + 1], t.hi[jj], t.mid[jj], t.lo[jj]);
                };
                const uint2* lutl = lut + (ln & 31);
                auto mma = [&](const Terms& t, uint32_t aw) {
                    const bf16x8 bl = frag(t.lo[0], t.lo[1], t.lo[2], t.lo[3]);
                    const bf16x8 bm = frag(t.mid[0], t.mid[1], t.mid[2], t.mid[3]);
                    const bf16x8 bh = frag(t.hi[0], t.hi[1], t.hi[2], t.hi[3]);
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) {
                        const uint32_t b8 = (aw >> (8 * tt)) & 255u;
                        const uint2 l0 = lutl[(b8 & 15u) * 32], l1 = lutl[(b8 >> 4) * 32];
                        const bf16x8 a = frag(l0.x, l0.y, l1.x, l1.y);
                        acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bl, acc[tt], 0, 0, 0);
                        acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bm, acc[tt], 0, 0, 0);
                        acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bh, acc[tt], 0, 0, 0);
                    }
                };
                // the ids of quads 0 and 1 came with the item; the ids of quad k + 2 are loaded while quad k - 1 runs and
                // take quad k's place (all of whose steps have been requested by then) before the last step of quad k
                idr[ln] = pc.ida;
                idr[64 + ln] = pc.idb;
                int ids_fly = ld_ids(cur.quad0 + min(q_lo + 2, q_hi - 1), ln);
                const uint4* abase = p.abits + size_t(cur.quad0 + q_lo) * 64;
                uint4 aw_a = abase[unsigned(ln)];
                uint4 aw_b = abase[unsigned(min(1, q_hi - 1 - q_lo) * 64 + ln)];
                wave_lds_order();
                F2_COMPILER_FENCE();
                dma(0); dma(1); dma(2); dma(3);
                Terms tt;
                float x[8];
                F2_WAIT_VM(6);                                   // step 0 has landed (three steps younger)
                fetch(0, x);
                split3(x, tt);
                F2_COMPILER_FENCE();
                if (4 < T) dma(4);
                // iteration i: step i + 1 is read from its slot, the MFMAs of step i are issued, step i + 1 is split
                // into the same term registers (the matrix pipe has taken them by then; at four or five steps of
                // load latency per step the phase is not MFMA-bound, and a second set of terms spilled), step i + 5
                // is requested into the slot step i + 1 came from
                auto body = [&](int i, uint32_t aw) {
                    if (i + 1 < T) {
                        // DMA requests younger than step i + 1: steps i + 2 .. min(i + 4, T - 1)
                        const int younger = min(i + 4, T - 1) - (i + 1);
                        if (younger >= 3) F2_WAIT_VM(6);
                        else if (younger == 2) F2_WAIT_VM(4);
                        else if (younger == 1) F2_WAIT_VM(2);
                        else F2_WAIT_VM(0);
                        fetch(i + 1, x);
                    }
                    mma(tt, aw);
                    if (i + 1 < T) split3(x, tt);
                    F2_COMPILER_FENCE();
                    if (i + 5 < T) dma(i + 5);
                };
                for (int i = 0; i < T; i += 4) {
                    const int qk = i >> 2;                       // quad of the wave these four steps belong to
                    body(i, aw_a.x);
                    body(i + 1, aw_a.y);
                    body(i + 2, aw_a.z);
                    // (the last body of the quad requests step 0 of quad qk + 2: its ids enter the ring here)
                    idr[(qk & 1) * 64 + ln] = ids_fly;
                    wave_lds_order();
                    body(i + 3, aw_a.w);
                    aw_a = aw_b;
                    ids_fly = ld_ids(cur.quad0 + min(q_lo + qk + 3, q_hi - 1), ln);
                    aw_b = abase[unsigned(min(qk + 2, q_hi - 1 - q_lo) * 64 + ln)];
                }
                F2_WAIT_VM(0);                                   // (nothing of this wave may still land in the tile's memory)
            }
            // ------------------------------------------------------------ 2. sum of the waves, in wave order
            // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
            __syncthreads();                                     // every wave is done with its ring
            {
                const int ln = f2_opaque(lane0);
                const int n = ln & 31, h = ln >> 5;
                for (int w = 0; w < n_active; ++w) {
                    if (wave == w) {
#pragma unroll
                        for (int t = 0; t < 4; ++t)
#pragma unroll
                            for (int i4 = 0; i4 < 4; ++i4) {
                                float4* dst = reinterpret_cast<float4*>(tile + n * kTS2 + 32 * t + 8 * i4 + 4 * h);
                                float4 v = make_float4(acc[t][4 * i4], acc[t][4 * i4 + 1], acc[t][4 * i4 + 2], acc[t][4 * i4 + 3]);
                                if (w > 0) {
                                    const float4 o = *dst;
                                    v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                                }
                                *dst = v;
                            }
                    }
                    __syncthreads();
                }
            }
            F2ST(2);
            gather_phase(std::true_type{});
        } else {
            F2ST(2);
            if (split) {
                // a piece without columns of the set: the rows other pieces own stay zero in its partial tile
                for (int x4 = threadIdx.x; x4 < 32 * kTS2 / 4; x4 += 256)
                    reinterpret_cast<float4*>(tile)[x4] = make_float4(0.f, 0.f, 0.f, 0.f);
                __syncthreads();
            }
            gather_phase(std::false_type{});
        }
        F2ST(3);
        __syncthreads();
        F2ST(4);

        // ---------------------------------------------------------------- 4. pieces of a split block meet in memory
        bool store_it = true;
        const int row0 = cur.b0 * kFB;
        if (split) {
            typedef unsigned v4u __attribute__((ext_vector_type(4)));
            const int tid = f2_opaque(int(threadIdx.x));
            float* mine = p.partials + (size_t(cur.pslot) * p.cap_panels + panel) * (32 * kFB);
            const __amdgpu_buffer_rsrc_t msrd = __builtin_amdgcn_make_buffer_rsrc(mine, 0, 32 * kFB * 4, 0x00020000);
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int x = tid + it * 256;                   // 1024 float4 = 32 columns x 128 rows
                const int c = x >> 5, r4 = (x & 31) * 4;
                const float4 v = *reinterpret_cast<const float4*>(tile + c * kTS2 + r4);
                v4u o;
                o.x = __float_as_uint(v.x); o.y = __float_as_uint(v.y); o.z = __float_as_uint(v.z); o.w = __float_as_uint(v.w);
                __builtin_amdgcn_raw_buffer_store_b128(o, msrd, (c * kFB + r4) * 4, 0, 16);      // sc1: written through
            }
            F2_WAIT_VM(0);
            __syncthreads();
            if (threadIdx.x == 0) {
                int* tk = p.tickets + size_t(cur.cslot) * p.cap_panels + panel;
                const int t = __hip_atomic_fetch_add(tk, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int last = t == cur.np - 1;
                if (last) __hip_atomic_store(tk, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
                sh_flag = last;
            }
            __syncthreads();
            store_it = sh_flag != 0;
            if (store_it) {
                const int first = cur.pslot - cur.kidx;          // the block's pieces own consecutive slots
                for (int it = 0; it < 4; ++it) {
                    const int x = tid + it * 256;
                    const int c = x >> 5, r4 = (x & 31) * 4;
                    float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f);
                    for (int k = 0; k < cur.np; ++k) {
                        const float* src = p.partials + (size_t(first + k) * p.cap_panels + panel) * (32 * kFB);
                        const __amdgpu_buffer_rsrc_t ssrd =
                            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 32 * kFB * 4, 0x00020000);
                        const v4u w = __builtin_amdgcn_raw_buffer_load_b128(ssrd, (c * kFB + r4) * 4, 0, 16);   // sc1: past the L1
                        const float4 v = make_float4(__uint_as_float(w.x), __uint_as_float(w.y), __uint_as_float(w.z), __uint_as_float(w.w));
                        if (k == 0) a4 = v;
                        else { a4.x += v.x; a4.y += v.y; a4.z += v.z; a4.w += v.w; }
                    }
                    float sc[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) sc[j] = row0 + r4 + j < p.M ? p.rowscale[row0 + r4 + j] : 0.f;
                    a4.x *= sc[0]; a4.y *= sc[1]; a4.z *= sc[2]; a4.w *= sc[3];
                    *reinterpret_cast<float4*>(tile + c * kTS2 + r4) = a4;
                }
            }
            __syncthreads();
        }

        // ---------------------------------------------------------------- 5. transposed store
        if (store_it) {
            const int ln = f2_opaque(lane0);
            const int nrows = int(min(int64_t(kFB), p.M - row0));
            const int rows_out = max(0, min(32, nrows - 32 * wave));
            const int cols_here = int(min(int64_t(32), p.L - c0));
            if (rows_out > 0 && !(p.probe & 2)) {
                // panel-blocked Tt: the wave's 32 x 32 tile is 4 KiB contiguous, element (c, r) at c * 32 + r
                float* base = p.Y + ((int64_t(row0 >> 5) + wave) * p.y_rows_pad + c0) * 32;
                const float* tw = tile + 32 * wave;
                const __amdgpu_buffer_rsrc_t ysrd = __builtin_amdgcn_make_buffer_rsrc(base, 0, 4096, 0x00020000);
                if ((rows_out & 3) == 0) {
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int x = ln + it * 64;
                        const int c = x >> 3;
                        const int r4 = (x & 7) * 4;
                        if (c < cols_here && r4 < rows_out) {
                            const float4 v = *reinterpret_cast<const float4*>(tw + c * kTS2 + r4);
                            typedef unsigned v4u __attribute__((ext_vector_type(4)));
                            v4u o;
                            o.x = __float_as_uint(v.x); o.y = __float_as_uint(v.y);
                            o.z = __float_as_uint(v.z); o.w = __float_as_uint(v.w);
                            const int off = (c * 32 + r4) * 4;
                            if (p.nt) __builtin_amdgcn_raw_buffer_store_b128(o, ysrd, off, 0, 2);
                            else __builtin_amdgcn_raw_buffer_store_b128(o, ysrd, off, 0, 0);
                        }
                    }
                } else {
                    for (int x = ln; x < 32 * 32; x += 64) {
                        const int c = x >> 5, r = x & 31;
                        if (c < cols_here && r < rows_out) base[c * 32 + r] = tw[c * kTS2 + r];
                    }
                }
            }
        }

        // ---------------------------------------------------------------- next item
        if (threadIdx.x == 0) sh_k[(u + 3) & 3] = kf;
        __syncthreads();                                         // the tile, the row records and sh_k change hands
#ifdef SIMRANK_F2_STAMPS
        F2ST(5);
        if (threadIdx.x == 0) {
            const unsigned at = atomicAdd(&g_f2n, 1u);
            if (at < kF2Cap) {
                unsigned long long* o = g_f2st + size_t(at) * 8;
                for (int i = 0; i < 6; ++i) o[i] = f2t[i];
                o[6] = (unsigned long long)xcd << 56 | (unsigned long long)unsigned(panel) << 32 | unsigned(cur.b0) << 8 | unsigned(cur.kidx);
                o[7] = (unsigned long long)unsigned(nq) << 32 | unsigned(n_rounds) << 8 | unsigned(cur.np);
            }
        }
#endif
        cur = nx1;
        nx1 = nx2;
        pc = pn;
        ++u;
    }
}

#endif  // SIMRANK_HOST_ONLY

template <typename T>
static int upload_vec2(T** d, const std::vector<T>& h) {
    const size_t bytes = std::max<size_t>(16, h.size() * sizeof(T));
    SR_HIP(plan_alloc((void**)d, bytes));
    if (!h.empty()) SR_HIP(plan_upload(*d, h.data(), h.size() * sizeof(T)));
    return SIMRANK_OK;
}

void free_fused2_plan(simrank_fused2_plan* p) {
    if (!p) return;
    plan_free(p->items); plan_free(p->dcols16); plan_free(p->dcols32); plan_free(p->abits);
    plan_free(p->gmeta); plan_free(p->sids16); plan_free(p->sids32);
#ifndef SIMRANK_HOST_ONLY
    (void)pool_free(p->partials);
    (void)hipFree(p->tickets);
    (void)hipFree(p->heads);
#endif
    delete p;
}

// Host side.  Per 128-row block: the dense set (as fused.hip: columns referenced by >= fuse_min rows, plus
// every column of a row whose remainder would exceed SIMRANK_F2_MAXREM; dropped when it makes fewer than
// fuse_steps steps), its pattern bits in A-fragment order, and the block's PIECES: the estimated duration of
// the block (matrix-core steps x c_step + gather rounds x c_round, in cycles of one workgroup) is cut into
// pieces of at most fuse_cap cycles; piece k of n owns quads [nq k / n, nq (k + 1) / n) of the set and every
// n-th row of the block in descending remainder order; its rows are dealt to 32 lane groups (four rows each
// at most) so that the groups' totals balance, and laid out as id streams of 64 per round exactly as fused.hip.
int build_fused2_plan(simrank_graph* g, const int32_t* rowptr, const int32_t* col, const float* rowscale) {
    (void)rowscale;
    const int64_t M = g->n_rows, K = g->n_cols;
    const int64_t thr = std::max<int64_t>(2, g->tun.fuse_min);
    const int64_t min_steps = fuse_min_steps(g->tun, g->n_cols);
    const int64_t nblk = (M + kFB - 1) / kFB;
    const bool ids16 = K < 65535;
    const double c_step = 200.0;          // cycles per 16-column step of a workgroup (4 waves side by side)
    const double c_round = 550.0;         // cycles per round (256 gathered entries of the workgroup)
    const double cap = (double)std::max<int64_t>(1000, g->tun.fuse_cap);
    std::vector<int32_t> dcols;
    std::vector<uint32_t> abits;
    std::vector<int32_t> sids, gmeta, items;
    struct Item { int32_t b0, quad0, nq, gmi, np, k; int32_t round0[4], nr[4]; double cost; };
    std::vector<Item> list;
    std::vector<uint16_t> cnt(size_t(K), 0);
    std::vector<int32_t> kpos(size_t(K), -1), touched, set;
    std::vector<int32_t> rem[kFB];
    int64_t covered = 0, steps_total = 0, r_nnz = 0, n_quads = 0;
    for (int64_t b = 0; b < nblk; ++b) {
        const int64_t lo = b * kFB, hi = std::min<int64_t>(M, lo + kFB);
        touched.clear();
        set.clear();
        for (int32_t j = rowptr[lo]; j < rowptr[hi]; ++j)
            if (cnt[col[j]]++ == 0) touched.push_back(col[j]);
        for (int32_t c : touched)
            if (cnt[c] >= thr) { set.push_back(c); kpos[c] = 0; }
        if ((int64_t)(set.size() + 15) / 16 < min_steps) {
            for (int32_t c : set) kpos[c] = -1;
            set.clear();
        }
        for (int64_t a = lo; a < hi; ++a) {
            int32_t r = 0;
            for (int32_t j = rowptr[a]; j < rowptr[a + 1]; ++j) r += kpos[col[j]] < 0;
            if (r > SIMRANK_F2_MAXREM)
                for (int32_t j = rowptr[a]; j < rowptr[a + 1]; ++j)
                    if (kpos[col[j]] < 0) { set.push_back(col[j]); kpos[col[j]] = 0; }
        }
        std::sort(set.begin(), set.end());
        const int32_t U = (int32_t)set.size();
        const int32_t nq = (U + 63) / 64;
        steps_total += (U + 15) / 16;
        const size_t q0 = size_t(n_quads);
        n_quads += nq;
        dcols.resize((q0 + size_t(nq)) * 64, U ? set[0] : 0);      // padding: a real row, pattern bits zero
        abits.resize((q0 + size_t(nq)) * 64 * 4, 0u);
        for (int32_t i = 0; i < U; ++i) {
            dcols[q0 * 64 + size_t(i)] = set[size_t(i)];
            kpos[set[size_t(i)]] = i;
        }
        const int nr = int(hi - lo);
        int64_t rem_total = 0;
        for (int rr = 0; rr < nr; ++rr) {
            const int64_t a = lo + rr;
            rem[rr].clear();
            for (int32_t j = rowptr[a]; j < rowptr[a + 1]; ++j) {
                const int32_t i = kpos[col[j]];
                if (i < 0) {
                    rem[rr].push_back(col[j]);
                } else {
                    const int kk = i & 15, s = (i & 63) >> 4;
                    const int ln = (kk >> 3) * 32 + (rr & 31);
                    abits[((q0 + size_t(i >> 6)) * 64 + size_t(ln)) * 4 + size_t(s)] |= 1u << (8 * (rr >> 5) + (kk & 7));
                    ++covered;
                }
            }
            rem_total += (int64_t)rem[rr].size();
        }
        r_nnz += rem_total;
        // pieces
        const double dur = double(nq) * 4 * c_step + double(rem_total) / 256.0 * c_round;
        int np = (int)std::min<double>(std::min<double>(64.0, std::max(1, nr)), std::max(1.0, std::ceil(dur / cap)));
        int order[kFB];
        std::iota(order, order + nr, 0);
        std::stable_sort(order, order + nr, [&](int x, int y) { return rem[x].size() > rem[y].size(); });
        for (int k = 0; k < np; ++k) {
            Item it{};
            it.b0 = (int32_t)b;
            it.np = np;
            it.k = k;
            it.quad0 = (int32_t)(q0 + size_t(int64_t(nq) * k / np));
            it.nq = (int32_t)(int64_t(nq) * (k + 1) / np - int64_t(nq) * k / np);
            it.gmi = (int32_t)(gmeta.size() / (32 * 4 * 2));
            // this piece's rows: every np-th of the descending order
            int prow[kFB], pn = 0;
            for (int i = k; i < nr; i += np) prow[pn++] = order[i];
            int grp_rows[32][4], grp_n[32];
            int64_t grp_tot[32];
            for (int i = 0; i < 32; ++i) { grp_n[i] = 0; grp_tot[i] = 0; }
            for (int i = 0; i < pn; ++i) {
                int best = -1;
                for (int gi = 0; gi < 32; ++gi)
                    if (grp_n[gi] < 4 && (best < 0 || grp_tot[gi] < grp_tot[best])) best = gi;
                grp_rows[best][grp_n[best]++] = prow[i];
                grp_tot[best] += (int64_t)rem[prow[i]].size();
            }
            int gorder[32];
            std::iota(gorder, gorder + 32, 0);
            std::stable_sort(gorder, gorder + 32, [&](int x, int y) { return grp_tot[x] > grp_tot[y]; });
            const size_t gm0 = gmeta.size();
            gmeta.resize(gm0 + 32 * 4 * 2, 0);
            int64_t slots = 0;
            for (int w = 0; w < 4; ++w) {
                int64_t longest = 0;
                for (int gg = 0; gg < 8; ++gg) longest = std::max(longest, grp_tot[gorder[gg * 4 + w]]);
                const int rounds = (int)((longest + 7) / 8 + 1) & ~1;      // even: the kernel keeps two rounds in flight
                it.round0[w] = (int32_t)(sids.size() / 64);
                it.nr[w] = rounds;
                const size_t base = sids.size();
                sids.resize(base + size_t(rounds) * 64, -1);
                for (int gg = 0; gg < 8; ++gg) {
                    const int gi = gorder[gg * 4 + w];
                    int32_t* gm = &gmeta[gm0 + ((size_t(w) * 8 + size_t(gg)) * 4) * 2];
                    int f = 0;
                    for (int kk = 0; kk < 4; ++kk) {
                        if (kk < grp_n[gi]) {
                            const int rr = grp_rows[gi][kk];
                            for (int32_t id : rem[rr]) {
                                sids[base + size_t(f >> 3) * 64 + size_t(gg) * 8 + size_t(f & 7)] = id;
                                ++f;
                            }
                            const uint32_t end = rem[rr].empty() ? 0xFFFFFFu : uint32_t(f);
                            gm[2 * kk] = int32_t(end << 8 | uint32_t(rr));
                            const float sc = rowscale[size_t(lo + rr)];
                            memcpy(&gm[2 * kk + 1], &sc, 4);
                        } else {
                            gm[2 * kk] = int32_t(0xFFFFFFFFu);
                        }
                    }
                }
                slots += rounds;
            }
            it.cost = double(it.nq) * 4 * c_step + double(slots) / 4.0 * c_round + 1.0;
            list.push_back(it);
        }
        for (int32_t c : touched) { cnt[c] = 0; kpos[c] = -1; }
    }
    // launch order inside a panel: most expensive first; the pieces of a block stay together (consecutive slots)
    std::vector<int32_t> perm(list.size());
    std::iota(perm.begin(), perm.end(), 0);
    std::stable_sort(perm.begin(), perm.end(), [&](int32_t x, int32_t y) {
        const Item& a = list[size_t(x)];
        const Item& c = list[size_t(y)];
        if (a.b0 == c.b0) return a.k < c.k;
        // (pieces of one block carry nearly equal costs; order blocks by the cost of their first piece)
        const double ka = list[size_t(x - a.k)].cost, kc = list[size_t(y - c.k)].cost;
        return ka != kc ? ka > kc : a.b0 < c.b0;
    });
    const size_t n_items = list.size();
    items.assign((n_items + 1) * kRec, 0);
    int32_t n_pslots = 0, n_cslots = 0;
    for (size_t i = 0; i < n_items; ++i) {
        const Item& it = list[size_t(perm[i])];
        int32_t* rec = &items[i * kRec];
        rec[0] = it.b0; rec[1] = it.quad0; rec[2] = it.nq; rec[3] = it.gmi; rec[4] = it.np;
        rec[5] = -1; rec[6] = -1; rec[7] = it.k;
        if (it.np > 1) {
            if (it.k == 0) { rec[5] = n_pslots; rec[6] = n_cslots; }
            else { rec[5] = items[(i - 1) * kRec + 5] + 1; rec[6] = items[(i - 1) * kRec + 6]; }
            if (it.k == it.np - 1) { n_pslots += it.np; ++n_cslots; }
        }
        for (int w = 0; w < 4; ++w) { rec[8 + w] = it.round0[w]; rec[12 + w] = it.nr[w]; }
    }
    {   // the empty record behind the last one: one piece, nothing to do, addresses that exist
        int32_t* rec = &items[n_items * kRec];
        rec[4] = 1; rec[5] = -1; rec[6] = -1;
    }
    sids.resize(sids.size() + 128, -1);                           // (round 0 / 1 of an item without rounds are still loaded)
    gmeta.resize(gmeta.size() + 32 * 4 * 2, int32_t(0xFFFFFFFFu));
    dcols.resize(dcols.size() + 128, 0);

    simrank_fused2_plan* pl = new simrank_fused2_plan;
    pl->n_items = (int32_t)n_items;
    pl->n_blocks = (int32_t)nblk;
    pl->n_pslots = n_pslots;
    pl->n_cslots = n_cslots;
    pl->n_quads = n_quads;
    pl->n_steps = steps_total;
    pl->nnz_covered = covered;
    pl->r_nnz = r_nnz;
    pl->ids16 = ids16 ? 1 : 0;
    pl->cap_panels = (int32_t)((std::max<int64_t>(M, K) + 31) / 32);
    int rc = upload_vec2(&pl->items, items);
    if (!rc) {
        if (ids16) {
            std::vector<uint16_t> d16(dcols.begin(), dcols.end());
            rc = upload_vec2(&pl->dcols16, d16);
        } else {
            rc = upload_vec2(&pl->dcols32, dcols);
        }
    }
    if (!rc) rc = upload_vec2(reinterpret_cast<uint32_t**>(&pl->abits), abits);
    if (!rc) rc = upload_vec2(reinterpret_cast<int32_t**>(&pl->gmeta), gmeta);
    if (!rc) {
        if (ids16) {
            std::vector<uint16_t> s16(sids.size());
            for (size_t i = 0; i < sids.size(); ++i) s16[i] = sids[i] < 0 ? uint16_t(0xFFFF) : uint16_t(sids[i]);
            rc = upload_vec2(&pl->sids16, s16);
        } else {
            rc = upload_vec2(&pl->sids32, sids);
        }
    }
#ifndef SIMRANK_HOST_ONLY
    if (!rc && n_pslots > 0) {
        rc = pool_alloc((void**)&pl->partials, size_t(n_pslots) * size_t(pl->cap_panels) * 32 * kFB * sizeof(float));
        if (!rc) {
            const size_t tb = size_t(n_cslots) * size_t(pl->cap_panels) * sizeof(int32_t);
            hipError_t e = hipMalloc((void**)&pl->tickets, tb);
            if (e == hipSuccess) e = hipMemset(pl->tickets, 0, tb);
            if (e != hipSuccess) { set_error("fused2 tickets: %s", hipGetErrorString(e)); rc = SIMRANK_ERR_HIP; }
        }
    }
    if (!rc) {
        hipError_t e = hipMalloc((void**)&pl->heads, 8 * 32 * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMemset(pl->heads, 0, 8 * 32 * sizeof(uint32_t));
        if (e != hipSuccess) { set_error("fused2 heads: %s", hipGetErrorString(e)); rc = SIMRANK_ERR_HIP; }
    }
#endif
    if (rc) {
        free_fused2_plan(pl);
        return rc;
    }
    g->fused2 = pl;
    return SIMRANK_OK;
}

// Tt (panel-blocked, y_rows_pad rows per panel) = (diag(rowscale) . A . X)^T, X panel-blocked
int launch_fused2_trans(const simrank_graph* g, const float* X, int64_t x_rows_pad, int64_t L, float* Y,
                        int64_t y_rows_pad, hipStream_t st) {
    simrank_fused2_plan* pl = g->fused2;
    SR_REQUIRE(pl, "graph has no persistent one-launch plan");
    SR_REQUIRE(aligned16(X) && aligned16(Y), "fused leg needs 16-byte aligned operands");
    SR_REQUIRE(x_rows_pad >= g->n_cols && (x_rows_pad + 1) * 128 < (int64_t(1) << 31) && x_rows_pad < (int64_t(1) << 24) - 1,
               "fused leg: operand of %lld rows per panel", (long long)x_rows_pad);
    Fused2Args a{};
    a.X = X; a.Y = Y;
    a.x_rows_pad = x_rows_pad; a.y_rows_pad = y_rows_pad;
    a.L = L; a.M = g->n_rows;
    a.n_panels = int32_t((L + 31) / 32);
    SR_REQUIRE(pl->n_pslots == 0 || a.n_panels <= pl->cap_panels, "fused leg: %d panels, plan sized for %d", a.n_panels,
               pl->cap_panels);
    a.n_items = pl->n_items;
    a.nt = (int32_t)(g->tun.stream_nt ? 1 : 0);
    a.x_sentinel = (int32_t)x_rows_pad;
    a.probe = (int32_t)g->tun.probe_flags;
    a.idx_mask = (int32_t)g->tun.probe_mask;
    a.cap_panels = pl->cap_panels;
    a.items = pl->items;
    a.rowscale = g->rowscale;
    a.partials = pl->partials; a.tickets = pl->tickets;
    a.heads = pl->heads;
    a.dcols16 = pl->dcols16; a.dcols32 = pl->dcols32; a.abits = pl->abits;
    a.gmeta = pl->gmeta; a.sids16 = pl->sids16; a.sids32 = pl->sids32;
#ifdef SIMRANK_HOST_ONLY
    SR_REQUIRE(false, "host-only build: no kernels");
#else
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        SR_HIP(hipGetDevice(&dev));
        SR_HIP(hipGetDeviceProperties(&prop, dev));
        n_cu = prop.multiProcessorCount;
    }
    const int64_t wg_per_cu = std::max<int64_t>(1, std::min<int64_t>(8, g->tun.fuse_wgs));
    const unsigned grid = (unsigned)(int64_t(n_cu) * wg_per_cu);
    SR_HIP(hipMemsetAsync(pl->heads, 0, 8 * 32 * sizeof(uint32_t), st));          // (calls on one graph are stream-ordered)
    if (pl->ids16)
        hipLaunchKernelGGL(fused2_kernel<true>, dim3(grid), dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL(fused2_kernel<false>, dim3(grid), dim3(256), 0, st, a);
#endif
    SR_HIP(hipGetLastError());
    return SIMRANK_OK;
}

}  // namespace simrank

#ifdef SIMRANK_F2_STAMPS
extern "C" __attribute__((visibility("default"))) int simrank_read_fused2_stamps(unsigned long long* out, int64_t cap, int32_t reset) {
    SR_HIP(hipDeviceSynchronize());
    unsigned n = 0;
    SR_HIP(hipMemcpyFromSymbol(&n, HIP_SYMBOL(simrank::g_f2n), sizeof(n)));
    n = std::min<unsigned>(n, simrank::kF2Cap);
    const int64_t m = std::min<int64_t>(n, cap);
    if (out && m > 0) SR_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(simrank::g_f2st), size_t(m) * 8 * sizeof(unsigned long long)));
    if (reset) {
        const unsigned z = 0;
        SR_HIP(hipMemcpyToSymbol(HIP_SYMBOL(simrank::g_f2n), &z, sizeof(z)));
    }
    return (int)m;
}
#endif

