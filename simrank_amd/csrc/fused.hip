// Leg 1 of the single-rank update as ONE launch: matrix cores and row gathers on the same L2-resident
// panel slice (gfx950, round 3).
//
//   Tt = (diag(rowscale) . A . X)^T        A = 0/1 CSR pattern, X and Tt panel-blocked
//
// Why.  The gather kernel of spmm.hip pulls one 128-byte segment per (entry, 32-column panel) through
// the texture path, and a vector-memory instruction costs that path ~16-20 cycles whatever its width
// (tools/micro/gather_shapes.hip: 8 lines x 16 B per lane = 50 B/clk/CU, 2 lines x 4 B per lane = 16):
// the leg is bound by the NUMBER of gathered segments.  Rows sorted by length put rows that share their
// hub columns next to each other: in a 128-row block of the bench graph two thirds of the entries sit
// in columns that at least two of the block's rows reference.  A column referenced by c rows of a block
// costs c segments when gathered and ONE when it feeds the matrix cores (B operand of a
// 128 x 16 x 32 step), so where a block shares enough columns (its DENSE SET) they are multiplied on
// MFMA — 0/1 pattern in bf16 x the operand split into three bf16 terms = exact f32 products — and
// only the REMAINDER is gathered, both by the SAME workgroup while the panel's slice of X is in its
// XCD's L2.  The separate dense_tiles launch of round 2 (blockdense.hip) streamed its operand rows a
// second time from HBM and handed partial sums over through memory; here a segment is loaded once per
// (block, panel) and nothing but the finished tile leaves the workgroup.
//
// One workgroup (4 waves) = one 128-row block x one 32-column panel:
//   1. MFMA phase (blocks with a dense set): the set is cut into steps of 16 columns, 4 steps = a quad;
//      the waves split the quads.  Per step a wave gathers the 16 operand segments with two
//      16-byte-per-lane loads (8 rows each: the fast shape), turns them into a B fragment through a
//      2 KiB LDS buffer of its own, expands the step's pattern bits (one byte per lane and 32-row
//      tile) into A fragments by a 4 KiB lookup table in LDS, and issues 4 tiles x 3 terms = 12
//      v_mfma_f32_32x32x16_bf16; software-pipelined, every wait counted.
//   2. The waves' accumulators are added into one LDS tile in wave order (deterministic).
//   3. Gather phase: the block's rows in descending remainder length, dealt to the waves 8 at a time
//      (one row per group of 8 lanes, 16 B per lane).  A wave's four passes form ONE stream of slot
//      instructions (slot j of pass ps gathers the j-th neighbour of each of the pass's 8 rows); the
//      host lays the ids out in that order, 64 per round of 8 slots, so a round is one coalesced id
//      load and 8 gathers with nothing else in front of them, and round r + 1 is in flight while round
//      r is summed.  Row sum = (MFMA part + gathered part) x rowscale, into the LDS tile.
//   4. Each wave stores one transposed 32 x 32 tile = 4 KiB contiguous of the panel-blocked Tt.
// Panels are bound to XCDs as in spmm.hip (blocks equal mod 8 share an L2; speed only); within a
// panel the heaviest blocks are launched first.  Bitwise reproducible: fixed order everywhere.
//
// Round 6.  (a) Which columns make a block's dense set is decided by what a QUAD of 64 columns saves in gathered entries
// (build_fused_plan, tuning fuse_min = 0 / fuse_pays) instead of a fixed count per column.  (b) The SYM instantiation is
// LEG 2 of a symmetric update in the same launch: the same product with Tt as the operand, then — instead of step 4 alone —
// the epilogue of SimRank.py:139-140 / :316 / :362 / :453 and the count of :74 on the wave's 32 x 32 tile in LDS, its store,
// and (right of the diagonal) step 4 as the store of the mirror image; tiles left of the diagonal are not computed
// (launch_fused_sym; taken where the dense sets hold most of the pattern, tuning fuse_sym).
#include <algorithm>
#include <cstring>
#include <numeric>
#include <vector>

#include <cstdlib>

#include <mutex>
#include <thread>

#include "common.h"
#include "fused_dev.h"

namespace simrank {

constexpr int kTS = 132;          // floats per column of the LDS tile (32 columns x 128 rows, transposed)
#ifndef SIMRANK_KMINUNITS
#define SIMRANK_KMINUNITS 64
#endif
#ifndef SIMRANK_KGROUPENTRIES
#define SIMRANK_KGROUPENTRIES 6144
#endif
#ifndef SIMRANK_KMAXREM
#define SIMRANK_KMAXREM 256
#endif
constexpr int kMinUnits = SIMRANK_KMINUNITS;       // units per panel below which blocks are grouped less
constexpr int kGroupEntries = SIMRANK_KGROUPENTRIES; // gathered entries a unit of several set-less blocks may hold
// fuse_min = 0 (build_fused_plan): what a quad of 64 columns must cover, and one 16-column step priced in gathered entries
constexpr int64_t kPaysGatherBound = 192, kPaysMfmaBound = 256, kStepInEntries = 77;
constexpr int kMaxRem = SIMRANK_KMAXREM;      // a row whose remainder would be longer sends all its columns to the dense set

struct FusedArgs {
    const float* X;
    float* Y;
    int64_t x_rows_pad, y_rows_pad;
    int64_t L, M;
    // (round 4: the operand may also be ROW-MAJOR — what a rank of a sharded update holds — and the transposed
    // result may go out in the chunks an all-to-all between column shards needs, simrank_spmm's t_block layout)
    int64_t x_panel_stride;   // floats between the slices of consecutive panels (blocked: x_rows_pad * 32, row-major: 32)
    uint32_t x_pitch;         // bytes between operand rows inside a slice (blocked: 128, row-major: 4 ld)
    int32_t x_bytes;          // bytes of panel 0's slice the buffer descriptor may read (panel p: minus 128 p when row-major)
    int32_t y_chunked;        // 1: Y[h n_cols (t_block + t_pad) + c (rows_in_block(h) + t_pad) + (a - h t_block)]
    int64_t t_block, t_pad;
    int32_t n_panels, n_units, nt;
    int32_t x_sentinel;
    int32_t idx_mask;         // DIAGNOSTIC (tuning "probe_mask"): operand row ids are ANDed with it (wrong results)
    int32_t meta_nt;          // the id streams are loaded non-temporally (they are re-read once per panel)
    int32_t probe;            // DIAGNOSTIC (tuning "probe_flags"): 1 no gather phase, 2 no stores, 4 no MFMA phase
    const float* rowscale;    // [M] (the last arriver of a split block scales the rows)
    const int32_t* units;     // [n_units][32]: first block, first quad (absolute), quads, index among the block's
                              // units, units of the block, partial-sum slot (-1: none), counter slot, block has a
                              // set, blocks of the unit (1..4; > 1 only without a set), then per wave: first round
                              // of its id stream and the round count after each of the unit's blocks
    float* partials;          // [partial slot][panel][32 x 128] sums of the units of split blocks
    int32_t* tickets;         // [counter slot][panel] arrivals (the last arriver resets it)
    int32_t n_pslots, n_cslots;
    const uint16_t* dcols16;
    const int32_t* dcols32;
    const uint4* abits;
    const int2* gmeta;        // [block][wave][lane group][4 rows]: (end of the row in the lane group's stream << 8 | row
                              // of the block; row 255: none, end 0xFFFFFF: no remainder), rowscale bits
    const uint16_t* sids16;   // id stream of the gather phase, 64 per round (0xFFFF: no neighbour), or
    const int32_t* sids32;    // the same in 32 bits (-1: no neighbour)
    // ---- leg 2 of a symmetric update (SYM, round 6): Y = epilogue(diag(rowscale) . A . X), X = Tt (K rows), Y = S' (M x M, M = L):
    // the 32 x 32 tiles on or above the diagonal are computed, put through the epilogue (SimRank.py:139-140, :316, :362,
    // :453 and the count of :74), stored, and their mirror image stored too.  The epilogue's operands share Y's layout.
    float coef, lbd;
    double eps;
    const uint8_t* ev;        // common-neighbour counts (u8), panel-blocked like Y (32 bytes per row and panel), or NULL
    const float* ap;          // prior, or NULL
    const float* prev;        // previous iterate (the count), or NULL
    unsigned long long* n_changed;
    int32_t set_diag, count_any;
};

#ifndef SIMRANK_HOST_ONLY          // (the sanitizer build of the host logic has no device code: common.h)
// The units of a split block hand their partial sums over with sc1 (write-through) stores, a drained vmcnt, a RELAXED
// agent-scope ticket and sc1 loads past the L1 (step 4b below): that is the cache-policy behaviour of gfx942 / gfx950,
// not a guarantee of the HIP memory model — refuse to build the device code for anything else.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx942__) && !defined(__gfx950__)
#error "fused.hip: the sc1 hand-off between the units of a split block is written for gfx942 / gfx950"
#endif
#ifdef SIMRANK_FUSED_STAMPS
// Diagnostic build only (bash tools/build_variant.sh fst -DSIMRANK_FUSED_STAMPS; tools/fused_stamps.py): the
// timeline of the workgroups with blockIdx in [g_fst_base, g_fst_base + kFstCap): s_memtime at the phase
// boundaries of wave 0, the XCC and the hardware id, into a __device__ array nothing else reads.
constexpr int kFstCap = 1 << 15;
__device__ unsigned long long g_fst[kFstCap * 8];
__device__ unsigned int g_fst_base;
__device__ __forceinline__ unsigned long long fst_now() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define FST(i)                                                                              \
    do {                                                                                    \
        const unsigned long long now_ = fst_now();                                          \
        if (threadIdx.x == 0 && blockIdx.x - g_fst_base < unsigned(kFstCap))                \
            g_fst[size_t(blockIdx.x - g_fst_base) * 8 + (i)] = now_;                        \
    } while (0)
#else
#define FST(i) do {} while (0)
#endif

#ifndef SIMRANK_FUSED_LB
#define SIMRANK_FUSED_LB 4     // waves per SIMD the register allocation aims at
#endif
// IDS16: 16-bit ids (fewer than 65535 operand rows; 0xFFFF marks an empty slot of the gather stream)
// SYM: leg 2 (epilogue, upper triangle + mirror) instead of leg 1 (transposed store)
#ifndef SIMRANK_FUSED_LB_SYM
#define SIMRANK_FUSED_LB_SYM 3     // ... of leg 2 (its epilogue's operands on top of the gather pipeline's two register sets: at 4
#endif                             // the compiler kept the accumulators of the matrix-core phase in scratch)
template <bool IDS16, bool SYM = false>
__global__ __launch_bounds__(256, SYM ? SIMRANK_FUSED_LB_SYM : SIMRANK_FUSED_LB) void fused_trans_kernel(const FusedArgs p) {
    __shared__ __attribute__((aligned(16))) float tile[32 * kTS];          // [column][row] of the block's result
    __shared__ __attribute__((aligned(16))) float bbuf_all[4 * 16 * 32];    // per wave: 16 operand segments
    __shared__ __attribute__((aligned(16))) uint4 lut[256];                 // pattern byte -> 8 bf16 (0 / 1.0)
    __shared__ __attribute__((aligned(16))) int2 gm_lds[kSub * 4 * 8 * 4];   // per block of the unit, wave, lane group: four rows

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t bid = blockIdx.x;
    const uint32_t local = bid >> 3;
    const int panel = int(local / uint32_t(p.n_units)) * 8 + int(bid & 7);
    if (panel >= p.n_panels) return;
    const uint32_t unit = local % uint32_t(p.n_units);
    const int32_t* un = p.units + size_t(unit) * 32;
    FST(0);
#ifdef SIMRANK_FUSED_STAMPS
    if (threadIdx.x == 0 && blockIdx.x - g_fst_base < unsigned(kFstCap)) {
        const unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);     // HW_REG_XCC_ID[3:0]
        const unsigned hw = __builtin_amdgcn_s_getreg((16 - 1) << 11 | 0 << 6 | 4);       // HW_REG_HW_ID[15:0]
        g_fst[size_t(blockIdx.x - g_fst_base) * 8 + 5] = (unsigned long long)xcc << 32 | hw;
        g_fst[size_t(blockIdx.x - g_fst_base) * 8 + 6] = (unsigned long long)panel << 32 | unit;
    }
#endif
    const int b0 = un[0];                                   // first (usually only) block of the unit
    const int64_t c0 = int64_t(panel) * 32;
    // (SYM) a block wholly left of the diagonal — its first row beyond the panel's last column — does nothing; the blocks of
    // a unit are consecutive, so the leading ones stay (every unit of a split block decides alike: no ticket is ever short)
    const int n_sub_all = un[8];
    const int n_sub_sym = SYM ? min(n_sub_all, int((c0 + 31) >> 7) - b0 + 1) : n_sub_all;
    if (SYM && n_sub_sym <= 0) return;
    const int g = lane >> 3, q = lane & 7, gbase = lane & ~7;
    const uint32_t qoff = uint32_t(q) * 16u;

    const float* xbase = p.X + int64_t(panel) * p.x_panel_stride;
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(xbase), 0, p.x_bytes - (p.x_pitch == 128u ? 0 : panel * 128), 0x00020000);
    const uint32_t pitch = p.x_pitch;
    const int sent = p.x_sentinel;

    const int quad0 = un[1];                                // this unit's quads of the block's dense set
    const int nq = (p.probe & 4) ? 0 : un[2];
    const int unit_k = un[3], unit_nb = un[4], pslot = un[5], cslot = un[6];
    const bool has_set = !(p.probe & 4) && un[7] != 0;      // the block has a dense set (this unit may be one of several)
    const int n_sub = n_sub_sym;                            // blocks of the unit (SYM: those that reach the diagonal)
    const int per = (nq + 3) >> 2;                          // quads per wave
    const int n_active = per ? (nq + per - 1) / per : 0;    // waves that have MFMA work

    // ---- (round 5) the matrix-core phase's first loads depend on the unit record only: the operand-row ids and pattern
    // bits of the wave's first two quads are requested before anything else, so that the chain unit record -> ids ->
    // operand rows -> first MFMA (three dependent trips to memory, ~3 000 cycles each under the gather traffic, for a set
    // of 1 - 3 quads per wave) runs beside the prologue instead of behind it.  Unconditional (the arrays carry one quad
    // of padding): every load count stays static.
    const int q_lo = wave * per, q_hi = min(nq, q_lo + per);
    auto ld_ids = [&](int qd) -> int {
        const size_t at = size_t(quad0 + qd) * 64 + lane;
        if constexpr (IDS16) return int(p.dcols16[at]) & p.idx_mask;
        else return p.dcols32[at] & p.idx_mask;
    };
    const int qa0 = max(0, min(q_lo, nq - 1)), qa1 = max(0, min(q_lo + 1, q_hi - 1));
    int ids_a = ld_ids(qa0);
    uint4 aw_a = p.abits[size_t(quad0 + qa0) * 64 + lane];
    int ids_b = ld_ids(qa1);
    uint4 aw_b = p.abits[size_t(quad0 + qa1) * 64 + lane];

    // ---- everything the gather phase needs that does not depend on another load is requested now: the
    // wave's stream geometry (scalar, part of the unit record), the rows of every lane group (into LDS),
    // the ids of the first two rounds
    const int32_t* wm = un + 9 + wave * 5;
    const int round0 = wm[0];
    const int e1 = wm[1], e2 = wm[2], e3 = wm[3], e4 = wm[4];      // rounds up to the end of each block of the unit
    const int n_rounds = (p.probe & 1) ? 0 : (n_sub == 1 ? e1 : n_sub == 2 ? e2 : n_sub == 3 ? e3 : e4);
    auto fix_sid = [&](int v) -> int {                       // a raw id of the stream -> operand row (empty slot: past the end)
        if constexpr (IDS16) return v == 0xFFFF ? sent : (v & p.idx_mask);
        else return v < 0 ? sent : (v & p.idx_mask);
    };
    auto ld_sid_raw = [&](int r) -> int {                    // round r of the stream: this lane's id, as stored
        const size_t at = (size_t(round0) + size_t(min(r, max(n_rounds - 1, 0)))) * 64 + lane;
        if constexpr (IDS16) return p.meta_nt ? int(__builtin_nontemporal_load(p.sids16 + at)) : int(p.sids16[at]);
        else return p.meta_nt ? __builtin_nontemporal_load(p.sids32 + at) : p.sids32[at];
    };
    auto ld_sid = [&](int r) -> int { return fix_sid(ld_sid_raw(r)); };
    // (round 5) the row records and the ids of the first two rounds are REQUESTED together and only then used: the
    // compiler had put the use of the first id load in front of the second load (two trips to memory one after the other
    // in every workgroup's prologue) and read the row records through a flat load selected against a stack slot
    int iv0 = sent, iv1 = sent;
    {   // lane (g, q): row q & 3 of lane group g, blocks q >> 2 and 2 + (q >> 2) of the unit
        const int sbA = q >> 2, sbB = 2 + (q >> 2);
        const int2* src = p.gmeta + ((size_t(un[29]) * 4 + wave) * 8 + g) * 4 + (q & 3);
        int2* dst = gm_lds + (wave * 8 + g) * 4 + (q & 3);
        int2 a = src[size_t(min(sbA, n_sub - 1)) * 4 * 8 * 4];
        int2 c = src[size_t(min(sbB, n_sub - 1)) * 4 * 8 * 4];
        int raw0 = 0, raw1 = 0;
        if (n_rounds > 0) {
            raw0 = ld_sid_raw(0);
            raw1 = ld_sid_raw(1);
        }
        __builtin_amdgcn_sched_barrier(0);                   // (every load of the prologue is on its way before the first use)
        if (sbA >= n_sub) a = make_int2(int(0xFFFFFFFFu), 0);
        if (sbB >= n_sub) c = make_int2(int(0xFFFFFFFFu), 0);
        dst[sbA * 4 * 8 * 4] = a;
        dst[sbB * 4 * 8 * 4] = c;
        if (n_rounds > 0) {
            iv0 = fix_sid(raw0);
            iv1 = fix_sid(raw1);
        }
    }

    // ... and the operand rows of the wave's first two matrix-core steps follow as soon as their ids are there: they travel
    // while the pattern table is built and the workgroup meets at its first barrier
    float4 rA0 = make_float4(0.f, 0.f, 0.f, 0.f), rA1 = rA0, rB0 = rA0, rB1 = rA0;
    auto issue = [&](int ids, int s4, float4& x0, float4& x1) {   // two wave instructions, 8 operand rows each
        x0 = ld_seg_p(srd, __shfl(ids, s4 * 16 + g), pitch, qoff);
        x1 = ld_seg_p(srd, __shfl(ids, s4 * 16 + 8 + g), pitch, qoff);
    };
    if (q_lo < q_hi) {
        issue(ids_a, 0, rA0, rA1);
        issue(ids_a, 1, rB0, rB1);
    }

    FST(1);
    // ---------------------------------------------------------------- 1. MFMA phase
    if (nq > 0) {
        {   // lookup table: entry e, dword d: low half = bit 2d, high half = bit 2d + 1 (bf16 1.0 = 0x3F80)
            const unsigned e = threadIdx.x;
            uint4 v;
            v.x = ((e >> 0) & 1u) * 0x3F80u | ((e >> 1) & 1u) * 0x3F800000u;
            v.y = ((e >> 2) & 1u) * 0x3F80u | ((e >> 3) & 1u) * 0x3F800000u;
            v.z = ((e >> 4) & 1u) * 0x3F80u | ((e >> 5) & 1u) * 0x3F800000u;
            v.w = ((e >> 6) & 1u) * 0x3F80u | ((e >> 7) & 1u) * 0x3F800000u;
            lut[e] = v;
        }
        __syncthreads();
        f32x16 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
        if (q_lo < q_hi) {
            // Software pipeline over the wave's steps, one quad (4 steps) per loop iteration: while the 12
            // MFMAs of a step run, the next step goes registers -> LDS -> fragment -> three bf16 terms and
            // the operand loads of the two steps after it are in flight (register sets rA / rB, one 2 KiB
            // LDS buffer per wave).  Every iteration issues the same loads in the same order — the prefetch
            // past the wave's last quad re-reads that quad — so the waits are counted, never drained; the
            // steps that pad a set to whole quads multiply zero pattern bits.
            float* bbuf = bbuf_all + wave * (16 * 32);
            const int n = lane & 31, h = lane >> 5;
            // (round 6, measured and dropped: TWO fp16 terms, x 2^14 = hi + lo 2^-11 — 22 significant bits, two MFMAs per tile
            // and step instead of three; pl32768d32 4.78 -> 4.75 ms, N = 65536 17.80 -> 17.58, MovieLens-shaped 1.39 -> 1.32 ms per
            // loop body: the phase is not bound by its MFMAs, and the operand would need a proven range;
            // profiles/r06_terms2_ab.log)
            struct Terms { uint32_t lo[4], mid[4], hi[4]; };
            // B fragment through the wave's LDS buffer: [k][n] -> lane (n, h) holds k = 8h .. 8h + 7
            auto stage = [&](const float4& x0, const float4& x1, float (&x)[8]) {
                wave_lds_order();
                *reinterpret_cast<float4*>(bbuf + g * 32 + q * 4) = x0;
                *reinterpret_cast<float4*>(bbuf + (8 + g) * 32 + q * 4) = x1;
                wave_lds_order();
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = bbuf[(8 * h + j) * 32 + n];
            };
            auto split = [&](const float (&x)[8], Terms& t) {
#pragma unroll
                for (int j = 0; j < 4; ++j) split3f(x[2 * j], x[2 * j + 1], t.hi[j], t.mid[j], t.lo[j]);
            };
            auto mma = [&](const Terms& t, uint32_t aw) {
                const bf16x8 bl = frag(t.lo[0], t.lo[1], t.lo[2], t.lo[3]);
                const bf16x8 bm = frag(t.mid[0], t.mid[1], t.mid[2], t.mid[3]);
                const bf16x8 bh = frag(t.hi[0], t.hi[1], t.hi[2], t.hi[3]);
                // per tile: smallest term first (as blockdense.hip)
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    const bf16x8 a = __builtin_bit_cast(bf16x8, lut[(aw >> (8 * tt)) & 255u]);
                    acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bl, acc[tt], 0, 0, 0);
                    acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bm, acc[tt], 0, 0, 0);
                    acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bh, acc[tt], 0, 0, 0);
                }
            };
            Terms tA, tB;
            float x[8];
            stage(rA0, rA1, x);
            split(x, tA);
            issue(ids_a, 2, rA0, rA1);
            for (int qd = q_lo; qd < q_hi; ++qd) {
                // tA = step 0 of quad qd; rB = its step 1 (arriving); rA = its step 2 (in flight)
                stage(rB0, rB1, x);
                issue(ids_a, 3, rB0, rB1);
                mma(tA, aw_a.x);
                split(x, tB);
                stage(rA0, rA1, x);
                issue(ids_b, 0, rA0, rA1);
                mma(tB, aw_a.y);
                split(x, tA);
                stage(rB0, rB1, x);
                issue(ids_b, 1, rB0, rB1);
                mma(tA, aw_a.z);
                split(x, tB);
                stage(rA0, rA1, x);
                issue(ids_b, 2, rA0, rA1);
                mma(tB, aw_a.w);
                split(x, tA);
                ids_a = ids_b;
                aw_a = aw_b;
                const int nx = min(qd + 2, q_hi - 1);
                ids_b = ld_ids(nx);
                aw_b = p.abits[size_t(quad0 + nx) * 64 + lane];
            }
        }
        // ------------------------------------------------------------ 2. sum of the waves, in wave order
        // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5):
        // four consecutive registers are four consecutive rows = 16 bytes of the transposed tile
        {
            const int n = lane & 31, h = lane >> 5;
            for (int w = 0; w < n_active; ++w) {
                if (wave == w) {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int i4 = 0; i4 < 4; ++i4) {
                            float4* dst = reinterpret_cast<float4*>(tile + n * kTS + 32 * t + 8 * i4 + 4 * h);
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
    }
    // (round 4) a block cut into units: matrix-core units (a share of the set's columns, no rows) and gather units (a
    // share of the rows, no columns of the set) all leave RAW sums in their tile and publish it after the gather
    // phase (step 4b); a gather unit's tile starts as zeros
    const bool split_unit = un[30] != 0;
    if (split_unit && nq == 0) {
        for (int x4 = threadIdx.x; x4 < 32 * kTS / 4; x4 += 256)
            reinterpret_cast<float4*>(tile)[x4] = make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();
    }
    const bool has_d = has_set || split_unit;                 // the tile holds something before the rows are emitted
    // (SYM) the convergence count of this wave's elements; count_any: once this wave's striped counter is known to be non-zero
    // the previous iterate is not read any more (`_converged` uses the sum as a truth value only, SimRank.py:74-77)
    unsigned changed = 0;
    const unsigned slot = ((blockIdx.x * 4u + unsigned(wave)) * 7u) % SIMRANK_CHANGED_SLOTS;
    bool check_prev = false;
    if constexpr (SYM) {
        check_prev = p.prev != nullptr;
        if (check_prev && p.count_any) check_prev = __hip_atomic_load(p.n_changed + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;
        check_prev = __builtin_amdgcn_readfirstlane(int(check_prev)) != 0;
    }

    FST(2);
    // ---------------------------------------------------------------- 3. gather phase (the remainder)
    // Every lane group owns four rows of the block, their remainder ids concatenated into ONE stream (the
    // host balances the 32 streams of a block); a slot instruction gathers the next neighbour of all 8
    // lane groups, a round is 8 slots = one coalesced id load.  Each lane group keeps one running sum;
    // when its stream reaches the end of a row the row goes into the LDS tile ((MFMA part + gathered
    // part) x rowscale) and the next row's (row, scale, end) comes from LDS.
    // Round r + 1 is in flight while round r is summed; the streams carry no padding except the tail of
    // the last round (ids marked empty load zeros through the buffer range check).
    {
        // the unit's blocks one after the other; the wave's rounds run through them without a gap (the
        // gathers of the next block's first round are in flight while this block's tile is stored)
        int sb = 0;                                           // block of the unit the stream is in
        int r_base = 0;                                       // first round of that block
        auto end_of = [&](int k) __attribute__((always_inline)) -> int { return k == 0 ? e1 : k == 1 ? e2 : k == 2 ? e3 : e4; };
        int r_end = (p.probe & 1) ? 0 : e1;                   // rounds up to the end of block sb
        const int2* gmp = gm_lds + (wave * 8 + g) * 4;        // + sb * 128: the lane group's rows in block sb
        wave_lds_order();
        // (row, scale, end) of the row the lane group is in: .x = row of the block or -1, .y = scale bits,
        // .z = end of the row in the group's stream or -1
        auto unpack = [](const int2& m) __attribute__((always_inline)) -> int3 {
            const unsigned u = unsigned(m.x);
            const int row = int(u & 255u), end = int(u >> 8);
            return make_int3(row == 255 ? -1 : row, m.y, end == 0xFFFFFF ? -1 : end);
        };
        int3 m_cur = unpack(gmp[0]);
        float4 cur = make_float4(0.f, 0.f, 0.f, 0.f);
        int krow = 0;                                         // rows of the lane group finished in this block
        auto emit = [&](const int3& m, const float4& sv) __attribute__((always_inline)) {
            if (m.x >= 0) {
                const float sc = split_unit ? 1.0f : __int_as_float(m.y);     // (a split block is scaled by its last arriver)
                float* tp = tile + (4 * q) * kTS + m.x;
                const float d0 = has_d ? tp[0] : 0.f, d1 = has_d ? tp[kTS] : 0.f;
                const float d2 = has_d ? tp[2 * kTS] : 0.f, d3 = has_d ? tp[3 * kTS] : 0.f;
                tp[0] = (sv.x + d0) * sc;
                tp[kTS] = (sv.y + d1) * sc;
                tp[2 * kTS] = (sv.z + d2) * sc;
                tp[3 * kTS] = (sv.w + d3) * sc;
            }
        };
        // slot f of the lane group's stream (of this block) has been added: was it the last of the current row?
        auto row_end = [&](int f) __attribute__((always_inline)) {
            if (f + 1 == m_cur.z) {
                emit(m_cur, cur);
                cur = make_float4(0.f, 0.f, 0.f, 0.f);
                ++krow;
                m_cur = unpack(gmp[sb * 128 + min(krow, 3)]);
                if (krow > 3) m_cur.z = -1;
            }
        };
        // every round of block sb has been summed: its rows without a remainder, then the tile goes out
        auto finish = [&]() __attribute__((always_inline)) {
            for (int k = krow; k < 4; ++k) emit(unpack(gmp[sb * 128 + k]), make_float4(0.f, 0.f, 0.f, 0.f));
            FST(3);
            __syncthreads();
            const int row0 = (b0 + sb) * kFB;
            // ------------------------------------------------------------ 4b. the units of a split block meet in memory
            // (a workgroup must not outlive its panel's turn in the L2): every unit publishes its raw sums — written
            // THROUGH (sc1) and read past the L1 (sc1): round 3's agent-scope release / acquire pair around plain
            // accesses wrote back every dirty line of the XCD's L2 — and takes a ticket; the LAST arriver adds them in
            // unit order (whoever it is: same bits), scales the rows and stores the tile, the others are done.
            // No unit ever waits for another.
            if (unit_nb > 1) {
                typedef unsigned v4u __attribute__((ext_vector_type(4)));
                float* mine = p.partials + (size_t(pslot) * p.n_panels + panel) * (32 * kFB);
                const __amdgpu_buffer_rsrc_t msrd = __builtin_amdgcn_make_buffer_rsrc(mine, 0, 32 * kFB * 4, 0x00020000);
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int x = threadIdx.x + it * 256;      // 1024 float4 = 32 columns x 128 rows
                    const int c = x >> 5, r4 = (x & 31) * 4;
                    const float4 v = *reinterpret_cast<const float4*>(tile + c * kTS + r4);
                    v4u o;
                    o.x = __float_as_uint(v.x); o.y = __float_as_uint(v.y); o.z = __float_as_uint(v.z); o.w = __float_as_uint(v.w);
                    __builtin_amdgcn_raw_buffer_store_b128(o, msrd, (c * kFB + r4) * 4, 0, 16);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                int* flag = reinterpret_cast<int*>(bbuf_all);
                if (threadIdx.x == 0) {
                    int* tk = p.tickets + size_t(cslot) * p.n_panels + panel;
                    const int t = __hip_atomic_fetch_add(tk, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const int last = t == unit_nb - 1;
                    if (last) __hip_atomic_store(tk, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
                    *flag = last;
                }
                __syncthreads();
                if (!*flag) {
                    sb = n_sub;                                  // (nothing left to do for this unit)
                    return;
                }
                const int first = pslot - unit_k;                // the block's units own consecutive slots
                // (round 5) a thread's four pieces of a unit's tile are requested together: the loop used to wait for every
                // single 16-byte load — 4 x units dependent trips to the L2 in the last arriver (two units per trip as well
                // spilled 47 registers: the gather pipeline's two register sets are live across this code)
                float4 acc4[4];
                auto piece = [&](int k, int it) -> float4 {
                    const int x = threadIdx.x + it * 256;
                    const int c = x >> 5, r4 = (x & 31) * 4;
                    const float* src = p.partials + (size_t(first + k) * p.n_panels + panel) * (32 * kFB);
                    const __amdgpu_buffer_rsrc_t ssrd =
                        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 32 * kFB * 4, 0x00020000);
                    const v4u w = __builtin_amdgcn_raw_buffer_load_b128(ssrd, (c * kFB + r4) * 4, 0, 16);
                    return make_float4(__uint_as_float(w.x), __uint_as_float(w.y), __uint_as_float(w.z), __uint_as_float(w.w));
                };
                for (int k = 0; k < unit_nb; ++k) {
                    float4 va[4];
#pragma unroll
                    for (int it = 0; it < 4; ++it) va[it] = piece(k, it);
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        if (k == 0) acc4[it] = va[it];
                        else { acc4[it].x += va[it].x; acc4[it].y += va[it].y; acc4[it].z += va[it].z; acc4[it].w += va[it].w; }
                    }
                }
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int x = threadIdx.x + it * 256;
                    const int c = x >> 5, r4 = (x & 31) * 4;
                    float scv[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) scv[j] = row0 + r4 + j < p.M ? p.rowscale[row0 + r4 + j] : 0.f;
                    float4 a4 = acc4[it];
                    a4.x *= scv[0]; a4.y *= scv[1]; a4.z *= scv[2]; a4.w *= scv[3];
                    *reinterpret_cast<float4*>(tile + c * kTS + r4) = a4;
                }
                __syncthreads();
            }
            const int nrows = int(min(int64_t(kFB), p.M - row0));
            const int rows_out = max(0, min(32, nrows - 32 * wave));
            const int cols_here = int(min(int64_t(32), p.L - c0));
            bool mirror = true;                               // (leg 1: every tile goes out transposed)
            if constexpr (SYM) {
                // ------------------------------------------------------------ 3b. leg 2: the epilogue, in the tile
                // The wave owns rows 32 wave .. + 31 of the block: one 32 x 32 tile of the result.  Left of the diagonal: nothing
                // (the tile above it supplies it); on it: every element computed and stored once; right of it: stored, and its
                // mirror image (step 4, what leg 1 calls the transposed store).  Lane group g takes rows g, g + 8, g + 16,
                // g + 24 of the tile, lane q of it columns 4 q .. 4 q + 3: one 16-byte piece of a row of Y / the previous
                // iterate / the prior, one 4-byte word of the counts.  Order of operations as the gather kernel's epilogue
                // (spmm.hip emit_row3): x coef, x (1 - 2^-count), blend with the prior, diagonal, count, store.
                const int64_t rt_row0 = int64_t(row0) + 32 * wave;
                const bool on_diag = c0 == rt_row0;
                mirror = c0 > rt_row0;                         // (both are multiples of 32)
                if (rows_out > 0 && (mirror || on_diag)) {
                    typedef unsigned v4u __attribute__((ext_vector_type(4)));
                    const int64_t pbase = int64_t(panel) * p.y_rows_pad;      // rows of this panel in Y, prev, prior, counts
                    const int bytes_f = int(p.y_rows_pad * 128);              // (below 2^31: launch_fused_sym)
                    const __amdgpu_buffer_rsrc_t ysrd2 = __builtin_amdgcn_make_buffer_rsrc(p.Y + pbase * 32, 0, bytes_f, 0x00020000);
                    const __amdgpu_buffer_rsrc_t osrd = __builtin_amdgcn_make_buffer_rsrc(
                        const_cast<float*>(p.prev ? p.prev + pbase * 32 : p.Y), 0, p.prev ? bytes_f : 0, 0x00020000);
                    const __amdgpu_buffer_rsrc_t asrd = __builtin_amdgcn_make_buffer_rsrc(
                        const_cast<float*>(p.ap ? p.ap + pbase * 32 : p.Y), 0, p.ap ? bytes_f : 0, 0x00020000);
                    const __amdgpu_buffer_rsrc_t esrd = __builtin_amdgcn_make_buffer_rsrc(
                        const_cast<uint8_t*>(p.ev ? p.ev + pbase * 32 : reinterpret_cast<const uint8_t*>(p.Y)), 0,
                        p.ev ? int(p.y_rows_pad * 32) : 0, 0x00020000);
                    const int nvalid = max(0, min(4, cols_here - 4 * q));
                    const float keep = 1.0f - p.lbd;
                    // (two rows of a lane group per batch of epilogue loads: with all four the two register sets of the gather
                    // pipeline, which are live across this code, spilled)
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                    unsigned w_ev[2];
                    v4u w_ap[2], w_old[2];
#pragma unroll
                    for (int it = 0; it < 2; ++it) {          // both rows' loads are requested before the first use
                        const int rr = 8 * (2 * half + it) + g;
                        const uint32_t a = uint32_t(rt_row0 + rr);
                        const bool on = rr < rows_out;
                        const uint32_t row = on ? a : 0x7FFFFF0u;             // (off: past the descriptor's end: zeros)
                        w_ev[it] = __builtin_amdgcn_raw_buffer_load_b32(esrd, int(row * 32u + 4u * q), 0, 2);
                        w_ap[it] = __builtin_amdgcn_raw_buffer_load_b128(asrd, int(row * 128u + qoff), 0, 2);
                        w_old[it] = check_prev ? __builtin_amdgcn_raw_buffer_load_b128(osrd, int(row * 128u + qoff), 0, 2)
                                               : v4u{0u, 0u, 0u, 0u};
                    }
#pragma unroll
                    for (int it = 0; it < 2; ++it) {
                        const int rr = 8 * (2 * half + it) + g;
                        if (rr < rows_out) {
                            const int64_t a = rt_row0 + rr;
                            float* tp = tile + (4 * q) * kTS + 32 * wave + rr;
                            float o[4] = {tp[0] * p.coef, tp[kTS] * p.coef, tp[2 * kTS] * p.coef, tp[3 * kTS] * p.coef};
                            if (p.ev) {
                                const unsigned w = w_ev[it];
                                o[0] *= 1.0f - __builtin_ldexpf(1.0f, -int(w & 255u));
                                o[1] *= 1.0f - __builtin_ldexpf(1.0f, -int((w >> 8) & 255u));
                                o[2] *= 1.0f - __builtin_ldexpf(1.0f, -int((w >> 16) & 255u));
                                o[3] *= 1.0f - __builtin_ldexpf(1.0f, -int(w >> 24));
                            }
                            if (p.ap) {
                                const float pr[4] = {__uint_as_float(w_ap[it].x), __uint_as_float(w_ap[it].y),
                                                     __uint_as_float(w_ap[it].z), __uint_as_float(w_ap[it].w)};
#pragma unroll
                                for (int i = 0; i < 4; ++i) o[i] = keep * o[i] + p.lbd * pr[i];
                            }
                            if (p.set_diag) {
                                const int64_t d = a - (c0 + 4 * q);
#pragma unroll
                                for (int i = 0; i < 4; ++i)
                                    if (d == i) o[i] = 1.0f;
                            }
                            if (check_prev) {
                                const float old[4] = {__uint_as_float(w_old[it].x), __uint_as_float(w_old[it].y),
                                                      __uint_as_float(w_old[it].z), __uint_as_float(w_old[it].w)};
#pragma unroll
                                for (int i = 0; i < 4; ++i)
                                    changed += (i < nvalid && fabs(double(o[i]) - double(old[i])) > p.eps) ? (mirror ? 2u : 1u) : 0u;
                            }
                            tp[0] = o[0]; tp[kTS] = o[1]; tp[2 * kTS] = o[2]; tp[3 * kTS] = o[3];     // (what the mirror image gets)
                            if (!(p.probe & 2)) {
                                const int yoff = int(uint32_t(a) * 128u + qoff);
                                if (nvalid == 4) {
                                    v4u out;
                                    out.x = __float_as_uint(o[0]); out.y = __float_as_uint(o[1]);
                                    out.z = __float_as_uint(o[2]); out.w = __float_as_uint(o[3]);
                                    if (p.nt) __builtin_amdgcn_raw_buffer_store_b128(out, ysrd2, yoff, 0, 2);
                                    else __builtin_amdgcn_raw_buffer_store_b128(out, ysrd2, yoff, 0, 0);
                                } else {
#pragma unroll
                                    for (int i = 0; i < 4; ++i)
                                        if (i < nvalid) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o[i]), ysrd2, yoff + 4 * i, 0, 0);
                                }
                            }
                        }
                    }
                    }
                    wave_lds_order();                          // the mirror image reads what the wave's other lanes wrote
                }
            }
            // ------------------------------------------------------------ 4. transposed store (SYM: the mirror image)
            if (rows_out > 0 && mirror && !(p.probe & 2)) {
                // panel-blocked Tt: the wave's 32 x 32 tile is 4 KiB contiguous, element (c, r) at c * 32 + r;
                // chunked Tt (a sharded rank): 32 segments of 128 bytes, one per output column, in the chunk of the
                // row block the rows belong to (a 128-row block never straddles two chunks: t_block % 128 == 0)
                int cstride = 32;
                float* base = p.Y + ((int64_t(row0 >> 5) + wave) * p.y_rows_pad + c0) * 32;
                if (p.y_chunked) {
                    const int64_t h = row0 / p.t_block;
                    const int64_t rows_h = min(p.t_block, p.M - h * p.t_block);
                    cstride = int(rows_h + p.t_pad);
                    base = p.Y + h * p.L * (p.t_block + p.t_pad) + c0 * cstride + (row0 - h * p.t_block) + 32 * wave;
                }
                const float* tw = tile + 32 * wave;
                const __amdgpu_buffer_rsrc_t ysrd =
                    __builtin_amdgcn_make_buffer_rsrc(base, 0, (31 * cstride + 32) * 4, 0x00020000);
                if ((rows_out & 3) == 0) {
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int x = lane + it * 64;
                        const int c = x >> 3;
                        const int r4 = (x & 7) * 4;
                        if (c < cols_here && r4 < rows_out) {
                            const float4 v = *reinterpret_cast<const float4*>(tw + c * kTS + r4);
                            typedef unsigned v4u __attribute__((ext_vector_type(4)));
                            v4u o;
                            o.x = __float_as_uint(v.x); o.y = __float_as_uint(v.y);
                            o.z = __float_as_uint(v.z); o.w = __float_as_uint(v.w);
                            const int off = (c * cstride + r4) * 4;
                            // cache policy of the tile store (aux: 1 = sc0, 2 = nt, 16 = sc1): sc1 writes through
                            // and drops the line, so the output does not take the L2 from the panel's slice
                            switch (p.nt) {
                                case 0: __builtin_amdgcn_raw_buffer_store_b128(o, ysrd, off, 0, 0); break;
                                case 1: __builtin_amdgcn_raw_buffer_store_b128(o, ysrd, off, 0, 2); break;
                                case 2: __builtin_amdgcn_raw_buffer_store_b128(o, ysrd, off, 0, 16); break;
                                default: __builtin_amdgcn_raw_buffer_store_b128(o, ysrd, off, 0, 17); break;
                            }
                        }
                    }
                } else {
                    for (int x = lane; x < 32 * 32; x += 64) {
                        const int c = x >> 5, r = x & 31;
                        if (c < cols_here && r < rows_out) base[c * cstride + r] = tw[c * kTS + r];
                    }
                }
            }
            ++sb;
            if (sb < n_sub) {
                __syncthreads();                              // the tile is free for the next block's rows
                r_base = r_end;
                r_end = (p.probe & 1) ? 0 : end_of(sb);
                krow = 0;
                cur = make_float4(0.f, 0.f, 0.f, 0.f);
                m_cur = unpack(gmp[sb * 128]);
            }
        };
        auto issue8 = [&](int iv, float4 (&v)[8]) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = ld_seg_p(srd, __shfl(iv, gbase + j), pitch, qoff);
        };
        auto consume = [&](const float4 (&v)[8], int r) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                cur.x += v[j].x; cur.y += v[j].y; cur.z += v[j].z; cur.w += v[j].w;
                row_end(8 * (r - r_base) + j);
            }
            while (sb < n_sub && r + 1 == r_end) finish();    // (also the blocks after it that have no rounds)
        };
        while (sb < n_sub && r_end == r_base) finish();       // leading blocks without rounds
        if (n_rounds > 0) {
            float4 vA[8], vB[8];
            issue8(iv0, vA);                                  // round 0
            int r = 0;
            // (round 6: the ids requested three rounds ahead instead of one — four id registers, ten spills around the loops —
            // made the leg SLOWER, 5.07 against 4.82 ms at pl32768d32, 18.8 against 17.9 at N = 65536: the loop does not wait
            // for its ids; profiles/r06_ids_ahead_ab.log, and the stand-alone replay tools/micro/gather_depth.hip agrees)
            while (r + 2 < n_rounds) {                        // at least two more rounds after r
                const int iv2 = ld_sid(r + 2);
                issue8(iv1, vB);
                consume(vA, r);
                iv1 = ld_sid(r + 3);
                issue8(iv2, vA);
                consume(vB, r + 1);
                r += 2;
            }
            if (r + 1 < n_rounds) {                           // vA = round r in flight, one more after it
                issue8(iv1, vB);
                consume(vA, r);
                consume(vB, r + 1);
            } else {
                consume(vA, r);
            }
        }
        while (sb < n_sub) finish();
    }
    if constexpr (SYM) {
        if (p.prev) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) changed += __shfl_down(changed, off);
            if (lane == 0 && changed) atomicAdd(p.n_changed + slot, (unsigned long long)changed);
        }
    }
    FST(4);
}

#endif  // SIMRANK_HOST_ONLY

template <typename T>
static int upload_vec(T** d, const std::vector<T>& h) {
    const size_t bytes = std::max<size_t>(16, h.size() * sizeof(T));
    SR_HIP(plan_alloc((void**)d, bytes));
    if (!h.empty()) SR_HIP(plan_upload(*d, h.data(), h.size() * sizeof(T)));
    return SIMRANK_OK;
}

void free_fused_plan(simrank_fused_plan* p) {
    if (!p) return;
    plan_free(p->units);
#ifndef SIMRANK_HOST_ONLY
    (void)pool_free(p->partials); (void)pool_free(p->tickets);
#endif
    plan_free(p->dcols16); plan_free(p->dcols32); plan_free(p->abits);
    plan_free(p->gmeta); plan_free(p->sids16); plan_free(p->sids32);
    delete p;
}

// Host side, per 128-row block:
//  * the dense set: columns referenced by >= fuse_min of its rows, plus every column of a row whose
//    remainder would otherwise exceed kMaxRem entries; dropped again (everything gathered) when it would
//    make fewer than fuse_steps 16-column steps — a short matrix-core phase costs a workgroup more in
//    dependent latencies than its few shared columns save;
//  * its pattern bits in A-fragment order;
//  * the gather streams: the block's rows are dealt to 32 lane groups (4 waves x 8), four rows each, so
//    that the groups' totals of remainder entries balance (longest row first to the lightest group with
//    room); a group's stream is its rows' remainder ids one after the other (rows without a remainder
//    last), a wave's stream is 64 ids per round: lane group g, slot q = id 8 r + q of the group's stream
//    (past its end: a marker).
int build_fused_plan(simrank_graph* g, const int32_t* rowptr, const int32_t* col, const float* rowscale) {
    const int64_t M = g->n_rows, K = g->n_cols;
    // fuse_min = 0: no fixed threshold — a block's columns by descending count, 64 at a time, while such a quad covers at
    // least `pays` entries (what its four matrix-core steps cost in gathered entries: fuse_quad_pays in common.h)
    const bool by_quads = g->tun.fuse_min == 0;
    const int64_t thr = std::max<int64_t>(2, g->tun.fuse_min);
    const int64_t min_steps = fuse_min_steps(g->tun, g->n_cols);
    const int64_t nblk = (M + kFB - 1) / kFB;
    // What a quad must cover (fuse_pays = -1): the two phases of the launch overlap across the workgroups of a CU, so work is
    // cheap to move TO the phase that is not the critical one.  A sample of the blocks (16 of them, evenly spaced) at the lower price says which that
    // is: where the matrix-core steps outweigh the gathered remainder (a MovieLens-shaped pattern: 90 % of the entries in
    // sets) a quad has to cover 256 entries, elsewhere (power-law graphs: the gathers are what the launch waits for) 192.
    // profiles/r06_fuse_pays_sweep.log: 256 is -9 % on the MovieLens-shaped update and +5 % on leg 1 at pl32768d32.
    int64_t pays = g->tun.fuse_pays;
    if (by_quads && pays < 0) {
        pays = kPaysGatherBound;
        std::vector<uint16_t> cnt(size_t(K), 0);
        std::vector<int32_t> touched;
        std::vector<int32_t> hist(size_t(kFB) + 1);
        int64_t s_steps = 0, s_rem = 0;
        const int64_t stride = std::max<int64_t>(1, nblk / 16);
        for (int64_t b = 0; b < nblk; b += stride) {
            const int64_t lo = b * kFB, hi = std::min<int64_t>(M, lo + kFB);
            touched.clear();
            std::fill(hist.begin(), hist.end(), 0);
            for (int32_t j = rowptr[lo]; j < rowptr[hi]; ++j)
                if (cnt[col[j]]++ == 0) touched.push_back(col[j]);
            for (int32_t c : touched) { ++hist[cnt[c]]; cnt[c] = 0; }
            // quads of the columns in descending count, from the histogram
            int64_t nq = 0, cov = 0, in_quad = 0, gain = 0;
            bool open = true;
            for (int c = kFB; c >= 1 && open; --c) {
                int64_t left = hist[size_t(c)];
                while (left > 0) {
                    const int64_t take = std::min<int64_t>(left, 64 - in_quad);
                    in_quad += take; gain += take * c; left -= take;
                    if (in_quad == 64) {
                        if (gain < pays) { open = false; in_quad = 0; gain = 0; break; }
                        ++nq; cov += gain; in_quad = 0; gain = 0;
                    }
                }
            }
            if (open && in_quad > 0 && gain >= pays) { ++nq; cov += gain; }
            if (4 * nq < min_steps) { nq = 0; cov = 0; }
            s_steps += 4 * nq;
            s_rem += int64_t(rowptr[hi] - rowptr[lo]) - cov;
        }
        if (kStepInEntries * s_steps > s_rem) pays = kPaysMfmaBound;
    }
    // 16-bit ids while 0xFFFF, the marker of an empty slot of a gather stream, is no column's id.  (Exactly 65536 operand rows
    // — BASELINE config 5 — would need that one column kept out of the streams: forcing it into the dense set of every
    // block that references it was tried and made config 5 slower, the column being a hub that nearly every block
    // references; what 16-bit ids are worth was measured at config 4: 1.3-2.7 % of leg 1.)
    const bool ids16 = K <= 65535;
    std::vector<int32_t> blk_quad0(size_t(nblk) + 1, 0);
    std::vector<int32_t> dcols;                       // padded to 64 per block
    std::vector<uint32_t> abits;                      // [quad][lane][4 steps]
    // per block and wave: its rounds (64 ids each) and its lane groups' rows; laid out per unit below
    std::vector<std::vector<int32_t>> blk_sids((size_t(nblk) + size_t(kSub)) * 4);
    std::vector<int32_t> blk_gmeta((size_t(nblk) + size_t(kSub)) * 32 * 4 * 2, int32_t(0xFFFFFFFFu));
    std::fill(blk_gmeta.begin(), blk_gmeta.begin() + size_t(nblk) * 32 * 4 * 2, 0);
    std::vector<int64_t> cost(size_t(nblk), 0), blk_rem(size_t(nblk), 0);
    std::vector<int32_t> blk_ng(size_t(nblk), 0), blk_gslot(size_t(nblk), 0);   // split blocks: gather units, slot of gather unit 1
    size_t extra_gm = 0;
    int64_t covered = 0, steps_total = 0, r_nnz = 0;
    // The blocks are independent of each other: each is worked out on its own (what it adds to the plan's arrays lands in a
    // BlockOut), by a few threads on large graphs, and the results are put together in block order afterwards — the same
    // arrays, entry for entry, as the one-thread builder's (config 5: 18 ms of the graph's set-up -> 5).
    struct GatherOut { std::vector<int32_t> sids[4]; int32_t gmeta[32 * 4 * 2]; };
    struct BlockOut {
        std::vector<int32_t> dcols;                   // nq x 64, padded
        std::vector<uint32_t> abits;                  // nq x 64 x 4
        int32_t nq = 0, U = 0, ng = 0;
        int64_t covered = 0, rem = 0, cost = 0;
        std::vector<GatherOut> gus;
    };
    struct Scratch {
        std::vector<uint16_t> cnt;
        std::vector<int32_t> kpos, touched, set;
        std::vector<int32_t> rem[kFB];
        explicit Scratch(int64_t K) : cnt(size_t(K), 0), kpos(size_t(K), -1) {}
    };
    auto process_block = [&](int64_t b, Scratch& sc, BlockOut& out) {
        std::vector<uint16_t>& cnt = sc.cnt;
        std::vector<int32_t>& kpos = sc.kpos;
        std::vector<int32_t>& touched = sc.touched;
        std::vector<int32_t>& set = sc.set;
        std::vector<int32_t>* rem = sc.rem;
        const int64_t lo = b * kFB, hi = std::min<int64_t>(M, lo + kFB);
        touched.clear();
        set.clear();
        for (int32_t j = rowptr[lo]; j < rowptr[hi]; ++j)
            if (cnt[col[j]]++ == 0) touched.push_back(col[j]);
        if (by_quads) {
            std::stable_sort(touched.begin(), touched.end(), [&](int32_t x, int32_t y) { return cnt[x] > cnt[y]; });
            size_t n = 0;
            while (n < touched.size()) {
                const size_t e = std::min(touched.size(), n + 64);
                int64_t gain = 0;
                for (size_t i = n; i < e; ++i) gain += cnt[touched[i]];
                if (gain < pays) break;
                n = e;
            }
            for (size_t i = 0; i < n; ++i) { set.push_back(touched[i]); kpos[touched[i]] = 0; }
        } else {
            for (int32_t c : touched)
                if (cnt[c] >= thr) { set.push_back(c); kpos[c] = 0; }
        }
        {
            const int64_t steps = (int64_t)(set.size() + 15) / 16;
            bool keep = steps >= min_steps;
            if (!keep && g->tun.fuse_dens > 0 && steps >= 4) {      // a short set that is dense all the same
                int64_t cov = 0;
                for (int32_t c : set) cov += cnt[c];
                keep = cov >= g->tun.fuse_dens * steps;
            }
            if (!keep) {
                for (int32_t c : set) kpos[c] = -1;
                set.clear();
            }
        }
        // rows that would keep a long remainder go to the matrix cores whole
        for (int64_t a = lo; a < hi; ++a) {
            int32_t r = 0;
            for (int32_t j = rowptr[a]; j < rowptr[a + 1]; ++j) r += kpos[col[j]] < 0;
            if (r > kMaxRem)
                for (int32_t j = rowptr[a]; j < rowptr[a + 1]; ++j)
                    if (kpos[col[j]] < 0) { set.push_back(col[j]); kpos[col[j]] = 0; }
        }
        std::sort(set.begin(), set.end());
        const int32_t U = (int32_t)set.size();
        const int32_t nq = (U + 63) / 64;
        out.U = U;
        out.nq = nq;
        out.dcols.assign(size_t(nq) * 64, U ? set[0] : 0);         // padding: a real row, pattern bits zero
        out.abits.assign(size_t(nq) * 64 * 4, 0u);
        for (int32_t i = 0; i < U; ++i) {
            out.dcols[size_t(i)] = set[size_t(i)];
            kpos[set[size_t(i)]] = i;
        }
        const int nr = int(hi - lo);
        for (int rr = 0; rr < nr; ++rr) {
            const int64_t a = lo + rr;
            rem[rr].clear();
            for (int32_t j = rowptr[a]; j < rowptr[a + 1]; ++j) {
                const int32_t i = kpos[col[j]];
                if (i < 0) {
                    rem[rr].push_back(col[j]);
                } else {
                    // column i of the set: quad i / 64, step (i % 64) / 16, k = i % 16 = 8 h + jj;
                    // lane (h, m = row % 32), byte = 32-row tile, bit jj
                    const int kk = i & 15, s = (i & 63) >> 4;
                    const int lane = (kk >> 3) * 32 + (rr & 31);
                    out.abits[(size_t(i >> 6) * 64 + size_t(lane)) * 4 + size_t(s)] |= 1u << (8 * (rr >> 5) + (kk & 7));
                    ++out.covered;
                }
            }
            // (round 5: ordering a row's gathered ids by how many rows of the block share the column — siblings at the head
            // of both streams, the same few rounds of the workgroup — changed nothing: 4.84 against 4.82 ms at pl32768d32,
            // 18.1 against 17.8 at N = 65536; a sibling's second access hits the L1 or the L2 either way.  Not kept.)
            out.rem += (int64_t)rem[rr].size();
        }
        int order[kFB];
        std::iota(order, order + nr, 0);
        std::stable_sort(order, order + nr, [&](int x, int y) { return rem[x].size() > rem[y].size(); });
        // Gather units of the block (round 4).  A block whose set is cut into units (fuse_unit) or whose remainder
        // exceeds fuse_rows entries is SPLIT: its matrix-core units and its gather units (each owning every n-th
        // row of the descending remainder order) all publish raw sums, the last arriver adds and scales them.
        // An unsplit block is one unit doing both phases, as before.
        const int64_t unit_q0 = std::max<int64_t>(4, g->tun.fuse_unit);
        const bool may_split = unit_q0 < (int64_t(1) << 20);
        const int n_m = nq > 0 ? (int)std::max<int64_t>(1, (nq + unit_q0 - 1) / unit_q0) : 0;
        int n_g = 1;
        if (may_split) n_g = (int)std::min<int64_t>(32, std::max<int64_t>(1, (out.rem + g->tun.fuse_rows - 1) / g->tun.fuse_rows));
        const bool split = may_split && (n_m > 1 || n_g > 1);
        if (!split) n_g = 1;
        out.ng = split ? n_g : 0;
        out.gus.resize(size_t(n_g));
        for (int gu = 0; gu < n_g; ++gu) {
            GatherOut& go = out.gus[size_t(gu)];
            std::fill(go.gmeta, go.gmeta + 32 * 4 * 2, 0);
            // this gather unit's rows: every n_g-th of the descending order
            int prow[kFB], pn = 0;
            for (int i = gu; i < nr; i += n_g) prow[pn++] = order[i];
            // 32 lane groups x 4 rows: longest row first, to the group with the smallest total that has room
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
            // groups by total, dealt to the waves in turn: every wave gets the same mix
            int gorder[32];
            std::iota(gorder, gorder + 32, 0);
            std::stable_sort(gorder, gorder + 32, [&](int x, int y) { return grp_tot[x] > grp_tot[y]; });
            int64_t slots_block = 0;
            for (int w = 0; w < 4; ++w) {
                int64_t longest = 0;
                for (int gg = 0; gg < 8; ++gg) longest = std::max(longest, grp_tot[gorder[gg * 4 + w]]);
                const int rounds = (int)((longest + 7) / 8);
                std::vector<int32_t>& sids = go.sids[w];
                sids.assign(size_t(rounds) * 64, -1);
                for (int gg = 0; gg < 8; ++gg) {
                    const int gi = gorder[gg * 4 + w];
                    int32_t* gm = &go.gmeta[(size_t(w) * 8 + size_t(gg)) * 8];
                    int f = 0;
                    for (int k = 0; k < 4; ++k) {
                        if (k < grp_n[gi]) {
                            const int rr = grp_rows[gi][k];
                            for (int32_t id : rem[rr]) {
                                sids[size_t(f >> 3) * 64 + size_t(gg) * 8 + size_t(f & 7)] = id;
                                ++f;
                            }
                            // (an empty row never ends a stream slot: no end)
                            const uint32_t end = rem[rr].empty() ? 0xFFFFFFu : uint32_t(f);
                            gm[2 * k] = int32_t(end << 8 | uint32_t(rr));
                            const float scv = rowscale[size_t(lo + rr)];
                            memcpy(&gm[2 * k + 1], &scv, 4);
                        } else {
                            gm[2 * k] = int32_t(0xFFFFFFFFu);
                        }
                    }
                }
                slots_block += longest;
            }
            if (gu == 0) out.cost = int64_t((U + 15) / 16) * 24 + slots_block * 20 + 100;
        }
        for (int32_t c : touched) { cnt[c] = 0; kpos[c] = -1; }
    };
    std::vector<BlockOut> outs((size_t)nblk);
    {
        // (SIMRANK_BUILD_THREADS: a test aid — tools/host/host_fuzz.cpp builds the same plan on one thread and on several
        // and compares the arrays)
        const char* forced = std::getenv("SIMRANK_BUILD_THREADS");
        const int n_thr = forced ? std::max(1, std::atoi(forced))
                                 : (int)std::min<int64_t>(8, (rowptr[M] >= 100000 && nblk >= 8)
                                                                 ? std::min<int64_t>(nblk / 2, std::max<int64_t>(nblk / 8, rowptr[M] / 120000)) : 1);
        // (by blocks on sparse patterns, by entries on dense ones: 29 blocks of 34 000 entries each — the MovieLens-shaped
        // item side — are worth eight threads, not three)
        if (n_thr <= 1) {
            Scratch sc(K);
            for (int64_t b = 0; b < nblk; ++b) process_block(b, sc, outs[(size_t)b]);
        } else {
            // equal shares of the entries (the solver's order is ascending in row length: equal block counts would not balance)
            std::vector<std::thread> th;
            int64_t b0 = 0;
            for (int t = 0; t < n_thr; ++t) {
                int64_t b1 = nblk;
                if (t + 1 < n_thr) {
                    const int64_t target = int64_t(rowptr[M]) * (t + 1) / n_thr;
                    b1 = b0;
                    while (b1 < nblk && rowptr[std::min<int64_t>(M, b1 * kFB)] < target) ++b1;
                }
                th.emplace_back([&, b0, b1]() {
                    Scratch sc(K);
                    for (int64_t b = b0; b < b1; ++b) process_block(b, sc, outs[(size_t)b]);
                });
                b0 = b1;
            }
            for (std::thread& x : th) x.join();
        }
    }
    // put together in block order
    for (int64_t b = 0; b < nblk; ++b) {
        BlockOut& o = outs[(size_t)b];
        blk_quad0[size_t(b) + 1] = blk_quad0[size_t(b)] + o.nq;
        steps_total += (o.U + 15) / 16;
        dcols.insert(dcols.end(), o.dcols.begin(), o.dcols.end());
        abits.insert(abits.end(), o.abits.begin(), o.abits.end());
        covered += o.covered;
        r_nnz += o.rem;
        blk_rem[size_t(b)] = o.rem;
        blk_ng[size_t(b)] = o.ng;
        cost[size_t(b)] = o.cost;
        for (size_t gu = 0; gu < o.gus.size(); ++gu) {
            // where this unit's row records and streams live: gather unit 0 of a block keeps the block's own slot
            const size_t slot = gu == 0 ? size_t(b) : size_t(nblk) + size_t(kSub) + extra_gm++;
            if (gu > 0) {
                blk_gmeta.resize((slot + 1) * 32 * 4 * 2, 0);
                blk_sids.resize((slot + 1) * 4);
            }
            if (gu == 1) blk_gslot[size_t(b)] = (int32_t)slot;      // (gather units 1 .. of a block own consecutive slots)
            std::copy(o.gus[gu].gmeta, o.gus[gu].gmeta + 32 * 4 * 2, blk_gmeta.begin() + slot * 32 * 4 * 2);
            for (int w = 0; w < 4; ++w) blk_sids[slot * 4 + size_t(w)].swap(o.gus[gu].sids[w]);
        }
        o = BlockOut();
    }
    // Units (one workgroup per unit and panel).  A block whose set has more than fuse_unit quads is cut
    // into units of about that many (its partial sums meet in memory, the last arriver finishes the
    // block).  Blocks WITHOUT a set are light — a workgroup's fixed latencies (unit record -> ids ->
    // first gathers ... barrier -> store) would outweigh their gathers — so up to fuse_group consecutive
    // ones share a unit, their rounds one stream per wave.  Launch order by cost, heaviest first.
    const int64_t unit_q = std::max<int64_t>(4, g->tun.fuse_unit);
    struct Unit { int32_t b0, nsub, q0, nq, k, nb; int64_t cost; int32_t gslot, split; };
    std::vector<Unit> ulist;
    // (a panel should offer an XCD more workgroups than it has slots — 32 CUs x 4 — or several panels are
    // in flight at once and share its L2: with too few units the grouping is halved; N = 8192, 64 blocks:
    // groups of four left 16 units per panel and eight 1 MiB slices in a 4 MiB L2, leg 1 +8 %)
    for (int64_t group = std::min<int64_t>(kSub, std::max<int64_t>(1, g->tun.fuse_group));; group /= 2) {
        ulist.clear();
        for (int64_t b = 0; b < nblk;) {
            const int32_t q0 = blk_quad0[size_t(b)], nqb = blk_quad0[size_t(b) + 1] - q0;
            if (blk_ng[size_t(b)] > 0) {
                // a split block: matrix-core units (no rows) and gather units (no columns of the set); the slot
                // behind the last block holds no rows at all
                const int32_t n_m = nqb > 0 ? (int32_t)std::max<int64_t>(1, (nqb + unit_q - 1) / unit_q) : 0;
                const int32_t n_g = blk_ng[size_t(b)], nb = n_m + n_g;
                for (int32_t k = 0; k < n_m; ++k) {
                    const int32_t lo = (int32_t)(int64_t(nqb) * k / n_m), hi = (int32_t)(int64_t(nqb) * (k + 1) / n_m);
                    ulist.push_back({(int32_t)b, 1, q0 + lo, hi - lo, k, nb, cost[size_t(b)], (int32_t)nblk, 1});
                }
                for (int32_t k = 0; k < n_g; ++k)
                    ulist.push_back({(int32_t)b, 1, q0, 0, n_m + k, nb, cost[size_t(b)],
                                     k == 0 ? (int32_t)b : blk_gslot[size_t(b)] + (k - 1), 1});
                ++b;
            } else if (nqb > 0) {
                ulist.push_back({(int32_t)b, 1, q0, nqb, 0, 1, cost[size_t(b)], (int32_t)b, 0});
                ++b;
            } else {
                // (at most kGroupEntries gathered entries per unit: blocks of an Erdos-Renyi pattern — every
                // row its 32 entries, no set — are work enough alone; four of them in one workgroup left half
                // of an XCD's workgroup slots empty: leg 1 +47 % at N = 32768, mean degree 32)
                int64_t e = b + 1, c = cost[size_t(b)], entries = blk_rem[size_t(b)];
                while (e < nblk && e - b < group && blk_quad0[size_t(e) + 1] == blk_quad0[size_t(e)] && blk_ng[size_t(e)] == 0 &&
                       entries + blk_rem[size_t(e)] <= kGroupEntries) {
                    entries += blk_rem[size_t(e)];
                    c += cost[size_t(e++)];
                }
                ulist.push_back({(int32_t)b, (int32_t)(e - b), q0, 0, 0, 1, c, (int32_t)b, 0});
                b = e;
            }
        }
        if (group <= 1 || (int64_t)ulist.size() >= kMinUnits) break;
    }
    std::stable_sort(ulist.begin(), ulist.end(), [](const Unit& x, const Unit& y) {
        return x.cost != y.cost ? x.cost > y.cost : (x.b0 != y.b0 ? x.b0 < y.b0 : x.k < y.k);
    });
    if (g->tun.fuse_order > 0 && ulist.size() > 8) {
        // units with a matrix-core phase spread evenly over the first 1 / fuse_order of the launch order
        // instead of all in front: a CU then holds matrix-core work and gather work at the same time
        // (the two phases use different pipes), not one kind after the other.  A block's units stay together.
        std::vector<Unit> heavy, light, mixed;
        for (const Unit& u : ulist) ((u.nq > 0 || u.split) ? heavy : light).push_back(u);
        const size_t span = std::max(heavy.size(), std::min(ulist.size(), ulist.size() / size_t(g->tun.fuse_order)));
        size_t hi = 0, li = 0;
        for (size_t pos = 0; pos < ulist.size(); ++pos) {
            const bool want_heavy = hi < heavy.size() && (li >= light.size() || hi * span <= pos * heavy.size());
            if (want_heavy) mixed.push_back(heavy[hi++]);
            else mixed.push_back(light[li++]);
        }
        ulist.swap(mixed);
    }
    std::vector<int32_t> units(ulist.size() * 32, 0), sids;
    blk_gmeta.resize(blk_gmeta.size() + size_t(kSub) * 32 * 4 * 2, int32_t(0xFFFFFFFFu));   // (a unit's loads may run past its blocks)
    sids.reserve(size_t(r_nnz) + ulist.size() * 512);
    int32_t n_pslots = 0, n_cslots = 0;
    std::vector<int32_t> blk_pslot(size_t(nblk), -1), blk_cslot(size_t(nblk), -1);
    for (size_t i = 0; i < ulist.size(); ++i) {
        const Unit& u = ulist[i];
        int32_t* rec = &units[i * 32];
        rec[0] = u.b0; rec[1] = u.q0; rec[2] = u.nq; rec[3] = u.k; rec[4] = u.nb;
        rec[5] = -1; rec[6] = -1;
        rec[7] = u.nq > 0 ? 1 : 0;
        rec[8] = u.nsub;
        rec[29] = u.gslot;                          // where the unit's row records live (an unsplit block: its own slot)
        rec[30] = u.split;
        if (u.nb > 1) {
            // the units of a block own consecutive slots, handed out when its first unit comes by — wherever the launch
            // order (fuse_order) puts the others
            if (blk_pslot[size_t(u.b0)] < 0) {
                blk_pslot[size_t(u.b0)] = n_pslots;
                blk_cslot[size_t(u.b0)] = n_cslots;
                n_pslots += u.nb;
                ++n_cslots;
            }
            rec[5] = blk_pslot[size_t(u.b0)] + u.k;
            rec[6] = blk_cslot[size_t(u.b0)];
        }
        for (int w = 0; w < 4; ++w) {
            int32_t* wm = rec + 9 + w * 5;
            wm[0] = (int32_t)(sids.size() / 64);
            int32_t rounds = 0;
            for (int sb = 0; sb < kSub; ++sb) {
                if (sb < u.nsub) {
                    const std::vector<int32_t>& bs = blk_sids[size_t(u.gslot + sb) * 4 + size_t(w)];
                    sids.insert(sids.end(), bs.begin(), bs.end());
                    rounds += (int32_t)(bs.size() / 64);
                }
                wm[1 + sb] = rounds;
            }
        }
    }

    if (const char* dump = std::getenv("SIMRANK_DUMP_FUSED_PLAN")) {
        // Measurement aid (tools/micro/gather_depth.hip): the unit records and the id streams of the gather phase as the
        // launch reads them, so that a stand-alone kernel can replay the REAL access pattern.  Header of eight int64, then
        // units[n_units][32], sids[n_rounds][64] (int32, -1 = empty slot), dcols[n_quads][64].
        if (FILE* f = std::fopen(dump, "wb")) {
            const int64_t hdr[8] = {0x53524450, M, K, (int64_t)(units.size() / 32), (int64_t)(sids.size() / 64),
                                    (int64_t)(dcols.size() / 64), r_nnz, covered};
            std::fwrite(hdr, sizeof(int64_t), 8, f);
            std::fwrite(units.data(), sizeof(int32_t), units.size(), f);
            std::fwrite(sids.data(), sizeof(int32_t), sids.size(), f);
            std::fwrite(dcols.data(), sizeof(int32_t), dcols.size(), f);
            std::fclose(f);
        }
    }
    simrank_fused_plan* pl = new simrank_fused_plan;
    pl->n_units = (int32_t)(units.size() / 32);
    pl->n_blocks = (int32_t)nblk;
    pl->n_pslots = n_pslots;
    pl->n_cslots = n_cslots;
    pl->n_quads = blk_quad0[size_t(nblk)];
    pl->n_steps = steps_total;
    pl->nnz_covered = covered;
    pl->r_nnz = r_nnz;
    pl->ids16 = ids16 ? 1 : 0;
    // (one quad of padding behind the last set: the kernel requests the first ids and pattern bits of a unit's share of its
    // set before it knows whether the unit has one — a unit of the last blocks without a set reads here)
    dcols.resize(dcols.size() + 64, 0);
    abits.resize(abits.size() + 64 * 4, 0u);
    int rc = upload_vec(&pl->units, units);
    if (!rc) {
        if (ids16) {
            std::vector<uint16_t> d16(dcols.begin(), dcols.end());
            rc = upload_vec(&pl->dcols16, d16);
        } else {
            rc = upload_vec(&pl->dcols32, dcols);
        }
    }
    if (!rc) rc = upload_vec(reinterpret_cast<uint32_t**>(&pl->abits), abits);
    if (!rc) rc = upload_vec(reinterpret_cast<int32_t**>(&pl->gmeta), blk_gmeta);
    if (!rc) {
        if (ids16) {
            std::vector<uint16_t> s16(sids.size());
            for (size_t i = 0; i < sids.size(); ++i) s16[i] = sids[i] < 0 ? uint16_t(0xFFFF) : uint16_t(sids[i]);
            rc = upload_vec(&pl->sids16, s16);
        } else {
            rc = upload_vec(&pl->sids32, sids);
        }
    }
#ifndef SIMRANK_HOST_ONLY
    if (!rc && n_pslots > 0) {
        // partial sums and tickets of the split blocks for the widest operand a solver hands over (S of the other
        // node set: max(M, K) columns); a wider one grows them at its first launch
        const int32_t panels = (int32_t)((std::max<int64_t>(M, K) + 31) / 32);
        hipError_t e = pool_hip_alloc((void**)&pl->partials, size_t(n_pslots) * panels * 32 * kFB * sizeof(float));
        if (e == hipSuccess) e = pool_hip_alloc((void**)&pl->tickets, size_t(n_cslots) * panels * sizeof(int32_t));
        if (e == hipSuccess) e = hipMemset(pl->tickets, 0, size_t(n_cslots) * panels * sizeof(int32_t));
        if (e != hipSuccess) {
            set_error("one-launch plan: partial sums of %d split blocks: %s", n_cslots, hipGetErrorString(e));
            (void)hipGetLastError();
            rc = e == hipErrorOutOfMemory ? SIMRANK_ERR_ALLOC : SIMRANK_ERR_HIP;
        } else {
            pl->cap_panels = panels;
        }
    }
#endif
    if (rc) {
        free_fused_plan(pl);
        return rc;
    }
    g->fused = pl;
    return SIMRANK_OK;
}

static int launch_fused(const simrank_graph* g, FusedArgs& a, hipStream_t st, bool sym = false) {
    simrank_fused_plan* pl = g->fused;
    // (the partial sums and tickets of the split blocks belong to the GRAPH: launches on one graph must be stream-ordered —
    // one solver per graph object, include/simrank_hip.h — and two host threads must not launch on it at once: this lock
    // covers the part that may re-allocate them; advisor, round 4)
    static std::mutex launch_mutex;
    std::lock_guard<std::mutex> launch_lock(launch_mutex);
    a.M = g->n_rows;
    a.n_panels = int32_t((a.L + 31) / 32);
    a.n_units = pl->n_units;
    a.nt = (int32_t)(g->tun.stream_nt ? g->tun.fuse_store : 0);
    a.probe = (int32_t)g->tun.probe_flags;
    a.meta_nt = (int32_t)g->tun.fuse_meta_nt;
    a.idx_mask = (int32_t)g->tun.probe_mask;
    a.units = pl->units;
    a.rowscale = g->rowscale;
    if (pl->n_pslots > 0 && a.n_panels > pl->cap_panels) {
        // (an operand wider than the plan was built for — not what a solver does: grow once)
        SR_HIP(hipStreamSynchronize(st));                  // an earlier launch may still use the old ones
        (void)pool_free(pl->partials); (void)pool_free(pl->tickets);
        pl->partials = nullptr; pl->tickets = nullptr; pl->cap_panels = 0;
        SR_HIP(pool_hip_alloc((void**)&pl->partials, size_t(pl->n_pslots) * a.n_panels * 32 * kFB * sizeof(float)));
        SR_HIP(pool_hip_alloc((void**)&pl->tickets, size_t(pl->n_cslots) * a.n_panels * sizeof(int32_t)));
        SR_HIP(hipMemsetAsync(pl->tickets, 0, size_t(pl->n_cslots) * a.n_panels * sizeof(int32_t), st));
        pl->cap_panels = a.n_panels;
    }
    // (tickets are indexed with the launch's own panel count: a narrower launch uses a prefix, all zero)
    a.partials = pl->partials; a.tickets = pl->tickets;
    a.n_pslots = pl->n_pslots; a.n_cslots = pl->n_cslots;
    a.dcols16 = pl->dcols16; a.dcols32 = pl->dcols32; a.abits = pl->abits;
    a.gmeta = pl->gmeta; a.sids16 = pl->sids16; a.sids32 = pl->sids32;
    const int64_t grid = int64_t((a.n_panels + 7) / 8) * 8 * a.n_units;
    SR_REQUIRE(grid > 0 && grid < (int64_t(1) << 31), "grid of %lld blocks", (long long)grid);
#ifdef SIMRANK_FUSED_STAMPS
    static const unsigned lds_pad = getenv("SIMRANK_FUSED_PAD") ? (unsigned)atoi(getenv("SIMRANK_FUSED_PAD")) : 0u;
    {
        const char* e = getenv("SIMRANK_FST_BASE");
        const unsigned base = e ? (unsigned)atoll(e) : 0u;
        SR_HIP(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_fst_base), &base, sizeof(base), 0, hipMemcpyHostToDevice, st));
    }
#else
    const unsigned lds_pad = 0;
#endif
#ifdef SIMRANK_HOST_ONLY
    (void)lds_pad;
    SR_REQUIRE(false, "host-only build: no kernels");
#else
    if (sym) {
        if (pl->ids16)
            hipLaunchKernelGGL((fused_trans_kernel<true, true>), dim3((unsigned)grid), dim3(256), lds_pad, st, a);
        else
            hipLaunchKernelGGL((fused_trans_kernel<false, true>), dim3((unsigned)grid), dim3(256), lds_pad, st, a);
    } else if (pl->ids16) {
        hipLaunchKernelGGL((fused_trans_kernel<true, false>), dim3((unsigned)grid), dim3(256), lds_pad, st, a);
    } else {
        hipLaunchKernelGGL((fused_trans_kernel<false, false>), dim3((unsigned)grid), dim3(256), lds_pad, st, a);
    }
#endif
    SR_HIP(hipGetLastError());
    return SIMRANK_OK;
}

// Tt (panel-blocked, y_rows_pad rows per panel) = (diag(rowscale) . A . X)^T, X panel-blocked
int launch_fused_trans(const simrank_graph* g, const float* X, int64_t x_rows_pad, int64_t L, float* Y,
                       int64_t y_rows_pad, hipStream_t st) {
    SR_REQUIRE(g->fused, "graph has no fused plan");
    SR_REQUIRE(aligned16(X) && aligned16(Y), "fused leg needs 16-byte aligned operands");
    SR_REQUIRE(x_rows_pad >= g->n_cols && (x_rows_pad + 1) * 128 < (int64_t(1) << 31) && x_rows_pad < (int64_t(1) << 24) - 1,
               "fused leg: operand of %lld rows per panel", (long long)x_rows_pad);
    FusedArgs a{};
    a.X = X; a.Y = Y;
    a.x_rows_pad = x_rows_pad; a.y_rows_pad = y_rows_pad;
    a.L = L;
    a.x_panel_stride = x_rows_pad * 32;
    a.x_pitch = 128;
    a.x_bytes = int32_t(x_rows_pad * 128);
    a.x_sentinel = (int32_t)x_rows_pad;
    return launch_fused(g, a, st);
}

// Leg 2 of a symmetric panel-blocked update in one launch (SYM): Y (M x M, M = n_rows(g) = L) = epilogue(diag(rowscale) . A . X),
// X panel-blocked with K = n_cols(g) rows; the epilogue's operands (counts, prior, previous iterate) share Y's layout.
// false from fused_sym_applies: the two-launch leg (dense_tiles + gather3<kSym>) stays.
bool fused_sym_applies(const simrank_graph* g, int64_t x_rows_pad, int64_t L, int64_t y_rows_pad) {
    const simrank_fused_plan* pl = g->fused;
    if (!pl || !g->tun.fuse || g->tun.fuse_sym == 0 || g->tun.dense_terms != 3) return false;
    if (L != g->n_rows || g->n_cols > g->tun.fuse_max_rows) return false;
    if ((x_rows_pad + 1) * 128 >= (int64_t(1) << 31) || (y_rows_pad + 1) * 128 >= (int64_t(1) << 31) ||
        x_rows_pad >= (int64_t(1) << 24) - 1 || y_rows_pad >= (int64_t(1) << 24) - 1)
        return false;
    if (g->tun.fuse_sym < 0) return 2 * pl->nnz_covered >= g->nnz;      // (the dense sets hold at least half of the entries)
    return true;
}

int launch_fused_sym(const simrank_graph* g, const float* X, int64_t x_rows_pad, int64_t L, float* Y, int64_t y_rows_pad,
                     float coef, float lbd, double eps, const uint8_t* ev, const float* ap, const float* prev,
                     unsigned long long* n_changed, int32_t set_diag, int32_t count_any, hipStream_t st) {
    SR_REQUIRE(fused_sym_applies(g, x_rows_pad, L, y_rows_pad), "fused leg 2 does not apply");
    SR_REQUIRE(aligned16(X) && aligned16(Y) && (!ap || aligned16(ap)) && (!prev || aligned16(prev)) &&
                   (!ev || reinterpret_cast<uintptr_t>(ev) % 4 == 0),
               "fused leg 2 needs 16-byte aligned operands");
    SR_REQUIRE(x_rows_pad >= g->n_cols && y_rows_pad >= g->n_rows && (!prev || n_changed), "fused leg 2: bad operands");
    FusedArgs a{};
    a.X = X; a.Y = Y;
    a.x_rows_pad = x_rows_pad; a.y_rows_pad = y_rows_pad;
    a.L = L;
    a.x_panel_stride = x_rows_pad * 32;
    a.x_pitch = 128;
    a.x_bytes = int32_t(x_rows_pad * 128);
    a.x_sentinel = (int32_t)x_rows_pad;
    a.coef = coef; a.lbd = lbd; a.eps = eps;
    a.ev = ev; a.ap = ap; a.prev = prev;
    a.n_changed = n_changed;
    a.set_diag = set_diag; a.count_any = count_any;
    return launch_fused(g, a, st, true);
}

// The same leg on a ROW-MAJOR operand (K x L, leading dimension ldx) with the transposed result in the chunked
// layout of simrank_spmm(transpose_out = 1, t_block, t_pad): what one rank of a column-sharded update runs
// (its operand is S[:, C_g]; chunk h of the result goes to rank h).  false: the shapes do not fit this path.
bool fused_rowmajor_fits(const simrank_graph* g, const float* X, int64_t ldx, int64_t L, const float* Y, int64_t t_block,
                         int64_t t_pad) {
    const int64_t K = g->n_cols, M = g->n_rows;
    const int64_t tb = (t_block <= 0 || t_block > M) ? M : t_block;
    return g->fused && aligned16(X) && aligned16(Y) && ldx % 4 == 0 && ldx * 4 < (int64_t(1) << 24) && ldx >= 32 &&
           (K + 1) * ldx * 4 < (int64_t(1) << 32) && K < (int64_t(1) << 24) - 1 && L > 0 &&
           (tb % kFB == 0 || tb == M) && (tb + t_pad) % 4 == 0 && (M - (M - 1) / tb * tb + t_pad) % 4 == 0;
}

int launch_fused_trans_rowmajor(const simrank_graph* g, const float* X, int64_t ldx, int64_t L, float* Y, int64_t t_block,
                                int64_t t_pad, hipStream_t st) {
    SR_REQUIRE(fused_rowmajor_fits(g, X, ldx, L, Y, t_block, t_pad), "fused leg: row-major operand does not fit");
    FusedArgs a{};
    a.X = X; a.Y = Y;
    a.L = L;
    a.x_panel_stride = 32;
    a.x_pitch = uint32_t(ldx * 4);
    // (to the last column of the last row: nothing beyond is read.  The descriptor's byte count and the offsets
    // id x pitch are UNSIGNED 32-bit quantities: a rank's block may span up to 4 GiB — config 5 on eight ranks is 2.2)
    a.x_bytes = int32_t(uint32_t((g->n_cols - 1) * ldx * 4 + L * 4));
    a.x_sentinel = (int32_t)g->n_cols;
    a.y_chunked = 1;
    a.t_block = (t_block <= 0 || t_block > g->n_rows) ? g->n_rows : t_block;
    a.t_pad = t_pad;
    return launch_fused(g, a, st);
}

}  // namespace simrank

using namespace simrank;

extern "C" {

#ifdef SIMRANK_FUSED_STAMPS
__attribute__((visibility("default"))) int simrank_read_fused_stamps(unsigned long long* out, int64_t n_wg) {
    SR_HIP(hipDeviceSynchronize());
    SR_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(simrank::g_fst), size_t(std::min<int64_t>(n_wg, simrank::kFstCap)) * 8 * sizeof(unsigned long long)));
    return SIMRANK_OK;
}
#endif

int simrank_graph_fused_stats(const simrank_graph* g, int64_t* n_steps, int64_t* nnz_covered,
                              int64_t* nnz_remainder) {
    SR_REQUIRE(g, "graph is NULL");
    const simrank_fused_plan* pl = g->fused;
    if (n_steps) *n_steps = pl ? pl->n_steps : 0;
    if (nnz_covered) *nnz_covered = pl ? pl->nnz_covered : 0;
    if (nnz_remainder) *nnz_remainder = pl ? pl->r_nnz : g->nnz;
    return SIMRANK_OK;
}

}  // extern "C"
