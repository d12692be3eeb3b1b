// Both legs of the single-rank update on matrices STORED IN FP16 (gfx950, round 3): BASELINE.json config 5's
// reduced-precision mode, `fit(storage_precision="fp16")`.  Never the default, outside the parity bar.
//
//   leg 1   Tt = (diag(rowscale) . A . S)^T                                  first `.dot` of SimRank.py:361
//   leg 2   S' = epilogue(diag(rowscale) . A . Tt), upper triangle + mirror   second `.dot`, :315-316, :453, :362
//
// Why this and not an fp16 operand for the matrix cores only (blockdense.hip's one-term mode, which buys
// nothing): both legs are bound by the NUMBER of 128-byte lines the texture path moves (fused.hip) and by
// the HBM/L2 bytes behind them.  An fp16 line holds 64 columns instead of 32, so the same gathers, id
// streams and stores serve twice the columns: S, Tt and the previous iterate are fp16, PANEL-BLOCKED with
// 64-column panels (element (r, c) at ((c >> 6) * rows_pad + r) * 64 + (c & 63): a row segment is still one
// 128-byte line), sums are f32, the epilogue is f32, and a value is rounded to fp16 (nearest even) once,
// when it is stored.  SCALE: the matrices hold value x 2^k (the solver: 2^14) because most similarities of a
// large sparse graph lie below fp16's normal range; both legs are linear, so only the diagonal, the prior,
// eps and the hand-back see the scale.  CONVERGENCE (SimRank.py:74): the previous iterate exists only rounded,
// so an element counts as moved when |new (before rounding) - old (stored)| > eps + half the fp16 spacing at
// the old value, i.e. by more than eps beyond what the old value's rounding explains.  Below 1/8 that spacing
// is under eps / 4 at the default eps and scale and the test is the reference's; for the few larger values
// it keeps a drift of less than eps per iteration from being counted each time it crosses a rounding
// boundary (comparing rounded with rounded ran 52 iterations where the reference stops at 20).
//
// One kernel for both legs, built on the one-launch plan of fused.hip (same units, pattern bits, balanced
// gather streams): a workgroup owns a 128-row block x one 64-column panel.
//   1. MFMA phase: wave (column half, K half) multiplies the block's dense set on
//      v_mfma_f32_32x32x16_f16 — 0/1 pattern x the fp16 operand = exact products, ONE term instead of the
//      three bf16 terms of the f32 path; the operand's 64-byte half segments are gathered 16 rows per
//      instruction and transposed into B fragments through a 1.25 KiB LDS buffer per wave.
//   2. The two K halves are added in order into an f32 LDS tile (64 columns x 128 rows).
//   3. Gather phase: as fused.hip, 8 columns per lane (16 bytes = 8 halves), f32 running sums.
//   4. leg 1: the tile goes out transposed, each wave 4 KiB contiguous of Tt.
//      leg 2: epilogue per row (evidence counts and prior in their f32-era 32-column panel layout), the
//      rounded value is compared, stored for a <= c, written back to the tile and mirrored for a < c.
//      Blocks that lie wholly left of the diagonal do nothing.
// Deterministic: fixed summation order everywhere.
#include <cmath>

#include "common.h"

namespace simrank {

constexpr int kHR = 68;           // floats per ROW of the LDS tile (64 columns + 4: rows stay 16-byte aligned)
constexpr int kBRow = 40;         // halves per operand row of a wave's B buffer (64 bytes + 16: the two K groups
                                  // of a fragment read from different banks)

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16h __attribute__((ext_vector_type(16)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

struct HalfArgs {
    const uint16_t* X;
    uint16_t* Y;
    int64_t x_rows_pad, y_rows_pad;
    int64_t L, M;
    int32_t n_panels, n_units;
    int32_t x_sentinel;
    const int32_t* units;
    const uint16_t* dcols16;
    const int32_t* dcols32;
    const uint4* abits;
    const int2* gmeta;
    const uint16_t* sids16;
    const int32_t* sids32;
    // leg 2
    float coef, lbd;
    float scale;               // stored value = true value x scale (a power of two): diagonal, prior and eps follow it
    double eps;
    const uint8_t* ev;         // counts, 32-column panels of ev_rows_pad rows
    int64_t ev_rows_pad;
    const float* ap;           // prior, 32-column panels of ap_rows_pad rows
    int64_t ap_rows_pad;
    const uint16_t* prev;      // previous iterate, fp16, 64-column panels of prev_rows_pad rows
    int64_t prev_rows_pad;
    unsigned long long* n_changed;
    int32_t set_diag, count_any;
    int32_t full;              // leg 2 of a SHARDED update: every element of the n_rows x L column block (no triangle, no
    int32_t col0;              // mirror image); col0 = global index of the block's column 0 (diagonal: row == col0 + column)
};

#ifndef SIMRANK_HOST_ONLY
__device__ __forceinline__ v4u ld16(__amdgpu_buffer_rsrc_t srd, int id, uint32_t boff) {
    return __builtin_amdgcn_raw_buffer_load_b128(srd, int(__umul24(uint32_t(id), 128u) + boff), 0, 0);
}

__device__ __forceinline__ void wave_lds_order_h() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ void add8(float (&c)[8], const v4u& v) {
    // (the whole vector is cast: __builtin_bit_cast of ONE ELEMENT of a vector reference reads element 0 for
    // every element with this compiler)
    const v4u w = v;
    const f16x8 h = __builtin_bit_cast(f16x8, w);
#pragma unroll
    for (int i = 0; i < 8; ++i) c[i] += float(h[i]);
}

__device__ __forceinline__ uint32_t pack2(float lo, float hi) {      // two f32 -> two fp16, nearest even
    f16x2 h;
    h.x = _Float16(lo);
    h.y = _Float16(hi);
    return __builtin_bit_cast(uint32_t, h);
}

__device__ __forceinline__ float half_bits_to_float(uint32_t bits16) {
    return float(__builtin_bit_cast(_Float16, uint16_t(bits16)));
}

// SYM: leg 2 (epilogue, upper triangle + mirror); otherwise leg 1 (transposed store)
// PRIOR: leg 2 with a prior matrix (two rows of a lane group per batch of epilogue loads instead of four)
#ifndef SIMRANK_HALF_LB1
#define SIMRANK_HALF_LB1 4      // waves per SIMD the register allocation of leg 1 aims at
#endif
#ifndef SIMRANK_HALF_RB
#define SIMRANK_HALF_RB 4       // rows of a lane group whose epilogue operands are loaded together (leg 2)
#endif
#ifndef SIMRANK_HALF_LB2
#define SIMRANK_HALF_LB2 3      // ... of leg 2
#endif
// FULL: leg 2 of one rank of a sharded update — the whole column block, each element once, no mirror image
template <bool IDS16, bool SYM, bool PRIOR, bool FULL = false>
__global__ __launch_bounds__(256, SYM ? SIMRANK_HALF_LB2 : SIMRANK_HALF_LB1) void half_leg_kernel(const HalfArgs p) {
    // [row][column] of the block's result: a lane group finishes a row with two 16-byte writes per lane; the
    // strided reads are left to the transposed store, which runs with all lanes
    __shared__ __attribute__((aligned(16))) float tile[kFB * kHR];
    // the MFMA phase's buffers live in the tile's memory (the tile is first written when that phase has ended):
    // 37.8 KiB per workgroup = four workgroups per CU
    uint4* const lut = reinterpret_cast<uint4*>(tile);                          // pattern byte -> 8 fp16 (0 / 1.0), 4 KiB
    uint16_t* const bbuf_all = reinterpret_cast<uint16_t*>(tile + 1024);        // per wave: 16 operand half segments
    static_assert(1024 * 4 + 4 * 16 * kBRow * 2 <= kFB * kHR * 4, "MFMA buffers inside the tile");
    __shared__ __attribute__((aligned(16))) int2 gm_lds[kSub * 4 * 8 * 4];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t bid = blockIdx.x;
    const uint32_t local = bid >> 3;
    const int panel = int(local / uint32_t(p.n_units)) * 8 + int(bid & 7);
    if (panel >= p.n_panels) return;
    const uint32_t unit = local % uint32_t(p.n_units);
    const int32_t* un = p.units + size_t(unit) * 32;
    const int b0 = un[0];
    const int64_t c0 = int64_t(panel) * 64;
    const int g = lane >> 3, q = lane & 7, gbase = lane & ~7;
    const uint32_t qoff = uint32_t(q) * 16u;
    int n_sub = un[8];
    if constexpr (SYM && !FULL) {
        // blocks whose first row lies right of the panel's last column have nothing in the upper triangle
        const int64_t last_c = (p.L < c0 + 64 ? p.L : c0 + 64) - 1;
        int k = 0;
        while (k < n_sub && int64_t(b0 + k) * kFB <= last_c) ++k;
        n_sub = k;
        if (n_sub == 0) return;
    }

    const uint16_t* xbase = p.X + (int64_t(panel) * p.x_rows_pad) * 64;
    const __amdgpu_buffer_rsrc_t srd =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(xbase), 0, int(p.x_rows_pad * 128), 0x00020000);
    const int sent = p.x_sentinel;

    const int quad0 = un[1];
    const int nq = un[2];
    const bool has_set = un[7] != 0;

    const int32_t* wm = un + 9 + wave * 5;
    const int round0 = wm[0];
    const int e1 = wm[1], e2 = wm[2], e3 = wm[3], e4 = wm[4];
    const int n_rounds = n_sub == 1 ? e1 : n_sub == 2 ? e2 : n_sub == 3 ? e3 : e4;
    // round r of the wave's id stream, as stored (the marker becomes the out-of-range id where it is used: no
    // arithmetic on a loaded value before then, so the load can stay in flight)
    auto ld_raw = [&](int r) -> int {
        const size_t at = (size_t(round0) + size_t(min(r, max(n_rounds - 1, 0)))) * 64 + lane;
        if constexpr (IDS16) return int(p.sids16[at]);
        else return p.sids32[at];
    };
    auto to_id = [&](int raw, bool ok) -> int {
        if constexpr (IDS16) return (raw == 0xFFFF || !ok) ? sent : raw;
        else return (raw < 0 || !ok) ? sent : raw;
    };
    // (round 5, as fused.hip / spmm.hip) everything the prologue reads is REQUESTED first and used afterwards: the row
    // records through plain global loads of a clamped index (the select against a stack slot compiled to flat loads
    // and scratch), the ids of the first two rounds side by side, the striped counter of the short-circuit test through an
    // unconditional buffer load past the L1 (a zero-byte descriptor when there is nothing to watch) instead of a
    // generic-pointer atomic load with its wait right behind it
    int iv0 = 0, iv1 = 0;
    unsigned long long seen = 0;
    const unsigned slot = ((blockIdx.x * 4u + unsigned(wave)) * 7u) % SIMRANK_CHANGED_SLOTS;
    {
        const int sbA = q >> 2, sbB = 2 + (q >> 2);
        const int2* src = p.gmeta + ((size_t(b0) * 4 + wave) * 8 + g) * 4 + (q & 3);
        int2* dst = gm_lds + (wave * 8 + g) * 4 + (q & 3);
        int2 a = src[size_t(min(sbA, n_sub - 1)) * 4 * 8 * 4];
        int2 c = src[size_t(min(sbB, n_sub - 1)) * 4 * 8 * 4];
        if (n_rounds > 0) {
            iv0 = ld_raw(0);
            iv1 = ld_raw(1);
        }
        if constexpr (SYM) {
            typedef unsigned v2u __attribute__((ext_vector_type(2)));
            const bool watch = p.prev && p.count_any;
            const __amdgpu_buffer_rsrc_t csrd = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<unsigned long long*>(p.n_changed), 0, watch ? SIMRANK_CHANGED_SLOTS * 8 : 0, 0x00020000);
            const v2u w = __builtin_amdgcn_raw_buffer_load_b64(csrd, int(slot * 8u), 0, 16);
            seen = (unsigned long long)w.x | ((unsigned long long)w.y << 32);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (sbA >= n_sub) a = make_int2(int(0xFFFFFFFFu), 0);
        if (sbB >= n_sub) c = make_int2(int(0xFFFFFFFFu), 0);
        dst[sbA * 4 * 8 * 4] = a;
        dst[sbB * 4 * 8 * 4] = c;
    }

    // ---------------------------------------------------------------- 1. MFMA phase
    if (nq > 0) {
        {   // entry e, dword d: low half = bit 2d, high half = bit 2d + 1 (fp16 1.0 = 0x3C00)
            const unsigned e = threadIdx.x;
            uint4 v;
            v.x = ((e >> 0) & 1u) * 0x3C00u | ((e >> 1) & 1u) * 0x3C000000u;
            v.y = ((e >> 2) & 1u) * 0x3C00u | ((e >> 3) & 1u) * 0x3C000000u;
            v.z = ((e >> 4) & 1u) * 0x3C00u | ((e >> 5) & 1u) * 0x3C000000u;
            v.w = ((e >> 6) & 1u) * 0x3C00u | ((e >> 7) & 1u) * 0x3C000000u;
            lut[e] = v;
        }
        __syncthreads();
        f32x16h acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
        const int ch = wave & 1, kp = wave >> 1;               // column half of the panel, half of the set's quads
        const int per = (nq + 1) >> 1;
        const int q_lo = kp * per, q_hi = min(nq, q_lo + per);
        if (q_lo < q_hi) {
            // software pipeline as fused.hip: every iteration issues the same loads in the same order
            uint16_t* bbuf = bbuf_all + wave * (16 * kBRow);
            const int n = lane & 31, h = lane >> 5;
            const int lrow = lane >> 2;                        // operand row of the step this lane loads
            const uint32_t boff = uint32_t(ch) * 64u + uint32_t(lane & 3) * 16u;
            auto ld_ids = [&](int qd) -> int {
                const size_t at = size_t(quad0 + qd) * 64 + lane;
                if constexpr (IDS16) return int(p.dcols16[at]);
                else return p.dcols32[at];
            };
            int ids_a = ld_ids(q_lo);
            uint4 aw_a = p.abits[size_t(quad0 + q_lo) * 64 + lane];
            int ids_b = ld_ids(min(q_lo + 1, q_hi - 1));
            uint4 aw_b = p.abits[size_t(quad0 + min(q_lo + 1, q_hi - 1)) * 64 + lane];
            auto issue = [&](int ids, int s4) -> v4u { return ld16(srd, __shfl(ids, s4 * 16 + lrow), boff); };
            // B fragment through the wave's LDS buffer: [k][n] -> lane (n, h) holds k = 8h .. 8h + 7
            auto stage = [&](const v4u& x, v4u& f) {
                wave_lds_order_h();
                *reinterpret_cast<v4u*>(bbuf + lrow * kBRow + (lane & 3) * 8) = x;
                wave_lds_order_h();
                const uint16_t* col = bbuf + (8 * h) * kBRow + n;
                f.x = uint32_t(col[0]) | uint32_t(col[kBRow]) << 16;
                f.y = uint32_t(col[2 * kBRow]) | uint32_t(col[3 * kBRow]) << 16;
                f.z = uint32_t(col[4 * kBRow]) | uint32_t(col[5 * kBRow]) << 16;
                f.w = uint32_t(col[6 * kBRow]) | uint32_t(col[7 * kBRow]) << 16;
            };
            auto mma = [&](const v4u& f, uint32_t aw) {
                const f16x8 b = __builtin_bit_cast(f16x8, f);
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    const f16x8 a = __builtin_bit_cast(f16x8, lut[(aw >> (8 * tt)) & 255u]);
                    acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[tt], 0, 0, 0);
                }
            };
            v4u rA, rB, fA, fB;
            rA = issue(ids_a, 0);
            rB = issue(ids_a, 1);
            stage(rA, fA);
            rA = issue(ids_a, 2);
            for (int qd = q_lo; qd < q_hi; ++qd) {
                stage(rB, fB);
                rB = issue(ids_a, 3);
                mma(fA, aw_a.x);
                stage(rA, fA);
                rA = issue(ids_b, 0);
                mma(fB, aw_a.y);
                stage(rB, fB);
                rB = issue(ids_b, 1);
                mma(fA, aw_a.z);
                stage(rA, fA);
                rA = issue(ids_b, 2);
                mma(fB, aw_a.w);
                ids_a = ids_b;
                aw_a = aw_b;
                const int nx = min(qd + 2, q_hi - 1);
                ids_b = ld_ids(nx);
                aw_b = p.abits[size_t(quad0 + nx) * 64 + lane];
            }
        }
        // (the pipeline's last prefetches are never used: retire them here, so that no wait on their registers
        // lands inside the gather loop)
        __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0), in a form the compiler's wait insertion sees
        __syncthreads();                                     // every wave is done with the table and its operand buffer
        // ------------------------------------------------------------ 2. the two K halves, in order
        {
            const int n = lane & 31, h = lane >> 5;
            for (int w = 0; w < 2; ++w) {
                if (kp == w && q_lo < q_hi) {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int i4 = 0; i4 < 4; ++i4) {
                            float* dst = tile + (32 * t + 8 * i4 + 4 * h) * kHR + 32 * ch + n;   // four consecutive rows
                            float4 v = make_float4(acc[t][4 * i4], acc[t][4 * i4 + 1], acc[t][4 * i4 + 2], acc[t][4 * i4 + 3]);
                            if (w > 0) {
                                v.x += dst[0]; v.y += dst[kHR]; v.z += dst[2 * kHR]; v.w += dst[3 * kHR];
                            }
                            dst[0] = v.x; dst[kHR] = v.y; dst[2 * kHR] = v.z; dst[3 * kHR] = v.w;
                        }
                }
                __syncthreads();
            }
        }
    }

    // ---------------------------------------------------------------- 3. gather phase (the remainder)
    unsigned changed = 0;
    {
        int sb = 0;
        int r_base = 0;
        auto end_of = [&](int k) -> int { return k == 0 ? e1 : k == 1 ? e2 : k == 2 ? e3 : e4; };
        int r_end = e1;
        const int2* gmp = gm_lds + (wave * 8 + g) * 4;
        wave_lds_order_h();
        auto unpack = [](const int2& m) -> int3 {
            const unsigned u = unsigned(m.x);
            const int row = int(u & 255u), end = int(u >> 8);
            return make_int3(row == 255 ? -1 : row, m.y, end == 0xFFFFFF ? -1 : end);
        };
        int3 m_cur = unpack(gmp[0]);
        float cur[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) cur[i] = 0.f;
        int krow = 0;
        const float post = SYM ? p.coef : 1.0f;
        auto emit = [&](const int3& m, const float (&sv)[8]) __attribute__((always_inline)) {
            if (m.x >= 0) {
                const float sc = __int_as_float(m.y) * post;
                float4* tp = reinterpret_cast<float4*>(tile + m.x * kHR + 8 * q);
                float4 d0 = make_float4(0.f, 0.f, 0.f, 0.f), d1 = d0;
                if (has_set) { d0 = tp[0]; d1 = tp[1]; }
                tp[0] = make_float4((sv[0] + d0.x) * sc, (sv[1] + d0.y) * sc, (sv[2] + d0.z) * sc, (sv[3] + d0.w) * sc);
                tp[1] = make_float4((sv[4] + d1.x) * sc, (sv[5] + d1.y) * sc, (sv[6] + d1.z) * sc, (sv[7] + d1.w) * sc);
            }
        };
        auto row_end = [&](int f) __attribute__((always_inline)) {
            if (f + 1 == m_cur.z) {
                emit(m_cur, cur);
#pragma unroll
                for (int i = 0; i < 8; ++i) cur[i] = 0.f;
                ++krow;
                m_cur = unpack(gmp[sb * 128 + min(krow, 3)]);
                if (krow > 3) m_cur.z = -1;
            }
        };
        auto finish = [&]() __attribute__((always_inline)) {
            const float zero[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int k = krow; k < 4; ++k) emit(unpack(gmp[sb * 128 + k]), zero);
            __syncthreads();
            const int row0 = (b0 + sb) * kFB;
            const int nrows = int(min(int64_t(kFB), p.M - row0));
            if constexpr (SYM) {
                // ---------------------------------------------------- 4b. epilogue, direct store, tile updated
                // lane group (wave, g): rows 32 wave + 8 it + g, lane q: columns 8q .. 8q + 7 of the panel.
                // Everything is addressed through buffer descriptors over the panel's slices (one 32-bit
                // offset per access instead of a 64-bit pointer: this block is inlined twice into a loop
                // that holds 16 gathers in flight).
                const int cb = int(c0) + 8 * q;                // first of this lane's columns
                const int Lc = int(p.L);
                constexpr bool full = FULL;                    // (sharded leg 2: the whole block, each element once)
                const bool check = p.prev && !(p.count_any && seen);
                const uint32_t half32 = uint32_t(q >> 2), o32 = uint32_t(8 * q) & 31u;
                const __amdgpu_buffer_rsrc_t ysrd = __builtin_amdgcn_make_buffer_rsrc(
                    p.Y + int64_t(panel) * p.y_rows_pad * 64, 0, int(p.y_rows_pad * 128), 0x00020000);
                const __amdgpu_buffer_rsrc_t osrd = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<uint16_t*>(p.prev ? p.prev + int64_t(panel) * p.prev_rows_pad * 64 : p.X), 0,
                    p.prev ? int(p.prev_rows_pad * 128) : 0, 0x00020000);
                const __amdgpu_buffer_rsrc_t esrd = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<uint8_t*>(p.ev ? p.ev + int64_t(2 * panel) * p.ev_rows_pad * 32 : (const uint8_t*)p.X), 0,
                    p.ev ? int(p.ev_rows_pad * 64) : 0, 0x00020000);
                const __amdgpu_buffer_rsrc_t asrd = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<float*>(PRIOR ? p.ap + int64_t(2 * panel) * p.ap_rows_pad * 32 : (const float*)p.X), 0,
                    PRIOR ? int(p.ap_rows_pad * 256) : 0, 0x00020000);
                const uint32_t ev_off = (half32 * uint32_t(p.ev_rows_pad)) * 32u + o32;
                const uint32_t ap_off = (half32 * uint32_t(p.ap_rows_pad)) * 128u + o32 * 4u;
                // (the loads of RB rows of the lane group together, then the arithmetic; aux 2 = non-temporal)
                constexpr int RB = PRIOR ? 2 : SIMRANK_HALF_RB;
#pragma unroll 1
                for (int pair = 0; pair < 4 / RB; ++pair) {
                    v2u evw[RB];
                    v4u pr0[PRIOR ? RB : 1], pr1[PRIOR ? RB : 1];
                    v4u old[RB];
#pragma unroll
                    for (int it = 0; it < RB; ++it) {
                        const int r = 32 * wave + 8 * (RB * pair + it) + g;
                        const int a = row0 + r;
                        const bool on = r < nrows && (full || a <= cb + 7) && cb < Lc;
                        // (rows that are off load from an offset past the descriptor's end: zeros, no branch)
                        const uint32_t off = on ? uint32_t(a) : 0x7FFFFFF0u / 256u;
                        evw[it] = __builtin_amdgcn_raw_buffer_load_b64(esrd, int(off * 32u + ev_off), 0, 2);
                        if constexpr (PRIOR) {
                            pr0[it] = __builtin_amdgcn_raw_buffer_load_b128(asrd, int(off * 128u + ap_off), 0, 2);
                            pr1[it] = __builtin_amdgcn_raw_buffer_load_b128(asrd, int(off * 128u + ap_off + 16u), 0, 2);
                        }
                        old[it] = check ? __builtin_amdgcn_raw_buffer_load_b128(osrd, int(off * 128u + qoff), 0, 2)
                                        : v4u{0u, 0u, 0u, 0u};
                    }
#pragma unroll
                    for (int it = 0; it < RB; ++it) {
                        const int r = 32 * wave + 8 * (RB * pair + it) + g;
                        const int a = row0 + r;
                        const bool on = r < nrows && (full || a <= cb + 7) && cb < Lc;
                        if (on) {
                            float4* tp = reinterpret_cast<float4*>(tile + r * kHR + 8 * q);
                            const float4 t0 = tp[0], t1 = tp[1];
                            float o[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
                            if (p.ev) {
                                const unsigned w0 = evw[it].x, w1 = evw[it].y;
                                o[0] *= 1.0f - __builtin_ldexpf(1.0f, -int(w0 & 255u));
                                o[1] *= 1.0f - __builtin_ldexpf(1.0f, -int((w0 >> 8) & 255u));
                                o[2] *= 1.0f - __builtin_ldexpf(1.0f, -int((w0 >> 16) & 255u));
                                o[3] *= 1.0f - __builtin_ldexpf(1.0f, -int(w0 >> 24));
                                o[4] *= 1.0f - __builtin_ldexpf(1.0f, -int(w1 & 255u));
                                o[5] *= 1.0f - __builtin_ldexpf(1.0f, -int((w1 >> 8) & 255u));
                                o[6] *= 1.0f - __builtin_ldexpf(1.0f, -int((w1 >> 16) & 255u));
                                o[7] *= 1.0f - __builtin_ldexpf(1.0f, -int(w1 >> 24));
                            }
                            if constexpr (PRIOR) {
                                const float keep = 1.0f - p.lbd;
                                const v4u u0 = pr0[it], u1 = pr1[it];
                                const float pr[8] = {__uint_as_float(u0.x), __uint_as_float(u0.y), __uint_as_float(u0.z),
                                                     __uint_as_float(u0.w), __uint_as_float(u1.x), __uint_as_float(u1.y),
                                                     __uint_as_float(u1.z), __uint_as_float(u1.w)};
#pragma unroll
                                for (int i = 0; i < 8; ++i) o[i] = keep * o[i] + (p.lbd * p.scale) * pr[i];
                            }
                            if (p.set_diag) {
                                const int d = a - cb - (FULL ? p.col0 : 0);
#pragma unroll
                                for (int i = 0; i < 8; ++i)
                                    if (d == i) o[i] = p.scale;
                            }
                            v4u out;
                            out.x = pack2(o[0], o[1]); out.y = pack2(o[2], o[3]);
                            out.z = pack2(o[4], o[5]); out.w = pack2(o[6], o[7]);
                            const v4u od = old[it];
                            const uint32_t ow[4] = {out.x, out.y, out.z, out.w};
                            const uint32_t dw[4] = {od.x, od.y, od.z, od.w};
                            // what was stored is what the next update reads and what the mirror image gets
                            float nvs[8];
#pragma unroll
                            for (int i = 0; i < 8; ++i) {
                                const float nv = half_bits_to_float((ow[i >> 1] >> (16 * (i & 1))) & 0xFFFFu);
                                nvs[i] = nv;
                                if (check) {
                                    // moved by more than eps beyond what the rounding of the stored old value explains:
                                    // the new value before rounding against the old one, eps widened by half the fp16
                                    // spacing at the old value (far below eps for values under 1/8; see the file header)
                                    const uint32_t ob = (dw[i >> 1] >> (16 * (i & 1))) & 0xFFFFu;
                                    const float ov = half_bits_to_float(ob);
                                    const float hu = __builtin_ldexpf(1.0f, max(int((ob >> 10) & 31u), 1) - 26);
                                    const int c = cb + i;
                                    const bool counts = (full || c >= a) && c < Lc;
                                    changed += (counts && fabs(double(o[i]) - double(ov)) > p.eps + double(hu))
                                                   ? (c > a && !full ? 2u : 1u) : 0u;
                                }
                            }
                            tp[0] = make_float4(nvs[0], nvs[1], nvs[2], nvs[3]);
                            tp[1] = make_float4(nvs[4], nvs[5], nvs[6], nvs[7]);
                            const int yoff = int(uint32_t(a) * 128u + qoff);
                            if ((full || a <= cb) && cb + 7 < Lc) {
                                __builtin_amdgcn_raw_buffer_store_b128(out, ysrd, yoff, 0, 2);
                            } else {
#pragma unroll
                                for (int i = 0; i < 8; ++i)
                                    if ((full || cb + i >= a) && cb + i < Lc)
                                        __builtin_amdgcn_raw_buffer_store_b16(
                                            (unsigned short)((ow[i >> 1] >> (16 * (i & 1))) & 0xFFFFu), ysrd, yoff + 2 * i, 0, 0);
                            }
                        }
                    }
                }
                __syncthreads();
            }
            // ------------------------------------------------------------ 4. transposed store
            // wave w: rows 64 (w >> 1) .. + 63 of the block (one 64-column panel of the output), columns
            // 32 (w & 1) .. + 31 of this panel (32 consecutive rows of the output) = 4 KiB contiguous
            if constexpr (!(SYM && FULL)) {
                const int ap = wave >> 1, cw = wave & 1;
                const int rows_here = nrows - 64 * ap;                 // rows of the block in this output panel
                if (rows_here > 0) {
                    uint16_t* base = p.Y + ((int64_t(row0 >> 6) + ap) * p.y_rows_pad + c0 + 32 * cw) * 64;
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int x = lane + it * 64;
                        const int c = x >> 3;                          // column of the wave's half
                        const int a8 = (x & 7) * 8;                    // first of 8 rows of the output panel
                        const int64_t cg = c0 + 32 * cw + c;
                        if (cg < p.L && a8 < rows_here) {
                        const float* t = tile + (64 * ap + a8) * kHR + 32 * cw + c;     // eight rows of one column
                        v4u out;
                        out.x = pack2(t[0], t[kHR]); out.y = pack2(t[2 * kHR], t[3 * kHR]);
                        out.z = pack2(t[4 * kHR], t[5 * kHR]); out.w = pack2(t[6 * kHR], t[7 * kHR]);
                        uint16_t* y = base + c * 64 + a8;
                        bool full = a8 + 7 < rows_here;
                        int64_t lim = rows_here;                       // rows a8 + i < lim are stored
                        if constexpr (SYM) {                           // the mirror image: only a < c
                            const int64_t ag = int64_t(row0) + 64 * ap + a8;
                            full = full && ag + 7 < cg;
                            lim = min(lim, cg - int64_t(row0) - 64 * ap);
                        }
                        if (full) {
                            __builtin_nontemporal_store(out, reinterpret_cast<v4u*>(y));
                        } else {
                            const uint32_t ow[4] = {out.x, out.y, out.z, out.w};
#pragma unroll
                            for (int i = 0; i < 8; ++i)
                                if (a8 + i < lim) y[i] = uint16_t((ow[i >> 1] >> (16 * (i & 1))) & 0xFFFFu);
                        }
                        }
                    }
                }
            }
            ++sb;
            if (sb < n_sub) {
                __syncthreads();
                r_base = r_end;
                r_end = end_of(sb);
                krow = 0;
#pragma unroll
                for (int i = 0; i < 8; ++i) cur[i] = 0.f;
                m_cur = unpack(gmp[sb * 128]);
            }
        };
        auto issue8 = [&](int iv, v4u (&v)[8]) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = ld16(srd, __shfl(iv, gbase + j), qoff);
        };
        auto consume = [&](const v4u (&v)[8], int r) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                add8(cur, v[j]);
                row_end(8 * (r - r_base) + j);
            }
            while (sb < n_sub && r + 1 == r_end) finish();    // (also the blocks after it that have no rounds)
        };
        while (sb < n_sub && r_end == r_base) finish();       // leading blocks without rounds
        if (n_rounds > 0) {
            v4u vA[8], vB[8];
            issue8(to_id(iv0, true), vA);                     // round 0
            int r = 0;
            int ivn = iv1;
            while (r + 2 < n_rounds) {                        // at least two more rounds after r
                const int iv2 = ld_raw(r + 2);
                issue8(to_id(ivn, true), vB);
                consume(vA, r);
                ivn = ld_raw(r + 3);
                issue8(to_id(iv2, true), vA);
                consume(vB, r + 1);
                r += 2;
            }
            if (r + 1 < n_rounds) {                           // vA = round r in flight, one more after it
                issue8(to_id(ivn, true), vB);
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
}

__global__ __launch_bounds__(256) void half_diagonal_kernel(uint16_t* S, int64_t n_rows, int64_t n_cols,
                                                            int64_t rows_pad, int64_t col0, float one) {
    const int64_t c = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
    const int64_t a = col0 + c;
    if (c < n_cols && a < n_rows)
        S[((c >> 6) * rows_pad + a) * 64 + (c & 63)] = __builtin_bit_cast(uint16_t, _Float16(one));
}

// fp16, 64-column panels -> f32, 32-column panels (the layout every hand-back routine reads)
__global__ __launch_bounds__(256) void half_widen_kernel(const uint16_t* __restrict__ src, int64_t src_rows_pad,
                                                         float* __restrict__ dst, int64_t dst_rows_pad,
                                                         int64_t n_rows, int64_t n_panels64, int64_t n_panels32,
                                                         float inv_scale) {
    // one thread = 8 columns of one row
    const int64_t total = n_panels64 * n_rows * 8;
    for (int64_t t = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; t < total; t += int64_t(gridDim.x) * blockDim.x) {
        const int q = int(t & 7);
        const int64_t pr = t >> 3;
        const int64_t pnl = pr / n_rows, r = pr - pnl * n_rows;
        const int64_t p32 = 2 * pnl + (q >> 2);
        if (p32 >= n_panels32) continue;
        const v4u v = *reinterpret_cast<const v4u*>(src + (pnl * src_rows_pad + r) * 64 + 8 * q);
        float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        add8(o, v);
        float4* d = reinterpret_cast<float4*>(dst + (p32 * dst_rows_pad + r) * 32 + ((8 * q) & 31));
        d[0] = make_float4(o[0] * inv_scale, o[1] * inv_scale, o[2] * inv_scale, o[3] * inv_scale);
        d[1] = make_float4(o[4] * inv_scale, o[5] * inv_scale, o[6] * inv_scale, o[7] * inv_scale);
    }
}
// flat conversions, 8 elements per thread (the tail element by element); saturating: a value beyond fp16's range
// becomes the largest finite fp16, not an infinity the receiver would turn into NaN behind a zero evidence factor
__global__ __launch_bounds__(256) void flat_narrow_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst,
                                                          int64_t n, float scale, int vec) {
    const int64_t n8 = vec ? n >> 3 : 0;
    auto cv = [scale](float v) { return fminf(fmaxf(v * scale, -65504.f), 65504.f); };
    for (int64_t t = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; t < n8; t += int64_t(gridDim.x) * blockDim.x) {
        const float4 a = reinterpret_cast<const float4*>(src)[2 * t], b = reinterpret_cast<const float4*>(src)[2 * t + 1];
        v4u o;
        o.x = pack2(cv(a.x), cv(a.y)); o.y = pack2(cv(a.z), cv(a.w));
        o.z = pack2(cv(b.x), cv(b.y)); o.w = pack2(cv(b.z), cv(b.w));
        reinterpret_cast<v4u*>(dst)[t] = o;
    }
    for (int64_t i = 8 * n8 + blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x)
        dst[i] = __builtin_bit_cast(uint16_t, _Float16(cv(src[i])));
}

__global__ __launch_bounds__(256) void flat_widen_kernel(const uint16_t* __restrict__ src, float* __restrict__ dst,
                                                         int64_t n, float inv_scale, int vec) {
    const int64_t n8 = vec ? n >> 3 : 0;
    for (int64_t t = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; t < n8; t += int64_t(gridDim.x) * blockDim.x) {
        const v4u v = reinterpret_cast<const v4u*>(src)[t];
        float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        add8(o, v);
        reinterpret_cast<float4*>(dst)[2 * t] = make_float4(o[0] * inv_scale, o[1] * inv_scale, o[2] * inv_scale, o[3] * inv_scale);
        reinterpret_cast<float4*>(dst)[2 * t + 1] = make_float4(o[4] * inv_scale, o[5] * inv_scale, o[6] * inv_scale, o[7] * inv_scale);
    }
    for (int64_t i = 8 * n8 + blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x)
        dst[i] = half_bits_to_float(src[i]) * inv_scale;
}
#endif  // SIMRANK_HOST_ONLY

}  // namespace simrank

using namespace simrank;

extern "C" {

static bool power_of_two_scale(float s) {
    int e = 0;
    return s >= 1.0f && s <= 32768.0f && std::frexp(s, &e) == 0.5f;
}

int simrank_fill_identity_blocked_h16(void* S, int64_t n_rows, int64_t n_cols, int64_t rows_pad, int64_t col0,
                                      float scale, void* stream) {
    SR_REQUIRE(S && n_rows > 0 && n_cols > 0 && rows_pad >= n_rows, "bad identity block");
    SR_REQUIRE(power_of_two_scale(scale), "scale must be a power of two in 1 .. 32768");
    const size_t bytes = size_t((n_cols + 63) / 64) * size_t(rows_pad) * 64 * sizeof(uint16_t);
    SR_HIP(hipMemsetAsync(S, 0, bytes, as_stream(stream)));
#ifndef SIMRANK_HOST_ONLY
    const int grid = (int)((n_cols + 255) / 256);
    hipLaunchKernelGGL(half_diagonal_kernel, dim3(grid), dim3(256), 0, as_stream(stream), (uint16_t*)S, n_rows,
                       n_cols, rows_pad, col0, scale);
#endif
    SR_HIP(hipGetLastError());
    return SIMRANK_OK;
}

int simrank_widen_blocked_h16(const void* src, int64_t src_rows_pad, float* dst, int64_t dst_rows_pad,
                              int64_t n_rows, int64_t n_cols, float scale, void* stream) {
    SR_REQUIRE(power_of_two_scale(scale), "scale must be a power of two in 1 .. 32768");
    SR_REQUIRE(src && dst && n_rows > 0 && n_cols > 0 && src_rows_pad >= n_rows && dst_rows_pad >= n_rows,
               "bad arguments");
    SR_REQUIRE(aligned16(src) && aligned16(dst), "operands must be 16-byte aligned");
#ifndef SIMRANK_HOST_ONLY
    const int64_t p64 = (n_cols + 63) / 64, p32 = (n_cols + 31) / 32;
    const int64_t total = p64 * n_rows * 8;
    const int grid = (int)std::min<int64_t>((total + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(half_widen_kernel, dim3(grid), dim3(256), 0, as_stream(stream), (const uint16_t*)src,
                       src_rows_pad, dst, dst_rows_pad, n_rows, p64, p32, 1.0f / scale);
#endif
    SR_HIP(hipGetLastError());
    return SIMRANK_OK;
}

// Flat f32 <-> fp16 conversions (value x scale, round to nearest even, saturating): the WIRE format of a sharded
// update's exchange buffers (driver: exchange_precision = "fp16") — the kernels on both sides stay f32.
int simrank_narrow_h16(const float* src, void* dst, int64_t n, float scale, void* stream) {
    SR_REQUIRE(power_of_two_scale(scale), "scale must be a power of two in 1 .. 32768");
    SR_REQUIRE(n >= 0 && (n == 0 || (src && dst)), "bad arguments");
    SR_REQUIRE(reinterpret_cast<uintptr_t>(src) % 4 == 0 && reinterpret_cast<uintptr_t>(dst) % 2 == 0, "misaligned operands");
#ifndef SIMRANK_HOST_ONLY
    if (n > 0) {
        const int vec = aligned16(src) && aligned16(dst);        // (16 bytes per lane where both sides allow it)
        const int grid = (int)std::min<int64_t>(((n + 7) / 8 + 255) / 256, 256 * 32);
        hipLaunchKernelGGL(flat_narrow_kernel, dim3(grid), dim3(256), 0, as_stream(stream), src, (uint16_t*)dst, n, scale, vec);
    }
#endif
    SR_HIP(hipGetLastError());
    return SIMRANK_OK;
}

int simrank_widen_h16(const void* src, float* dst, int64_t n, float scale, void* stream) {
    SR_REQUIRE(power_of_two_scale(scale), "scale must be a power of two in 1 .. 32768");
    SR_REQUIRE(n >= 0 && (n == 0 || (src && dst)), "bad arguments");
    SR_REQUIRE(reinterpret_cast<uintptr_t>(src) % 2 == 0 && reinterpret_cast<uintptr_t>(dst) % 4 == 0, "misaligned operands");
#ifndef SIMRANK_HOST_ONLY
    if (n > 0) {
        const int vec = aligned16(src) && aligned16(dst);
        const int grid = (int)std::min<int64_t>(((n + 7) / 8 + 255) / 256, 256 * 32);
        hipLaunchKernelGGL(flat_widen_kernel, dim3(grid), dim3(256), 0, as_stream(stream), (const uint16_t*)src, dst, n,
                           1.0f / scale, vec);
    }
#endif
    SR_HIP(hipGetLastError());
    return SIMRANK_OK;
}

int simrank_spmm_blocked_h16(const simrank_graph* g, const void* X, int64_t x_rows_pad, int64_t n_cols_x, void* Y,
                             int64_t y_rows_pad, int32_t transpose_out, const simrank_epilogue* ep,
                             int64_t aux_rows_pad, float scale, void* stream) {
    SR_REQUIRE(g && X && Y && n_cols_x > 0, "bad arguments");
    SR_REQUIRE(power_of_two_scale(scale), "scale must be a power of two in 1 .. 32768");
    SR_REQUIRE(x_rows_pad >= g->n_cols && y_rows_pad >= (transpose_out ? n_cols_x : g->n_rows),
               "padded row counts %lld / %lld too small", (long long)x_rows_pad, (long long)y_rows_pad);
    SR_REQUIRE((transpose_out != 0) != (ep != nullptr), "leg 1: transpose_out and no epilogue; leg 2: an epilogue");
    const simrank_fused_plan* pl = g->fused;
    SR_REQUIRE(pl, "fp16 storage needs the one-launch plan (tuning fuse = 1 when the graph is created)");
    SR_REQUIRE(pl->n_pslots == 0, "fp16 storage: blocks cut into several units (tuning fuse_unit) are not supported");
    SR_REQUIRE(aligned16(X) && aligned16(Y), "operands must be 16-byte aligned");
    SR_REQUIRE((x_rows_pad + 1) * 128 < (int64_t(1) << 31) && x_rows_pad < (int64_t(1) << 24) - 1,
               "operand of %lld rows per panel", (long long)x_rows_pad);
    hipStream_t st = as_stream(stream);
    HalfArgs a{};
    a.X = (const uint16_t*)X; a.Y = (uint16_t*)Y;
    a.x_rows_pad = x_rows_pad; a.y_rows_pad = y_rows_pad;
    a.L = n_cols_x; a.M = g->n_rows;
    a.n_panels = int32_t((n_cols_x + 63) / 64);
    a.n_units = pl->n_units;
    a.x_sentinel = (int32_t)x_rows_pad;
    a.units = pl->units;
    a.dcols16 = pl->dcols16; a.dcols32 = pl->dcols32; a.abits = pl->abits;
    a.gmeta = pl->gmeta; a.sids16 = pl->sids16; a.sids32 = pl->sids32;
    if (ep) {
        // symmetric = 1: one rank holds the whole matrix (upper triangle + mirror image); symmetric = 0: the column block
        // [diag_col0, diag_col0 + n_cols_x) of one rank of a sharded update, every element computed (csrc/shardplan.hip)
        SR_REQUIRE(ep->symmetric ? (n_cols_x == g->n_rows && ep->diag_col0 == 0)
                                 : (ep->diag_col0 >= 0 && ep->diag_col0 + n_cols_x <= g->n_rows),
                   "fp16 storage: leg 2 is the symmetric single-rank form, or a column block of a sharded update");
        SR_REQUIRE(ep->symmetric || !ep->apriori, "fp16 storage: a sharded leg 2 takes no prior");
        a.full = ep->symmetric ? 0 : 1;
        a.col0 = (int32_t)ep->diag_col0;
        a.coef = ep->coef; a.lbd = ep->lbd; a.eps = ep->eps * double(scale);
        a.scale = scale;
        a.ev = ep->evidence; a.ap = ep->apriori;
        a.ev_rows_pad = a.ap_rows_pad = aux_rows_pad;
        a.prev = (const uint16_t*)ep->previous; a.prev_rows_pad = y_rows_pad;
        a.n_changed = ep->n_changed;
        a.set_diag = ep->set_diag; a.count_any = ep->count_any;
        SR_REQUIRE(!(a.ev || a.ap) || aux_rows_pad >= g->n_rows, "evidence / prior: padded rows too small");
        SR_REQUIRE(!a.ev || (reinterpret_cast<uintptr_t>(a.ev) % 8 == 0), "evidence must be 8-byte aligned");
        SR_REQUIRE(!a.ap || aligned16(a.ap), "prior must be 16-byte aligned");
        SR_REQUIRE(!a.prev || (aligned16(a.prev) && a.n_changed), "previous needs 16-byte alignment and a counter");
        if (a.prev) SR_HIP(hipMemsetAsync(a.n_changed, 0, sizeof(unsigned long long) * SIMRANK_CHANGED_SLOTS, st));
    }
    const int64_t grid = int64_t((a.n_panels + 7) / 8) * 8 * a.n_units;
    SR_REQUIRE(grid > 0 && grid < (int64_t(1) << 31), "grid of %lld blocks", (long long)grid);
#ifdef SIMRANK_HOST_ONLY
    SR_REQUIRE(false, "host-only build: no kernels");
#else
#define SR_HALF(SYM, PRI) \
    do { if (pl->ids16) hipLaunchKernelGGL((half_leg_kernel<true, SYM, PRI>), dim3((unsigned)grid), dim3(256), 0, st, a); \
         else hipLaunchKernelGGL((half_leg_kernel<false, SYM, PRI>), dim3((unsigned)grid), dim3(256), 0, st, a); } while (0)
    if (ep && a.full) {
        if (pl->ids16) hipLaunchKernelGGL((half_leg_kernel<true, true, false, true>), dim3((unsigned)grid), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((half_leg_kernel<false, true, false, true>), dim3((unsigned)grid), dim3(256), 0, st, a);
    } else if (ep && a.ap) SR_HALF(true, true);
    else if (ep) SR_HALF(true, false);
    else SR_HALF(false, false);
#undef SR_HALF
#endif
    SR_HIP(hipGetLastError());
    return SIMRANK_OK;
}

}  // extern "C"
