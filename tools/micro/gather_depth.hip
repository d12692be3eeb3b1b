// Measurement aid (not part of the library) — the GATE of round 6 for leg 1:
// how fast can gfx950 gather the REAL remainder id streams of a workload's one-launch plan, as a function of the
// bytes a CU keeps in flight and of where they land (VGPRs, or LDS through `buffer_load_dwordx4 ... lds`)?
//
// Input: the plan dump of tools/dump_fused_plan.py (SIMRANK_DUMP_FUSED_PLAN in fused.hip): unit records and the id
// streams (64 ids per round = 8 slots x 8 lane groups) exactly as fused_trans_kernel reads them.  The grid, the
// panel -> XCD binding, the launch order of a panel's units and the access shape (a wave instruction = 64 x 16 B, each
// group of 8 lanes one 128-byte line of the panel's 4 MiB slice) are the launch's own; what is left out is everything
// else a workgroup of the real kernel does (matrix-core phase, row ends, the LDS tile), so the figure is the ceiling
// of the gather phase with that many loads in flight — not a kernel.
//
//   DEPTH d: rounds r+1 .. r+d-1 are in flight while round r is summed (the product kernel is DEPTH 2).
//   "builtin": the product's own form — compiler-visible buffer loads into two register sets, the compiler's waits.
//   "vgpr":    d register sets of 8 x float4, loads and waits written by hand (inline asm, `s_waitcnt vmcnt(8 (d-1))`):
//              this compiler rotates a deeper pipeline of VISIBLE loads through register copies behind `vmcnt(0)`.
//   "lds":     a ring of d slots x 8 KiB per wave filled by `buffer_load_dwordx4 ... lds`, read back with ds_read_b128;
//              by hand as well (the compiler puts `vmcnt(0)` in front of every LDS read that follows a visible LDS-DMA).
//   The ids of a wave's stream are staged in LDS first (chunks of 32 rounds), so that no compiler-visible vector load
//   is pending while the hand-counted ones are in flight.
//   Every variant is CHECKED (operand of ones: the sum over all lane groups must equal the number of ids x panels).
//
//   hipcc -O3 -std=c++20 --offload-arch=gfx950 tools/micro/gather_depth.hip -o build/gather_depth
//   build/gather_depth gpurun_out/plan_pl32768d32.bin
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kChunk = 32;          // rounds of ids staged in LDS at a time (4 KiB per wave)
constexpr int kIdRows = kChunk + 8; // + empty rounds behind the chunk (what the prologue of a short chunk reads)

struct Args {
    const float* X;
    float* Y;
    const int32_t* units;
    const uint16_t* sids;
    int64_t rows_pad;
    int32_t n_panels, n_units;
    int32_t mask;       // ids are ANDed with it (0xFFFF: the real stream; 8191: L2-resident; 255: L1-resident)
    int32_t store;      // 1: every wave also stores 4 KiB per block of its unit (the transposed tile's bytes, nt)
    int32_t check;      // 1: the lane groups' sums are added into *total (untimed launch)
    unsigned long long* total;
    float* sink;
};

typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));

enum { kBuiltin = 0, kVgpr = 1, kLds = 2 };

template <int N> using IC = std::integral_constant<int, N>;
template <int N, typename F> __device__ __forceinline__ void unroll(F&& f) {
    [&]<int... Is>(std::integer_sequence<int, Is...>) { (f(IC<Is>{}), ...); }(std::make_integer_sequence<int, N>{});
}


// (the asm statements live in plain functions: this clang rejects asm operands that name captured variables inside a generic lambda)
__device__ __forceinline__ void asm_load_vgpr(v4u& dst, int voff, v4i srd) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(dst) : "v"(voff), "s"(srd) : "memory");
}
__device__ __forceinline__ void asm_load_lds(int voff, v4i srd, uint32_t lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(srd), "s"(lds_addr) : "memory");
}
template <int AFTER> __device__ __forceinline__ void asm_wait_set(v4u (&v)[8]) {
    asm volatile("s_waitcnt vmcnt(%8)"
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])
                 : "n"(AFTER) : "memory");
}
template <int AFTER, int BASE> __device__ __forceinline__ void asm_read_slot(v4u (&x)[8], uint32_t at) {
    asm volatile("s_waitcnt vmcnt(%9)\n\t"
                 "ds_read_b128 %0, %8 offset:%10\n\tds_read_b128 %1, %8 offset:%11\n\t"
                 "ds_read_b128 %2, %8 offset:%12\n\tds_read_b128 %3, %8 offset:%13\n\t"
                 "ds_read_b128 %4, %8 offset:%14\n\tds_read_b128 %5, %8 offset:%15\n\t"
                 "ds_read_b128 %6, %8 offset:%16\n\tds_read_b128 %7, %8 offset:%17\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(x[4]), "=&v"(x[5]), "=&v"(x[6]), "=&v"(x[7])
                 : "v"(at), "n"(AFTER), "n"(BASE), "n"(BASE + 1024), "n"(BASE + 2048), "n"(BASE + 3072), "n"(BASE + 4096),
                   "n"(BASE + 5120), "n"(BASE + 6144), "n"(BASE + 7168)
                 : "memory");
}

template <int D, int LAND, int WPS>
__global__ __launch_bounds__(256, WPS) void gather_kernel(const Args p) {
    __shared__ __attribute__((aligned(16))) uint16_t ids_lds[4][kIdRows][64];
    extern __shared__ __attribute__((aligned(16))) unsigned char ring_all[];   // LDS landing: [wave][D][8][64 x 16 B]
    static_assert(D - 1 <= 8, "the empty rounds behind a chunk cover the prologue");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t bid = blockIdx.x, local = bid >> 3;
    const int panel = int(local / uint32_t(p.n_units)) * 8 + int(bid & 7);
    if (panel >= p.n_panels) return;
    const uint32_t unit = local % uint32_t(p.n_units);
    const int32_t* un = p.units + size_t(unit) * 32;
    const int n_sub = un[8];
    const int round0 = un[9 + wave * 5];
    const int n_all = un[9 + wave * 5 + n_sub];              // rounds of this wave's stream
    const int q = lane & 7, gbase = lane & ~7;
    const uint32_t qoff = uint32_t(q) * 16u;
    const float* xbase = p.X + int64_t(panel) * p.rows_pad * 32;
    const int sent = int(p.rows_pad);
    float4 cur = make_float4(0.f, 0.f, 0.f, 0.f);
    auto fix = [&](int v) -> int { return v == 0xFFFF ? sent : (v & p.mask); };

    if constexpr (LAND == kBuiltin) {
        // ---- the product's gather loop (fused.hip, step 3): two sets, visible loads, ids one round ahead
        static_assert(D == 2, "builtin form is depth 2");
        const __amdgpu_buffer_rsrc_t srd =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xbase), 0, int32_t(p.rows_pad * 128), 0x00020000);
        const int n = n_all;
        auto ld_raw = [&](int r) -> int { return int(p.sids[(size_t(round0) + size_t(min(r, max(n - 1, 0)))) * 64 + lane]); };
        auto issue8 = [&](int iv, float4 (&v)[8]) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const v4u w = __builtin_amdgcn_raw_buffer_load_b128(srd, int(__umul24(uint32_t(__shfl(iv, gbase + j)), 128u) + qoff), 0, 0);
                v[j] = make_float4(__uint_as_float(w.x), __uint_as_float(w.y), __uint_as_float(w.z), __uint_as_float(w.w));
            }
        };
        auto consume = [&](const float4 (&v)[8]) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { cur.x += v[j].x; cur.y += v[j].y; cur.z += v[j].z; cur.w += v[j].w; }
        };
        if (n > 0) {
            const int raw0 = ld_raw(0), raw1 = ld_raw(1);
            __builtin_amdgcn_sched_barrier(0);
            int iv1 = fix(raw1);
            float4 vA[8], vB[8];
            issue8(fix(raw0), vA);
            int r = 0;
            while (r + 2 < n) {
                const int iv2 = fix(ld_raw(r + 2));
                issue8(iv1, vB);
                consume(vA);
                iv1 = fix(ld_raw(r + 3));
                issue8(iv2, vA);
                consume(vB);
                r += 2;
            }
            if (r + 1 < n) {
                issue8(iv1, vB);
                consume(vA);
                consume(vB);
            } else {
                consume(vA);
            }
        }
    } else {
        v4i srd_s;                                           // the buffer descriptor as four scalars (for the asm loads)
        {
            const uint64_t a = reinterpret_cast<uint64_t>(xbase);
            srd_s.x = __builtin_amdgcn_readfirstlane(int(uint32_t(a)));
            srd_s.y = __builtin_amdgcn_readfirstlane(int(uint32_t(a >> 32) & 0xFFFFu));
            srd_s.z = __builtin_amdgcn_readfirstlane(int(p.rows_pad * 128));
            srd_s.w = 0x00020000;
        }
        unsigned char* ring = ring_all + size_t(wave) * D * 8192;
        const uint32_t ring_addr = __builtin_amdgcn_readfirstlane(uint32_t(reinterpret_cast<uintptr_t>(ring)));   // LDS byte address
        v4u v[LAND == kVgpr ? D : 1][8];
        for (int c0 = 0; c0 < n_all; c0 += kChunk) {
            const int n = min(kChunk, n_all - c0);           // rounds of this chunk
            // stage the chunk's ids in LDS (visible loads: everything of the previous chunk has been summed)
            for (int r = 0; r < n; ++r) ids_lds[wave][r][lane] = p.sids[(size_t(round0) + size_t(c0 + r)) * 64 + lane];
            for (int r = n; r < min(n + 8, kIdRows); ++r) ids_lds[wave][r][lane] = 0xFFFF;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            auto id_of = [&](int r) -> int { return fix(int(ids_lds[wave][min(r, kIdRows - 1)][lane])); };   // (past the chunk: empty)

            auto issue = [&](int id, auto SLOT) {
                constexpr int slot = decltype(SLOT)::value;
                unroll<8>([&](auto J) {
                    constexpr int j = decltype(J)::value;
                    const int voff = int(__umul24(uint32_t(__shfl(id, gbase + j)), 128u) + qoff);
                    if constexpr (LAND == kVgpr) asm_load_vgpr(v[slot][j], voff, srd_s);
                    else asm_load_lds(voff, srd_s, ring_addr + uint32_t(slot * 8192 + j * 1024));
                });
            };
            // the round in slot `slot` has landed when at most `after` newer loads are outstanding; sum it
            auto consume = [&](auto SLOT, auto AFTER) {
                constexpr int slot = decltype(SLOT)::value, after = decltype(AFTER)::value;
                if constexpr (LAND == kVgpr) {
                    asm_wait_set<after>(v[slot]);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        cur.x += __uint_as_float(v[slot][j].x); cur.y += __uint_as_float(v[slot][j].y);
                        cur.z += __uint_as_float(v[slot][j].z); cur.w += __uint_as_float(v[slot][j].w);
                    }
                } else {
                    v4u x[8];
                    asm_read_slot<after, slot * 8192>(x, ring_addr + uint32_t(lane) * 16u);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        cur.x += __uint_as_float(x[j].x); cur.y += __uint_as_float(x[j].y);
                        cur.z += __uint_as_float(x[j].z); cur.w += __uint_as_float(x[j].w);
                    }
                }
            };
            // prologue: rounds 0 .. D-2 (past the chunk: empty ids); stage s (slot j = s % D): round s + D - 1 issued, round s
            // summed.  Whole groups of D stages in a branch-free loop body, the last stages and the drain as straight-line
            // code per case.
            unroll<D - 1>([&](auto Kc) { issue(id_of(decltype(Kc)::value), Kc); });
            int s = 0;
            auto stage = [&](auto J) {
                constexpr int j = decltype(J)::value;
                issue(id_of(s + D - 1), IC<(j + D - 1) % D>{});
                consume(J, IC<8 * (D - 1)>{});
                ++s;
            };
            auto drain = [&](auto J) {
                constexpr int j = decltype(J)::value;
                unroll<D - 1>([&](auto Kc) {
                    constexpr int k = decltype(Kc)::value;
                    consume(IC<(j + k) % D>{}, IC<8 * (D - 2 - k)>{});
                });
            };
            auto tail = [&](auto self, auto J, int rem) -> void {
                constexpr int j = decltype(J)::value;
                if constexpr (j < D - 1) {
                    if (rem > j) {
                        stage(J);
                        self(self, IC<j + 1>{}, rem);
                    } else {
                        drain(J);
                    }
                } else {
                    drain(J);
                }
            };
            const int n_main = max(n - (D - 1), 0);          // stages that issue a round
            while (s + D <= n_main) unroll<D>([&](auto J) { stage(J); });
            tail(tail, IC<0>{}, n_main - s);
        }
    }
    if (p.check) {
        if (q == 0) atomicAdd(p.total, (unsigned long long)(cur.x + 0.5f));
    }
    if (p.store) {
        // the bytes of the transposed tiles: one 4 KiB tile per wave and block of the unit (panel-blocked Tt)
        const int b0 = un[0];
        for (int sb = 0; sb < n_sub; ++sb) {
            float* base = p.Y + ((int64_t((b0 + sb) * 4) + wave) * p.rows_pad + int64_t(panel) * 32) * 32;
            const __amdgpu_buffer_rsrc_t ysrd = __builtin_amdgcn_make_buffer_rsrc(base, 0, 4096, 0x00020000);
            v4u o;
            o.x = __float_as_uint(cur.x); o.y = __float_as_uint(cur.y); o.z = __float_as_uint(cur.z); o.w = __float_as_uint(cur.w);
#pragma unroll
            for (int it = 0; it < 4; ++it) __builtin_amdgcn_raw_buffer_store_b128(o, ysrd, (lane + it * 64) * 16, 0, 2);
        }
    } else if (cur.x + cur.y + cur.z + cur.w == 12345.678f) {
        p.sink[0] = cur.x;
    }
}

// The product's form (visible loads, two register sets) with the ids requested THREE rounds ahead instead of one
// (four id registers), or staged in LDS a chunk at a time (LDSIDS): does the gather loop wait for its id loads?
template <int IDK, bool LDSIDS>
__global__ __launch_bounds__(256, 4) void gather_builtin_kernel(const Args p) {
    __shared__ __attribute__((aligned(16))) uint16_t ids_lds[4][kIdRows][64];
    extern __shared__ __attribute__((aligned(16))) unsigned char pad_all[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t bid = blockIdx.x, local = bid >> 3;
    const int panel = int(local / uint32_t(p.n_units)) * 8 + int(bid & 7);
    if (panel >= p.n_panels) return;
    const uint32_t unit = local % uint32_t(p.n_units);
    const int32_t* un = p.units + size_t(unit) * 32;
    const int n_sub = un[8];
    const int round0 = un[9 + wave * 5];
    const int n_all = un[9 + wave * 5 + n_sub];
    const int q = lane & 7, gbase = lane & ~7;
    const uint32_t qoff = uint32_t(q) * 16u;
    const float* xbase = p.X + int64_t(panel) * p.rows_pad * 32;
    const int sent = int(p.rows_pad);
    const __amdgpu_buffer_rsrc_t srd =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xbase), 0, int32_t(p.rows_pad * 128), 0x00020000);
    float4 cur = make_float4(0.f, 0.f, 0.f, 0.f);
    auto fix = [&](int v) -> int { return v == 0xFFFF ? sent : (v & p.mask); };
    float4 v[2][8];
    auto issue8 = [&](int iv, float4 (&d)[8]) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const v4u w = __builtin_amdgcn_raw_buffer_load_b128(srd, int(__umul24(uint32_t(__shfl(iv, gbase + j)), 128u) + qoff), 0, 0);
            d[j] = make_float4(__uint_as_float(w.x), __uint_as_float(w.y), __uint_as_float(w.z), __uint_as_float(w.w));
        }
    };
    auto consume = [&](const float4 (&d)[8]) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { cur.x += d[j].x; cur.y += d[j].y; cur.z += d[j].z; cur.w += d[j].w; }
    };
    // (written out by hand in the product loop's own style — named variables, plain while loops: the generic version of this
    // loop came back from the compiler with a drained wait at the loop head)
    for (int c0 = 0; c0 < n_all; c0 += (LDSIDS ? kChunk : (1 << 30))) {
        const int n = LDSIDS ? min(kChunk, n_all - c0) : n_all;
        if constexpr (LDSIDS) {
            for (int r = 0; r < n; ++r) ids_lds[wave][r][lane] = p.sids[(size_t(round0) + size_t(c0 + r)) * 64 + lane];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        auto ld = [&](int r) -> int {                        // (past the stream: the last round again; never used)
            const int rc = min(r, max(n - 1, 0));
            if constexpr (LDSIDS) return int(ids_lds[wave][rc][lane]);
            else return int(p.sids[(size_t(round0) + size_t(rc)) * 64 + lane]);
        };
        float4 (&vA)[8] = v[0];
        float4 (&vB)[8] = v[1];
        int raw0 = ld(0), raw1 = ld(1), raw2 = ld(2), raw3 = ld(3);
        __builtin_amdgcn_sched_barrier(0);
        issue8(fix(raw0), vA);
        int r = 0;
#define SB __builtin_amdgcn_sched_barrier(0)   /* (the scheduler otherwise sums a set BEFORE it issues the next one) */
        while (r + 4 < n) {
            issue8(fix(raw1), vB); raw0 = ld(r + 4); SB; consume(vA); SB;
            issue8(fix(raw2), vA); raw1 = ld(r + 5); SB; consume(vB); SB;
            issue8(fix(raw3), vB); raw2 = ld(r + 6); SB; consume(vA); SB;
            issue8(fix(raw0), vA); raw3 = ld(r + 7); SB; consume(vB); SB;
            r += 4;
        }
#undef SB
        if (r + 1 < n) {
            issue8(fix(raw1), vB); consume(vA);
            if (r + 2 < n) {
                issue8(fix(raw2), vA); consume(vB);
                if (r + 3 < n) {
                    issue8(fix(raw3), vB); consume(vA); consume(vB);
                } else {
                    consume(vA);
                }
            } else {
                consume(vB);
            }
        } else {
            consume(vA);
        }
    }
    if (p.check) {
        if (q == 0) atomicAdd(p.total, (unsigned long long)(cur.x + 0.5f));
    }
    if (p.store) {
        const int b0 = un[0];
        for (int sb = 0; sb < n_sub; ++sb) {
            float* base = p.Y + ((int64_t((b0 + sb) * 4) + wave) * p.rows_pad + int64_t(panel) * 32) * 32;
            const __amdgpu_buffer_rsrc_t ysrd = __builtin_amdgcn_make_buffer_rsrc(base, 0, 4096, 0x00020000);
            v4u o;
            o.x = __float_as_uint(cur.x); o.y = __float_as_uint(cur.y); o.z = __float_as_uint(cur.z); o.w = __float_as_uint(cur.w);
#pragma unroll
            for (int it = 0; it < 4; ++it) __builtin_amdgcn_raw_buffer_store_b128(o, ysrd, (lane + it * 64) * 16, 0, 2);
        }
    } else if (cur.x + cur.y + cur.z + cur.w == 12345.678f) {
        p.sink[0] = cur.x;
    }
}

__global__ void fill_ones(float4* x, size_t n4) {
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n4; i += size_t(gridDim.x) * blockDim.x)
        x[i] = make_float4(1.f, 1.f, 1.f, 1.f);
}

typedef void (*kern_t)(const Args);
static void report_kernel(const char* name, kern_t kern, Args a, double ids_per_panel, int wgs_per_cu, size_t ring, int depth) {
    hipFuncAttributes fa;
    CHECK(hipFuncGetAttributes(&fa, (const void*)kern));
    // dynamic LDS: the ring (LDS landing) plus padding so that exactly wgs_per_cu workgroups fit a CU's 160 KiB
    const size_t per_wg = (size_t(160) * 1024 / wgs_per_cu) & ~size_t(255);
    if (per_wg < fa.sharedSizeBytes + ring) { printf("%-24s: does not fit %d workgroups per CU\n", name, wgs_per_cu); return; }
    const size_t dyn = per_wg - fa.sharedSizeBytes;
    CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
    const int64_t grid = int64_t((a.n_panels + 7) / 8) * 8 * a.n_units;
    const int waves_per_cu = wgs_per_cu * 4;
    // check launch
    CHECK(hipMemset(a.total, 0, 8));
    a.check = 1; a.store = 0;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), dyn, 0, a);
    CHECK(hipDeviceSynchronize());
    unsigned long long total = 0;
    CHECK(hipMemcpy(&total, a.total, 8, hipMemcpyDeviceToHost));
    const unsigned long long want = (unsigned long long)(ids_per_panel * a.n_panels + 0.5);
    char verdict[96];
    if (total == want) snprintf(verdict, sizeof verdict, "ok");
    else snprintf(verdict, sizeof verdict, "MISMATCH (%llu of %llu)", total, want);
    a.check = 0;
    for (int store = 0; store < 2; ++store) {
        a.store = store;
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), dyn, 0, a);
        CHECK(hipEventRecord(e0));
        for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), dyn, 0, a);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        ms /= 3;
        CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
        const double bytes = ids_per_panel * 128.0 * a.n_panels;
        printf("%-24s mask %5d store %d: %7.3f ms %6.2f TB/s gathered | %3d VGPRs %2d waves/CU %3d KiB in flight/CU (%d rounds/wave) check %s\n",
               name, a.mask, store, ms, bytes / (ms * 1e-3) / 1e12, fa.numRegs, waves_per_cu, waves_per_cu * (depth - 1) * 8, depth - 1,
               verdict);
        fflush(stdout);
    }
}

template <int D, int LAND, int WPS>
static void report(const char* name, Args a, double ids_per_panel, int wgs_per_cu) {
    report_kernel(name, gather_kernel<D, LAND, WPS>, a, ids_per_panel, wgs_per_cu, LAND == kLds ? size_t(4) * D * 8192 : 0, D);
}

int main(int argc, char** argv) {
    if (argc < 2) { printf("usage: gather_depth plan.bin\n"); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { printf("cannot open %s\n", argv[1]); return 2; }
    int64_t hdr[8];
    if (fread(hdr, 8, 8, f) != 8 || hdr[0] != 0x53524450) { printf("bad header\n"); return 2; }
    const int64_t M = hdr[1], K = hdr[2], n_units = hdr[3], n_rounds = hdr[4], n_quads = hdr[5], r_nnz = hdr[6], covered = hdr[7];
    std::vector<int32_t> units(size_t(n_units) * 32), sids(size_t(n_rounds) * 64);
    if (fread(units.data(), 4, units.size(), f) != units.size() || fread(sids.data(), 4, sids.size(), f) != sids.size()) { printf("short file\n"); return 2; }
    fclose(f);
    if (K > 65535) { printf("this aid reads 16-bit ids (K = %lld)\n", (long long)K); return 2; }
    std::vector<uint16_t> s16(sids.size() + 64 * 8, 0xFFFF);
    int64_t real = 0, max_rounds = 0, waves = 0;
    for (size_t i = 0; i < sids.size(); ++i) { s16[i] = sids[i] < 0 ? 0xFFFF : uint16_t(sids[i]); real += sids[i] >= 0; }
    for (int64_t u = 0; u < n_units; ++u)
        for (int w = 0; w < 4; ++w) {
            const int nr = units[size_t(u) * 32 + 9 + w * 5 + units[size_t(u) * 32 + 8]];
            max_rounds = nr > max_rounds ? nr : max_rounds;
            waves += nr > 0;
        }
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("%s: %d CUs | plan: M %lld K %lld, %lld units per panel, %lld rounds (%lld ids, %lld in the file's count; fill %.3f), "
           "dense sets %lld quads / %lld entries; mean %.1f rounds per working wave, longest %lld\n",
           prop.name, prop.multiProcessorCount, (long long)M, (long long)K, (long long)n_units, (long long)n_rounds,
           (long long)real, (long long)r_nnz, double(real) / double(n_rounds * 64), (long long)n_quads, (long long)covered,
           double(n_rounds) / double(waves ? waves : 1), (long long)max_rounds);
    const int64_t rows_pad = (std::max(M, K) + 127) / 128 * 128;
    const int n_panels = int((M + 31) / 32);          // (the operand of leg 1 is S: M = K columns)
    Args a{};
    float *X, *Y, *sink;
    const size_t xb = size_t(n_panels) * rows_pad * 128;
    CHECK(hipMalloc(&X, xb)); CHECK(hipMalloc(&Y, xb)); CHECK(hipMalloc(&sink, 64));
    hipLaunchKernelGGL(fill_ones, dim3(4096), dim3(256), 0, 0, reinterpret_cast<float4*>(X), xb / 16);
    CHECK(hipDeviceSynchronize());
    int32_t* d_units; uint16_t* d_sids;
    CHECK(hipMalloc(&d_units, units.size() * 4)); CHECK(hipMalloc(&d_sids, s16.size() * 2));
    CHECK(hipMemcpy(d_units, units.data(), units.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_sids, s16.data(), s16.size() * 2, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&a.total, 8));
    a.X = X; a.Y = Y; a.units = d_units; a.sids = d_sids; a.rows_pad = rows_pad;
    a.n_panels = n_panels; a.n_units = int(n_units); a.sink = sink;
    const double ipp = double(real);
    const bool brief = argc > 2 && std::string(argv[2]) == "--brief";      // (for a counter pass: the product's loop at the three masks)
    for (int mask : {0xFFFF, 8191, 255}) {
        a.mask = mask;
        report<2, kBuiltin, 4>("builtin depth 2 (today)", a, ipp, 4);
        if (brief) continue;
        // the same loop with its ids requested further ahead / staged in LDS
        report_kernel("builtin, ids 3 ahead", gather_builtin_kernel<4, false>, a, ipp, 4, 0, 2);
        report_kernel("builtin, ids via LDS", gather_builtin_kernel<4, true>, a, ipp, 4, 0, 2);
        if (mask != 0xFFFF) continue;
        // the same code at other occupancies (its 70 VGPRs allow 7 waves per SIMD): bytes in flight per CU by WAVES
        report<2, kBuiltin, 4>("builtin depth 2", a, ipp, 2);
        report<2, kBuiltin, 4>("builtin depth 2", a, ipp, 3);
        report<2, kBuiltin, 4>("builtin depth 2", a, ipp, 5);
        report<2, kBuiltin, 4>("builtin depth 2", a, ipp, 6);
        report<2, kBuiltin, 4>("builtin depth 2", a, ipp, 7);
        report_kernel("builtin, ids 3 ahead", gather_builtin_kernel<4, false>, a, ipp, 6, 0, 2);
        // deeper per wave, by hand
        report<2, kVgpr, 4>("vgpr depth 2", a, ipp, 4);
        report<3, kVgpr, 3>("vgpr depth 3", a, ipp, 3);
        report<3, kVgpr, 3>("vgpr depth 3", a, ipp, 2);
        report<4, kVgpr, 2>("vgpr depth 4", a, ipp, 2);
        // landing in LDS (the ring limits a CU to one workgroup)
        report<3, kLds, 4>("lds depth 3", a, ipp, 1);        // 96 KiB of ring
        report<4, kLds, 4>("lds depth 4", a, ipp, 1);        // 128 KiB
    }
    return 0;
}
