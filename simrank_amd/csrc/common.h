// Internal declarations shared by the translation units of libsimrank_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "simrank_hip.h"

namespace simrank {

void set_error(const char* fmt, ...);

#define SR_HIP(call)                                                                   \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess) {                                                        \
            ::simrank::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                                 __FILE__, __LINE__);                                  \
            return SIMRANK_ERR_HIP;                                                    \
        }                                                                              \
    } while (0)

#define SR_REQUIRE(cond, ...)                 \
    do {                                      \
        if (!(cond)) {                        \
            ::simrank::set_error(__VA_ARGS__); \
            return SIMRANK_ERR_INVALID;       \
        }                                     \
    } while (0)

struct Tuning {
    int64_t panel = 0;    // 0 = automatic
    int64_t xcd_map = 1;  // panel -> XCD affinity (blockIdx % 8 shares an L2)
    int64_t stream_nt = 1; // non-temporal loads/stores for streamed-once data
    int64_t tile = 0;     // rows per wave tile (16, 32, 64; 0 = automatic)
    int64_t huge = 512;   // rows of at least this many entries are split over a workgroup's waves
    int64_t triangle = 1; // allow the upper-triangle + mirror form of a symmetric leg 2
    int64_t balance = 4;  // cut 32-row tiles heavier than balance x the mean tile (0 = uniform tiles)
};
Tuning& tuning();

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace simrank

struct simrank_lds_plan;   // lds.hip: SELL / long-row packing for the LDS-tiled kernel
namespace simrank { void free_lds_plan(simrank_lds_plan* p); }

// The graph object: device CSR of the 0/1 pattern + per-row scale, and the transposed
// pattern (CSC) used by the evidence kernel.
struct simrank_graph {
    int64_t n_rows = 0, n_cols = 0, nnz = 0;
    int32_t* rowptr = nullptr;    // [n_rows+1]
    int32_t* col = nullptr;       // [nnz]
    float* rowscale = nullptr;    // [n_rows]
    int32_t* t_rowptr = nullptr;  // [n_cols+1]  transposed pattern
    int32_t* t_col = nullptr;     // [nnz]       row ids, ascending per column
    int32_t max_row_nnz = 0;
    // balanced tiling of the rows (api.hip: build_tiles): tile t = rows [tile_row0[t],
    // tile_row0[t+1]) — 32-row blocks, the heavy ones cut into aligned halves — and, for the
    // upper-triangle form of leg 2, the per-XCD list of (panel, workgroup) pairs to launch
    int32_t* tile_row0 = nullptr;
    int32_t n_tiles = 0;
    int32_t* sym_map = nullptr;
    int32_t sym_blocks = 0;
    // host copies, kept for the lazily built LDS plan (lds.hip)
    std::vector<int32_t> h_rowptr, h_col;
    simrank_lds_plan* lds_plan = nullptr;
    bool lds_plan_failed = false;
};
