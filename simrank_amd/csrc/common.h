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
    int64_t hub = -1;     // rows of X kept in LDS by spmm_hub_kernel: -1 automatic, 0 off, n forced
    int64_t hub_waves = 8;   // waves per hub workgroup (8, 12, 16)
    int64_t hub_rounds = 0;  // tile rounds per hub workgroup (0 = automatic)
};
Tuning& tuning();

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace simrank

struct simrank_lds_plan;   // lds.hip: SELL / long-row packing for the LDS-tiled kernel
namespace simrank {
void free_lds_plan(simrank_lds_plan* p);
int hub_capacity(int n_waves);   // spmm.hip: rows of X a hub workgroup's LDS tile can hold
}

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
    // hub plan (api.hip: build_hub_plan; spmm.hip: spmm_hub_kernel).  When hub_n > 0, `col`
    // lists each row's hub entries first (by slot), then the others (ascending), and
    // `col_enc` is the same list with the hub entries written as -1-slot.
    int32_t hub_n = 0, hub_waves = 0;
    int32_t* hub_ids = nullptr;   // [hub_n] slot -> column
    int32_t* col_enc = nullptr;   // [nnz]
    double hub_share = 0.0;       // fraction of the entries that are hub entries
    // `col` / `col_enc` with every id >= 0 multiplied by scaled_ld / 4 (spmm.hip: scaled_ids)
    int32_t* col_s = nullptr;
    int32_t* col_enc_s = nullptr;
    int64_t scaled_ld = 0;
    int32_t* huge_rows = nullptr; // rows of >= huge_len_built entries, ascending (built on demand)
    int32_t n_huge_rows = 0, huge_len_built = 0;
    // host copies, kept for the lazily built LDS plan (lds.hip)
    std::vector<int32_t> h_rowptr, h_col;
    simrank_lds_plan* lds_plan = nullptr;
    bool lds_plan_failed = false;
};
