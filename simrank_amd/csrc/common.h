// Internal declarations shared by the translation units of libsimrank_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
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
    int64_t balance = 2;  // cut 32-row tiles heavier than balance x the mean tile (0 = uniform tiles); 2 keeps an
                          // XCD on one panel at a time: L2 fills 17.9 -> 10.7 GB per leg 1 at pl32768
    int64_t dense_min = 4;   // block-dense MFMA part: a column joins a 128-row tile's dense set when
                             // at least this many of the tile's rows reference it (0 = off)
    int64_t dense_cols = 128; // ... and a tile gets a dense set only with this many such columns
    int64_t dense_terms = 3; // operand terms of the matrix-core part: 3 = bf16 hi+mid+lo (exact f32 products),
                             // 1 = one fp16 term (reduced precision: config 5's "fp16 MFMA dense leg")
    int64_t probe_mask = -1; // DIAGNOSTIC ONLY: gathered row ids are ANDed with this (wrong results; prices
                             // the memory path of the gather kernel: 255 = L1-resident operand, 8191 = L2-resident)
    int64_t probe_flags = 0; // DIAGNOSTIC ONLY (lean kernel): 1 no gathers, 2 no stores, 4 no dense partial sums, 8 no ids
    int64_t addr32 = 1;      // lean kernel: 32-bit buffer addressing of the operand where its rows span < 2 GiB
    int64_t lean = 1;        // 1: the lean gather kernel (32-float panels, 32-row tiles) wherever it applies,
                             // 0: the generic kernel everywhere (row-major operands only)
    int64_t ids16 = 1;       // stream neighbour ids as 16-bit values when the graph allows it
    int64_t sym_desc = 1;    // upper-triangle leg 2: an XCD takes its panels in descending order — the big ones (N/128
                             // workgroups: one panel at a time in its L2) first, the small ones as the tail: leg 2 -4.7 %
    int64_t ev_tri = 1;      // evidence counts of a whole square block: paths to b >= a only + a mirror pass (round 4)
    int64_t ev_hub = 14;     // ... columns with at least this many thousandths of n_rows live rows (and >= 48) are counted on the
                             // matrix cores (i8 product of their 0/1 image), the rest on the LDS counters (round 5); 0 = off
    int64_t fuse = 1;        // leg 1 of a panel-blocked update as ONE launch (fused.hip): the columns shared by
                             // >= fuse_min rows of a 128-row block on the matrix cores, the rest gathered, by the
                             // same workgroup on the same L2-resident panel slice; 0 = dense_tiles + gather3 launches
    int64_t fuse_min = 0;    // ... a column joins a block's dense set when this many of its rows reference it (3 was the default
                             // of rounds 3-5; 2 and 4: +8 % and +1 % on the leg at pl32768d32); 0 (round 6): by quads that pay
    int64_t fuse_pays = -1;  // ... fuse_min = 0: a block's columns by descending count, 64 (one quad = four matrix-core steps) at a
                             // time, while the quad covers at least this many entries; -1: 192, or 256 where the matrix-core
                             // steps outweigh the gathered remainder (build_fused_plan; profiles/r06_fuse_pays_sweep.log)
    int64_t fuse_steps = -1; // ... and a block keeps its set only when it makes this many 16-column steps; -1: by the size
                             // of a panel's operand slice (fuse_min_steps below)
    int64_t fuse_dens = 0;   // ... or (> 0) when it makes at least 4 and covers this many entries per step
    int64_t fuse_unit = 48;  // ... sets of more than this many 64-column groups are cut into units (workgroups whose
                             // partial sums meet in memory, written through and read past the L1 — round 4; with round 3's
                             // agent-scope release / acquire pair 64 cost +3 %): pl32768d32 leg 1 5.50 -> 5.20 ms at 64
                             // (128: 5.34, 32: 5.23, 16: 5.94); with the gather units of fuse_rows 48 -> 5.0 ms; 1 << 20 = never
    int64_t fuse_rows = 8192; // ... and a block whose gathered remainder exceeds this many entries is cut into gather units
                             // (every n-th row of its descending remainder order each), split blocks only when fuse_unit is on
    int64_t fuse_store = 1;  // ... cache policy of its tile stores: 0 plain, 1 nt, 2 sc1 (write through, drop), 3 sc0 sc1
    int64_t fuse_meta_nt = 0; // ... id streams loaded non-temporally
    int64_t fuse_order = 0;  // ... launch order of a panel's units: 0 heaviest first, k > 0: the units with a matrix-core
                             // phase spread over the first 1/k of the order
    int64_t fuse_max_rows = 1 << 20;  // ... operands with more rows than this keep the two-launch leg (N = 65536: 20.4
                             // against 21.8 ms, so practically never)
    int64_t fuse_cap = 40000;  // (experiment build only, tools/experiments/fused2.hip) piece size of the persistent leg
    int64_t fuse_wgs = 4;    // (experiment build only) its resident workgroups per CU
    int64_t fuse_shards = 1; // ... also for the row-major column block of a sharded rank (result in the all-to-all's chunks)
    int64_t fuse_group = 3;  // ... and up to this many consecutive blocks without a set share a workgroup (1..4)
    int64_t fuse_sym = -1;   // (round 6) leg 2 of a symmetric update as ONE launch too (fused.hip, SYM: matrix cores + gathers +
                             // epilogue + both stores of a tile): 1 wherever it applies, 0 never, -1 where the dense sets hold at
                             // least half of the pattern's entries (MovieLens-shaped graphs: the two-launch leg's second pass)
    int64_t dense_lazy = 0;  // (set by simrank_plan_create, not a knob) the graph's only launch that could use the dense-block
                             // plan is the upper-triangle leg 2: build it only if that leg would take it (dense_sym), i.e. skip
                             // the 256-byte-per-column fragment image and its upload on power-law graphs (8 ms at N = 65536)
    int64_t dense_sym = -1;  // dense part in the upper-triangle form of leg 2: 1 yes, 0 no, -1 = when the
                             // dense sets hold at least half of the pattern's entries
};
Tuning& tuning();            // the process-wide defaults: simrank_set_tuning writes them, simrank_graph_create
Tuning tuning_snapshot();    // copies them (under a lock) into the graph it builds; launches read the copy


// The arrays of a graph object (CSR, plans): allocated, filled once and freed through these.  A sanitizer
// build of the HOST logic (make asan: -DSIMRANK_HOST_ONLY, host side only, no GPU needed) keeps them in
// host memory, so simrank_graph_create — validation, transposition, tiling, dense and fused plans —
// runs under AddressSanitizer / UBSan and tools/host_fuzz.cpp can check the plans it built.
#ifdef SIMRANK_HOST_ONLY
inline hipError_t plan_alloc(void** p, size_t bytes) {
    *p = malloc(bytes ? bytes : 1);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
inline hipError_t plan_upload(void* d, const void* h, size_t bytes) { memcpy(d, h, bytes); return hipSuccess; }
inline hipError_t plan_download(void* h, const void* d, size_t bytes) { memcpy(h, d, bytes); return hipSuccess; }
inline void plan_free(void* p) { free(p); }
#else
// (through the device pool since round 6: the small blocks of a graph object are kept in size classes between fits)
int pool_alloc(void** dptr, size_t bytes);
int pool_free(void* ptr);
inline hipError_t plan_alloc(void** p, size_t bytes) {
    const int rc = pool_alloc(p, bytes);
    return rc == SIMRANK_OK ? hipSuccess : (rc == SIMRANK_ERR_ALLOC ? hipErrorOutOfMemory : hipErrorUnknown);
}
inline hipError_t plan_upload(void* d, const void* h, size_t bytes) { return hipMemcpy(d, h, bytes, hipMemcpyHostToDevice); }
inline hipError_t plan_download(void* h, const void* d, size_t bytes) { return hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost); }
inline void plan_free(void* p) { (void)pool_free(p); }
#endif

// simrank_graph_create with the knobs given (the plans set some per graph: fp16-held fits take no split blocks)
// after_base (optional): called once the CSR / CSC arrays are on the device, while the host plan builders still run on their
// threads — what a plan queues there (the evidence counts, which read those arrays only) runs beside them
int graph_create_with(const Tuning& tun, int64_t n_rows, int64_t n_cols, int64_t nnz, const int32_t* rowptr,
                      const int32_t* col, const float* rowscale, simrank_graph** out,
                      const std::function<int(simrank_graph*)>* after_base = nullptr);

// device block pool (api.hip): what simrank_malloc / simrank_free and the plans allocate through
int pool_alloc(void** dptr, size_t bytes);
int pool_free(void* ptr);
void pool_trim(int device);          // -1: every device
void pool_stats(int device, int64_t* cached_bytes, int64_t* cached_blocks, int64_t* limit_bytes);
void handback_release_slabs(int device);   // handback.hip: the pinned slabs of the f64 hand-back (-1: every device)
// handback.hip: rows `rows` (caller's ids, host) of a panel-blocked n x n matrix held in the solver's order, in the caller's
// column order, into host memory: dst[i][j] = S[inv[rows[i]]][inv[j]].  elem 4: f32, 32-column panels; elem 2: binary16 x
// `scale`, 64-column panels (widened on the device).  What simrank_plan_rows_f32 / simrank_biplan_rows_f32 do.
int rows_to_host(const void* S, int64_t rows_pad, int64_t n, const int32_t* inv_dev, const int32_t* rows, int32_t n_rows,
                 float* dst, int64_t ld, int elem, float scale, hipStream_t stream);
inline hipError_t pool_hip_alloc(void** p, size_t bytes) {       // (for call sites that speak hipError_t)
    const int rc = pool_alloc(p, bytes);
    return rc == SIMRANK_OK ? hipSuccess : (rc == SIMRANK_ERR_ALLOC ? hipErrorOutOfMemory : hipErrorUnknown);
}

// spmm.hip: (W . I)^T = W^T into a panel-blocked matrix of n_cols(g) rows x n_rows(g) columns — leg 1 of a fit's first update
int identity_leg1_blocked(const simrank_graph* g, float* Tt, int64_t t_rows_pad, void* stream);
int identity_leg1_blocked_h16(const simrank_graph* g, uint16_t* Tt, int64_t t_rows_pad, float scale, void* stream);

// planprep.hip: the host-only half of the plans (validation, solver node order, renamed patterns)
struct PlanPrep {
    std::vector<int32_t> ord, inv;          // ord[new] = old, inv[old] = new
    std::vector<int32_t> rp, cl;            // the pattern in the solver's order
    std::vector<float> rs;
    bool asym = false;                      // the prior is not symmetric: un-fused epilogue (SimRank.py:453 on asymmetric iterates)
};
struct BiPlanPrep {
    std::vector<int32_t> rowptr21, col21;   // the group-2 pattern (transpose), caller's order
    std::vector<int32_t> ord[2], inv[2];
    std::vector<int32_t> rp[2], cl[2];      // group w's pattern with both sides renamed
    std::vector<float> rs[2];
    bool asym = false;                      // a prior of either group is not symmetric: both iterates are asymmetric
};
int plan_prepare(int64_t n, int64_t nnz, const int32_t* rowptr, const int32_t* col, const float* rowscale,
                 const simrank_plan_options* opt, PlanPrep* out);
// the sharded plan's host half: the single plan's, plus the ascending order DEALT to `deal` shards in runs of 128 (32)
// nodes when n divides evenly (driver.dealt_order); deal <= 1: plain ascending order
// (allow_asym: a prior that is not symmetric is reported in out->asym instead of refused)
int shard_prepare(int64_t n, int64_t nnz, const int32_t* rowptr, const int32_t* col, const float* rowscale,
                  const float* apriori, int64_t ld_apriori, bool reorder, int32_t deal, PlanPrep* out, bool allow_asym = false);
int shard_biplan_prepare(int64_t n1, int64_t n2, int64_t nnz, const int32_t* rowptr12, const int32_t* col12,
                         const float* rowscale1, const float* rowscale2, const simrank_biplan_options* opt, int32_t deal1,
                         int32_t deal2, BiPlanPrep* out, bool allow_asym = false);
// is a[i][j] == a[j][i] everywhere (NULL: yes)?  What decides between the fused and the un-fused epilogue before a plan is laid out.
bool prior_symmetric(const float* a, int64_t ld, int64_t n);
int biplan_prepare(int64_t n1, int64_t n2, int64_t nnz, const int32_t* rowptr12, const int32_t* col12,
                   const float* rowscale1, const float* rowscale2, const simrank_biplan_options* opt, BiPlanPrep* out);

// How many 16-column steps a block's dense set must make to be kept.  A short matrix-core phase costs a workgroup more in
// fixed work (pattern table, tile hand-over, barriers) than its few shared columns save in gathers — unless the gathers
// are expensive: once a panel's operand slice (K rows x 128 bytes) no longer fits the XCD's 4 MiB L2, more of them miss
// and the break-even moves to smaller sets.  Measured after the round-4 split into matrix-core and gather units
// (profiles/r04_fuse_steps_sweep.log): K = 32768: 20 steps 4.81 ms against 4.97 with 8 (pl32768d32), 4.27 / 4.33 (pl32768);
// K = 65536 (pl65536): 8 steps 18.2 ms against 18.5 with 20.
inline int64_t fuse_min_steps(const Tuning& t, int64_t K) {
    return t.fuse_steps >= 0 ? t.fuse_steps : (K * 128 > (int64_t(4) << 20) ? 8 : 20);
}

// The run loops queue update k + 1 BEFORE they read the count of update k only while an update is short (small graphs:
// the host round trip per update is what they save); from this many nodes on an update takes milliseconds, the round trip
// is noise, and the speculative update would just be 1 / k of the fit thrown away (config 5: 27 ms of 140).
constexpr int64_t kSpeculateBelow = 16384;

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace simrank


// blockdense.hip: the part of the pattern that is dense enough for the matrix cores.  Rows go in
// aligned blocks of 128; inside a block the columns referenced by at least `dense_min` rows form
// the block's DENSE SET.  Those entries are multiplied on MFMA (0/1 pattern in bf16 x the operand
// split into three bf16 terms = exact f32 products, f32 accumulation) into a partial-sum buffer;
// the gather kernel runs on the REMAINDER pattern and adds the partial sums before its epilogue.
struct simrank_dense_plan {
    // A row block's dense set is cut into UNITS of at most 2048 columns (one workgroup per unit
    // and 256 output columns); unit u writes its raw sums into slab unit_slab[u] of the partial
    // buffer, and the slabs of a block are consecutive: block b owns slabs
    // [block_slab0[b], block_slab0[b] + block_nslab[b]).
    int32_t n_units = 0;            // launch order: heaviest first
    int32_t n_slabs = 0;
    int32_t n_blocks_dense = 0;     // 128-row blocks with a dense set
    int64_t total_k = 0;            // sum of the dense set sizes (each unit padded to 16)
    int64_t nnz_covered = 0;        // entries that moved to the dense part
    int32_t* unit_row0 = nullptr;   // [n_units]   first row of the unit's block
    int32_t* unit_slab = nullptr;   // [n_units]
    int32_t* unit_kofs = nullptr;   // [n_units+1] offsets into dcols (multiples of 16), launch order
    int32_t* dcols = nullptr;       // [total_k]   operand rows of the dense sets (padding: 0)
    uint4* afrag = nullptr;         // [total_k/16][4 row blocks][64 lanes] pattern in MFMA A-operand order
    int32_t* block_slab0 = nullptr; // [ceil(n_rows/128)] first slab of a row block
    int32_t* block_nslab = nullptr; // [ceil(n_rows/128)] number of slabs (0 = no dense set)
    // the remainder pattern and its balanced tiling (as simrank_graph's own)
    int64_t r_nnz = 0;
    int32_t r_max_row = 0;
    int32_t* r_rowptr = nullptr;
    int32_t* r_col = nullptr;
    uint16_t* r_col16 = nullptr;    // the same ids in 16 bits when n_cols <= 65536
    int32_t* r_tile_row0 = nullptr;
    int32_t r_n_tiles = 0;
    int32_t* r_sym_map = nullptr;
    int32_t r_sym_blocks = 0;
    // partial sums of the last launch: [n_slabs * 128][ldp] f32, grown on demand.  Calls on one
    // graph must be stream-ordered (one solver per graph object).
    float* part = nullptr;
    size_t part_cap = 0;            // floats
};

// fused.hip: leg 1 as one launch.  Per 128-row block: the dense set (columns referenced by >= fuse_min
// of the block's rows) with its 0/1 pattern as bits in MFMA A-fragment order, the remainder CSR, and
// the order in which the gather phase takes the rows (descending remainder length).
struct simrank_fused_plan {
    int32_t n_units = 0;            // workgroups per panel, launch order: heaviest block first
    int32_t n_blocks = 0;
    int32_t ids16 = 0;              // ids stored in 16 bits (fewer than 65535 operand rows)
    int64_t n_quads = 0;            // 64-column groups of all dense sets
    int64_t n_steps = 0;            // 16-column MFMA steps of all dense sets
    int64_t nnz_covered = 0;        // entries on the matrix cores
    int64_t r_nnz = 0;              // entries gathered
    int32_t* units = nullptr;       // [n_units][32] first block, first quad, quads, index in block, units of the
                                    // block, partial slot, counter slot, block has a set, blocks of the unit, then per
                                    // wave: first round of its id stream, rounds up to the end of each block
    int32_t n_pslots = 0, n_cslots = 0;
    float* partials = nullptr;      // [n_pslots][cap_panels][32 x 128] partial sums of split blocks (grown on demand;
    int32_t* tickets = nullptr;     // [n_cslots][cap_panels]            calls on one graph are stream-ordered)
    int32_t cap_panels = 0;
    uint16_t* dcols16 = nullptr;    // [n_quads*64] operand rows of the sets, or
    int32_t* dcols32 = nullptr;     //              the same in 32 bits; padding = a real row, bits zero
    uint4* abits = nullptr;         // [n_quads*64] per lane: 4 steps x (4 row tiles x 8 k) pattern bits
    int2* gmeta = nullptr;          // [n_blocks*32*4] per block, wave, lane group and row: (end of the row in the
                                    // group's stream << 8 | row of the block; 255: no row, 0xFFFFFF: no remainder),
                                    // rowscale bits
    uint16_t* sids16 = nullptr;     // gather id stream, 64 per round (0xFFFF: no neighbour), or
    int32_t* sids32 = nullptr;      //              32-bit ids (-1: no neighbour)
};

#ifdef SIMRANK_EXPERIMENT_FUSED2
// tools/experiments/fused2.hip (NOT in the product library; tools/build_variant.sh ... -DSIMRANK_EXPERIMENT_FUSED2): leg 1 as
// one PERSISTENT launch: the same dense sets and id streams, per PIECE of a block (a share of the set's columns + a share of
// the rows), pulled from per-XCD queues by resident workgroups.  Round 4: correct, slower than fused.hip (HISTORY.md).
struct simrank_fused2_plan {
    int32_t n_items = 0;            // pieces per panel, launch order
    int32_t n_blocks = 0;
    int32_t ids16 = 0;
    int32_t cap_panels = 0;         // panels the partial-sum slots are sized for
    int64_t n_quads = 0, n_steps = 0, nnz_covered = 0, r_nnz = 0;
    int32_t* items = nullptr;       // [n_items + 1][16]
    int32_t n_pslots = 0, n_cslots = 0;
    float* partials = nullptr;      // [n_pslots][cap_panels][32 x 128]
    int32_t* tickets = nullptr;     // [n_cslots][cap_panels]
    uint32_t* heads = nullptr;      // [8][32]
    uint16_t* dcols16 = nullptr;
    int32_t* dcols32 = nullptr;
    uint4* abits = nullptr;
    int2* gmeta = nullptr;
    uint16_t* sids16 = nullptr;
    int32_t* sids32 = nullptr;
};
#endif

namespace simrank {
constexpr int kFB = 128;          // rows per block of the one-launch plan
constexpr int kSub = 4;           // blocks a unit without a dense set may hold
void free_fused_plan(simrank_fused_plan* p);
int build_fused_plan(simrank_graph* g, const int32_t* rowptr, const int32_t* col, const float* rowscale);
int launch_fused_trans(const simrank_graph* g, const float* X, int64_t x_rows_pad, int64_t L, float* Y,
                       int64_t y_rows_pad, hipStream_t st);
bool fused_sym_applies(const simrank_graph* g, int64_t x_rows_pad, int64_t L, int64_t y_rows_pad);
int launch_fused_sym(const simrank_graph* g, const float* X, int64_t x_rows_pad, int64_t L, float* Y, int64_t y_rows_pad,
                     float coef, float lbd, double eps, const uint8_t* ev, const float* ap, const float* prev,
                     unsigned long long* n_changed, int32_t set_diag, int32_t count_any, hipStream_t st);
bool fused_rowmajor_fits(const simrank_graph* g, const float* X, int64_t ldx, int64_t L, const float* Y, int64_t t_block,
                         int64_t t_pad);
int launch_fused_trans_rowmajor(const simrank_graph* g, const float* X, int64_t ldx, int64_t L, float* Y, int64_t t_block,
                                int64_t t_pad, hipStream_t st);
#ifdef SIMRANK_EXPERIMENT_FUSED2
void free_fused2_plan(simrank_fused2_plan* p);
int build_fused2_plan(simrank_graph* g, const int32_t* rowptr, const int32_t* col, const float* rowscale);
int launch_fused2_trans(const simrank_graph* g, const float* X, int64_t x_rows_pad, int64_t L, float* Y,
                        int64_t y_rows_pad, hipStream_t st);
#endif
struct DenseUse {                   // what the gather kernel needs from a dense launch
    const float* part = nullptr;
    int64_t ldp = 0;
    const int32_t* block_slab0 = nullptr;
    const int32_t* block_nslab = nullptr;
};
void free_dense_plan(simrank_dense_plan* p);
int build_dense_plan(simrank_graph* g, const int32_t* rowptr, const int32_t* col);
int launch_dense_tiles(const simrank_graph* g, const float* X, int64_t ldx, int64_t L, bool tri,
                       hipStream_t st, DenseUse* use);
// balanced 32-row tiling of a pattern (api.hip): tile list and the upper-triangle launch list
// returns the number of tiles; tile_row0 = n_tiles + 1 row offsets, then the launch order of the
// ceil(n_tiles / 4) workgroup groups (most entries first)
int64_t build_tiles(const int32_t* rowptr, int64_t n_rows, int64_t nnz, int64_t balance,
                    std::vector<int32_t>& tile_row0, std::vector<int32_t>& sym_map, bool sym_descending);
}

// The graph object: device CSR of the 0/1 pattern + per-row scale, and the transposed
// pattern (CSC) used by the evidence kernel.
struct simrank_graph {
    int64_t n_rows = 0, n_cols = 0, nnz = 0;
    int32_t* rowptr = nullptr;    // [n_rows+1]
    int32_t* col = nullptr;       // [nnz]
    uint16_t* col16 = nullptr;    // [nnz] the same ids in 16 bits when n_cols <= 65536 (what the gather
                                  // kernel streams once per panel: half the bytes, half the L2 lines)
    float* rowscale = nullptr;    // [n_rows]
    int32_t* t_rowptr = nullptr;  // [n_cols+1]  transposed pattern
    int32_t* t_col = nullptr;     // [nnz]       row ids, ascending per column
    int32_t* t_pos = nullptr;     // [nnz]       slot of CSR entry j in its column's list (live rows only)
    int32_t* ev_hubidx = nullptr; // [n_cols]    index of a HUB column in the 0/1 image of the evidence counts, -1: none
    int32_t ev_hubs = 0;          // columns of that image (a multiple of 32; 0: no hub columns)
    uint8_t* ev_hub_image = nullptr;  // [rows padded to 128][ev_hubs] the image itself, built at the first evidence call
    void* ev_hub_ready = nullptr;     // hipEvent_t recorded behind that build: an evidence call on ANOTHER stream waits for it
    int32_t max_row_nnz = 0;
    // balanced tiling of the rows (api.hip: build_tiles): tile t = rows [tile_row0[t],
    // tile_row0[t+1]) — 32-row blocks, the heavy ones cut into aligned halves — and, for the
    // upper-triangle form of leg 2, the per-XCD list of (panel, workgroup) pairs to launch
    int32_t* tile_row0 = nullptr;
    int32_t n_tiles = 0;
    int32_t* sym_map = nullptr;
    int32_t sym_blocks = 0;
    simrank_dense_plan* dense = nullptr;   // NULL: no block of the pattern is dense enough
    simrank_fused_plan* fused = nullptr;   // leg 1 as one launch (fused.hip); NULL: tuning "fuse" = 0
    std::atomic<int> fused_build{0};       // its builder thread during simrank_graph_create: 0 none, 1 running, 2 finished (the
                                           // dense-block builder of a lazy graph waits for it: the one-launch leg 2 may take its place)
#ifdef SIMRANK_EXPERIMENT_FUSED2
    simrank_fused2_plan* fused2 = nullptr; // leg 1 as one persistent launch (experiment build); tuning "fuse" = 2
#endif
    simrank::Tuning tun;                   // knobs in force when the graph was created (every launch on
                                           // this graph uses these, whatever is set afterwards)
};
