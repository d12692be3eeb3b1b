/*
 * simrank_hip.h — C ABI of the MI355X (gfx950) SimRank / SimRank++ iteration engine.
 *
 * The reference (ysong1231/SimRank) has no FFI: its hot path is inline NumPy inside the
 * `fit` methods of SimRank/SimRank.py.  This header is what a binding for that path
 * binds instead; each entry point names the reference lines it replaces.  The Python
 * host (simrank_amd/engine.py, ctypes) is the only caller in this repository, and
 * INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch types.
 *   - every function returns 0 on success or a negative simrank_status; the message of
 *     the last failure on the calling thread is simrank_last_error().
 *   - "device pointer" = HIP device memory of the current device (from simrank_malloc or
 *     from any other allocator, e.g. torch.Tensor.data_ptr()).  Host buffers are caller
 *     owned; device buffers passed in are caller owned; a simrank_graph owns its CSR copy.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  All work is
 *     asynchronous on that stream unless a function says it synchronises.
 *   - matrices are row-major float32 with an explicit leading dimension `ld` (elements).
 *     The fast (16-byte vector) kernels run when pointers are 16-byte aligned and every
 *     ld is a multiple of 4; otherwise a scalar variant runs (same results).
 *
 * Data model
 *   W = diag(rowscale) . A, A the 0/1 pattern of a CSR matrix with n_rows x n_cols.
 *   Every normalised adjacency the reference builds has this form (SimRank.py:45-52,
 *   :191-200: each stored value is 1/in-degree or 1/sum-of-weights of its ROW node; the
 *   SimRank++ "spread" factor of :326-333 is again per row).
 *   One similarity update  S' = coef . W . S . W^T (.*E) (+ lbd.A), diag <- 1  is two calls
 *   of simrank_spmm:   Tt = (W . S)^T      (transpose_out = 1, no epilogue)
 *                      S' = W . Tt          (epilogue)            [S symmetric]
 *   or, for dense graphs, simrank_spmm / simrank_gemm_nt on a densified W (MFMA).
 */
#ifndef SIMRANK_HIP_H
#define SIMRANK_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SIMRANK_ABI_VERSION 7
#define SIMRANK_CHANGED_SLOTS 1024

#if defined(__GNUC__)
#define SIMRANK_API __attribute__((visibility("default")))
#else
#define SIMRANK_API
#endif

typedef enum simrank_status {
    SIMRANK_OK = 0,
    SIMRANK_ERR_INVALID = -1,   /* bad argument (shape, NULL, alignment the call requires) */
    SIMRANK_ERR_HIP = -2,       /* a HIP runtime call failed; see simrank_last_error()      */
    SIMRANK_ERR_NO_DEVICE = -3, /* no gfx950 device visible                                 */
    SIMRANK_ERR_ALLOC = -4
} simrank_status;

SIMRANK_API int simrank_abi_version(void);
SIMRANK_API const char* simrank_last_error(void);

/* ---- device, memory, stream, event plumbing (thin HIP wrappers) ------------------- */
SIMRANK_API int simrank_device_count(int* count);
SIMRANK_API int simrank_set_device(int device);
SIMRANK_API int simrank_device_info(int device, char* name, int name_len, int64_t* total_bytes,
                        int* compute_units, char* arch, int arch_len);
/* Device memory.  Blocks of at least 64 MiB come from / go back to a per-device pool inside the library
 * (hipMalloc maps a 17 GiB similarity matrix in up to 0.5 s; the reference pays nothing comparable for
 * np.zeros at SimRank.py:124-125): a request takes the smallest cached block of its size up to 1/8 more,
 * simrank_free synchronises the device (as hipFree does) and keeps the block, least recently freed blocks
 * leave when more than SIMRANK_POOL_GIB (environment, default 56: what one N = 65536 plan holds) GiB are at rest, and an allocation that
 * fails empties the pool and is tried once more.  Smaller blocks (the arrays of a graph object and its plans) are kept in size
 * classes (2^k x {1, 1.25, 1.5, 1.75} bytes from 512; at most 1 GiB at rest per device; SIMRANK_POOL_GIB=0 turns every kind
 * of pooling off).  The plans (simrank_plan_*, simrank_biplan_*) allocate
 * their matrices and scratch through the same pool.  Cached blocks are invisible to other allocators of
 * the process: simrank_pool_trim(device) (-1 = every device) returns them to the driver. */
SIMRANK_API int simrank_malloc(void** dptr, size_t bytes);
SIMRANK_API int simrank_free(void* dptr);
SIMRANK_API int simrank_pool_trim(int device);
SIMRANK_API int simrank_pool_stats(int device, int64_t* cached_bytes, int64_t* cached_blocks, int64_t* limit_bytes);
SIMRANK_API int simrank_memset(void* dptr, int byte_value, size_t bytes, void* stream);
/* copies are enqueued on `stream` and the call returns after they completed */
SIMRANK_API int simrank_memcpy_h2d(void* dst_device, const void* src_host, size_t bytes, void* stream);
SIMRANK_API int simrank_memcpy_d2h(void* dst_host, const void* src_device, size_t bytes, void* stream);
SIMRANK_API int simrank_memcpy_d2d(void* dst_device, const void* src_device, size_t bytes, void* stream);
/* 2-D device->host copy of a row-major block, converting float32 -> float64 on the host
 * side (replaces the DataFrame hand-back of SimRank.py:141, :303). */
SIMRANK_API int simrank_download_f64(double* dst_host, int64_t ld_dst, const float* src_device,
                         int64_t ld_src, int64_t n_rows, int64_t n_cols, void* stream);
/* Device->host hand-back of an n x n float32 matrix as float64 in another node order (replaces `pd.DataFrame(new_S)` of
 * SimRank.py:141, :303): dst[i][j] = src[idx[i]][idx[j]] (idx: DEVICE int32[n], NULL = identity; src row-major with
 * ld_src, or panel-blocked when src_rows_pad > 0).  Pipelined band by band: re-ordering on the device, PCIe and the
 * widening on the host overlap.  mode 0: every element crosses PCIe.  mode 1 (SYMMETRIC form): the matrix is checked on
 * the device to be bitwise mirror-equal outside the 32 x 32 diagonal blocks of the source order (what the upper-triangle
 * leg 2 leaves); if so only the elements on or above the diagonal of the result cross PCIe and the host writes each
 * twice, the diagonal blocks come over whole; if not, mode 0 runs.  Same bits either way.  (On a host share of 16 CPUs
 * mode 1 is the slower one — the mirrored writes cost more than the PCIe bytes they save; csrc/handback.hip.) */
SIMRANK_API int simrank_handback_f64(double* dst_host, int64_t ld_dst, const float* src_device, int64_t ld_src,
                                     int64_t src_rows_pad, int64_t n, const int32_t* idx_device, int32_t mode, void* stream);
/* The convergence count of an update without stopping the stream (`_converged`, SimRank.py:54-77, as a caller
 * that drives the legs itself reads it): a COUNTER SET owns four pinned slots and their events; _fetch queues the copy
 * of n counters (the simrank_epilogue.n_changed array) into slot `slot` (0..3) and records an event behind it; _wait
 * returns their sum once that copy has landed, while whatever was queued behind the fetch — update k + 1 of the loop
 * at :129-140, issued before the count of update k is known — keeps running.  One set per engine (stream): two fits
 * running side by side on one device never see each other's counts. */
typedef struct simrank_counter_set simrank_counter_set;
SIMRANK_API int simrank_counters_create(simrank_counter_set** out);
SIMRANK_API int simrank_counters_destroy(simrank_counter_set* set);
SIMRANK_API int simrank_counters_fetch(simrank_counter_set* set, const unsigned long long* counters, int32_t n, int32_t slot,
                                       void* stream);
SIMRANK_API int simrank_counters_wait(simrank_counter_set* set, int32_t slot, unsigned long long* sum);
SIMRANK_API int simrank_stream_create(void** stream);
SIMRANK_API int simrank_stream_destroy(void* stream);
SIMRANK_API int simrank_stream_synchronize(void* stream);
SIMRANK_API int simrank_event_create(void** event);
SIMRANK_API int simrank_event_destroy(void* event);
SIMRANK_API int simrank_event_record(void* event, void* stream);
SIMRANK_API int simrank_event_synchronize(void* event);   /* host waits for the recorded point */
/* waits for `stop`, then returns the time between the two records in milliseconds */
SIMRANK_API int simrank_event_elapsed_ms(void* start, void* stop, float* ms);

/* ---- graph: normalised adjacency in CSR (replaces the dense `Graph` DataFrame built at
 *      SimRank.py:43-52 and the pivots at :199-200 / :390-391) ---------------------- */
typedef struct simrank_graph simrank_graph;

/* rowptr[n_rows+1], col[nnz] (column indices ascending within a row, < n_cols),
 * rowscale[n_rows]: host arrays, copied.  Builds the transposed pattern (CSC) too. */
SIMRANK_API int simrank_graph_create(int64_t n_rows, int64_t n_cols, int64_t nnz, const int32_t* rowptr,
                         const int32_t* col, const float* rowscale, simrank_graph** out);
/* The same, with the common-in-neighbour counts of the pattern (simrank_evidence_counts / _blocked below: columns
 * [col0, col0 + n_cols_ev) into `counts`, zeroed by the caller; rows_pad > 0: panel-blocked, ld ignored) queued on `stream` as
 * soon as the pattern is on the device: the counting kernel runs beside the host threads that build the graph's plans. */
SIMRANK_API int simrank_graph_create_counting(int64_t n_rows, int64_t n_cols, int64_t nnz, const int32_t* rowptr,
                                              const int32_t* col, const float* rowscale, int64_t col0, int64_t n_cols_ev,
                                              uint8_t* counts, int64_t ld, int64_t rows_pad, void* stream,
                                              simrank_graph** out);
SIMRANK_API int simrank_graph_destroy(simrank_graph* g);
SIMRANK_API int simrank_graph_shape(const simrank_graph* g, int64_t* n_rows, int64_t* n_cols, int64_t* nnz);
/* Operand terms of the matrix-core part for THIS graph (3 = exact f32 products from three bf16 terms, the
 * default; 1 = one fp16 term: BASELINE.json config 5's reduced-precision dense leg, outside the 1e-5 parity
 * bar).  Per graph, so that two fits of one process can differ (the process-wide "dense_terms" knob is
 * only the default a graph starts with).  With 1 the legs run as dense_tiles + gather launches. */
SIMRANK_API int simrank_graph_set_dense_terms(simrank_graph* g, int32_t terms);

/* ---- K0: S[:, block] <- columns [col0, col0+n_cols) of the identity
 *      (SimRank.py:124-126, :280-285, :346-348, :402-407) --------------------------- */
SIMRANK_API int simrank_fill_identity(float* S, int64_t n_rows, int64_t n_cols, int64_t ld, int64_t col0,
                          void* stream);

/* ---- fused epilogue of an update: scale, evidence, prior, diagonal, convergence count
 *      (SimRank.py:139-140, :361-362, :453-454 and `_converged` :74) ---------------- */
typedef struct simrank_epilogue {
    float coef;                      /* C (C1/C2)                                         */
    float lbd;                       /* prior blend, used only when apriori != NULL       */
    const uint8_t* evidence;         /* device, common-in-neighbour counts (saturated at
                                        255) of this column block, or NULL: value is
                                        multiplied by 1 - 2^-count  (SimRank.py:315-316)  */
    int64_t ld_evidence;
    const float* apriori;            /* device, prior block or NULL:
                                        v <- (1-lbd).v + lbd.apriori  (SimRank.py:453)    */
    int64_t ld_apriori;
    const float* previous;           /* device, previous iterate block or NULL            */
    int64_t ld_previous;
    double eps;                      /* strict |new - previous| > eps, every element      */
    unsigned long long* n_changed;   /* device array of SIMRANK_CHANGED_SLOTS counters, zeroed
                                        by the call; the count is their SUM (striped so the
                                        workgroups do not serialise on one address)       */
    int64_t diag_col0;               /* global column index of local column 0             */
    int32_t set_diag;                /* 1: element (a, a - diag_col0) <- 1                */
    int32_t symmetric;               /* 1: the block is the whole n x n result and evidence,
                                        prior and previous iterate are exactly symmetric: the
                                        kernel may compute the upper triangle only and store
                                        its mirror image (off-diagonal tiles come out exactly
                                        symmetric; n_changed counts mirrored elements twice) */
    int32_t restrict_support;        /* 1 (with evidence): the gathers of a row's 32-column segment
                                        are skipped when all its evidence counts are zero — the
                                        product is multiplied by E = 0 there (SimRank.py:315-316,
                                        :361: S stays inside supp(E)); same bits, fewer gathers.
                                        Worth it when few segments are live
                                        (simrank_evidence_live_segments) */
    int32_t count_any;               /* 1: the caller only needs to know WHETHER any element moved by
                                        more than eps — what `_converged` returns (SimRank.py:74-77:
                                        the sum is used as a truth value).  The lean gather kernel
                                        then stops comparing once a difference has been found: a
                                        wave that finds its own striped counter non-zero when it
                                        starts skips the reads of `previous` (half a pass over S per
                                        update; ~0.5 % of the waves still compare).  The
                                        sum of the counters is 0 exactly when nothing moved and >= 1
                                        otherwise, no longer the exact count; the result block is
                                        the same bits.  0: exact count (the other kernels always) */
} simrank_epilogue;

/* ---- the convergence count on the host: sum of the n striped counters an epilogue wrote
 *      (`_converged`, SimRank.py:74, called once per loop index at :130).  Synchronises the
 *      stream; the copy goes through pinned memory. */
SIMRANK_API int simrank_read_counters(const unsigned long long* counters, int32_t n,
                                      unsigned long long* sum, void* stream);

/* ---- K2 (+K3 sparse form, K4, K5): Y = diag(rowscale).A.X with optional transposed
 *      store and fused epilogue.  X: n_cols(g) x n_cols_x, Y: n_rows(g) x n_cols_x.
 *      transpose_out = 0: Y[a*ldy + c]
 *      transpose_out = 1: rows are grouped in blocks of t_block rows (t_block <= 0 or
 *        >= n_rows(g): one block); block h holds Y^T of its rows, contiguous, its rows
 *        padded by t_pad floats (so the receiver's leading dimension need not be a power
 *        of two):
 *        Y[h*n_cols_x*(t_block+t_pad) + c*(rows_in_block(h)+t_pad) + (a - h*t_block)]
 *        (ldy is ignored, except that with a single block and ldy >= n_rows(g) it is the
 *        row stride of Y^T: Y[c*ldy + a])
 *      This is the layout an all-to-all between column-sharded ranks needs.
 *      (first `.dot` of SimRank.py:139/:298/:301/:361/:420/:423; with the epilogue also
 *      the second, using S' = W.(W.S)^T for symmetric S.) */
SIMRANK_API int simrank_spmm(const simrank_graph* g, const float* X, int64_t ldx, int64_t n_cols_x,
                 float* Y, int64_t ldy, int32_t transpose_out, int64_t t_block, int64_t t_pad,
                 const simrank_epilogue* epilogue, void* stream);

/* ---- leg 2 of ONE RANK of a sharded symmetric update, half the gathers.  The graph has
 *      n_rows = world * mb rows, mb a multiple of 32; the rank owns columns [rank*mb, (rank+1)*mb)
 *      (epilogue->diag_col0 = rank*mb); X is the n_cols(g) x mb operand, Y the n_rows x mb block.
 *      For the 32-row tile i of shard h and the rank's column tile j only i <= j is computed; for
 *      i < j the transposed tile is also stored — inside Y when h == rank, else packed into
 *      send[h*chunk_floats + (j(j-1)/2 + i)*1024 ...] (32 x 32 floats, row = the rank's column).
 *      After an all-to-all of the chunks, simrank_shard_unpack puts what rank h sent at rows
 *      (h, j) x columns (rank, i) of Y.  The node order must make every shard an equal mix of short
 *      and long rows (driver.dealt_order) for the work to halve evenly.  n_changed counts mirrored
 *      elements twice.  Replaces what simrank_spmm with an epilogue replaces — the second `.dot(G.T)`
 *      with `C *`, `Evidence *`, the prior blend and `fill_diagonal` of SimRank.py:139-140, :361-362,
 *      :453-454 and the count of `_converged` (:74) — for S split by column block over `world` GPUs. */
SIMRANK_API int simrank_spmm_shard(const simrank_graph* g, const float* X, int64_t ldx, float* Y,
                                   int64_t ldy, const simrank_epilogue* epilogue, int32_t rank,
                                   int32_t world, float* send, int64_t chunk_floats, void* stream);
SIMRANK_API int simrank_shard_unpack(float* Y, int64_t ldy, const float* recv, int64_t chunk_floats,
                                     int32_t rank, int32_t world, int64_t n_rows, void* stream);
/* One STAGE of the half-form leg 2: the column tiles [tile_lo, tile_hi) of the rank's block only.  The packed
 * mirrored tiles of those columns occupy slots [tile_lo (tile_lo - 1) / 2, tile_hi (tile_hi - 1) / 2) of a
 * destination's chunk; `send_stage` / `recv_stage` hold just that range for every rank (stage_chunk_floats
 * per rank), so the second all-to-all can be cut into stages too and a stage's tiles leave while the next
 * stage computes (driver.Side: exchange 2 overlapped with leg 2).  The striped counters are
 * zeroed when zero_counters is set (first stage of an update) and accumulate otherwise. */
SIMRANK_API int simrank_spmm_shard_stage(const simrank_graph* g, const float* X, int64_t ldx, float* Y, int64_t ldy,
                                         const simrank_epilogue* ep, int32_t rank, int32_t world, float* send_stage,
                                         int64_t stage_chunk_floats, int32_t tile_lo, int32_t tile_hi,
                                         int32_t zero_counters, void* stream);
SIMRANK_API int simrank_shard_unpack_stage(float* Y, int64_t ldy, const float* recv_stage, int64_t stage_chunk_floats,
                                           int32_t rank, int32_t world, int64_t n_rows, int32_t tile_lo,
                                           int32_t tile_hi, void* stream);

/* ---- K4/K5 alone: Y = epilogue(Q), element-wise over an n_rows x n_cols block (Q and Y may be
 *      the same buffer).  Used when an update cannot fuse its epilogue into leg 2: a prior
 *      that is not symmetric makes S asymmetric, and leg 2 must then store its product
 *      transposed (no epilogue) before SimRank.py:453 is applied. */
SIMRANK_API int simrank_epilogue_apply(const float* Q, int64_t ldq, float* Y, int64_t ldy,
                                       int64_t n_rows, int64_t n_cols,
                                       const simrank_epilogue* epilogue, void* stream);

/* ---- partial hand-back: the k largest entries of every row of an n_rows x n_cols block,
 *      largest first, ties by lower column; with exclude_diag the element (a, a - col0) is
 *      skipped.  idx_out[a*k + j] is a GLOBAL column (col0 + c), -1 when the row has fewer
 *      than k candidates.  Lets a caller take "the 10 most similar nodes" of every node
 *      without moving N^2 floats over PCIe (the reference returns the dense matrix,
 *      SimRank.py:141). */
SIMRANK_API int simrank_topk_rows(const float* S, int64_t ld, int64_t n_rows, int64_t n_cols,
                                  int64_t col0, int32_t k, int32_t exclude_diag,
                                  int32_t* idx_out, float* val_out, void* stream);

/* The same with the ids of the block's columns given explicitly (device array col_ids[n_cols];
 * NULL = col0 + c): ids are what is reported and what breaks ties.  For callers that keep the
 * matrix in a permuted node order (the host side sorts nodes by row length). */
SIMRANK_API int simrank_topk_rows_ids(const float* S, int64_t ld, int64_t n_rows, int64_t n_cols,
                                      int64_t col0, const int32_t* col_ids, int32_t k,
                                      int32_t exclude_diag, int32_t* idx_out, float* val_out,
                                      void* stream);

/* ---- dst[i][j] = src[row_idx[i]][col_idx[j]] for an n_rows x n_cols destination; a NULL
 *      index list is the identity.  elem_bytes 4 (f32) or 1 (u8 evidence counts).  Moves
 *      priors into, and results out of, the permuted node order; the reference has no
 *      counterpart (it keeps list(self.Nodes) order throughout, SimRank.py:43,141). */
SIMRANK_API int simrank_permute(const void* src, int64_t ld_src, void* dst, int64_t ld_dst,
                                int64_t n_rows, int64_t n_cols, const int32_t* row_idx,
                                const int32_t* col_idx, int32_t elem_bytes, void* stream);

/* ---- K7: counts of common in-neighbours, saturated at 255, for columns
 *      [col0, col0+n_cols) of the n_rows x n_rows evidence matrix; only rows with
 *      rowscale > 0 take part (pattern G > 0).  Replaces the int64 matmul of
 *      SimRank.py:315; E = 1 - 2^-count is applied in the epilogue (:316). */
SIMRANK_API int simrank_evidence_counts(const simrank_graph* g, int64_t col0, int64_t n_cols,
                            uint8_t* counts, int64_t ld, void* stream);

/* ---- how many aligned 32-column segments of a u8 count block hold a nonzero count (`live`) out
 *      of `total` = n_rows * ceil(n_cols / 32): the support density of the evidence matrix
 *      E = 1 - 2^-count (SimRank.py:315-316) at the granularity simrank_spmm can skip.
 *      Synchronises the stream. */
SIMRANK_API int simrank_evidence_live_segments(const uint8_t* counts, int64_t ld, int64_t rows_pad,
                                               int64_t n_rows, int64_t n_cols, int64_t* live,
                                               int64_t* total, void* stream);
/* (rows_pad > 0: the counts are panel-blocked, see below; ld is then ignored) */

/* ---- PANEL-BLOCKED operands.  A matrix of R rows x C columns is stored as ceil(C/32) panels of
 *      rows_pad >= R rows x 32 elements; element (r, c) sits at
 *          ((c >> 5) * rows_pad + r) * 32 + (c & 31)          (floats; bytes for the u8 counts).
 *      What the gather legs read per 32-column panel — one 128-byte segment of every source row —
 *      is then ONE contiguous slice (4 MiB at N = 32768) instead of N segments 128 KiB apart: the
 *      gathers stop missing the TLB (measured on one XCD, K = 32768: 3.3 vs 1.7 TB/s) and every
 *      tile a leg stores is a contiguous 4 KiB.  The single-rank solver keeps S, the transposed
 *      product, the evidence counts and the prior in this layout; results leave it through
 *      simrank_permute_layout.  Same reference lines as the row-major entry points. */
SIMRANK_API int simrank_fill_identity_blocked(float* S, int64_t n_rows, int64_t n_cols,
                                              int64_t rows_pad, int64_t col0, void* stream);
/* X: n_cols(g) rows (x_rows_pad per panel) x n_cols_x columns.  transpose_out = 0: Y is n_rows(g)
 * rows x n_cols_x columns and the epilogue operands share its layout (their ld_* are ignored);
 * transpose_out = 1: Y is n_cols_x rows x n_rows(g) columns.  y_rows_pad = padded rows of Y.
 * Which launches serve a call (tuning, below): transpose_out = 1 (leg 1, first `.dot` of SimRank.py:139) is ONE launch,
 * matrix cores + gathers ("fuse"); transpose_out = 0 with epilogue.symmetric (leg 2, the second `.dot` and the element-wise
 * lines) is the upper-triangle gather + mirror ("triangle"), or the same ONE launch as leg 1 with the epilogue applied to
 * the tile where the graph's dense sets hold most of its entries ("fuse_sym"). */
SIMRANK_API int simrank_spmm_blocked(const simrank_graph* g, const float* X, int64_t x_rows_pad,
                                     int64_t n_cols_x, float* Y, int64_t y_rows_pad,
                                     int32_t transpose_out, const simrank_epilogue* epilogue,
                                     void* stream);
SIMRANK_API int simrank_epilogue_apply_blocked(const float* Q, float* Y, int64_t n_rows, int64_t n_cols,
                                               int64_t rows_pad, const simrank_epilogue* epilogue,
                                               void* stream);
SIMRANK_API int simrank_topk_rows_blocked(const float* S, int64_t rows_pad, int64_t n_rows,
                                          int64_t n_cols, int64_t col0, const int32_t* col_ids,
                                          int32_t k, int32_t exclude_diag, int32_t* idx_out,
                                          float* val_out, void* stream);
SIMRANK_API int simrank_evidence_counts_blocked(const simrank_graph* g, int64_t col0, int64_t n_cols,
                                                uint8_t* counts, int64_t rows_pad, void* stream);
/* simrank_permute with a layout per side: *_rows_pad = 0 row-major (ld_* used), > 0 panel-blocked */
SIMRANK_API int simrank_permute_layout(const void* src, int64_t ld_src, int64_t src_rows_pad,
                                       void* dst, int64_t ld_dst, int64_t dst_rows_pad,
                                       int64_t n_rows, int64_t n_cols, const int32_t* row_idx,
                                       const int32_t* col_idx, int32_t elem_bytes, void* stream);

/* ---- FP16 STORAGE (round 3; BASELINE.json config 5's reduced-precision mode, both `.dot`s of
 *      SimRank.py:361 and the element-wise lines :315-316, :362, :453 on matrices held in fp16).
 *      NEVER the default and outside the 1e-5 parity bar: S, the transposed product and the previous
 *      iterate are IEEE binary16, panel-blocked with 64-COLUMN panels (element (r, c) at
 *      ((c >> 6) * rows_pad + r) * 64 + (c & 63): a row segment is one 128-byte line, as in the f32
 *      layout, and serves twice the columns); sums and the epilogue are f32, a value is rounded (nearest
 *      even) once, when stored; the convergence count compares the rounded values.  One GPU, symmetric
 *      iterates, graphs created with tuning "fuse" = 1 (half.hip).
 *      simrank_spmm_blocked_h16: transpose_out = 1 and epilogue = NULL is leg 1 (Y = transposed
 *      product); transpose_out = 0 with an epilogue (symmetric = 1, diag_col0 = 0) is leg 2.  The
 *      epilogue's evidence counts and prior keep the f32 era's 32-column panels of aux_rows_pad rows;
 *      its `previous` is an fp16 matrix laid out like Y.
 *      simrank_widen_blocked_h16 converts to the f32 panel-blocked layout every hand-back entry reads.
 *      SCALE: similarities of a large sparse graph are mostly far below fp16's normal range (6.1e-5),
 *      so the matrices hold value x scale, scale a power of two in 1 .. 32768 (the solver uses 16384:
 *      the diagonal 1.0 is stored as 16384, the smallest normal number stands for 3.7e-9).  Both legs are
 *      linear, so only the diagonal, the prior (x scale), eps (x scale) and the hand-back (/ scale) see it. */
SIMRANK_API int simrank_fill_identity_blocked_h16(void* S, int64_t n_rows, int64_t n_cols,
                                                  int64_t rows_pad, int64_t col0, float scale,
                                                  void* stream);
SIMRANK_API int simrank_spmm_blocked_h16(const simrank_graph* g, const void* X, int64_t x_rows_pad,
                                         int64_t n_cols_x, void* Y, int64_t y_rows_pad,
                                         int32_t transpose_out, const simrank_epilogue* epilogue,
                                         int64_t aux_rows_pad, float scale, void* stream);
SIMRANK_API int simrank_widen_blocked_h16(const void* src, int64_t src_rows_pad, float* dst,
                                          int64_t dst_rows_pad, int64_t n_rows, int64_t n_cols,
                                          float scale, void* stream);
/* Flat conversions f32 <-> fp16 (value x scale, nearest even, saturating at fp16's largest finite value): the WIRE
 * format of a sharded update's exchange buffers — the transposed product of SimRank.py:139's first `.dot` on its way to
 * the rank that needs it, and the mirrored tiles of the second — when the caller trades precision for link bytes
 * (driver: TorchWorld(exchange_precision="fp16"); the kernels on both sides stay f32; outside the 1e-5 parity bar,
 * never the default).  n elements (16 bytes per lane where both operands are 16-byte aligned), scale a power of two in
 * 1 .. 32768. */
SIMRANK_API int simrank_narrow_h16(const float* src_device, void* dst_device_fp16, int64_t n, float scale, void* stream);
SIMRANK_API int simrank_widen_h16(const void* src_device_fp16, float* dst_device, int64_t n, float scale, void* stream);

/* ---- dense MFMA path (second `.dot(G.T)` of SimRank.py:139 when W really is dense) -- */
/* Wd[a*ld + i] = rowscale[a] where (a,i) is stored, 0 elsewhere */
SIMRANK_API int simrank_graph_densify(const simrank_graph* g, float* Wd, int64_t ld, void* stream);
/* C[M x N] = epilogue(A[M x K] . B[N x K]^T) on v_mfma_f32_32x32x2_f32 (exact f32).
 * epilogue may be NULL (plain product).  Requires 16-byte aligned pointers and
 * lda, ldb, ldc multiples of 4. */
SIMRANK_API int simrank_gemm_nt(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                    const float* B, int64_t ldb, float* C, int64_t ldc,
                    const simrank_epilogue* epilogue, void* stream);


/* ---- block-dense part of a pattern.  When a graph is created, every aligned
 *      block of 128 rows gets a DENSE SET: the columns referenced by at least `dense_min` of its
 *      rows (kept only if there are `dense_cols` of them).  simrank_spmm multiplies those
 *      entries on the matrix cores (0/1 pattern in bf16 x the operand split into three bf16
 *      terms: exact f32 products, f32 accumulation) and gathers only the remainder; this is
 *      the "dense S.W contraction on MFMA" of the north star, applied where W really is dense
 *      (first and second `.dot` of SimRank.py:139 and their bipartite / PP forms).
 *      The statistics: blocks with a dense set, total size of the sets, entries covered. */
SIMRANK_API int simrank_graph_dense_stats(const simrank_graph* g, int64_t* n_tiles,
                                          int64_t* dense_cols, int64_t* nnz_covered);

/* ---- leg 1 as ONE launch (round 3; first `.dot` of SimRank.py:139, :298, :301, :361, :420, :423).
 *      simrank_spmm_blocked(transpose_out = 1) on a graph created with tuning "fuse" = 1 (default)
 *      runs one kernel in which a workgroup owns a 128-row block x one 32-column panel: the columns
 *      that at least "fuse_min" (default 2) rows of the block share are multiplied on the matrix
 *      cores (each operand segment loaded ONCE per block and panel instead of once per entry), the
 *      other entries are gathered, and the finished tile is stored transposed.  Statistics:
 *      16-column matrix-core steps per panel, entries on the matrix cores, entries gathered. */
SIMRANK_API int simrank_graph_fused_stats(const simrank_graph* g, int64_t* n_steps,
                                          int64_t* nnz_covered, int64_t* nnz_remainder);
/* The matrix-core part of simrank_spmm alone, into the graph's partial-sum buffer (measurement
 * harness: its HIP-event time and 2 * 3 * 128 * dense_cols * n_cols_x bf16 flop give the MFMA
 * rate).  X needs 8-byte alignment and an even ldx.  Fails when the graph has no dense sets. */
SIMRANK_API int simrank_dense_part(const simrank_graph* g, const float* X, int64_t ldx,
                                   int64_t n_cols_x, void* stream);

/* ---- PLAN: the loop of SimRank.fit / SimRankPP.fit / AprioriSimRank.fit on one GPU behind five calls
 *      (SimRank.py:124-141, :346-363, :440-455; SURVEY.md §8b create_plan / step / download).
 *      simrank_plan_create takes the normalised adjacency as CSR + per-row scale in the CALLER's node
 *      order (row = target node, SimRank.py:45-52), re-orders the nodes for speed (ascending in-degree)
 *      when options.reorder is set, builds the graph object, the three panel-blocked matrices of an
 *      update, the evidence counts of SimRank++ (options.evidence) and the prior (options.apriori: HOST
 *      n x n row-major; a prior that is not symmetric makes the iterates asymmetric (SimRank.py:453): the plan then runs
 *      leg 2 as leg 1's launch again — its product stored transposed — and the epilogue as a pass of its own with the exact
 *      count; f32 only, SIMRANK_ERR_INVALID with storage_fp16).  simrank_plan_run is the reference loop: at most
 *      `iterations` updates, stopping at loop index k when no element moved by more than eps
 *      (converged_at = k, exactly the "Converged at iteration k" of SimRank.py:132; -1 when the loop ran
 *      out); below 16384 nodes update k + 1 is queued before the count of update k is read.  simrank_plan_step is one
 *      update with its count; simrank_plan_result(_f64) hands S back in the caller's node order
 *      (device row-major f32 / host f64: SimRank.py:141).  One plan per host thread; all work goes to the
 *      stream given at creation. */
typedef struct simrank_plan simrank_plan;
typedef struct simrank_plan_options {
    float coef;                 /* C */
    float lbd;                  /* prior blend (used when apriori != NULL) */
    const float* apriori;       /* HOST n x n row-major prior (symmetric or not, see above), or NULL */
    int64_t ld_apriori;
    int32_t evidence;           /* 1: SimRank++ evidence factor 1 - 2^-|common in-neighbours| */
    int32_t reorder;            /* 1: iterate in ascending-row-length node order (recommended) */
    int32_t storage_fp16;       /* 1: the matrices are HELD in fp16 ("FP16 STORAGE" above: value x 2^14 on 64-column
                                   panels, f32 sums, one rounding per stored value) — BASELINE config 5's
                                   reduced-precision mode, outside the parity bar; 0 (default): f32 */
    int32_t dense_terms;        /* operand terms of the matrix-core part: 0 or 3 = exact f32 products (three bf16 terms);
                                   1 = one fp16 term (BASELINE config 5 read literally; outside the parity bar) */
} simrank_plan_options;
SIMRANK_API int simrank_plan_create(int64_t n, int64_t nnz, const int32_t* rowptr, const int32_t* col,
                                    const float* rowscale, const simrank_plan_options* options, void* stream,
                                    simrank_plan** out);
SIMRANK_API int simrank_plan_reset(simrank_plan* p);
SIMRANK_API int simrank_plan_step(simrank_plan* p, double eps, int32_t exact_count, int64_t* n_changed);
SIMRANK_API int simrank_plan_run(simrank_plan* p, int32_t iterations, double eps, int32_t* updates_done,
                                 int32_t* converged_at);
/* The same loop with the reference's console hooks: progress(user, k, 0) is called when loop index k goes on to an
 * update (SimRank.py:135 `update_progress(k / iterations)`), progress(user, k, 1) when the test passes at loop index k
 * (:131-133 "Converged at iteration k").  A nonzero return value of the first form ends the loop after k updates
 * (how a binding propagates an exception raised in its callback).  This is what the Python classes' fit() runs. */
typedef int32_t (*simrank_progress_fn)(void* user, int32_t k, int32_t converged);
SIMRANK_API int simrank_plan_run_cb(simrank_plan* p, int32_t iterations, double eps, simrank_progress_fn progress,
                                    void* user, int32_t* updates_done, int32_t* converged_at);
SIMRANK_API int simrank_plan_result(simrank_plan* p, float* dst, int64_t ld);
/* HOST float64 n x n in the caller's order (SimRank.py:141), through simrank_handback_f64 (pipelined band by band) */
SIMRANK_API int simrank_plan_result_f64(simrank_plan* p, double* dst, int64_t ld);
/* HOST u8 n x n in the caller's order: common in-neighbour counts, saturated at 255 (`Evidence` = 1 - 0.5 ** count,
 * SimRank.py:311-320); plans created with options.evidence only */
SIMRANK_API int simrank_plan_evidence_u8(simrank_plan* p, uint8_t* dst, int64_t ld);
/* releases the iterates, the transposed product and the prior (a finished fit keeps its plan only for
 * simrank_plan_evidence_u8); every entry point that needs them fails with SIMRANK_ERR_INVALID afterwards */
SIMRANK_API int simrank_plan_trim(simrank_plan* p);
/* the k most similar nodes of every node (HOST int32 / float [n][k], caller's ids, largest first, ties by the
 * lower id, -1 / 0 where a row has fewer; exclude_diag = 1 leaves the node itself out): 2 n k values cross
 * PCIe instead of n^2 */
SIMRANK_API int simrank_plan_topk(simrank_plan* p, int32_t k, int32_t exclude_diag, int32_t* idx_host,
                                  float* val_host);
/* some ROWS of the current similarity matrix (rows of the DataFrame of SimRank.py:141): dst[i][j] = S[rows[i]][j] in the
 * caller's node order, float32 as stored (fp16-held plans: widened), host memory, ld >= n floats per row — n_rows x n
 * values across PCIe instead of n^2 (a query for a few nodes; what the full-size parity tests sample) */
SIMRANK_API int simrank_plan_rows_f32(simrank_plan* p, const int32_t* rows, int32_t n_rows, float* dst, int64_t ld);
/* measurement: HIP events on the plan's own stream around both legs of the next `updates` updates (created at this call,
 * outside the timed region; 0 = off); simrank_plan_leg_times drains the stream and returns the mean duration of leg 1
 * (first .dot of SimRank.py:139) and of leg 2 (second .dot + :140 + :74) over the updates stamped since — what bench.py's
 * roofline.achieved is computed from */
SIMRANK_API int simrank_plan_set_timing(simrank_plan* p, int32_t updates);
SIMRANK_API int simrank_plan_leg_times(simrank_plan* p, double* leg1_ms, double* leg2_ms, int32_t* updates);
SIMRANK_API int simrank_plan_info(const simrank_plan* p, int64_t* n, int32_t* updates, const simrank_graph** graph);
SIMRANK_API int simrank_plan_destroy(simrank_plan* p);

/* ---- BIPARTITE PLAN: the loop of BipartiteSimRank.fit / BipartiteSimRankPP.fit / BipartitleAprioriSimRank.fit
 *      on one GPU (SimRank.py:288-302, :410-424, :478-492; SURVEY.md §8b create_plan(N or (n1, n2))).
 *      ONE edge set describes both patterns: the CSR of group 1 (n1 rows, columns = group-2 ids) with the
 *      per-row scales of both groups — W12 = diag(rowscale1) . A, W21 = diag(rowscale2) . A^T.  Each loop index
 *      updates S1 from S2 and then S2 from the NEW S1 (Gauss-Seidel, :300-302); the loop ends at index k when
 *      neither matrix moved by more than eps (:289).  Evidence (options.evidence): E1 from the group-1
 *      pattern, E2 from the group-2 pattern — the corrected form — unless options.strict_reference = 1, which is
 *      what the reference does (SimRank.py:420-423, :488-491): both updates are multiplied by
 *      Evidence_N1, position by position; with n1 != n2 (and n1 != 1, where NumPy broadcasts the 1 x 1 array)
 *      simrank_biplan_step / _run fail with NumPy's "operands could not be broadcast together with shapes
 *      (n1,n1) (n2,n2) " at the first group-2 update, i.e. not for iterations = 0 or eps >= 1.
 *      Priors: HOST row-major.  When either is not symmetric both iterates are asymmetric (SimRank.py:488, :491): both
 *      updates then store leg 2 transposed and run the epilogue as a pass of its own (exact counts).
 *      simrank_biplan_result_f64(group = 1 | 2) hands S1 / S2 back in the caller's node order. */
typedef struct simrank_biplan simrank_biplan;
typedef struct simrank_biplan_options {
    float c1, c2;               /* C1, C2 */
    float lbd1, lbd2;           /* prior blends (used when the prior is given) */
    const float* apriori1;      /* HOST n1 x n1 row-major prior, or NULL */
    int64_t ld_apriori1;
    const float* apriori2;      /* HOST n2 x n2, or NULL */
    int64_t ld_apriori2;
    int32_t evidence;           /* 1: SimRank++ evidence factors */
    int32_t reorder;            /* 1: iterate in ascending-row-length node order within each group (recommended) */
    int32_t strict_reference;   /* 1: Evidence_N1 on BOTH updates, as SimRank.py:423 / :491 (quirk Q2); 0: E2 on group 2 */
} simrank_biplan_options;
SIMRANK_API int simrank_biplan_create(int64_t n1, int64_t n2, int64_t nnz, const int32_t* rowptr12,
                                      const int32_t* col12, const float* rowscale1, const float* rowscale2,
                                      const simrank_biplan_options* options, void* stream, simrank_biplan** out);
SIMRANK_API int simrank_biplan_reset(simrank_biplan* p);
SIMRANK_API int simrank_biplan_step(simrank_biplan* p, double eps, int32_t exact_count, int64_t* changed1,
                                    int64_t* changed2);
SIMRANK_API int simrank_biplan_run(simrank_biplan* p, int32_t iterations, double eps, int32_t* updates_done,
                                   int32_t* converged_at);
/* with the reference's console hooks, as simrank_plan_run_cb (SimRank.py:289-296) */
SIMRANK_API int simrank_biplan_run_cb(simrank_biplan* p, int32_t iterations, double eps, simrank_progress_fn progress,
                                      void* user, int32_t* updates_done, int32_t* converged_at);
SIMRANK_API int simrank_biplan_result_f64(simrank_biplan* p, int32_t group, double* dst, int64_t ld);
/* as simrank_plan_topk / simrank_plan_evidence_u8 / simrank_plan_trim, per group (1 | 2).  The counts are those that
 * GATE the group's update: with strict_reference the group-2 counts are Evidence_N1's, position by position */
SIMRANK_API int simrank_biplan_topk(simrank_biplan* p, int32_t group, int32_t k, int32_t exclude_diag, int32_t* idx_host,
                                    float* val_host);
SIMRANK_API int simrank_biplan_rows_f32(simrank_biplan* p, int32_t group, const int32_t* rows, int32_t n_rows, float* dst,
                                        int64_t ld);
SIMRANK_API int simrank_biplan_evidence_u8(simrank_biplan* p, int32_t group, uint8_t* dst, int64_t ld);
SIMRANK_API int simrank_biplan_trim(simrank_biplan* p);
SIMRANK_API int simrank_biplan_destroy(simrank_biplan* p);

/* ---- SHARDED PLAN: the same loop with S split by COLUMN BLOCK over `world` GPUs, one process per GPU
 *      ("step(...) includes K10 when P>1"; the loop of SimRank.py:129-140, :351-362, :443-454 — the
 *      reference has no counterpart, it is one NumPy process).  Every rank holds the whole (small) graph and the
 *      N x N/P column block of S it owns; one update is
 *          leg 1   (W.S_block)^T, stored straight into the chunks of the all-to-all        [first .dot of :139]
 *          exchange 1   all-to-all of the chunks (in `stages` slices, each leaving while the next is computed)
 *          leg 2   W.(received operand) with the fused epilogue and the convergence count  [second .dot, :140, :74]
 *                  full form: every 32 x 32 tile; half form: tiles i <= j only, the transposed tiles i < j go to the
 *                  ranks that own them in a second, half-size all-to-all (exchange 2) and are put in place
 *          count   summed over the ranks on the device (all-reduce), read by every rank
 *      A COMMUNICATOR carries the exchanges: RCCL (loaded with dlopen at the first use: `librccl.so.1`, or what the
 *      environment variable SIMRANK_RCCL_LIB names; ncclSend / ncclRecv groups on a stream of their own, ordered
 *      against the kernels by events), or an IN-PROCESS GROUP of `world` virtual ranks on one device whose exchanges
 *      are device copies (tests, and the per-rank kernel times of DESIGN.md §5 on the one GPU available).
 *      The entry points that move data between ranks take ALL of this process's plans: one in a multi-process world,
 *      the `world` plans of an in-process group (which then advance in lockstep on one stream).
 *      Nodes: ascending row length, dealt to the shards in runs of 128 in the half form (as driver.dealt_order).
 *      Results: simrank_shardplan_block_f64 = the rank's columns (all n rows, caller's row order) + their node ids by
 *      simrank_shardplan_columns; simrank_shardplan_result_f64 assembles the whole matrix on rank `root`.
 *      f32, or fp16-held matrices (options.storage_fp16).  A prior that is not symmetric makes the iterates asymmetric
 *      (SimRank.py:453): leg 2's product then goes round a SECOND all-to-all (W . Tt is the transpose of the wanted block) and
 *      the epilogue runs as a pass of its own with the exact count — f32 matrices, leg2_form 0 or -1, else SIMRANK_ERR_INVALID.
 *      Lifetime: destroy the plans before their communicator.  Failure: these calls are collectives — a rank whose call
 *      fails (out of memory, a bad argument the others did not pass) leaves its peers waiting inside RCCL, as in any
 *      RCCL program; the host program owns that failure mode (validate on every rank before, tear the job down after). */
typedef struct simrank_comm simrank_comm;
#define SIMRANK_COMM_ID_BYTES 128
/* rank 0 makes an id, the host program hands its bytes to every rank (its own job: MPI, a file, a socket), every rank
 * creates its communicator from it (collective: returns when all `world` ranks have called) */
SIMRANK_API int simrank_comm_unique_id(void* id_bytes);
SIMRANK_API int simrank_comm_create(const void* id_bytes, int32_t rank, int32_t world, simrank_comm** out);
/* an RCCL communicator the host program already has (ncclComm_t; not destroyed with the handle) */
SIMRANK_API int simrank_comm_adopt(void* rccl_comm, int32_t rank, int32_t world, simrank_comm** out);
/* `world` virtual ranks inside this process, on the current device: out[0 .. world) */
SIMRANK_API int simrank_comm_local_group(int32_t world, simrank_comm** out);
/* `world` ranks inside this process, on the current device, for `world` HOST THREADS — one rank each, every thread with one
 * plan, exactly as the processes of an RCCL world (a plan per call, collectives entered by every rank): the loops take the
 * code path of an RCCL rank (exchange stream, stage events, grouped sends and receives, all-reduced count) over an in-process
 * transport with RCCL's interface whose sends and receives rendezvous between the threads and copy behind the sender's
 * event.  For tests of that code path with ranks that run at the same time where one GPU is all there is; a peer that never
 * arrives is an error after SIMRANK_THREAD_COMM_TIMEOUT seconds (default 120) on every rank, not a hang.  out[0 .. world). */
SIMRANK_API int simrank_comm_thread_group(int32_t world, simrank_comm** out);
SIMRANK_API int simrank_comm_destroy(simrank_comm* c);

typedef struct simrank_shardplan simrank_shardplan;
typedef struct simrank_shardplan_options {
    float coef;                 /* C */
    float lbd;                  /* prior blend (used when apriori != NULL) */
    const float* apriori;       /* HOST n x n row-major prior (the same on every rank; symmetric or not, see above), or NULL */
    int64_t ld_apriori;
    int32_t evidence;           /* 1: SimRank++ evidence factor */
    int32_t reorder;            /* 1: ascending-row-length node order (dealt to the shards in the half form) */
    int32_t leg2_form;          /* 0 full, 1 half (needs n % (32 world) == 0), -1: half from 8 ranks on where it applies */
    int32_t stages;             /* exchange 1 in that many overlapped stages (0: by the width of a rank's block) */
    int32_t wire_fp16;          /* 1: the exchanges move fp16 x 2^14 (half the link bytes, f32 kernels, one fp16 rounding
                                   of the transposed product per update: outside the parity bar); 0 (default): f32 */
    int32_t storage_fp16;       /* 1: S, the transposed product and the leg-2 operand HELD in fp16 on 64-column panels ("FP16
                                   STORAGE" above; BASELINE config 5 in its stated form: reduced precision AND shards): half the
                                   kernel bytes and half the link bytes; needs n % (64 world) == 0, leg 2 in its full form,
                                   no prior; outside the parity bar, the convergence index is not reference-comparable */
} simrank_shardplan_options;
SIMRANK_API int simrank_shardplan_create(int64_t n, int64_t nnz, const int32_t* rowptr, const int32_t* col,
                                         const float* rowscale, const simrank_shardplan_options* options,
                                         simrank_comm* comm, void* stream, simrank_shardplan** out);
SIMRANK_API int simrank_shardplan_reset(simrank_shardplan* const* plans, int32_t n_local);
/* one update on every rank; n_changed (may be NULL) = the count over ALL ranks */
SIMRANK_API int simrank_shardplan_step(simrank_shardplan* const* plans, int32_t n_local, double eps, int32_t exact_count,
                                       int64_t* n_changed);
/* the reference loop (while a rank's update is short, update k + 1 is queued before the count of update k is read); every
 * rank returns the same numbers */
SIMRANK_API int simrank_shardplan_run(simrank_shardplan* const* plans, int32_t n_local, int32_t iterations, double eps,
                                      int32_t* updates_done, int32_t* converged_at);
/* dst[i][j] (HOST, n x n_cols_of_the_rank) = S[caller's node i][the rank's column j]; ids[j] = caller's id of column j */
SIMRANK_API int simrank_shardplan_block_f64(simrank_shardplan* p, double* dst, int64_t ld);
SIMRANK_API int simrank_shardplan_columns(const simrank_shardplan* p, int32_t* ids);
/* the whole matrix in the caller's order on rank `root` (dst ignored elsewhere); collective */
SIMRANK_API int simrank_shardplan_result_f64(simrank_shardplan* const* plans, int32_t n_local, int32_t root, double* dst,
                                             int64_t ld);
/* the k most similar nodes of every node on rank `root` (HOST int32 / float [n][k], caller's ids, largest first, ties by the
 * lower id, -1 / 0 where a row has fewer): every rank selects among its own columns on the device, n k values per rank cross
 * the links and PCIe instead of n^2 — the hand-back config 5 is meant for (its dense result is 34 GB of float64); collective */
SIMRANK_API int simrank_shardplan_topk(simrank_shardplan* const* plans, int32_t n_local, int32_t root, int32_t k,
                                       int32_t exclude_diag, int32_t* idx_host, float* val_host);
/* measurement: HIP events at the boundaries of the next `updates` updates of this rank's plan (0 = off), each on the stream its
 * piece runs on; simrank_shardplan_timings drains both streams and returns mean milliseconds per update: ms[0] leg 1 (the
 * stages' kernels), ms[1] exchange 1 (its stages, on the exchange stream: what RCCL took), ms[2] what the kernels' stream
 * waited between the last stage's kernel and leg 2 (the part of exchange 1 leg 1 did not hide), ms[3] leg 2, ms[4] the
 * all-reduce of the count + exchange 2, ms[5] the whole update on the kernels' stream */
SIMRANK_API int simrank_shardplan_set_timing(simrank_shardplan* p, int32_t updates);
SIMRANK_API int simrank_shardplan_timings(simrank_shardplan* p, double* ms, int32_t n_ms, int32_t* updates);
SIMRANK_API int simrank_shardplan_info(const simrank_shardplan* p, int64_t* n, int64_t* col_lo, int64_t* col_hi,
                                       int32_t* half_form, int32_t* stages, int32_t* updates);
SIMRANK_API int simrank_shardplan_destroy(simrank_shardplan* p);

/* ---- SHARDED BIPARTITE PLAN: the loops of BipartiteSimRank.fit / BipartiteSimRankPP.fit / BipartitleAprioriSimRank.fit
 *      (SimRank.py:288-302, :410-424, :478-492) with S1 and S2 each split by column block over the ranks of a communicator:
 *      a pair of sharded plans, one per group, each reading the other group's blocks in its leg 1 — two exchanges per loop
 *      body in strict order (the group-2 update consumes the NEW S1, :300-302), the loop ends when neither matrix moved
 *      (:289).  `options` as for simrank_biplan_create (evidence in its corrected form unless strict_reference: quirk Q2,
 *      incl. NumPy's broadcast error at the first group-2 update); leg2_form / stages / wire_fp16 as in
 *      simrank_shardplan_options, the half form taken group by group where that group's size allows it.  A prior that
 *      is not symmetric (either group) makes both iterates asymmetric: f32, leg 2 in its full form, its product sent
 *      round a second all-to-all (as simrank_shardplan_create).  simrank_shardbiplan_side hands out group 1 | 2's plan for the hand-back entry points of the
 *      single-matrix plan (simrank_shardplan_result_f64 / _block_f64 / _columns / _topk / _info / _set_timing) — it
 *      stays owned by the pair. */
typedef struct simrank_shardbiplan simrank_shardbiplan;
SIMRANK_API int simrank_shardbiplan_create(int64_t n1, int64_t n2, int64_t nnz, const int32_t* rowptr12, const int32_t* col12,
                                           const float* rowscale1, const float* rowscale2,
                                           const simrank_biplan_options* options, int32_t leg2_form, int32_t stages,
                                           int32_t wire_fp16, simrank_comm* comm, void* stream, simrank_shardbiplan** out);
SIMRANK_API int simrank_shardbiplan_side(simrank_shardbiplan* bp, int32_t group, simrank_shardplan** out);
SIMRANK_API int simrank_shardbiplan_reset(simrank_shardbiplan* const* plans, int32_t n_local);
SIMRANK_API int simrank_shardbiplan_step(simrank_shardbiplan* const* plans, int32_t n_local, double eps, int32_t exact_count,
                                         int64_t* changed1, int64_t* changed2);
SIMRANK_API int simrank_shardbiplan_run(simrank_shardbiplan* const* plans, int32_t n_local, int32_t iterations, double eps,
                                        int32_t* updates_done, int32_t* converged_at);
SIMRANK_API int simrank_shardbiplan_destroy(simrank_shardbiplan* bp);

/* ---- tuning knobs (measurement harness; defaults are the tuned values).  simrank_set_tuning
 *      changes the process-wide DEFAULTS; simrank_graph_create copies them into the graph it
 *      builds (under a lock), and every launch on that graph uses its copy — a knob set later does
 *      not reach graphs that already exist.
 *      "panel"    columns per gather panel (16, 32, 64, 128, 256; 0 = automatic = 32)
 *      "tile"     rows per wave tile (16, 32, 64; 0 = automatic = 32)
 *      "lean"     0/1  the lean gather kernel (32-column panels, 32-row tiles) wherever it applies;
 *                 0 = the generic kernel (row-major operands only)
 *      "xcd_map"  0/1  panel -> XCD affinity
 *      "triangle" 0/1  allow the upper-triangle + mirror form when epilogue.symmetric
 *      "stream_nt" 0/1 non-temporal access for streamed-once data
 *      "huge"     rows of at least this many entries are split over a workgroup's waves
 *      "balance"  32-row tiles heavier than this many times the mean tile are cut in halves
 *                 (0 = uniform tiles; default 2)
 *      "dense_min" / "dense_cols"  selection of the block-dense part (see above); dense_min 0 = off
 *      "fuse" / "fuse_min" / "fuse_pays" / "fuse_steps"  leg 1 of a panel-blocked update as one launch (see above):
 *                 fuse_min 0 (default): a 128-row block's columns in descending count join its dense set 64 at a
 *                 time (one quad = four 16-column steps) while the quad covers fuse_pays entries (default -1: 192,
 *                 or 256 where the plan's matrix-core steps outweigh its gathered remainder); fuse_min >= 2: every
 *                 column that many of the block's rows reference; a block keeps its set when it makes fuse_steps
 *                 16-column steps (default -1: 20 while a panel's operand slice fits the XCD's L2, i.e. up to
 *                 32768 operand rows, 8 beyond); fuse 0 = the dense_tiles + gather launches of round 2
 *      "fuse_sym"  leg 2 of a symmetric panel-blocked update in the same launch (epilogue on the tile, upper
 *                 triangle + mirror): 1 wherever it applies, 0 never, -1 (default) where the dense sets hold at
 *                 least half of the pattern's entries
 *      "fuse_max_rows" operands with more rows than this (default 2^20: never) keep the two-launch leg 1
 *      "fuse_group" up to this many (1..4, default 3) consecutive blocks without a set share a workgroup
 *      "fuse_unit"  sets of more 64-column groups than this (default 48) are cut into several workgroups whose
 *                 partial sums meet in memory (1 << 20: never)
 *      "fuse_order" launch order of a panel's workgroups: 0 heaviest first, k: matrix-core units spread
 *                 over the first 1/k of the order
 *      "fuse_store" cache policy of the tile stores (0 plain, 1 nt, 2 sc1, 3 sc0 sc1); "fuse_meta_nt"
 *      "dense_terms" operand terms of the block-dense part: 3 = bf16 hi+mid+lo (exact f32
 *                 products, default), 1 = one fp16 term (reduced precision, BASELINE config 5)
 *      "ids16"    0/1  stream the neighbour ids as 16-bit values (graphs with <= 65536 columns)
 *      "dense_sym" dense part in the upper-triangle form of leg 2: 1 always, 0 never, -1 when
 *                 the dense sets hold at least half of the entries
 *      "sym_desc" 0/1  upper-triangle leg 2: an XCD takes its panels in descending order (default 1)
 *      "addr32"   0/1  32-bit buffer addressing of the gather operand where it spans < 2 GiB (default 1)
 *      "probe_mask", "probe_flags"  DIAGNOSTIC ONLY (wrong results; refused unless the environment
 *                 variable SIMRANK_ENABLE_PROBES is set): price parts of the gather
 *                 kernel — ids ANDed with a mask; 1 no gathers, 2 no stores, 4 no dense partial
 *                 sums, 8 no id loads, 16 one XCD's share of the launch only ---- */
SIMRANK_API int simrank_set_tuning(const char* key, int64_t value);
SIMRANK_API int simrank_get_tuning(const char* key, int64_t* value);

#ifdef __cplusplus
}
#endif
#endif /* SIMRANK_HIP_H */
